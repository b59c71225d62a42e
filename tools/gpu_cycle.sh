#!/bin/bash
# One GPU-box cycle: parity tests, a short bench line and the rocprofv3 kernel stats of the bench command.
# usage (from the repo root, through gpurun): bash tools/gpu_cycle.sh <tag> [pytest-selector]
set -o pipefail
TAG=${1:-cycle}
SEL=${2:-tests}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG /tmp/prof_$TAG
timeout -k 10 600 python -m pytest $SEL -q -m gpu -x > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/$TAG/tests.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-bert > gpurun_out/$TAG/bench_prof.json 2> gpurun_out/$TAG/bench_prof.err
echo "prof rc=$?"
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f gpurun_out/$TAG/kernel_stats.csv; done
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$TAG/kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
python3 -c "
import json;d=json.load(open('gpurun_out/$TAG/bench_prof.json'));print('tok/s',round(d['value']),'ms/gen',round(d['ms_per_step'],1),'nodes',d['graph']['graph_nodes'],'attn GB/s',d['roofline'] and round(d['roofline']['achieved']), d['roofline'] and d['roofline']['step_breakdown_ms'])"
