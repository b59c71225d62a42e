"""bench.py's own launch path at world size 2 over gloo, on CPU (`python bench.py --gpus N` used to run ONE
rank and print n_gpus 1).  `--dry-run` keeps everything of the N-rank flow except the GPU work: the parent starts
N children through torch.distributed.run, the ranks rendezvous on 127.0.0.1, broadcast an arena, barrier,
max-reduce and rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=600)


def last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {stdout[-400:]}"
    return json.loads(lines[0])


def test_gpus2_without_a_launcher_starts_two_ranks():
    r = run(["--gpus", "2", "--dry-run"], MGEA_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-800:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["backend"] == "gloo"
    assert line["per_rank_tokens_per_sec"] == [1000.0, 1001.0]          # every rank's own figure, in rank order
    assert "gloo weight broadcast" in line["config"]["parallelism"] and "RCCL" not in line["config"]["parallelism"]


def test_gpus8_dry_run_is_the_shape_of_baseline_config3():
    """BASELINE configs[3] (B = 512 prompts over 8 GPUs) as far as this container can take it: `--gpus 8 --dry-run`
    over gloo = 8 CPU ranks through bench.py's own launch path -- rendezvous on 127.0.0.1, ONE arena broadcast that every rank checks,
    barrier, max-reduce, the 8 per-rank figures in rank order, global_batch 512 in the line, and every rank's prompt slice
    (mgea.dist.shard_rows: 64 contiguous rows each, no overlap, no gap).  No scaling number can be measured here."""
    r = run(["--gpus", "8", "--dry-run"], MGEA_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    assert r.returncode == 0, r.stderr[-800:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 8 and line["dry_run"] is True and line["backend"] == "gloo"
    assert line["per_rank_tokens_per_sec"] == [1000.0 + i for i in range(8)]
    assert line["config"]["global_batch"] == 512
    assert line["rows_per_rank"] == [64] * 8 and line["first_row_per_rank"] == [64 * i for i in range(8)]
    assert "dp8 replicas, one gloo weight broadcast" in line["config"]["parallelism"]


def test_shard_rows_covers_512_prompts_over_8_ranks_and_ragged_totals():
    sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
    from mgea.dist import shard_rows
    got = [shard_rows(512, r, 8) for r in range(8)]
    assert [len(g) for g in got] == [64] * 8 and [g.start for g in got] == list(range(0, 512, 64))
    for n, w in ((513, 8), (7, 8), (100, 3), (1, 2)):
        rows = [i for r in range(w) for i in shard_rows(n, r, w)]
        assert rows == list(range(n))                                   # every prompt exactly once, in order
        sizes = [len(shard_rows(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_rccl_job_on_too_few_gpus_fails_loudly_instead_of_running_one_rank():
    r = run(["--gpus", "2"])                      # backend nccl (default); this container has no GPU at all
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = run(["--gpus", "2", "--dry-run"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MGEA_DIST_BACKEND="gloo")
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_launcher_command_is_the_drivers():
    sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
    from mgea import launch
    cmd = launch.rank_command("bench.py", ["--gpus", "4"], 4, port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                       "--master-port", "29511", "bench.py", "--gpus", "4"]
    with pytest.raises(ValueError):
        launch.spawn_ranks("bench.py", [], 1)


def test_algorithmic_bytes_match_the_survey_figures():
    """Decoder-S, B = 64, 1019 steps = 921.5 GB in fp32 (14.13 MB per token) / 460.8 GB at 2 bytes; the
    whole-step HBM fraction of the bench line is computed from exactly this formula."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    f32 = b.whole_step_bytes(b.DEC, 64, 5, 1019, 4, 4)
    f16 = b.whole_step_bytes(b.DEC, 64, 5, 1019, 2, 2)
    assert abs(f32 / 1e9 - 921.5) < 0.5 and abs(f16 / 1e9 - 460.8) < 0.3
    assert abs(f32 / (64 * 1019) / 1e6 - 14.13) < 0.01
    C, NL, V = b.DEC["d_model"], b.DEC["n_layer"], b.DEC["vocab"]
    assert NL * (12 * C * C + 13 * C) + V * C + V == 23_184_516          # P_step of SURVEY §8d


def test_gpu_count_comes_from_sysfs_not_from_hip(tmp_path, monkeypatch):
    """The self-launching parent counts GPUs from the KFD topology (nodes with SIMDs), capped by *_VISIBLE_DEVICES; it never calls
    torch.cuda / HIP (a parent that has initialised the GPU must not start GPU children on this pool)."""
    sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
    from mgea import launch
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):                 # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert launch.count_gpus(str(tmp_path), env={}) == 3
    assert launch.count_gpus(str(tmp_path), env={"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert launch.count_gpus(str(tmp_path), env={"ROCR_VISIBLE_DEVICES": ""}) == 0
    assert launch.count_gpus(str(tmp_path / "missing"), env={}) is None
    src = open(os.path.join(ROOT, "music-generation-emotion-adaptive_amd", "mgea", "launch.py")).read()
    assert "import torch" not in src
    bench_src = open(BENCH).read()
    parent = bench_src[bench_src.index("if args.gpus > 1 and not launch.launched_by_a_launcher()"):bench_src.index("if args.dry_run:\n        return dry_run")]
    assert "torch.cuda" not in parent and "count_gpus()" in parent
