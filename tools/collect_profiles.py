#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left in gpurun_out/prof_<tag>/ into profiles/<tag>_* (the committed evidence):  collect_profiles.py r4"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
S, D = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
same = ["bench_kernel_stats.csv", "bench_under_rocprof.json", "f16_kernel_stats.csv", "bert_bf16_kernel_stats.csv", "bert_bf16_packed_kernel_stats.csv",
        "prefill_f16_kernel_stats.csv", "prefill_f32_kernel_stats.csv", "pmc_fetch_size_attn.json", "pmc_fetch_size_attn_graph_replay.json",
        "pmc_mfma_util.json", "pmc_mfma_util_prefill_f32.json", "b1_kernel_stats.csv"]
renamed = {"rc.txt": "profile_round_rc.txt", "sampler_bench.out": "sampler_bench.txt", "head_phases.out": "head_phases.txt",
           "bert_packed_ab.out": "bert_packed_ab.txt", "attn16_pipe_ab.out": "attn16_pipe_ab.txt", "prefill16_pages_ab.out": "prefill16_pages_ab.txt"}
for f in same:
    if os.path.exists(os.path.join(S, f)): shutil.copy(os.path.join(S, f), os.path.join(D, f"{tag}_{f}"))
for f, g in renamed.items():
    if os.path.exists(os.path.join(S, f)): shutil.copy(os.path.join(S, f), os.path.join(D, f"{tag}_{g}"))
with open(os.path.join(D, f"{tag}_step_ab.txt"), "w") as o:
    for f in ("step_ab_head.out", "step_ab_one_per_cu.out"):
        if os.path.exists(os.path.join(S, f)): o.write(open(os.path.join(S, f)).read())
# FETCH_SIZE of the skinny GEMMs and the head: the two passes of tools/pmc_skinny.sh in one file, with the algorithmic bytes beside them
parts = [os.path.join(S, f) for f in ("pmc_fetch_size_skinny_nch2.json", "pmc_fetch_size_head.json")]
if all(os.path.exists(p) for p in parts):
    a, b = (json.load(open(p)) for p in parts)
    out = {k: v for k, v in list(a.items()) + list(b.items()) if not k.startswith("_")}
    alg = {"gemm_skinny_kernel<0, true, 2, 1, false, 2>": ("QKV 512 -> 1536", 1536 * 512 * 4 + 64 * 512 * 4),
           "gemm_skinny_kernel<2, true, 2, 1, false, 2>": ("FC1 512 -> 2048", 2048 * 512 * 4 + 64 * 512 * 4),
           "gemm_skinny_kernel<1, false, 1, 1, false, 2>": ("out-proj 512 -> 512 (+ residual)", 512 * 512 * 4 + 2 * 64 * 512 * 4),
           "head_balanced_kernel": ("LM head 512 -> 8324", 8324 * 512 * 4 + 64 * 512 * 4)}
    for k, v in out.items():
        for pat, (name, nbytes) in alg.items():
            if pat in k:
                v.update(what=name, bytes_fetched_per_launch=v["mean"] * 2048, algorithmic_bytes_per_launch=nbytes,
                         fetched_over_algorithmic=round(v["mean"] * 2048 / nbytes, 3))
    out.update(_commit=a.get("_commit"), _counter="FETCH_SIZE (x 1024 x 2 bytes, MI355X_MICROARCH.md)", _commands=[a.get("_command"), b.get("_command")],
               _note="two passes (tools/pmc_skinny.sh); gemm_skinny_kernel<1, false, 1, 1, false, 8> (FC2) is left out: rocprofv3 --pmc dies inside its "
                     "interception of that launch (r4_pmc_fetch_size_skinny_profiler_crash.err)")
    json.dump(out, open(os.path.join(D, f"{tag}_pmc_fetch_size_skinny.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict): print(k[:64], v.get("what"), v.get("fetched_over_algorithmic"))
