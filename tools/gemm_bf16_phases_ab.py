#!/usr/bin/env python3
"""In-process A/B of the whole-tile K-loop schedules of the persistent bf16 GEMM (switch bf16_gemm_phases: 4 / 2 phases per K-tile with
the waves of a SIMD in opposite roles, 1 = software-pipelined with one barrier per K-tile) on the DistilBERT shapes at M = 32768 and on
two long-K shapes; schedules interleaved, medians of 5 rounds of 10 launches.   python3 tools/gemm_bf16_phases_ab.py [4,2,1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops

phases = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 2, 1]
old = _lib.tune_get("bf16_gemm_phases")
for name, M, N, K, epi in [("qkv", 32768, 2304, 768, 3), ("out", 32768, 768, 768, 5), ("fc1", 32768, 3072, 768, 4), ("fc2", 32768, 768, 3072, 5),
                           ("qkv0", 32768, 2304, 768, 0), ("sq4k", 4096, 4096, 4096, 0), ("sq8k", 8192, 8192, 8192, 0)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16()
    st = torch.stack([torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5], 1).contiguous()
    kw = {}
    if epi in (3, 4):
        kw = dict(ln=dict(rowstat=st, c1=w.float().sum(1)), gelu=epi == 4)
    elif epi == 5:
        kw = dict(res=r, ln=dict(rowstat=st, g=torch.ones(N, device="cuda"), b=torch.zeros(N, device="cuda"),
                                 stats=torch.zeros(M, N // 256, 2, device="cuda")))
    kw["out"] = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res, outs = {}, {}
    for rep in range(5):
        for ph in phases:
            _lib.tune_set("bf16_gemm_phases", ph)
            f = lambda: ops.gemm_bf16(a, w, b, **kw)
            for _ in range(2): f()
            torch.cuda.synchronize()
            if rep == 0: outs[ph] = kw["out"].clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(ph, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    same = all(torch.equal(outs[phases[0]], outs[p]) for p in phases)
    for ph, v in res.items():
        us = sorted(v)[len(v) // 2]
        print(f"{name:4s} M={M:5d} N={N:5d} K={K:5d} epi {epi} phases {ph}: median {us:7.1f} us (min {min(v):7.1f})  "
              f"{2 * M * N * K / us / 1e6:7.1f} TFLOP/s" + ("" if same else "   OUTPUTS DIFFER between schedules"), flush=True)
_lib.tune_set("bf16_gemm_phases", old)
