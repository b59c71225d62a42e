#!/usr/bin/env python3
"""Time the persistent bf16 GEMM on the DistilBERT shapes at M = 32768 (GPU box), every epilogue the engine uses, with the
three tail schedules interleaved in one process (switch bf16_gemm_tail: 0 whole tiles, 1 half-tile tail units, 2 = 1 with the odd
slots running their half unit first); medians of 5 rounds of 10 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops

M = 32768
tails = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
for name, N, K, epi in [("qkv", 2304, 768, 3), ("out", 768, 768, 5), ("fc1", 3072, 768, 4), ("fc2", 768, 3072, 5), ("qkv0", 2304, 768, 0)]:
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
    res = {}
    for rep in range(5):                      # the schedules alternate: clocks drift by a few % over a run
        for tail in tails:
            _lib.tune_set("bf16_gemm_tail", tail)
            f = lambda: ops.gemm_bf16(a, w, b, **kw)
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tail, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    for tail, v in res.items():
        us = sorted(v)[len(v) // 2]
        print(f"{name:4s} N={N:5d} K={K:5d} epi {epi} tail {tail}: median {us:7.1f} us (min {min(v):7.1f})  "
              f"{2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
_lib.tune_set("bf16_gemm_tail", 1)
