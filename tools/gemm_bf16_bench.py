#!/usr/bin/env python3
"""Time the bf16 GEMM on the DistilBERT shapes (GPU box).  MGEA_BF16_GEMM_DBG=1 no loads after the
first tile, =2 no MFMAs (loads only).  Each shape is timed without and with the split-tail scratch (include/mgea.h)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import ops

M = 32768
for name, N, K, mode in [("qkv", 2304, 768, "bias"), ("out", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "res")]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").bfloat16()
    scratch = ops.GemmScratch()
    res = {}
    for rep in range(5):                      # the two schedules alternate: clocks drift by a few % over a run
        for sc in (None, scratch):
            f = lambda: ops.gemm_bf16(a, w, b, r if mode == "res" else None, gelu=(mode == "gelu"), scratch=sc)
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(sc is not None, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    for split, v in res.items():
        us = sorted(v)[len(v) // 2]
        print(f"{name:4s} N={N:5d} K={K:5d} {'split tail ' if split else 'whole tiles'}: median {us:7.1f} us (min {min(v):7.1f})  "
              f"{2 * M * N * K / us / 1e6:7.1f} TFLOP/s  dbg={os.environ.get('MGEA_BF16_GEMM_DBG', '0')}", flush=True)
