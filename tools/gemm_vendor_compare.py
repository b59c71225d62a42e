#!/usr/bin/env python3
"""How far is the hand-written 16-bit GEMM (csrc/bf16.hip) from the vendor library on the shapes the engines run?  torch.matmul
(hipBLASLt / rocBLAS) against ops.gemm_bf16 with the PLAIN bias epilogue, bf16, medians of interleaved rounds (GPU box).  A measurement
only: nothing in the product calls the vendor GEMM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import ops

def timed(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [("bert qkv", 32768, 2304, 768), ("bert out", 32768, 768, 768), ("bert fc1", 32768, 3072, 768), ("bert fc2", 32768, 768, 3072),
          ("bert packed fc1", 18831, 3072, 768), ("dec qkv", 65536, 1536, 512), ("dec out", 65536, 512, 512), ("dec fc1", 65536, 2048, 512),
          ("dec fc2", 65536, 512, 2048), ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    wt = w.t()
    res = {"mgea": [], "vendor": [], "vendor+bias": []}
    for rep in range(5):
        res["mgea"].append(timed(lambda: ops.gemm_bf16(a, w, b, out=out)))
        res["vendor"].append(timed(lambda: torch.matmul(a, wt, out=out)))
        res["vendor+bias"].append(timed(lambda: torch.addmm(b.bfloat16(), a, wt, out=out)))
    fl = 2 * M * N * K
    line = f"{name:16s} M={M:6d} N={N:5d} K={K:5d}:"
    for k, v in res.items():
        us = sorted(v)[len(v) // 2]
        line += f"  {k} {us:7.1f} us = {fl / us / 1e6:6.0f} TF ({fl / us / 1e6 / 2500:.3f})"
    print(line, flush=True)

# ---- exact-fp32: ops.gemm (v_mfma_f32_16x16x4_f32, csrc/gemm_f32.hip) against torch fp32 matmul (TF32-style shortcuts off: the library's
# own fp32 path), on the f32 engines' big-M shapes
torch.backends.cuda.matmul.allow_tf32 = False
for name, M, N, K in [("dec qkv f32", 65536, 1536, 512), ("dec fc1 f32", 65536, 2048, 512), ("dec fc2 f32", 65536, 512, 2048),
                      ("dec head f32", 65536, 8324, 512), ("bert fc1 f32", 32768, 3072, 768), ("bert fc2 f32", 32768, 768, 3072)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    wt = w.t()
    res = {"mgea": [], "vendor+bias": []}
    for rep in range(3):
        res["mgea"].append(timed(lambda: ops.gemm(a, w, b), 3))
        res["vendor+bias"].append(timed(lambda: torch.addmm(b, a, wt, out=out), 3))
    fl = 2 * M * N * K
    line = f"{name:16s} M={M:6d} N={N:5d} K={K:5d}:"
    for k, v in res.items():
        us = sorted(v)[len(v) // 2]
        line += f"  {k} {us:8.1f} us = {fl / us / 1e6:6.1f} TF ({fl / us / 1e6 / 157.3:.3f} of the 157.3 TF fp32 matrix peak)"
    print(line, flush=True)
