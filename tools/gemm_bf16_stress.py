#!/usr/bin/env python3
"""Race screen for the persistent phase-interleaved bf16 GEMM (hand-counted vmcnt, LDS-DMA hidden from the compiler, two wave groups
one barrier apart): the same product N times, every output compared BITWISE with the first run and once against fp64 math.
A DMA / ds_read race shows up as rare wrong tiles, never as a rounding-sized error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
_lib.tune_set("bf16_gemm_tile", 4)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
noise = len(sys.argv) > 2 and sys.argv[2] == "noise"   # a second stream copies 512 MB buffers meanwhile: uneven memory load shifts DMA timing
side = torch.cuda.Stream()
nbuf = [torch.empty(128 << 20, dtype=torch.float32, device="cuda") for _ in range(2)] if noise else None
bad = 0
for (M, N, K, mode) in [(32768, 2304, 768, "bias"), (32768, 768, 3072, "res"), (32768, 3072, 768, "gelu"), (1300, 3072, 768, "gelu"), (768, 768, 3072, "res"),
                        (512, 256, 128, "bias"), (5000, 1024, 1536, "res")]:
    g = torch.Generator().manual_seed(M + N + K)
    a = ((torch.rand(M, K, generator=g) * 2 - 1)).bfloat16().cuda()
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * K ** -0.5).bfloat16().cuda()
    b = (torch.rand(N, generator=g) * 2 - 1).cuda()
    r = ((torch.rand(M, N, generator=g) * 2 - 1)).bfloat16().cuda() if mode == "res" else None
    first = ops.gemm_bf16(a, w, b, r, gelu=(mode == "gelu"))
    sub = slice(0, min(M, 2048))
    want = a[sub].double() @ w.double().t() + b.double()
    if mode == "gelu": want = torch.nn.functional.gelu(want)
    if mode == "res": want = want + r[sub].double()
    err = float(((first[sub].double() - want).abs() / (want.abs() + 1.0)).max())
    mism = 0
    for i in range(reps):
        if noise and i % 3 == 0:
            with torch.cuda.stream(side):
                nbuf[1].copy_(nbuf[0])
        out = ops.gemm_bf16(a, w, b, r, gelu=(mode == "gelu"))
        if not torch.equal(out, first):
            mism += 1
    bad += mism + (err >= 6e-3)
    print(f"M={M:6d} N={N:5d} K={K:5d} {mode:5s}: rel err vs fp64 {err:.2e}, {mism} of {reps} repetitions differ bitwise from the first", flush=True)
print("RACE SCREEN", "(with a concurrent copy stream)" if noise else "", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
