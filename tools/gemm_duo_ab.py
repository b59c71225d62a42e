#!/usr/bin/env python3
"""In-process A/B: the persistent 256 x 256 kernel (one 8-wave workgroup per CU) against the two-workgroups-per-CU kernel (256 x 128 per
4-wave workgroup; switch bf16_gemm_tile = 5) on the DistilBERT shapes with the plain epilogues; outputs compared bitwise (the MFMA sequence
over K is the same).  python3 tools/gemm_duo_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops

old = _lib.tune_get("bf16_gemm_tile")
for name, M, N, K, epi in [("qkv0", 32768, 2304, 768, 0), ("fc1g", 32768, 3072, 768, 1), ("out0", 32768, 768, 768, 0), ("fc2_0", 32768, 768, 3072, 0),
                           ("rag", 5000, 2304, 192, 0), ("sq4k", 4096, 4096, 4096, 0), ("sq8k", 8192, 8192, 8192, 0),
                           # Decoder-S prefill [64, 1024] (fp16 there; the bf16 kernels run at the same rate): K = 512 is 8 K-tiles per tile
                           ("dSqkv", 65536, 1536, 512, 0), ("dSfc1", 65536, 2048, 512, 1), ("dSout", 65536, 512, 512, 0), ("dSfc2", 65536, 512, 2048, 0)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    kw = dict(gelu=epi == 1, out=torch.empty(M, N, dtype=torch.bfloat16, device="cuda"))
    res, outs = {}, {}
    for rep in range(5):
        for tile in (0, 5):
            _lib.tune_set("bf16_gemm_tile", tile)
            f = lambda: ops.gemm_bf16(a, w, b, **kw)
            for _ in range(2): f()
            torch.cuda.synchronize()
            if rep == 0: outs[tile] = kw["out"].clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tile, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    same = torch.equal(outs[0], outs[5])
    ref = (a[:256].float() @ w.float().T + b)
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    err = (outs[5][:256].float() - ref).abs().max().item()
    for tile, v in res.items():
        us = sorted(v)[len(v) // 2]
        print(f"{name:5s} M={M:5d} N={N:5d} K={K:5d} epi {epi} {'duo 256x128 x2' if tile else 'one 256x256   '}: median {us:7.1f} us (min {min(v):7.1f})  "
              f"{2 * M * N * K / us / 1e6:7.1f} TFLOP/s" + (f"   bitwise equal: {same}, max |err| vs fp32 {err:.4f}" if tile else ""), flush=True)
_lib.tune_set("bf16_gemm_tile", old)
