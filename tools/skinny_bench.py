#!/usr/bin/env python3
"""Micro-benchmark / ablation of the fused skinny GEMM on the decode-step shapes (runs on the GPU box).
Times back-to-back launches from a captured hipGraph-free loop; the ~8 us floor is host launch cost,
so read DIFFERENCES between variants, not absolutes.  (dbg >> 8) & 15 forces the row-tile count MT,
(dbg >> 12) & 31 the waves per workgroup; tools/skinny_phases.py gives the in-kernel phase timing."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
from mgea._lib import ptr, check, stream_ptr

lib = _lib.load()
dev = "cuda:0"


def run(name, epi, N, K, ln, dbg, iters=400):
    M = 64
    a = torch.randn(M, K, device=dev); w = ops.tile_weights(torch.randn(N, K, device=dev) * K ** -0.5)
    bias = torch.randn(N, device=dev); lnw = torch.ones(N, device=dev)   # stands in for the folded-LN c1 vector
    stats = torch.zeros(M, 32, 2, device=dev); stats[:, :, 1] = 16.0
    out = torch.zeros(M, N, device=dev); so = torch.zeros(M, N // 16, 2, device=dev)

    def go():
        check(lib.mgea_op_skinny(epi, ptr(a), ptr(w), ptr(bias), ptr(lnw) if ln else None,
                                 ptr(stats), 32, 16, ptr(out), ptr(so), M, N, K, 1 if epi == 2 else 0, dbg, stream_ptr()))
    # capture a chain of launches in a graph so the host launch cost does not hide the kernel time
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5): go()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50): go()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(iters // 50): g.replay()
        e1.record(s); s.synchronize()
    print(f"{name:22s} N={N:5d} K={K:5d} ln={int(ln)} mt={(dbg >> 8) & 15} dbg={dbg & 255}: "
          f"{e0.elapsed_time(e1) / (iters // 50 * 50) * 1e3:7.2f} us/launch", flush=True)


SHAPES = (("qkv", 2, 1536, 512, True), ("fc1", 2, 2048, 512, True), ("out-proj", 1, 512, 512, False), ("fc2", 1, 512, 2048, False))
if len(sys.argv) > 1 and sys.argv[1] == "default":   # the launcher's own choice only (e.g. under rocprofv3 --kernel-trace --stats)
    for name, epi, N, K, ln in SHAPES:
        run(name, epi, N, K, ln, 0)
else:   # sweep of row tiles (mt) and waves (nw) on the decode-step shapes
    for name, epi, N, K, ln in SHAPES:
        for mt in (1, 2):
            for nw in (4, 8):
                run(f"{name} mt={mt} nw={nw}", epi, N, K, ln, (mt << 8) | (nw << 12))
