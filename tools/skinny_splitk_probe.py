#!/usr/bin/env python3
"""Would the N = 512 decode GEMMs (out-proj, FC2: 128 workgroups of 16 x 16 tiles, half the chip) gain from a 2-way split of K over
256 workgroups with the partial sums added by the consumer's operand load?  Upper bound without building it: the same residual
kernel at (N, K) = (1024, K / 2) moves the same weight bytes with 256 workgroups and half the K per workgroup, and writes two
[64, 512] images.  Each shape in a graph-replayed chain of itself (24 kernels, 6 weight sets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
from mgea._lib import ptr, check, stream_ptr
lib = _lib.load(); dev = "cuda:0"
M = 64
def run(N, K):
    ws = [ops.tile_weights(torch.randn(N, K, device=dev) * K ** -0.5) for _ in range(6)]
    bias = torch.randn(N, device=dev)
    a = torch.randn(M, max(K, N), device=dev)            # A and the residual / output share one buffer as in the step (x -> x)
    out = torch.zeros(M, max(K, N), device=dev)
    stats = torch.zeros(M, N // 16, 2, device=dev); stats[:, :, 1] = 16.0
    def step():
        for n in range(24):
            check(lib.mgea_op_skinny(1, ptr(a), ptr(ws[n % 6]), ptr(bias), None, ptr(stats), 32, 16, ptr(out), ptr(stats), M, N, K, 0, 0, stream_ptr()))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(3): g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(200): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3 / 24
for name, N, K in (("out-proj", 512, 512), ("out-proj split 2", 1024, 256), ("FC2", 512, 2048), ("FC2 split 2", 1024, 1024)):
    print(f"{name:18s} N={N:5d} K={K:5d}: {run(N, K):5.2f} us per kernel", flush=True)
