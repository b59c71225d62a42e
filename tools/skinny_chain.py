#!/usr/bin/env python3
"""The GEMM chain of a decode step WITHOUT the attention kernels, from the op-level entry points: per layer
QKV-like (LayerNorm-folded, N = 1536) -> out-proj (residual) -> FC1 (LayerNorm-folded, GELU) -> FC2 (residual), 6 layers with their own
weights, every kernel reading what its predecessor wrote, captured in one hipGraph.  Compares the time per kernel with the same four
kernels timed each in a chain of itself (tools/skinny_bench.py): is a kernel slower behind DIFFERENT kernels?
usage: skinny_chain.py [same | pattern QOFG...]   ("same": all layers share one set of weights; pattern: which of Q = QKV-like,
O = out-proj, F = FC1, G = FC2 to chain, repeated to 24 kernels, e.g. "pattern QO")"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
from mgea._lib import ptr, check, stream_ptr
lib = _lib.load(); dev = "cuda:0"
M, C, F, NL = 64, 512, 2048, 6
same = len(sys.argv) > 1 and sys.argv[1] == "same"
def W(n, k): return ops.tile_weights(torch.randn(n, k, device=dev) * k ** -0.5)
layers = []
for l in range(NL if not same else 1):
    layers.append(dict(qkv=W(3 * C, C), out=W(C, C), fc1=W(F, C), fc2=W(C, F), b3=torch.randn(3 * C, device=dev), bc=torch.randn(C, device=dev),
                       bf=torch.randn(F, device=dev), c3=torch.ones(3 * C, device=dev), cf=torch.ones(F, device=dev)))
x = torch.randn(M, C, device=dev); q = torch.zeros(M, 3 * C, device=dev); h = torch.zeros(M, F, device=dev)
stats = torch.zeros(M, C // 16, 2, device=dev); stats[:, :, 1] = 16.0; so_dummy = torch.zeros(M, F // 16, 2, device=dev)
def sk(epi, a, w, bias, c1, st_in, out, st_out, N, K, act=0):
    check(lib.mgea_op_skinny(epi, ptr(a), ptr(w), ptr(bias), ptr(c1) if c1 is not None else None, ptr(st_in), 32, 16, ptr(out), ptr(st_out),
                             M, N, K, act, 0, stream_ptr()))
pattern = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "pattern" else "QOFG"
def step():
    n = 0
    while n < 24:
        for ch in pattern:
            L = layers[0 if same else (n // 4) % NL]
            if ch == "Q": sk(2, x, L["qkv"], L["b3"], L["c3"], stats, q, so_dummy, 3 * C, C)          # LN + in_proj (ACT epilogue, no activation)
            if ch == "O": sk(1, q, L["out"], L["bc"], None, stats, x, stats, C, C)                    # out-proj + residual (+ stats); A = what QKV wrote
            if ch == "F": sk(2, x, L["fc1"], L["bf"], L["cf"], stats, h, so_dummy, F, C, 1)           # LN + fc1 + GELU
            if ch == "G": sk(1, h, L["fc2"], L["bc"], None, stats, x, stats, C, F)                    # fc2 + residual (+ stats)
            n += 1
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
us = e0.elapsed_time(e1) / 200 * 1e3
print(f"GEMM chain {pattern} ({'one weight set' if same else '6 weight sets'}): {us:.1f} us per 24 kernels = {us / 24:.2f} us per kernel "
      f"(each in a chain of itself: 4.12 + 2.90 + 4.18 + 5.40 = 16.6 us per layer = 4.15 us per kernel)", flush=True)
