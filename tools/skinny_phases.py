#!/usr/bin/env python3
"""Where the time of one fused skinny GEMM goes (runs on the GPU box): dbg bit 20 makes every workgroup write
100 MHz timestamps of its phases (entry, loads requested, loads landed, K loop done, LDS reduce ready, stores
done); this prints the launch ramp and per-phase medians over workgroups for the decode-step shapes.
The waits that mode inserts serialise loads and MFMAs, so read it as an upper bound per phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
from mgea._lib import ptr, check, stream_ptr

lib = _lib.load()
dev = "cuda:0"
TS = 1 << 20


def run(name, epi, N, K, ln, mt=0, nw=0, abl=0):
    M = 64
    a = torch.randn(M, K, device=dev); w = ops.tile_weights(torch.randn(N, K, device=dev) * K ** -0.5)
    bias = torch.randn(N, device=dev); lnw = torch.ones(N, device=dev)   # stands in for the folded-LN c1 vector
    stats = torch.zeros(M, 32, 2, device=dev); stats[:, :, 1] = 16.0
    out = torch.zeros(M, N, device=dev)
    n_stat = 64 * (N // 16) * 2
    max_wg = (N // 16 + 8) * 4
    so = torch.zeros(n_stat + max_wg * 16, device=dev)
    dbg = (mt << 8) | (nw << 12) | abl

    def go(d):
        check(lib.mgea_op_skinny(epi, ptr(a), ptr(w), ptr(bias), ptr(lnw) if ln else None,
                                 ptr(stats), 32, 16, ptr(out) if epi != 3 else None, ptr(so), M, N, K, 1 if epi == 2 else 0, d, stream_ptr()))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5): go(dbg)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(8): go(dbg | TS)
        g.replay(); s.synchronize()
        g.replay(); s.synchronize()
    t8 = so[n_stat:].cpu().view(torch.int64).view(-1, 8)
    t8 = t8[t8[:, 0] != 0].double() * 0.01   # us
    t = t8[:, :6]
    t0 = t[:, 0].min()
    ramp = t[:, 0] - t0
    ph = t[:, 1:] - t[:, :-1]
    med = lambda x: x.median().item()
    print(f"{name:10s} N={N:5d} K={K:5d} ln={int(ln)} wgs={t.shape[0]:4d}  span {(t[:, 5].max() - t0).item():6.2f} us | "
          f"start med {med(ramp):5.2f} max {ramp.max().item():5.2f} | issue+LN {med(ph[:, 0]):5.2f} | land {med(ph[:, 1]):5.2f} "
          f"(max {ph[:, 1].max().item():5.2f}) | mfma {med(ph[:, 2]):5.2f} | reduce {med(ph[:, 3]):5.2f} | epi {med(ph[:, 4]):5.2f} "
          f"(max {ph[:, 4].max().item():5.2f}) | wg life med {med(t[:, 5] - t[:, 0]):5.2f}", flush=True)


run("qkv-like", 2, 1536, 512, True)
run("fc1", 2, 2048, 512, True)
run("out-proj", 1, 512, 512, False)
run("fc2", 1, 512, 2048, False)
run("head", 3, 8324, 512, False)
run("ln N1024", 2, 1024, 512, True, mt=2)
run("ln N4096", 2, 4096, 512, True, mt=2)
run("res N1024", 1, 1024, 512, False)
run("res N2048", 1, 2048, 512, False, mt=1)
run("res N256", 1, 256, 512, False)
