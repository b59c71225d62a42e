#!/usr/bin/env python3
"""Where the time of the balanced LM-head kernel goes (csrc/head_gemm.hip; runs on the GPU box): dbg bit 21 launches the stamped build
of the benchmark's instantiation (64 x 8324 x 512, fp32), in which lane 0 of EVERY wave writes 100 MHz stamps: entry, loads issued,
chunk 0 landed, everything landed, MFMAs done (+ bias requested), barrier passed, stores acknowledged.  The explicit waits of that
build serialise loads and MFMAs: read the shares, not the length.  Also times the product kernel against the generic one in a graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
from mgea._lib import ptr, check, stream_ptr

lib = _lib.load()
dev = "cuda:0"
M, N, K = 64, 8324, 512
a = ops.tile_rows(torch.randn(M, K, device=dev))
w = ops.tile_weights(torch.randn(N, K, device=dev) * K ** -0.5)
bias = torch.randn(N, device=dev)
P = int(lib.mgea_op_skinny_logits_partials(M, N, K))
so = torch.zeros(2 * 64 * 521 + 256 * 8 * 8 * 2 + 4096, device=dev)
# a 96 MB buffer rotated through between launches stands in for the other weights of a step (the head's W is not L2-resident in the step)
rot = torch.empty(96 << 18, device=dev)


def go(d):
    check(lib.mgea_op_skinny(3, ptr(a), ptr(w), ptr(bias), None, None, 0, 16, None, ptr(so), M, N, K, 0, d, stream_ptr()))


def timed(label, d, reps=20, cold=True):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        go(d); s.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            if cold: rot.add_(1.0)
            e0.record(s); go(d); e1.record(s)
        s.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    print(f"{label:34s} median {ts[len(ts) // 2]:6.2f} us  min {ts[0]:6.2f}  (event pair, {'cold' if cold else 'back to back'})", flush=True)


for sw, name in ((1, "balanced head kernel"), (0, "generic skinny kernel")):
    _lib.tune_set("head_balanced", sw)
    timed(name, 0, cold=True)
    timed(name, 0, cold=False)
_lib.tune_set("head_balanced", 1)
for cold in (True, False):
    if cold: rot.add_(1.0)
    torch.cuda.synchronize()
    go(1 << 21); torch.cuda.synchronize()
    t = so[2 * 64 * P: 2 * 64 * P + P * 8 * 8 * 2].cpu().view(torch.int64).view(P, 8, 8).double() * 0.01    # us, [wg][wave][stamp]
    t0 = t[:, :, 0].min()
    names = ["issue", "chunk0 landed", "all landed", "mfma", "lds+barrier", "epilogue+stores"]
    ph = t[:, :, 1:7] - t[:, :, 0:6]
    print(f"stamped build, {'cold' if cold else 'warm'}: span {(t[:, :, 6].max() - t0):.2f} us; start ramp med {(t[:, :, 0] - t0).median():.2f} max {(t[:, :, 0] - t0).max():.2f}")
    for i, nme in enumerate(names):
        x = ph[:, :, i].flatten()
        print(f"   {nme:16s} med {x.median():5.2f}  p90 {x.kthvalue(int(0.9 * x.numel())).values:5.2f}  max {x.max():5.2f} us")
    life = (t[:, :, 6] - t[:, :, 0]).flatten()
    print(f"   wave life        med {life.median():5.2f}  max {life.max():5.2f} us")
