#!/usr/bin/env python3
"""Time mgea_op_sample on a [64, 8324] logits matrix in a captured graph (GPU box): which part of the sampler costs what."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib
from mgea._lib import SamplerConfig, ptr, check, stream_ptr
lib = _lib.load()
B, V = 64, 8324
logits = torch.randn(B, V, device="cuda") * 2
ids = torch.zeros(B, dtype=torch.int32, device="cuda")
for name, kw in (("keep all", dict(top_k=0, top_p=0.0)), ("top-k 50", dict(top_k=50, top_p=0.0)), ("top-p 0.9", dict(top_k=0, top_p=0.9)),
                 ("top-k 50 + top-p 0.9", dict(top_k=50, top_p=0.9))):
    sc = SamplerConfig(temperature=1.0, eos_id=-1, seed=7, **kw)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        go = lambda: check(lib.mgea_op_sample(ptr(logits), B, V, C.byref(sc), 3, ptr(ids), None, stream_ptr()))
        for _ in range(3): go()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): go()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); s.synchronize()
    print(f"{name:22s}: {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per launch", flush=True)
