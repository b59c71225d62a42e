#!/usr/bin/env python3
"""Single-request serving latency of the reference's own endpoint shape (api_cache.py:204: B = 1, top_k = 50, a FRESH seed per
request): per-request wall time of generate() incl. prefill, reset and the D2H copy of the ids, for 1 / 16 / 256 / 1019 decode
steps, with the step-graph instantiation counter before and after -- a new seed must not re-capture.
Then alternating batch sizes 1 / 2 (two cached graphs, no re-capture either)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
sd = synth.decoder_state_dict(5, 8324, 1024, 512, 6)
eng = DecoderEngine(sd, n_head=8, max_batch=8, max_ctx=1024)
p = [[1, 2, 3, 4, 5]]
for n in (1, 16, 256, 1019):
    eng.generate(p, n, top_k=50, seed=1); torch.cuda.synchronize()
    i0 = eng.stats()["graph_instantiates"]
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); out = eng.generate(p, n, top_k=50, seed=100 + r).cpu(); ts.append(time.perf_counter() - t0)
    print(f"B=1 top-k=50 n_steps={n:4d}: {min(ts)*1e3:8.3f} ms per request (min of 5, incl. prefill, reset, D2H)  -> {min(ts)/n*1e6:7.1f} us/step; "
          f"graph instantiations during the 5 requests with 5 new seeds: {eng.stats()['graph_instantiates'] - i0}")
i0 = eng.stats()["graph_instantiates"]
ts = []
for r in range(8):
    pp = p * (1 + r % 2)
    t0 = time.perf_counter(); eng.generate(pp, 16, top_k=50, seed=r).cpu(); ts.append(time.perf_counter() - t0)
print("alternating B=1/B=2, 16 steps:", [round(t*1e3, 2) for t in ts], "ms; graph instantiations:", eng.stats()["graph_instantiates"] - i0,
      "(the B=2 graph is captured once, in the second request)")
g = eng.generate(p, 256, top_k=1).cpu(); t0 = time.perf_counter(); g = eng.generate(p, 256, top_k=1).cpu(); dt = time.perf_counter() - t0
print(f"B=1 greedy n_steps=256: {dt*1e3:.3f} ms per request -> {dt/256*1e6:.1f} us/step")
