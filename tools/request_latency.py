import os, sys, time
sys.path.insert(0, "/root/repo/music-generation-emotion-adaptive_amd")
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
sd = synth.decoder_state_dict(5, 8324, 1024, 512, 6)
eng = DecoderEngine(sd, n_head=8, max_batch=8, max_ctx=1024)
p = [[1, 2, 3, 4, 5]]
for n in (1, 16, 256):
    eng.generate(p, n, top_k=50, seed=1); torch.cuda.synchronize()
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); out = eng.generate(p, n, top_k=50, seed=r).cpu(); ts.append(time.perf_counter() - t0)
    print(f"B=1 top-k=50 n_steps={n:4d}: {min(ts)*1e3:8.3f} ms per request (incl. prefill, reset, D2H)  -> {min(ts)/n*1e6:7.1f} us/step")
# alternating batch sizes (graph re-capture?)
ts = []
for r in range(6):
    pp = p * (1 + r % 2)
    t0 = time.perf_counter(); eng.generate(pp, 16, top_k=50, seed=r).cpu(); ts.append(time.perf_counter() - t0)
print("alternating B=1/B=2, 16 steps:", [round(t*1e3, 2) for t in ts], "ms")
