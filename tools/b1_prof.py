#!/usr/bin/env python3
"""One B = 1 (or B = argv[1]) greedy generation of 1019 steps for rocprofv3 --kernel-trace --stats: what a single stream's step
is made of (the reference's endpoint serves one request at a time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import time
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = synth.decoder_state_dict(5, 8324, 1024, 512, 6)
eng = DecoderEngine(sd, n_head=8, max_batch=8, max_ctx=1024)
p = [[1, 2, 3, 4, 5]] * B
eng.generate(p, 32, top_k=1); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.generate(p, 1019, top_k=1).cpu(); dt = time.perf_counter() - t0
print(f"B={B} greedy 1019 steps: {dt * 1e3:.1f} ms -> {dt / 1019 * 1e6:.1f} us per step")
