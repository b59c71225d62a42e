#!/usr/bin/env python3
"""One 32-token prompt through the f32 DistilBERT engine, 20 forwards (for a rocprofv3 kernel trace: GPU time against wall time per request)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.bert import BertEngine
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
eng = BertEngine(sd, n_heads=12, max_tokens=4096, dtype="f32")
ids, mask = synth.bert_inputs(2, 1, 32, 30522)
ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
for _ in range(5): eng.forward(ids, mask)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.forward(ids, mask)
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per request (wall)")
