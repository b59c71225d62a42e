#!/usr/bin/env python3
"""Single-prompt DistilBERT latency (the endpoint classifies one text per request, api_cache.py:189) by sequence length."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.bert import BertEngine
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
ad = synth.lora_adapter(41, 768, 6)
eng = BertEngine(sd, n_heads=12, adapter=ad, max_tokens=256 * 128, dtype="f32")
for B, S in ((1, 16), (1, 32), (1, 128), (8, 32), (256, 128)):
    ids, mask = synth.bert_inputs(2, B, S, 30522)
    ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for _ in range(3): eng.forward(ids, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.forward(ids, mask)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d} S={S:3d}: {dt*1e3:7.3f} ms per forward", flush=True)
