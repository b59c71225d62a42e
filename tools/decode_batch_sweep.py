#!/usr/bin/env python3
"""Decoder-S greedy decode throughput and per-step latency by batch size (the reference serves B = 1,
api_cache.py:159-184; the benchmark config is B = 64).  5-token prompts, generation to 1024 tokens."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
Tp, TL = 5, 1024
sd = synth.decoder_state_dict(5, 8324, 1024, 512, 6)
BATCHES = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]
eng = DecoderEngine(sd, n_head=8, max_batch=max(64, max(BATCHES)), max_ctx=TL)
for B in BATCHES:
    prompts = torch.from_numpy(synth.integers(1, "p", (B, Tp), 0, 8324)).to(torch.int32).cuda()
    for mode, kw in (("greedy", dict(top_k=1)), ("top-k 50 (reference default)", dict(top_k=50, seed=3)), ("top-p 0.9", dict(top_k=None, top_p=0.9, seed=3))):
        eng.generate(prompts, TL - Tp, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.generate(prompts, TL - Tp, **kw); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B:3d} {mode:30s}: {dt*1e3:7.1f} ms per generation, {dt/(TL-Tp)*1e6:6.1f} us per step, {B*(TL-Tp)/dt:9.0f} tokens/s", flush=True)
