#!/usr/bin/env python3
"""The fp16 storage mode under rocprofv3: one warm-up + one generation of Decoder-S (B = 64, 1024 tokens, greedy) and of
BASELINE configs[4] (12L / 768d, B = 64, 2048 tokens, top-p 0.9), so that the kernel-trace averages are per decode step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
which = sys.argv[1] if len(sys.argv) > 1 else "both"
runs = []
if which in ("S", "both"):
    runs.append(("Decoder-S f16", dict(seed=0, vocab=8324, seq_len=1024, d_model=512, n_layer=6), 8, 1024, dict(top_k=1)))
if which in ("L", "both"):
    runs.append(("Decoder-L f16", dict(seed=5, vocab=8324, seq_len=2048, d_model=768, n_layer=12), 12, 2048, dict(top_k=None, top_p=0.9, seed=1)))
for name, g, heads, TL, samp in runs:
    sd = synth.decoder_state_dict(g["seed"], g["vocab"], g["seq_len"], g["d_model"], g["n_layer"])
    eng = DecoderEngine(sd, n_head=heads, max_batch=64, max_ctx=TL, dtype="f16")
    p = torch.from_numpy(synth.integers(9, "prompts", (64, 5), 0, g["vocab"])).to(torch.int32).cuda()
    eng.generate(p, TL - 5, **samp); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.generate(p, TL - 5, **samp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {64 * (TL - 5) / dt:.0f} tokens/s, {dt / (TL - 5) * 1e6:.1f} us/step", flush=True)
    eng.close()
