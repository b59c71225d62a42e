#!/usr/bin/env python3
"""A/B of a library switch on the fp16 matrix-core prefill of Decoder-S [64, 1024] (cache fill, logits dropped), variants interleaved in
one process:  python3 tools/prefill_ab.py decoder_prefill16_pages 0 1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, synth
from mgea.decoder import DecoderEngine
name = sys.argv[1] if len(sys.argv) > 1 else "decoder_prefill16_pages"
vals = [int(v) for v in sys.argv[2:]] or [0, 1]
B, T = 64, 1024
sd = synth.decoder_state_dict(0, 8324, 1024, 512, 6)
eng = DecoderEngine(sd, n_head=8, max_batch=B, max_ctx=T, dtype="f16")
ids = torch.from_numpy(synth.integers(1, "p", (B, T), 0, 8324)).cuda()
old = _lib.tune_get(name)
res = {}
WL = os.environ.get("PREFILL_LOGITS", "0") == "1"      # PREFILL_LOGITS=1: with the logits of every position (the head GEMM, 2.2 GB of fp32 rows)
for rep in range(5):
    for v in vals:
        _lib.tune_set(name, v)
        for _ in range(2): eng.reset_and_prefill(ids, want_logits=WL)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.reset_and_prefill(ids, want_logits=WL)
        torch.cuda.synchronize(); res.setdefault(v, []).append((time.perf_counter() - t0) / 5 * 1e3)
_lib.tune_set(name, old)
flops = 2 * B * T * (6 * 12 * 512 * 512) + 4 * B * T * T * 512 * 6
for v, t in res.items():
    ms = sorted(t)[len(t) // 2]
    print(f"{name} = {v}: median {ms:.3f} ms (min {min(t):.3f}), {B * T / ms / 1e3:.2f} M tokens/s, {flops / ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s", flush=True)
