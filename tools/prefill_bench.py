#!/usr/bin/env python3
"""Decoder-S prefill [64, 1024] (non-causal, no past, logits dropped like the sampler): time, tokens/s and TFLOP/s vs the MFMA
peak of the dtype (3.86 TFLOP incl. the head; without head 3.30).   python3 tools/prefill_bench.py [f32|f16] [logits]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
B, T = 64, 1024
DEC = dict(vocab=8324, seq_len=1024, d_model=512, n_layer=6, d_ff=2048)
sd = synth.decoder_state_dict(0, DEC["vocab"], DEC["seq_len"], DEC["d_model"], DEC["n_layer"])
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
peak = 157.3 if dtype == "f32" else 2500.0
eng = DecoderEngine(sd, n_head=8, max_batch=B, max_ctx=T, dtype=dtype)
ids = torch.from_numpy(synth.integers(1, "p", (B, T), 0, DEC["vocab"])).cuda()
for want_logits in ((False, True) if "logits" in sys.argv[2:] else (False,)):
    for _ in range(2): eng.reset_and_prefill(ids, want_logits=want_logits)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): eng.reset_and_prefill(ids, want_logits=want_logits)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    C, NL, V = 512, 6, 8324
    flops = 2 * B * T * (NL * 12 * C * C) + 4 * B * T * T * C * NL + (2 * B * T * V * C if want_logits else 0)
    print(f"prefill [64,1024] {dtype} logits={want_logits}: {dt*1e3:.2f} ms  {B*T/dt:.0f} tok/s  {flops/dt/1e12:.1f} TFLOP/s ({flops/dt/1e12/peak:.3f} of the {dtype} MFMA peak)")
