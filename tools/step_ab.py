#!/usr/bin/env python3
"""In-process A/B of a library switch on the headline generation (Decoder-S, B = 64, 5 -> 1024 tokens, greedy, f32): one engine per value
(switches that shape the captured graph are read when it is captured), generations interleaved.   python3 tools/step_ab.py <switch> <values...> [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, synth
from mgea.decoder import DecoderEngine
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:]] or [0, 1]
B, Tp, TL = int(os.environ.get("STEP_AB_B", 64)), 5, 1024      # STEP_AB_B=1: the single-stream case
sd = synth.decoder_state_dict(0, 8324, 1024, 512, 6)
prompts = torch.from_numpy(synth.integers(1, "prompts", (B, Tp), 0, 8324)).to(torch.int32).cuda()
old = _lib.tune_get(name)
engs, outs, res = {}, {}, {}
for v in vals:
    _lib.tune_set(name, v)
    engs[v] = DecoderEngine(sd, n_head=8, max_batch=B, max_ctx=TL)
    outs[v] = engs[v].generate(prompts, TL - Tp, top_k=1)
for rep in range(4):
    for v in vals:
        _lib.tune_set(name, v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engs[v].generate(prompts, TL - Tp, top_k=1)
        torch.cuda.synchronize(); res.setdefault(v, []).append(time.perf_counter() - t0)
_lib.tune_set(name, old)
for v in vals:
    dt = sorted(res[v])[len(res[v]) // 2]
    print(f"{name} = {v}: {dt * 1e3:.2f} ms per generation, {B * (TL - Tp) / dt / 1e3:.1f} k tokens/s, {dt / (TL - Tp) * 1e6:.2f} us per step; "
          f"ids equal to {name} = {vals[0]}: {bool(torch.equal(outs[v], outs[vals[0]]))}", flush=True)
