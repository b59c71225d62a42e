#!/usr/bin/env python3
"""Experiment: the B = 64 generation as N independent engines of 64/N rows each, driven from N host threads on their own
streams (each engine replays its own hipGraph), so one lane's latency-bound GEMMs can overlap another's HBM-bound attention."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.decoder import DecoderEngine
Tp, TL, B = 5, 1024, 64
sd = synth.decoder_state_dict(5, 8324, 1024, 512, 6)
prompts = torch.from_numpy(synth.integers(1, "p", (B, Tp), 0, 8324)).to(torch.int32).cuda()
for n in (1, 2, 4):
    engs = [DecoderEngine(sd, n_head=8, max_batch=B // n, max_ctx=TL) for _ in range(n)]
    parts = [prompts[i * (B // n):(i + 1) * (B // n)].contiguous() for i in range(n)]
    outs = [None] * n
    def work(i):
        outs[i] = engs[i].generate(parts[i], TL - Tp, top_k=1)
    def run():
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run()
    dt = min(run(), run())
    print(f"{n} engine(s) x {B // n} rows: {dt*1e3:7.1f} ms per generation, {B*(TL-Tp)/dt:9.0f} tokens/s", flush=True)
    del engs
