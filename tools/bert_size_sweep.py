#!/usr/bin/env python3
"""DistilBERT-base forward time by batch size, f32 engine against bf16 engine with its 16-bit kernels forced for every size (the threshold
below which a bf16 engine runs on the exact-fp32 kernels comes from this sweep).  python3 tools/bert_size_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.bert import BertEngine
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
engs = {dt: BertEngine(sd, n_heads=12, max_tokens=128 * 128, dtype=dt) for dt in ("f32", "bf16")}
for B, S in ((1, 32), (4, 64), (4, 128), (8, 128), (16, 128), (32, 128), (64, 128), (128, 128)):
    ids, mask = synth.bert_inputs(2, B, S, 30522)
    ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    row = []
    for dt, eng in engs.items():
        for _ in range(3): eng.forward(ids, mask)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): eng.forward(ids, mask)
        torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / n * 1e3
        st = eng.stats()
        row.append(f"{dt} {dt_ms:7.3f} ms ({'16-bit kernels' if st['gemm_persistent'] + st['gemm_ring'] + st['gemm_small'] else 'fp32 kernels'})")
    print(f"[{B:3d}, {S:3d}] = {B * S:5d} tokens: " + "   ".join(row), flush=True)
