#!/usr/bin/env python3
"""In-pipeline A/B of a library switch on the bf16 DistilBERT [256, 128] forward: the same engine, the variants interleaved in
one process (clocks drift by a few % over a run), medians of 7 rounds of 10 forwards.
  python3 tools/bert_ab.py bf16_gemm_tail 0 1 2      python3 tools/bert_ab.py bert_bf16_nofold 0 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, synth
from mgea.bert import BertEngine
name = sys.argv[1] if len(sys.argv) > 1 else "bf16_gemm_tail"
vals = [int(v) for v in sys.argv[2:]] or [0, 1, 2]
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
eng = BertEngine(sd, n_heads=12, adapter=synth.lora_adapter(41, 768, 6), max_tokens=256 * 128, dtype="bf16")
ids, mask = synth.bert_inputs(2, 256, 128, 30522)
ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
old = _lib.tune_get(name)
res = {}
for rep in range(7):
    for v in vals:
        _lib.tune_set(name, v)
        for _ in range(2): eng.forward(ids, mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): eng.forward(ids, mask)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(v, []).append(e0.elapsed_time(e1) / 10)
_lib.tune_set(name, old)
for v, t in res.items():
    ms = sorted(t)[len(t) // 2]
    print(f"{name} = {v}: median {ms:.3f} ms per [256, 128] forward (min {min(t):.3f}), {256 / ms:.1f} k prompts/s, "
          f"{2.861 / ms / 2.5:.3f} of 2.5 PFLOP/s", flush=True)
