#!/usr/bin/env python3
"""Run the DistilBERT-base [256,128] forward a few times (for rocprofv3): python3 tools/bert_prof.py bf16|f32 [packed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import synth
from mgea.bert import BertEngine
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
ad = synth.lora_adapter(41, 768, 6)
eng = BertEngine(sd, n_heads=12, adapter=ad, max_tokens=256 * 128, dtype=dtype)
ids, mask = synth.bert_inputs(2, 256, 128, 30522)
ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
if "packed" in sys.argv[2:]:
    pk = tuple(t.cuda() if isinstance(t, torch.Tensor) else t for t in BertEngine.pack(ids.cpu(), mask.cpu()))
    fwd = lambda: eng.forward_packed(*pk)
else:
    fwd = lambda: eng.forward(ids, mask)
for _ in range(2): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fwd()
torch.cuda.synchronize(); print(dtype, "packed" if "packed" in sys.argv[2:] else "padded", "ms/batch", (time.perf_counter() - t0) / 5 * 1e3, "rows", eng.stats()["rows"])
