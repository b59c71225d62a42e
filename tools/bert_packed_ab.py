#!/usr/bin/env python3
"""DistilBERT bf16 on the bench batch [256, 128] (lengths 16..128): padded forward against the PACKED forward (real tokens only), and the
packed forward with / without the folded-LayerNorm pipeline (switch bert_bf16_nofold), interleaved in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, synth
from mgea.bert import BertEngine
B, S = 256, 128
sd = synth.distilbert_state_dict(41, 30522, 512, 768, 6, 3072)
eng = BertEngine(sd, n_heads=12, max_tokens=B * S, dtype="bf16")
ids_np, mask_np = synth.bert_inputs(2, B, S, 30522)
ids, mask = torch.from_numpy(ids_np), torch.from_numpy(mask_np)
pk = tuple(t.cuda() if isinstance(t, torch.Tensor) else t for t in BertEngine.pack(ids, mask))
idd, md = ids.cuda(), mask.cuda()
res = {}
def run(label, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); res.setdefault(label, []).append((time.perf_counter() - t0) / 10 * 1e3)
for rep in range(5):
    for nofold in (0, 1):
        _lib.tune_set("bert_bf16_nofold", nofold)
        run(f"padded nofold={nofold}", lambda: eng.forward(idd, md))
        run(f"packed nofold={nofold}", lambda: eng.forward_packed(*pk))
_lib.tune_set("bert_bf16_nofold", 0)
print(f"real tokens {int(mask.sum())} of {B * S}; last forward ran on {eng.stats()['rows']} rows")
for k, t in res.items():
    ms = sorted(t)[len(t) // 2]
    print(f"{k:20s} median {ms:.3f} ms (min {min(t):.3f})  {B / ms:.1f} k prompts/s", flush=True)
