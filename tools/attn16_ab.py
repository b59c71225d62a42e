#!/usr/bin/env python3
"""In-process A/B of a library switch on the 16-bit flash attention alone (bf16; the fp16 instantiation runs the same instruction
stream), on the decoder prefill shape [64, 1024, 8 x 64] and on the DistilBERT shape [256, 128, 12 x 64], variants interleaved:
    python3 tools/attn16_ab.py attn16_pipe 0 1"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "music-generation-emotion-adaptive_amd"))
import torch
from mgea import _lib, ops
name = sys.argv[1] if len(sys.argv) > 1 else "attn16_pipe"
vals = [int(v) for v in sys.argv[2:]] or [0, 1]
old = _lib.tune_get(name)
for (B, T, H) in ((64, 1024, 8), (256, 128, 12)):
    qkv = (torch.randn(B, T, 3 * 64 * H, device="cuda") * 1.5).bfloat16()
    res, outs = {}, {}
    for rep in range(5):
        for v in vals:
            _lib.tune_set(name, v)
            for _ in range(3): outs[v] = ops.attention_bf16(qkv, H, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.attention_bf16(qkv, H, None)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 1e3 / 20)
    flops = 4.0 * B * H * T * T * 64
    for v, t in res.items():
        us = sorted(t)[len(t) // 2]
        same = all(torch.equal(outs[v], outs[vals[0]]) for _ in (0,))
        print(f"[{B}, {T}, {H} x 64] {name} = {v}: median {us:7.1f} us (min {min(t):7.1f})  {flops / us / 1e6:7.1f} TFLOP/s = {flops / us / 1e6 / 2500:.3f} of 2.5 PF; "
              f"bitwise equal to {name} = {vals[0]}: {same}", flush=True)
_lib.tune_set(name, old)
