#!/usr/bin/env python3
"""Where a unit (128 queries x 128 keys) of the 16-bit flash attention spends its time: in-kernel 100 MHz stamps from the tools-only
build (tools/build_stamps.sh).  Stamps per unit: 0 top | 1 DMA of this unit landed (vmcnt(0)) | 2 barrier passed | 3 next unit's DMA,
mask and Q requested, previous output stored | 4 the two 64-key tiles computed | 5 end (output staged if it was the item's last block).
  python3 tools/attn_stamps.py [B T H]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGEA_LIB_PATH", os.path.join(ROOT, "tools", "libmgea_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
import numpy as np
import torch
from mgea import _lib, ops
lib = _lib.load()
lib.mgea_dbg_set_ph_stamps.restype = C.c_int
lib.mgea_dbg_set_ph_stamps.argtypes = [C.c_void_p]
B, T, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1024, 8)
stamps = torch.zeros(512, 64, dtype=torch.int64, device="cuda")
assert lib.mgea_dbg_set_ph_stamps(C.c_void_p(stamps.data_ptr())) == 0
qkv = (torch.randn(B, T, 3 * H * 64, device="cuda") * 1.0).bfloat16()
for _ in range(3): ops.attention_bf16(qkv, H, None)
torch.cuda.synchronize(); stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attention_bf16(qkv, H, None); e1.record(); torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64) / 100.0
print(f"attention [{B}, {T}, {H} x 64] bf16: launch {e0.elapsed_time(e1) * 1e3:.1f} us, {4 * B * H * T * T * 64 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.0f} TFLOP/s")
names = ["vmcnt(0) wait", "barrier", "issue next DMA + flush", "2 tiles of 64 keys", "unit end"]
for u in range(1, 6):
    base = u * 8
    wg = s[:, base] > 0
    if not wg.any(): break
    d = [np.median(s[wg, base + i + 1] - s[wg, base + i]) for i in range(5)]
    print(f"  unit {u} ({int(wg.sum())} workgroups, {np.median(s[wg, base + 5] - s[wg, base]):.2f} us): " + " | ".join(f"{n} +{x:.2f}" for n, x in zip(names, d)))
