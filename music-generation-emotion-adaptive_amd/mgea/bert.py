"""Host side of the DistilBERT(+LoRA) classifier engine (replaces the `model(**inputs).logits`
call of emotion_analysis/inference.py:16-20).  Torch = device memory + streams only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import BertConfig, check, ptr


def _f32(t, device):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def geometry_from_state_dict(sd: Dict) -> Dict[str, int]:
    we = sd["distilbert.embeddings.word_embeddings.weight"]
    pe = sd["distilbert.embeddings.position_embeddings.weight"]
    n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("distilbert.transformer.layer."))
    return dict(vocab=int(we.shape[0]), dim=int(we.shape[1]), max_pos=int(pe.shape[0]), n_layers=n_layers,
                hidden=int(sd["distilbert.transformer.layer.0.ffn.lin1.weight"].shape[0]),
                num_labels=int(sd["classifier.weight"].shape[0]))


class BertEngine:
    def __init__(self, state_dict: Optional[Dict], n_heads: int = 12, adapter: Optional[Dict] = None,
                 lora_alpha: float = 16.0, lora_r: Optional[int] = None, max_tokens: int = 256 * 128,
                 device="cuda:0", geometry: Optional[Dict] = None, arena: Optional[torch.Tensor] = None,
                 ln_eps: float = 1e-12, dtype: str = "f32"):
        """dtype "f32" = parity mode (exact-fp32 MFMA); "bf16" = perf mode (bf16 weights/activations,
        fp32 accumulate, fp32 LayerNorm/softmax/GELU; the arena stays fp32 and is converted once)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BertEngine needs a ROCm device ('cuda:N'); there is no CPU path")
        geo = dict(geometry) if geometry is not None else geometry_from_state_dict(state_dict)
        self.geo = geo
        self.n_heads = int(n_heads)
        self.num_labels = geo["num_labels"]
        self.cfg = BertConfig(vocab=geo["vocab"], max_pos=geo["max_pos"], dim=geo["dim"], n_heads=self.n_heads,
                              n_layers=geo["n_layers"], hidden=geo["hidden"], num_labels=geo["num_labels"],
                              max_tokens=int(max_tokens),
                              dtype=_lib.DTYPE_BF16 if dtype == "bf16" else _lib.DTYPE_F32, ln_eps=ln_eps)
        self.dtype = dtype
        n = C.c_int32(0)
        total = C.c_int64(0)
        check(self.lib.mgea_bert_arena_layout(C.byref(self.cfg), None, C.byref(n), C.byref(total)))
        offs = (C.c_int64 * n.value)()
        check(self.lib.mgea_bert_arena_layout(C.byref(self.cfg), offs, C.byref(n), C.byref(total)))
        self.offsets, self.arena_floats = list(offs), total.value
        torch.cuda.set_device(self.device)
        if arena is not None:
            if arena.numel() != self.arena_floats or arena.dtype != torch.float32 or arena.device != self.device:
                raise ValueError("arena tensor has the wrong size / dtype / device")
            self.arena = arena
        else:
            self.arena = self.pack_arena(state_dict, adapter, lora_alpha, lora_r)
        torch.cuda.synchronize(self.device)
        h = C.c_void_p(0)
        check(self.lib.mgea_bert_create(C.byref(self.cfg), ptr(self.arena), C.byref(h)))
        self.h = h

    def pack_arena(self, sd, adapter, lora_alpha, lora_r) -> torch.Tensor:
        """Canonical arena of mgea.h.  q_lin/k_lin/v_lin are stacked into one [3D, D] matrix; LoRA
        A/B (peft names `base_model.model.<...>.lora_A.weight`, optionally `.default.`) are folded
        in place on the device by mgea_lora_merge: W' = W + (alpha/r) B A."""
        geo, dev = self.geo, self.device
        D = geo["dim"]
        arena = torch.zeros(self.arena_floats, dtype=torch.float32, device=dev)
        it = iter(self.offsets)

        def put(t):
            off = next(it)
            t = _f32(t, dev).reshape(-1)
            arena[off:off + t.numel()].copy_(t)
            return off

        e = "distilbert.embeddings."
        put(sd[e + "word_embeddings.weight"]); put(sd[e + "position_embeddings.weight"])
        put(sd[e + "LayerNorm.weight"]); put(sd[e + "LayerNorm.bias"])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        keep = []
        for i in range(geo["n_layers"]):
            p = f"distilbert.transformer.layer.{i}."
            qkv_w = torch.cat([_f32(sd[p + f"attention.{nm}.weight"], dev) for nm in ("q_lin", "k_lin", "v_lin")], 0)
            qkv_b = torch.cat([_f32(sd[p + f"attention.{nm}.bias"], dev) for nm in ("q_lin", "k_lin", "v_lin")], 0)
            off = put(qkv_w)
            put(qkv_b)
            if adapter:
                for j, nm in enumerate(("q_lin", "k_lin", "v_lin")):
                    A = self._find(adapter, f"transformer.layer.{i}.attention.{nm}.lora_A")
                    Bm = self._find(adapter, f"transformer.layer.{i}.attention.{nm}.lora_B")
                    if A is None or Bm is None:
                        continue
                    A, Bm = _f32(A, dev), _f32(Bm, dev)
                    r = A.shape[0]
                    keep += [A, Bm]
                    scale = float(lora_alpha) / float(lora_r or r)
                    w_ptr = C.c_void_p(arena.data_ptr() + 4 * (off + j * D * D))
                    check(self.lib.mgea_lora_merge(w_ptr, ptr(A), ptr(Bm), D, D, r, scale, stream))
            put(sd[p + "attention.out_lin.weight"]); put(sd[p + "attention.out_lin.bias"])
            put(sd[p + "sa_layer_norm.weight"]); put(sd[p + "sa_layer_norm.bias"])
            put(sd[p + "ffn.lin1.weight"]); put(sd[p + "ffn.lin1.bias"])
            put(sd[p + "ffn.lin2.weight"]); put(sd[p + "ffn.lin2.bias"])
            put(sd[p + "output_layer_norm.weight"]); put(sd[p + "output_layer_norm.bias"])
        # peft `modules_to_save` heads override the base ones when present in the adapter
        for nm in ("pre_classifier", "classifier"):
            for part in ("weight", "bias"):
                t = self._find(adapter, f"model.{nm}.", part) if adapter else None
                put(t if t is not None else sd[f"{nm}.{part}"])
        torch.cuda.synchronize(dev)
        del keep
        return arena

    @staticmethod
    def _find(adapter, frag, suffix="weight"):
        """First adapter tensor whose name contains `frag` and ends with `suffix` (peft writes
        `...q_lin.lora_A.weight` to disk and `...q_lin.lora_A.default.weight` in memory; saved heads
        are `base_model.model.pre_classifier[.modules_to_save.default].weight`)."""
        for k, v in adapter.items():
            if frag in k and k.endswith(suffix):
                return v
        return None

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgea_bert_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None, want_logits=True, want_argmax=True):
        """ids [B,S] (any int dtype), mask [B,S] 0/1 -> (logits [B,labels] fp32, argmax [B] int32)."""
        if ids.dim() != 2:
            raise RuntimeError("input_ids must be [B, S]")
        B, S = ids.shape
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.geo["vocab"]):
            raise IndexError("index out of range in self")
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        m32 = None if mask is None else mask.to(device=self.device, dtype=torch.int32).contiguous()
        logits = torch.empty(B, self.num_labels, dtype=torch.float32, device=self.device) if want_logits else None
        amax = torch.empty(B, dtype=torch.int32, device=self.device) if want_argmax else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.mgea_bert_forward(self.h, ptr(ids32), ptr(m32), B, S, ptr(logits), ptr(amax), st))
        return logits, amax
