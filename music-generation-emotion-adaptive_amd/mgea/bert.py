"""Host side of the DistilBERT(+LoRA) classifier engine (replaces the `model(**inputs).logits`
call of emotion_analysis/inference.py:16-20).  Torch = device memory + streams only."""
from __future__ import annotations

import ctypes as C
import math
import re
from typing import Dict, List, Optional, Tuple

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


_LORA_KEY = re.compile(r"^(?P<mod>.+)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")


def _base_name(key: str) -> str:
    """peft tensor name -> name in the base model's state dict: `base_model.model.` prefix,
    `.modules_to_save.<adapter>` / `.base_layer` wrappers removed."""
    k = key[len("base_model.model."):] if key.startswith("base_model.model.") else key
    k = re.sub(r"\.modules_to_save(\.[^.]+)?(?=\.(weight|bias)$)", "", k)
    return k.replace(".base_layer.", ".")


def _pattern_value(pattern: Optional[Dict], module: str, default):
    """peft's rank_pattern / alpha_pattern lookup: the first key that is the module name or a suffix of it
    (regex allowed) wins."""
    for key, val in (pattern or {}).items():
        if re.match(rf"(.*\.)?{key}$", module):
            return val
    return default


def resolve_adapter(sd: Dict, adapter: Optional[Dict], adapter_config: Optional[Dict] = None, lora_alpha: float = 16.0,
                    lora_r: Optional[int] = None) -> Tuple[Dict, Dict]:
    """Turn a peft adapter (what PeftModel.from_pretrained applies, emotion_analysis/modeling.py:14-21) into
    (overrides, loras): `overrides[base_key]` replaces a base tensor (modules_to_save heads, trained biases) and
    `loras[module] = (A [r, in], B [out, r], scale)` is folded as W' = W + scale * B @ A for EVERY Linear the adapter
    targets (q_lin / k_lin / v_lin / out_lin / ffn.lin1 / ffn.lin2 / pre_classifier / classifier).  scale follows
    peft: lora_alpha / r, or lora_alpha / sqrt(r) with use_rslora, with rank_pattern / alpha_pattern per module.
    Anything in the adapter that is not consumed (DoRA magnitudes, embedding LoRA, unknown names) raises: a dropped
    tensor would silently change the logits.  peft is not importable here, so this is restated from its published
    definition (parity unpinned; merged == unmerged is what the tests can check)."""
    cfg = dict(adapter_config or {})
    if cfg.get("use_dora"):
        raise NotImplementedError("adapter_config.use_dora: DoRA adapters are not supported")
    if cfg.get("fan_in_fan_out"):
        raise NotImplementedError("adapter_config.fan_in_fan_out: transposed base weights are not supported")
    alpha0 = float(cfg.get("lora_alpha", lora_alpha))
    r0 = cfg.get("r", lora_r)
    overrides: Dict = {}
    halves: Dict[str, Dict[str, object]] = {}
    leftover: List[str] = []
    for key, t in (adapter or {}).items():
        if ".original_module." in key:      # peft keeps the frozen copy of a modules_to_save layer beside the trained one
            continue
        name = _base_name(key)
        m = _LORA_KEY.match(name)
        if m:
            halves.setdefault(m.group("mod"), {})[m.group("ab")] = t
        elif name in sd:
            if tuple(t.shape) != tuple(sd[name].shape):
                raise ValueError(f"adapter tensor {key} has shape {tuple(t.shape)}, the base model's {name} has {tuple(sd[name].shape)}")
            overrides[name] = t
        else:
            leftover.append(key)
    loras: Dict = {}
    for mod, ab in halves.items():
        wkey = mod + ".weight"
        if "A" not in ab or "B" not in ab or wkey not in sd or len(sd[wkey].shape) != 2:
            leftover += [k for k in adapter if _base_name(k).startswith(mod + ".lora_")]
            continue
        A, Bm = ab["A"], ab["B"]
        out_dim, in_dim = (int(x) for x in sd[wkey].shape)
        r = int(A.shape[0])
        if tuple(A.shape) != (r, in_dim) or tuple(Bm.shape) != (out_dim, r):
            raise ValueError(f"LoRA shapes of {mod}: A {tuple(A.shape)} B {tuple(Bm.shape)} do not fit W [{out_dim}, {in_dim}]")
        r_cfg = _pattern_value(cfg.get("rank_pattern"), mod, r0)
        if r_cfg is not None and int(r_cfg) != r:
            raise ValueError(f"LoRA rank of {mod} is {r}, adapter_config says {r_cfg}")
        alpha = float(_pattern_value(cfg.get("alpha_pattern"), mod, alpha0))
        loras[mod] = (A, Bm, alpha / math.sqrt(r) if cfg.get("use_rslora") else alpha / r)
    if leftover:
        raise ValueError(f"adapter tensors that this loader cannot apply (they would be dropped silently): {sorted(leftover)[:6]}"
                         f"{' ...' if len(leftover) > 6 else ''}")
    return overrides, loras


class BertEngine:
    def __init__(self, state_dict: Optional[Dict], n_heads: int = 12, adapter: Optional[Dict] = None,
                 lora_alpha: float = 16.0, lora_r: Optional[int] = None, max_tokens: int = 256 * 128,
                 device="cuda:0", geometry: Optional[Dict] = None, arena: Optional[torch.Tensor] = None,
                 ln_eps: float = 1e-12, dtype: str = "f32", adapter_config: Optional[Dict] = None):
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
            self.arena = self.pack_arena(state_dict, adapter, lora_alpha, lora_r, adapter_config)
        torch.cuda.synchronize(self.device)
        h = C.c_void_p(0)
        check(self.lib.mgea_bert_create(C.byref(self.cfg), ptr(self.arena), C.byref(h)))
        self.h = h

    def pack_arena(self, sd, adapter, lora_alpha, lora_r, adapter_config=None) -> torch.Tensor:
        """Canonical arena of mgea.h.  q_lin/k_lin/v_lin are stacked into one [3D, D] matrix; every LoRA pair of
        the adapter (resolve_adapter) is folded in place on the device by mgea_lora_merge: W' = W + scale * B A."""
        geo, dev = self.geo, self.device
        D, Hd = geo["dim"], geo["hidden"]
        overrides, loras = resolve_adapter(sd, adapter, adapter_config, lora_alpha, lora_r)
        sd = {**sd, **overrides}      # modules_to_save heads / trained biases replace the base tensors
        arena = torch.zeros(self.arena_floats, dtype=torch.float32, device=dev)
        it = iter(self.offsets)
        slot = {}                      # Linear module -> (arena offset of its weight, out_dim, in_dim)

        def put(t):
            off = next(it)
            t = _f32(t, dev).reshape(-1)
            arena[off:off + t.numel()].copy_(t)
            return off

        e = "distilbert.embeddings."
        put(sd[e + "word_embeddings.weight"]); put(sd[e + "position_embeddings.weight"])
        put(sd[e + "LayerNorm.weight"]); put(sd[e + "LayerNorm.bias"])
        for i in range(geo["n_layers"]):
            p = f"distilbert.transformer.layer.{i}."
            qkv_w = torch.cat([_f32(sd[p + f"attention.{nm}.weight"], dev) for nm in ("q_lin", "k_lin", "v_lin")], 0)
            qkv_b = torch.cat([_f32(sd[p + f"attention.{nm}.bias"], dev) for nm in ("q_lin", "k_lin", "v_lin")], 0)
            off = put(qkv_w)
            put(qkv_b)
            for j, nm in enumerate(("q_lin", "k_lin", "v_lin")):
                slot[p + f"attention.{nm}"] = (off + j * D * D, D, D)
            slot[p + "attention.out_lin"] = (put(sd[p + "attention.out_lin.weight"]), D, D); put(sd[p + "attention.out_lin.bias"])
            put(sd[p + "sa_layer_norm.weight"]); put(sd[p + "sa_layer_norm.bias"])
            slot[p + "ffn.lin1"] = (put(sd[p + "ffn.lin1.weight"]), Hd, D); put(sd[p + "ffn.lin1.bias"])
            slot[p + "ffn.lin2"] = (put(sd[p + "ffn.lin2.weight"]), D, Hd); put(sd[p + "ffn.lin2.bias"])
            put(sd[p + "output_layer_norm.weight"]); put(sd[p + "output_layer_norm.bias"])
        slot["pre_classifier"] = (put(sd["pre_classifier.weight"]), D, D); put(sd["pre_classifier.bias"])
        slot["classifier"] = (put(sd["classifier.weight"]), geo["num_labels"], D); put(sd["classifier.bias"])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        keep = []
        for mod, (A, Bm, scale) in loras.items():
            if mod not in slot:
                raise ValueError(f"LoRA target {mod} is not a Linear of the classifier")
            off, out_dim, in_dim = slot[mod]
            A, Bm = _f32(A, dev), _f32(Bm, dev)
            keep += [A, Bm]
            w_ptr = C.c_void_p(arena.data_ptr() + 4 * off)
            check(self.lib.mgea_lora_merge(w_ptr, ptr(A), ptr(Bm), out_dim, in_dim, int(A.shape[0]), float(scale), stream))
        torch.cuda.synchronize(dev)
        del keep
        return arena

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgea_bert_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self) -> Dict[str, int]:
        """What the last forward ran (mgea_bert_stats): lets a test assert which kernels produced the logits it checks."""
        out = (C.c_int64 * 16)()
        check(self.lib.mgea_bert_stats(self.h, out))
        return dict(forwards=out[0], folded_layernorm=bool(out[1]), gemm_persistent=out[2], gemm_ring=out[3], gemm_small=out[4],
                    gemm_half_tile_tails=out[5], layernorm_kernels=out[6], last_layer_cls_only=bool(out[7]),
                    gemm_by_epilogue=[int(out[8 + e]) for e in range(6)], rows=int(out[14]))

    BF16_MIN_TOKENS = 512        # a bf16 engine's routing threshold (csrc/bert.hip): smaller calls run on its exact-fp32 kernels
    PACKED_MAX_LEN_16BIT = 256   # one query block / key stage of the 16-bit flash attention per sequence

    @staticmethod
    def pack(ids: torch.Tensor, mask: Optional[torch.Tensor]):
        """HOST tensors [B, S] + 0/1 mask -> (ids [n] int32, pos [n] int32, cu_seqlens [B + 1] int32, max_len) of the real tokens, or None
        when the mask is not a prefix mask (holes / leading padding: only the padded form represents those) or a sequence is empty.
        The tokenizer of emotion_analysis/inference.py:16 (padding=True) pads on the right, so its masks are prefix masks."""
        B, S = ids.shape
        if mask is None:
            return None
        m = mask.to(torch.bool)
        lens = m.sum(1)
        if int(lens.min()) < 1 or not bool((m == (torch.arange(S)[None, :] < lens[:, None])).all()):
            return None
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = lens.cumsum(0).to(torch.int32)
        pos = torch.arange(S, dtype=torch.int32)[None, :].expand(B, S)[m].contiguous()
        return ids[m].to(torch.int32).contiguous(), pos, cu, int(lens.max())

    def forward_packed(self, ids: torch.Tensor, pos: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, want_logits=True, want_argmax=True):
        """The forward on PACKED rows (pack() above; mgea_bert_forward_packed): ids / pos [n_tokens], cu_seqlens [B + 1], max_len = the longest
        sequence (any length on the exact-fp32 kernels; <= 256 on a bf16 engine's 16-bit kernels, i.e. from 512 tokens on).
        Same (logits [B, labels], argmax [B]) as forward() on the padded batch."""
        B = int(cu_seqlens.numel()) - 1
        n = int(ids.numel())
        if not ids.is_cuda and n and (int(ids.min()) < 0 or int(ids.max()) >= self.geo["vocab"]):
            raise IndexError("index out of range in self")
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        pos32 = pos.to(device=self.device, dtype=torch.int32).contiguous()
        cu32 = cu_seqlens.to(device=self.device, dtype=torch.int32).contiguous()
        logits = torch.empty(B, self.num_labels, dtype=torch.float32, device=self.device) if want_logits else None
        amax = torch.empty(B, dtype=torch.int32, device=self.device) if want_argmax else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.mgea_bert_forward_packed(self.h, ptr(ids32), ptr(pos32), ptr(cu32), B, n, int(max_len), ptr(logits), ptr(amax), st))
        return logits, amax

    def forward_auto(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None, want_logits=True, want_argmax=True):
        """forward(), on packed rows where that applies: HOST ids / mask as the tokenizer produces them, a right-padded batch with at
        least one padding token (and, on a bf16 engine's 16-bit kernels, no sequence longer than 256) -- otherwise the padded call."""
        if not ids.is_cuda and mask is not None and not mask.is_cuda and ids.dim() == 2:
            pk = self.pack(ids, mask)
            if pk is not None and pk[0].numel() < ids.numel():
                on16 = self.dtype == "bf16" and pk[0].numel() >= self.BF16_MIN_TOKENS
                if not on16 or pk[3] <= self.PACKED_MAX_LEN_16BIT:
                    return self.forward_packed(*pk, want_logits=want_logits, want_argmax=want_argmax)
        return self.forward(ids, mask, want_logits, want_argmax)

    def id_errors(self, raise_error: bool = True) -> int:
        """Read and clear the engine's sticky device flags (ONE stream sync): bit 0 = some token id handed over as a DEVICE tensor
        since the last call was outside the vocabulary (and was clamped); raises the IndexError nn.Embedding raises in the reference."""
        flags = C.c_int32(0)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.mgea_bert_error_flags(self.h, C.byref(flags), st))
        if raise_error and (flags.value & 1):
            raise IndexError("index out of range in self")
        return flags.value

    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None, want_logits=True, want_argmax=True):
        """ids [B,S] (any int dtype), mask [B,S] 0/1 -> (logits [B,labels] fp32, argmax [B] int32)."""
        if ids.dim() != 2:
            raise RuntimeError("input_ids must be [B, S]")
        B, S = ids.shape
        # ids outside the vocabulary: nn.Embedding raises IndexError in the reference.  A HOST tensor (what the tokenizer hands over,
        # emotion_analysis/inference.py:14-16) is checked here at no GPU cost; a DEVICE tensor is not read back -- that put a host
        # sync and two reduction kernels in front of every forward (~0.1 ms of idle GPU per [256, 128] batch) -- the embedding
        # kernel clamps such ids and sets a sticky device flag: id_errors() reads it (one sync, whenever the caller reads the logits
        # anyway) and raises the reference's IndexError then.
        if not ids.is_cuda and ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.geo["vocab"]):
            raise IndexError("index out of range in self")
        ids32 = ids.to(device=self.device, dtype=torch.int32).contiguous()
        m32 = None if mask is None else mask.to(device=self.device, dtype=torch.int32).contiguous()
        logits = torch.empty(B, self.num_labels, dtype=torch.float32, device=self.device) if want_logits else None
        amax = torch.empty(B, dtype=torch.int32, device=self.device) if want_argmax else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.lib.mgea_bert_forward(self.h, ptr(ids32), ptr(m32), B, S, ptr(logits), ptr(amax), st))
        return logits, amax
