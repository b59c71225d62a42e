"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the DistilBERT(+LoRA)
emotion classifier forward used by emotion_analysis/inference.py:12-22.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The arithmetic of this model is not in /root/reference: it lives in the third-party packages the
reference pins (requirements.txt:3,6,7: torch==2.2.2, transformers==4.40.0, peft==0.10.0) and is
reached through emotion_analysis/modeling.py:14-21.  Restated here from the published DistilBERT
definition (transformers modeling_distilbert.py: Embeddings, MultiHeadSelfAttention, FFN,
TransformerBlock, DistilBertForSequenceClassification) and the LoRA definition
y = W x + b + (alpha/r) B (A x) (r=8, alpha=16, targets q_lin/v_lin:
Scripts/finetuneDistillBert.ipynb:787-795; any other Linear target is handled the same way).  Pinned against logits produced by the container's
local `transformers` class on the same synthetic weights (tests/golden/distilbert_*.npz); the
LoRA branch itself is parity-unpinned (peft absent) beyond merged == unmerged equality.

  embeddings : word[ids] + pos[0..S) -> LayerNorm(eps 1e-12)
  x6 block   : q,k,v = Linear(h); softmax(q k^T / sqrt(dh) + additive key mask) v -> out_lin
               h = LN(attn + h); h = LN(lin2(gelu_erf(lin1(h))) + h)          (post-LN, eps 1e-12)
  head       : h[:,0] -> pre_classifier -> ReLU -> classifier (28 labels)
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu().contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class DistilBertRef:
    def __init__(self, state_dict: Dict, n_heads: int, adapter: Optional[Dict] = None,
                 lora_scale: float = 2.0, merge: bool = True):
        """adapter: peft-named tensors `base_model.model.<module>.lora_A.weight` [r, in] / `.lora_B.weight` [out, r]
        on ANY Linear of the model (the reference's adapter targets q_lin / v_lin; peft applies whatever the adapter
        holds) plus optional `base_model.model.<name>` tensors that replace base ones (modules_to_save heads).
        merge=True folds W' = W + lora_scale * B @ A; merge=False keeps the LoRA branch y += lora_scale * B (A x)."""
        self.sd = {k: _t(v) for k, v in state_dict.items()}
        self.ad = {}
        for k, v in (adapter or {}).items():
            name = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
            if ".lora_" in name:
                self.ad[name] = _t(v)
            else:
                self.sd[name] = _t(v)
        self.H = n_heads
        self.scale = lora_scale
        self.D = self.sd["distilbert.embeddings.word_embeddings.weight"].shape[1]
        self.L = 1 + max(int(k.split(".")[3]) for k in self.sd if k.startswith("distilbert.transformer.layer."))
        self.merge = merge
        if merge:
            for mod in sorted({k.split(".lora_")[0] for k in self.ad}):
                A, B = self._lora(mod)
                self.sd[mod + ".weight"] = self.sd[mod + ".weight"] + self.scale * (B @ A)

    def _lora(self, mod):
        if mod + ".lora_A.weight" not in self.ad:
            return None
        return self.ad[mod + ".lora_A.weight"], self.ad[mod + ".lora_B.weight"]

    def _linear(self, x, mod):
        y = x @ self.sd[mod + ".weight"].t() + self.sd[mod + ".bias"]
        ab = None if self.merge else self._lora(mod)
        if ab is not None:
            y = y + self.scale * ((x @ ab[0].t()) @ ab[1].t())
        return y

    def _lin(self, x, i, nm):
        return self._linear(x, f"distilbert.transformer.layer.{i}.attention.{nm}")

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids [B,S] int64, mask [B,S] 0/1 -> logits [B,28]."""
        sd, D, H = self.sd, self.D, self.H
        dh = D // H
        B, S = ids.shape
        e = "distilbert.embeddings."
        h = sd[e + "word_embeddings.weight"][ids] + sd[e + "position_embeddings.weight"][:S]
        h = F.layer_norm(h, (D,), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], 1e-12)
        add = None
        if mask is not None:
            add = torch.zeros(B, 1, 1, S)
            add = add.masked_fill(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
        for i in range(self.L):
            p = f"distilbert.transformer.layer.{i}."
            q = self._lin(h, i, "q_lin").view(B, S, H, dh).transpose(1, 2)
            k = self._lin(h, i, "k_lin").view(B, S, H, dh).transpose(1, 2)
            v = self._lin(h, i, "v_lin").view(B, S, H, dh).transpose(1, 2)
            s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
            if add is not None:
                s = s + add
            a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, D)
            a = self._lin(a, i, "out_lin")
            h = F.layer_norm(a + h, (D,), sd[p + "sa_layer_norm.weight"], sd[p + "sa_layer_norm.bias"], 1e-12)
            f = F.gelu(self._linear(h, p + "ffn.lin1"))
            f = self._linear(f, p + "ffn.lin2")
            h = F.layer_norm(f + h, (D,), sd[p + "output_layer_norm.weight"], sd[p + "output_layer_norm.bias"], 1e-12)
        pooled = torch.relu(self._linear(h[:, 0], "pre_classifier"))
        return self._linear(pooled, "classifier")
