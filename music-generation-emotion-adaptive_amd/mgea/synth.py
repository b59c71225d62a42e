"""Deterministic synthetic weights / inputs for the two models on the hot path.

No trained checkpoint of either model exists offline (SURVEY.md §2 "Missing blobs"), so every
test, fixture and benchmark uses weights produced here.  The generator is a counter-based
splitmix64 hash written with plain numpy uint64 arithmetic, so the same (seed, tensor name)
gives bit-identical float32 tensors in this container, on the GPU box and in the golden
generator -- fixtures therefore only need to store *outputs*, never weights.

Key layouts produced:
  * decoder: the *training-script* checkpoint layout consumed by the reference loader
    (`emb.weight`, `pos`, `tr.layers.N.self_attn.in_proj_weight`, ..., `fc.weight`),
    reference: train/train_large2.py:83-110 and api_cache.py:118-134.
  * DistilBERT: Hugging Face `DistilBertForSequenceClassification` state-dict names plus a
    LoRA adapter on q_lin / v_lin (r=8, alpha=16; Scripts/finetuneDistillBert.ipynb:787-795).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = ((seed & 0xFFFFFFFF) << 32) | h
    return _splitmix64(np.array([k], dtype=np.uint64))[0]


def uniform(seed: int, name: str, shape, scale: float = 1.0, shift: float = 0.0) -> np.ndarray:
    """float32 tensor with entries `shift + scale * u`, u uniform in [-1, 1) with 24-bit resolution."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _key(seed, name))
    u = (bits >> np.uint64(40)).astype(np.float64) * (2.0 / (1 << 24)) - 1.0
    return (shift + scale * u).astype(np.float32).reshape(shape)


def integers(seed: int, name: str, shape, lo: int, hi: int) -> np.ndarray:
    """int64 tensor uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + _key(seed, name))
    return (lo + (bits >> np.uint64(11)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape)


# --------------------------------------------------------------------------------------------
# Decoder (training-script checkpoint layout)
# --------------------------------------------------------------------------------------------

def decoder_state_dict(seed: int, vocab: int, seq_len: int, d_model: int, n_layer: int,
                       d_ff: int | None = None) -> Dict[str, np.ndarray]:
    """Random decoder checkpoint in the layout `remap_state_dict` consumes (api_cache.py:118-134).

    `pos` has `seq_len` rows (api_cache.py:36 reads SEQ_LEN = pos.shape[0]).  Scales follow
    torch's default initialisers so activations have realistic magnitudes; the reference's zero
    init of `pos` (api_cache.py:80) would make positions untestable, so it is N(0, ~0.02)-like.
    """
    C = d_model
    F = d_ff if d_ff is not None else 4 * C
    sd: Dict[str, np.ndarray] = {}
    sd["emb.weight"] = uniform(seed, "emb.weight", (vocab, C), 1.7)
    sd["pos"] = uniform(seed, "pos", (seq_len, C), 0.035)
    for i in range(n_layer):
        p = f"tr.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = uniform(seed, p + "in_w", (3 * C, C), (6.0 / (4 * C)) ** 0.5)
        sd[p + "self_attn.in_proj_bias"] = uniform(seed, p + "in_b", (3 * C,), 0.02)
        sd[p + "self_attn.out_proj.weight"] = uniform(seed, p + "out_w", (C, C), C ** -0.5)
        sd[p + "self_attn.out_proj.bias"] = uniform(seed, p + "out_b", (C,), 0.02)
        sd[p + "linear1.weight"] = uniform(seed, p + "l1_w", (F, C), C ** -0.5)
        sd[p + "linear1.bias"] = uniform(seed, p + "l1_b", (F,), C ** -0.5)
        sd[p + "linear2.weight"] = uniform(seed, p + "l2_w", (C, F), F ** -0.5)
        sd[p + "linear2.bias"] = uniform(seed, p + "l2_b", (C,), F ** -0.5)
        sd[p + "norm1.weight"] = uniform(seed, p + "n1_w", (C,), 0.05, 1.0)
        sd[p + "norm1.bias"] = uniform(seed, p + "n1_b", (C,), 0.05)
        sd[p + "norm2.weight"] = uniform(seed, p + "n2_w", (C,), 0.05, 1.0)
        sd[p + "norm2.bias"] = uniform(seed, p + "n2_b", (C,), 0.05)
    sd["fc.weight"] = uniform(seed, "fc.weight", (vocab, C), C ** -0.5)
    sd["fc.bias"] = uniform(seed, "fc.bias", (vocab,), C ** -0.5)
    return sd


def decoder_vocab(vocab: int, with_eos: bool = False) -> Dict[str, int]:
    """A token-string vocabulary of `vocab` entries shaped like the reference's
    (control tokens + `[NOTE] ...` strings, train/train_large2.py:23-30, api_cache.py:157,203)."""
    toks = ["[PAD]", "[START_SEQUENCE]"]
    if with_eos:
        toks.append("[END_SEQUENCE]")
    toks += [f"[BPM] {b}" for b in (60, 72, 84, 96, 108, 120, 132, 144, 160, 180)]
    for k in ("C", "D", "E", "F", "G", "A", "B", "B-", "E-", "F#"):
        toks += [f"[KEY_SIGNATURE] {k} major", f"[KEY_SIGNATURE] {k} minor"]
    toks += ["[INSTRUMENT] Violin", "[INSTRUMENT] Acoustic Grand Piano", "[INSTRUMENT] Flute"]
    names = ["C", "C#", "D", "E-", "E", "F", "F#", "G", "G#", "A", "B-", "B"]
    i = 0
    while len(toks) < vocab:
        pitch = f"{names[i % 12]}{2 + (i // 12) % 5}"
        start = round(0.25 * (i // 60), 2)
        dur = (0.25, 0.5, 1.0)[i % 3]
        toks.append(f"[NOTE] [PITCH:{pitch}] [START:{start}] [END:{round(start + dur, 2)}] [DURATION:{dur}]")
        i += 1
    toks = toks[:vocab]
    assert len(set(toks)) == vocab, "synthetic vocabulary must be collision free"
    return {t: i for i, t in enumerate(toks)}


# --------------------------------------------------------------------------------------------
# DistilBERT (+ LoRA adapter)
# --------------------------------------------------------------------------------------------

def distilbert_state_dict(seed: int, vocab: int, max_pos: int, dim: int, n_layers: int,
                          hidden: int, num_labels: int = 28) -> Dict[str, np.ndarray]:
    """HF `DistilBertForSequenceClassification` tensor names (transformers, modeling_distilbert.py)."""
    D = dim
    sd: Dict[str, np.ndarray] = {}
    e = "distilbert.embeddings."
    sd[e + "word_embeddings.weight"] = uniform(seed, "we", (vocab, D), 0.06)
    sd[e + "position_embeddings.weight"] = uniform(seed, "pe", (max_pos, D), 0.04)
    sd[e + "LayerNorm.weight"] = uniform(seed, "eln_w", (D,), 0.05, 1.0)
    sd[e + "LayerNorm.bias"] = uniform(seed, "eln_b", (D,), 0.05)
    for i in range(n_layers):
        p = f"distilbert.transformer.layer.{i}."
        for nm in ("q_lin", "k_lin", "v_lin", "out_lin"):
            sd[p + f"attention.{nm}.weight"] = uniform(seed, p + nm + "w", (D, D), 1.6 * D ** -0.5)
            sd[p + f"attention.{nm}.bias"] = uniform(seed, p + nm + "b", (D,), 0.05)
        sd[p + "sa_layer_norm.weight"] = uniform(seed, p + "saw", (D,), 0.05, 1.0)
        sd[p + "sa_layer_norm.bias"] = uniform(seed, p + "sab", (D,), 0.05)
        sd[p + "ffn.lin1.weight"] = uniform(seed, p + "l1w", (hidden, D), 1.6 * D ** -0.5)
        sd[p + "ffn.lin1.bias"] = uniform(seed, p + "l1b", (hidden,), 0.05)
        sd[p + "ffn.lin2.weight"] = uniform(seed, p + "l2w", (D, hidden), 1.6 * hidden ** -0.5)
        sd[p + "ffn.lin2.bias"] = uniform(seed, p + "l2b", (D,), 0.05)
        sd[p + "output_layer_norm.weight"] = uniform(seed, p + "olw", (D,), 0.05, 1.0)
        sd[p + "output_layer_norm.bias"] = uniform(seed, p + "olb", (D,), 0.05)
    sd["pre_classifier.weight"] = uniform(seed, "pcw", (D, D), 1.7 * D ** -0.5)
    sd["pre_classifier.bias"] = uniform(seed, "pcb", (D,), 0.05)
    sd["classifier.weight"] = uniform(seed, "clw", (num_labels, D), 3.0 * D ** -0.5)
    sd["classifier.bias"] = uniform(seed, "clb", (num_labels,), 0.05)
    return sd


def lora_adapter(seed: int, dim: int, n_layers: int, r: int = 8) -> Dict[str, np.ndarray]:
    """LoRA A/B for q_lin and v_lin of every layer, peft tensor naming
    (`...attention.q_lin.lora_A.weight` [r, D], `lora_B.weight` [D, r])."""
    ad: Dict[str, np.ndarray] = {}
    for i in range(n_layers):
        for nm in ("q_lin", "v_lin"):
            p = f"base_model.model.distilbert.transformer.layer.{i}.attention.{nm}."
            ad[p + "lora_A.weight"] = uniform(seed, p + "A", (r, dim), dim ** -0.5)
            ad[p + "lora_B.weight"] = uniform(seed, p + "B", (dim, r), 0.3 * r ** -0.5)
    return ad


def bert_inputs(seed: int, batch: int, seq: int, vocab: int, min_len: int = 16):
    """ids [B,S] int64 with first token 101-like CLS id, suffix padding (id 0) and a 0/1 mask
    (SURVEY.md §8d synthetic inputs)."""
    lo = min(1000, vocab // 4)
    ids = integers(seed, "bert_ids", (batch, seq), lo, vocab)
    lens = integers(seed, "bert_lens", (batch,), min(min_len, seq), seq + 1)
    lens[0] = seq  # at least one full-length row
    mask = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[: max(1, batch // 2), 0] = min(101, vocab - 1)  # CLS-like id; other rows keep a random first id
    return ids, mask
