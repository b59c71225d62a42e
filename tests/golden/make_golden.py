#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own classes.

Runs only in the build container (needs /root/reference, read-only); never on the GPU box.
Nothing from the reference is copied into the repo: this script parses
`/root/reference/api_cache.py` and `/root/reference/generate_music/generate.py` with `ast`,
executes only the class/function definitions on the hot path (`GPTBlock`, `GPTWithKV`,
`remap_state_dict`, `sample_kvcache`, `GPT`) in a scratch namespace (a plain module import is
impossible offline -- SURVEY.md §8c), feeds them the deterministic synthetic weights of
`mgea/synth.py`, and stores the *outputs* (logits, greedy token ids, masked probabilities)
as small .npz fixtures.  The DistilBERT fixtures come from the container's local
`transformers` `DistilBertForSequenceClassification` (the reference's model definition is that
third-party class, emotion_analysis/modeling.py:14-21) with LoRA folded as W + (alpha/r) B A.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import ast
import hashlib
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
from mgea import synth  # noqa: E402

REF = "/root/reference"


def lift(path, names, extra_globals):
    """exec only the named top-level ClassDef/FunctionDef nodes of a reference file."""
    src = open(path).read()
    tree = ast.parse(src)
    body = [n for n in tree.body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    missing = set(names) - {n.name for n in body}
    assert not missing, f"reference symbols not found: {missing}"
    ns = {"torch": torch, "nn": torch.nn, "re": re}
    ns.update(extra_globals)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def t(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def decoder_fixture(tag, seed, vocab, seq_len, d_model, n_head, n_layer, prompts, n_steps,
                    n_tf_steps, full_logits):
    tok2id = synth.decoder_vocab(vocab)
    id2tok = {i: s for s, i in tok2id.items()}
    ns = lift(os.path.join(REF, "api_cache.py"),
              ["GPTBlock", "GPTWithKV", "remap_state_dict", "sample_kvcache"],
              {"tok2id": tok2id, "id2tok": id2tok})
    sd = synth.decoder_state_dict(seed, vocab, seq_len, d_model, n_layer)
    model = ns["GPTWithKV"](vocab_size=vocab, seq_len=seq_len, d_model=d_model,
                            n_head=n_head, n_layer=n_layer)
    res = model.load_state_dict(ns["remap_state_dict"](t(sd)))
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    out = {"cfg": np.array([seed, vocab, seq_len, d_model, n_head, n_layer], dtype=np.int64)}
    torch.manual_seed(0)
    for pi, ids in enumerate(prompts):
        ptoks = [id2tok[i] for i in ids]
        inp = torch.tensor(ids).unsqueeze(0)
        with torch.no_grad():
            # prefill logits (api_cache.py:163 computes and discards them)
            logits, past = model(inp)
            # greedy = the reference sampler with top_k=1 (deterministic argmax)
            toks = ns["sample_kvcache"](model, ptoks, max_len=len(ids) + n_steps,
                                        temperature=1.0, top_k=1, device="cpu")
            gids = np.array([tok2id[s] for s in toks], dtype=np.int64)
            # teacher-forced step logits following the sampler's own loop (api_cache.py:166-179)
            gen = inp
            steps = []
            for s in range(n_tf_steps):
                lg, past = model(gen[:, -1:], past)
                lg = lg[:, -1, :]
                steps.append(lg[0].numpy().copy())
                gen = torch.cat([gen, lg.argmax(-1, keepdim=True)], 1)
            assert np.array_equal(gen[0].numpy(), gids[: len(ids) + n_tf_steps])
        out[f"prompt{pi}"] = np.array(ids, dtype=np.int64)
        out[f"greedy{pi}"] = gids
        steps = np.stack(steps)
        if full_logits:
            out[f"prefill_logits{pi}"] = logits[0].numpy()
            out[f"step_logits{pi}"] = steps
        else:  # big shape: keep a slice + digest only
            out[f"prefill_logits_head{pi}"] = logits[0, :, :64].numpy()
            out[f"step_logits_head{pi}"] = steps[:, :64]
            out[f"step_logits_max{pi}"] = steps.max(-1)
            srt = np.sort(steps, -1)
            out[f"step_top2_gap{pi}"] = srt[:, -1] - srt[:, -2]
        out[f"greedy_sha{pi}"] = np.frombuffer(
            hashlib.sha256(gids.astype("<i8").tobytes()).digest(), dtype=np.uint8)
    # one top-k=50 masked probability vector, pre-multinomial (api_cache.py:169-177)
    with torch.no_grad():
        inp = torch.tensor(prompts[0]).unsqueeze(0)
        _, past = model(inp)
        lg, _ = model(inp[:, -1:], past)
        lg = lg[:, -1, :] / 0.8
        k = min(50, vocab)
        vals, idxs = lg.topk(k)
        mask = torch.full_like(lg, -1e10)
        mask.scatter_(1, idxs, 0.0)
        probs = torch.softmax(lg + mask, dim=-1)
    out["topk_probs_T0.8_k50"] = probs[0].numpy()

    # no-cache twin: generate_music/generate.py GPT (post-LN, ReLU, full recompute), same weights
    ns2 = lift(os.path.join(REF, "generate_music", "generate.py"), ["GPT"], {})
    twin = ns2["GPT"](vocab, seq_len + 1, d_model, n_head=n_head, n_layer=n_layer)
    twin.load_state_dict(t(sd), strict=True)
    twin.eval()
    with torch.no_grad():
        ids = torch.tensor(prompts[1]).unsqueeze(0)
        tl = twin(ids)[0].numpy()
        # greedy continuation with the twin's full-recompute loop (generate.py:46-61, top_k=1)
        g = ids
        for _ in range(min(8, n_steps)):
            g = torch.cat([g, twin(g)[:, -1, :].argmax(-1, keepdim=True)], 1)
    out["twin_logits1"] = tl if full_logits else tl[:, :64]
    out["twin_greedy1"] = g[0].numpy().astype(np.int64)
    path = os.path.join(HERE, f"decoder_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def bert_fixture(tag, seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq):
    from transformers import DistilBertConfig, DistilBertForSequenceClassification
    cfg = DistilBertConfig(vocab_size=vocab, max_position_embeddings=max_pos, dim=dim,
                           n_heads=n_heads, n_layers=n_layers, hidden_dim=hidden,
                           num_labels=28, dropout=0.1, attention_dropout=0.1,
                           seq_classif_dropout=0.2, sinusoidal_pos_embds=False)
    model = DistilBertForSequenceClassification(cfg).eval()
    sd = synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden)
    ad = synth.lora_adapter(seed, dim, n_layers)
    merged = dict(sd)
    for i in range(n_layers):
        for nm in ("q_lin", "v_lin"):
            p = f"base_model.model.distilbert.transformer.layer.{i}.attention.{nm}."
            A = ad[p + "lora_A.weight"].astype(np.float32)
            B = ad[p + "lora_B.weight"].astype(np.float32)
            k = f"distilbert.transformer.layer.{i}.attention.{nm}.weight"
            # peft LoRA: W x + (alpha/r) B (A x), r=8, alpha=16 (finetuneDistillBert.ipynb:787-795)
            merged[k] = (torch.from_numpy(sd[k]) + 2.0 * (torch.from_numpy(B) @ torch.from_numpy(A))).numpy()
    res = model.load_state_dict(t(merged), strict=False)
    assert not res.unexpected_keys, res
    assert all("position_ids" in k for k in res.missing_keys), res
    ids, mask = synth.bert_inputs(seed + 1, batch, seq, vocab, min_len=4)
    with torch.no_grad():
        logits = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits
        # the reference's single-string path: one row, no padding (inference.py:16)
        n0 = int(mask[1].sum())
        solo = model(input_ids=torch.from_numpy(ids[1:2, :n0]),
                     attention_mask=torch.ones(1, n0, dtype=torch.long)).logits
    out = {
        "cfg": np.array([seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq], dtype=np.int64),
        "ids": ids, "mask": mask, "logits": logits.numpy(),
        "argmax": logits.argmax(1).numpy().astype(np.int64),
        "probs": torch.softmax(logits, 1).numpy(),
        "solo_row1_logits": solo.numpy(),
    }
    path = os.path.join(HERE, f"distilbert_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    torch.set_num_threads(8)
    # tiny decoder: 2 layers, C=128, 2 heads of 64, V=311 (not a multiple of 16), 64 positions
    decoder_fixture("tiny", seed=11, vocab=311, seq_len=64, d_model=128, n_head=2, n_layer=2,
                    prompts=[[1, 5, 14], [1, 7, 20, 33, 34], [1, 9, 25, 33, 34, 35]],
                    n_steps=32, n_tf_steps=16, full_logits=True)
    # 8-head tiny (head dim 32) -- the reference hard-codes n_head=8 (api_cache.py:112)
    decoder_fixture("tiny8h", seed=12, vocab=300, seq_len=48, d_model=256, n_head=8, n_layer=2,
                    prompts=[[1, 4, 15, 33], [1, 8, 22, 34, 35], [1, 3, 30]],
                    n_steps=24, n_tf_steps=8, full_logits=True)
    # Decoder-S real shape (train/train_large2.py:10-12,23-28): 6L/512d/8H, V=8324, 1024 positions
    decoder_fixture("S", seed=21, vocab=8324, seq_len=1024, d_model=512, n_head=8, n_layer=6,
                    prompts=[[1, 6, 17, 33, 34], [1, 10, 28, 35, 33], [1, 4, 13, 34, 35],
                             [1, 11, 30, 33, 35]],
                    n_steps=48, n_tf_steps=48, full_logits=False)
    bert_fixture("tiny", seed=31, vocab=100, max_pos=32, dim=128, n_heads=2, n_layers=2,
                 hidden=512, batch=6, seq=24)
    bert_fixture("base", seed=41, vocab=30522, max_pos=512, dim=768, n_heads=12, n_layers=6,
                 hidden=3072, batch=8, seq=128)


if __name__ == "__main__":
    main()
