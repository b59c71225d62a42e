"""Pins oracle/distilbert_ref.py against logits of the container's `transformers` DistilBERT class
on the same synthetic weights (tests/golden/distilbert_*.npz)."""
import numpy as np
import pytest
import torch

from mgea import synth
from oracle.distilbert_ref import DistilBertRef


def build(g, merge=True):
    seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq = (int(x) for x in g["cfg"])
    sd = synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden)
    ad = synth.lora_adapter(seed, dim, n_layers)
    return DistilBertRef(sd, n_heads, ad, merge=merge)


@pytest.mark.parametrize("tag", ["tiny", "base"])
def test_logits_and_labels(golden, tag):
    g = golden("distilbert_" + tag)
    ref = build(g)
    logits = ref.forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5, rtol=0)
    assert logits.argmax(1).tolist() == g["argmax"].tolist()
    np.testing.assert_allclose(torch.softmax(logits, 1).numpy(), g["probs"], atol=1e-6, rtol=0)


def test_inputs_are_the_synthetic_ones(golden):
    g = golden("distilbert_tiny")
    seed, vocab, *_rest, batch, seq = (int(x) for x in g["cfg"])
    ids, mask = synth.bert_inputs(seed + 1, batch, seq, vocab, min_len=4)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(mask, g["mask"])


def test_padded_row_equals_solo_run(golden):
    g = golden("distilbert_tiny")
    ref = build(g)
    n = int(g["mask"][1].sum())
    solo = ref.forward(torch.from_numpy(g["ids"][1:2, :n]))
    np.testing.assert_allclose(solo.numpy(), g["solo_row1_logits"], atol=3e-5, rtol=0)


def test_lora_merged_equals_unmerged(golden):
    g = golden("distilbert_tiny")
    a = build(g, merge=True).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    b = build(g, merge=False).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-5, rtol=0)
