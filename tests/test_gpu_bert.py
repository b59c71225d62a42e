"""Parity of the HIP DistilBERT(+LoRA) forward against the golden logits of the `transformers`
class (bit-exact argmax labels, logits within 1e-3)."""
import numpy as np
import pytest
import torch

from mgea import synth

pytestmark = pytest.mark.gpu


def make(g, with_adapter=True):
    from mgea.bert import BertEngine
    seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq = (int(x) for x in g["cfg"])
    sd = synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden)
    ad = synth.lora_adapter(seed, dim, n_layers) if with_adapter else None
    return BertEngine(sd, n_heads=n_heads, adapter=ad, max_tokens=batch * seq)


@pytest.mark.parametrize("tag", ["tiny", "base"])
def test_logits_and_labels_vs_golden(golden, tag):
    g = golden("distilbert_" + tag)
    eng = make(g)
    logits, amax = eng.forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-3, rtol=0)
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4
    assert amax.cpu().tolist() == g["argmax"].tolist()


def test_solo_row_without_padding(golden):
    g = golden("distilbert_tiny")
    eng = make(g)
    n = int(g["mask"][1].sum())
    logits, _ = eng.forward(torch.from_numpy(g["ids"][1:2, :n]), None)
    np.testing.assert_allclose(logits.cpu().numpy(), g["solo_row1_logits"], atol=1e-4, rtol=0)


def test_lora_changes_the_result(golden):
    g = golden("distilbert_tiny")
    a, _ = make(g, True).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    b, _ = make(g, False).forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    assert float((a - b).abs().max()) > 1e-3


def test_sequence_longer_than_position_table(golden):
    g = golden("distilbert_tiny")
    eng = make(g)
    with pytest.raises(RuntimeError):
        eng.forward(torch.zeros(1, 33, dtype=torch.long), None)
