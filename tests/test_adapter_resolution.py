"""Host logic of the real-weights ingest (no GPU): what mgea.bert.resolve_adapter makes of a peft adapter
directory's tensors.  The reference applies its adapter with PeftModel.from_pretrained
(emotion_analysis/modeling.py:14-21), which honours every target module and modules_to_save head the adapter
holds; peft is not importable here, so the LoRA definition is restated (parity unpinned) and checked through
merged == unmerged equality in the oracle."""
import math

import numpy as np
import pytest
import torch

from mgea import synth
from mgea.bert import resolve_adapter
from oracle.distilbert_ref import DistilBertRef

D, HID, NL, H = 64, 128, 2, 2      # head_dim 32: the smallest the HIP attention kernels take


def base_sd():
    return synth.distilbert_state_dict(51, 100, 32, D, NL, HID)


def adapter_all_targets(r=4, seed=52, default_infix=False):
    """LoRA on all six Linear kinds of every layer + trained copies of both heads (peft on-disk names)."""
    ad = {}
    shapes = {"attention.q_lin": (D, D), "attention.k_lin": (D, D), "attention.v_lin": (D, D), "attention.out_lin": (D, D),
              "ffn.lin1": (HID, D), "ffn.lin2": (D, HID)}
    inf = ".default" if default_infix else ""
    for i in range(NL):
        for nm, (o, k) in shapes.items():
            p = f"base_model.model.distilbert.transformer.layer.{i}.{nm}."
            ad[p + f"lora_A{inf}.weight"] = synth.uniform(seed, p + "A", (r, k), k ** -0.5)
            ad[p + f"lora_B{inf}.weight"] = synth.uniform(seed, p + "B", (o, r), 0.3 * r ** -0.5)
    ad["base_model.model.pre_classifier.modules_to_save.default.weight"] = synth.uniform(seed, "pcw2", (D, D), D ** -0.5)
    ad["base_model.model.pre_classifier.modules_to_save.default.bias"] = synth.uniform(seed, "pcb2", (D,), 0.05)
    ad["base_model.model.classifier.weight"] = synth.uniform(seed, "clw2", (28, D), 2.0 * D ** -0.5)
    ad["base_model.model.classifier.bias"] = synth.uniform(seed, "clb2", (28,), 0.05)
    return ad


def merged_sd(sd, overrides, loras):
    out = {k: torch.from_numpy(np.asarray(v)).float() for k, v in {**sd, **overrides}.items()}
    for mod, (A, B, scale) in loras.items():
        out[mod + ".weight"] = out[mod + ".weight"] + scale * (torch.from_numpy(np.asarray(B)) @ torch.from_numpy(np.asarray(A)))
    return out


@pytest.mark.parametrize("default_infix", [False, True])
def test_every_target_and_saved_head_is_applied(default_infix):
    sd, ad = base_sd(), adapter_all_targets(default_infix=default_infix)
    overrides, loras = resolve_adapter(sd, ad, dict(r=4, lora_alpha=8))
    assert len(loras) == 6 * NL and set(overrides) == {"pre_classifier.weight", "pre_classifier.bias", "classifier.weight", "classifier.bias"}
    assert all(abs(s - 2.0) < 1e-12 for _, _, s in loras.values())
    ids, mask = synth.bert_inputs(53, 5, 20, 100, min_len=3)
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    want = DistilBertRef(sd, H, {k.replace(".modules_to_save.default", "").replace(".default.weight", ".weight"): v for k, v in ad.items()},
                         lora_scale=2.0, merge=False).forward(ids, mask)
    got = DistilBertRef(merged_sd(sd, overrides, loras), H).forward(ids, mask)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=3e-5, rtol=0)
    base = DistilBertRef(sd, H).forward(ids, mask)
    assert float((base - want).abs().max()) > 1e-2          # the adapter really matters


def test_scales_rslora_and_patterns():
    sd, ad = base_sd(), adapter_all_targets(r=4)
    _, l1 = resolve_adapter(sd, ad, dict(r=4, lora_alpha=8, use_rslora=True))
    assert all(abs(s - 8 / math.sqrt(4)) < 1e-12 for _, _, s in l1.values())
    _, l2 = resolve_adapter(sd, ad, dict(r=4, lora_alpha=8, alpha_pattern={"q_lin": 32}))
    assert all(abs(s - (8.0 if "q_lin" in m else 2.0)) < 1e-12 for m, (_, _, s) in l2.items())
    _, l3 = resolve_adapter(sd, ad, None, lora_alpha=16.0)    # no adapter_config.json: alpha 16 over the tensors' rank
    assert all(abs(s - 4.0) < 1e-12 for _, _, s in l3.values())
    with pytest.raises(ValueError, match="rank"):
        resolve_adapter(sd, ad, dict(r=8, lora_alpha=16))     # config disagrees with the tensors
    resolve_adapter(sd, ad, dict(r=8, lora_alpha=16, rank_pattern={r"layer\.\d+\..*": 4}))   # ... unless rank_pattern says so


def test_unapplied_tensors_raise_instead_of_being_dropped():
    sd, ad = base_sd(), adapter_all_targets()
    bad = dict(ad)
    bad["base_model.model.distilbert.transformer.layer.0.attention.q_lin.lora_magnitude_vector"] = np.ones(D, np.float32)
    with pytest.raises(ValueError, match="cannot apply"):
        resolve_adapter(sd, bad)
    bad = dict(ad)
    bad["base_model.model.distilbert.embeddings.word_embeddings.lora_embedding_A"] = np.ones((4, 100), np.float32)
    with pytest.raises(ValueError, match="cannot apply"):
        resolve_adapter(sd, bad)
    bad = {k: v for k, v in ad.items() if not k.endswith("layer.1.ffn.lin2.lora_B.weight")}    # A without its B
    with pytest.raises(ValueError, match="cannot apply"):
        resolve_adapter(sd, bad)
    with pytest.raises(NotImplementedError):
        resolve_adapter(sd, ad, dict(use_dora=True))
    with pytest.raises(ValueError, match="shape"):
        resolve_adapter(sd, {"base_model.model.classifier.weight": np.zeros((3, D), np.float32)})
    assert resolve_adapter(sd, None) == ({}, {})
