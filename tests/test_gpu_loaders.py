"""Real-weights ingest end to end, from files written into a temp directory:
  * emotion_analysis.modeling.load_model  -- the local-directory counterpart of the reference's hub loads
    (emotion_analysis/modeling.py:14-21): safetensors / .bin weights + peft adapter + adapter_config.json + vocab.txt;
  * generate_music.generate.load_checkpoint -- api_cache.py:26-37,108-138: torch.load(weights_only=True) ->
    geometry from tensor shapes -> remap -> load.
The real model files exist only on the hub (unreachable), so the files hold the synthetic weights whose outputs the
golden fixtures pin (tests/golden/distilbert_tiny.npz from `transformers`, decoder_tiny.npz from the reference classes)."""
import json
import os

import numpy as np
import pytest
import torch

from mgea import synth

pytestmark = pytest.mark.gpu

WORDS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "i", "am", "walk", "##ing", "down", "a", "road", "and", "see", "rain",
         "##bow", "it", "is", "sunny", ".", ",", "!", "love", "life", "so", "happy", "sad", "the", "##s", "?"]


def write_model_dir(d, g, fmt="safetensors", adapter=True, adapter_cfg=True):
    seed, vocab, max_pos, dim, n_heads, n_layers, hidden, batch, seq = (int(x) for x in g["cfg"])
    sd = {k: torch.from_numpy(v) for k, v in synth.distilbert_state_dict(seed, vocab, max_pos, dim, n_layers, hidden).items()}
    ad = {k: torch.from_numpy(v) for k, v in synth.lora_adapter(seed, dim, n_layers).items()}
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file(sd, os.path.join(d, "model.safetensors"))
        if adapter:
            save_file(ad, os.path.join(d, "adapter_model.safetensors"))
    else:
        torch.save(sd, os.path.join(d, "pytorch_model.bin"))
        if adapter:
            torch.save(ad, os.path.join(d, "adapter_model.bin"))
    if adapter and adapter_cfg:   # what peft writes for Scripts/finetuneDistillBert.ipynb:787-795
        json.dump(dict(peft_type="LORA", r=8, lora_alpha=16, lora_dropout=0.1, target_modules=["q_lin", "v_lin"], bias="none",
                       use_rslora=False, use_dora=False, rank_pattern={}, alpha_pattern={}, fan_in_fan_out=False,
                       modules_to_save=["pre_classifier", "classifier"]), open(os.path.join(d, "adapter_config.json"), "w"))
    json.dump(dict(n_heads=n_heads, dim=dim, n_layers=n_layers, hidden_dim=hidden, vocab_size=vocab), open(os.path.join(d, "config.json"), "w"))
    words = WORDS + [f"w{i}" for i in range(vocab - len(WORDS))]
    with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(words[:vocab]) + "\n")
    return sd, ad, n_heads


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_load_model_from_local_dir_matches_transformers_golden(golden, tmp_path, fmt):
    from emotion_analysis.modeling import load_model
    g = golden("distilbert_tiny")
    write_model_dir(str(tmp_path), g, fmt)
    tokenizer, engine = load_model(str(tmp_path), max_tokens=6 * 24)
    logits, amax = engine.forward(torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-4, rtol=0)
    assert amax.cpu().tolist() == g["argmax"].tolist()
    enc = tokenizer("i am walking down a road!", return_tensors="pt", truncation=True, padding=True)
    assert enc["input_ids"][0].tolist() == [2, 5, 6, 7, 8, 9, 10, 11, 21, 3]
    engine.close()


def test_load_model_without_adapter_config_and_without_adapter(golden, tmp_path):
    """No adapter_config.json: alpha 16 over the tensors' rank (= the reference's r 8 / alpha 16).  No adapter at all:
    the base weights alone, which must differ."""
    from emotion_analysis.modeling import load_model
    g = golden("distilbert_tiny")
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    write_model_dir(str(a), g, adapter_cfg=False)
    write_model_dir(str(b), g, adapter=False)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    _, ea = load_model(str(a), max_tokens=6 * 24)
    la, _ = ea.forward(ids, mask)
    np.testing.assert_allclose(la.cpu().numpy(), g["logits"], atol=1e-4, rtol=0)
    _, eb = load_model(str(b), max_tokens=6 * 24)
    lb, _ = eb.forward(ids, mask)
    assert float((la - lb).abs().max()) > 1e-3
    with pytest.raises(FileNotFoundError):
        load_model(str(tmp_path / "nowhere"))


def test_inference_module_over_a_loaded_directory(golden, tmp_path):
    """emotion_analysis.inference.predict & co. on a directory: tokenizer -> HIP forward -> label, against the oracle
    fed with the same token ids."""
    import emotion_analysis.inference as inf
    from emotion_analysis.config import ID2LABEL
    from oracle.distilbert_ref import DistilBertRef
    g = golden("distilbert_tiny")
    sd, ad, n_heads = write_model_dir(str(tmp_path), g)
    inf.configure(model_dir=str(tmp_path))
    text = "i am so happy, i love life and the rainbow!"
    enc = inf.tokenizer(text, return_tensors="pt", truncation=True, padding=True)
    want = DistilBertRef(sd, n_heads, ad).forward(enc["input_ids"], enc["attention_mask"])
    assert inf.predict(text) == ID2LABEL[int(want.argmax(1))]
    probs = torch.softmax(want, 1)[0]
    assert inf.predict_all_labels(text) == {ID2LABEL[i]: round(float(p), 4) for i, p in enumerate(probs)}
    assert [l for l, _ in inf.predict_top_k_labels(text, 3)] == [ID2LABEL[int(i)] for i in probs.topk(3).indices]
    inf.model.close()
    inf.tokenizer = inf.model = None


def test_adapter_on_every_linear_target_matches_unmerged_oracle():
    """LoRA on all six Linear kinds + modules_to_save heads, folded on the device, against the oracle that keeps the LoRA
    branch unmerged (out_lin / lin1 / lin2 adapters used to be dropped silently)."""
    from mgea.bert import BertEngine
    from oracle.distilbert_ref import DistilBertRef
    from test_adapter_resolution import D, H, adapter_all_targets, base_sd
    sd, ad = base_sd(), adapter_all_targets()
    eng = BertEngine(sd, n_heads=H, adapter=ad, adapter_config=dict(r=4, lora_alpha=8), max_tokens=5 * 20)
    ids, mask = synth.bert_inputs(53, 5, 20, 100, min_len=3)
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    logits, amax = eng.forward(ids, mask)
    want = DistilBertRef(sd, H, {k.replace(".modules_to_save.default", ""): v for k, v in ad.items()}, lora_scale=2.0,
                         merge=False).forward(ids, mask)
    np.testing.assert_allclose(logits.cpu().numpy(), want.numpy(), atol=1e-4, rtol=0)
    assert amax.cpu().tolist() == want.argmax(1).tolist()
    bad = dict(ad)
    bad["base_model.model.distilbert.transformer.layer.0.attention.q_lin.lora_magnitude_vector"] = np.ones(D, np.float32)
    with pytest.raises(ValueError):
        BertEngine(sd, n_heads=H, adapter=bad, max_tokens=64)


def test_load_checkpoint_roundtrip_gives_the_reference_golden_ids(golden, tmp_path):
    """torch.save({"model", "vocab"}) in the training-script layout (train/train_large2.py:100-110) -> load_checkpoint ->
    the greedy ids the reference's own classes produced for these weights (tests/golden/decoder_tiny.npz)."""
    import generate_music.generate as gen
    g = golden("decoder_tiny")
    seed, vocab, seq_len, d_model, n_head, n_layer = (int(x) for x in g["cfg"])
    sd = {k: torch.from_numpy(v) for k, v in synth.decoder_state_dict(seed, vocab, seq_len, d_model, n_layer).items()}
    path = str(tmp_path / "music_generator.pt")
    torch.save({"model": sd, "vocab": synth.decoder_vocab(vocab), "cfg": {"d_model": d_model}}, path)
    model, tok2id, id2tok, SEQ_LEN, D_MODEL = gen.load_checkpoint(path, n_head=n_head)
    assert (SEQ_LEN, D_MODEL, len(tok2id)) == (seq_len, d_model, vocab) and model.n_layer == n_layer
    assert gen.model is model and id2tok[tok2id["[START_SEQUENCE]"]] == "[START_SEQUENCE]"
    for i in range(3):
        want = [id2tok[j] for j in g[f"greedy{i}"].tolist()]
        prompt = [id2tok[j] for j in g[f"prompt{i}"].tolist()]
        assert gen.sample_kvcache(model, prompt, max_len=len(want), temperature=1.0, top_k=1, device="cpu") == want
    logits, _ = model(torch.tensor([g["prompt1"].tolist()]))
    np.testing.assert_allclose(logits[0].cpu().numpy(), g["prefill_logits1"], atol=1e-3, rtol=0)
    # a checkpoint pickled with arbitrary objects is refused by the weights_only loader, not executed
    evil = str(tmp_path / "evil.pt")
    torch.save({"model": sd, "vocab": {"a": 0}, "hook": np.random.RandomState(0)}, evil)
    with pytest.raises(Exception):
        gen.load_checkpoint(evil)
