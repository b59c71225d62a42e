"""Model loading for the emotion classifier (reference: emotion_analysis/modeling.py:8-25 pulls
tokenizer, base model and peft adapter from the hub).  Here everything comes from a LOCAL
directory in Hugging Face layout, read with loaders that execute nothing from the files
(safetensors, or torch.load(weights_only=True)):
    vocab.txt                      WordPiece vocabulary
    model.safetensors | pytorch_model.bin          DistilBertForSequenceClassification weights
    adapter_model.safetensors | adapter_model.bin  peft LoRA tensors (+ modules_to_save heads), optional
    adapter_config.json            r / lora_alpha / use_rslora / rank_pattern / alpha_pattern, optional
                                   (defaults r = rank of the tensors, alpha = 16)
    config.json                    n_heads (default 12)
Every LoRA pair of the adapter, whatever Linear it targets, is folded into the weights on the device
(W' = W + scale B A, mgea.bert.resolve_adapter); adapter tensors the loader cannot apply raise instead
of being dropped.  The forward runs in libmgea_hip.so."""
import json
import os
from typing import Dict, Optional

import torch

from mgea.bert import BertEngine
from mgea.tokenizer import WordPieceTokenizer

from .config import NUM_LABELS


def _load_tensors(path_st: str, path_bin: str) -> Optional[Dict]:
    if os.path.exists(path_st):
        from safetensors.torch import load_file
        return load_file(path_st)
    if os.path.exists(path_bin):
        return torch.load(path_bin, map_location="cpu", weights_only=True)
    return None


def load_model(model_dir: Optional[str] = None, device: str = "cuda:0", max_tokens: int = 64 * 512):
    """Returns (tokenizer, engine) like the reference's load_model() returns (tokenizer, model)."""
    model_dir = model_dir or os.environ.get("MGEA_DISTILBERT_DIR")
    if not model_dir or not os.path.isdir(model_dir):
        raise FileNotFoundError(
            "DistilBERT weights are not bundled and the hub is unreachable: set MGEA_DISTILBERT_DIR to a local "
            "directory with vocab.txt + model.safetensors (+ adapter_model.safetensors), or call "
            "emotion_analysis.inference.configure(...)")
    j = lambda n: os.path.join(model_dir, n)
    sd = _load_tensors(j("model.safetensors"), j("pytorch_model.bin"))
    if sd is None:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {model_dir}")
    ad = _load_tensors(j("adapter_model.safetensors"), j("adapter_model.bin"))
    n_heads, ac = 12, None
    if os.path.exists(j("config.json")):
        with open(j("config.json")) as f:
            n_heads = int(json.load(f).get("n_heads", 12))
    if os.path.exists(j("adapter_config.json")):
        with open(j("adapter_config.json")) as f:
            ac = json.load(f)
    if int(sd["classifier.weight"].shape[0]) != NUM_LABELS and not ad:
        raise RuntimeError(f"classifier has {sd['classifier.weight'].shape[0]} labels, expected {NUM_LABELS}")
    tokenizer = WordPieceTokenizer(j("vocab.txt"))
    engine = BertEngine(sd, n_heads=n_heads, adapter=ad, adapter_config=ac, max_tokens=max_tokens, device=device)
    return tokenizer, engine
