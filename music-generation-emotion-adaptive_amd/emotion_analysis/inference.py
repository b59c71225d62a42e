"""Drop-in for emotion_analysis/inference.py (reference lines 12-94): same function names,
signatures and return types; `classify` is the north-star alias of `predict` with a batched form.
The forward pass runs on the MI355X through libmgea_hip.so (mgea.bert.BertEngine).  Unlike the
reference (inference.py:10 loads the model from the hub at import), loading is lazy: the first
call reads MGEA_DISTILBERT_DIR, or use configure() to hand over a tokenizer + engine."""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch

from .config import ID2LABEL
from .data_preprocessing import segment_text

tokenizer = None
model = None


def configure(tokenizer_=None, engine=None, model_dir: Optional[str] = None, device: str = "cuda:0") -> None:
    """Install the (tokenizer, engine) pair; with model_dir, load it like modeling.load_model()."""
    global tokenizer, model
    if model_dir is not None or tokenizer_ is None or engine is None:
        from .modeling import load_model
        tokenizer, model = load_model(model_dir, device)
    else:
        tokenizer, model = tokenizer_, engine


def _logits(text: Union[str, Sequence[str]]) -> torch.Tensor:
    if model is None:
        configure()
    inputs = tokenizer(text, return_tensors="pt", truncation=True, padding=True)     # inference.py:16
    # (a batch of texts runs on packed rows where the engine can: a bf16 engine, >= 512 real tokens -- same logits, no padding rows computed)
    logits, _ = model.forward_auto(inputs["input_ids"], inputs["attention_mask"], want_argmax=False)
    return logits.cpu()


def predict(text: str) -> str:
    """Emotion label of one text (inference.py:12-22): argmax over the 28 logits -- the engine's own device argmax
    (ties to the lowest id, like torch.argmax), so no torch arithmetic sits between the kernels and the label."""
    if model is None:
        configure()
    inputs = tokenizer(text, return_tensors="pt", truncation=True, padding=True)     # inference.py:16
    _, amax = model.forward(inputs["input_ids"], inputs["attention_mask"], want_logits=False)
    return ID2LABEL[int(amax[0])]


def classify(text: Union[str, Sequence[str], torch.Tensor], attention_mask: Optional[torch.Tensor] = None):
    """North-star name.  str -> label; list of str or an id tensor [B,S] -> list of labels."""
    if isinstance(text, str):
        return predict(text)
    if isinstance(text, torch.Tensor):
        if model is None:
            configure()
        _, amax = model.forward(text, attention_mask, want_logits=False)
        labels = [ID2LABEL[int(i)] for i in amax.cpu().tolist()]
        if text.is_cuda:
            model.id_errors()     # device-resident ids were not read back before the forward: the IndexError of nn.Embedding comes here
        return labels
    logits = _logits(list(text))
    return [ID2LABEL[int(i)] for i in logits.argmax(1).tolist()]


def predict_all_labels(text: str) -> dict:
    """label -> round(probability, 4) for all 28 labels (inference.py:26-38)."""
    probabilities = torch.softmax(_logits(text), dim=1).squeeze().tolist()
    return {ID2LABEL[i]: round(prob, 4) for i, prob in enumerate(probabilities)}


def predict_top_k_labels(text: str, k: int = 3) -> list:
    """[(label, round(p, 4))] of the k most probable labels, descending (inference.py:41-60)."""
    probabilities = torch.softmax(_logits(text), dim=1).squeeze()
    topk = torch.topk(probabilities, k)
    return [(ID2LABEL[idx.item()], round(prob.item(), 4)) for idx, prob in zip(topk.indices, topk.values)]


def predict_labels_above_threshold(text: str, threshold: float = 0.2) -> list:
    """[(label, round(p, 4))] with p > threshold in label order (inference.py:62-80)."""
    probabilities = torch.softmax(_logits(text), dim=1).squeeze()
    return [(ID2LABEL[i], round(p.item(), 4)) for i, p in enumerate(probabilities) if p.item() > threshold]


def analyze_emotion_transitions(text: str):
    """(segment, emotion) per sentence (inference.py:83-94)."""
    return [(segment, predict(segment)) for segment in segment_text(text)]
