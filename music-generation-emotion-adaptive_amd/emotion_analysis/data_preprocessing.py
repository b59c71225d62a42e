"""Sentence segmentation used only by analyze_emotion_transitions (reference:
emotion_analysis/data_preprocessing.py:5-11 uses nltk punkt, which needs a download).  A small
rule-based splitter stands in; it is host-side text handling outside the accelerated path."""
import re

_SPLIT = re.compile(r"(?<=[.!?])\s+(?=[\"'(\[]?[A-Z0-9])")


def segment_text(text: str):
    return [s for s in (p.strip() for p in _SPLIT.split(text.strip())) if s]
