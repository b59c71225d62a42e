"""WordPiece tokenizer for DistilBERT-uncased given a user-supplied vocab.txt.

The reference gets its tokenizer from the hub (`AutoTokenizer.from_pretrained(REPO_ID)`,
emotion_analysis/modeling.py:14) and calls it as `tokenizer(text, return_tensors="pt",
truncation=True, padding=True)` (emotion_analysis/inference.py:16).  No vocabulary exists offline,
so this is an own implementation of the published BERT tokenisation (BasicTokenizer: clean,
lower-case, strip accents, split punctuation, space CJK; WordpieceTokenizer: greedy longest match
with the `##` continuation prefix, `[UNK]` for unmatchable words, words > 100 chars -> `[UNK]`),
checked in tests against `transformers.BertTokenizer` on a local vocab file.
Host-side string work only; ids are int64 like the reference's.
"""
from __future__ import annotations

import unicodedata
from typing import Dict, Iterable, List, Optional, Sequence, Union

import torch


def load_vocab(path: str) -> Dict[str, int]:
    vocab = {}
    with open(path, encoding="utf-8") as f:
        for i, line in enumerate(f):
            vocab[line.rstrip("\n")] = i
    return vocab


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    def __init__(self, vocab: Union[str, Dict[str, int]], do_lower_case: bool = True, model_max_length: int = 512,
                 unk="[UNK]", cls="[CLS]", sep="[SEP]", pad="[PAD]"):
        self.vocab = load_vocab(vocab) if isinstance(vocab, str) else dict(vocab)
        self.do_lower_case = do_lower_case
        self.model_max_length = model_max_length
        self.unk, self.cls, self.sep, self.pad = unk, cls, sep, pad
        for t in (unk, cls, sep, pad):
            if t not in self.vocab:
                raise ValueError(f"vocabulary lacks the special token {t}")
        self.special = {unk, cls, sep, pad, "[MASK]"}
        self.pad_token_id = self.vocab[pad]

    # ---- basic tokenisation ---------------------------------------------------------------
    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            else:
                out.append(" " if _is_whitespace(ch) else ch)
        text = unicodedata.normalize("NFC", "".join(out))
        words = []
        for tok in text.strip().split():
            if tok in self.special:
                words.append(tok)
                continue
            if self.do_lower_case:
                tok = tok.lower()
                tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in tok:
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    def _wordpiece(self, word: str) -> List[str]:
        if len(word) > 100:
            return [self.unk]
        pieces, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        toks = []
        for w in self._basic(text):
            toks += [w] if w in self.special else self._wordpiece(w)
        return toks

    def encode(self, text: str, truncation: bool = True, max_length: Optional[int] = None) -> List[int]:
        ids = [self.vocab[t] for t in self.tokenize(text)]
        limit = (max_length or self.model_max_length) - 2
        if truncation and len(ids) > limit:
            ids = ids[:limit]
        return [self.vocab[self.cls]] + ids + [self.vocab[self.sep]]

    def __call__(self, text: Union[str, Sequence[str]], return_tensors: Optional[str] = "pt", truncation: bool = True,
                 padding: bool = True, max_length: Optional[int] = None):
        texts = [text] if isinstance(text, str) else list(text)
        enc = [self.encode(t, truncation, max_length) for t in texts]
        width = max(len(e) for e in enc)
        ids = [e + [self.pad_token_id] * (width - len(e)) for e in enc]
        mask = [[1] * len(e) + [0] * (width - len(e)) for e in enc]
        if return_tensors == "pt":
            return {"input_ids": torch.tensor(ids, dtype=torch.long), "attention_mask": torch.tensor(mask, dtype=torch.long)}
        return {"input_ids": ids, "attention_mask": mask}
