"""Emotion -> music parameters (reference: emotion_analysis/EATS.py:21-42 over lookup_table.csv, the
reference's own 28-row data table, carried here unchanged as data).  Host-side table lookup, no
compute; kept so `from emotion_analysis import inference, EATS` (api_cache.py:4,190) resolves."""
import csv
import json
import os
import random
from typing import Dict, List, Tuple, Union

LOOKUP_PATH = os.path.join(os.path.dirname(__file__), "lookup_table.csv")


def _load(path: str) -> Dict[str, Dict]:
    table = {}
    with open(path, newline="", encoding="utf-8") as f:
        for row in csv.DictReader(f):
            table[row["emotion"]] = {
                "bpm_min": int(row["bpm_min"]), "bpm_max": int(row["bpm_max"]), "key": row["key"],
                "scale_type": row["scale_type"], "instrument_families": json.loads(row["instrument_families"]),
            }
    return table


EATS = _load(LOOKUP_PATH)


def _params_for_label(label: str) -> Dict:
    key = label.lower()
    if key not in EATS:
        raise ValueError(f"Emotion '{label}' not in lookup table")     # EATS.py:23-24
    e = EATS[key]
    bpm = random.randint(e["bpm_min"], e["bpm_max"])                   # same two draws, same order (EATS.py:27-28)
    fam = random.choice(e["instrument_families"])
    return {"emotion": key, "bpm": bpm, "key": e["key"], "scale_type": e["scale_type"], "inst_family": fam,
            "all_families": e["instrument_families"]}


def get_music_params(emotions: Union[str, List[str], Tuple[str, ...]]) -> Union[Dict, List[Dict]]:
    if isinstance(emotions, str):
        return _params_for_label(emotions)
    return [_params_for_label(lab) for lab in emotions]
