"""Label map of the 28-way GoEmotions classifier (reference: emotion_analysis/config.py:3-36).
REPO_ID names the hub model the reference downloads; this build never fetches it -- point
MGEA_DISTILBERT_DIR (or inference.configure) at a local copy instead."""

REPO_ID = "SaiRohitMurali/distilbertmodel-598"

_LABELS = ("admiration amusement anger annoyance approval caring confusion curiosity desire disappointment "
           "disapproval disgust embarrassment excitement fear gratitude grief joy love nervousness optimism "
           "pride realization relief remorse sadness surprise neutral").split()
ID2LABEL = dict(enumerate(_LABELS))
NUM_LABELS = len(_LABELS)
assert NUM_LABELS == 28
