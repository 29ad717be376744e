"""`import gru` as the reference's drivers do (vqa_model.py:14): the HIP-backed encoder lives in vqa_amd/gru.py."""
from vqa_amd.gru import GRU, NUMBERS  # noqa: F401
