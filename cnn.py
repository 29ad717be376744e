"""`import cnn` as the reference's drivers do (vqa_model.py:13): the HIP-backed encoder lives in vqa_amd/cnn.py."""
from vqa_amd.cnn import CNN  # noqa: F401
