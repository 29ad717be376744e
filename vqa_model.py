"""`from vqa_model import evaluate, VQA` as the reference's main.py:1-2 does: the HIP-backed model lives in
vqa_amd/vqa_model.py."""
from vqa_amd.vqa_model import VQA, VQAModel, SoftScoreBCELoss, evaluate  # noqa: F401
