"""1-rank RCCL worker for tests/test_gpu_dp.py: the real data-parallel step (hook-driven bucketed all-reduce on the
`nccl` backend, side-stream GRU backward, in-place flat gradients) must reproduce the non-distributed Trainer bit for
bit when the group has one rank (an all-reduce over one rank is the identity)."""
import copy
import os
import pathlib
import sys
import tempfile

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import build_model, load_case  # noqa: E402
from vqa_amd.optim import Trainer  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    c = load_case('h64_p5_max')
    m0 = build_model(c, pathlib.Path(tempfile.mkdtemp()), 'bf16').cuda().train()
    m1 = copy.deepcopy(m0)
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    ref, dp = Trainer(m0), Trainer(m1, force_buckets=True)
    assert dp.buckets.active and not ref.buckets.active
    for it in range(3):                                   # the hooks must re-arm every step
        l0, l1 = ref.step(img, q, t, flip=False), dp.step(img, q, t, flip=False)
        torch.cuda.synchronize()
        assert float(l0) == float(l1), (it, float(l0), float(l1))
    for (k, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(a, b), k
    # evaluate() under an initialised process group: its 3-scalar all-reduce over one rank changes nothing
    import contextlib
    import io
    import vqa_model

    class DS:
        phase, original_length = 'val', 7
    batches = [{'image': c['images'], 'question': c['questions'], 'answer': c['targets']}]
    with contextlib.redirect_stdout(io.StringIO()):
        loss, _, acc = vqa_model.evaluate(batches, m1, vqa_model.SoftScoreBCELoss(), 0.0, DS())
    m1.eval()
    with torch.no_grad():
        lo = m1(img, q)
    from vqa_amd import ops
    assert abs(acc - float(ops.soft_accuracy_sum(lo, t)) / 7) < 1e-9
    assert abs(loss - float(vqa_model.SoftScoreBCELoss()(lo, t))) < 1e-3 * abs(loss)
    print('DP_GPU_OK', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
