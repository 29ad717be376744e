"""world_size-2 gloo worker for tests/test_cpu_host.py::test_grad_buckets_gloo_world2 (CPU only).

The DP parity definition of DESIGN.md section 5 is what is checked: every rank runs the ORACLE's forward + backward
(oracle/vqa_oracle.py, BatchNorm statistics per rank) on its shard of the global batch, over the parameters of the
drop-in module re-pointed into the flat buffers (FlatParams) with the hook-driven bucketed all-reduce armed
(GradBuckets); the reduced gradient must equal the SUM of the per-shard oracle gradients (the loss is a sum,
vqa_model.py:215).  The HIP model cannot run here; its module is used as the parameter container only."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vqa_oracle as O  # noqa: E402
from tests.helpers import build_model  # noqa: E402
from vqa_amd import synth  # noqa: E402
from vqa_amd.optim import FlatParams, GradBuckets  # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(2)
    tmp = os.path.join(sys.argv[1], f'rank{rank}')
    os.makedirs(tmp, exist_ok=True)
    H, C, V, per = 32, 20, 82, 2
    case = dict(hidden=H, classes=C, padding=2, pooling='max', vocab=V, sd=synth.state_dict(7, H, C, V))
    img, q, t = synth.images(3, per * world, 224), synth.questions(3, per * world, 6, V), synth.targets(3, per * world, C)
    # the reference: the sum over shards of the oracle's per-shard gradients (every rank computes it in full)
    want = None
    for r in range(world):
        sh = slice(r * per, (r + 1) * per)
        _, _, g, _ = O.train_step_grads(case['sd'], img[sh].float(), q[sh], t[sh], 2, 'max')
        want = g if want is None else {k: want[k] + g[k] for k in g}
    # data parallel: this rank's shard, gradients into the flat buffer, buckets reduced from the hooks
    m = build_model(case, tmp, 'fp32')                       # parameter container (CPU: its forward is never called)
    flat = FlatParams(m)
    buckets = GradBuckets(flat)
    assert buckets.active and len(flat.bucket_ranges) == 3
    assert flat.names[0].startswith('linear') and flat.names[-1].startswith('cnn')
    sh = slice(rank * per, (rank + 1) * per)
    sd = dict(m.named_parameters())
    sd.update(dict(m.named_buffers()))
    for it in range(2):                                      # twice: the hooks must re-arm
        flat.zero_grad()
        O.bce_sum(O.forward(sd, img[sh].float(), q[sh], 2, 'max', True), t[sh]).backward()
        buckets.wait()
        for n, p in m.named_parameters():
            ref = want[n]
            err = float((p.grad - ref).norm() / (ref.norm() + 1e-12))
            assert err < 1e-5 or float(ref.norm()) < 1e-6, (n, it, err)
            assert p.grad.data_ptr() == p._vqa_grad_dst.data_ptr()
    # a parameter that gets no gradient in a pass (an unused branch): its bucket's countdown never reaches zero - wait()
    # must still reduce that bucket, or the ranks would silently keep different gradients
    m.fc.weight_g.requires_grad_(False)
    flat.zero_grad()
    O.bce_sum(O.forward(sd, img[sh].float(), q[sh], 2, 'max', True), t[sh]).backward()
    buckets.wait()
    for n, p in m.named_parameters():
        if n == 'fc.weight_g':
            assert float(p._vqa_grad_dst.abs().max()) == 0.0
        else:
            ref = want[n]
            assert float((p.grad - ref).norm() / (ref.norm() + 1e-12)) < 1e-5 or float(ref.norm()) < 1e-6, n
    print(f'DP_OK rank{rank}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
