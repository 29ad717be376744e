"""world_size-2 gloo worker for tests/test_cpu_host.py::test_grad_buckets_gloo_world2 (CPU only).

The HIP model cannot run here, so the compute is the oracle on a tiny CNN-free stand-in: what is under test is
FlatParams + GradBuckets (flat layout, bucket order, hook-driven asynchronous SUM all-reduce)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqa_amd.optim import FlatParams, GradBuckets  # noqa: E402


class Toy(nn.Module):
    """Parameter groups named like the real model's (head / gru.* / cnn.*) so the bucket layout is exercised."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.cnn = nn.Sequential(nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 8))
        self.gru = nn.Linear(6, 8)
        self.fc = nn.Linear(8, 5)

    def forward(self, a, b):
        return self.fc(torch.relu(self.cnn(a)) * torch.tanh(self.gru(b)))


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(123)
    A, Bq, T = torch.randn(8, 12), torch.randn(8, 6), torch.rand(8, 5)
    crit = nn.BCEWithLogitsLoss(reduction='sum')
    # single-process reference: gradient of the SUM loss over the global batch
    ref = Toy()
    crit(ref(A, Bq), T).backward()
    ref_g = {n: p.grad.clone() for n, p in ref.named_parameters()}
    # data parallel: each rank takes its shard
    m = Toy()
    flat = FlatParams(m)
    buckets = GradBuckets(flat)
    assert len(flat.bucket_ranges) == 3 and flat.names[0].startswith('fc') and flat.names[-1].startswith('cnn')
    sh = slice(rank * 4, rank * 4 + 4)
    for it in range(2):                      # twice: the hooks must re-arm
        flat.zero_grad()
        crit(m(A[sh], Bq[sh]), T[sh]).backward()
        buckets.wait()
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, ref_g[n], atol=1e-5), (n, it)
            assert p.grad.data_ptr() >= flat.grad.data_ptr()
    print(f'DP_OK rank{rank}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
