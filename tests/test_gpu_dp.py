"""GPU: the data-parallel training step over RCCL (backend 'nccl') with one rank on one GPU - the exact code path the
driver's multi-GPU bench takes (hooks, bucketed async all-reduce, NCCL-stream ordering against the side stream),
checked against the non-distributed Trainer; plus bench.py launched the way the driver launches it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    return env


def test_rccl_one_rank_step_matches_local():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')], cwd=ROOT, env=_env(29731),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DP_GPU_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_under_torchrun_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: one JSON line with the contract's keys."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1',
                        '--master-addr', '127.0.0.1', '--master-port', '29733', os.path.join(ROOT, 'bench.py'),
                        '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in out, k
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['roofline']['frac'] > 0


@pytest.mark.parametrize('extra', [[], ['--image', '448', '--tokens', '30', '--hidden', '2048']])
def test_full_size_step_with_forced_buckets_under_torchrun(extra):
    """The N > 1 step at the bench's full sizes on one GPU (`bench.py --force-buckets`: hook-driven bucketed all-reduces
    on RCCL's stream beside the trainer's, the weight-gradient and the question encoder's streams - five streams on four
    hardware queues): it must neither hang (a timeout fails the test) nor change the contract's line.  224x224 / H = 1024
    and the 448x448 / T = 30 / H = 2048 variant, whose step once hung when a fifth compute stream was added."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1',
                        '--master-addr', '127.0.0.1', '--master-port', '29735', os.path.join(ROOT, 'bench.py'),
                        '--gpus', '1', '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--force-buckets'] + extra,
                       cwd=ROOT, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['ms_per_step'] > 0
