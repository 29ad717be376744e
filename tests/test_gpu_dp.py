"""GPU: the data-parallel training step over RCCL (backend 'nccl') with one rank on one GPU - the exact code path the
driver's multi-GPU bench takes (hooks, bucketed async all-reduce, NCCL-stream ordering against the side stream),
checked against the non-distributed Trainer; plus bench.py launched the way the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port the OS says is free right now (no hard-coded rendezvous ports)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    return env


def test_rccl_one_rank_step_matches_local():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_gpu_worker.py')], cwd=ROOT, env=_env(_free_port()),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DP_GPU_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_under_torchrun_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: one JSON line with the contract's keys."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
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
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
                        '--gpus', '1', '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--force-buckets'] + extra,
                       cwd=ROOT, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['ms_per_step'] > 0
    assert out['config'].get('force_buckets') is True


class _FakeWork:
    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)      # ProcessGroupNCCL's Work.wait(): a stream wait, no host block


@pytest.mark.parametrize('size,tokens,hidden,batch,fold', [(224, 14, 1024, 256, False), (224, 14, 1024, 256, True),
                                                            (448, 30, 2048, 64, False)])
def test_bucket_stream_ordering_with_a_fake_collective(size, tokens, hidden, batch, fold, tmp_path, monkeypatch):
    """an all-reduce over ONE rank leaves the buffer as it is whenever it runs, so the 1-rank RCCL
    tests cannot fail on a bucket that is reduced before its last writer has finished.  Here the collective is a fake
    with ProcessGroupNCCL's stream semantics that CHANGES the data: it runs on a private stream that first waits for
    the stream current at the call, doubles the bucket in place, and returns a handle whose wait() makes the current
    stream wait for it.  After three Trainer steps every parameter must equal, bit for bit, a non-distributed run whose
    flat gradient is doubled after a full device synchronisation: a gradient kernel (weight-gradient stream, question
    encoder's stream, the trainer's own) that lands after its bucket was "reduced" overwrites the doubled values and
    the parameters differ.  Full bench size (with the question encoder on its own stream and folded onto the weight-
    gradient stream, optim.DP_FOLD_STREAMS) and the 448x448 / H = 2048 variant."""
    import copy
    import torch
    import bench
    import vqa_model
    from vqa_amd import synth
    from vqa_amd import optim
    from vqa_amd.optim import Trainer
    monkeypatch.setattr(optim, 'DP_FOLD_STREAMS', fold)
    C, V = 3129, 10000
    qpath, lpath = bench.make_fixture(str(tmp_path), V, C)
    torch.manual_seed(42)
    m0 = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': hidden, 'n_layers': 1,
                        'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu', precision='bf16').cuda().train()
    m1 = copy.deepcopy(m0)
    batchT = (synth.images(1234, batch, size).cuda(), synth.questions(1234, batch, tokens, V).cuda(),
              synth.targets(1234, batch, C).cuda())
    coll = torch.cuda.Stream()
    calls = []

    def fake_all_reduce(buf):
        coll.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(coll):
            buf.mul_(2.0)
            calls.append(buf.numel())
            return _FakeWork(coll.record_event())

    ref = Trainer(m0)
    dp = Trainer(m1, force_buckets=True, all_reduce=fake_all_reduce)
    assert dp.buckets.active and not ref.buckets.active and getattr(m1, '_dp_fold_streams', False) == fold

    def doubled():                                   # stands in for buckets.wait(): everything done, then g *= 2
        torch.cuda.synchronize()
        ref.flat.grad.mul_(2.0)
    ref.buckets.wait = doubled
    for it in range(3):
        l0, l1 = ref.step(*batchT, flip=False), dp.step(*batchT, flip=False)
        torch.cuda.synchronize()
        assert float(l0) == float(l1), (it, float(l0), float(l1))
        assert float(ref.opt.total_norm) == float(dp.opt.total_norm), it
    assert len(calls) == 9 and sum(calls[:3]) == dp.flat.numel          # three buckets per step, the whole buffer
    for (k, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(a, b), k
    assert torch.equal(ref.flat.grad, dp.flat.grad)
