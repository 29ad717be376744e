"""Which torch device operations are still on the training step?  torch.profiler over three steady-state steps: every
device activity that is not a kernel of libvqa_hip.so, with the ATen op and the Python frame that issued it.
python scripts/torch_ops_on_step.py"""
import os, sys, tempfile, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import vqa_model
from vqa_amd import synth
from vqa_amd.optim import Trainer

dev = torch.device('cuda', 0)
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1,
                           'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu')
model = model.to(dev).train()
tr = Trainer(model)
batch = (synth.images(1234, 256, 224).to(dev), synth.questions(1234, 256, 14, 10000).to(dev),
         synth.targets(1234, 256, 3129).to(dev))
with torch.cuda.stream(tr.stream):
    for _ in range(10):
        tr.step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            tr.step(*batch)
        torch.cuda.synchronize()
ev = prof.events()
by_corr = collections.defaultdict(list)
cnt = collections.Counter()
ndev = 0
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        n = e.name
        ndev += 1
        if 'anonymous namespace' in n or n.startswith('_ZN12_GLOBAL'):
            continue
        cnt[n[:80]] += 1
print(f'{ndev} device activities in 3 steps; those that are not libvqa kernels:')
for n, c in cnt.most_common():
    print(f'{c:4d}  {n}')
# the CPU ops that launched them
print('--- CPU-side ops with device time and a stack')
seen = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and getattr(e, 'device_time_total', 0) > 0 \
            and not e.cpu_children:
        st = [s for s in (e.stack or []) if 'vqa_amd' in s or 'bench.py' in s or 'scripts/' in s]
        seen[(e.name, st[0] if st else '?')] += 1
for (n, s), c in seen.most_common():
    print(f'{c:4d}  {n:28s} {s}')
