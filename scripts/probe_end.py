"""Which stream finishes the backward last?  Events recorded on the trainer's stream, the weight-gradient stream and the
question encoder's stream at the moment loss.backward() has been issued, read against each other over steady-state steps."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, vqa_model
from vqa_amd import synth, ops, cnn as CN, vqa_model as VM, optim
from vqa_amd.optim import Trainer
dev = torch.device('cuda', 0)
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1,
                           'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu')
model = model.to(dev).train()
tr = Trainer(model)
batch = (synth.images(1234, 256, 224).to(dev), synth.questions(1234, 256, 14, 10000).to(dev), synth.targets(1234, 256, 3129).to(dev))
tr.stream.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(tr.stream)
probe = []
orig = ops.join_deferred
state = {'on': False}
def patched(lo=None, hi=None):
    if state['on'] and lo is None:
        evs = []
        for s in (torch.cuda.current_stream(), CN._wgrad_stream(dev), VM._side_stream(dev)):
            e = torch.cuda.Event(enable_timing=True); e.record(s); evs.append(e)
        probe.append(evs)
    return orig(lo, hi)
ops.join_deferred = patched
optim.ops.join_deferred = patched
for _ in range(30): tr.step(*batch)
state['on'] = True
for _ in range(40): tr.step(*batch)
torch.cuda.synchronize()
rows = [(m.elapsed_time(w) * 1e3, m.elapsed_time(q) * 1e3) for m, w, q in probe[5:]]
print('per step, relative to the trainer stream finishing its backward (us; > 0 = finishes later):')
print(' weight-gradient stream:', ' '.join(f'{a:.0f}' for a, _ in rows))
print(' question stream       :', ' '.join(f'{b:.0f}' for _, b in rows))
