"""GPU time of each of the first steps after a synchronize (events on the trainer's stream): where does the fixed cost
of a timed region come from?  python scripts/step_series.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
    for _ in range(20):
        tr.step(*batch)
    for rep in range(3):
        torch.cuda.synchronize()
        evs, hs = [], []
        t0 = time.perf_counter()
        for i in range(16):
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); hs.append(time.perf_counter() - t0)
            tr.step(*batch)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); hs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(16)]
        print('gpu ms/step :', ' '.join(f'{g:5.2f}' for g in gpu), f'| wall {t1 * 1e3 / 16:.3f} ms/step')
        print('host issue  :', ' '.join(f'{(hs[i + 1] - hs[i]) * 1e3:5.2f}' for i in range(16)))
