"""Per-step time of the first 30 steps of a cold process (wall clock with a synchronize after every step, and the
memory the caching allocator has reserved): what does the driver's `--steps 20 --warmup 5` window still contain?"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, vqa_model
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
mode = sys.argv[1] if len(sys.argv) > 1 else 'sync'
tr.stream.wait_stream(torch.cuda.current_stream()); torch.cuda.set_stream(tr.stream)
if mode == 'sync':
    for i in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.step(*batch)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'step {i:2d}: {dt * 1e3:8.3f} ms   reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB', flush=True)
elif mode == 'events':   # no synchronize at all: host issue time and GPU time (events) of each of the first 40 steps
    evs, hs = [], []
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    for i in range(40):
        t0 = time.perf_counter()
        tr.step(*batch)
        hs.append(time.perf_counter() - t0)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    torch.cuda.synchronize()
    for i in range(1, 40):
        print(f'step {i:2d}: gpu {evs[i].elapsed_time(evs[i + 1]):7.3f} ms   host issue (incl. run-ahead wait) {hs[i] * 1e3:7.3f} ms')
else:       # the driver's shape: 5 warm-up steps, synchronize, then blocks of 5 steps timed without synchronizing inside
    for _ in range(5):
        tr.step(*batch)
    torch.cuda.synchronize()
    for blk in range(6):
        t0 = time.perf_counter()
        for _ in range(5):
            tr.step(*batch)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'steps {5 + 5 * blk:2d}-{9 + 5 * blk:2d}: {dt * 1e3 / 5:8.3f} ms/step   reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB', flush=True)
