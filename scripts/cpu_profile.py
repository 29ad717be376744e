"""CPU cost of issuing one training step: tiny batch (GPU work negligible, same launch count) + cProfile.
With autograd multithreading off the backward's Python code runs on the calling thread, where cProfile sees it."""
import sys, tempfile, time, cProfile, pstats, torch
sys.path.insert(0, '.')
from bench import make_fixture
from vqa_amd import synth
from vqa_amd.optim import Trainer
import vqa_model
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = make_fixture(tmp, 10000, 3129)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1, 'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu').cuda().train()
tr = Trainer(m)
B = 4
batch = (synth.images(1, B, 224).cuda(), synth.questions(1, B, 14, 10000).cuda(), synth.targets(1, B, 3129).cuda())
def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): tr.step(*batch, flip=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e3 * (t1 - t0) / n, 1e3 * (time.perf_counter() - t0) / n
run(5)
print('B=%d multithreaded autograd: issue %.2f ms/step, done %.2f' % ((B,) + run(30)))
torch.autograd.set_multithreading_enabled(False)
run(5)
print('B=%d single-thread autograd: issue %.2f ms/step, done %.2f' % ((B,) + run(30)))
pr = cProfile.Profile()
pr.enable()
run(30)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(45)
