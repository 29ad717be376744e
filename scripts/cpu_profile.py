"""CPU cost of issuing one training step: tiny batch (GPU work negligible, same launch count) + cProfile."""
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
for _ in range(5): tr.step(*batch, flip=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.step(*batch, flip=False)
torch.cuda.synchronize()
print(f'B={B}: {1e3*(time.perf_counter()-t0)/30:.2f} ms/step (CPU-bound)')
pr = cProfile.Profile()
pr.enable()
for _ in range(30): tr.step(*batch, flip=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
