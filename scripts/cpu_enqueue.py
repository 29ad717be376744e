import sys, tempfile, time, torch
sys.path.insert(0, '.')
from bench import make_fixture
from vqa_amd import synth
from vqa_amd.optim import Trainer
import vqa_model
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = make_fixture(tmp, 10000, 3129)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1, 'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu').cuda().train()
tr = Trainer(m)
B = 256
batch = (synth.images(1, B, 224).cuda(), synth.questions(1, B, 14, 10000).cuda(), synth.targets(1, B, 3129).cuda())
for _ in range(5): tr.step(*batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): tr.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step')
