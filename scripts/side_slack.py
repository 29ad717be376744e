"""How much slack does the question encoder's stream have?  Events at the end of the recurrence's forward (question
stream) and of the CNN's forward (trainer's stream), and at the end of the BPTT and of the CNN's backward, over steady
steps.   python scripts/side_slack.py [image tokens hidden batch [eval]]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqa_model
from vqa_amd import synth
from vqa_amd.optim import Trainer

dev = torch.device('cuda', 0)
IMG, TOK, HID, BATCH = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (224, 14, 1024, 256)))
with tempfile.TemporaryDirectory() as tmp:
    qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
    torch.manual_seed(42)
    model = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': HID, 'n_layers': 1,
                           'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu')
model = model.to(dev).train()
tr = Trainer(model)
batch = (synth.images(1234, BATCH, IMG).to(dev), synth.questions(1234, BATCH, TOK, 10000).to(dev),
         synth.targets(1234, BATCH, 3129).to(dev))
marks = []
gf, cf = model.gru.forward, model.cnn.forward


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def gru_fwd(*a, **k):
    s = ev()
    out = gf(*a, **k)
    marks.append(('gru_start', s))
    marks.append(('gru_end', ev()))
    return out


def cnn_fwd(*a, **k):
    s = ev()
    out = cf(*a, **k)
    marks.append(('cnn_start', s))
    marks.append(('cnn_end', ev()))
    return out


model.gru.forward, model.cnn.forward = gru_fwd, cnn_fwd
EVAL = len(sys.argv) > 5 and sys.argv[5] == 'eval'
if EVAL:
    model.eval()


def one_step():
    if EVAL:
        with torch.no_grad():
            model(batch[0], batch[1])
    else:
        tr.step(*batch)


with torch.cuda.stream(tr.stream):
    for _ in range(20):
        one_step()
    torch.cuda.synchronize()
    marks.clear()
    t0 = ev()
    for _ in range(5):
        one_step()
    t1 = ev()
    torch.cuda.synchronize()
print(f'{t0.elapsed_time(t1) / 5:.3f} ms/step')
for i in range(0, len(marks), 4):
    d = dict(marks[i:i + 4])
    print('gru fwd %.3f ms (start %+.3f after cnn start), cnn fwd %.3f ms, gru ends %.3f ms before cnn' % (
        d['gru_start'].elapsed_time(d['gru_end']), d['cnn_start'].elapsed_time(d['gru_start']),
        d['cnn_start'].elapsed_time(d['cnn_end']), d['gru_end'].elapsed_time(d['cnn_end'])))
