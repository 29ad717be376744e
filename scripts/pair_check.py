"""Eval-mode sample independence across the paired / unpaired conv tile forms: logits of the first 255 samples with
B = 256 (two images per tile in the last conv) and B = 255 (odd: one image per tile) must be identical."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, vqa_model
from vqa_amd import synth
from vqa_amd import cnn as C

dev = torch.device('cuda', 0)
for prec in ('fp32', 'bf16'):
    with tempfile.TemporaryDirectory() as tmp:
        qpath, lpath = bench.make_fixture(tmp, 10000, 3129)
        torch.manual_seed(42)
        m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 1024, 'n_layers': 1,
                           'train_question_path': qpath}, lpath, 256, 2, 0.0, 'max', 'relu', precision=prec)
    m = m.to(dev).eval()
    img, q = synth.images(5, 256, 224).to(dev), synth.questions(5, 256, 14, 10000).to(dev)
    with torch.no_grad():
        a = m(img, q)
        b = m(img[:255].contiguous(), q[:255].contiguous())
        v1 = m.cnn(img)
        v2 = m.cnn(img[:255].contiguous())
    print(prec, 'logits max diff', float((a[:255] - b).abs().max()), 'cnn max diff', float((v1[:255] - v2).abs().max()),
          'scale', float(a.abs().max()), float(v1.abs().max()))
    # the last conv alone, eval form (relu, no statistics), random data
    dt = torch.float32 if prec == 'fp32' else torch.bfloat16
    torch.manual_seed(1)
    x = torch.randn(6, 12, 12, 256, device=dev).to(dt)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
    bias = torch.randn(256, device=dev)
    wpk = C.pack_weights(w, 256, dt, dgrad=False)
    y6, _ = C.conv_fwd(x, wpk, bias, 256, 0, False, relu=True)
    y5, _ = C.conv_fwd(x[:5].contiguous(), wpk, bias, 256, 0, False, relu=True)
    print(prec, 'conv relu 6 vs 5:', float((y6[:5].float() - y5.float()).abs().max()))
