"""GPU: the whole drop-in VQA module (HIP kernels through the C-ABI) against the golden vectors captured from the
reference and against the oracle on the same synthetic inputs and weights."""
import os
import numpy as np
import pytest
import torch

from tests.helpers import CONFIGS, VARIANT_CONFIGS, build_model, load_case, rel_err, sample_idx

pytestmark = pytest.mark.gpu


# (the B=32 case has its own, tighter test: tests/test_gpu_train_parity.py)
@pytest.mark.parametrize('name', [n for n in CONFIGS if not n.endswith('_b32')])
def test_fp32_model_vs_golden(name, tmp_path):
    """fp32 path: logits / loss <= 1e-4 relative of the reference; GRU/head gradients <= 2e-3 (weight_g: scalar sums
    with cancellation), CNN gradients within the ReLU/arg-max flip band (3e-2 at these tiny batches)."""
    import vqa_model
    from oracle import vqa_oracle as O
    c = load_case(name)
    g = c['golden']
    m = build_model(c, tmp_path, 'fp32').cuda()
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    m.eval()
    with torch.no_grad():
        lo = m(img, q)
    assert rel_err(lo.cpu(), g['eval_logits']) < 1e-4
    from vqa_amd import ops
    assert float(ops.soft_accuracy_sum(lo, t)) == pytest.approx(float(g['eval_acc_sum']), abs=1e-6)
    m.train()
    crit = vqa_model.SoftScoreBCELoss()
    logits = m(img, q)
    loss = crit(logits, t)
    loss.backward()
    assert rel_err(logits.detach().cpu(), g['logits']) < 1e-4
    # the reference's own fp32 Frobenius norm (torch CPU norm()) is 5e-5 off on fc.weight_v; that scales every logit
    # the same way and moves the summed loss by ~1e-4 (oracle header).  Against the oracle's accurate norm: 2e-5.
    assert abs(float(loss.detach()) - float(g['loss'])) < 3e-4 * abs(float(g['loss']))
    o_loss, o_logits, _, _ = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'], c['padding'],
                                                c['pooling'])
    assert rel_err(logits.detach().cpu(), o_logits) < 2e-5
    assert abs(float(loss.detach()) - float(o_loss)) < 2e-5 * abs(float(o_loss))
    for k, b in m.named_buffers():
        if 'running' in k:
            assert rel_err(b.cpu(), g[f'stat/{k}']) < 1e-4, k
    worst = {}
    for k, p in m.named_parameters():
        flat = p.grad.detach().reshape(-1).cpu()
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            assert float(flat.abs().max()) == 0.0
            continue
        e = rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}'])
        grp = 'cnn' if k.startswith('cnn.') else ('weight_g' if k.endswith('weight_g') else 'gru+head')
        worst[grp] = max(worst.get(grp, 0.0), e)
        tol = 3e-2 if grp == 'cnn' else (2e-3 if grp == 'weight_g' else 5e-4)
        assert e < tol, (k, e)
    print(f'\n[{name}] logits {rel_err(logits.detach().cpu(), g["logits"]):.2e} worst grads {worst}')


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('name', list(VARIANT_CONFIGS))
def test_variants_vs_golden(name, prec, tmp_path):
    """activation='selu' (nn.SELU + nn.AlphaDropout(0.5), vqa_model.py:54-55; the reference's recorded keep-masks
    replayed through forward(dropout_noise=...)) and n_layers=2 (stacked nn.GRU, gru.py:51) against the golden vectors
    of the imported reference.  fp32: the bounds of test_fp32_model_vs_golden.  bf16: outputs only - at B=4 the bf16
    image encoder's training-mode output moves every gradient behind it; the bf16 gradients of these two
    branches are pinned where that noise is absent: the head on given encoder outputs
    (test_gpu_train_parity.py::test_head_fwd_bwd_vs_oracle, selu cases) and the stacked encoder alone
    (test_gpu_gru.py::test_gru_vs_oracle, layers > 1)."""
    import vqa_model
    from oracle import vqa_oracle as O
    c = load_case(name)
    g = c['golden']
    m = build_model(c, tmp_path, prec).cuda()
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    tol_out = 1e-4 if prec == 'fp32' else 3e-2
    m.eval()
    with torch.no_grad():
        lo = m(img, q)
    assert rel_err(lo.cpu(), g['eval_logits']) < tol_out
    m.train()
    noise = c.get('noise')
    if noise is not None:
        noise = [tuple(x.cuda() for x in st) if isinstance(st, (tuple, list)) else st.cuda() for st in noise]
    logits = m(img, q, dropout_noise=noise)
    loss = vqa_model.SoftScoreBCELoss()(logits, t)
    loss.backward()
    assert rel_err(logits.detach().cpu(), g['logits']) < tol_out
    assert abs(float(loss.detach()) - float(g['loss'])) < (3e-4 if prec == 'fp32' else 3e-2) * abs(float(g['loss']))
    worst = {}
    for k, p in m.named_parameters():
        flat = p.grad.detach().reshape(-1).cpu()
        assert bool(torch.isfinite(flat).all()), k
        if prec == 'bf16' or (k.startswith('cnn.') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX):
            continue
        e = rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}'])
        grp = 'cnn' if k.startswith('cnn.') else ('weight_g' if k.endswith('weight_g') else 'gru+head')
        worst[grp] = max(worst.get(grp, 0.0), e)
        tol = 3e-2 if grp == 'cnn' else (2e-3 if grp == 'weight_g' else 5e-4)
        assert e < tol, (k, e)
    print(f'\n[{name} {prec}] logits {rel_err(logits.detach().cpu(), g["logits"]):.2e} worst grads {worst}')
    if c['activation'] != 'relu':
        # module-drawn AlphaDropout (no injected noise): training output differs from eval, p is 0.5 whatever `dropout`
        with torch.no_grad():
            a, b = m(img, q), m(img, q)
        assert not torch.equal(a, b)
        mk, ad = m._dropout_mask(4, img.device)
        assert 0.3 < float((mk != 0).float().mean()) < 0.7 and ad is not None


def test_eval32_soft_accuracy_identical(tmp_path):
    """VQA soft accuracy on the fixed synthetic eval set is identical to the reference's (fp32 and bf16 paths)."""
    from vqa_amd import ops, synth
    c = load_case('h64_p5_max')
    img = synth.images(1235, 32, 224).cuda()
    q = synth.questions(1235, 32, c['tokens'], c['vocab']).cuda()
    t = synth.targets(1235, 32, c['classes']).cuda()
    ref = c['golden']['eval32_logits']
    # logit margin of the reference's top-1 on this set: bf16 may only flip an arg-max inside its error band
    srt = np.sort(ref, axis=1)
    margin = float((srt[:, -1] - srt[:, -2]).min())
    for prec in ('fp32', 'bf16'):
        m = build_model(c, tmp_path, prec).cuda().eval()
        with torch.no_grad():
            lo = m(img, q)
        err = float(np.abs(lo.cpu().numpy() - ref).max())
        print(f'\n[{prec}] max |logit err| {err:.3e}  min top-1 margin {margin:.3e}')
        # bf16 may only keep every arg-max if its logit error stays inside half the smallest top-1 margin: asserted,
        # not assumed (a bf16 regression past the margin must fail here, not skip the accuracy check)
        assert err < margin / 2, (prec, err, margin)
        assert float(ops.soft_accuracy_sum(lo, t)) == pytest.approx(float(c['golden']['eval32_acc_sum']), abs=1e-6)


def test_evaluate_loop(tmp_path):
    """evaluate() keeps the reference's signature / return tuple (vqa_model.py:106-141)."""
    import vqa_model
    from vqa_amd import synth
    c = load_case('h64_p5_max')
    m = build_model(c, tmp_path, 'fp32').cuda()

    class DS:
        phase, original_length = 'val', 40
    batches = []
    for i in range(2):
        batches.append({'image': synth.images(1235, 32, 224)[16 * i:16 * i + 16],
                        'question': synth.questions(1235, 32, c['tokens'], c['vocab'])[16 * i:16 * i + 16],
                        'answer': synth.targets(1235, 32, c['classes'])[16 * i:16 * i + 16]})
    loss, not_improved, acc = vqa_model.evaluate(batches, m, vqa_model.SoftScoreBCELoss(), 0.0, DS())
    assert acc == pytest.approx(float(c['golden']['eval32_acc_sum']) / 40, abs=1e-6)
    assert not_improved is False and loss > 0


def test_pickle_roundtrip(tmp_path):
    """The reference checkpoints by pickling the whole module (vqa_model.py:296): no native handle may live on it."""
    c = load_case('h64_p5_max')
    m = build_model(c, tmp_path, 'fp32').cuda()
    path = tmp_path / 'm.pth'
    torch.save(m, path)
    m2 = torch.load(path, weights_only=False)
    with torch.no_grad():
        a = m.eval()(c['images'].cuda(), c['questions'].cuda())
        b = m2.eval()(c['images'].cuda(), c['questions'].cuda())
    assert torch.equal(a, b)


def test_cpu_tensor_raises(tmp_path):
    from vqa_amd._lib import VqaHipError
    c = load_case('h64_p5_max')
    m = build_model(c, tmp_path, 'fp32')
    with pytest.raises(VqaHipError):
        m(c['images'], c['questions'])


def test_trainer_flat_gradients_and_step(tmp_path):
    """optim.Trainer: the HIP backward writes every gradient straight into the flat buffer (autograd adopts the views:
    no accumulation kernels, no copies), the result equals a plain loss.backward() on the same module, and the
    clip + Adamax tail reproduces the reference's post-step parameters (golden psamp/*, total_norm)."""
    import copy
    import vqa_model
    from oracle import vqa_oracle as O
    from vqa_amd.optim import Trainer
    c = load_case('h64_p5_max')
    g = c['golden']
    m = build_model(c, tmp_path, 'fp32').cuda().train()
    m2 = copy.deepcopy(m)
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    vqa_model.SoftScoreBCELoss()(m2(img, q), t).backward()           # plain autograd: fresh gradient tensors
    ref = {k: p.grad.clone() for k, p in m2.named_parameters()}
    tr = Trainer(m, lr=2e-3, max_norm=0.25)
    copies = []
    orig = tr.flat.collect
    def spy(params=None):
        copies.extend(p for p in tr.flat.params if p.grad is None or p.grad.data_ptr() != p._vqa_grad_dst.data_ptr())
        return orig(params)
    tr.flat.collect = spy
    tr.step(img, q, t, flip=False)
    assert not copies, 'some gradients were not written in place'
    names = dict(zip(tr.flat.names, tr.flat.params))
    for k, r in ref.items():
        got = names[k]._vqa_grad_dst.view(r.shape)
        assert torch.equal(got, r), k                        # same kernels, same bits
    assert float(tr.opt.total_norm) == pytest.approx(float(g['total_norm']), rel=1e-3)
    for k, p in m.named_parameters():
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            continue
        d = np.abs(p.detach().reshape(-1).cpu().numpy()[sample_idx(p.numel())] - g[f'psamp/{k}'])
        assert (d > 1e-4).mean() <= 0.07, k
    # a second step: gradients are overwritten, not accumulated
    tr.step(img, q, t, flip=False)
    assert not copies


def test_sparse_targets_match_dense(tmp_path):
    """Sparse (labels, scores) targets: loss, dlogits and soft accuracy equal the dense [B,C] path
    and the oracle's BCE on the densified matrix."""
    import vqa_model
    from oracle import vqa_oracle as O
    from vqa_amd import ops, synth
    B, C, L = 37, 3129, 10
    torch.manual_seed(3)
    logits = (torch.randn(B, C) * 3).cuda().requires_grad_(True)
    dense = synth.targets(77, B, C)
    labels = torch.full((B, L), -1, dtype=torch.int64)
    scores = torch.zeros(B, L)
    for b in range(B):
        nz = dense[b].nonzero().flatten()
        labels[b, :len(nz)] = nz
        scores[b, :len(nz)] = dense[b, nz]
    with torch.no_grad():                                  # make every row's arg-max a scored label now and then
        for b in range(0, B, 3):
            logits[b, labels[b, 0]] = 50.0
    crit = vqa_model.SoftScoreBCELoss()
    ld = crit(logits, dense.cuda())
    gd, = torch.autograd.grad(ld, logits)
    ls = crit(logits, (labels.cuda(), scores.cuda()))
    gs, = torch.autograd.grad(ls, logits)
    assert float(ls.detach()) == pytest.approx(float(ld.detach()), rel=1e-6)
    assert torch.allclose(gs, gd, atol=1e-7)
    assert float(ls.detach()) == pytest.approx(float(O.bce_sum(logits.detach().cpu(), dense)), rel=1e-5)
    a_d = float(ops.soft_accuracy_sum(logits.detach(), dense.cuda()))
    a_s = float(ops.soft_accuracy_sum(logits.detach(), (labels.cuda(), scores.cuda())))
    assert a_s == pytest.approx(a_d, abs=1e-6) and a_d > 0


def test_softscore_targets_feed_the_loss_kernel():
    """Annotations -> vqa_amd.softscore -> packed sparse targets -> HIP loss / accuracy == the dense rows the
    reference's dataset would build (dataset.py:135-138) through the oracle's BCE."""
    import json
    import vqa_model
    from oracle import vqa_oracle as O
    from vqa_amd import ops, softscore as S
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'softscore.json')) as f:
        gold = json.load(f)
    a2l, l2a, train_t, _ = S.prepare(gold['train_annotations'], gold['val_annotations'])
    kept, _, _ = S.drop_unanswered(train_t, list(range(len(train_t))))
    labels, scores = S.pack_sparse(kept)
    dense = torch.from_numpy(S.densify(kept, len(l2a)))
    torch.manual_seed(5)
    logits = (torch.randn(len(kept), len(l2a)) * 2).cuda()
    loss = vqa_model.SoftScoreBCELoss()(logits, (torch.from_numpy(labels).cuda(), torch.from_numpy(scores).cuda()))
    assert float(loss) == pytest.approx(float(O.bce_sum(logits.cpu(), dense)), rel=1e-5)
    acc = float(ops.soft_accuracy_sum(logits, (torch.from_numpy(labels).cuda(), torch.from_numpy(scores).cuda())))
    want = float(dense[torch.arange(len(kept)), logits.cpu().argmax(1)].sum())
    assert acc == pytest.approx(want, abs=1e-5)


@pytest.mark.parametrize('B,T,H,W', [(1, 1, 224, 224), (3, 5, 200, 260), (5, 14, 224, 224), (2, 30, 190, 190)])
def test_edge_shapes_vs_oracle(B, T, H, W, tmp_path):
    """Ragged ends of the input space: one sample, one token, non-square / non-224 images (odd maps under the floor-mode
    pool), questions that are all <pad>.  fp32 path against the oracle: eval logits, train logits, loss, GRU/head
    gradients."""
    import vqa_model
    from oracle import vqa_oracle as O
    from vqa_amd import synth
    c = load_case('h64_p5_max')
    m = build_model(c, tmp_path, 'fp32').cuda()
    S = max(H, W)
    img = synth.images(77, B, S)[:, :, :H, :W].contiguous()
    q = synth.randint(77, 'edge_q', (B, T), 2, c['vocab'])
    q[0, T // 2:] = 1                                        # <pad> tail (gru.py:94-97) ...
    q[-1, :] = 1                                             # ... and a question of nothing but <pad>
    t = synth.targets(77, B, c['classes'])
    m.eval()
    with torch.no_grad():
        lo = m(img.cuda(), q.cuda())
    want = O.forward(c['sd'], img.float(), q, c['padding'], c['pooling'], training=False)
    assert lo.shape == (B, c['classes']) and rel_err(lo.cpu(), want) < 2e-5
    m.train()
    logits = m(img.cuda(), q.cuda())
    loss = vqa_model.SoftScoreBCELoss()(logits, t.cuda())
    loss.backward()
    o_loss, o_logits, o_grads, _ = O.train_step_grads(c['sd'], img.float(), q, t, c['padding'], c['pooling'])
    assert rel_err(logits.detach().cpu(), o_logits) < 2e-5
    assert abs(float(loss.detach()) - float(o_loss)) < 2e-5 * abs(float(o_loss))
    for k, p in m.named_parameters():
        if k.startswith('cnn.'):
            continue                                         # ReLU / arg-max flip band at tiny batches
        tol = 2e-3 if k.endswith('weight_g') else 5e-4
        assert rel_err(p.grad.detach().cpu(), o_grads[k]) < tol, k


@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
def test_mirror_equals_flipped_batch(prec, tmp_path):
    """model(images, q, mirror=True) == model(images.flip(-1), q): logits and every gradient, bit for bit (the bf16
    path reads the images mirrored inside the first-layer kernels, the fp32 path flips the batch)."""
    import vqa_model
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    out = []
    for mirror in (True, False):
        m = build_model(c, tmp_path, prec).cuda().train()
        logits = m(img, q, mirror=True) if mirror else m(img.flip(-1).contiguous(), q)
        vqa_model.SoftScoreBCELoss()(logits, t).backward()
        out.append((logits.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    assert torch.equal(out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
def test_side_streams_do_not_change_results(prec, tmp_path, monkeypatch):
    """Question encoder, conv weight gradients and head parameter gradients on their side streams (the default) against
    everything on the caller's stream: logits and every gradient bit for bit, read right after backward() returns
    (plain loss.backward(), no Trainer: the end-of-backward callback of ops.defer_join is what joins the streams)."""
    import vqa_model
    from vqa_amd import cnn as CN, vqa_model as VM
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    out = []
    for streams in (True, False):
        monkeypatch.setattr(VM, 'SIDE_STREAM', streams)
        monkeypatch.setattr(VM, 'HEAD_WGRAD_STREAM', streams)
        monkeypatch.setattr(CN, 'WGRAD_STREAM', streams)
        m = build_model(c, tmp_path, prec).cuda().train()
        for _ in range(3):                                       # repeated: a missed join shows as a stale gradient
            m.zero_grad(set_to_none=True)
            logits = m(img, q)
            vqa_model.SoftScoreBCELoss()(logits, t).backward()
        out.append((logits.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    assert torch.equal(out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('sparse', [False, True])
def test_fused_classifier_loss_equals_the_criterion(prec, sparse, tmp_path):
    """VQA.forward_loss (the classifier GEMM with the sigmoid-BCE epilogue, vqa_model.py:103 + :215) against forward() +
    SoftScoreBCELoss: the same logits bit for bit, the same gradient of the loss w.r.t. the logits bit for bit (hence
    every parameter gradient), the loss up to the summation order of its partials; with a loss scale other than 1 and
    with dense and sparse targets; and against the oracle's BCE."""
    import vqa_model
    from oracle import vqa_oracle as O
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    tgt = t
    if sparse:
        L_ = 10
        labels = torch.full((t.shape[0], L_), -1, dtype=torch.int64)
        scores = torch.zeros(t.shape[0], L_)
        for b in range(t.shape[0]):
            nz = c['targets'][b].nonzero().flatten()
            labels[b, :len(nz)] = nz
            scores[b, :len(nz)] = c['targets'][b, nz]
        tgt = (labels.cuda(), scores.cuda())
    out = []
    for fused in (True, False):
        m = build_model(c, tmp_path, prec).cuda().train()
        if fused:
            logits, loss = m.forward_loss(img, q, tgt)
        else:
            logits = m(img, q)
            loss = vqa_model.SoftScoreBCELoss()(logits, tgt)
        (loss * 0.37).backward()
        out.append((logits.detach(), float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == pytest.approx(out[1][1], rel=1e-6)
    assert out[0][1] == pytest.approx(float(O.bce_sum(out[0][0].cpu(), c['targets'])), rel=1e-5)
    for k in out[0][2]:
        a, b = out[0][2][k], out[1][2][k]
        # (the unfused path multiplies sigmoid(x) - t by the loss scale in fp32 before the bf16 cast, the fused one
        # inside the cast: the same fp32 product, the same rounding)
        assert torch.equal(a, b), k
