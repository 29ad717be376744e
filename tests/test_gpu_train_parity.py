"""GPU: parity of the TRAINING path at the benchmark's precision and model size.

* the fusion head in bf16 (attn_bwd / hadamard_bwd / relu_bwd / group_sum / the TN GEMMs in their bf16 instantiations)
  against the oracle's autograd on given encoder outputs;
* the whole model, fp32 and bf16, against the B=32 golden captured from the imported reference (H=1024, padding 2,
  224x224): at this batch the ReLU / arg-max flips of the tiny-batch goldens average out and the bounds tighten;
* dropout p = 0.0381 (the authors' final config) with the reference's recorded keep-masks replayed;
* three clip + Adamax steps against the reference's parameters, and the fused optimizer element-wise against
  torch.optim.Adamax + clip_grad_norm_ (the calls at vqa_model.py:259, :218, :266)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, load_case, rel_err, sample_idx

pytestmark = pytest.mark.gpu


def _is_conv_bias(k):
    from oracle import vqa_oracle as O
    return k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX


# bf16 bounds per head tensor (measured in brackets): activations / weights are rounded to 8 mantissa bits at every
# GEMM input, accumulation is fp32
# (att = gradients that pass through the region softmax: dv, dq, the two attention layers)
# grad: a bf16 forward moves pre-activations by ~2e-3 of their scale, which flips the ReLU mask of ~1e-3 of the units;
# the mask is binary in the gradient, so weight gradients move by ~sqrt(1e-3) = 3e-2 [measured 2.1e-2 .. 3.8e-2]
HEAD_BF16 = dict(logits=1.5e-2, dv=6e-2, dq=6e-2, att=1e-1, grad=6e-2, weight_g=2e-2)


@pytest.mark.parametrize('K,H,B,act', [(25, 1024, 32, 'relu'), (144, 512, 24, 'relu'), (25, 512, 40, 'relu'),
                                       (144, 1024, 16, 'relu'),
                                       # the reference's other branch (vqa_model.py:54-55): nn.SELU at the four
                                       # activation sites, nn.AlphaDropout(p=0.5) noise at the two dropout sites
                                       (25, 1024, 32, 'selu'), (144, 512, 24, 'selu')])
@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
def test_head_fwd_bwd_vs_oracle(prec, K, H, B, act, tmp_path):
    """_HeadFunction on GIVEN fp32 v [B,k,256], q [B,H]: logits, dv, dq and all 16 head parameter gradients against
    oracle.head_forward autograd (vqa_model.py:83-103) with the sum-BCE loss on top (:215)."""
    import vqa_model
    from oracle import vqa_oracle as O
    from vqa_amd import synth
    C = 3129
    case = dict(hidden=H, classes=C, padding=2, pooling='max', vocab=82, sd=synth.state_dict(42, H, C, 82),
                activation=act)
    m = build_model(case, tmp_path, prec).cuda().train()
    noise = noise_dev = None
    if act != 'relu':
        noise = [O.alpha_dropout_noise(synth.randint(5, f'head.keep{i}', (B, H), 0, 2)) for i in (1, 2)]
        noise_dev = [tuple(x.cuda() for x in st) for st in noise]
    v = torch.relu(synth.uniform(5, 'head.v', (B, K, 256), -0.6, 1.4))      # pooled post-ReLU features, some zeros
    q = synth.uniform(5, 'head.q', (B, H), -1.0, 1.0)                       # GRU hidden state range
    t = synth.targets(5, B, C)
    p = {k: (x.clone().requires_grad_(True) if O.is_param(k) else x) for k, x in case['sd'].items()
         if not k.startswith(('cnn.', 'gru.'))}
    vr, qr = v.clone().requires_grad_(True), q.clone().requires_grad_(True)
    lo_ref = O.head_forward(p, vr, qr, masks=noise, activation=act)
    O.bce_sum(lo_ref, t).backward()
    vd, qd = v.cuda().requires_grad_(True), q.cuda().requires_grad_(True)
    lo = m.head(vd, qd, dropout_noise=noise_dev)
    vqa_model.SoftScoreBCELoss()(lo, t.cuda()).backward()
    tol = HEAD_BF16 if prec == 'bf16' else dict(logits=2e-5, dv=2e-5, dq=2e-5, grad=5e-5, weight_g=2e-5)
    if prec == 'bf16' and act != 'relu':
        # SELU's derivative is formed from the bf16-stored activation over its whole negative half (ReLU's is a sign),
        # and half the units are replaced by AlphaDropout's constant [measured att 8.6e-2 .. 9.8e-2]
        tol = dict(tol, att=1.5e-1)
    errs = dict(logits=rel_err(lo.detach().cpu(), lo_ref.detach()), dv=rel_err(vd.grad.cpu(), vr.grad),
                dq=rel_err(qd.grad.cpu(), qr.grad))
    for k, e in errs.items():
        assert e < tol[k], (k, e)
    worst = {}
    for k, prm in m.named_parameters():
        if k.startswith(('cnn.', 'gru.')):
            continue
        if k.endswith('weight_g'):        # a dot product with cancellation: error against the scale of its terms
            kv = k[:-1] + 'v'
            scale = float(p[kv].grad.norm()) * float(case['sd'][kv].norm()) / abs(float(case['sd'][k]))
            e, grp = abs(float(prm.grad) - float(p[k].grad)) / scale, 'weight_g'
        else:
            e = rel_err(prm.grad.cpu(), p[k].grad)
            grp = 'att' if (prec == 'bf16' and k.startswith(('linear_inside_relu_attention',
                                                             'linear_after_relu_attention'))) else 'grad'
        worst[grp] = max(worst.get(grp, 0.0), e)
        assert e < tol[grp], (k, e)
    print(f'\n[head {prec} {act} K={K} H={H} B={B}] ' + ' '.join(f'{k} {e:.2e}' for k, e in {**errs, **worst}.items()))


# ---- whole model against the B=32 golden.
# fp32 bounds per parameter group (measured on MI355X in brackets).  Two groups are ill-conditioned in this synthetic-
# weight model, for the reference as much as for this path:
#   cnn        the reference's own fp32 gradients sit 4-5e-3 from an fp64 run of the same algorithm (ReLU / arg-max
#              decisions flipping on 1e-7 forward differences; scripts/diag_b32.py)                      [<= 3.7e-3]
#   attention  ONE of the layer's 819,200 ReLU units has a pre-activation of 4.7e-9 and lands on the other side of zero
#              than in the reference's run; it is worth 2.0e-3 on the bias gradient (scripts/fp64_anchor.py,
#              test_fp32_gradients_are_as_close_to_an_fp64_run_as_the_reference_is); on the REFERENCE's encoder
#              outputs the head alone agrees with the oracle to 3e-6 (asserted below)                    [<= 2.0e-3]
#   rest       GRU, projections, classifier: 3-8e-5 = the reference's own 5e-5 Frobenius-norm error     [<= 8.2e-5]
# A scalar weight_g gradient is <dW, v> / ||v||: a dot product with cancellation, so its error is measured against
# the scale of its terms, ||dW||_F (recovered as ||dweight_v|| * ||v|| / |g|), not against its own magnitude.
# bf16: the image encoder stores activations in bf16, and that storage ALONE (oracle with the same rounding points,
# all arithmetic fp32 on the CPU) moves this model's encoder gradients by 0.4-0.6 and its attention gradients by
# 0.07-0.2 relative to the fp32 reference.  So the bf16 statement per tensor is: not further from the fp32 reference
# than 1.3 x what bf16 storage does to the same algorithm + 0.03 (cnn, attention), and fixed bounds where the storage
# noise is small (rest 3e-2 [1.5e-2], logits 1e-2, loss 2e-3).
ATT = ('linear_inside_relu_attention', 'linear_after_relu_attention')
B32_TOL = {
    'fp32': dict(logits=1e-4, loss=3e-4, cnn=1e-2, attention=1e-2, rest=3e-4, weight_g=3e-4, stats=1e-4,
                 head_alone=2e-5),
    'bf16': dict(logits=1e-2, loss=2e-3, cnn=None, attention=None, rest=3e-2, weight_g=3e-2, stats=5e-3,
                 head_alone=None),
}


def _group(k):
    if k.endswith('weight_g'):
        return 'weight_g'
    return 'cnn' if k.startswith('cnn.') else ('attention' if k.startswith(ATT) else 'rest')


def _bf16_storage_errors(c, noise=None):
    """What bf16 activation storage in the image encoder does to every gradient of the SAME algorithm: the oracle with
    the bf16 path's rounding points (oracle._q) and fp32 arithmetic, against the fp32 golden."""
    from oracle import vqa_oracle as O
    g = c['golden']
    p = {k: (v.detach().clone().requires_grad_(True) if O.is_param(k) else v) for k, v in c['sd'].items()}
    v = O.cnn_forward(p, c['images'].float(), c['padding'], c['pooling'], True, None, None, act_dtype=torch.bfloat16)
    lo = O.head_forward(p, v, O.gru_forward(p, c['questions']), None, noise)
    names = [k for k in p if O.is_param(k)]
    grads = torch.autograd.grad(O.bce_sum(lo, c['targets']), [p[k] for k in names])
    return {k: rel_err(gr.reshape(-1).numpy()[sample_idx(gr.numel())], g[f'gsamp/{k}']) for k, gr in zip(names, grads)}


def _check_train_step(m, c, prec, noise=None):
    import vqa_model
    from oracle import vqa_oracle as O
    g = c['golden']
    tol = B32_TOL[prec]
    storage = _bf16_storage_errors(c, noise) if prec == 'bf16' else None
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    logits = m(img, q, dropout_noise=noise)
    loss = vqa_model.SoftScoreBCELoss()(logits, t)
    loss.backward()
    e_logits = rel_err(logits.detach().cpu(), g['logits'])
    e_loss = abs(float(loss.detach()) - float(g['loss'])) / abs(float(g['loss']))
    assert e_logits < tol['logits'] and e_loss < tol['loss'], (e_logits, e_loss)
    for k, b in m.named_buffers():
        if 'running' in k:
            assert rel_err(b.cpu(), g[f'stat/{k}']) < tol['stats'], k
    worst = {}
    for k, p in m.named_parameters():
        flat = p.grad.detach().reshape(-1).cpu()
        if _is_conv_bias(k):
            assert float(flat.abs().max()) == 0.0
            continue
        grp = _group(k)
        if grp == 'weight_g':
            kv = k[:-1] + 'v'
            scale = float(g[f'gnorm/{kv}']) * float(c['sd'][kv].norm()) / abs(float(c['sd'][k]))   # ||dW||_F
            e = abs(float(flat[0]) - float(g[f'gsamp/{k}'][0])) / scale
            bound = tol['weight_g']
        else:
            e = rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}'])
            bound = tol[grp] if tol[grp] is not None else 1.3 * storage[k] + 0.03
        worst[grp] = max(worst.get(grp, 0.0), e if tol.get(grp) is not None else e / bound)
        assert e < bound, (k, e, bound)
    # the head alone on the REFERENCE's encoder outputs (golden cnn_out / gru_out) against the oracle's autograd
    m.zero_grad(set_to_none=True)
    vg = torch.from_numpy(g['cnn_out']).cuda().requires_grad_(True)
    hg = torch.from_numpy(g['gru_out']).cuda().requires_grad_(True)
    vqa_model.SoftScoreBCELoss()(m.head(vg, hg, noise), t).backward()
    ph = {k: (x.clone().requires_grad_(True) if O.is_param(k) else x) for k, x in c['sd'].items()
          if not k.startswith(('cnn.', 'gru.'))}
    vr = torch.from_numpy(g['cnn_out']).requires_grad_(True)
    hr = torch.from_numpy(g['gru_out']).requires_grad_(True)
    O.bce_sum(O.head_forward(ph, vr, hr, None, noise), c['targets']).backward()
    alone = {'dv': rel_err(vg.grad.cpu(), vr.grad), 'dq': rel_err(hg.grad.cpu(), hr.grad)}
    for k, prm in m.named_parameters():
        if not k.startswith(('cnn.', 'gru.')):
            if k.endswith('weight_g'):
                kv = k[:-1] + 'v'
                scale = float(ph[kv].grad.norm()) * float(c['sd'][kv].norm()) / abs(float(c['sd'][k]))
                alone[k] = abs(float(prm.grad) - float(ph[k].grad)) / scale
            else:
                alone[k] = rel_err(prm.grad.cpu(), ph[k].grad)
    worst['head_alone'] = max(alone.values())
    if tol['head_alone'] is not None:
        assert worst['head_alone'] < tol['head_alone'], alone
    else:                                                    # bf16: the bounds of test_head_fwd_bwd_vs_oracle
        for k, e in alone.items():
            assert e < (HEAD_BF16['weight_g'] if k.endswith('weight_g') else
                        HEAD_BF16['att'] if k.startswith(ATT) or k in ('dv', 'dq') else HEAD_BF16['grad']), (k, e)
    return e_logits, e_loss, worst


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_b32_train_step_vs_golden(prec, tmp_path):
    """Full training forward + backward at H=1024, padding 2, 224x224, B=32 against the imported reference."""
    c = load_case('h1024_p2_max_b32')
    m = build_model(c, tmp_path, prec).cuda().train()
    e_logits, e_loss, worst = _check_train_step(m, c, prec)
    print(f'\n[b32 {prec}] logits {e_logits:.2e} loss {e_loss:.2e} ' + ' '.join(f'{k} {e:.2e}' for k, e in worst.items()))


# What ONE flipped ReLU decision may add on top of "1.5 x the reference's own distance from fp64" (below).  Of the
# attention layer's B * k * H = 819,200 units one has an fp64 pre-activation of 4.7e-9 (typical: 0.27).  Every fp32
# perturbation decides it anew - scripts/fp64_anchor.py: the fp64 head fed with this path's encoder outputs, with the
# imported reference's, with only v or only q swapped, flips it each time - and that one unit is worth 2.0e-3 on the
# layer's bias gradient and 9.8e-4 on its weight_v gradient.  Torch's fp32 run happens to leave it on fp64's side (3.4e-5
# from fp64), this path's puts it on the other (1.97e-3): a coin, not an accuracy gap - the two fp32 encoders sit 3.2e-6
# and 4.0e-6 from the fp64 encoder output, and on IDENTICAL encoder outputs the head agrees with the oracle to 3e-6
# (_check_train_step).  What that unit sends back through dv reaches the last conv's weight gradient (1.1e-4 against
# 3.5e-5); below it both encoders' own decisions dominate and the factor alone holds.
FP64_SLACK = dict(cnn=2e-4, attention=4e-3, rest=2e-5, weight_g=2e-5)


def test_fp32_gradients_are_as_close_to_an_fp64_run_as_the_reference_is(tmp_path):
    """The fp32 statement without a hand-picked tolerance ("what's weak" 2).  The B=32 bounds above are 1e-2
    for the image encoder's gradients where the north_star says 1e-4, because the reference's OWN fp32 gradients are only
    that reproducible: ReLU / arg-max decisions flip on 1e-7 forward differences.  Here the same algorithm is run once
    more in fp64 (the oracle on the CPU, double precision end to end: the exact answer to fp32 accuracy) and every
    gradient of the HIP fp32 path must be no further from it than 1.5 x the imported reference's fp32 gradient is, plus
    a stated slack per group (FP64_SLACK: rounding level for the question encoder, the projections and the classifier;
    one flipped unit for the attention layer) - per parameter tensor; logits and loss likewise (there this path is
    closer to fp64 than the reference: its weight norm sums pairwise, torch's naively - DESIGN.md section 2)."""
    import vqa_model
    from oracle import vqa_oracle as O
    c = load_case('h1024_p2_max_b32')
    g = c['golden']
    dt = torch.float64
    p = {k: (v.to(dt).detach().clone().requires_grad_(O.is_param(k)) if v.is_floating_point() else v)
         for k, v in c['sd'].items()}
    lo64 = O.head_forward(p, O.cnn_forward(p, c['images'].to(dt), c['padding'], c['pooling'], True, None, None),
                          O.gru_forward(p, c['questions']))
    loss64 = O.bce_sum(lo64, c['targets'].to(dt))
    names = [k for k in p if O.is_param(k)]
    g64 = dict(zip(names, torch.autograd.grad(loss64, [p[k] for k in names])))
    m = build_model(c, tmp_path, 'fp32').cuda().train()
    logits = m(c['images'].cuda(), c['questions'].cuda())
    loss = vqa_model.SoftScoreBCELoss()(logits, c['targets'].cuda())
    loss.backward()
    e_hip, e_ref = rel_err(logits.detach().cpu(), lo64.detach()), rel_err(g['logits'], lo64.detach())
    assert e_hip <= 1.5 * e_ref + 2e-6, (e_hip, e_ref)
    l64 = float(loss64.detach())
    assert abs(float(loss.detach()) - l64) <= 1.5 * abs(float(g['loss']) - l64) + 2e-6 * abs(l64)
    worst, bad = {}, []
    for k, prm in m.named_parameters():
        if _is_conv_bias(k):
            continue                                # (identically zero behind a training-mode BatchNorm)
        exact = g64[k].reshape(-1).numpy()
        idx = sample_idx(exact.size)
        hip = prm.grad.detach().reshape(-1).cpu().numpy()[idx]
        if k.endswith('weight_g'):                  # a dot product with cancellation: against the scale of its terms
            kv = k[:-1] + 'v'
            scale = float(g64[kv].norm()) * float(c['sd'][kv].norm()) / abs(float(c['sd'][k]))
            eh, er = abs(float(hip[0]) - float(exact[0])) / scale, abs(float(g[f'gsamp/{k}'][0]) - float(exact[0])) / scale
        else:
            eh, er = rel_err(hip, exact[idx]), rel_err(g[f'gsamp/{k}'], exact[idx])
        grp = _group(k)
        if eh / (er + 2e-5) > worst.get(grp, (0.0,))[0]:
            worst[grp] = (eh / (er + 2e-5), eh, er, k)
        if not eh <= 1.5 * er + FP64_SLACK[grp]:
            bad.append((k, eh, er))
    print(f'\n[fp64 anchor] logits hip {e_hip:.2e} ref {e_ref:.2e}; worst hip / ref per group: ' +
          '; '.join(f'{grp} {w[1]:.2e} / {w[2]:.2e} ({w[3]})' for grp, w in worst.items()))
    assert not bad, bad


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_dropout_replayed_vs_golden(prec, tmp_path):
    """Dropout p = 0.0381 (vqa_model.py:55, :101, :103, :407-416): with the reference's two recorded keep-masks the HIP
    path reproduces its logits, loss and gradients; without them (its own draws) it does not - the masks are live."""
    c = load_case('h1024_p2_max_b32_drop')
    m = build_model(c, tmp_path, prec).cuda().train()
    assert m.dropout_p == pytest.approx(0.0381)
    e_logits, e_loss, worst = _check_train_step(m, c, prec, noise=c['noise'])
    print(f'\n[drop {prec}] logits {e_logits:.2e} loss {e_loss:.2e} ' + ' '.join(f'{k} {e:.2e}' for k, e in worst.items()))
    m.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    own = m(c['images'].cuda(), c['questions'].cuda())                        # its own Bernoulli draws
    assert rel_err(own.detach().cpu(), c['golden']['logits']) > 10 * B32_TOL['fp32']['logits']
    # the module's own draws are inverted dropout: kept fraction ~ 1 - p, kept values scaled by 1 / (1 - p)
    mask, add = m._dropout_mask(4096, torch.device('cuda'))
    assert add is None                                                        # nn.Dropout: multiplicative only
    assert abs(float((mask > 0).float().mean()) - (1 - 0.0381)) < 5e-3
    assert float(mask.max()) == pytest.approx(1 / (1 - 0.0381), rel=1e-6)
    m.eval()
    assert m._dropout_mask(4, torch.device('cuda')) == (None, None)


# bf16 row: what three steps of the bf16 path keep of the reference's fp32 trajectory on this synthetic-weight model (the
# loss goes 12104 -> 852 -> 2641).  Scalars and the smooth parameters' updates hold to a few percent; the running
# MEANS of the deep layers do not (measured 0.11 on cnn.convolutions.22): after step 1 every weight has moved by lr *
# sign(g), the image encoder's step-2 / step-3 activations are those of a different network wherever a near-zero
# step-1 gradient took the other sign, and a batch mean is a small difference of large numbers.  The bound on them is
# therefore loose and says so; how the bf16 path TRAINS is pinned by test_bf16_trajectory_tracks_fp32 below.
@pytest.mark.parametrize('prec,tol', [('fp32', dict(tol_scalar=5e-3, tol_update=1e-2)),
                                      ('bf16', dict(tol_scalar=5e-2, tol_update=8e-2, tol_var=5e-2, tol_mean=3e-1))])
def test_three_steps_vs_golden(prec, tol, tmp_path):
    """Three whole training steps (Trainer: zero_grad -> fwd -> loss -> bwd -> clip 0.25 -> Adamax) on the B=32 case
    against the reference's state after its third step: losses and clip norms of steps 2 and 3, running statistics,
    and the accumulated updates of the smoothly-behaving parameters (tests/test_oracle_golden.py::SMOOTH)."""
    from tests.test_oracle_golden import check_three_steps
    from vqa_amd.optim import Trainer
    c = load_case('h1024_p2_max_b32')
    m = build_model(c, tmp_path, prec).cuda().train()
    tr = Trainer(m, lr=2e-3, max_norm=0.25)
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    losses, norms = [], []
    for _ in range(3):
        losses.append(float(tr.step(img, q, t, flip=False)))
        norms.append(float(tr.opt.total_norm))
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    worst = check_three_steps(c, sd, losses, norms, **tol)
    for k in range(10):
        assert int(m.cnn.convolutions[[1, 4, 8, 11, 15, 18, 22, 25, 29, 32][k]].num_batches_tracked) == 3
    g = c['golden']
    stat = {kind: max(rel_err(sd[k], g[f'stat3/{k}']) for k in c['sd'] if k.endswith(kind))
            for kind in ('running_mean', 'running_var')}
    print(f'\n[3 steps {prec}] losses {losses} norms {norms} worst mean |update diff| / lr {worst:.2e} '
          f'running stats {stat}')


def test_reference_loop_as_written(tmp_path):
    """The reference's training loop, line for line as vqa_model.py:246-266 writes it, on the HIP module - no Trainer, no
    SoftScoreBCELoss: `model.flip(images)`, `nn.BCEWithLogitsLoss(reduction='sum')`, `loss.backward()`,
    `nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.25, norm_type=2)`, `float(loss)`, `optim.Adamax(...).step()`
    - for three steps on the B=32 case (fp32), against the imported reference's state after ITS third step (losses /
    clip norms of steps 2 and 3, running statistics, p3samp/*).  The flip's coin is pinned to "no flip" (p = 0), as the
    golden was recorded; that model.flip flips when the coin says so is checked at the end."""
    import torch.nn as nn
    import torch.optim as optim
    from tests.test_oracle_golden import check_three_steps
    c = load_case('h1024_p2_max_b32')
    model = build_model(c, tmp_path, 'fp32')
    model = model.to(model.device)
    batch = {'image': c['images'], 'question': c['questions'], 'answer': c['targets']}
    criterion = nn.BCEWithLogitsLoss(reduction='sum')
    optimizer = optim.Adamax(model.parameters(), lr=2e-3)
    model.flip.p = 0.0
    train_epoch_losses, norms = list(), list()
    model.train()
    for i_batch in range(3):
        optimizer.zero_grad()

        images_batch_ = model.flip(batch['image'].cuda())
        questions_batch_ = batch['question'].cuda()
        target = batch['answer'].cuda()

        output = model(images_batch_, questions_batch_)
        loss = criterion(output, target)
        loss.backward()

        norms.append(float(nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.25, norm_type=2)))

        train_epoch_losses.append(float(loss))
        optimizer.step()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    worst = check_three_steps(c, sd, train_epoch_losses, norms)
    assert train_epoch_losses[0] == pytest.approx(float(c['golden']['loss']), rel=3e-4)
    print(f'\n[reference loop] losses {train_epoch_losses} norms {norms} worst mean |update diff| / lr {worst:.2e}')
    model.flip.p = 1.0
    x = batch['image'][:2].cuda()
    assert torch.equal(model.flip(x), x.flip(-1))


def test_bf16_trajectory_tracks_fp32(tmp_path):
    """How the bf16 path TRAINS, against the fp32 path of the same library (which the tests above pin to the reference):
    same module-default initialisation (seed 42, as bench.py and vqa_model.py:334), H = 1024, 224x224, four fixed
    batches of 128 samples visited in turn for 40 Trainer steps (clip 0.25 + Adamax lr 2e-3, flips on with the same
    coin sequence).  Per-step gradients of this model are chaotic in bf16 (test_b32_train_step_vs_golden), so the
    statement is about the trajectory: the bf16 loss stays within a band of the fp32 loss at EVERY step, the two end at
    the same mean loss over the last visit of the four batches, and the eval-mode soft accuracy on a fixed 256-sample
    set (BatchNorm running statistics and all) differs by at most a stated delta.  Measured values: DESIGN.md section 2."""
    import copy
    import bench
    import vqa_model
    from vqa_amd import ops, synth
    from vqa_amd.optim import Trainer
    H, C, V, B, NB, STEPS = 1024, 3129, 10000, 128, 4, 40
    qpath, lpath = bench.make_fixture(str(tmp_path), V, C)
    torch.manual_seed(42)
    base = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': H, 'n_layers': 1, 'train_question_path': qpath},
                         lpath, 256, 2, 0.0, 'max', 'relu', precision='fp32')
    batches = []
    for i in range(NB):
        q, t = synth.questions(100 + i, B, 14, V), synth.targets(100 + i, B, C)
        # something to learn within 40 steps, so that the soft accuracy says more than chance: one of seven answers,
        # chosen by the question's first word, carries the full score (VQA-v2's "yes" / "no" / "2" play this role)
        t[torch.arange(B), q[:, 0] % 7] = 1.0
        batches.append((synth.images(100 + i, B, 224).cuda(), q.cuda(), t.cuda()))
    ev = (torch.cat([b[0] for b in batches[:2]]), torch.cat([b[1] for b in batches[:2]]),
          torch.cat([b[2] for b in batches[:2]]))
    curves, accs, eval_loss = {}, {}, {}
    for prec in ('fp32', 'bf16'):
        m = copy.deepcopy(base).set_precision(prec).cuda().train()
        tr = Trainer(m, lr=2e-3, max_norm=0.25)
        torch.manual_seed(7)                                   # the flip's coin sequence
        curves[prec] = [float(tr.step(*batches[s % NB])) for s in range(STEPS)]
        m.eval()
        with torch.no_grad():
            lo = m(ev[0], ev[1])
            accs[prec] = float(ops.soft_accuracy_sum(lo, ev[2])) / ev[0].shape[0]
            eval_loss[prec] = float(vqa_model.SoftScoreBCELoss()(lo, ev[2])) / ev[0].shape[0]
    f, b = curves['fp32'], curves['bf16']
    rel = [abs(x - y) / x for x, y in zip(f, b)]
    print(f'\n[trajectory] fp32 loss {[round(x, 1) for x in f]}\n[trajectory] bf16 loss {[round(x, 1) for x in b]}\n'
          f'[trajectory] max rel diff {max(rel):.3e} (first 10 steps {max(rel[:10]):.3e}), soft accuracy fp32 '
          f'{accs["fp32"]:.4f} bf16 {accs["bf16"]:.4f}, eval loss / sample fp32 {eval_loss["fp32"]:.3f} bf16 '
          f'{eval_loss["bf16"]:.3f}')
    assert all(x == x for x in f + b)
    assert f[-1] < 0.25 * f[0] and b[-1] < 0.25 * b[0]                     # both train
    assert max(rel) < TRAJ_BAND, rel
    last_f, last_b = sum(f[-NB:]) / NB, sum(b[-NB:]) / NB
    assert abs(last_f - last_b) < TRAJ_FINAL * last_f
    assert abs(accs['fp32'] - accs['bf16']) <= TRAJ_ACC_DELTA
    assert abs(eval_loss['fp32'] - eval_loss['bf16']) < TRAJ_EVAL_LOSS * eval_loss['fp32']


# bands of test_bf16_trajectory_tracks_fp32 (measured values in DESIGN.md section 2).  The pinned statement is the
# TRAINING-loss band (stable across every variant of the library measured: max 0.26-0.28 at step 17, 0.055-0.067 over
# the first ten steps) and the final mean loss.  The eval-mode figures after 40 steps are weak evidence and bounded as
# such: they are only as reproducible as the fp32 path is against ITSELF - the same fp32 run with another fp32
# summation order of one layer's sums (two images per conv tile on / off in the forward or the weight gradient, batch
# slices, fused / separate reduce passes: gradients equal to 1e-7 in step 1, 1e-3 in step 2, 4e-2 in step 3 - Adamax's
# first update is lr * sign(g)) ends at an eval loss per sample between 16.1 and 18.7 (bf16: 15.8-16.5), and a soft
# accuracy of 0.14-0.19 on 256 samples of a one-in-seven task is within two standard errors (0.023 each) of chance:
# the accuracy bound is three standard errors of a difference, 3 * sqrt(2 * 0.16 * 0.84 / 256) = 0.097.
TRAJ_BAND, TRAJ_FINAL, TRAJ_ACC_DELTA, TRAJ_EVAL_LOSS = 0.35, 0.05, 0.097, 0.25


def test_clip_adamax_vs_torch_optim():
    """vqa_clip_adamax element-wise against nn.utils.clip_grad_norm_(0.25) + torch.optim.Adamax(lr=2e-3) (the calls
    at vqa_model.py:259, :218, :266) over six steps of synthetic gradients whose scale shrinks and grows - so the
    running max u alternates between its decayed value and |g| + eps, the bias correction runs past step 1, and the
    norm falls on both sides of the clip threshold."""
    from vqa_amd import synth
    from vqa_amd.optim import FlatAdamax, FlatParams
    mod = torch.nn.ParameterDict({'a': torch.nn.Parameter(synth.uniform(1, 'p.a', (257, 33), -1, 1)),
                                  'b': torch.nn.Parameter(synth.uniform(1, 'p.b', (1000,), -1, 1))}).cuda()
    ref = {k: torch.nn.Parameter(v.detach().cpu().clone()) for k, v in mod.items()}
    opt_ref = torch.optim.Adamax(ref.values(), lr=2e-3)
    flat = FlatParams(mod)
    opt = FlatAdamax(flat, lr=2e-3, max_norm=0.25)
    for step, scale in enumerate([1.0, 1e-3, 5e-6, 3.0, 1e-4, 0.5]):
        for k in ref:
            g = synth.uniform(2 + step, 'g.' + k, tuple(ref[k].shape), -1, 1) * scale
            ref[k].grad = g.clone()
            mod[k].grad = None
            mod[k]._vqa_grad_dst.copy_(g.reshape(-1))
            mod[k].grad = mod[k]._vqa_grad_dst.view(mod[k].shape)
        total_ref = torch.nn.utils.clip_grad_norm_(ref.values(), max_norm=0.25, norm_type=2)
        opt_ref.step()
        opt.step()
        assert float(opt.total_norm) == pytest.approx(float(total_ref), rel=1e-6)
        for k in ref:
            assert torch.allclose(mod[k].detach().cpu(), ref[k].detach(), rtol=0, atol=3e-7), (k, step)


@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
def test_gradient_accumulation_with_side_streams(prec, tmp_path, monkeypatch):
    """with an existing p.grad (two backward calls without zero_grad, or zero_grad(set_to_none=False))
    autograd runs `p.grad += g` on the calling stream right after the head's backward returns - the head's parameter
    gradients, computed on the weight-gradient stream, must be finished by then.  Streams on vs off, bit for bit."""
    import vqa_model
    from vqa_amd import cnn as CN, vqa_model as VM
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    crit = vqa_model.SoftScoreBCELoss()
    out = []
    for streams in (True, False):
        monkeypatch.setattr(VM, 'SIDE_STREAM', streams)
        monkeypatch.setattr(VM, 'HEAD_WGRAD_STREAM', streams)
        monkeypatch.setattr(CN, 'WGRAD_STREAM', streams)
        m = build_model(c, tmp_path, prec).cuda().train()
        m.zero_grad(set_to_none=True)
        crit(m(img, q), t).backward()
        crit(m(img.flip(0), q.flip(0)), t.flip(0)).backward()              # accumulates into the existing grads
        twice = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        m.zero_grad(set_to_none=False)                                     # grads stay allocated, zeroed
        crit(m(img, q), t).backward()
        seen = []
        h = m.fc.weight_v.register_hook(lambda g: seen.append(g.clone()))  # a tensor hook reads the gradient at once
        m.zero_grad(set_to_none=True)
        crit(m(img, q), t).backward()
        h.remove()
        # a post-accumulate-grad hook that is not GradBuckets' own (a user's, DDP's, a logger's) reads p.grad on
        # the calling stream right after the adoption: the head must have waited for its side-stream gradients
        post = []
        h2 = m.linear_inside_relu_last.weight_v.register_post_accumulate_grad_hook(lambda p: post.append(p.grad.clone()))
        m.zero_grad(set_to_none=True)
        crit(m(img, q), t).backward()
        h2.remove()
        out.append((twice, {k: p.grad.detach().clone() for k, p in m.named_parameters()}, seen[0], post[0]))
    for a, b in zip(out[0][:2], out[1][:2]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][2], out[0][1]['fc.weight_v'])
    assert torch.equal(out[0][3], out[1][3]) and torch.equal(out[0][3], out[0][1]['linear_inside_relu_last.weight_v'])


@pytest.mark.parametrize('knob', ['HEAD_DQ_SIDE', 'HEAD_WN_EARLY', 'HEAD_FUSE_ELEMENTWISE'])
def test_head_stream_knobs_are_bit_identical(knob, tmp_path, monkeypatch):
    """The two measured-and-not-kept placements of head work (vqa_model.HEAD_DQ_SIDE: the question gradient leads the
    parameter-gradient work on the weight-gradient stream; HEAD_WN_EARLY: the weight norms issued at the start of the
    forward on that stream) change streams, not launches, and HEAD_FUSE_ELEMENTWISE (the default: vqa_hadamard_fwd and
    vqa_relu_bwd as epilogues of the GEMMs beside them) changes launches, not arithmetic: logits and every gradient bit
    for bit, two steps in a row (the second starts while the first one's side-stream work may still be running)."""
    import vqa_model
    from vqa_amd import vqa_model as VM
    c = load_case('h64_p5_max')
    img, q, t = c['images'].cuda(), c['questions'].cuda(), c['targets'].cuda()
    crit = vqa_model.SoftScoreBCELoss()
    out = []
    for on in (True, False):
        monkeypatch.setattr(VM, knob, on)
        m = build_model(c, tmp_path, 'bf16').cuda().train()
        res = []
        for rep in range(2):
            m.zero_grad(set_to_none=True)
            logits = m(img, q)
            crit(logits, t).backward()
            torch.cuda.synchronize()
            res.append((logits.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
        out.append(res)
    for (la, ga), (lb, gb) in zip(*out):
        assert torch.equal(la, lb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


def test_soft_accuracy_on_nan_and_inf_rows():
    """A diverged model reaching evaluate(): rows of NaN / -inf logits.  torch.argmax returns a valid index (the first
    NaN, else the first maximum) and the reference's loop carries on (vqa_model.py:127-130); so must the kernels -
    and never gather out of range."""
    from vqa_amd import ops, synth
    B, C, L = 9, 3129, 10
    torch.manual_seed(0)
    x = torch.randn(B, C)
    x[1] = float('nan')
    x[2] = float('-inf')
    x[3, 100:] = float('nan')                       # first NaN at 100
    x[4, 7] = float('inf')
    x[5] = 0.25                                     # all equal: first index
    x[6, 2000] = float('nan'); x[6, 5] = float('inf')
    t = synth.targets(3, B, C)
    t[1, 0] = 0.9; t[2, 0] = 0.6; t[3, 100] = 1.0; t[4, 7] = 0.3; t[5, 0] = 0.3; t[6, 2000] = 0.6
    want = float(t[torch.arange(B), torch.argmax(x, dim=1)].sum())
    assert float(ops.soft_accuracy_sum(x.cuda(), t.cuda())) == pytest.approx(want, abs=1e-6)
    labels = torch.full((B, L), -1, dtype=torch.int64)
    scores = torch.zeros(B, L)
    for b in range(B):
        nz = t[b].nonzero().flatten()
        labels[b, :len(nz)] = nz
        scores[b, :len(nz)] = t[b, nz]
    assert float(ops.soft_accuracy_sum(x.cuda(), (labels.cuda(), scores.cuda()))) == pytest.approx(want, abs=1e-6)


def test_out_of_vocabulary_token_poisons_instead_of_faulting(tmp_path):
    """A token id outside the embedding table (nn.Embedding raises IndexError): the gather reads nothing, the row
    becomes NaN and reaches the output; the backward leaves the table's gradient untouched by it."""
    from tests.test_gpu_gru import _mk
    g, _ = _mk(64, 82, tmp_path, 'fp32')
    q = torch.randint(2, 82, (4, 6))
    q[2, 3] = 82
    q[3, 0] = -5
    h = g(q.cuda())
    assert torch.isfinite(h[:2]).all() and torch.isnan(h[2]).all() and torch.isnan(h[3]).all()
