"""CPU: the oracle (oracle/vqa_oracle.py) against golden vectors captured from the imported reference.

Pins SURVEY.md §8(c): the restatement equals the reference's torch modules on identical synthetic inputs
and weights (fp32, <=1e-5 relative; most tensors agree to ~1e-7)."""
import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from tests.helpers import CONFIGS, VARIANT_CONFIGS, load_case, rel_err, sample_idx

TOL = 1e-5


@pytest.mark.parametrize('name', list(CONFIGS))
def test_oracle_matches_reference(name, monkeypatch):
    c = load_case(name)
    g = c['golden']
    monkeypatch.setattr(O, 'REFERENCE_NORM', True)   # tight pin: same Frobenius-norm call as the reference
    # eval forward + soft accuracy
    lo = O.forward(c['sd'], c['images'], c['questions'], c['padding'], c['pooling'], training=False)
    assert rel_err(lo, g['eval_logits']) < TOL
    assert abs(float(O.bce_sum(lo, c['targets'])) - float(g['eval_loss'])) < TOL * abs(float(g['eval_loss']))
    assert O.soft_accuracy_sum(lo, c['targets']) == pytest.approx(float(g['eval_acc_sum']), abs=1e-6)
    # train step
    inter = {}
    O.forward(c['sd'], c['images'], c['questions'], c['padding'], c['pooling'], True, None, inter)
    assert rel_err(inter['cnn_out'], g['cnn_out']) < TOL
    assert rel_err(inter['gru_out'], g['gru_out']) < TOL
    loss, logits, grads, stats = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'],
                                                    c['padding'], c['pooling'])
    assert rel_err(logits, g['logits']) < TOL
    assert abs(float(loss) - float(g['loss'])) < TOL * abs(float(g['loss']))
    for k, gr in grads.items():
        flat = gr.reshape(-1)
        ref_norm = float(g[f'gnorm/{k}'])
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            # conv bias grads are mathematically zero under training-mode BN: only rounding noise
            assert float(flat.double().norm()) < 1e-3 * float(g['total_norm'])
            continue
        # CNN gradients are only reproducible to ~1e-3 in fp32: one ReLU/arg-max decision that flips on a
        # 1e-7 forward difference in a small deep layer moves every upstream gradient by ~1/sqrt(N_active)
        # (measured: the reference's own fp32 gradients sit 3.5e-3 from an fp64 run on the 448 case).
        tol = 3e-2 if k.startswith('cnn.') else (1e-3 if k.endswith('weight_g') else 2e-4)
        assert abs(float(flat.double().norm()) - ref_norm) < tol * ref_norm + 1e-7, k
        assert rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}']) < tol, k
    for k, v in stats.items():
        assert rel_err(v, g[f'stat/{k}']) < TOL, k
    # clip + Adamax
    total, clipped = O.clip_grad_norm(grads)
    assert float(total) == pytest.approx(float(g['total_norm']), rel=1e-3)   # dominated by CNN grads, see above
    params = {k: v for k, v in c['sd'].items() if O.is_param(k)}
    new = O.adamax_step(params, clipped, {})
    for k, p in new.items():   # Adamax's first step moves every weight by lr*sign(g): insensitive to the above
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            continue           # ... except conv biases, whose gradient is pure rounding noise (sign is arbitrary)
        d = np.abs(p.reshape(-1).numpy()[sample_idx(p.numel())] - g[f'psamp/{k}'])
        assert (d > 1e-4).mean() <= 0.07, k   # a near-zero gradient whose sign differs moves a weight by 2*lr


def test_accurate_norm_is_closer_to_fp64():
    """The oracle's default (pairwise) weight-norm is the accurate one: fp32 oracle vs an fp64 run of itself
    agrees to 1e-6, while the reference's own fp32 output is ~5e-5 away (torch CPU norm(), see oracle header)."""
    c = load_case('h1024_p2_max')
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in c['sd'].items()}
    lo64 = O.forward(sd64, c['images'], c['questions'], c['padding'], c['pooling'], training=False)
    lo32 = O.forward(c['sd'], c['images'], c['questions'], c['padding'], c['pooling'], training=False)
    assert rel_err(lo32, lo64) < 2e-6
    assert rel_err(c['golden']['eval_logits'], lo64) < 1e-4


def test_eval32_soft_accuracy(monkeypatch):
    monkeypatch.setattr(O, 'REFERENCE_NORM', True)
    c = load_case('h64_p5_max')
    from vqa_amd import synth
    img = synth.images(1235, 32, 224)
    q = synth.questions(1235, 32, c['tokens'], c['vocab'])
    t = synth.targets(1235, 32, c['classes'])
    lo = O.forward(c['sd'], img, q, c['padding'], c['pooling'], training=False)
    assert rel_err(lo, c['golden']['eval32_logits']) < TOL
    assert O.soft_accuracy_sum(lo, t) == pytest.approx(float(c['golden']['eval32_acc_sum']), abs=1e-6)


def oracle_steps(c, n_steps):
    """n_steps of fwd + bwd + clip + Adamax on the case's batch (reference vqa_model.py:247-266 per iteration, no flip)
    -> (state dict after the last step, [loss], [total_norm])."""
    sd = dict(c['sd'])
    state, losses, norms = {}, [], []
    for _ in range(n_steps):
        loss, _, grads, stats = O.train_step_grads(sd, c['images'], c['questions'], c['targets'], c['padding'],
                                                   c['pooling'])
        total, clipped = O.clip_grad_norm(grads)
        new = O.adamax_step({k: v for k, v in sd.items() if O.is_param(k)}, clipped, state)
        sd.update(new)
        sd.update(stats)
        losses.append(float(loss))
        norms.append(float(total))
    return sd, losses, norms


# Parameters whose three-step trajectory is reproducible: the step moves EVERY weight by ~lr (Adamax), the loss goes
# 12104 -> 852 -> 2641, and the image encoder's gradients at steps 2 and 3 are chaotic in the ~0.07 % of weights whose
# near-zero step-1 gradient has the other sign (ReLU / arg-max flips downstream): even this oracle, which runs the same
# ATen CPU kernels as the reference, ends 0.2-0.3 lr away from it on the conv weights.  The question encoder and the
# layers after the attention are smooth (mean difference < 0.003 lr) and carry the pin; Adamax itself is pinned
# element-wise against torch.optim.Adamax in test_adamax_matches_torch_optim.
SMOOTH = ('gru.', 'linear_inside_relu_image', 'linear_inside_relu_question', 'linear_inside_relu_last', 'fc.')


def check_three_steps(c, sd, losses, norms, lr=2e-3, tol_scalar=5e-3, tol_update=1e-2, tol_var=2e-3, tol_mean=5e-2):
    """Shared with the GPU test: state after three steps against the reference's."""
    g = c['golden']
    for s in (2, 3):
        assert losses[s - 1] == pytest.approx(float(g[f'loss{s}']), rel=tol_scalar)
        assert norms[s - 1] == pytest.approx(float(g[f'total_norm{s}']), rel=tol_scalar)
    worst = 0.0
    for k in c['sd']:
        if k.endswith('running_var'):
            assert rel_err(sd[k], g[f'stat3/{k}']) < tol_var, k
        elif k.endswith('running_mean'):
            assert rel_err(sd[k], g[f'stat3/{k}']) < tol_mean, k
        elif O.is_param(k) and k.startswith(SMOOTH):
            # compare the accumulated UPDATE (three steps of at most lr each), not the parameter it is added to
            idx = sample_idx(c['sd'][k].numel())
            p0 = c['sd'][k].reshape(-1).numpy()[idx]
            d = np.abs((np.asarray(sd[k]).reshape(-1)[idx] - p0) - (g[f'p3samp/{k}'] - p0)) / lr
            worst = max(worst, float(d.mean()))
            assert d.mean() < tol_update, (k, float(d.mean()), float(d.max()))
    return worst


def test_three_adamax_steps(monkeypatch):
    """Bias correction at step >= 2 and the running max u of Adamax (vqa_model.py:218, :266): three whole steps on the
    B=32 case against the reference's parameters / losses / clip norms / running statistics after step 3."""
    monkeypatch.setattr(O, 'REFERENCE_NORM', True)
    c = load_case('h1024_p2_max_b32')
    sd, losses, norms = oracle_steps(c, 3)
    check_three_steps(c, {k: v.numpy() if torch.is_tensor(v) else v for k, v in sd.items()}, losses, norms)


def test_adamax_matches_torch_optim():
    """oracle.clip_grad_norm + adamax_step against the calls the reference makes (vqa_model.py:259
    nn.utils.clip_grad_norm_(0.25), :218 optim.Adamax(lr), :266 step) on a synthetic gradient sequence: six steps,
    gradients that shrink and grow (so u alternates between its decayed value and |g| + eps) and norms on both sides
    of the clip threshold."""
    from vqa_amd import synth
    p0 = {'a': synth.uniform(1, 'p.a', (257, 33), -1, 1), 'b': synth.uniform(1, 'p.b', (1000,), -1, 1)}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in p0.items()}
    opt = torch.optim.Adamax(ref.values(), lr=2e-3)
    mine, state = {k: v.clone() for k, v in p0.items()}, {}
    for step, scale in enumerate([1.0, 1e-3, 5e-6, 3.0, 1e-4, 0.5]):
        grads = {k: synth.uniform(2 + step, 'g.' + k, tuple(v.shape), -1, 1) * scale for k, v in p0.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        total_ref = torch.nn.utils.clip_grad_norm_(ref.values(), max_norm=0.25, norm_type=2)
        opt.step()
        total, clipped = O.clip_grad_norm(grads)
        mine = O.adamax_step(mine, clipped, state)
        assert float(total) == pytest.approx(float(total_ref), rel=1e-6)
        for k in ref:
            assert torch.allclose(mine[k], ref[k].detach(), rtol=0, atol=2e-7), (k, step)


def test_dropout_masks_replayed(monkeypatch):
    """Dropout p = 0.0381 (the authors' final config, vqa_model.py:407-416): the oracle with the reference's two
    recorded keep-masks multiplied in at vqa_model.py:101 / :103 reproduces its logits, loss and gradients."""
    monkeypatch.setattr(O, 'REFERENCE_NORM', True)
    c = load_case('h1024_p2_max_b32_drop')
    g = c['golden']
    assert 0 < int((~c['keep'][0]).sum()) < c['keep'][0].numel() // 10
    loss, logits, grads, _ = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'], c['padding'],
                                                c['pooling'], masks=c['noise'])
    assert rel_err(logits, g['logits']) < TOL
    assert abs(float(loss) - float(g['loss'])) < TOL * abs(float(g['loss']))
    plain = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'], c['padding'], c['pooling'])[1]
    assert rel_err(plain, g['logits']) > 1e-3                       # the masks matter
    for k, gr in grads.items():
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            continue
        tol = 1e-2 if k.startswith('cnn.') else (1e-3 if k.endswith('weight_g') else 2e-4)
        flat = gr.reshape(-1)
        assert rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}']) < tol, k


@pytest.mark.parametrize('name', list(VARIANT_CONFIGS))
def test_oracle_variants_match_reference(name, monkeypatch):
    """The reference's other branches, imported and run like the main cases (tests/golden/make_golden.py):
    activation='selu' (vqa_model.py:54-55: nn.SELU at the four activation sites + nn.AlphaDropout(p=0.5) at the two
    dropout sites, the recorded keep-masks replayed) and n_layers=2 (gru.py:51: stacked nn.GRU)."""
    monkeypatch.setattr(O, 'REFERENCE_NORM', True)
    c = load_case(name)
    g = c['golden']
    act = c['activation']
    lo = O.forward(c['sd'], c['images'], c['questions'], c['padding'], c['pooling'], training=False, activation=act)
    assert rel_err(lo, g['eval_logits']) < TOL
    inter = {}
    O.forward(c['sd'], c['images'], c['questions'], c['padding'], c['pooling'], True, None, inter, activation=act)
    assert rel_err(inter['gru_out'], g['gru_out']) < TOL
    loss, logits, grads, _ = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'], c['padding'],
                                                c['pooling'], masks=c.get('noise'), activation=act)
    assert rel_err(logits, g['logits']) < TOL
    assert abs(float(loss) - float(g['loss'])) < TOL * abs(float(g['loss']))
    if act != 'relu':
        for k in c['keep']:
            assert 0.3 < float(k.float().mean()) < 0.7                  # p = 0.5 whatever `dropout` says
        plain = O.train_step_grads(c['sd'], c['images'], c['questions'], c['targets'], c['padding'], c['pooling'],
                                   activation=act)[1]
        assert rel_err(plain, g['logits']) > 1e-2                       # the AlphaDropout noise matters
    assert set(grads) == {k[len('gnorm/'):] for k in g.files if k.startswith('gnorm/')}
    for k, gr in grads.items():
        if k.startswith('cnn.convolutions') and k.endswith('.bias') and int(k.split('.')[2]) in O.CONV_IDX:
            continue
        tol = 3e-2 if k.startswith('cnn.') else (1e-3 if k.endswith('weight_g') else 2e-4)
        flat = gr.reshape(-1)
        assert rel_err(flat.numpy()[sample_idx(flat.numel())], g[f'gsamp/{k}']) < tol, k
