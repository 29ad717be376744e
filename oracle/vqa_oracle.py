"""CPU oracle for the VQA hot path — TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it.  The product path (vqa_amd/) never routes through it and fails loudly
when the HIP library is missing.

It restates, in plain fp32 PyTorch-CPU arithmetic with explicit formulas, the algorithm of the reference's
training/eval step:

  * CNN            reference cnn.py:22-42 (conv3x3+bias, BatchNorm2d, ReLU, 2x2 pool) and cnn.py:52-53
  * GRU            reference gru.py:48-52, :101-106 (embedding gather + nn.GRU equations, h0=0, h_T)
  * fusion head    reference vqa_model.py:83-103 (concat attention, softmax over regions, projections)
  * weight_norm    reference vqa_model.py:59-75 (dim=None: W = g * v / ||v||_F, scalar g)
  * loss           reference vqa_model.py:215 (BCEWithLogitsLoss(reduction='sum'))
  * soft accuracy  reference vqa_model.py:127-132
  * clip + Adamax  reference vqa_model.py:259, :218, :266

Pinning: the reference has no tests or golden vectors of its own.  This restatement is pinned
by golden outputs captured from the reference modules imported in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz); tests/test_oracle_golden.py re-checks it against them.

State is a plain dict with the reference's state_dict names, so the same dict loads into the reference,
into this oracle and into the HIP-backed module.
"""
import math

import torch
import torch.nn.functional as F

CONV_IDX = [0, 3, 7, 10, 14, 17, 21, 24, 28, 31]   # positions of the convs in reference cnn.py:22-42
POOL_AFTER = {1, 3, 5, 7, 9}                       # a pool follows every second conv block
PAD_LAYERS = {0, 2, 4, 6}                          # convs that take `padding` (cnn.py:23,27,31,35)
HEAD_LAYERS = ['linear_inside_relu_attention', 'linear_after_relu_attention', 'linear_inside_relu_image',
               'linear_inside_relu_question', 'linear_inside_relu_last', 'fc']
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# torch's CPU fp32 `norm()` (what reference weight_norm(dim=None) calls) accumulates naively: on the 3.2 M-element
# fc.weight_v it is off by 5e-5 relative to fp64, and that error lands 1:1 on the logits.  The explicit
# pairwise sum below is accurate to 1e-7.  REFERENCE_NORM=True swaps in torch's own call so that the
# golden pin (tests/test_oracle_golden.py) can be tight; HIP parity tests use the accurate default.
REFERENCE_NORM = False


def weight_norm_w(g, v):
    """weight_norm(dim=None): W = g * v / ||v||_F   (reference vqa_model.py:59-75)."""
    nrm = torch.norm(v) if REFERENCE_NORM else torch.sqrt(torch.sum(v * v))
    return v * (g / nrm)


class _Round(torch.autograd.Function):
    """Storage-precision emulation for the bf16 path: round the value on the way forward (`fwd`) and / or the
    gradient on the way back (`bwd`) to `dtype`, all arithmetic staying fp32.  With dtype=None it is the identity.
    Used only to compare the bf16 HIP kernels against the same algorithm with the same rounding points."""

    @staticmethod
    def forward(ctx, t, dtype, fwd, bwd):
        ctx.dtype, ctx.bwd = dtype, bwd
        return t.to(dtype).to(t.dtype) if fwd else t.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.bwd else g), None, None, None


def _q(t, dtype, fwd=True, bwd=False):
    return t if dtype is None else _Round.apply(t, dtype, fwd, bwd)


class _BatchNormTrain(torch.autograd.Function):
    """Training-mode BatchNorm2d with the closed-form backward
        dy = (gamma * invstd) * (dz - mean(dz) - xhat * mean(dz * xhat)),  dgamma = sum(dz*xhat),  dbeta = sum(dz)
    (the form ATen's native batch_norm backward uses).  Differentiating the mean/variance expressions term by
    term through autograd instead loses ~1e-3 on early-layer gradients in fp32 (measured against an fp64 run)."""

    @staticmethod
    def forward(ctx, y, gamma, beta):
        mean = y.mean(dim=(0, 2, 3))
        var = ((y - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        invstd = 1.0 / torch.sqrt(var + BN_EPS)
        xh = (y - mean[None, :, None, None]) * invstd[None, :, None, None]
        ctx.save_for_backward(xh, gamma, invstd)
        ctx.mark_non_differentiable(mean, var)
        return xh * gamma[None, :, None, None] + beta[None, :, None, None], mean, var

    @staticmethod
    def backward(ctx, dz, _dm, _dv):
        xh, gamma, invstd = ctx.saved_tensors
        n = dz.numel() // dz.shape[1]
        dbeta = dz.sum(dim=(0, 2, 3))
        dgamma = (dz * xh).sum(dim=(0, 2, 3))
        dy = (gamma * invstd)[None, :, None, None] * (
            dz - (dbeta / n)[None, :, None, None] - xh * (dgamma / n)[None, :, None, None])
        return dy, dgamma, dbeta


def cnn_forward(sd, x, padding=2, pooling='max', training=True, new_stats=None, inter=None, act_dtype=None):
    """reference cnn.py:49-54.  x [B,3,S,S] fp32 -> [B,k,256].

    act_dtype=torch.bfloat16 emulates the bf16 path's storage rounding (image, conv weights, conv output y,
    activation a, and the gradients dy / da) with fp32 arithmetic in between; None is the plain fp32 algorithm.

    Training mode: batch statistics, biased variance for the normalisation, unbiased for the running
    update (written to `new_stats` if given); eval mode: running statistics."""
    x = _q(x, act_dtype)
    for li, ci in enumerate(CONV_IDX):
        w, b = sd[f'cnn.convolutions.{ci}.weight'], sd[f'cnn.convolutions.{ci}.bias']
        gamma, beta = sd[f'cnn.convolutions.{ci + 1}.weight'], sd[f'cnn.convolutions.{ci + 1}.bias']
        y = F.conv2d(x, _q(w, act_dtype), b, padding=padding if li in PAD_LAYERS else 0)
        y = _q(y, act_dtype, True, True)
        if training:
            z, mean, var = _BatchNormTrain.apply(y, gamma, beta)
            if new_stats is not None:
                n = y.numel() // y.shape[1]
                rm, rv = sd[f'cnn.convolutions.{ci + 1}.running_mean'], sd[f'cnn.convolutions.{ci + 1}.running_var']
                new_stats[f'cnn.convolutions.{ci + 1}.running_mean'] = \
                    ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
                new_stats[f'cnn.convolutions.{ci + 1}.running_var'] = \
                    ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / (n - 1)).detach()
        else:
            mean, var = sd[f'cnn.convolutions.{ci + 1}.running_mean'], sd[f'cnn.convolutions.{ci + 1}.running_var']
            xh = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
            z = xh * gamma[None, :, None, None] + beta[None, :, None, None]
        x = torch.relu(z)
        if li in POOL_AFTER:
            x = F.max_pool2d(x, 2, 2) if pooling == 'max' else F.avg_pool2d(x, 2, 2)
        x = _q(x, act_dtype, True, True)
        if inter is not None:
            inter[f'cnn.act{li}'] = x
    x = x.permute(0, 2, 3, 1)
    return x.reshape(x.shape[0], x.shape[1] * x.shape[2], -1)


def gru_forward(sd, q, inter=None):
    """reference gru.py:101-106: embedding gather, nn.GRU (gate rows [r;z;n], h0 = 0, layer k > 0 fed by the output
    sequence of layer k-1, no inter-layer dropout: gru.py:51-52), returns the top layer's h_T."""
    seq = sd['gru.word_embedding.weight'][q]                          # [B,T,E]
    layer = 0
    while f'gru.encoder.weight_ih_l{layer}' in sd:
        w_ih, w_hh = sd[f'gru.encoder.weight_ih_l{layer}'], sd[f'gru.encoder.weight_hh_l{layer}']
        b_ih, b_hh = sd[f'gru.encoder.bias_ih_l{layer}'], sd[f'gru.encoder.bias_hh_l{layer}']
        H = w_hh.shape[1]
        h = torch.zeros(q.shape[0], H, dtype=seq.dtype)
        outs = []
        for t in range(q.shape[1]):
            gi = seq[:, t] @ w_ih.t() + b_ih
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            outs.append(h)
        seq = torch.stack(outs, dim=1)
        layer += 1
    return h


SELU_SCALE, SELU_ALPHA = 1.0507009873554805, 1.6732632423543772


def selu(x):
    """nn.SELU (reference vqa_model.py:54, the non-'relu' branch): scale * (max(0,x) + min(0, alpha * (e^x - 1)))."""
    return SELU_SCALE * (torch.clamp(x, min=0) + torch.clamp(SELU_ALPHA * torch.expm1(x), max=0))


def alpha_dropout_noise(keep, p=0.5):
    """nn.AlphaDropout (reference vqa_model.py:55: nn.AlphaDropout(), p = 0.5) from a 0/1 keep tensor, as ATen forms it:
    alpha' = scale * alpha, a = ((alpha'^2 p + 1)(1 - p))^-1/2, out = x * (a keep) + ((keep - 1) alpha' a + alpha' a p).
    Returns the (mask, add) pair head_forward takes for a dropout site."""
    ap = SELU_SCALE * SELU_ALPHA
    a = 1.0 / ((ap * ap * p + 1.0) * (1.0 - p)) ** 0.5
    keep = keep.to(torch.float32)
    return keep * a, (keep - 1.0) * (ap * a) + ap * a * p


def _drop(x, site):
    if site is None:
        return x
    if isinstance(site, (tuple, list)):
        mask, add = site
        return x * mask + add
    return x * site


def head_forward(sd, v, q, inter=None, masks=None, activation='relu'):
    """reference vqa_model.py:83-103.  v [B,k,D], q [B,H] -> logits [B,C].
    masks = (site1, site2): the two dropout sites of :101 and :103 - an nn.Dropout noise tensor [B,H] (keep / (1-p),
    vqa_model.py:55), an nn.AlphaDropout (mask, add) pair (alpha_dropout_noise), or None = identity (p=0 / eval).
    activation: 'relu' -> nn.ReLU, anything else -> nn.SELU (vqa_model.py:54)."""
    act = torch.relu if activation == 'relu' else selu
    W = {n: weight_norm_w(sd[f'{n}.weight_g'], sd[f'{n}.weight_v']) for n in HEAD_LAYERS}
    B, k, _ = v.shape
    concat = torch.cat((v, q[:, None, :].expand(B, k, q.shape[1])), dim=2)
    a = act(concat @ W['linear_inside_relu_attention'].t() + sd['linear_inside_relu_attention.bias'])
    s = a @ W['linear_after_relu_attention'].t()                      # [B,k,1]
    p = torch.softmax(s, dim=1)
    vhat = torch.sum(p * v, dim=1)
    fv = act(vhat @ W['linear_inside_relu_image'].t() + sd['linear_inside_relu_image.bias'])
    fq = act(q @ W['linear_inside_relu_question'].t() + sd['linear_inside_relu_question.bias'])
    had = _drop(fv * fq, None if masks is None else masks[0])
    m = act(had @ W['linear_inside_relu_last'].t() + sd['linear_inside_relu_last.bias'])
    m = _drop(m, None if masks is None else masks[1])
    if inter is not None:
        inter['att_p'] = p[:, :, 0]
    return m @ W['fc'].t()


def forward(sd, images, questions, padding=2, pooling='max', training=True, new_stats=None, inter=None, masks=None,
            activation='relu'):
    """reference vqa_model.py:77-103."""
    dt = sd['cnn.convolutions.0.weight'].dtype      # fp32; tests may pass an fp64 copy for a truth run
    v = cnn_forward(sd, images.to(dt), padding, pooling, training, new_stats, inter)
    q = gru_forward(sd, questions, inter)
    if inter is not None:
        inter['cnn_out'], inter['gru_out'] = v, q
    return head_forward(sd, v, q, inter, masks, activation)


def bce_sum(logits, target):
    """BCEWithLogitsLoss(reduction='sum'): sum max(x,0) - x*t + log1p(exp(-|x|))  (reference vqa_model.py:215)."""
    x = logits
    return torch.sum(torch.clamp(x, min=0) - x * target + torch.log1p(torch.exp(-torch.abs(x))))


def soft_accuracy_sum(logits, target):
    """reference vqa_model.py:127-130: sum_i target[i, argmax_c logits[i]] (caller divides by original_length)."""
    pred = torch.argmax(logits, dim=1)
    return float(target.gather(1, pred[:, None]).sum())


def is_param(name):
    return not (name.endswith('running_mean') or name.endswith('running_var')
                or name.endswith('num_batches_tracked'))


def dropout_noise(keep, p):
    """nn.Dropout's noise tensor from a boolean keep-mask: keep / (1 - p)   (reference vqa_model.py:55)."""
    return keep.to(torch.float32).div_(1.0 - p)


def train_step_grads(sd, images, questions, target, padding=2, pooling='max', masks=None, activation='relu'):
    """One fwd+bwd (reference vqa_model.py:254-256).  Returns (loss, logits, grads, new_running_stats)."""
    p = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in sd.items()}
    new_stats = {}
    logits = forward(p, images, questions, padding, pooling, True, new_stats, None, masks, activation)
    loss = bce_sum(logits, target)
    names = [k for k in p if is_param(k)]
    gs = torch.autograd.grad(loss, [p[k] for k in names])
    return loss.detach(), logits.detach(), dict(zip(names, gs)), new_stats


def clip_grad_norm(grads, max_norm=0.25):
    """nn.utils.clip_grad_norm_(L2) (reference vqa_model.py:259).  Returns (total_norm, clipped grads)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adamax_step(params, grads, state, lr=2e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adamax (weight_decay 0), reference vqa_model.py:218, :266.  state: {name: (step, m, u)}."""
    out = {}
    for k, p in params.items():
        g = grads[k]
        step, m, u = state.get(k, (0, torch.zeros_like(p), torch.zeros_like(p)))
        step += 1
        m = betas[0] * m + (1 - betas[0]) * g
        u = torch.maximum(betas[1] * u, g.abs() + eps)
        out[k] = p - (lr / (1 - betas[0] ** step)) * m / u
        state[k] = (step, m, u)
    return out
