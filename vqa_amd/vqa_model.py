"""Drop-in for the reference's `vqa_model.VQA` / `vqa_model.evaluate` (vqa_model.py:20-141), HIP-backed.

`VQA(gru_params, label2ans_path, img_feature_dim, padding, dropout, pooling, activation)` and
`model(images_batch, questions_batch) -> logits [B, num_classes]` are unchanged; parameter names match the
reference's state_dict (six weight-normalised linears expose `weight_g` (0-dim) / `weight_v` / `bias`).

The attention + fusion + classifier head (vqa_model.py:83-103) is one torch.autograd.Function over hand-written
HIP kernels: the q-term of the concat-attention linear is hoisted (W_att = [W_v | W_q], so
relu(W_att [v;q] + b) = relu(W_v v + (W_q q + b)) is computed once per sample instead of once per region), the
region softmax / weighted sum is one kernel per sample, every linear is an MFMA GEMM with bias/ReLU fused.
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import cnn, gru, ops


import os

_SIDE = {}
SIDE_STREAM = os.environ.get('VQA_SIDE_STREAM', '1') != '0'


def _side_stream(device):
    """One extra HIP stream per device (module scope: nothing native is stored on the nn.Module)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class RandomHorizontalFlip(nn.Module):
    """torchvision.transforms.RandomHorizontalFlip as the reference uses it (vqa_model.py:41, :250): one draw
    flips the whole batch along the width axis."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def draw(self):
        """The coin of one forward() call (same draw from torch's CPU generator)."""
        return bool(torch.rand(1) < self.p)

    def forward(self, img):
        return img.flip(-1) if self.draw() else img


class WeightNormLinear(nn.Module):
    """Parameter container for weight_norm(nn.Linear(in, out, bias), dim=None): `weight_g` (0-dim), `weight_v`
    [out, in] and `bias`, initialised like torch (g = ||v||_F).  The arithmetic lives in the HIP head."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=bias)
        self.in_features, self.out_features = in_features, out_features
        if bias:
            self.bias = nn.Parameter(lin.bias.detach().clone())
        else:
            self.register_parameter('bias', None)
        self.weight_g = nn.Parameter(lin.weight.detach().norm().clone())
        self.weight_v = nn.Parameter(lin.weight.detach().clone())


def alpha_dropout_noise(keep, p=0.5):
    """nn.AlphaDropout's noise as torch forms it (ATen Dropout.cpp, alpha dropout): with alpha' = 1.7580993408473766
    and a = ((alpha'^2 p + 1)(1 - p))^-1/2 the output is x * (a * keep) + ((keep - 1) * alpha' * a + alpha' * a * p).
    keep: 0/1 tensor.  Returns (mask, add) for the head's dropout sites."""
    alpha = 1.7580993408473766
    a = 1.0 / ((alpha * alpha * p + 1.0) * (1.0 - p)) ** 0.5
    keep = keep.float()
    return keep * a, (keep - 1.0) * (alpha * a) + alpha * a * p


def _lowp_hint(t, shape, dtype):
    """The MFMA-dtype copy of an encoder output that its producer attached to it (`_vqa_lowp`), if it fits."""
    h = getattr(t, '_vqa_lowp', None)
    if (h is not None and dtype != torch.float32 and h.dtype == dtype and tuple(h.shape) == tuple(shape) and
            h.is_contiguous() and h.device == t.device):
        return h
    return None


HEAD_LAYERS = ['linear_inside_relu_attention', 'linear_after_relu_attention', 'linear_inside_relu_image',
               'linear_inside_relu_question', 'linear_inside_relu_last', 'fc']


# VQA_HEAD_DQ_STREAM=1: the question encoder's gradient (three launches that only its BPTT waits for) is formed on the
# question encoder's stream instead of the chain that leads to the CNN's backward.  Off: measured SLOWER (same-box A/B,
# 4 alternations of 60 steps: 4.169 vs 4.139 ms/step) - the two event hops between hardware queues cost the chain more
# than three 6 us kernels do.
HEAD_DQ_STREAM = os.environ.get('VQA_HEAD_DQ_STREAM', '0') != '0'
# VQA_HEAD_DQ_SIDE=1: those three launches lead the head's parameter-gradient work on the WEIGHT-GRADIENT stream (which
# that work is ordered behind the chain by one event anyway) and the question encoder's stream waits for them: the chain
# to the CNN's backward neither runs them nor records anything more.  Off: 29 us of kernels less on the chain buy 0 to
# 0.3 % (150-step runs alternated on one box: 3.954-3.966 against 3.956-3.959 ms/step on a fast box, 4.057-4.075
# against 4.070-4.090 on a slower one) - what the chain sheds still runs beside it on the same CUs.
HEAD_DQ_SIDE = os.environ.get('VQA_HEAD_DQ_SIDE', '0') != '0'
# VQA_HEAD_WN_EARLY=1: the six weight norms of the head (W = g v / ||v||: two launches, ~30 us, parameters only) are
# issued at the start of forward() on the weight-gradient stream, which is idle during the forward pass, instead of on
# the chain between the CNN's forward and backward.  Off: measured SLOWER in the same A/B (4.109 / 4.126 ms/step) - the
# 28 MB of bf16 matrices are written 1.2 ms before the head's GEMMs read them, and the CNN's forward evicts them from
# the Infinity Cache in between (round 1 saw the same with them on the question encoder's stream).
HEAD_WN_EARLY = os.environ.get('VQA_HEAD_WN_EARLY', '0') != '0'
# VQA_HEAD_FUSE_ELEMENTWISE=0: vqa_hadamard_fwd / vqa_relu_bwd as launches of their own instead of GEMM epilogues
# (vqa_gemm's emul / emode).  The head's phase is the one stretch of the step where nothing else runs, so a launch there
# costs its latency in full: four 256-byte fills put into it cost the step 0.020 ms, ten 0.040 (150-step runs).
HEAD_FUSE_ELEMENTWISE = os.environ.get('VQA_HEAD_FUSE_ELEMENTWISE', '1') != '0'
# Head parameter gradients on the weight-gradient stream (see _HeadFunction.backward); VQA_HEAD_WGRAD_STREAM=0: inline
HEAD_WGRAD_STREAM = os.environ.get('VQA_HEAD_WGRAD_STREAM', '1') != '0'


class _HeadFunction(torch.autograd.Function):
    """(v [B,k,D] fp32, q [B,H] fp32, weights) -> (logits [B,C] fp32, loss or None).
    With `target` (dense [B,C] or the sparse pair) the classifier GEMM forms the sum-BCE loss and sigmoid(x) - t in its
    epilogue (ops.fc_bce): the second output is the loss, and the backward starts from the saved gradient."""

    @staticmethod
    def forward(ctx, v, q, dtype, masks, target, g_att, v_att, b_att, g_s, v_s, g_img, v_img, b_img, g_q, v_q, b_q,
                g_last, v_last, b_last, g_fc, v_fc, act=1, pre_wn=None):
        # act: 1 = nn.ReLU, 3 = nn.SELU (vqa_model.py:54).  masks = (mask1, mask2, add1, add2): the two dropout sites
        # compute x * mask + add (nn.Dropout: add None; nn.AlphaDropout: see _alpha_dropout_noise)
        B, K, D = v.shape
        H, C = v_q.shape[0], v_fc.shape[0]
        dev = v.device
        v = v.contiguous()
        q = q.contiguous()
        mask1, mask2, add1, add2 = masks
        # all six weight-normed layers: W = g * v / ||v||_F (and W^T for the backward) in two launches
        # (pre_wn: the same, issued earlier on another stream by VQA.forward - HEAD_WN_EARLY)
        wn, ss = pre_wn if pre_wn is not None else ops.weight_norm_fwd_all(
            [(v_att, g_att, True), (v_s, g_s, False), (v_img, g_img, True), (v_q, g_q, True), (v_last, g_last, True),
             (v_fc, g_fc, True)], dtype)
        (W_att, W_att_T), (w_s, _), (W_img, W_img_T), (W_q, W_q_T), (W_last, W_last_T), (W_fc, W_fc_T) = wn

        # the encoders hand over the MFMA-dtype copies they already hold (the last block's bf16 output, the recurrence's
        # bf16 h_T) beside their f32 outputs: two cast kernels less on the chain between the CNN's forward and backward
        v_t = _lowp_hint(v, (B, K, D), dtype)
        v_t = v_t.reshape(B * K, D) if v_t is not None else ops.cast_pad(v.reshape(B * K, D), dtype)
        q_t = _lowp_hint(q, (B, H), dtype)
        if q_t is None:
            q_t = ops.cast_pad(q, dtype)
        qa = ops.gemm(q_t, W_att[:, D:], bias=b_att)                                        # [B,H]
        a = ops.gemm(v_t, W_att[:, :D], out_dtype=dtype, rowadd=qa, group=K, act=act)       # [B*K,H]
        p = torch.empty(B, K, device=dev, dtype=torch.float32)
        vhat = torch.empty(B, D, device=dev, dtype=torch.float32)
        vhat_t = torch.empty(B, D, device=dev, dtype=dtype)
        L.call('vqa_attn_fwd', L.dt(a), a, w_s, v, B, K, H, D, p, vhat, vhat_t)
        fv = ops.gemm(vhat_t, W_img, bias=b_img, act=act)
        m0 = torch.empty(B, H, device=dev, dtype=dtype)
        if HEAD_FUSE_ELEMENTWISE and mask1 is None and add1 is None and dtype == torch.bfloat16:
            # no dropout: the Hadamard product leaves the second projection's GEMM as its bf16 copy (one launch less
            # on the chain between the CNN's forward and backward; same bits as vqa_hadamard_fwd)
            fq = ops.gemm(q_t, W_q, bias=b_q, act=act, c2=m0, emul=fv, emode=1)
        else:
            fq = ops.gemm(q_t, W_q, bias=b_q, act=act)
            L.call('vqa_hadamard_fwd', L.dt(m0), fv, fq, mask1, B * H, m0, add1)
        m_t = torch.empty(B, H, device=dev, dtype=dtype)
        if mask2 is None and add2 is None and dtype == torch.bfloat16:
            m = ops.gemm(m0, W_last, bias=b_last, act=act, c2=m_t)    # no dropout: the GEMM writes the bf16 copy itself
        else:
            m = ops.gemm(m0, W_last, bias=b_last, act=act)
            L.call('vqa_mask_mul', L.dt(m_t), m, mask2, B * H, m_t, add2)
        if target is None:
            logits, loss, dl = ops.gemm(m_t, W_fc), None, None
        else:
            logits, loss, dl = ops.fc_bce(m_t, W_fc, target)
            loss = loss.reshape(())
        ctx.save_for_backward(v, q_t, v_t, a, w_s, p, vhat_t, fv, fq, m0, m, m_t, W_att_T, W_img_T, W_q_T, W_last_T,
                              W_fc_T, ss, g_att, v_att, g_s, v_s, g_img, v_img,
                              g_q, v_q, g_last, v_last, g_fc, v_fc, b_att, b_img, b_q, b_last,
                              *[t for t in (mask1, mask2) if t is not None], *([dl] if dl is not None else []))
        ctx.dims, ctx.dtype, ctx.has_masks = (B, K, D, H, C), dtype, (mask1 is not None, mask2 is not None)
        ctx.act = act
        ctx.q_stream = getattr(q, '_vqa_stream', None)
        ctx.fused_loss = dl is not None
        ctx.set_materialize_grads(False)
        return logits, loss

    @staticmethod
    def backward(ctx, dlogits, dloss=None):
        (v, q_t, v_t, a, w_s, p, vhat_t, fv, fq, m0, m, m_t, W_att_T, W_img_T, W_q_T, W_last_T, W_fc_T, ss,
         g_att, v_att, g_s, v_s, g_img, v_img, g_q, v_q, g_last, v_last, g_fc,
         v_fc, b_att, b_img, b_q, b_last) = ctx.saved_tensors[:34]
        extra = list(ctx.saved_tensors[34:])
        mask1 = extra.pop(0) if ctx.has_masks[0] else None
        mask2 = extra.pop(0) if ctx.has_masks[1] else None
        dl_saved = extra.pop(0) if ctx.fused_loss else None
        B, K, D, H, C = ctx.dims
        dtype, dev, act = ctx.dtype, v.device, ctx.act
        new = lambda *s, dt=dtype: torch.empty(*s, device=dev, dtype=dt)

        ops.defer_reset()
        # ---- data path first: what the image and question encoders wait for
        if dl_saved is not None and dloss is not None and dlogits is None:
            # the loss came out of the classifier's epilogue: d logits = (sigmoid(x) - t) * d loss, scaled in the cast
            dl_t = ops.cast_pad(dl_saved, dtype, ops.pad8(C), scale=dloss.reshape(1).float())
        else:
            if dl_saved is not None and dloss is not None:               # the logits fed something else as well
                dlogits = dlogits + dl_saved * dloss
            if dlogits is None:
                dlogits = torch.zeros(B, C, device=dev, dtype=torch.float32)
            dl_t = ops.cast_pad(dlogits.contiguous(), dtype, ops.pad8(C))        # [B,C] (padded rows)
        if HEAD_FUSE_ELEMENTWISE and mask2 is None:
            # no dropout: the activation's derivative is applied in the epilogue of the GEMM that forms dm (its
            # split-K reduce pass): one launch less on the chain, same bits as vqa_relu_bwd
            dpre_last = ops.gemm(dl_t, W_fc_T, out_dtype=dtype, emul=m, emode=16 + act)      # [B,H]
        else:
            dm = ops.gemm(dl_t, W_fc_T)                                          # [B,H]
            dpre_last = new(B, H)
            L.call('vqa_relu_bwd', L.dt(dpre_last), dm, m, mask2, B * H, dpre_last, act)
        dpv, dpq = new(B, H), new(B, H)
        if HEAD_FUSE_ELEMENTWISE and mask1 is None and dtype == torch.bfloat16:
            # (no dropout) both outputs of the Hadamard product's backward from the GEMM that forms dm0
            ops.gemm(dpre_last, W_last_T, out=dpv, c2=dpq, emul=fv, emul2=fq, emode=32 + act)
        else:
            dm0 = ops.gemm(dpre_last, W_last_T)
            L.call('vqa_hadamard_bwd', L.dt(dpv), dm0, fv, fq, mask1, B * H, dpv, dpq, act)
        dvhat = ops.gemm(dpv, W_img_T)                                           # [B,D]
        dv = new(B * K, D, dt=torch.float32)
        dpre_a = new(B * K, H)
        dws_part = new(B, H, dt=torch.float32)
        L.call('vqa_attn_bwd', L.dt(a), a, w_s, v, p, dvhat, B, K, H, D, dv, dpre_a, dws_part, act)
        ops.gemm(dpre_a, W_att_T[:D], out=dv, accumulate=True)                   # += dpre_a @ W_v
        # dv is what the image encoder's backward - the critical chain - waits for.  The question encoder's gradient
        # (three more launches) only feeds the recurrence's BPTT, which ends long before the CNN's backward does: it
        # leads the parameter-gradient work on the weight-gradient stream (HEAD_DQ_SIDE), or is formed on the stream the
        # BPTT will run on (HEAD_DQ_STREAM; ctx.q_stream), or stays on this chain
        def question_grad():
            dq = ops.gemm(dpq, W_q_T)                                            # [B,H]
            dqa, dqa_t = new(B, H, dt=torch.float32), new(B, H)
            L.call('vqa_group_sum', L.dt(dpre_a), dpre_a, B, K, H, dqa, dqa_t)
            ops.gemm(dqa_t, W_att_T[D:], out=dq, accumulate=True)                # += dqa @ W_q-part
            return dq, dqa, dqa_t

        main = torch.cuda.current_stream()
        dq_side = HEAD_DQ_SIDE and HEAD_WGRAD_STREAM
        qs = ctx.q_stream if (HEAD_DQ_STREAM and not dq_side and ctx.q_stream is not None) else None
        if qs is not None and qs != main:
            qs.wait_event(main.record_event())
            for t in (dpq, dpre_a, W_q_T, W_att_T):
                t.record_stream(qs)
        else:
            qs = None
        dq_done = None
        if not dq_side:
            with torch.cuda.stream(qs if qs is not None else main):
                dq, dqa, dqa_t = question_grad()
            dq_done = qs.record_event() if qs is not None else None

        # ---- parameter gradients: nothing downstream in this backward needs them, so they run on the weight-gradient
        # stream beside the encoders' backward (ops.defer_join: joined before backward() returns / before a bucket
        # all-reduce reads them)
        def weights():
            dW_fc = ops.gemm(dl_t, m_t, trans=True)                              # [C,H]
            dW_last = ops.gemm(dpre_last, m0, trans=True)
            db_last = ops.colsum(dpre_last, out=ops.grad_dst(b_last))
            dW_img = ops.gemm(dpv, vhat_t, trans=True)                           # [H,D]
            db_img = ops.colsum(dpv, out=ops.grad_dst(b_img))
            dW_q = ops.gemm(dpq, q_t, trans=True)
            db_q = ops.colsum(dpq, out=ops.grad_dst(b_q))
            dW_s = ops.colsum(dws_part).reshape(1, H)
            db_att = ops.colsum(dqa, out=ops.grad_dst(b_att))
            dW_att = new(H, D + H, dt=torch.float32)
            ops.gemm(dpre_a, v_t, trans=True, out=dW_att[:, :D])
            ops.gemm(dqa_t, q_t, trans=True, out=dW_att[:, D:])
            wn = ops.weight_norm_bwd_all([(dW_att, v_att, g_att), (dW_s, v_s, g_s), (dW_img, v_img, g_img),
                                          (dW_q, v_q, g_q), (dW_last, v_last, g_last), (dW_fc, v_fc, g_fc)], ss)
            return wn, db_att, db_img, db_q, db_last

        if HEAD_WGRAD_STREAM:
            from .cnn import _wgrad_stream
            side = _wgrad_stream(dev)
            cur = torch.cuda.current_stream()
            side.wait_event(cur.record_event())
            if dq_done is not None:
                side.wait_event(dq_done)                                         # (dqa / dqa_t come from that stream)
            # Autograd hands the returned gradients to each parameter's AccumulateGrad node right after this function,
            # on THIS stream.  With p.grad None and no tensor hooks that node only adopts the tensor (no kernel), so
            # the side-stream work may still be running and is joined later (ops.defer_join).  With an existing
            # p.grad (gradient accumulation, zero_grad(set_to_none=False), a second backward) it runs `p.grad += g`,
            # and a tensor hook reads g: both must see finished gradients, so this stream waits before returning.
            head_params = (g_att, v_att, b_att, g_s, v_s, g_img, v_img, b_img, g_q, v_q, b_q, g_last, v_last, b_last,
                           g_fc, v_fc)
            # (post-accumulate-grad hooks read p.grad right after the adoption, on this stream: GradBuckets' own are
            # marked join-aware - they order their collective behind the side stream themselves - any other hook, a
            # user's, DDP's or a logger's, must find finished gradients)
            adopt_only = all(p.grad is None and not p._backward_hooks and
                             all(getattr(h, '_vqa_join_aware', False)
                                 for h in (getattr(p, '_post_accumulate_grad_hooks', None) or {}).values())
                             for p in head_params)
            with torch.cuda.stream(side):
                if dq_side:
                    dq, dqa, dqa_t = question_grad()
                    dq_ev = side.record_event()
                wn, db_att, db_img, db_q, db_last = weights()
            if dq_side:
                # the recurrence's BPTT runs on the stream its forward ran on (autograd), else on this one
                bptt = ctx.q_stream if ctx.q_stream is not None else cur
                bptt.wait_event(dq_ev)
                for t in (dq, dqa, dqa_t):
                    t.record_stream(cur)                # (made on the side stream; dq travels through autograd from here)
                    if bptt != cur:
                        t.record_stream(bptt)
                for t in (W_q_T, W_att_T):
                    t.record_stream(side)
            for t in (dl_t, m_t, dpre_last, m0, dpv, vhat_t, dpq, q_t, dws_part, dqa, dpre_a, v_t, dqa_t, ss):
                t.record_stream(side)                                            # (freed on this stream, read on that)
            if adopt_only:
                ops.defer_join(side, outputs=[x for pair in wn for x in pair] + [db_att, db_img, db_q, db_last])
            else:
                cur.wait_event(side.record_event())
                for t in [x for pair in wn for x in pair] + [db_att, db_img, db_q, db_last]:
                    t.record_stream(cur)                                         # (allocated on that stream, read on this)
        else:
            if dq_done is not None:
                torch.cuda.current_stream().wait_event(dq_done)
            wn, db_att, db_img, db_q, db_last = weights()
        if dq_done is not None:
            for t in (dqa, dqa_t, dq):
                t.record_stream(main)           # (made on the question stream; dq travels through autograd from here)
        (dg_att, dv_att), (dg_s, dv_s), (dg_img, dv_img), (dg_q, dv_q), (dg_last, dv_last), (dg_fc, dv_fc) = wn
        return (dv.reshape(B, K, D), dq, None, None, None, dg_att, dv_att, db_att, dg_s, dv_s, dg_img, dv_img, db_img,
                dg_q, dv_q, db_q, dg_last, dv_last, db_last, dg_fc, dv_fc, None, None)


class VQA(nn.Module):
    def __init__(self, gru_params: dict, label2ans_path: str, img_feature_dim: int, padding: int, dropout: float,
                 pooling: str, activation: str, precision: str = 'bf16'):
        """Arguments as the reference (vqa_model.py:21-32); `precision` ('bf16' | 'fp32') is the only addition."""
        super(VQA, self).__init__()
        self.device = 'cuda:0' if torch.cuda.is_available() else 'cpu'
        self.precision = precision
        self.cnn = cnn.CNN(padding=padding, pooling=pooling, precision=precision)
        self.padding = padding
        self.pooling = pooling
        self.flip = RandomHorizontalFlip(p=0.5)
        self.gru = gru.GRU(gru_params['word_embd_dim'], gru_params['question_hidden_dim'], gru_params['n_layers'],
                           gru_params['train_question_path'], precision=precision)
        self.word_embd_dim = gru_params['word_embd_dim']
        self.question_hidden_dim = gru_params['question_hidden_dim']
        self.hidden_dim = gru_params['question_hidden_dim']
        self.n_layers = gru_params['n_layers']

        with open(label2ans_path, 'rb') as f:
            self.lbl2ans = pickle.load(f)
        self.num_classes = len(self.lbl2ans)
        self.activation = activation
        self.dropout_p = dropout
        H = self.hidden_dim
        self.linear_inside_relu_attention = WeightNormLinear(img_feature_dim + H, H)
        self.linear_after_relu_attention = WeightNormLinear(H, 1, bias=False)
        self.linear_inside_relu_image = WeightNormLinear(img_feature_dim, H)
        self.linear_inside_relu_question = WeightNormLinear(H, H)
        self.linear_inside_relu_last = WeightNormLinear(H, H)
        self.fc = WeightNormLinear(H, self.num_classes, bias=False)

    def set_precision(self, precision):
        self.precision = self.cnn.precision = self.gru.precision = precision
        return self

    ALPHA_DROPOUT_P = 0.5       # the reference's selu branch builds nn.AlphaDropout() with torch's default p

    def _dropout_mask(self, B, dev):
        """(mask, add) of one dropout site, out = x * mask + add; (None, None) = identity (eval, or nn.Dropout(0)).
        activation == 'relu': nn.Dropout(dropout) - inverted dropout, keep-mask scaled by 1/(1-p).
        any other activation: nn.AlphaDropout() (vqa_model.py:55 ignores `dropout` there: p = 0.5)."""
        if not self.training:
            return None, None
        if self.activation == 'relu':
            if self.dropout_p <= 0.0:
                return None, None
            keep = 1.0 - self.dropout_p
            return torch.empty(B, self.hidden_dim, device=dev).bernoulli_(keep).div_(keep), None
        return alpha_dropout_noise(torch.empty(B, self.hidden_dim, device=dev).bernoulli_(1.0 - self.ALPHA_DROPOUT_P),
                                   self.ALPHA_DROPOUT_P)

    def forward_loss(self, images_batch, questions_batch, target, mirror=False, dropout_noise=None):
        """forward() and the reference's criterion (nn.BCEWithLogitsLoss(reduction='sum'), vqa_model.py:215, :255) in
        one: the classifier GEMM's epilogue forms the loss and its gradient, so the logits are written once and never
        re-read (BASELINE.json north_star: "one MFMA GEMM followed by a fused sigmoid-BCE epilogue").  forward() itself
        is unchanged - the reference's loop calls the model and the criterion separately, and that keeps working.
        target: dense fp32 [B,C] or the sparse pair (labels, scores).  Returns (logits, loss)."""
        return self.forward(images_batch, questions_batch, mirror, dropout_noise, _target=target)

    def forward(self, images_batch, questions_batch, mirror=False, dropout_noise=None, _target=None):
        # mirror=True: the batch as model.flip would have flipped it, without the flipped copy (Trainer.step)
        # dropout_noise=(n1, n2): the two nn.Dropout noise tensors [B,H] (keep / (1-p)) of vqa_model.py:101 / :103
        # given by the caller instead of drawn here (parity tests replay the reference's recorded draws)
        if not images_batch.is_cuda:
            raise L.VqaHipError('vqa_amd.VQA runs on an MI355X only: there is no CPU fallback for the hot path')
        # The question encoder (28 small latency-bound launches per direction) is independent of the image
        # encoder: it runs on a side HIP stream, under the CNN kernels.  Autograd replays each backward on its
        # forward's stream, so the GRU's BPTT overlaps the CNN backward the same way.
        pre_wn = None
        if SIDE_STREAM:
            cur = torch.cuda.current_stream()
            if HEAD_WN_EARLY and cnn.WGRAD_STREAM:
                pre_wn = self._weight_norms_early(cur, images_batch.device)
            # (data parallelism, optim.DP_FOLD_STREAMS: the Trainer marks the model and the question encoder shares the
            # weight-gradient stream - two compute streams + RCCL's own)
            side = (cnn._wgrad_stream if getattr(self, '_dp_fold_streams', False) else _side_stream)(images_batch.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                questions_representation = self.gru(questions_batch)    # [B, H]
            images_representation = self.cnn(images_batch, mirror)      # [B, k, d]
            cur.wait_stream(side)
            questions_representation.record_stream(cur)
            questions_representation._vqa_stream = side      # (where its backward will run: _HeadFunction.backward)
        else:
            images_representation = self.cnn(images_batch, mirror)
            questions_representation = self.gru(questions_batch)
        if pre_wn is not None:
            cur.wait_event(pre_wn[2])
        return self.head(images_representation, questions_representation, dropout_noise, _target, _pre_wn=pre_wn)

    def _weight_norms_early(self, cur, dev):
        """The head's six weight norms (parameters only) on the weight-gradient stream, which is idle during the forward
        pass: -> (views, sumsq, event).  Everything it allocates is used on `cur` later (record_stream)."""
        dtype = torch.bfloat16 if self.precision == 'bf16' else torch.float32
        a, s, i, qq, l, f = (getattr(self, n) for n in HEAD_LAYERS)
        side = cnn._wgrad_stream(dev)
        side.wait_event(cur.record_event())         # (the parameters' last writer - the optimizer - ran on `cur`)
        with torch.cuda.stream(side), torch.no_grad():
            wn, ss = ops.weight_norm_fwd_all([(a.weight_v, a.weight_g, True), (s.weight_v, s.weight_g, False),
                                              (i.weight_v, i.weight_g, True), (qq.weight_v, qq.weight_g, True),
                                              (l.weight_v, l.weight_g, True), (f.weight_v, f.weight_g, True)], dtype)
            ev = side.record_event()
        for w, wt in wn:
            for t in (w, wt):
                if t is not None:
                    t.record_stream(cur)
        ss.record_stream(cur)
        return wn, ss, ev

    def head(self, images_representation, questions_representation, dropout_noise=None, target=None, _pre_wn=None):
        """Attention + fusion + classifier (vqa_model.py:83-103) on given encoder outputs v [B,k,256], q [B,H] (fp32):
        the second half of forward(), callable on its own (parity tests feed it the oracle's encoder outputs).
        With `target` it returns (logits, loss) - see forward_loss()."""
        dtype = torch.bfloat16 if self.precision == 'bf16' else torch.float32
        B, dev = images_representation.shape[0], images_representation.device
        if dropout_noise is not None:
            # per site: a multiplicative noise tensor [B,H] (nn.Dropout), a (mask, add) pair (nn.AlphaDropout,
            # alpha_dropout_noise(keep)), or None
            sites = [t if isinstance(t, (tuple, list)) else (t, None) for t in dropout_noise]
            sites = [tuple(None if t is None else t.to(dev, torch.float32).contiguous() for t in st) for st in sites]
            assert all(t is None or tuple(t.shape) == (B, self.hidden_dim) for st in sites for t in st)
        else:
            sites = [self._dropout_mask(B, dev), self._dropout_mask(B, dev)]
        masks = (sites[0][0], sites[1][0], sites[0][1], sites[1][1])
        act = 1 if self.activation == 'relu' else 3         # vqa_model.py:54: anything but 'relu' is nn.SELU()
        a, s, i, qq, l, f = (getattr(self, n) for n in HEAD_LAYERS)
        if target is not None and not isinstance(target, (tuple, list)):
            target = target.float()
        logits, loss = _HeadFunction.apply(images_representation, questions_representation, dtype, masks, target,
                                   a.weight_g, a.weight_v, a.bias, s.weight_g, s.weight_v, i.weight_g, i.weight_v,
                                   i.bias, qq.weight_g, qq.weight_v, qq.bias, l.weight_g, l.weight_v, l.bias,
                                   f.weight_g, f.weight_v, act, None if _pre_wn is None else _pre_wn[:2])
        return logits if target is None else (logits, loss)


VQAModel = VQA   # the name BASELINE.json's north_star uses


class _BceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, scores):
        if scores is None:
            loss, dl = ops.bce_sum(logits, target.contiguous(), want_grad=True)
        else:                                           # sparse form: target = labels int64 [B,L], scores [B,L]
            loss, dl = ops.bce_sum_sparse(logits, target, scores, want_grad=True)
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, go):
        (dl,) = ctx.saved_tensors
        return dl * go, None, None


class SoftScoreBCELoss(nn.Module):
    """nn.BCEWithLogitsLoss(reduction='sum') against dense soft-score targets (vqa_model.py:215) as one fused HIP
    kernel producing the loss and sigmoid(x) - t in a single pass over the logits.  `target` is the reference's
    dense fp32 [B,C] matrix, or - SURVEY.md 8f-3 - the sparse pair (labels int64 [B,L] with -1 for empty slots,
    scores fp32 [B,L]) that dataset.py:135-138 would scatter into it."""

    def forward(self, logits, target):
        sp = ops._sparse(target)
        if sp is not None:
            return _BceSum.apply(logits, sp[0], sp[1])
        return _BceSum.apply(logits, target, None)


def evaluate(dataloader, model, criterion, last_epoch_acc, dataset):
    """reference vqa_model.py:106-141, same signature and return tuple.  The per-sample host loop
    (vqa_model.py:129-130) is replaced by an on-device argmax + gather + sum: one scalar per batch."""
    print(f"============ Evaluating on {'validation' if dataset.phase == 'val' else 'train'} set ============")
    model.eval()
    with torch.no_grad():
        acc_sum = None
        epoch_losses = list()                           # device scalars: ONE host sync per evaluation, not per batch
        for i_batch, batch in enumerate(dataloader):
            images_batch = batch['image'].cuda()
            questions_batch = batch['question'].cuda()
            target = batch['answer']
            target = tuple(t.cuda() for t in target) if isinstance(target, (tuple, list)) else target.cuda()
            output = model(images_batch, questions_batch)
            epoch_losses.append(criterion(output, target).detach().reshape(1).float())
            part = ops.soft_accuracy_sum(output, target)
            acc_sum = part if acc_sum is None else acc_sum + part
        dev = next(model.parameters()).device
        totals = torch.zeros(3, device=dev, dtype=torch.float64)    # accuracy sum, loss sum, batches
        if acc_sum is not None:
            totals[0] = acc_sum.double().sum()
            totals[1] = torch.cat(epoch_losses).double().sum()
            totals[2] = len(epoch_losses)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # data-parallel evaluation: every rank saw its shard of the loader; one 3-scalar all-reduce
            torch.distributed.all_reduce(totals, op=torch.distributed.ReduceOp.SUM)
        accuracy, loss_sum, batches = (float(v) for v in totals.tolist())
        acc = accuracy / dataset.original_length
        print(f"{'Validation' if dataset.phase == 'val' else 'Train'} accuracy = {round(acc, 5)}")
        cur_epoch_loss = loss_sum / batches if batches else float('nan')
        print(f"{'Validation' if dataset.phase == 'val' else 'Train'} loss = {round(cur_epoch_loss, 5)}")
        acc_not_improved = not (acc > last_epoch_acc)
        return cur_epoch_loss, acc_not_improved, acc
