"""Question encoder: drop-in for the reference's `gru.GRU` (gru.py:23-106), HIP-backed.

Same constructor, same parameter names (`word_embedding.weight`, `encoder.{weight,bias}_{ih,hh}_l0`), same text
helpers (`preprocess_question_string`, `words_to_idx`), same output: the hidden state after the LAST of the T
tokens (pads included, gru.py:105).  The vocabulary is built without torchtext: specials first, then words by
descending count, ties alphabetical, min_freq 2 (the legacy torchtext.vocab.Vocab order, gru.py:39).

Forward/backward: embedding gather -> one [T*B,E]x[E,3H] input-projection MFMA GEMM -> per-timestep recurrent
GEMM + fused gate kernel; BPTT with saved gates, weight gradients as two reduction-over-(T*B) MFMA GEMMs.
n_layers > 1 (gru.py:51) stacks the same recurrence: layer k reads layer k-1's output sequence, and its
input-projection gradient is added to every step of the lower layer's BPTT (vqa_gru_bwd_steps' dh_seq).
"""
import json
import re
from collections import defaultdict

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

NUMBERS = {'0': 'zero', '1': 'one', '2': 'two', '3': 'three', '4': 'four', '5': 'five', '6': 'six', '7': 'seven',
           '8': 'eight', '9': 'nine', '10': 'ten'}


def build_vocab(counts, min_freq=2, specials=('<unk>', '<pad>')):
    words = sorted(counts.items(), key=lambda kv: kv[0])
    words.sort(key=lambda kv: kv[1], reverse=True)
    itos = list(specials) + [w for w, c in words if c >= min_freq and w not in specials]
    return {w: i for i, w in enumerate(itos)}, itos


class _GruParams(nn.Module):
    """Holds nn.GRU's parameter names/shapes/init (weight_ih_l{k}, weight_hh_l{k}, bias_ih_l{k}, bias_hh_l{k} in
    nn.GRU's registration order; layer k > 0 reads the H outputs of layer k-1) without instantiating nn.GRU (whose
    .cuda() would route through MIOpen's weight flattening)."""

    def __init__(self, input_size, hidden_size, num_layers=1):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        k = 1.0 / hidden_size ** 0.5
        mk = lambda *shape: nn.Parameter(torch.empty(*shape).uniform_(-k, k))
        for layer in range(num_layers):
            setattr(self, f'weight_ih_l{layer}', mk(3 * hidden_size, input_size if layer == 0 else hidden_size))
            setattr(self, f'weight_hh_l{layer}', mk(3 * hidden_size, hidden_size))
            setattr(self, f'bias_ih_l{layer}', mk(3 * hidden_size))
            setattr(self, f'bias_hh_l{layer}', mk(3 * hidden_size))

    def layer(self, k):
        return tuple(getattr(self, f'{n}_l{k}') for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))


import os
# One-launch recurrence (bf16, H % 256 == 0, 48*H*2 bytes of W_hh per workgroup in LDS).  Off by default: the GRU
# runs on a side stream under the CNN, where the per-step launches fill idle CUs while a persistent grid that spins
# on its device-wide barrier holds CUs and LDS away from the conv kernels (measured 8.20 vs 7.89 ms/step).
USE_PERSISTENT = os.environ.get('VQA_GRU_PERSISTENT', '0') != '0'
# workgroups (= CUs: one fits per CU) the persistent recurrence may hold; 0 = as many as are resident.  H/16 (one
# batch slab) is the smallest grid: 64 CUs at H = 1024, the other 192 stay with the CNN.
PERSISTENT_MAX_WG = int(os.environ.get('VQA_GRU_MAX_WG', '0'))
# Default recurrence (bf16, H % 64 == 0): ONE launch per time step - the recurrent GEMM streams its operands, kept in
# MFMA-fragment order, from L2 and applies the gates in its epilogue (csrc/gru_step.hip).  VQA_GRU_FUSED_STEP=0: GEMM + split-K reduce + gate kernel per step.
FUSED_STEP = os.environ.get('VQA_GRU_FUSED_STEP', '1') != '0'


def _persistent_ok(dtype, H):
    return USE_PERSISTENT and dtype == torch.bfloat16 and H % 256 == 0 and 48 * (2 * H + 32) <= 160 * 1024


def _fused_ok(dtype, H):
    return FUSED_STEP and dtype == torch.bfloat16 and H >= 64 and H % 64 == 0


def _packed(B, k):
    """Elements of one fragment-order activation of the fused step kernels (rows padded to whole 64-row groups)."""
    return -(-B // 64) * 64 * k


def _recurrence_fwd(x, E, dtype, w_ih, w_hh, b_ih, b_hh, B, T, allow_persistent=True):
    """x [T*B, >=E] (MFMA dtype, time-major) -> h_f f32 / h_t dtype [T+1,B,H] (slot 0 = h_-1 = 0) and the saved gates."""
    H, dev = w_hh.shape[1], x.device
    wih = ops.cast_pad(w_ih, dtype, ops.pad8(E))             # [3H, E] (row stride padded to 8)
    gi = ops.gemm(x[:, :E], wih, bias=b_ih)                  # [T*B, 3H] fp32
    save = torch.empty(T, B, 4 * H, device=dev, dtype=torch.float32)
    persistent = allow_persistent and _persistent_ok(dtype, H)
    fused = not persistent and _fused_ok(dtype, H)
    # slot 0 = h_{-1} = 0: the fused step kernels write it themselves in their first launch; the other forms read it
    mk = torch.empty if fused else torch.zeros
    h_f = mk(T + 1, B, H, device=dev, dtype=torch.float32)            # h_{-1}, h_0 .. h_{T-1}
    h_t = mk(T + 1, B, H, device=dev, dtype=dtype)
    if fused:
        # one launch per time step: the recurrent GEMM with the gates in its epilogue, operands in MFMA-fragment order
        # (csrc/gru_step.hip); W_hh is packed straight from the f32 master
        wf = torch.empty(3 * H * H, device=dev, dtype=dtype)
        L.call('vqa_gru_pack_whh', w_hh.contiguous(), H, wf, None)
        h_p = torch.empty((T + 1) * _packed(B, H), device=dev, dtype=dtype)
        L.call('vqa_gru_fwd_fused', gi, wf, b_hh, h_f, h_t, h_p, save, B, T, H)
        return h_f, h_t, save
    whh = ops.cast_pad(w_hh, dtype)
    if not (persistent and
            L.fits('vqa_gru_fwd_persistent', gi, whh, whh.stride(0), b_hh, h_f, h_t, save, B, T, H,
                   torch.empty(2, device=dev, dtype=torch.int32), PERSISTENT_MAX_WG)):
        # T x (recurrent GEMM + gate kernel), issued by one call (vqa_gru_fwd_steps: same launches, no per-step
        # trip through the binding)
        gh = torch.empty(B, 3 * H, device=dev, dtype=torch.float32)
        splits = ops.gemm_splits(B, 3 * H, H, dtype)
        ws = torch.empty(splits, B, 3 * H, device=dev, dtype=torch.float32) if splits > 1 else None
        L.call('vqa_gru_fwd_steps', L.dt(x), gi, whh, whh.stride(0), b_hh, h_f, h_t, save, gh, ws, splits, B, T, H)
    return h_f, h_t, save


def _recurrence_bwd(x, E, dtype, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh, dh_last, dh_seq, B, T):
    """BPTT of one layer.  dh_last [B,H] / dh_seq [T,B,H] (either may be None): gradients w.r.t. the last / every
    step's output.  Returns (dx [T*B,E] f32, dw_ih, dw_hh, db_ih, db_hh)."""
    H, dev = w_hh.shape[1], x.device
    wih_T = ops.cast_transpose(w_ih, dtype)                  # [E, 3H]: W_ih^T in the MFMA dtype, one launch
    dgi = torch.empty(T * B, 3 * H, device=dev, dtype=dtype)
    dgh = torch.empty(T * B, 3 * H, device=dev, dtype=dtype)
    if dh_seq is not None:
        dh_seq = dh_seq.contiguous().float()
        dh = dh_seq[T - 1].clone() if dh_last is None else dh_last.float() + dh_seq[T - 1]
    elif dh_last is not None:
        dh = dh_last.contiguous().float()
    else:
        dh = torch.zeros(B, H, device=dev, dtype=torch.float32)
    persistent = dh_seq is None and _persistent_ok(dtype, H)
    if not persistent and _fused_ok(dtype, H):
        wb = torch.empty(3 * H * H, device=dev, dtype=dtype)
        L.call('vqa_gru_pack_whh', w_hh.contiguous(), H, None, wb)            # W_hh^T, fragment order
        dgh_p = torch.empty(2 * _packed(B, 3 * H), device=dev, dtype=dtype)
        dh_ws = torch.empty(2, B, H, device=dev, dtype=torch.float32)
        L.call('vqa_gru_bwd_fused', dh, save, h_f, wb, dgi, dgh, dgh_p, dh_ws, B, T, H, dh_seq)
    else:
        whh_T = ops.cast_pad(w_hh.t().contiguous(), dtype)   # [H, 3H]
        if not (persistent and
                L.fits('vqa_gru_bwd_persistent', whh_T, whh_T.stride(0), h_f, save, dh, dgi, dgh, B, T, H,
                       torch.empty(2, device=dev, dtype=torch.int32), PERSISTENT_MAX_WG)):
            dh_ws = torch.empty(2, B, H, device=dev, dtype=torch.float32)
            splits = ops.gemm_splits(B, H, 3 * H, dtype)
            ws = torch.empty(splits, B, H, device=dev, dtype=torch.float32) if splits > 1 else None
            L.call('vqa_gru_bwd_steps', L.dt(x), dh, save, h_f, whh_T, whh_T.stride(0), dgi, dgh, dh_ws, ws, splits,
                   B, T, H, dh_seq)
    hprev = h_t[:T].reshape(T * B, H)
    # parameter gradients are written into the flat gradient buffer when there is one (ops.grad_dst)
    dw_hh = ops.gemm(dgh, hprev, trans=True, out=ops.grad_dst(w_hh))          # [3H, H]
    dw_ih = ops.gemm(dgi, x[:, :E], trans=True, out=ops.grad_dst(w_ih))       # [3H, E]
    db_ih, db_hh = ops.colsum(dgi, out=ops.grad_dst(b_ih)), ops.colsum(dgh, out=ops.grad_dst(b_hh))
    dx = ops.gemm(dgi, wih_T)                                # [T*B, E] fp32
    return dx, dw_ih, dw_hh, db_ih, db_hh


class _GruFunction(torch.autograd.Function):
    """Embedding + the first nn.GRU layer.  Returns (h_last [B,H] f32, h_seq [T,B,H] f32, h_seq_t [T*B,H] MFMA dtype):
    the sequences feed a stacked layer (num_layers > 1) and are not differentiable views otherwise."""

    @staticmethod
    def forward(ctx, q, dtype, emb, w_ih, w_hh, b_ih, b_hh):
        B, T = q.shape
        E, H = emb.shape[1], w_hh.shape[1]
        Ep = ops.pad8(E)
        q = q.contiguous()
        x = torch.empty(T * B, Ep, device=q.device, dtype=dtype)
        L.call('vqa_embed_fwd', L.dt(x), q, emb, B, T, E, emb.shape[0], x, Ep)
        h_f, h_t, save = _recurrence_fwd(x, E, dtype, w_ih, w_hh, b_ih, b_hh, B, T)
        ctx.save_for_backward(q, x, h_f, h_t, save, w_ih, w_hh, emb, b_ih, b_hh)
        ctx.dtype = dtype
        ctx.set_materialize_grads(False)
        hseq_t = h_t[1:].reshape(T * B, H)
        ctx.mark_non_differentiable(hseq_t)
        return h_f[T], h_f[1:], hseq_t

    @staticmethod
    def backward(ctx, dh_last, dh_seq, _dt=None):
        q, x, h_f, h_t, save, w_ih, w_hh, emb, b_ih, b_hh = ctx.saved_tensors
        B, T = q.shape
        E = emb.shape[1]
        dx, dw_ih, dw_hh, db_ih, db_hh = _recurrence_bwd(x, E, ctx.dtype, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh,
                                                         dh_last, dh_seq, B, T)
        demb = ops.grad_dst(emb)
        if demb is None:
            demb = torch.empty_like(emb)
        L.call('vqa_zero', demb, demb.numel() * 4)            # (the scatter below writes only the rows that occur)
        ws = torch.empty(L.lib().vqa_embed_bwd_ws_bytes(B, T, E), device=q.device, dtype=torch.uint8)
        L.call('vqa_embed_bwd', q, dx, B, T, E, emb.shape[0], dx.stride(0), demb, ws)
        return None, None, demb, dw_ih, dw_hh, db_ih, db_hh


class _GruStackedFunction(torch.autograd.Function):
    """nn.GRU layer k > 0 (gru.py:51 num_layers): its input is the H-wide output sequence of the layer below
    (x_f [T,B,H] f32 carries the gradient, x_t [T*B,H] is the same values in the MFMA dtype).  Same outputs as
    _GruFunction."""

    @staticmethod
    def forward(ctx, x_f, x_t, dtype, w_ih, w_hh, b_ih, b_hh):
        T, B, E = x_f.shape
        H = w_hh.shape[1]
        h_f, h_t, save = _recurrence_fwd(x_t, E, dtype, w_ih, w_hh, b_ih, b_hh, B, T, allow_persistent=False)
        ctx.save_for_backward(x_t, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh)
        ctx.dtype, ctx.dims = dtype, (T, B, E)
        ctx.set_materialize_grads(False)
        hseq_t = h_t[1:].reshape(T * B, H)
        ctx.mark_non_differentiable(hseq_t)
        return h_f[T], h_f[1:], hseq_t

    @staticmethod
    def backward(ctx, dh_last, dh_seq, _dt=None):
        x_t, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh = ctx.saved_tensors
        T, B, E = ctx.dims
        dx, dw_ih, dw_hh, db_ih, db_hh = _recurrence_bwd(x_t, E, ctx.dtype, h_f, h_t, save, w_ih, w_hh, b_ih, b_hh,
                                                         dh_last, dh_seq, B, T)
        return dx.reshape(T, B, E), None, None, dw_ih, dw_hh, db_ih, db_hh


class GRU(nn.Module):
    def __init__(self, word_embd_dim, hidden_dim, n_layers, train_question_path, precision='bf16'):
        super(GRU, self).__init__()
        if n_layers < 1:
            raise ValueError('n_layers must be >= 1')
        self.n_layers = n_layers
        self.device = 'cuda:0' if torch.cuda.is_available() else 'cpu'
        self.train_question_path = train_question_path
        self.precision = precision
        self.word_dict = self.get_vocabs_counts()
        self.word_idx_mappings, self.idx_word_mappings = build_vocab(self.word_dict, min_freq=2)
        self.word_embedding = nn.Embedding(len(self.word_idx_mappings), word_embd_dim)   # N(0,1), trainable
        self.encoder = _GruParams(word_embd_dim, hidden_dim, n_layers)

    @staticmethod
    def preprocess_question_string(question):
        """reference gru.py:55-68: capitalise the first letter, lower the rest, digits 0-10 to words, any other
        token containing a digit -> '<number>', strip non-alphanumerics."""
        result = question[0].upper() + question[1:].lower()
        words = [NUMBERS[word] if word in NUMBERS else word for word in result.split(' ')]
        return [('<number>' if any(ch.isdigit() for ch in word) else re.sub(r'[\W_]+', '', word)) for word in words]

    def get_vocabs_counts(self):
        word_dict = defaultdict(int)
        with open(self.train_question_path) as json_file:
            for q_object in json.load(json_file)['questions']:
                for word in self.preprocess_question_string(q_object['question']):
                    word_dict[word] += 1
        return word_dict

    def words_to_idx(self, question: list):
        """reference gru.py:87-99: indices, <unk> for unknown words, trimmed / <pad>-padded to 14."""
        unk, pad = self.word_idx_mappings['<unk>'], self.word_idx_mappings['<pad>']
        idx = [self.word_idx_mappings.get(word, unk) for word in question][:14]
        return torch.tensor(idx + [pad] * (14 - len(idx))).long()

    def forward(self, questions_batch):
        if not questions_batch.is_cuda:
            raise L.VqaHipError('vqa_amd.GRU runs on an MI355X only (no CPU fallback)')
        dtype = torch.bfloat16 if self.precision == 'bf16' else torch.float32
        e = self.encoder
        h_last, h_seq, h_seq_t = _GruFunction.apply(questions_batch, dtype, self.word_embedding.weight, *e.layer(0))
        for k in range(1, e.num_layers):                    # stacked layers (no dropout between them: nn.GRU default)
            h_last, h_seq, h_seq_t = _GruStackedFunction.apply(h_seq, h_seq_t, dtype, *e.layer(k))
        if dtype != torch.float32:
            T, B = questions_batch.shape[1], questions_batch.shape[0]
            h_last._vqa_lowp = h_seq_t[(T - 1) * B:]        # the recurrence's own MFMA-dtype h_T (for the head's GEMMs)
        return h_last                                       # gru.py:105: the top layer's hidden state after token T
