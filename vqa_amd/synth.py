"""Deterministic synthetic data for the VQA hot path (weights, images, questions, soft-score targets).

Everything here is a pure function of (seed, name, shape): a counter-based 64-bit integer hash
(splitmix64) mapped to floats, so the same tensors can be re-created bit-for-bit in this container
(golden-vector generation, CPU tests) and on the GPU box (parity tests, bench) without shipping them.

Shapes follow SURVEY.md §8(d): images are k/255 stored as fp16 (reference dataset.py:96-98), questions are
int64 [B,T] with ids in [2,V) padded with <pad>=1 at the end (reference gru.py:94-97), targets are dense
fp32 [B,C] with 1..4 labels scored from {0.3,0.6,0.9,1.0} (reference compute_softscore.py:92-103).
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u64(seed, name, n):
    """n 64-bit hashes for stream (seed, name)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.uint64(seed) ^ (np.uint64(zlib.crc32(name.encode())) << np.uint64(32)))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    """float32 tensor, uniform in [lo, hi), from the top 24 bits of the hash."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (hash_u64(seed, name, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def randint(seed, name, shape, lo, hi):
    """int64 tensor, uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (hash_u64(seed, name, n) >> np.uint64(11)) % np.uint64(hi - lo)
    return torch.from_numpy(u.astype(np.int64).reshape(shape) + lo)


def images(seed, batch, size=224, dtype=torch.float16):
    """[B,3,S,S], values k/255 (k in 0..255), stored in fp16 as the reference dataset does."""
    k = randint(seed, 'images', (batch, 3, size, size), 0, 256)
    return (k.to(torch.float32) / 255.0).to(dtype)


def questions(seed, batch, tokens=14, vocab=10000):
    """int64 [B,T]: real length L~U{3..T}, ids U{2..V-1}, then <pad>=1 up to T."""
    ids = randint(seed, 'q_ids', (batch, tokens), 2, vocab)
    length = randint(seed, 'q_len', (batch, 1), 3, tokens + 1)
    pos = torch.arange(tokens).unsqueeze(0)
    return torch.where(pos < length, ids, torch.ones_like(ids))


_SCORES = torch.tensor([0.3, 0.6, 0.9, 1.0])


def targets(seed, batch, classes=3129):
    """fp32 [B,C] soft scores: n~U{1..4} distinct labels per row scored from {0.3,0.6,0.9,1.0}."""
    t = torch.zeros(batch, classes)
    n = randint(seed, 't_n', (batch,), 1, 5)
    lab = randint(seed, 't_lab', (batch, 4), 0, classes)
    sc = randint(seed, 't_sc', (batch, 4), 0, 4)
    for b in range(batch):
        for j in range(int(n[b])):
            t[b, int(lab[b, j])] = _SCORES[int(sc[b, j])]
    return t


def state_dict_spec(hidden, classes, vocab, word_dim=300, img_dim=256, n_layers=1):
    """(name, shape, kind) for every parameter/buffer of the reference VQA module, in state_dict order.

    Names/shapes as printed by importing the reference (reference cnn.py:22-42,
    gru.py:48-52, vqa_model.py:59-75)."""
    spec = []
    chans = [3, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256]
    seq_idx = [0, 3, 7, 10, 14, 17, 21, 24, 28, 31]
    for i, si in enumerate(seq_idx):
        cin, cout = chans[i], chans[i + 1]
        spec += [(f'cnn.convolutions.{si}.weight', (cout, cin, 3, 3), ('fan', cin * 9)),
                 (f'cnn.convolutions.{si}.bias', (cout,), ('fan', cin * 9)),
                 (f'cnn.convolutions.{si + 1}.weight', (cout,), ('range', 0.6, 1.4)),
                 (f'cnn.convolutions.{si + 1}.bias', (cout,), ('range', -0.3, 0.3)),
                 (f'cnn.convolutions.{si + 1}.running_mean', (cout,), ('range', -0.2, 0.2)),
                 (f'cnn.convolutions.{si + 1}.running_var', (cout,), ('range', 0.5, 1.5)),
                 (f'cnn.convolutions.{si + 1}.num_batches_tracked', (), ('zero_long',))]
    H = hidden
    spec += [('gru.word_embedding.weight', (vocab, word_dim), ('range', -1.5, 1.5))]
    for k in range(n_layers):                                # nn.GRU(num_layers=n_layers), gru.py:51
        spec += [(f'gru.encoder.weight_ih_l{k}', (3 * H, word_dim if k == 0 else H), ('fan', H)),
                 (f'gru.encoder.weight_hh_l{k}', (3 * H, H), ('fan', H)),
                 (f'gru.encoder.bias_ih_l{k}', (3 * H,), ('fan', H)),
                 (f'gru.encoder.bias_hh_l{k}', (3 * H,), ('fan', H))]
    for nm, out, inn, bias in [('linear_inside_relu_attention', H, img_dim + H, True),
                               ('linear_after_relu_attention', 1, H, False),
                               ('linear_inside_relu_image', H, img_dim, True),
                               ('linear_inside_relu_question', H, H, True),
                               ('linear_inside_relu_last', H, H, True),
                               ('fc', classes, H, False)]:
        if bias:
            spec.append((f'{nm}.bias', (out,), ('range', 0.0, 0.3)))
        spec.append((f'{nm}.weight_g', (), ('g', out, inn)))
        # fc rows lean negative (trained VQA logits are mostly negative: sparse soft-score targets, no fc bias)
        spec.append((f'{nm}.weight_v', (out, inn), ('fan', inn, -0.5) if nm == 'fc' else ('fan', inn)))
    return spec


def state_dict(seed, hidden, classes, vocab, word_dim=300, img_dim=256, n_layers=1):
    """Deterministic full state_dict for the VQA module (loads into the reference and into the build)."""
    sd = {}
    for name, shape, kind in state_dict_spec(hidden, classes, vocab, word_dim, img_dim, n_layers):
        if kind[0] == 'fan':
            b = 1.0 / float(kind[1]) ** 0.5
            shift = kind[2] * b if len(kind) > 2 else 0.0
            sd[name] = uniform(seed, name, shape, -b + shift, b + shift)
        elif kind[0] == 'range':
            sd[name] = uniform(seed, name, shape, kind[1], kind[2])
        elif kind[0] == 'zero_long':
            sd[name] = torch.zeros((), dtype=torch.long)
        elif kind[0] == 'g':
            # a scalar g that is NOT ||v||, so the weight-norm scaling is exercised
            _, out, inn = kind
            sd[name] = uniform(seed, name, (), 0.8, 1.2) * (out / 3.0) ** 0.5
    return sd
