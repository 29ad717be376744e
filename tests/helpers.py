"""Shared helpers for the parity tests: golden loading and the synthetic cases they were made from."""
import os

import numpy as np
import torch

from vqa_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED_W, SEED_X, SAMPLE = 42, 1234, 2048

# name -> hidden, classes, padding, pooling, batch, image size, tokens   (tests/golden/make_golden.py CONFIGS)
CONFIGS = {
    'h1024_p2_max': (1024, 3129, 2, 'max', 4, 224, 14),
    'h512_p0_avg': (512, 3129, 0, 'avg', 2, 224, 14),
    'h64_p5_max': (64, 50, 5, 'max', 3, 224, 9),
    'h128_p2_max_448': (128, 50, 2, 'max', 2, 448, 30),
    'h1024_p2_max_b32': (1024, 3129, 2, 'max', 32, 224, 14),      # + parameters after 3 clip + Adamax steps
}
# cases with the reference's dropout > 0: the two keep-masks of the recorded step are part of the golden file
DROPOUT_CONFIGS = {
    'h1024_p2_max_b32_drop': (1024, 3129, 2, 'max', 32, 224, 14),
}


# the reference's other branches: name -> (config, activation, n_layers).  'selu' = nn.SELU + nn.AlphaDropout(p=0.5)
# (vqa_model.py:54-55; the recorded keep-masks of the two sites are in the golden file); n_layers = 2 = stacked nn.GRU
VARIANT_CONFIGS = {
    'h64_p2_max_selu': ((64, 50, 2, 'max', 4, 224, 9), 'selu', 1),
    'h64_p2_max_l2': ((64, 50, 2, 'max', 4, 224, 9), 'relu', 2),
}


def sample_idx(n):
    if n <= SAMPLE:
        return np.arange(n)
    return (np.arange(SAMPLE, dtype=np.int64) * 2654435761 + 12345) % n


def load_case(name):
    activation, n_layers = 'relu', 1
    if name in VARIANT_CONFIGS:
        (hidden, classes, padding, pooling, batch, size, tokens), activation, n_layers = VARIANT_CONFIGS[name]
    else:
        hidden, classes, padding, pooling, batch, size, tokens = (CONFIGS.get(name) or DROPOUT_CONFIGS[name])
    g = np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'))
    vocab = int(g['vocab'])
    case = dict(hidden=hidden, classes=classes, padding=padding, pooling=pooling, batch=batch, size=size,
                tokens=tokens, vocab=vocab, golden=g, activation=activation, n_layers=n_layers,
                sd=synth.state_dict(SEED_W, hidden, classes, vocab, n_layers=n_layers),
                images=synth.images(SEED_X, batch, size),
                questions=synth.questions(SEED_X, batch, tokens, vocab),
                targets=synth.targets(SEED_X, batch, classes))
    if 'dropout_p' in g.files:
        p = float(g['dropout_p'])
        keep = [torch.from_numpy(np.unpackbits(g[k])[:batch * hidden].reshape(batch, hidden).astype(bool))
                for k in ('keep1', 'keep2')]
        if activation == 'relu':
            case.update(dropout=p, keep=keep, noise=[k.to(torch.float32).div_(1.0 - p) for k in keep])
        else:       # nn.AlphaDropout: every site is a (mask, add) pair
            from oracle import vqa_oracle as O
            case.update(keep=keep, noise=[O.alpha_dropout_noise(k, p) for k in keep])
    return case


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def fixture_files(tmp, classes, vocab_words=40):
    """The tiny question JSON + label2ans pickle tests/golden/make_golden.py built the reference model from
    (same words, so the vocabulary has the same 82 rows)."""
    import json
    import pickle
    alphabet = 'abcdefghijklmnopqrstuvwxyz'
    words = [alphabet[i % 26] + alphabet[(i // 26) % 26] + 'x' for i in range(vocab_words)]
    qs = [{'question': ' '.join(words[(3 * i + j) % vocab_words] for j in range(6))} for i in range(2 * vocab_words)]
    qpath = os.path.join(str(tmp), 'questions.json')
    with open(qpath, 'w') as f:
        json.dump({'questions': qs}, f)
    lpath = os.path.join(str(tmp), f'label2ans_{classes}.pkl')
    with open(lpath, 'wb') as f:
        pickle.dump([f'ans{i}' for i in range(classes)], f)
    return qpath, lpath


def build_model(case, tmp, precision):
    """The HIP-backed VQA for a golden case, loaded with the case's synthetic state dict."""
    import vqa_model
    qpath, lpath = fixture_files(tmp, case['classes'])
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': case['hidden'], 'n_layers': case.get('n_layers', 1),
                       'train_question_path': qpath}, lpath, 256, case['padding'], case.get('dropout', 0.0),
                      case['pooling'], case.get('activation', 'relu'), precision=precision)
    assert len(m.gru.word_idx_mappings) == case['vocab']
    m.load_state_dict(case['sd'], strict=True)
    return m
