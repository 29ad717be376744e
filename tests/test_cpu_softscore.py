"""vqa_amd.softscore against vectors made by the reference's compute_softscore.py (tests/golden/softscore.json,
generator: tests/golden/make_softscore_golden.py).  Exact: strings, labels and scores must be equal."""
import json
import os
import pickle

import numpy as np

from vqa_amd import softscore as S

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'softscore.json')) as f:
    GOLD = json.load(f)


def test_normalize_answer_matches_reference_vectors():
    assert len(GOLD['normalize']) > 200
    bad = [(a, S.normalize_answer(a), want) for a, want in GOLD['normalize'] if S.normalize_answer(a) != want]
    assert not bad, bad[:5]


def test_soft_score():
    for n, want in GOLD['score']:
        assert S.soft_score(n) == want


def _plain(targets):
    return [dict(t, label_counts=sorted([list(kv) for kv in t['label_counts'].items()])) for t in targets]


def test_vocab_and_targets_match_reference(tmp_path):
    a2l, l2a, train_t, val_t = S.prepare(GOLD['train_annotations'], GOLD['val_annotations'], str(tmp_path),
                                         GOLD['min_occurrence'])
    assert l2a == GOLD['label2ans'] and len(l2a) > 10
    assert a2l == {a: i for i, a in enumerate(l2a)}
    assert _plain(train_t) == GOLD['train_targets']
    assert _plain(val_t) == GOLD['val_targets']
    # the cache files have the reference's names and load back to the same objects
    for name, obj in (('train_ans2label', a2l), ('train_label2ans', l2a), ('train_target', train_t), ('val_target', val_t)):
        with open(tmp_path / f'{name}.pkl', 'rb') as f:
            assert pickle.load(f) == obj
    assert any(not t['labels'] for t in train_t + val_t)        # the fixture has unanswerable questions


def test_sparse_and_dense_forms_agree():
    a2l, l2a, train_t, _ = S.prepare(GOLD['train_annotations'], GOLD['val_annotations'])
    kept, questions, original = S.drop_unanswered(train_t, list(range(len(train_t))))
    assert original == len(train_t) and 0 < len(kept) < original and all(t['labels'] for t in kept)
    assert [train_t[i] for i in questions] == kept
    labels, scores = S.pack_sparse(kept)
    assert labels.dtype == np.int64 and scores.dtype == np.float32 and labels.shape == (len(kept), 10)
    dense = S.densify(kept, len(l2a))
    rebuilt = np.zeros_like(dense)
    for i in range(len(kept)):
        for lab, sc in zip(labels[i], scores[i]):
            if lab >= 0:
                rebuilt[i, lab] = sc
    assert np.array_equal(dense, rebuilt)
    assert ((labels < 0) == (scores == 0)).all()
