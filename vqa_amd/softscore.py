"""Answer-side preprocessing: VQA-v2 annotations -> answer vocabulary -> soft-score targets.

Host-side text work that feeds the loss kernel; it restates what the reference's compute_softscore.py does
(reference compute_softscore.py:92-239) in this package's own form, and adds the packed sparse target format the
HIP loss / accuracy kernels take directly (`pack_sparse`, consumed by `SoftScoreBCELoss` and `soft_accuracy_sum`).

  normalize_answer(text)                 compute_softscore.py:106-143 (punctuation, number words, articles,
                                         contractions, commas), including its quirks - see the notes in the code
  soft_score(count)                      compute_softscore.py:92-103: 0, 0.3, 0.6, 0.9, then 1
  build_answer_vocab(annotations, 9)     compute_softscore.py:146-189: answers whose normalised
                                         `multiple_choice_answer` is the majority answer of >= 9 questions, labelled
                                         in order of first appearance
  compute_targets(annotations, a2l)      compute_softscore.py:192-239: per question the labels / scores / label
                                         counts of its ten human answers (matched as written, as the reference does)
  write_cache(...)                       the three pickle files the reference's dataset.py:30 and vqa_model.py:49 read
  pack_sparse / densify                  [N, width] label + score arrays (label -1 = empty slot) / dense [N, C] rows
"""
import os
import pickle
import re

import numpy as np

# ---------------------------------------------------------------------------------------------- normalisation
# Contractions (the table of the VQA evaluation script, reference compute_softscore.py:17-61) in a compact form: each
# entry is the corrected spelling, with '^' standing for an apostrophe that the misspelt key lacks.  Keys containing
# capitals in the reference table ("Im", "Ive", ...) can never match the lower-cased words and are left out.
_CONTRACTED = """
ain^t aren^t can^t could^ve couldn^t couldn't^ve couldn^t've didn^t doesn^t don^t hadn^t hadn^t've hadn't^ve hasn^t
haven^t he^d he^d've he'd^ve he^s how^d how^ll how^s isn^t it^d it^d've it'd^ve it^ll let's ma^am mightn^t mightn^t've
mightn't^ve might^ve mustn^t must^ve needn^t not^ve o^clock oughtn^t ^ow's'at 'ow^s'at 'ow's^at shan^t she^d've she'd^ve
she's should^ve shouldn^t shouldn^t've shouldn't^ve somebody^d've somebody'd^ve somebody^ll somebody^s someone^d
someone^d've someone'd^ve someone^ll someone^s something^d something^d've something'd^ve something^ll that^s there^d
there^d've there'd^ve there^re there^s they^d they^d've they'd^ve they^ll they^re they^ve ^twas wasn^t we^d've we'd^ve
we^ve weren^t what^ll what^re what^s what^ve when^s where^d where^s where^ve who^d who^d've who'd^ve who^ll who^s
who^ve why^ll why^re why^s won^t would^ve wouldn^t wouldn^t've wouldn't^ve y^all y^all'll y'all^ll y^all'd've y'all^d've
y'all'd^ve you^d you^d've you'd^ve you^ll you^re you^ve
"""
CONTRACTIONS = {e.replace('^', ''): e.replace('^', "'") for e in _CONTRACTED.split()}
CONTRACTIONS["somebody'd"] = 'somebodyd'          # the one entry the reference table has the wrong way round (:36)

NUMBER_WORDS = {w: str(i) for i, w in enumerate('zero one two three four five six seven eight nine ten'.split())}
NUMBER_WORDS['none'] = '0'
ARTICLES = frozenset(('a', 'an', 'the'))
PUNCTUATION = ';/[]"{}()=+\\_-><@`,?!'
_DIGIT_COMMA_DIGIT = re.compile(r'\d,\d')
# The reference pattern (?!<=\d)(\.)(?!\d) (:76) opens with a look-AHEAD for the literal text "<=" + digit, which can
# never hold in front of a '.', so all it asks is "a full stop not followed by a digit"; and its re.UNICODE flag is
# passed in the `count` position of sub() (:114), which caps the removals at int(re.UNICODE) = 32 per answer.
_LOOSE_STOP = re.compile(r'\.(?!\d)')
_STOP_LIMIT = int(re.UNICODE)


def _strip_punctuation(text):
    # a mark that touches a space anywhere in the ORIGINAL text - or any mark at all when the text holds a
    # digit,digit group - is deleted; otherwise it becomes a space (reference :106-113)
    delete_all = _DIGIT_COMMA_DIGIT.search(text) is not None
    out = text
    for mark in PUNCTUATION:
        gone = delete_all or (mark + ' ') in text or (' ' + mark) in text
        out = out.replace(mark, '' if gone else ' ')
    return _LOOSE_STOP.sub('', out, count=_STOP_LIMIT)


def _words(text):
    kept = []
    for word in text.lower().split():
        word = NUMBER_WORDS.get(word, word)
        if word not in ARTICLES:                   # checked after the number mapping, as in the reference (:122-123)
            kept.append(CONTRACTIONS.get(word, word))
    return kept


def normalize_answer(text):
    """'Two, dogs!' -> '2 dogs' (reference preprocess_answer, compute_softscore.py:140-143)."""
    return ' '.join(_words(_strip_punctuation(text))).replace(',', '')


def soft_score(count):
    """Score of an answer given by `count` of the ten annotators (reference get_score, :92-103)."""
    return (0, 0.3, 0.6, 0.9)[count] if count < 4 else 1


# ---------------------------------------------------------------------------------------------- vocabulary, targets
def build_answer_vocab(annotations, min_occurrence=9):
    """-> (ans2label dict, label2ans list).  An answer is kept when it is the `multiple_choice_answer` (normalised) of
    at least `min_occurrence` DISTINCT questions; labels follow first appearance (reference :146-189)."""
    asked = {}
    for entry in annotations:
        asked.setdefault(normalize_answer(entry['multiple_choice_answer']), set()).add(entry['question_id'])
    label2ans = [ans for ans, qids in asked.items() if len(qids) >= min_occurrence]
    return {ans: i for i, ans in enumerate(label2ans)}, label2ans


def compute_targets(annotations, ans2label):
    """One record per annotation, in order, with the reference's keys (:221-228).  The ten answers are looked up
    exactly as written - the reference's load_v2 (:254-268) never normalises them, whatever its docstring asks -
    so an answer that only matches the vocabulary after normalisation is dropped there and here."""
    records = []
    for entry in annotations:
        tally = {}
        for given in entry['answers']:
            tally[given['answer']] = tally.get(given['answer'], 0) + 1
        known = [(ans2label[ans], n) for ans, n in tally.items() if ans in ans2label]
        records.append({
            'question_id': entry['question_id'],
            'question_type': entry['question_type'],
            'image_id': entry['image_id'],
            'label_counts': {label: n for label, n in known},
            'labels': [label for label, _ in known],
            'scores': [soft_score(n) for _, n in known],
        })
    return records


def write_cache(cache_root, name, ans2label=None, label2ans=None, targets=None):
    """`<name>_ans2label.pkl`, `<name>_label2ans.pkl`, `<name>_target.pkl` under cache_root - the file names and
    contents the reference's VQADataset (dataset.py:30) and VQA.__init__ (vqa_model.py:49) load."""
    os.makedirs(cache_root, exist_ok=True)
    for suffix, obj in (('ans2label', ans2label), ('label2ans', label2ans), ('target', targets)):
        if obj is not None:
            with open(os.path.join(cache_root, f'{name}_{suffix}.pkl'), 'wb') as f:
                pickle.dump(obj, f)


def prepare(train_annotations, val_annotations, cache_root=None, min_occurrence=9):
    """The reference's load_v2 (:254-268) on already-parsed annotation lists: vocabulary from the training split,
    targets for both.  -> (ans2label, label2ans, train_targets, val_targets)."""
    ans2label, label2ans = build_answer_vocab(train_annotations, min_occurrence)
    train_t, val_t = compute_targets(train_annotations, ans2label), compute_targets(val_annotations, ans2label)
    if cache_root is not None:
        write_cache(cache_root, 'train', ans2label, label2ans, train_t)
        write_cache(cache_root, 'val', targets=val_t)
    return ans2label, label2ans, train_t, val_t


# ---------------------------------------------------------------------------------------------- device formats
def drop_unanswered(targets, questions):
    """Questions none of whose answers is in the vocabulary are left out of training and evaluation
    (reference dataset.py:34-38); the pre-filter length stays the accuracy denominator (dataset.py:32)."""
    keep = [i for i, t in enumerate(targets) if t['labels']]
    return [targets[i] for i in keep], [questions[i] for i in keep], len(targets)


def pack_sparse(targets, width=10):
    """-> (labels int64 [N, width], scores float32 [N, width]); unused slots hold label -1 / score 0.  A question has
    ten answers, so at most ten distinct labels: width 10 always fits.  This is the form the HIP loss and accuracy
    kernels take (vqa_amd.ops.bce_sum_sparse), 120 bytes a row instead of the 12.5 KB dense row of dataset.py:135-138."""
    labels = np.full((len(targets), width), -1, dtype=np.int64)
    scores = np.zeros((len(targets), width), dtype=np.float32)
    for i, t in enumerate(targets):
        n = len(t['labels'])
        if n > width:
            raise ValueError(f'record {i} has {n} labels, width is {width}')
        labels[i, :n] = t['labels']
        scores[i, :n] = t['scores']
    return labels, scores


def densify(targets, num_classes):
    """Dense float32 [N, num_classes] rows as the reference's dataset builds them one by one (dataset.py:135-138)."""
    out = np.zeros((len(targets), num_classes), dtype=np.float32)
    for i, t in enumerate(targets):
        out[i, t['labels']] = t['scores']
    return out
