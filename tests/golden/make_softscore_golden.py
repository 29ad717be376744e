"""Golden vectors for vqa_amd.softscore, made by importing the reference's compute_softscore.py (standard library only)
in this container.   Run:  python tests/golden/make_softscore_golden.py   -> tests/golden/softscore.json

Stored: inputs and the reference's outputs only - (answer string -> preprocess_answer), (count -> get_score), and a
small synthetic annotation set -> (label2ans, targets).  The answer strings are written here; every lower-case key of
the reference's contraction table is also fed through preprocess_answer as one more input.
"""
import contextlib
import io
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
import compute_softscore as ref  # noqa: E402

ANSWERS = [
    'yes', 'No', 'Two', 'two dogs', 'a dog', 'The Cat', 'an apple and the pear', 'none', 'ten', 'eleven',
    'dont know', "don't know", 'thats it', 'Whats this', 'its', 'ive', 'Im', 'im here', "somebody'd", 'somebodyd',
    "let's go", "she's", 'oclock', '3 oclock', 'yall', "y'all", 'twas night',
    'red, white', 'red,white', 'red , white', '1,000', '1,000 people!', '10,5 and a.b', 'a,b', 'a, b',
    '3.5', '3.', '.5', 'end.', 'U.S.A.', 'u.s.a', 'e.g. this', '...', '. . .', 'v1.2.3.', 'a.1', '1.a',
    'what?', 'what ?', 'what? now', 'wow!', 'wow !', 'a/b', 'a / b', 'a /b', 'left-right', 'left - right', '-5', '- 5',
    '(maybe)', '( maybe )', '[x]', '{x}', '"quoted"', 'a_b', 'a=b', 'a+b', 'a\\b', 'a>b', 'a<b', 'a<=1.', 'me@home', '`tick`',
    'semi;colon', 'semi; colon', "it's", "it's a dog's", 'mr. smith', 'st. 5', 'no. 7', 'no.7',
    '', ' ', '   spaced   out   ', 'TAB\there', 'new\nline', 'MiXeD CaSe An THE a', 'the', 'a an the', 'one a two',
    'zero', 'None', 'NONE of the above', 'one-two', 'one - two', 'ten,ten', 'nine, ten', 'a,', ',a', ',', '1,2,3', '1, 2, 3',
    'hello world.', 'hello. world', 'x' + '.' * 40, ('a. ' * 40).strip(), '.' * 33 + '5', 'café', 'naïve the', '日本 the 語',
    'couldntve', "couldn'tve", "couldnt've", "y'all'dve", "'ow'sat", "ow's'at", "'ows'at", 'wed', "wed've", 'hes a he',
    "I'dve", "Id've", 'Ive', 'theyre there', 'whos who', 'wont', 'cant', 'aint no', 'maam', 'yes sir, maam.',
]


def synth_annotations(seed, n, pool):
    rnd = random.Random(seed)
    qtypes = ['what is', 'how many', 'is the', 'what color']
    out = []
    for i in range(n):
        # a skewed choice so that some answers pass the >= 9 filter and others do not
        main = pool[min(int(rnd.expovariate(0.09)), len(pool) - 1)]
        given = []
        for _ in range(10):
            r = rnd.random()
            given.append(main if r < 0.55 else pool[rnd.randrange(len(pool))])
        # every 7th record repeats the previous question id: the >= 9 filter counts DISTINCT questions
        out.append({'question_id': 1000 * seed + i - (1 if i % 7 == 6 else 0),
                    'question_type': qtypes[rnd.randrange(4)], 'image_id': 7000 + i // 3,
                    'multiple_choice_answer': main,
                    'answers': [{'answer': a} for a in given]})
    return out


POOL = ['yes', 'no', '2', 'two', 'Two', 'a dog', 'dog', 'the dog', 'white', 'red, white', 'red white', 'dont know',
        "don't know", '1,000', '1000', 'frisbee', 'tennis.', 'tennis', 'left', 'right', 'none', '0', 'U.S.', 'us',
        'stop sign', 'stop-sign', 'water', 'grass', 'wood', 'pizza', '3', 'three', 'blue', 'Blue', 'green', 'nothing']


def main():
    keys = sorted(k for k in ref.contractions if k == k.lower())
    answers = ANSWERS + keys + [k + ' ' + k for k in keys[:5]]
    train, val = synth_annotations(1, 300, POOL), synth_annotations(2, 60, POOL)
    with contextlib.redirect_stdout(io.StringIO()), tempfile.TemporaryDirectory() as tmp:
        occ = ref.filter_answers(train, 9)
        ans2label = ref.create_ans2label(occ, 'train', tmp)
        label2ans = [a for a, _ in sorted(ans2label.items(), key=lambda t: t[1])]
        ttrain = ref.compute_target(train, ans2label, 'train', tmp)
        tval = ref.compute_target(val, ans2label, 'val', tmp)

    def plain(targets):   # JSON has no integer dict keys: label_counts as sorted pairs
        return [dict(t, label_counts=sorted(t['label_counts'].items())) for t in targets]

    doc = {
        'normalize': [[a, ref.preprocess_answer(a)] for a in answers],
        'score': [[n, ref.get_score(n)] for n in range(12)],
        'train_annotations': train, 'val_annotations': val, 'min_occurrence': 9,
        'label2ans': label2ans, 'train_targets': plain(ttrain), 'val_targets': plain(tval),
    }
    with open(os.path.join(HERE, 'softscore.json'), 'w') as f:
        json.dump(doc, f, ensure_ascii=True, separators=(',', ':'))
    print('answers', len(answers), 'vocab', len(label2ans), 'bytes', os.path.getsize(os.path.join(HERE, 'softscore.json')))


if __name__ == '__main__':
    main()
