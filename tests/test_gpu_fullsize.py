"""GPU: size-independent properties at BASELINE.json's full configs[1] size (bf16, B=256, 224x224, T=14, H=1024,
C=3129), where the CPU oracle is too slow to be the checker."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S, T, H, C, V = 256, 224, 14, 1024, 3129, 10000


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    import bench
    import vqa_model
    from vqa_amd import synth
    tmp = tmp_path_factory.mktemp('full')
    qpath, lpath = bench.make_fixture(str(tmp), V, C)
    torch.manual_seed(42)
    m = vqa_model.VQA({'word_embd_dim': 300, 'question_hidden_dim': H, 'n_layers': 1, 'train_question_path': qpath},
                      lpath, 256, 2, 0.0, 'max', 'relu', precision='bf16').cuda()
    batch = (synth.images(1234, B, S).cuda(), synth.questions(1234, B, T, V).cuda(), synth.targets(1234, B, C).cuda())
    return m, batch


def test_eval_samples_are_independent(setup):
    """Eval-mode BatchNorm makes samples independent: the image encoder on a 64-sample slice is bit-identical to the
    slice of the full batch (same kernels, same tiles per image), and the logits agree to bf16 GEMM-tiling noise."""
    m, (img, q, t) = setup
    m.eval()
    with torch.no_grad():
        full_c, part_c = m.cnn(img), m.cnn(img[64:128])
        assert torch.equal(full_c[64:128], part_c)
        full, part = m(img, q), m(img[64:128], q[64:128])
    assert float((full[64:128] - part).abs().max()) < 2e-2 * float(full.abs().max())


def test_soft_accuracy_is_additive(setup):
    from vqa_amd import ops
    m, (img, q, t) = setup
    m.eval()
    with torch.no_grad():
        lo = m(img, q)
    whole = float(ops.soft_accuracy_sum(lo, t))
    parts = sum(float(ops.soft_accuracy_sum(lo[i:i + 64].contiguous(), t[i:i + 64].contiguous())) for i in range(0, B, 64))
    assert whole == pytest.approx(parts, abs=1e-4)


def test_training_step_is_deterministic_and_learns(setup):
    """Two backward passes from the same state give bit-identical gradients (every reduction is fixed-order; the
    embedding scatter-add by float atomics is the documented exception), all finite; five Adamax steps on one batch
    lower the loss."""
    import copy
    import vqa_model
    from vqa_amd.optim import Trainer
    m, (img, q, t) = setup
    m = copy.deepcopy(m).train()
    crit = vqa_model.SoftScoreBCELoss()
    grads = []
    state = copy.deepcopy(m.state_dict())
    for _ in range(2):
        m.load_state_dict(state)
        m.zero_grad(set_to_none=True)
        crit(m(img, q), t).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    for k in grads[0]:
        assert torch.isfinite(grads[0][k]).all(), k
        if k == 'gru.word_embedding.weight':
            assert torch.allclose(grads[0][k], grads[1][k], rtol=1e-3, atol=1e-5)
        else:
            assert torch.equal(grads[0][k], grads[1][k]), k
    m.load_state_dict(state)
    tr = Trainer(m, lr=2e-3, max_norm=0.25)
    losses = [float(tr.step(img, q, t, flip=False)) for _ in range(5)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
