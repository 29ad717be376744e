"""GPU: question encoder (embedding + GRU) against the oracle; persistent one-launch recurrence vs per-step path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(H, V, tmp_path, prec, layers=1):
    from tests.helpers import fixture_files
    from vqa_amd import synth
    import gru
    qpath, _ = fixture_files(tmp_path, 5)
    g = gru.GRU(300, H, layers, qpath, precision=prec)
    sd = {k[4:]: v for k, v in synth.state_dict(3, H, 5, 82, n_layers=layers).items() if k.startswith('gru.')}
    g.load_state_dict(sd)
    return g.cuda(), {'gru.' + k: v for k, v in sd.items()}


@pytest.mark.parametrize('prec,H,B,T,tol,persistent,layers',
                         [('fp32', 64, 5, 7, 2e-5, False, 1), ('fp32', 256, 3, 14, 2e-5, False, 1),
                          ('bf16', 256, 70, 14, 3e-2, False, 1), ('bf16', 512, 130, 9, 3e-2, False, 1),
                          ('bf16', 256, 70, 14, 3e-2, True, 1), ('bf16', 512, 130, 9, 3e-2, True, 1),
                          # the benchmark's shape (configs[1]) on both launch structures, and configs[4]'s large hidden
                          # size (the persistent kernel's W_hh slice no longer fits LDS there: it must fall back)
                          ('bf16', 1024, 256, 14, 3e-2, False, 1), ('bf16', 1024, 256, 14, 3e-2, True, 1),
                          ('bf16', 2048, 64, 30, 3e-2, True, 1),
                          # stacked nn.GRU (gru.py:51 num_layers > 1): every layer's BPTT takes the gradient of the
                          # layer above at every step (vqa_gru_bwd_steps dh_seq)
                          ('fp32', 64, 5, 7, 2e-5, False, 2), ('fp32', 256, 3, 14, 3e-5, False, 3),
                          ('bf16', 256, 70, 14, 4e-2, False, 2), ('bf16', 1024, 64, 14, 4e-2, True, 2)])
def test_gru_vs_oracle(prec, H, B, T, tol, persistent, layers, tmp_path, monkeypatch):
    from oracle import vqa_oracle as O
    from vqa_amd import synth
    import vqa_amd.gru as G
    monkeypatch.setattr(G, 'USE_PERSISTENT', persistent)
    g, sd = _mk(H, 82, tmp_path, prec, layers)
    q = synth.questions(5, B, T, 82)
    dh = torch.randn(B, H)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    h_ref = O.gru_forward(p, q)
    grads = torch.autograd.grad((h_ref * dh).sum(), list(p.values()))
    h = g(q.cuda())
    (h * dh.cuda()).sum().backward()
    rel = lambda u, v: float((u.detach().cpu().double() - v.detach().double()).norm() / v.detach().double().norm())
    assert rel(h, h_ref) < tol
    got = dict(g.named_parameters())
    for (k, _), gr in zip(p.items(), grads):
        assert rel(got[k[4:]].grad, gr) < tol * 3, k


def test_persistent_matches_per_step(tmp_path, monkeypatch):
    """Same arithmetic, two launch structures: the one-launch recurrence must reproduce the per-step path."""
    import vqa_amd.gru as G
    from vqa_amd import synth
    H, B, T = 512, 200, 14
    g, _ = _mk(H, 82, tmp_path, 'bf16')
    q = synth.questions(6, B, T, 82).cuda()
    dh = torch.randn(B, H, device='cuda')
    outs = []
    for use in (True, False):
        monkeypatch.setattr(G, 'USE_PERSISTENT', use)
        g.zero_grad()
        h = g(q)
        (h * dh).sum().backward()
        outs.append((h.detach().clone(), {k: p.grad.clone() for k, p in g.named_parameters()}))
    # identical up to bf16 rounding flips of h_t (the two paths add b_hh in a different order: 1e-7 differences
    # occasionally land on the other side of a bf16 rounding boundary and propagate through the recurrence)
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-3
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[1][1][k]
        assert float((a - b).norm() / b.norm()) < 2e-2, k


@pytest.mark.parametrize('B,T,E,V,hot', [(256, 14, 300, 10000, 0.4), (64, 30, 300, 82, 0.5), (700, 14, 300, 500, 0.3),
                                         (5, 3, 6, 7, 0.0), (1, 1, 300, 82, 0.0), (33, 9, 10, 40, 0.9),
                                         # V >= 2^19 with more than 4096 tokens in a chunk: 64-bit sort keys, 64 KB of LDS keys
                                         (400, 14, 10, 524293, 0.2)])
def test_embedding_backward_is_exact_and_deterministic(B, T, E, V, hot):
    """vqa_embed_bwd (sorted, no atomics) against the scatter-add in fp64: hot rows (the <pad> row takes `hot` of the
    tokens), out-of-vocabulary ids (dropped), more tokens than one sort chunk (700 x 14 > 8192), widths that are not a
    multiple of 4 - and bit-identical on a second run."""
    from vqa_amd import _lib as L
    torch.manual_seed(B + T + E)
    q = torch.randint(0, V, (B, T))
    q[torch.rand(B, T) < hot] = 1
    if B * T > 20:
        q[0, 0], q[-1, -1] = V + 3, -2                       # nn.Embedding would raise: these contribute nothing
    ldx = (E + 7) // 8 * 8
    dx = torch.randn(T * B, ldx)
    want = torch.zeros(V, E, dtype=torch.float64)
    for b in range(B):
        for t in range(T):
            if 0 <= int(q[b, t]) < V:
                want[int(q[b, t])] += dx[t * B + b, :E].double()
    outs = []
    for _ in range(2):
        demb = torch.zeros(V, E, device='cuda')
        ws = torch.empty(L.lib().vqa_embed_bwd_ws_bytes(B, T, E), device='cuda', dtype=torch.uint8)
        L.call('vqa_embed_bwd', q.cuda(), dx.cuda(), B, T, E, V, ldx, demb, ws)
        outs.append(demb.cpu())
    assert torch.equal(outs[0], outs[1])
    scale = float(want.abs().max()) + 1e-30
    assert float((outs[0].double() - want).abs().max()) < 2e-6 * scale * max(1.0, (hot * B * T) ** 0.5)
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[q[(q >= 0) & (q < V)].unique()] = False
    assert float(outs[0][untouched].abs().max() if untouched.any() else 0.0) == 0.0
