"""Image encoder: drop-in for the reference's `cnn.CNN` (cnn.py:11-54), HIP-backed.

Same constructor (`padding`, `pooling`), same `convolutions` nn.Sequential layout (so parameter / buffer names
`convolutions.{0,1,3,4,7,8,...}.*` match a reference state_dict), same output `[B, k, 256]`.  The forward and
backward of the whole stack run as ONE torch.autograd.Function whose kernels are hand-written HIP
(vqa_amd/csrc/conv.hip, bn.hip) reached through the C-ABI library: NHWC activations, implicit-GEMM MFMA
convolutions with BatchNorm statistics in the epilogue, fused BN+ReLU+pool apply, closed-form BN backward.

precision: 'bf16' (bf16 activations/weights in the MFMA, fp32 accumulate, fp32 master weights and statistics)
or 'fp32' (exact-f32 MFMA).  There is no CPU path: on a CPU tensor the module raises.
"""
import os
import functools

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from . import plan as P

CHANNELS = [3, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256]      # reference cnn.py:23-40
CONV_IDX = [0, 3, 7, 10, 14, 17, 21, 24, 28, 31]
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
CIN0_PAD = 16          # the 3 image channels are zero-padded to one MFMA k-group


@functools.lru_cache(maxsize=None)
def pick_tile(oh, ow, mt_max=256, patch_max=384):
    """(R, CW): output-tile rows x cols of one workgroup.  Fewest tiles first, then the smallest halo."""
    best = None
    for cw in range(1, min(ow, mt_max) + 1):
        r = min(oh, mt_max // cw)
        while r >= 1 and (r + 2) * (cw + 2) > patch_max:
            r -= 1
        if r < 1:
            continue
        tiles = -(-oh // r) * -(-ow // cw)
        # shrink r as far as the tile count allows (less overhang)
        while r > 1 and -(-oh // (r - 1)) * -(-ow // cw) == tiles:
            r -= 1
        key = (tiles, (r + 2) * (cw + 2))
        if best is None or key < best[0]:
            best = (key, (r, cw))
    return best[1]


def _kp(ck, dtype):
    kc = 32 if dtype == torch.bfloat16 else 16
    return (9 * ck + kc - 1) // kc * kc


def _ck(cin, dtype):
    return min(cin, 32 if dtype == torch.bfloat16 else 16)


def pack_weights(w, cin_p, dtype, dgrad):
    """torch-layout fp32 [Cout,Cin,3,3] -> k-contiguous slab for conv3x3_igemm (forward or data-gradient)."""
    cout, cin = w.shape[0], w.shape[1]
    red = cout if dgrad else cin_p
    rows = cin_p if dgrad else cout
    ck = _ck(red, dtype)
    kp = _kp(ck, dtype)
    out = P.empty(((red // ck) * rows * kp,), dtype, w.device)
    L.call('vqa_conv_pack', L.dt(out), w, out, cout, cin, cin_p, ck, kp, int(dgrad))
    return out


def pack_all(jobs, dtype, scales=None):
    """jobs = [(w, cin_p, dgrad)]: every slab of the step in ONE launch; returns the slabs (views of one buffer).
    scales[i] (optional, forward jobs) = per-output-channel factor folded into slab i (eval-mode BatchNorm)."""
    if not jobs:
        return []
    desc, sizes = [], []
    for w, cin_p, dgrad in jobs:
        cout, cin = w.shape[0], w.shape[1]
        red, rows = (cout, cin_p) if dgrad else (cin_p, cout)
        ck = _ck(red, dtype)
        kp = _kp(ck, dtype)
        desc += [cout, cin, cin_p, ck, kp, int(dgrad)]
        sizes.append(-(-(red // ck) * rows * kp // 64) * 64)          # 128-byte aligned slices
    buf = P.empty((sum(sizes),), dtype, jobs[0][0].device)
    outs, off = [], 0
    for n in sizes:
        outs.append(buf[off:off + n])
        off += n
    n = len(jobs)
    L.call('vqa_conv_pack_all', L.dt(buf), n, [j[0] for j in jobs], outs, desc,
           None if scales is None else list(scales))
    return outs


def _conv_tag(kind, B, OH, OW, cin_real, cout_real, x, y_elems_bytes, w_bytes, kernel=None):
    """Algorithmic work of one conv launch for bench.py's roofline leg: MACs as the reference writes them
    (real channel counts), bytes = input read once + output written once + weights."""
    tag = dict(kind=kind, shape=f'{OH}x{OW} {cin_real}->{cout_real}', flops=2.0 * B * OH * OW * 9 * cin_real * cout_real,
               bytes=float(x.numel() * x.element_size() + y_elems_bytes + w_bytes))
    if kernel:
        tag['kernel'] = kernel
    return tag


def conv_pool_ok(x, pad):
    """vqa_conv3x3_fwd can apply ReLU AND the 2x2 max pool in its epilogue for this input (inference): bf16, an even
    output map, even tiles (16 columns wide in the single-chunk kernel)."""
    B, IH, IW, cin = x.shape
    OH, OW = IH + 2 * pad - 2, IW + 2 * pad - 2
    if not (EVAL_FUSE_POOL and x.dtype == torch.bfloat16 and cin % 16 == 0 and OH % 2 == 0 and OW % 2 == 0):
        return False
    R, CW = pick_tile(OH, OW)
    if cin in (16, 32):                 # single-chunk kernel: a window = two fragments of a lane and its neighbour
        return CW == 16 and R % 2 == 0
    # multi-chunk kernel: any even tile; not the one-tile maps (they share a tile between two images instead)
    return cin % 32 == 0 and R % 2 == 0 and CW % 2 == 0 and not (R >= OH and CW >= OW)


def conv_fwd(x, wpk, bias, cout, pad, want_stats, cin_real=None, kind='fwd', relu=False, in_coeffs=None, bst=None,
             out=None, stats_out=None, pool=False):
    """x NHWC [B,IH,IW,Cin] -> y [B,OH,OW,Cout] (+ partial BN statistics rows).
    pool (with relu, conv_pool_ok): -> [B,OH/2,OW/2,Cout], the 2x2 max pool of relu(y) from the same kernel.
    in_coeffs = (scale, shift): x is the previous conv's RAW output and the kernel applies that layer's BatchNorm +
    ReLU while it stages its input tile (pad must be 0).
    bst = (y_below, coef [4,cout]): a data-gradient call whose output is the gradient w.r.t. the BatchNorm + ReLU output
    of the layer below; the returned rows are that layer's BatchNorm-backward partial sums (for vqa_bn_bwd_finalize)."""
    B, IH, IW, cin = x.shape
    OH, OW = IH + 2 * pad - 2, IW + 2 * pad - 2
    R, CW = pick_tile(OH, OW)
    if pool:
        assert relu and not want_stats and in_coeffs is None and bst is None and out is None and conv_pool_ok(x, pad)
        y = P.empty((B, OH // 2, OW // 2, cout), x.dtype, x.device)
        G = _plan(L.dt(x), B, IH, IW, cin, OH, OW, cout, pad, R, CW, 4)
        tag = lambda: _conv_tag(kind, B, OH, OW, cin if cin_real is None else cin_real, cout, x,
                                y.numel() * y.element_size(), wpk.numel() * wpk.element_size(),
                                kernel='conv3x3_igemm_p' if cin == _ck(cin, x.dtype) else 'conv3x3_igemm_tile')
        L.call('vqa_conv3x3_fwd', L.dt(x), x, wpk, y, bias, None, B, IH, IW, cin, OH, OW, cout, pad, R, CW, G, 3,
               None, None, None, None, tag=tag)
        return y, None
    y = P.empty((B, OH, OW, cout), x.dtype, x.device) if out is None else out
    G = _plan(L.dt(x), B, IH, IW, cin, OH, OW, cout, pad, R, CW, 1 if in_coeffs is not None else (2 if bst else 0))
    stats = None
    if want_stats or bst:       # (out / stats_out: a batch slice of a larger call writes its part of the whole)
        stats = P.empty((G, cout, 2), torch.float32, x.device) if stats_out is None else stats_out
        assert stats.shape[0] == G and stats.is_contiguous()
    assert y.is_contiguous() and tuple(y.shape) == (B, OH, OW, cout)
    cr = cin if cin_real is None else cin_real
    # (algorithmic bytes of the launch: + the lower layer's y when its BatchNorm-backward sums are formed here)
    tag = lambda: _conv_tag(kind, B, OH, OW, cr, cout, x, y.numel() * y.element_size() * (2 if bst else 1),
                            wpk.numel() * wpk.element_size(),
                            kernel='conv3x3_igemm_p' if cin == _ck(cin, x.dtype) else 'conv3x3_igemm_tile')
    isc, ish = in_coeffs if in_coeffs is not None else (None, None)
    by, bcoef = bst if bst else (None, None)
    if bst:
        assert by.shape == y.shape and by.dtype == y.dtype and by.is_contiguous() and tuple(bcoef.shape) == (4, cout)
    L.call('vqa_conv3x3_fwd', L.dt(x), x, wpk, y, bias, stats, B, IH, IW, cin, OH, OW, cout, pad, R, CW, G, int(relu),
           isc, ish, by, bcoef, tag=tag)
    return y, stats


def _same(a, b):
    return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype


def dgrad_wgrad_ok(dy, cin, pad):
    """vqa_conv3x3_dgrad_wgrad has this conv's shape: bf16, pad 0, 16 -> 16 or 32 -> 32 channels, switched on."""
    C = dy.shape[3]
    return dy.dtype == torch.bfloat16 and pad == 0 and C == cin and C in (16, 32) and C <= FUSE_DGRAD_WGRAD


def conv_dgrad_wgrad(dy, wd, cin, bst, cin_real, out=None, stats_out=None, part_out=None):
    """Data gradient (with the lower layer's BatchNorm-backward sums, bst as in conv_fwd) AND weight-gradient slabs of
    a pad-0 conv from one pass over dy and the lower layer's raw output.  Returns (dx, below, part): part [G, C, 9, cin]
    f32 is left to wgrad_reduce_all."""
    B, H, W, C = dy.shape
    IH, IW = H + 2, W + 2
    R, CW = pick_tile(IH, IW)
    G = _plan(L.dt(dy), B, H, W, C, IH, IW, cin, 2, R, CW, 3)
    dx = P.empty((B, IH, IW, cin), dy.dtype, dy.device) if out is None else out
    below = P.empty((G, cin, 2), torch.float32, dy.device) if stats_out is None else stats_out
    part = P.empty((G, C, 9, cin), torch.float32, dy.device) if part_out is None else part_out
    by, bcoef = bst
    assert by.shape == dx.shape and by.dtype == dx.dtype and by.is_contiguous() and tuple(bcoef.shape) == (4, cin)
    assert dx.is_contiguous() and tuple(below.shape) == (G, cin, 2) and below.is_contiguous()
    assert tuple(part.shape) == (G, C, 9, cin) and part.is_contiguous()

    def tag():
        t = _conv_tag('dgrad', B, IH, IW, C, cin, dy, dx.numel() * dx.element_size() * 2, wd.numel() * wd.element_size(),
                      kernel='conv3x3_igemm_p')
        t['flops'] += 2.0 * B * H * W * 9 * cin_real * C      # + the weight gradient
        return t
    L.call('vqa_conv3x3_dgrad_wgrad', L.dt(dy), dy, wd, dx, below, part, B, H, W, C, IH, IW, cin, 2, R, CW, G, by, bcoef,
           tag=tag)
    return dx, below, part


_plans = {}


def _plan(*key):
    """Persistent-grid size of a conv shape (an occupancy query in the library, cached per shape)."""
    if key not in _plans:
        g = L.lib().vqa_conv3x3_plan(*[L.ctypes.c_int(int(v)) for v in key])
        if g < 1:
            raise L.VqaHipError(f'vqa_conv3x3_plan: unsupported conv shape {key}')
        _plans[key] = g
    return _plans[key]


def _wgrad_family(cin, cout):
    """bench.py's roofline leg prices the weight gradient in its two regimes: the conv3x3_wgrad<32, 64> instantiations
    (Cin >= 32 and Cout >= 64: MFMA-bound) apart from the shallow, HBM-bound ones."""
    return 'conv3x3_wgrad_deep' if (cin >= 32 and cout >= 64) else 'conv3x3_wgrad'


def wgrad_rows(x, dy, pad=None):
    """G of conv_wgrad for these shapes (rows of its partial-slab workspace)."""
    return _wgrad_geom(x, dy, pad)[2]


# two images per tile in the weight gradient of a small pad-0 map (vqa_conv3x3_wgrad's seam; the forward pairs inside
# the library): the 10x10 map of the last conv fills 100 of a tile's 256 reduction pixels - 4 steps of 32 for 3.1 steps
# of work, and a patch + dy staging round per image
WGRAD_PAIR = os.environ.get('VQA_WGRAD_PAIR', '1') != '0'


def _wgrad_geom(x, dy, pad=None):
    """-> (R, CW, G, seam, (B, IH, OH) as passed to the library: those of the PAIRS when seam > 0)."""
    B, IH, IW, cin = x.shape
    _, OH, OW, cout = dy.shape
    mt = 256 if x.dtype == torch.bfloat16 else 128
    pmax = 384 if mt == 256 else 320
    R, CW = pick_tile(OH, OW, mt, pmax)
    seam = 0
    if (WGRAD_PAIR and pad == 0 and R >= OH and CW >= OW and B % 2 == 0 and (2 * OH + 2) * CW <= mt and
            2 * IH * (CW + 2) <= pmax):
        seam, B, IH, OH = OH, B // 2, 2 * IH, 2 * OH + 2
        R = OH
    geo = (B, IH, OH)
    tiles = B * (-(-OH // R)) * (-(-OW // CW))
    ck = _ck(cin, x.dtype)
    pairs = (-(-cout // 64)) * (cin // ck)
    esz = x.element_size()
    pxb = 32 if ck * esz == 32 else ck * esz + 32
    cow = min(cout, 64)                                   # output channels per workgroup (conv.hip dy_stride)
    dyb = (32 if cow == 16 else cow * 2 + 32) if esz == 2 else (64 if cow == 16 else cow * 4 + 64)
    kc = 64 // esz
    lds = (R + 2) * (CW + 2) * pxb + (-(-R * CW // kc) * kc) * (dyb + 4) + 8 * ck
    # resident workgroups per CU: register-limited waves/SIMD of the instantiation, then LDS
    waves = 4 if cow == 16 else 2
    if WGRAD_COF4 and cow == 64 and ck == 32 and esz == 2:
        # the four-fragment form (conv.hip, COFX = 4): 360 registers = one wave per SIMD, and its cross-slice combine
        # buffer (73.7 KB) is the floor of its LDS
        waves, lds = 1, max(lds, 2 * 9 * 4 * 1024)
    per_cu = max(1, min(waves, (160 * 1024) // lds))
    if per_cu - WGRAD_RESERVE >= 2:
        per_cu -= WGRAD_RESERVE
    G = max(1, min(tiles, (256 * per_cu) // pairs if pairs <= 256 * per_cu else 1))
    return R, CW, G, seam, geo


def conv_wgrad(x, dy, cin_real, pad, out=None, in_coeffs=None, defer=None, part_out=None):
    """defer (a list): the partial slabs are left unreduced and (part, dw, G, cout, cin, cin_real) is appended to it
    for wgrad_reduce_all - one launch for all layers at the end of the backward.
    part_out ([G, cout, 9, cin] f32): the slabs are written there and NOT reduced (a batch slice of a larger call: the
    caller reduces the slabs of all slices at once, wgrad_reduce_all); returns None."""
    B, IH, IW, cin = x.shape
    _, OH, OW, cout = dy.shape
    R, CW, G, seam, (gB, gIH, gOH) = _wgrad_geom(x, dy, pad)
    isc, ish = in_coeffs if in_coeffs is not None else (None, None)
    if part_out is not None:
        assert tuple(part_out.shape) == (G, cout, 9, cin) and part_out.is_contiguous()
        tag = lambda: _conv_tag('wgrad', B, OH, OW, cin_real, cout, x, dy.numel() * dy.element_size(), cout * cin_real * 36,
                                kernel=_wgrad_family(cin, cout))
        L.call('vqa_conv3x3_wgrad', L.dt(x), x, dy, part_out, None, G, gB, gIH, IW, cin, cin_real,
               gOH, OW, cout, pad, R, CW, isc, ish, seam, tag=tag)
        return None
    part = P.empty((G, cout, 9, cin), torch.float32, x.device)
    dw = out if out is not None else P.empty((cout, cin_real, 3, 3), torch.float32, x.device)
    tag = lambda: _conv_tag('wgrad', B, OH, OW, cin_real, cout, x, dy.numel() * dy.element_size(), dw.numel() * 4,
                            kernel=_wgrad_family(cin, cout))
    L.call('vqa_conv3x3_wgrad', L.dt(x), x, dy, part, dw if defer is None else None, G, gB, gIH, IW, cin, cin_real,
           gOH, OW, cout, pad, R, CW, isc, ish, seam, tag=tag)
    if defer is not None:
        defer.append((part, dw, G, cout, cin, cin_real))
    return dw


def wgrad_reduce_all(jobs):
    """The deferred reductions of conv_wgrad(defer=jobs), one launch."""
    if not jobs:
        return
    desc = []
    for _, _, G, cout, cin, cin_real in jobs:
        desc += [G, cout, cin, cin_real]
    L.call('vqa_conv3x3_wgrad_reduce_all', len(jobs), [j[0] for j in jobs], [j[1] for j in jobs], desc)


def bn_act_pool(y, scale, shift, pool, want_ysel=False, out=None):
    """a = pool(relu(scale * y + shift)).  want_ysel (max pool): returns (a, ysel), ysel = the raw y at the selected
    position of every window (None when the shape is not served - the caller then runs the reduce pass).
    out: the tensor a is written to (same dtype as y)."""
    B, H, W, C = y.shape
    ashape = (B, H // 2 if pool else H, W // 2 if pool else W, C)
    a = P.empty(ashape, y.dtype, y.device) if out is None else out.view(ashape)
    tag = lambda: dict(kind='fwd', shape=f'{H}x{W}x{C}', flops=0.0, bytes=float((y.numel() + a.numel()) * y.element_size()))
    if want_ysel and pool == 1:
        ysel = P.empty(ashape, y.dtype, y.device)
        tag2 = lambda: dict(tag(), bytes=tag()['bytes'] + a.numel() * y.element_size())
        if L.fits('vqa_bn_act_pool_fwd', L.dt(y), y, a, scale, shift, B, H, W, C, pool, ysel, tag=tag2):
            return a, ysel
    L.call('vqa_bn_act_pool_fwd', L.dt(y), y, a, scale, shift, B, H, W, C, pool, None, tag=tag)
    return (a, None) if want_ysel else a


def to_nhwc(images, dtype):
    B, C, H, W = images.shape
    out = P.empty((B, H, W, CIN0_PAD), dtype, images.device)
    L.call('vqa_nchw_to_nhwc', L.dt(images), L.dt(out), images, out, B, C, H, W, CIN0_PAD)
    return out


def first_ok(images, w, dtype):
    """The dedicated first-layer kernels take the NCHW image as it is (bf16 path, Conv2d(3, 16))."""
    return (dtype == torch.bfloat16 and tuple(w.shape) == (16, 3, 3, 3) and images.dim() == 4 and
            images.shape[1] == 3 and images.dtype in (torch.float16, torch.float32, torch.bfloat16))


@functools.lru_cache(maxsize=None)
def pick_first_tile(oh, ow, patch_max=511):
    """Output tile (R, CW) of the first-layer kernels: R*CW <= 256 pixels, (R+2)*(CW+2) <= patch_max patch pixels
    (511 forward, 384 weight gradient).
    The kernel is bound by its stores (32 bytes per output pixel against 6 read), and long contiguous store runs are
    what the memory system rewards: 226x226 at B=256 runs 0.168 ms with 21x12 tiles, 0.147 with 6x38, 0.137-0.139 with
    3x76 / 2x113.  Cost = tiles (every tile costs a full 256-pixel pass) x (1 + 24 / CW)."""
    best = None
    for r in range(1, min(oh, 256) + 1):
        cw_max = min(256 // r, patch_max // (r + 2) - 2, ow)
        if cw_max < 1:
            break
        nx = -(-ow // cw_max)
        cw = -(-ow // nx)
        cost = nx * (-(-oh // r)) * (1.0 + 24.0 / cw)
        if best is None or cost < best[0]:
            best = (cost, r, cw)
    return best[1], best[2]


def first_fwd(images, w, bias, pad, want_stats, wscale=None, relu=False, mirror=False):
    B, _, IH, IW = images.shape
    OH, OW = IH + 2 * pad - 2, IW + 2 * pad - 2
    R, CW = pick_first_tile(OH, OW)
    y = P.empty((B, OH, OW, 16), torch.bfloat16, images.device)
    G = min(B * (-(-OH // R)) * (-(-OW // CW)), 256 * (5 - FIRST_RESERVE))   # persistent workgroups: 5 resident per CU (92 VGPRs)
    stats = P.empty((G, 16, 2), torch.float32, images.device) if want_stats else None
    tag = lambda: dict(kind='first', shape=f'{OH}x{OW} 3->16', flops=2.0 * B * OH * OW * 27 * 16,
               bytes=float(images.numel() * images.element_size() + y.numel() * 2 + w.numel() * 4))
    L.call('vqa_conv3x3_first_fwd', L.dt(images), images, w, y, bias, stats, B, IH, IW, OH, OW, pad, R, CW, G,
           wscale, int(relu), int(mirror), tag=tag)
    return y, stats


def first_wgrad_rows(B, OH, OW):
    """Partial slabs ([3][16][32] f32 each) the fused first-layer weight gradient leaves for B images."""
    R, CW = pick_first_tile(OH, OW, 384)
    return min(B * (-(-OH // R)) * (-(-OW // CW)), 256 * (4 - FIRST_WGRAD_RESERVE))


def first_wgrad(images, dy, pad, out=None, mirror=False, bn=None, part_out=None):
    """bn = (g, scale, shift, mean, invstd, c1, c2): `dy` is then the layer's raw conv output y and the BatchNorm-
    backward apply pass runs inside the kernel (vqa_conv3x3_first_wgrad_bn).
    part_out ([first_wgrad_rows(B, OH, OW), 3, 16, 32] f32, with bn = (g, scale, shift)): a batch slice - the three sums
    are left there for first_wgrad_reduce; returns None."""
    B, _, IH, IW = images.shape
    _, OH, OW, _ = dy.shape
    R, CW = pick_first_tile(OH, OW, 384)                      # (wide tiles: long runs of dy; 0.155 -> 0.139 ms)
    tiles = B * (-(-OH // R)) * (-(-OW // CW))
    # persistent workgroups: what is resident per CU (84 registers -> 5 waves per SIMD; the fused kernel: 124 -> 4)
    # less FIRST_WGRAD_RESERVE
    G = min(tiles, 256 * ((4 if bn is not None else 5) - FIRST_WGRAD_RESERVE))
    if part_out is not None:
        assert tuple(part_out.shape) == (G, 3, 16, 32) and part_out.is_contiguous() and len(bn) == 3
        tag = lambda: dict(kind='first_wgrad', shape=f'{OH}x{OW} 3->16', flops=2.0 * B * OH * OW * 27 * 16,
                           bytes=float(images.numel() * images.element_size() + 2 * dy.numel() * 2))
        L.call('vqa_conv3x3_first_wgrad_bn', L.dt(images), images, dy, *bn, None, None, None, None, part_out, None, G,
               B, IH, IW, OH, OW, pad, R, CW, int(mirror), tag=tag)
        return None
    # (the fused form leaves three slabs per workgroup: sum x dz, sum x y, sum x - conv_first.hip)
    part = P.empty((G, 3 if bn is not None else 1, 16, 32), torch.float32, dy.device)
    dw = out if out is not None else P.empty((16, 3, 3, 3), torch.float32, dy.device)
    tag = lambda: dict(kind='first_wgrad', shape=f'{OH}x{OW} 3->16', flops=2.0 * B * OH * OW * 27 * 16,
               bytes=float(images.numel() * images.element_size() + dy.numel() * 2))
    if bn is not None:
        tag0 = tag
        tag = lambda: dict(tag0(), bytes=tag0()['bytes'] + dy.numel() * 2)
        L.call('vqa_conv3x3_first_wgrad_bn', L.dt(images), images, dy, *bn, part, dw, G, B, IH, IW, OH, OW, pad, R, CW,
               int(mirror), tag=tag)
        return dw
    L.call('vqa_conv3x3_first_wgrad', L.dt(images), images, dy, part, dw, G, B, IH, IW, OH, OW, pad, R, CW,
           int(mirror), tag=tag)
    return dw


def first_wgrad_reduce(part, scale, mean, invstd, c1, c2, out=None):
    """The slabs first_wgrad(part_out=...) left for every batch slice, as one [G, 3, 16, 32] array -> dw [16,3,3,3]."""
    assert part.dim() == 4 and tuple(part.shape[1:]) == (3, 16, 32) and part.is_contiguous()
    dw = out if out is not None else P.empty((16, 3, 3, 3), torch.float32, part.device)
    L.call('vqa_conv3x3_first_wgrad_bn_reduce', part, part.shape[0], scale, mean, invstd, c1, c2, dw)
    return dw


def block_fwd(x, w, bias, gamma, beta, rm, rv, pad, pool, training, first=False, wpk=None, mirror=False,
              in_coeffs=None, apply=True, want_ysel=False, nbt=None, out=None):
    """conv3x3 -> BatchNorm -> ReLU -> (pool).  x NHWC (first=True: the NCHW image, dedicated kernel).  Returns
    (a, saved) with saved = (x, y, scale, shift, mean, invstd) for block_bwd.
    in_coeffs: x is the raw conv output of the previous block, whose BatchNorm + ReLU this block's conv applies itself.
    apply=False (pool == 0 only): the BatchNorm + ReLU pass is left to the consumer; returns a = None.
    nbt: the layer's num_batches_tracked counter (bumped by the statistics kernel); out: where `a` is written."""
    dev, cout = x.device, w.shape[0]
    if first:
        y, stats = first_fwd(x, w, bias, pad, training, mirror=mirror)
    else:
        if wpk is None:
            wpk = pack_weights(w, x.shape[3], x.dtype, dgrad=False)
        y, stats = conv_fwd(x, wpk, bias, cout, pad, training, cin_real=w.shape[1], in_coeffs=in_coeffs)
    scale, shift, mean, invstd = P.empty((4, cout), torch.float32, dev).unbind(0)   # ONE [4,C] block
    if training:
        count = float(y.shape[0] * y.shape[1] * y.shape[2])
        L.call('vqa_bn_finalize', stats, stats.shape[0], cout, count, gamma, beta, rm, rv, BN_MOMENTUM, BN_EPS, mean,
               invstd, scale, shift, nbt)
    else:
        L.call('vqa_bn_eval_coeffs', cout, gamma, beta, rm, rv, BN_EPS, scale, shift)
    if want_ysel:
        a, ysel = bn_act_pool(y, scale, shift, pool, want_ysel=True)
        return a, (x, y, scale, shift, mean, invstd), ysel
    a = bn_act_pool(y, scale, shift, pool, out=out) if apply else None
    return a, (x, y, scale, shift, mean, invstd)


def coef_block(scale, shift, mean, invstd):
    """The [4,C] tensor (scale, shift, mean, invstd) that block_fwd made the four vectors views of."""
    C = scale.numel()
    assert (shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()) == tuple(scale.data_ptr() + 4 * C * k for k in (1, 2, 3))
    return scale.as_strided((4, C), (C, 1))


def block_bwd(d, saved, w, cin_real, pad, pool, need_dx, gdst, first=False, wpk_d=None, mirror=False,
              side=None, in_coeffs=None, part=None, bst=None, defer=None, hold=None, first_below=None,
              first_part=None):
    """d = gradient w.r.t. the block output (NHWC, activation dtype).  gdst = (dw, dgamma, dbeta): where the layer's
    parameter gradients are written (slots of the flat gradient buffer, or of the pass's own gradient block; the conv
    bias feeds a training-mode BatchNorm, so its gradient is identically zero - the reference produces rounding noise
    ~1e-7 of the weight-gradient scale there - and its slot is never written: zero from the buffer's fill).
    Returns (dx, below).
    side: HIP stream for the weight gradient (off the dgrad -> BatchNorm-backward chain), None = current stream.
    part: this block's BatchNorm-backward partial sums, already produced by the data gradient of the block above.
    bst = (y, coef) of the block BELOW: this block's data gradient also produces that block's partial sums (`below`).
    hold (a list, side stream only): the weight gradient is not issued here - a closure that orders the side stream
    behind the main stream AT THE TIME IT IS CALLED and then issues it is appended instead (see WGRAD_HOLD).
    first_below = (images, pad, mirror, sink): the block below is the first layer; when this block runs in batch slices
    the three sums of that layer's weight gradient are formed per slice, right behind the data gradient that wrote the
    slice's d (FIRST_WGRAD_SLICED), and the slabs are appended to `sink`.  first_part: those slabs, for the first layer."""
    x, y, scale, shift, mean, invstd = saved
    dev = y.device
    B, H, W, C = y.shape
    count = float(B * H * W)
    have_part = part is not None
    if not have_part:
        rows = L.lib().vqa_bn_bwd_rows(L.dt(y), B, H, W, C, pool)
        part = P.empty((rows, C, 2), torch.float32, dev)
    rows = part.shape[0]
    dw_out, dgamma, dbeta = gdst
    c1, c2 = P.empty((2, C), torch.float32, dev).unbind(0)
    es = y.element_size()
    if not have_part:
        tag = lambda: dict(kind='reduce', shape=f'{H}x{W}x{C}', flops=0.0, bytes=float((y.numel() + d.numel()) * es))
        L.call('vqa_bn_bwd', L.dt(y), 0, y, d, None, scale, shift, mean, invstd, None, None, part, B, H, W, C, pool,
               tag=tag)
    L.call('vqa_bn_bwd_finalize', part, rows, C, count, dgamma, dbeta, c1, c2)
    cur = torch.cuda.current_stream()
    if first and pool == 0 and not need_dx and FUSE_FIRST_BN:
        # layer 1: nobody but its weight gradient reads dy - the apply pass runs inside that kernel, dy is never stored
        bn = (d, scale, shift, mean, invstd, c1, c2)
        if first_part is not None:
            first_wgrad_reduce(first_part, scale, mean, invstd, c1, c2, out=dw_out)
        elif side is None or FIRST_WGRAD_MAIN:
            first_wgrad(x, y, pad, out=dw_out, mirror=mirror, bn=bn)
        else:
            P.stream_wait(side, cur)
            with torch.cuda.stream(side):
                first_wgrad(x, y, pad, out=dw_out, mirror=mirror, bn=bn)
            for t in (x, y, bn[0], scale, c1, part):
                t.record_stream(side)
        return None, None
    S = _bwd_slices(B, y.numel() * es)
    if S > 1 and not first and need_dx and defer is None:
        # Batch slices: apply pass -> weight gradient -> data gradient per B / S images, so that the slice's dy (and
        # the lower layer's y, read by both gradients) is still in the 256 MB Infinity Cache when it is read again.
        # Same-box A/B, 6 alternations of 60 steps: 4.231 -> 4.209 ms/step with 4 slices of block 1's second conv
        # (411 MB per tensor); 2 slices the same, 8 slices (32 images per launch) +2.5 %, and slicing block 2 as well
        # (205 MB tensors, which fit the cache whole) +0.4 %.
        Bs = B // S
        wd = wpk_d if wpk_d is not None else pack_weights(w, x.shape[3], x.dtype, dgrad=True)
        dx = P.empty(tuple(x.shape), x.dtype, dev)
        dys = P.empty(tuple(y.shape), y.dtype, dev)
        IH, IW, cin = x.shape[1], x.shape[2], x.shape[3]
        both = bst is not None and in_coeffs is not None and _same(bst[0], x) and dgrad_wgrad_ok(y, cin, pad)
        gd = _plan(L.dt(y), Bs, H, W, C, IH, IW, cin, 2 - pad, *pick_tile(IH, IW), 3 if both else (2 if bst else 0))
        below = P.empty((S * gd, cin, 2), torch.float32, dev) if bst else None
        gw = gd if both else wgrad_rows(x[:Bs], dys[:Bs], pad)
        part_w = P.empty((S * gw, C, 9, cin), torch.float32, dev)
        fpart = None
        if first_below is not None and both and FIRST_WGRAD_SLICED and FUSE_FIRST_BN:
            g1 = first_wgrad_rows(Bs, IH, IW)
            fpart = P.empty((S * g1, 3, 16, 32), torch.float32, dev)
            first_below[3].append(fpart)
        for si in range(S):
            sl = slice(si * Bs, (si + 1) * Bs)
            tag = lambda sl=sl: dict(kind='apply', shape=f'{H}x{W}x{C}', flops=0.0,
                                     bytes=float((2 * y[sl].numel() + d[sl].numel()) * es))
            L.call('vqa_bn_bwd', L.dt(y), 1, y[sl], d[sl], dys[sl], scale, shift, mean, invstd, c1, c2, None, Bs, H, W,
                   C, pool, tag=tag)
            if both:        # one pass over (dy, x): the slabs come out of the data gradient's kernel
                conv_dgrad_wgrad(dys[sl], wd, cin, (bst[0][sl], bst[1]), cin_real, out=dx[sl],
                                 stats_out=below[si * gd:(si + 1) * gd], part_out=part_w[si * gw:(si + 1) * gw])
                if fpart is not None:
                    # the first layer's x (x) dz, x (x) y and x sums of this slice, while its d and y are still in the
                    # Infinity Cache (scale, shift: rows 0 and 1 of the coefficient block of the layer below)
                    first_wgrad(first_below[0][sl], bst[0][sl], first_below[1], mirror=first_below[2],
                                bn=(dx[sl], bst[1][0], bst[1][1]), part_out=fpart[si * g1:(si + 1) * g1])
                continue
            if side is not None:
                P.stream_wait(side, cur)
            with torch.cuda.stream(side if side is not None else cur):
                conv_wgrad(x[sl], dys[sl], cin_real, pad, in_coeffs=in_coeffs, part_out=part_w[si * gw:(si + 1) * gw])
            conv_fwd(dys[sl], wd, None, cin, 2 - pad, False, kind='dgrad',
                     bst=(bst[0][sl], bst[1]) if bst else None, out=dx[sl],
                     stats_out=below[si * gd:(si + 1) * gd] if bst else None)
        if side is not None and both:
            P.stream_wait(side, cur)
        with torch.cuda.stream(side if side is not None else cur):
            wgrad_reduce_all([(part_w, dw_out, S * gw, C, cin, cin_real)])
        if side is not None:
            for t in (dys, x, part_w):
                t.record_stream(side)
        return dx, below
    dy = P.empty(tuple(y.shape), y.dtype, dev)
    tag = lambda: dict(kind='apply', shape=f'{H}x{W}x{C}', flops=0.0, bytes=float((2 * y.numel() + d.numel()) * es))
    L.call('vqa_bn_bwd', L.dt(y), 1, y, d, dy, scale, shift, mean, invstd, c1, c2, None, B, H, W, C, pool, tag=tag)

    def wgrad():
        if first:
            first_wgrad(x, dy, pad, out=dw_out, mirror=mirror)
        else:
            conv_wgrad(x, dy, cin_real, pad, out=dw_out, in_coeffs=in_coeffs, defer=defer)

    def dgrad():
        if not need_dx:
            return None, None
        wd = wpk_d if wpk_d is not None else pack_weights(w, x.shape[3], x.dtype, dgrad=True)
        return conv_fwd(dy, wd, None, x.shape[3], 2 - pad, False, kind='dgrad', bst=bst)

    if (need_dx and not first and bst is not None and in_coeffs is not None and _same(bst[0], x)
            and dgrad_wgrad_ok(y, x.shape[3], pad)):
        cin = x.shape[3]
        wd = wpk_d if wpk_d is not None else pack_weights(w, cin, x.dtype, dgrad=True)
        dx, below, part_w = conv_dgrad_wgrad(dy, wd, cin, bst, cin_real)
        job = (part_w, dw_out, part_w.shape[0], C, cin, cin_real)
        if defer is not None:
            defer.append(job)
        elif side is None:
            wgrad_reduce_all([job])
        else:
            P.stream_wait(side, cur)
            with torch.cuda.stream(side):
                wgrad_reduce_all([job])
            part_w.record_stream(side)
        return dx, below
    if side is None:
        wgrad()
        dx, below = dgrad()
    elif hold is not None:
        def later():
            P.stream_wait(side, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrad()
        hold.append(later)
        dy.record_stream(side)
        x.record_stream(side)
        dx, below = dgrad()
    else:
        # (enqueueing the data gradient first instead measured the same)
        P.stream_wait(side, cur)
        with torch.cuda.stream(side):
            wgrad()
        dy.record_stream(side)
        x.record_stream(side)
        dx, below = dgrad()
    return dx, below


_POOL_ID = {}


def _eval_forward(x, first, params, buffers, padding, pool_mode, dtype):
    """Inference (BASELINE.json configs[3], "BN folded"): eval-mode BatchNorm is an affine map per channel, folded into
    the conv - W' = W * scale[co] at packing time, b' = bias * scale + shift - so conv + BN + ReLU is ONE kernel with a
    ReLU epilogue; only the pooled layers keep a second (pool-only) pass."""
    dev = x.device
    cinp = [CIN0_PAD] + list(CHANNELS[1:10])
    coef = torch.empty(10, 3, 256, device=dev, dtype=torch.float32)          # scale, shift, folded bias per layer
    ct = L.ctypes
    ptrs = lambda ts: (ct.c_void_p * 10)(*[t.data_ptr() for t in ts])
    L.check(L.lib().vqa_bn_eval_coeffs_all(L.stream(), 10, ptrs([params[4 * li + 2] for li in range(10)]),
                                           ptrs([params[4 * li + 3] for li in range(10)]),
                                           ptrs([buffers[2 * li] for li in range(10)]),
                                           ptrs([buffers[2 * li + 1] for li in range(10)]),
                                           ptrs([params[4 * li + 1] for li in range(10)]),
                                           (ct.c_int * 10)(*[params[4 * li].shape[0] for li in range(10)]),
                                           ct.c_float(BN_EPS), L.ptr(coef), 256), 'vqa_bn_eval_coeffs_all')
    jobs = [(params[4 * li], cinp[li], False) for li in range(10) if not (first and li == 0)]
    scales = [coef[li, 0] for li in range(10) if not (first and li == 0)]
    slabs = ([None] if first else []) + pack_all(jobs, dtype, scales)
    key = str(dev)
    if key not in _POOL_ID:
        _POOL_ID[key] = (torch.ones(256, device=dev), torch.zeros(256, device=dev))
    one, zero = _POOL_ID[key]
    for li in range(10):
        w = params[4 * li]
        c = w.shape[0]
        pad = padding if li in (0, 2, 4, 6) else 0
        bf = coef[li, 2, :c]
        if first and li == 0:
            y, _ = first_fwd(x, w, bf, pad, False, wscale=coef[0, 0], relu=True)
        elif li % 2 == 1 and pool_mode == 1 and conv_pool_ok(x, pad):
            # (blocks 1 to 4 at 224 x 224: the full-resolution activation - 1.6, 0.8, 0.4 and 0.2 GB at B = 1024 - is
            # never written; same bits as conv -> pool)
            x, _ = conv_fwd(x, slabs[li], bf, c, pad, False, cin_real=w.shape[1], relu=True, pool=True)
            continue
        else:
            y, _ = conv_fwd(x, slabs[li], bf, c, pad, False, cin_real=w.shape[1], relu=True)
        x = bn_act_pool(y, one, zero, pool_mode) if li % 2 == 1 else y       # relu(y) >= 0: pool-only pass
    B, PH, PW, C = x.shape
    if x.dtype == torch.float32:
        return x.reshape(B, PH * PW, C)
    out = torch.empty(B, PH * PW, C, device=dev, dtype=torch.float32)
    L.call('vqa_cast_pad', L.dt(x), L.F32, x, B * PH * PW, C, C, out, C, None)
    out._vqa_lowp = x.reshape(B, PH * PW, C)         # (the head's GEMMs take the bf16 copy as it is)
    return out


# The weight gradient of a layer is off the backward's critical chain (BatchNorm-backward -> dgrad -> next layer's
# BatchNorm-backward): it runs on its own HIP stream beside the dgrad that reads the same dy (same-box A/B:
# 5.26 -> 5.22 ms/step).  VQA_WGRAD_STREAM=0 keeps everything on one stream.
WGRAD_STREAM = os.environ.get('VQA_WGRAD_STREAM', '1') != '0'
# Training forward: the BatchNorm + ReLU of the first conv of each block (no pool behind it) is applied by its two
# consumers - the block's second conv and that conv's weight gradient - while they stage their input tiles; the
# normalised activation is never written (5 passes and 1.6 GB of traffic per step at B = 256).  VQA_FUSE_BN_FWD=0: the
# separate pass (bit-identical results).
FUSE_BN_FWD = os.environ.get('VQA_FUSE_BN_FWD', '1') != '0'
# Backward: the data gradient of a block's second conv produces the gradient d w.r.t. the first conv's BatchNorm + ReLU
# output; with the first conv's raw output y prefetched beside it, its epilogue also accumulates that BatchNorm's
# backward sums (sum dz, sum dz*xhat) - the reduce pass over (y, d) never runs.  VQA_FUSE_BN_BWD=0: the separate pass.
# Only where that data gradient is the single-chunk persistent kernel (16 / 32 channels: HBM-bound, one |y| of
# traffic less); in the multi-chunk kernel the extra reads cost more than the pass (58x58x64: +0.10 ms for 0.046).
FUSE_BN_BWD = int(os.environ.get('VQA_FUSE_BN_BWD', '32'))      # largest channel count it is applied to; 0 = never
FUSE_FIRST_BN = os.environ.get('VQA_FUSE_FIRST_BN', '1') != '0'
# Inference: ReLU and the 2x2 max pool in the epilogue of the pooled blocks' second conv where that conv is the
# single-chunk kernel (vqa_conv3x3_fwd, relu = 3).  VQA_EVAL_FUSE_POOL=0: conv -> pool-only pass (same bits).
EVAL_FUSE_POOL = os.environ.get('VQA_EVAL_FUSE_POOL', '1') != '0'
# The same for a MAX-POOLED block below: its forward pool pass also stores ysel, the raw conv output at the position each
# window selects (pooled resolution), and the data gradient above forms the BatchNorm-backward sums from (d, ysel) - the
# reduce pass over the full-resolution y (4x the bytes) never runs.  Largest channel count of the pooled layer it is
# applied to; 0 = never.
FUSE_POOL_BWD = int(os.environ.get('VQA_FUSE_POOL_BWD', '128'))
# VQA_DEFER_WGRAD_REDUCE=1: every layer's weight gradient leaves its partial slabs unreduced and one launch at the end of
# the backward sums them all (vqa_conv3x3_wgrad_reduce_all) instead of nine little launches.  Off by default: 0.1 ms less
# kernel time per step, but the one launch sits in the step's tail where nothing overlaps it, while the nine run under
# the data-gradient chain (same-box A/B: 4.402 vs 4.378 ms/step).  Bit-identical either way.
DEFER_WGRAD_REDUCE = os.environ.get('VQA_DEFER_WGRAD_REDUCE', '0') != '0'
# backward of the blocks whose conv output is at least VQA_BWD_SLICE_MIN_MB in batch slices (block_bwd):
# VQA_BWD_SLICES = 0 (default) picks the number of slices, 1 turns slicing off, N forces N slices
BWD_SLICES = int(os.environ.get('VQA_BWD_SLICES', '0'))
# both gradients of a block's second conv from one kernel (vqa_conv3x3_dgrad_wgrad) up to this channel count: 0 off,
# 16 block 1 only, 32 blocks 1 and 2.  Same-box A/B, 3 alternations of 60 steps: 16 takes 4.084 -> 4.008 ms/step (the
# kernel: 79.7 us per 64-image slice against 59.6 + 45.5 for the two it replaces); 32 ADDS 0.09 ms - alone its kernel is
# 15 % faster than the two (223 us against 158 + 103), but the 32 -> 32 weight gradient was hidden on the side stream
# and the 65 us it adds to the data gradient are on the critical chain.
FUSE_DGRAD_WGRAD = int(os.environ.get('VQA_FUSE_DGRAD_WGRAD', '16'))
BWD_SLICE_MIN_MB = float(os.environ.get('VQA_BWD_SLICE_MIN_MB', '300'))
# VQA_FIRST_WGRAD_SLICED=1: the first layer's weight gradient per batch slice of block 1's backward
# (vqa_conv3x3_first_wgrad_bn with dw = NULL, then _bn_reduce) - the kernel that runs last, reading y1 and d1 (418 MB
# each) back from HBM, runs right behind the data gradient that wrote the slice's d1 instead (its sums need only
# BatchNorm 1's scale and shift; c1 and c2, which need every slice, enter when the slabs are combined).  Off: measured
# on one box, 150-step runs: 3.953 / 3.961 / 3.957 against 3.960 / 3.955 / 3.961 ms/step with the default two slices
# (a 128-image slice's traffic, 620 MB, is 2.4 x the Infinity Cache: little of d1 and y1 is still there), and SLOWER
# with four (3.999 against 3.975-3.986; eight: 4.12): what the cache returns, the smaller launches take.  (The slices'
# launches on the weight-gradient stream instead of the chain: 4.000 against 3.951-3.956.)
FIRST_WGRAD_SLICED = os.environ.get('VQA_FIRST_WGRAD_SLICED', '0') != '0'
WGRAD_RESERVE = int(os.environ.get('VQA_WGRAD_RESERVE', '0'))
# the deep weight gradient's four-output-fragment form (read by the library from the same variable; measured slower:
# one wave per SIMD - conv.hip)
WGRAD_COF4 = os.environ.get('VQA_WGRAD_COF4', '0') != '0'
# The first layer's weight gradient is the LAST kernel of the backward and depends only on the data gradient above it
# (through BatchNorm 1's c1, c2).  On the weight-gradient stream it queued behind the weight gradient (last slice) and
# the slab sum of the layer above while the main stream had nothing left to do; on the main stream it starts at once
# and those two run beside it: 4.180 -> 4.120 ms/step (5 alternations of 60 steps).  VQA_FIRST_WGRAD_MAIN=0: as before.
FIRST_WGRAD_MAIN = os.environ.get('VQA_FIRST_WGRAD_MAIN', '1') != '0'
# Workgroup slots per CU that the first layer's persistent kernels leave to the other streams' kernels (the question
# encoder's beside the forward kernel, the last weight-gradient slab sums beside the weight gradient).  Alone the
# forward kernel is 3 % slower with four workgroups per CU than with five (147 against 143 us; three: 166); the step
# gains 0.6 %: 3.921 / 3.923 / 3.922 against 3.950 / 3.941 / 3.950 ms (150-step runs alternated on one box; forward
# only 3.940, two slots 3.927, three 3.966).  Round 2 had measured the same knob as neutral - before the trainer's
# stream had priority and the question encoder ran from one call.
FIRST_RESERVE = int(os.environ.get('VQA_FIRST_RESERVE', '1'))
FIRST_WGRAD_RESERVE = int(os.environ.get('VQA_FIRST_WGRAD_RESERVE', str(FIRST_RESERVE)))
# The deep layers' weight gradients (MFMA-bound, like the deep data gradients that run at the same time) are held back
# and released onto the weight-gradient stream when the backward's main chain reaches the HBM-bound shallow blocks:
# layers >= WGRAD_HOLD_FROM are issued after the main-stream launches of layer WGRAD_RELEASE_AT (0 / -1 = no holding).
WGRAD_HOLD_FROM = int(os.environ.get('VQA_WGRAD_HOLD_FROM', '0'))
WGRAD_RELEASE_AT = int(os.environ.get('VQA_WGRAD_RELEASE_AT', '3'))


def _bwd_slices(B, nbytes):
    """Batch slices of a block's backward: tensors larger than the Infinity Cache are cut to ~200 MB per slice, into at
    most 8 launches (more, smaller launches cost more than the cache gives back: 8 slices of the 224x224 first block,
    32 images each, measured +2.5 %), and only into equal slices.  (~100 MB = 4 slices of the 224x224 first block until
    that block's two gradients came from one kernel; since then 2 slices of 209 MB measure 0.5-0.8 % faster:
    3.949 / 3.946 / 3.951 against 3.974 / 3.986 / 3.983 ms/step, 150-step runs alternated on one box.)"""
    if BWD_SLICES == 1 or nbytes < BWD_SLICE_MIN_MB * 1e6:
        return 1
    S = BWD_SLICES
    if S == 0:
        S = 2
        while S < 8 and nbytes / S > 220e6:
            S *= 2
    while S > 1 and B % S:
        S //= 2
    return S
_WGRAD_STREAMS = {}


def _wgrad_stream(dev):
    key = str(dev)
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(dev)
    return _WGRAD_STREAMS[key]


def _knobs():
    """Everything in this module the recorded launch sequence depends on (part of every plan key)."""
    return (WGRAD_STREAM, FUSE_BN_FWD, FUSE_BN_BWD, FUSE_FIRST_BN, FUSE_POOL_BWD, DEFER_WGRAD_REDUCE, BWD_SLICES,
            BWD_SLICE_MIN_MB, WGRAD_RESERVE, FIRST_WGRAD_MAIN, FIRST_RESERVE, FIRST_WGRAD_RESERVE, WGRAD_HOLD_FROM, WGRAD_RELEASE_AT,
            FUSE_DGRAD_WGRAD,
            WGRAD_COF4, WGRAD_PAIR, FIRST_WGRAD_SLICED)


def out_regions(size, padding):
    """Side of the final feature map for a `size` x `size` image (cnn.py:22-42: five blocks of conv(pad) conv(0) pool)."""
    for li in range(10):
        size = size + 2 * (padding if li in (0, 2, 4, 6) else 0) - 2
        if li % 2 == 1:
            size //= 2
    return size


def _grad_layout():
    """Element offsets of the 40 parameter gradients in the pass's own gradient block (no flat buffer to write into):
    every gradient starts on a 256-byte boundary."""
    offs, off = [], 0
    for li in range(10):
        for n in (CHANNELS[li + 1] * CHANNELS[li] * 9, CHANNELS[li + 1], CHANNELS[li + 1], CHANNELS[li + 1]):
            offs.append(off)
            off += -(-n // 64) * 64
    return offs, off


def _train_forward(images, first, params, buffers, padding, pool_mode, dtype, mirror, out):
    """The training forward, launch by launch (recorded into a plan when one is being made).  Writes the [B,k,256] f32
    output into `out`; returns (saved, meta, ysels, wd)."""
    x = images if first else to_nhwc(images, dtype)
    saved, meta, ysels = [], [], [None] * 10
    # every packed weight slab of the step (forward and data-gradient) in one launch
    cinp = [CIN0_PAD] + list(CHANNELS[1:10])
    fjobs = [(params[4 * li], cinp[li], False) for li in range(10) if not (first and li == 0)]
    djobs = [(params[4 * li], cinp[li], True) for li in range(1, 10)]
    slabs = pack_all(fjobs + djobs, dtype)
    wf = ([None] if first else []) + slabs[:len(fjobs)]
    wd = [None] + slabs[len(fjobs):]
    in_coeffs = None
    for li in range(10):
        w, bias, gamma, beta = params[4 * li: 4 * li + 4]
        rm, rv, nbt = buffers[3 * li: 3 * li + 3]
        pad = padding if li in (0, 2, 4, 6) else 0
        pool = pool_mode if li % 2 == 1 else 0
        defer = FUSE_BN_FWD and pool == 0 and li < 9        # the next conv (pad 0) applies this block's BN + ReLU
        # (the pooled layer's channels are the output channels of the data gradient above it)
        ysel_ok = pool == 1 and li < 9 and CHANNELS[li + 1] <= FUSE_POOL_BWD
        # the last block's output IS the encoder's output: f32 [B,k,256] (cnn.py:52-53) - written in place on the f32
        # path, converted by one more launch on the bf16 path
        last_out = out if (li == 9 and dtype == torch.float32) else None
        if ysel_ok:
            a, sv, ysels[li] = block_fwd(x, w, bias, gamma, beta, rm, rv, pad, pool, True, first=first and li == 0,
                                         wpk=wf[li], mirror=mirror and li == 0, in_coeffs=in_coeffs, want_ysel=True,
                                         nbt=nbt)
        else:
            a, sv = block_fwd(x, w, bias, gamma, beta, rm, rv, pad, pool, True, first=first and li == 0, wpk=wf[li],
                              mirror=mirror and li == 0, in_coeffs=in_coeffs, apply=not defer, nbt=nbt, out=last_out)
        saved += list(sv)
        meta.append((pad, pool, in_coeffs is not None))
        x, in_coeffs = (sv[1], (sv[2], sv[3])) if defer else (a, None)
    if dtype != torch.float32:
        B, PH, PW, C = x.shape
        L.call('vqa_cast_pad', L.dt(x), L.F32, x, B * PH * PW, C, C, out, C, None)
    return saved, meta, ysels, wd, (x if dtype != torch.float32 else None)


def _train_backward(dout, saved, ysels, wd, meta, gdsts, dtype, first, mirror):
    """The training backward, launch by launch.  dout f32 [B,k,256] contiguous; gdsts[4 li + j] = where layer li's
    (weight, bias, gamma, beta) gradient goes."""
    B, H, W, C = saved[55].shape
    pool_last = meta[9][1]
    PH, PW = (H // 2, W // 2) if pool_last else (H, W)
    if dtype == torch.float32:
        d = dout.view(B, PH, PW, C)
    else:
        d = P.empty((B, PH, PW, C), dtype, dout.device)
        L.call('vqa_cast_pad', L.F32, L.dt(d), dout, B * PH * PW, C, C, d, C, None)
    side = _wgrad_stream(dout.device) if WGRAD_STREAM else None
    part = None
    deferred = [] if DEFER_WGRAD_REDUCE else None      # weight-gradient slabs, reduced by one launch at the end
    held = []
    fsink = []                                          # the first layer's weight-gradient slabs, formed per slice of block 1
    for li in range(9, -1, -1):
        pad, pool, fused_in = meta[li]
        # the block below has no pool: this block's data gradient can also do its BatchNorm-backward reduction
        bst = None
        if li > 0 and meta[li - 1][1] == 0 and CHANNELS[li] <= min(FUSE_BN_BWD, _ck(CHANNELS[li], dtype)):
            bst = (saved[6 * (li - 1) + 1], coef_block(*saved[6 * (li - 1) + 2: 6 * (li - 1) + 6]))
        elif ysels.get(li - 1) is not None:       # max-pooled block below: the same sums from (d, ysel) at pooled resolution
            bst = (ysels[li - 1], coef_block(*saved[6 * (li - 1) + 2: 6 * (li - 1) + 6]))
        # fused_in: this layer's input x is the previous layer's raw conv output; its weight gradient re-applies
        # that layer's BatchNorm + ReLU (saved scale / shift) as the forward conv did
        in_coeffs = (saved[6 * (li - 1) + 2], saved[6 * (li - 1) + 3]) if fused_in else None
        g = gdsts[4 * li: 4 * li + 4]
        d, part = block_bwd(d, saved[6 * li: 6 * li + 6], None, CHANNELS[li], pad, pool, li > 0, (g[0], g[2], g[3]),
                            first=first and li == 0, wpk_d=wd[li], mirror=mirror and li == 0, side=side,
                            in_coeffs=in_coeffs, part=part, bst=bst, defer=deferred,
                            hold=held if (side is not None and 0 < WGRAD_HOLD_FROM <= li) else None,
                            first_below=(saved[0], meta[0][0], mirror, fsink) if (first and li == 1) else None,
                            first_part=fsink[0] if (li == 0 and fsink) else None)
        if held and li <= WGRAD_RELEASE_AT:
            for fn in held:
                fn()
            held = []
    for fn in held:
        fn()
    if side is not None:
        with torch.cuda.stream(side):
            wgrad_reduce_all(deferred)
        P.stream_wait(torch.cuda.current_stream(), side)
    else:
        wgrad_reduce_all(deferred)


class _CnnFunction(torch.autograd.Function):
    """images [B,3,S,S] + the 40 parameters + 30 buffers (running_mean, running_var, num_batches_tracked per layer)
    -> [B,k,256] fp32.

    Training passes run from PLANS (vqa_amd/plan.py): the first pass of a shape runs launch by launch and is recorded;
    every later one allocates one scratch arena and makes one C-ABI call (vqa_run_plan) for the forward and one for the
    backward - the same launches with the same arguments on the same streams."""

    @staticmethod
    def forward(ctx, images, cfg, *tensors):
        padding, pool_mode, training, dtype, mirror = cfg
        params, buffers = tensors[:40], tensors[40:]
        first = first_ok(images, params[0], dtype)
        if mirror and not (first and training):
            # the mirrored read lives in the first-layer training kernels; everywhere else flip the batch itself
            images, mirror = images.flip(-1), False
        images = images.contiguous()
        dev = images.device
        if not training:
            ctx.cfg = cfg
            return _eval_forward(images if first else to_nhwc(images, dtype), first, params,
                                 [b for i, b in enumerate(buffers) if i % 3 != 2], padding, pool_mode, dtype)
        B, _, IH, IW = images.shape
        ko = out_regions(IH, padding) * out_regions(IW, padding)
        out = torch.empty(B, ko, 256, device=dev, dtype=torch.float32)
        # (the batch-flip flag is a run-time value of the plan, not part of its key: the mirrored and the plain step are
        # the same launches but for that one integer of the two first-layer kernels)
        key = ('cnn.fwd', dev.index, tuple(images.shape), images.dtype, padding, pool_mode, dtype, first, _knobs())
        ctx.cfg, ctx.first, ctx.mirror, ctx.key = cfg, first, mirror, key
        pl = P.lookup(key)
        if pl is not None:
            arena = pl.arena(dev)
            pl.run(arena, [images.data_ptr(), out.data_ptr()] + [t.data_ptr() for t in tensors],
                   patch={'mirror': int(mirror)})
            ctx.plan, ctx.meta, ctx.ysel_at = pl, pl.meta, pl.ysel_at
            ctx.save_for_backward(arena, images, *params)
            if 'lowp' in pl.exports:
                out._vqa_lowp = pl.view(arena, 'lowp').view(B, ko, 256)
            return out
        rec = None
        if P.ENABLED and L._PROFILE is None and P.recording() is None:
            rec = P.Recorder(key, dev)
            rec.bind('images', images)
            rec.bind('out', out)
            for i, t in enumerate(tensors):
                rec.bind(f't{i}', t)
            rec.patchable('mirror', 'vqa_conv3x3_first_fwd', 17)
        if rec is not None:
            with rec:
                saved, meta, ysels, wd, lowp = _train_forward(images, first, params, buffers, padding, pool_mode,
                                                              dtype, mirror, out)
                if lowp is not None:
                    rec.export('lowp', lowp)
                if first:
                    saved[0] = None                              # (layer 1's input is the image: an external base)
                for i, t in enumerate(saved):
                    if t is not None:
                        rec.export(f's{i}', t)
                for li in range(10):
                    if ysels[li] is not None:
                        rec.export(f'ysel{li}', ysels[li])
                    if wd[li] is not None:
                        rec.export(f'wd{li}', wd[li])
            pl = P.lookup(key)
            pl.meta, pl.ysel_at = meta, [li for li in range(10) if ysels[li] is not None]
            ctx.fwd_table = pl.table                             # this pass's own addresses: for its backward's recording
            if first:
                saved[0] = images
        else:
            saved, meta, ysels, wd, lowp = _train_forward(images, first, params, buffers, padding, pool_mode, dtype,
                                                          mirror, out)
            ctx.fwd_table = None
        ctx.plan = None
        ctx.ysel_at = [li for li in range(10) if ysels[li] is not None]
        ctx.wd_at = [li for li in range(10) if wd[li] is not None]
        ctx.save_for_backward(*saved, *params, *[ysels[li] for li in ctx.ysel_at], *[wd[li] for li in ctx.wd_at])
        ctx.meta = meta
        if lowp is not None:
            out._vqa_lowp = lowp.view(B, ko, 256)            # the MFMA-dtype copy the head's GEMMs take (no second cast)
        return out

    @staticmethod
    def backward(ctx, dout):
        padding, pool_mode, training, dtype, _ = ctx.cfg
        if not training:
            raise L.VqaHipError('CNN backward is implemented for training-mode BatchNorm only')
        dev = dout.device
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        if ctx.plan is not None:
            arena, images = ctx.saved_tensors[:2]
            params = ctx.saved_tensors[2:42]
        else:
            params = ctx.saved_tensors[60:100]
            images = ctx.saved_tensors[0] if ctx.first else None
        # where the 40 gradients go: their slots of the flat gradient buffer (optim.FlatParams; adopted by autograd as
        # p.grad without a copy - only valid while p.grad is None), else a gradient block of this pass
        flat = all(p.grad is None and getattr(p, '_vqa_grad_dst', None) is not None for p in params)
        gbase = params[0]._vqa_grad_dst._base if flat else None
        layout = getattr(gbase, '_vqa_layout_id', None) if flat else 0
        if flat and layout is None:
            flat, layout = False, 0
        if flat:
            gdsts = [p._vqa_grad_dst.view(p.shape) for p in params]
        else:
            offs, total = _grad_layout()
            gbase = torch.empty(total, device=dev, dtype=torch.float32)
            gdsts = [gbase[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
        key = ('cnn.bwd', ctx.key, layout)
        side = _wgrad_stream(dev) if WGRAD_STREAM else None
        pl = P.lookup(key) if ctx.plan is not None else None
        if pl is not None:
            arena_b = pl.arena(dev)
            # (bases in the order the recording bound them: fwd arena, [the image: first-layer kernels only], dout, grads)
            pl.run(arena_b, [arena.data_ptr()] + ([images.data_ptr()] if ctx.first else []) +
                   [dout.data_ptr(), gbase.data_ptr()], [side.cuda_stream] if side is not None else [],
                   patch={'mirror': int(ctx.mirror)})
            return (None, None, *gdsts, *([None] * 30))
        # ---- launch by launch (and recorded, when the forward pass has a plan to record against)
        if ctx.plan is not None:
            fp = ctx.plan
            saved = [fp.view(arena, f's{i}') if f's{i}' in fp.exports else None for i in range(60)]
            if ctx.first:
                saved[0] = images
            ysels = {li: fp.view(arena, f'ysel{li}') for li in ctx.ysel_at}
            wd = [fp.view(arena, f'wd{li}') if f'wd{li}' in fp.exports else None for li in range(10)]
            table = [(arena.data_ptr(), fp.arena_bytes, 0)]
        else:
            saved = list(ctx.saved_tensors[:60])
            rest = ctx.saved_tensors[100:]
            ysels = dict(zip(ctx.ysel_at, rest[:len(ctx.ysel_at)]))
            wd = [None] * 10
            for li, t in zip(ctx.wd_at, rest[len(ctx.ysel_at):]):
                wd[li] = t
            table = ctx.fwd_table
        rec = None
        if table is not None and P.ENABLED and L._PROFILE is None and P.recording() is None:
            rec = P.Recorder(key, dev)
            rec.bind_table('fwd', table)
            if ctx.first:
                rec.bind('images', images)
            rec.bind('dout', dout)
            rec.bind('grads', gbase)
            if side is not None:
                rec.stream('wgrad', side)
            rec.patchable('mirror', 'vqa_conv3x3_first_wgrad', 14)
            rec.patchable('mirror', 'vqa_conv3x3_first_wgrad_bn', 21)

        def run():
            if not flat:
                L.call('vqa_zero', gbase, gbase.numel() * 4)      # (the conv-bias slots are never written)
            _train_backward(dout, saved, ysels, wd, ctx.meta, gdsts, dtype, ctx.first, ctx.mirror)
        if rec is not None:
            with rec:
                run()
        else:
            run()
        return (None, None, *gdsts, *([None] * 30))


class CNN(nn.Module):
    def __init__(self, padding=0, pooling='max', precision='bf16'):
        super(CNN, self).__init__()
        self.device = 'cuda:0' if torch.cuda.is_available() else 'cpu'
        self.padding, self.pooling, self.precision = padding, pooling, precision

        def pool():
            return nn.MaxPool2d(2, 2) if pooling == 'max' else nn.AvgPool2d(2, 2)

        layers = []
        for li in range(10):
            pad = padding if li in (0, 2, 4, 6) else 0
            layers += [nn.Conv2d(CHANNELS[li], CHANNELS[li + 1], kernel_size=3, padding=pad),
                       nn.BatchNorm2d(CHANNELS[li + 1]), nn.ReLU()]
            if li % 2 == 1:
                layers.append(pool())
        # The Sequential only HOLDS the parameters / buffers under the reference's names; it is never called.
        self.convolutions = nn.Sequential(*layers)

    def forward(self, x, mirror=False):
        """mirror=True encodes the batch flipped along its width (the reference flips the batch tensor before the
        call, vqa_model.py:250; here the first layer reads it mirrored instead)."""
        if not x.is_cuda:
            raise L.VqaHipError('vqa_amd.CNN runs on an MI355X only (no CPU fallback); move the model and inputs '
                                'to cuda')
        dtype = torch.bfloat16 if self.precision == 'bf16' else torch.float32
        params, buffers = [], []
        for ci in CONV_IDX:
            conv, bn = self.convolutions[ci], self.convolutions[ci + 1]
            params += [conv.weight, conv.bias, bn.weight, bn.bias]
            buffers += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        cfg = (self.padding, 1 if self.pooling == 'max' else 2, self.training, dtype, bool(mirror))
        # (num_batches_tracked is bumped by each layer's statistics kernel, vqa_bn_finalize)
        return _CnnFunction.apply(x, cfg, *params, *buffers)
