// First conv layer of the image encoder (reference cnn.py:23, Conv2d(3, 16, 3, padding)), bf16 path.
// The generic implicit-GEMM kernel needs the 3 image channels padded to one 16-channel MFMA k-group in an NHWC copy
// (5.3x the image bytes, plus the copy kernel).  Here the whole reduction is k = (c, ky, kx) = 27, fed straight from
// the NCHW image (f16 / f32 / bf16, dataset.py:96):
//   conv3x3_first_fwd    y[B][OH][OW][16] (bf16, NHWC) = conv(images) + bias, with the BatchNorm partial statistics;
//                        LDS patch pixel-interleaved (4 bf16 per pixel), three 16x16x16 MFMAs per 16 output pixels
//   conv3x3_first_wgrad  dW[16][3][3][3] = sum over pixels of dy (x) image patch, persistent workgroups over tiles;
//                        the same pixel-interleaved LDS patch; both operands by transposing LDS reads (the reduction
//                        runs over pixels), columns n = 4 tap + c: three 16x16x32 MFMAs per 32 pixels
//   ..._first_wgrad_bn   the same from (y, g): the BatchNorm-backward apply pass of layer 1 folded in by linearity
// HBM-bound: bytes = |images| + |y|.
#include "common.h"
#include <type_traits>

namespace {

struct FirstArgs {
    const void* img;     // [B][3][IH][IW]
    const float* w;      // [16][27] f32 master weights (torch layout [16][3][3][3])
    bf16* y;             // fwd: output [B][OH][OW][16];  wgrad: dy
    const float* bias;   // [16] or null
    float* stats;        // fwd: [tiles*B][16][2] partial (sum, sumsq) or null;  wgrad: part [G][16][32]
    int B, IH, IW, OH, OW, pad;
    int R, CW, tiles_x, tiles_y;
    float inv_cw, inv_pw, inv_tx, inv_tiles;
    const float* wscale; // fwd: per-output-channel factor folded into the weights (eval-mode BatchNorm) or null
    int relu;            // fwd: y = max(conv + bias, 0)
    int flip;            // read every image mirrored along its width: the reference's whole-batch RandomHorizontalFlip
                         // (vqa_model.py:250) costs nothing here - only the sign of the patch's x step changes
    // fused wgrad: y = the layer's raw conv output, g = gradient w.r.t. the BatchNorm + ReLU output, and the BatchNorm
    // backward coefficients [16] each (bn.hip: dy = scale * (dz - c1 - xhat * c2), dz = z > 0 ? g : 0)
    const bf16* g;
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd, *bn_c1, *bn_c2;
    int rev;             // walk the tiles back to front (unit_walk, common.h)
};

constexpr int PMAX = 384;                       // weight gradient: host bound on (R+2)*(CW+2)

// One image element through the buffer resource: an offset at or beyond the buffer's size returns 0 (the zero padding).
template <typename TI> __device__ __forceinline__ float buffer_ld_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <> __device__ __forceinline__ float buffer_ld_f32<_Float16>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <> __device__ __forceinline__ float buffer_ld_f32<bf16>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0);
    return __uint_as_float((unsigned)h << 16);
}
template <> __device__ __forceinline__ float buffer_ld_f32<float>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// Both kernels are persistent: workgroup g walks the (image, tile) pairs t = g, g + G, ...; what does not depend on the
// tile is computed once per workgroup (a wave64 VALU instruction costs its SIMD 4 cycles, and per-tile index
// arithmetic is what these HBM-bound kernels spend their time on otherwise).

// ---- forward: pixel-interleaved patch.  LDS holds the (R+2) x (CW+2) input pixels as 4 bf16 each (3 channels and a
// zero), 8 bytes per pixel.  The K = 16 values of ONE kernel row ky - k = 4*kx + c with kx, c in 0..3, where kx = 3 and
// c = 3 carry zero weights - are then the 8 bytes of pixel (pr + ky, pc + q) for lane group q: one ds_read_b64 per
// kernel row and three chained 16x16x16 MFMAs per fragment, instead of eight 2-byte gathers with their address adds
// and v_perm packing in front of one 16x16x32 MFMA (the kernel is VALU-bound, not MFMA-bound).  Lane group 3 of the
// last pixels of a patch row reads one pixel past the row (weight 0): the next row's first pixel, or the spare pixel
// PP that every stash zeroes.
constexpr int PMAX_F = 511;                     // forward: host bound on (R+2)*(CW+2) (+ the spare pixel: 512)
constexpr int NPX = (PMAX_F + 1 + 255) / 256;   // patch pixels per thread per tile
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct PixelPlan {           // per thread: the NPX patch pixels it stages (all three channels of each)
    int rel[NPX];            // BYTES of channel 0 from the tile's top-left input pixel
    int pyx[NPX];            // (py-pad) << 16 | (px-pad+64); "never inside" rows for pixels beyond the patch
};

template <typename TI>
__device__ __forceinline__ void plan_pixels(const FirstArgs& a, int PW, int PP, PixelPlan& pl) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int i = tid + j * 256;
        const int py = fastdiv(i, PW, a.inv_pw), px = i - py * PW;
        pl.rel[j] = ((py - a.pad) * a.IW + (a.flip ? a.pad - px : px - a.pad)) * (int)sizeof(TI);
        pl.pyx[j] = ((i < PP ? py - a.pad : -0x4000) << 16) | (px - a.pad + 64);
    }
}

// all 3*NPX loads of tile (b, oy0, ox0) in flight together; out-of-image pixels are requested at the buffer's size
// (the launcher keeps the image tensor under 2 GiB, so size + 2 planes cannot wrap)
template <typename TI>
__device__ __forceinline__ void fetch_pixels(const FirstArgs& a, __amdgpu_buffer_rsrc_t ibuf, unsigned ibytes,
                                             unsigned plane, const PixelPlan& pl, int b, int oy0, int ox0,
                                             float (&v)[NPX][3]) {
    const unsigned org = (unsigned)((b * 3 * a.IH + oy0) * a.IW + (a.flip ? a.IW - 1 - ox0 : ox0)) * (unsigned)sizeof(TI);
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const bool ok = (unsigned)(oy0 + (pl.pyx[j] >> 16)) < (unsigned)a.IH &&
                        (unsigned)(ox0 - 64 + (pl.pyx[j] & 0xFFFF)) < (unsigned)a.IW;
        const unsigned o = ok ? org + (unsigned)pl.rel[j] : ibytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[j][c] = buffer_ld_f32<TI>(ibuf, o + c * plane);
    }
}

__device__ __forceinline__ void stash_pixels(unsigned char* patch, int PP, const float (&v)[NPX][3]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int i = tid + j * 256;
        const bf16x4 t = {(bf16)v[j][0], (bf16)v[j][1], (bf16)v[j][2], (bf16)0.f};
        if (i <= PP) *reinterpret_cast<bf16x4*>(patch + i * 8) = t;     // pixel PP: the zeroed spare
    }
}

template <typename TI, bool RELU, bool STATS>
__global__ __launch_bounds__(256) void conv3x3_first_fwd(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    const unsigned plane = (unsigned)a.IH * a.IW * (unsigned)sizeof(TI);
    const unsigned ibytes = (unsigned)a.B * 3 * plane;                                   // (launcher: < 2 GiB)
    const __amdgpu_buffer_rsrc_t ibuf = buffer_of(a.img, ibytes);
    if ((int)blockIdx.x >= total) return;

    PixelPlan pl;
    plan_pixels<TI>(a, PW, PP, pl);
    // weight fragments (MFMA A operand: row = output channel r, k = 4q + e <-> kx = q, c = e), one per kernel row,
    // straight from the f32 master
    const float wsc = a.wscale ? a.wscale[r] : 1.f;
    s16x4 fw[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        bf16x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool live = q < 3 && e < 3;
            const float wk = a.w[r * 27 + (live ? e * 9 + ky * 3 + q : 0)];   // unconditional load
            t[e] = (bf16)(live ? wk * wsc : 0.f);
        }
        fw[ky] = __builtin_bit_cast(s16x4, t);
    }
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = a.bias ? a.bias[q * 4 + e] : 0.f;
    // fragment pixel decode: fragment f of this wave = tile pixels (wave*4+f)*16 + r
    int paddr[4][3], frc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int m = (wave * 4 + f) * 16 + r;
        int pr = 0, pc = 0;
        if (m < MT) { pr = fastdiv(m, a.CW, a.inv_cw); pc = m - pr * a.CW; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) paddr[f][ky] = ((pr + ky) * PW + pc + q) * 8;
        frc[f] = ((m < MT ? pr : 0x4000) << 16) | pc;
    }
    const unsigned ybytes = (unsigned)a.B * a.OH * a.OW * 32u;                           // (launcher: < 4 GiB)
    const __amdgpu_buffer_rsrc_t ybuf = buffer_of(a.y, ybytes);
    f32x2 s1[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    const f32x4 bias4 = {bv[0], bv[1], bv[2], bv[3]};

    float pv[NPX][3];
    auto tile_org = [&](int t, int& b, int& oy0, int& ox0) {
        b = fastdiv(t, tiles, a.inv_tiles);
        const int tl = t - b * tiles, ty = fastdiv(tl, a.tiles_x, a.inv_tx);
        oy0 = ty * a.R;
        ox0 = (tl - ty * a.tiles_x) * a.CW;
    };
    int fb, foy0, fox0;
    const UnitWalk uw = unit_walk(blockIdx.x, gridDim.x, total, a.rev);
    tile_org(uw.unit(uw.first), fb, foy0, fox0);
    fetch_pixels<TI>(a, ibuf, ibytes, plane, pl, fb, foy0, fox0, pv);
    // Loads and stores share one counter per wave (s_waitcnt vmcnt) and complete out of order with respect to each
    // other, so the wait for a tile's pixels is also a wait for every store issued before it.  The stores of tile t
    // are therefore held in registers and issued after the wait of tile t+1: both the prefetch and the stores get
    // a whole iteration to complete, instead of the store round trip being exposed in every iteration.
    uint4 held[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    unsigned hoff[2] = {ybytes, ybytes};                     // (buffer size = dropped)
    for (int t = uw.first; t < uw.end; t += uw.step) {
        stash_pixels(smem, PP, pv);
        __syncthreads();
        const int b = fb, oy0 = foy0, ox0 = fox0;
        if (t + uw.step < uw.end) {
            tile_org(uw.unit(t + uw.step), fb, foy0, fox0);
            fetch_pixels<TI>(a, ibuf, ibytes, plane, pl, fb, foy0, fox0, pv);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) buffer_store16(ybuf, hoff[h], held[h]);     // (first tile: dropped)
        const unsigned ybase = (unsigned)((b * a.OH + oy0) * a.OW + ox0) * 32u + (q >> 1) * 16;
        // Fragments in pairs: after the MFMA a lane holds 4 channels (8 bytes) of one pixel per fragment.  One
        // v_permlane16_swap per dword between the fragments of a pair leaves every lane with 8 consecutive channels
        // (16 bytes) of ONE pixel - even lane groups the pair's first fragment, odd groups the second - so the tile
        // is written with half as many, twice as wide stores.
#pragma unroll
        for (int fp2 = 0; fp2 < 2; ++fp2) {
            uint2 pk[2];
            unsigned orel[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f = fp2 * 2 + h;
                const int pr = frc[f] >> 16, pc = frc[f] & 0xFFFF;
                const bool ok = oy0 + pr < a.OH && ox0 + pc < a.OW;
                orel[h] = ok ? ybase + (unsigned)(pr * a.OW + pc) * 32u : ybytes;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const s16x4 fp = *reinterpret_cast<const s16x4*>(smem + paddr[f][ky]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fw[ky], fp, acc, 0, 0, 0);
                }
                const float okf = ok ? 1.f : 0.f;
                const Packed4<bf16> o = epilogue4<bf16, RELU, STATS>(acc, bias4, okf, s1, s2);
                pk[h] = make_uint2(o.lo, o.hi);
            }
            auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
            auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
            // lane group q now holds channels (q>>1)*8 .. +7 of the pixel of fragment (q&1) of the pair
            hoff[fp2] = (q & 1) ? orel[1] : orel[0];
            held[fp2] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
        __syncthreads();                                      // fragment reads done before the next stash
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) buffer_store16(ybuf, hoff[h], held[h]);
    if (STATS) {                                              // one row of partial sums per workgroup
        float* red = reinterpret_cast<float*>(smem);          // [4 waves][16][2]
        float t1[4] = {s1[0].x, s1[0].y, s1[1].x, s1[1].y}, t2[4] = {s2[0].x, s2[0].y, s2[1].x, s2[1].y};
        sum16_x4(t1[0], t1[1], t1[2], t1[3]);
        sum16_x4(t2[0], t2[1], t2[2], t2[3]);
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(wave * 16 + q * 4 + e) * 2 + 0] = t1[e];
                red[(wave * 16 + q * 4 + e) * 2 + 1] = t2[e];
            }
        }
        __syncthreads();
        if (tid < 32) a.stats[(long)blockIdx.x * 32 + tid] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
    }
}

// Persistent workgroups over (image, tile); the 4 waves split the pixel reduction of a tile.  D rows = output channel
// (dy fragment by the transposing LDS read), D cols = k index (two 16-column tiles cover k = 0..31), MFMA K = 32
// pixels with the k <-> pixel map of conv.hip's wgrad: lane group q, element 4h+e <-> pixel ks + 16h + 4q + e.
// The pixel -> patch offset of every K element is a per-workgroup LDS table (no division in the loop).
// FUSED: the BatchNorm-backward apply pass of layer 1 happens here - layer 1 has no data gradient, so its dy was written
// (418 MB at B = 256) only to be read back by this kernel - and by LINEARITY: with dz = z > 0 ? g : 0,
//   dy = scale * dz - (A * y + Bc)   =>   dW[c][k] = scale_c * sum(x_k dz_c) - A_c * sum(x_k y_c) - Bc_c * sum(x_k),
// so the tile's dz and its raw y go to LDS side by side and the MFMAs form the three sums (the third against a column of
// ones over the pixels inside the image); first_wgrad_reduce_bn combines them.  What is left for the VALU is the ReLU
// mask - the per-element affine arithmetic in front of the LDS write was 44 of this kernel's 210 us (a build without it
// ran at the 167 us its 913 MB take), and each dy was rounded to bf16 before it met its x.
template <typename TI, bool FUSED>
__global__ __launch_bounds__(256) void conv3x3_first_wgrad(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int MTP = (MT + 31) / 32 * 32;
    unsigned char* patch = smem;                                  // [PP + 1 pixels][4] bf16: 3 channels and a zero
    unsigned char* dyl = smem + ((PP + 1) * 8 + 15) / 16 * 16;    // [MTP pixels][16 co] bf16, 32 bytes per pixel
    short* ptab = reinterpret_cast<short*>(dyl + MTP * 32);       // [MTP] byte offset of tile pixel m inside the patch
    float* coef = reinterpret_cast<float*>(dyl + MTP * 32 + ((MTP * 2 + 15) / 16) * 16);   // FUSED: [2][16] scale, shift
    unsigned char* yl = reinterpret_cast<unsigned char*>(coef + 6 * 16);  // FUSED: the raw y tile, laid out as dyl
    bf16* vl = reinterpret_cast<bf16*>(yl + MTP * 32);                   // FUSED: [MTP] 1 = tile pixel inside the image
    constexpr int NS = FUSED ? 3 : 1;                            // A operands: dz (or dy), y, ones
    constexpr int NA = NS * 3;                                   // x 3 column tiles: taps 0-3, 4-7, 8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4, rq = r >> 2, p = r & 3;
    const unsigned plane = (unsigned)a.IH * a.IW * (unsigned)sizeof(TI);
    const unsigned ibytes = (unsigned)a.B * 3 * plane;                                   // (launcher: < 2 GiB)
    const unsigned dbytes = (unsigned)a.B * a.OH * a.OW * 32u;                           // (launcher: < 4 GiB)
    const __amdgpu_buffer_rsrc_t ibuf = buffer_of(a.img, ibytes), dbuf = buffer_of(a.y, dbytes);
    const __amdgpu_buffer_rsrc_t gbuf = buffer_of(FUSED ? a.g : a.y, dbytes);
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    if ((int)blockIdx.x >= total) return;
    if (FUSED && tid < 16) {
        coef[tid] = a.bn_scale[tid];                              // z = scale * y + shift: the layer's pre-activation
        coef[16 + tid] = a.bn_shift[tid];
    }

    PixelPlan pl;
    plan_pixels<TI>(a, PW, PP, pl);
    // dy vectors of this thread: pixel m = i >> 1, half vv = i & 1
    int drel[2], drc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + j * 256, m = i >> 1, vv = i & 1;
        const int pr = fastdiv(m, a.CW, a.inv_cw), pc = m - pr * a.CW;
        drel[j] = ((pr * a.OW + pc) * 16 + vv * 8) * 2;          // bytes
        drc[j] = ((m < MT ? pr : 0x4000) << 16) | pc;
    }
    if (tid < MTP) {
        const int m = tid < MT ? tid : 0;                         // dy rows beyond the tile are zero: any finite value
        const int pr = fastdiv(m, a.CW, a.inv_cw);
        ptab[tid] = (short)((pr * PW + (m - pr * a.CW)) * 8);
    }
    // The x operand, B[pixel][n]: column tile j holds taps 4j .. 4j+3 with the 4 channel slots of each, n = 4 tap + c.
    // For a transposing read lane (rq, p) hands in the 8 bytes of pixel m0 + rq under tap 4j + p and lane r receives
    // column r of the 4 pixels m0 .. m0+3: (tap 4j + (r >> 2), channel r & 3).  Taps beyond 8 re-read tap 8 and channel
    // slot 3 is the zero: both land in columns nobody stores.
    int toff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tap = 4 * j + p < 8 ? 4 * j + p : 8;
        toff[j] = ((tap / 3) * PW + tap % 3) * 8;
    }
    f32x4 acc[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto tile_org = [&](int t, int& b, int& oy0, int& ox0) {
        b = fastdiv(t, tiles, a.inv_tiles);
        const int tl = t - b * tiles, ty = fastdiv(tl, a.tiles_x, a.inv_tx);
        oy0 = ty * a.R;
        ox0 = (tl - ty * a.tiles_x) * a.CW;
    };
    float pv[NPX][3];
    uint4 dv[2], gv[2];
    unsigned okbits = 0;
    auto fetch = [&](int t) {
        int b, oy0, ox0;
        tile_org(t, b, oy0, ox0);
        fetch_pixels<TI>(a, ibuf, ibytes, plane, pl, b, oy0, ox0, pv);
        const unsigned dorg = (unsigned)((b * a.OH + oy0) * a.OW + ox0) * 32u;
        okbits = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = oy0 + (drc[j] >> 16) < a.OH && ox0 + (drc[j] & 0xFFFF) < a.OW;
            dv[j] = buffer_load16(dbuf, ok ? dorg + (unsigned)drel[j] : dbytes);
            if (FUSED) {
                gv[j] = buffer_load16(gbuf, ok ? dorg + (unsigned)drel[j] : dbytes);
                okbits |= ok ? 1u << j : 0u;
            }
        }
    };
    // FUSED: (y, g) -> dz of 8 channels (thread's half vv = tid & 1): g where the pre-activation is positive.  Outside
    // the image both loads return zeros, so dz = 0 there without a test; the coefficients are read from LDS (the kernel
    // has no registers to spare for 16 of them)
    auto relu_mask = [&](const uint4& yv, const uint4& gq) {
        const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
        const f32x2* c2v = reinterpret_cast<const f32x2*>(coef + (tid & 1) * 8);
        unsigned ow[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x2 y = {__uint_as_float(yw[h] << 16), __uint_as_float(yw[h] & 0xFFFF0000u)};
            const f32x2 z = __builtin_elementwise_fma(y, c2v[h], c2v[8 + h]);
            const unsigned lo = z.x > 0.f ? gw[h] : 0u, hi = z.y > 0.f ? gw[h] : 0u;
            ow[h] = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
        }
        return make_uint4(ow[0], ow[1], ow[2], ow[3]);
    };
    const UnitWalk uw = unit_walk(blockIdx.x, gridDim.x, total, a.rev);
    fetch(uw.unit(uw.first));
    __syncthreads();                                              // (coef)
    typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4;
    for (int t = uw.first; t < uw.end; t += uw.step) {
        stash_pixels(patch, PP, pv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;
            const uint4 tv = FUSED ? relu_mask(dv[j], gv[j]) : dv[j];
            if (i < MTP * 2) {
                *reinterpret_cast<uint4*>(dyl + i * 16) = tv;
                if (FUSED) {
                    *reinterpret_cast<uint4*>(yl + i * 16) = dv[j];
                    if ((tid & 1) == 0) vl[i >> 1] = (bf16)(((okbits >> j) & 1) ? 1.f : 0.f);
                }
            }
        }
        __syncthreads();
        if (t + uw.step < uw.end) fetch(uw.unit(t + uw.step));
        for (int ks = wave * 32; ks < MTP; ks += 128) {
            bf16x8 fa[NS], fb[3];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m0 = ks + h * 16 + q * 4;               // first of the 4 pixels of K elements 4h..4h+3
                bf16x4 va[NS], vx[3];
                va[0] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(dyl + (m0 + rq) * 32 + 4 * p * 2));
                if (FUSED) {
                    va[1 % NS] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(yl + (m0 + rq) * 32 + 4 * p * 2));
                    va[2 % NS] = *reinterpret_cast<const bf16x4*>(vl + m0);      // every row of the ones operand alike
                }
                const unsigned char* px = patch + ptab[m0 + rq];
#pragma unroll
                for (int j = 0; j < 3; ++j) vx[j] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(px + toff[j]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) fa[s][4 * h + e] = va[s][e];
#pragma unroll
                    for (int j = 0; j < 3; ++j) fb[j][4 * h + e] = vx[j][e];
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[s * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s], fb[j], acc[s * 3 + j], 0, 0, 0);
        }
        __syncthreads();                                          // fragment reads done before the next stash
    }
    // combine the 4 waves (fixed order) and write this workgroup's partial slabs [NS][16 co][32 k]
    float* red = reinterpret_cast<float*>(smem);                  // [3 waves][NA][64 lanes][4]
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < NA; ++n) *reinterpret_cast<f32x4*>(red + (((wave - 1) * NA + n) * 64 + lane) * 4) = acc[n];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            for (int w = 0; w < 3; ++w) acc[n] += *reinterpret_cast<const f32x4*>(red + ((w * NA + n) * 64 + lane) * 4);
            // D row = co = 4q+e; D col r of column tile j = n % 3: tap 4j + (r >> 2), channel r & 3 -> k = 9 c + tap
            const int tap = 4 * (n % 3) + rq, c = p;
            if (tap < 9 && c < 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    a.stats[(((long)blockIdx.x * NS + n / 3) * 16 + q * 4 + e) * 32 + c * 9 + tap] = acc[n][e];
            }
        }
    }
}

// dw[co][k] (k < 27: torch layout [16][3][3][3]) = sum_g part[g][co][k]; one wave per output, fixed order
__global__ __launch_bounds__(256) void first_wgrad_reduce(const float* part, float* dw, int G) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= 16 * 27) return;
    const int co = o / 27, k = o - co * 27;
    float s = 0.f;
    for (int g = lane; g < G; g += 64) s += part[((long)g * 16 + co) * 32 + k];
    s = wave_sum(s);
    if (lane == 0) dw[o] = s;
}

// The fused form's three slabs per workgroup - S1 = sum x dz, S2 = sum x y, S3 = sum x (every row alike) - summed over
// the workgroups in fixed order (f64) and combined with the BatchNorm-backward coefficients of the apply pass
// (bn.hip: dy = scale * (dz - c1 - xhat * c2)  =  scale * dz - (A * y + Bc), A = scale * invstd * c2, Bc = scale * c1 -
// A * mean):  dw[co][k] = scale * S1 - A * S2 - Bc * S3.
__global__ __launch_bounds__(1024) void first_wgrad_reduce_bn(const float* part, float* dw, int G, const float* scale,
                                                              const float* mean, const float* invstd, const float* c1,
                                                              const float* c2) {
    // One workgroup per output channel: thread = (k, slab lane gl); a wave's load is 2 whole 128-byte rows of a slab
    // (one wave per output, lanes striding over the slabs, touched a cache line per lane and took 15 us - this kernel
    // is the last of the backward's chain).  Fixed order: slabs gl, gl + 32, ... per thread, then the 32 lanes in turn.
    __shared__ double red[3][32][33];
    const int co = blockIdx.x, k = threadIdx.x & 31, gl = threadIdx.x >> 5;
    double s[3] = {0.0, 0.0, 0.0};
    for (int g0 = gl; g0 < G; g0 += 32 * 8) {
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = g0 + 32 * u;
            const float* p = part + (long)(g < G ? g : 0) * 3 * 512;
            v[u][0] = p[co * 32 + k];
            v[u][1] = p[512 + co * 32 + k];
            v[u][2] = p[1024 + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) s[j] += g0 + 32 * u < G ? (double)v[u][j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) red[j][gl][k] = s[j];
    __syncthreads();
    if (threadIdx.x < 27) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int l = 0; l < 32; ++l)
#pragma unroll
            for (int j = 0; j < 3; ++j) t[j] += red[j][l][k];
        const float sc = scale[co], A = sc * invstd[co] * c2[co], Bc = sc * c1[co] - A * mean[co];
        dw[co * 27 + k] = (float)((double)sc * t[0] - (double)A * t[1] - (double)Bc * t[2]);
    }
}

bool first_args(FirstArgs& a, int B, int IH, int IW, int OH, int OW, int pad, int R, int CW, int pmax) {
    if (OH != IH + 2 * pad - 2 || OW != IW + 2 * pad - 2 || R < 1 || CW < 1 || R * CW > 256 ||
        (R + 2) * (CW + 2) > pmax || B < 1)
        return false;
    // the kernels address images and y through 32-bit buffer offsets (out-of-range offset = zero padding)
    if ((size_t)B * 3 * IH * IW * 4 >= (1ull << 31) || (size_t)B * OH * OW * 32 >= (1ull << 32)) return false;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.pad = pad; a.R = R; a.CW = CW;
    a.tiles_x = (OW + CW - 1) / CW; a.tiles_y = (OH + R - 1) / R;
    a.inv_cw = 1.0f / CW; a.inv_pw = 1.0f / (CW + 2); a.inv_tx = 1.0f / a.tiles_x;
    a.inv_tiles = 1.0f / (a.tiles_x * a.tiles_y);
    // fastdiv (common.h): float quotient + signed 24-bit products - tile numbers and image widths below 2^22 / 2^23
    if ((long)B * a.tiles_x * a.tiles_y >= (1 << 22) || (long)IH * IW >= (1 << 22)) return false;
    return true;
}

}  // namespace

extern "C" int vqa_conv3x3_first_fwd(void* stream, int in_dtype, const void* images, const float* w, void* y,
                                     const float* bias, float* stats, int B, int IH, int IW, int OH, int OW, int pad,
                                     int R, int CW, int G, const float* wscale, int relu, int flip) {
    FirstArgs a{images, w, static_cast<bf16*>(y), bias, stats};
    a.wscale = wscale;
    a.relu = relu;
    a.flip = flip;
    a.rev = vqa_walk_rev(16);
    if (!images || !w || !y || !first_args(a, B, IH, IW, OH, OW, pad, R, CW, PMAX_F)) return VQA_ERR_ARG;
    if (G < 1 || G > a.tiles_x * a.tiles_y * B) return VQA_ERR_ARG;
    size_t lds = (size_t)((R + 2) * (CW + 2) + 1) * 8;       // the patch pixels and the spare one, 8 bytes each
    if (lds < 512) lds = 512;
    dim3 grid(G);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype != VQA_F16 && in_dtype != VQA_F32 && in_dtype != VQA_BF16) return VQA_ERR_ARG;
    auto go = [&](auto relu_tag, auto stats_tag) {
        constexpr bool RL = decltype(relu_tag)::value, ST = decltype(stats_tag)::value;
        if (in_dtype == VQA_F16) hipLaunchKernelGGL((conv3x3_first_fwd<_Float16, RL, ST>), grid, dim3(256), lds, s, a);
        else if (in_dtype == VQA_F32) hipLaunchKernelGGL((conv3x3_first_fwd<float, RL, ST>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((conv3x3_first_fwd<bf16, RL, ST>), grid, dim3(256), lds, s, a);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (relu) { if (stats) go(T_{}, T_{}); else go(T_{}, F_{}); }
    else { if (stats) go(F_{}, T_{}); else go(F_{}, F_{}); }
    CHECK_LAUNCH();
    return VQA_OK;
}

static int first_wgrad_launch(void* stream, int in_dtype, FirstArgs& a, float* part, float* dw, int G, int B, int IH,
                              int IW, int OH, int OW, int pad, int R, int CW, bool fused) {
    // (dw == NULL, fused form only: the slabs are left in `part` for vqa_conv3x3_first_wgrad_bn_reduce)
    if (!a.img || !a.y || !part || (!dw && !fused) || G < 1 || !first_args(a, B, IH, IW, OH, OW, pad, R, CW, PMAX))
        return VQA_ERR_ARG;
    const int MTP = (R * CW + 31) / 32 * 32;
    if (G > a.tiles_x * a.tiles_y * B) return VQA_ERR_ARG;
    size_t lds = (size_t)(((R + 2) * (CW + 2) + 1) * 8 + 15) / 16 * 16 + (size_t)MTP * 32 + (size_t)(MTP * 2 + 15) / 16 * 16 +
                 6 * 16 * 4;
    if (fused) lds += (size_t)MTP * 32 + (size_t)(MTP * 2 + 15) / 16 * 16;      // the y tile and the column of ones
    const size_t red = (size_t)3 * (fused ? 9 : 3) * 64 * 16;                    // the cross-wave combine at the end
    if (lds < red) lds = red;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype != VQA_F16 && in_dtype != VQA_F32 && in_dtype != VQA_BF16) return VQA_ERR_ARG;
#define GO(F)                                                                                                         \
    do {                                                                                                              \
        if (in_dtype == VQA_F16) hipLaunchKernelGGL((conv3x3_first_wgrad<_Float16, F>), dim3(G), dim3(256), lds, s, a); \
        else if (in_dtype == VQA_F32) hipLaunchKernelGGL((conv3x3_first_wgrad<float, F>), dim3(G), dim3(256), lds, s, a); \
        else hipLaunchKernelGGL((conv3x3_first_wgrad<bf16, F>), dim3(G), dim3(256), lds, s, a);                        \
    } while (0)
    if (fused) GO(true); else GO(false);
#undef GO
    if (fused && !dw)
        ;
    else if (fused)
        hipLaunchKernelGGL(first_wgrad_reduce_bn, dim3(16), dim3(1024), 0, s, part, dw, G, a.bn_scale, a.bn_mean,
                           a.bn_invstd, a.bn_c1, a.bn_c2);
    else
        hipLaunchKernelGGL(first_wgrad_reduce, dim3(108), dim3(256), 0, s, part, dw, G);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_conv3x3_first_wgrad(void* stream, int in_dtype, const void* images, const void* dy, float* part,
                                       float* dw, int G, int B, int IH, int IW, int OH, int OW, int pad, int R,
                                       int CW, int flip) {
    FirstArgs a{images, nullptr, static_cast<bf16*>(const_cast<void*>(dy)), nullptr, part};
    a.wscale = nullptr;
    a.relu = 0;
    a.flip = flip;
    a.rev = vqa_walk_rev(32);
    a.g = nullptr;
    return first_wgrad_launch(stream, in_dtype, a, part, dw, G, B, IH, IW, OH, OW, pad, R, CW, false);
}

extern "C" int vqa_conv3x3_first_wgrad_bn(void* stream, int in_dtype, const void* images, const void* y, const void* g,
                                          const float* scale, const float* shift, const float* mean,
                                          const float* invstd, const float* c1, const float* c2, float* part,
                                          float* dw, int G, int B, int IH, int IW, int OH, int OW, int pad, int R,
                                          int CW, int flip) {
    if (!g || !scale || !shift || (dw && (!mean || !invstd || !c1 || !c2))) return VQA_ERR_ARG;
    FirstArgs a{images, nullptr, static_cast<bf16*>(const_cast<void*>(y)), nullptr, part};
    a.wscale = nullptr;
    a.relu = 0;
    a.flip = flip;
    a.rev = vqa_walk_rev(32);
    a.g = static_cast<const bf16*>(g);
    a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = mean; a.bn_invstd = invstd; a.bn_c1 = c1; a.bn_c2 = c2;
    return first_wgrad_launch(stream, in_dtype, a, part, dw, G, B, IH, IW, OH, OW, pad, R, CW, true);
}

extern "C" int vqa_conv3x3_first_wgrad_bn_reduce(void* stream, const float* part, int G, const float* scale,
                                                 const float* mean, const float* invstd, const float* c1,
                                                 const float* c2, float* dw) {
    if (!part || G < 1 || !scale || !mean || !invstd || !c1 || !c2 || !dw) return VQA_ERR_ARG;
    hipLaunchKernelGGL(first_wgrad_reduce_bn, dim3(16), dim3(1024), 0, static_cast<hipStream_t>(stream), part, dw, G,
                       scale, mean, invstd, c1, c2);
    CHECK_LAUNCH();
    return VQA_OK;
}
