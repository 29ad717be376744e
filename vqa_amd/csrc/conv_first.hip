// First conv layer of the image encoder (reference cnn.py:23, Conv2d(3, 16, 3, padding)), bf16 path.
// The generic implicit-GEMM kernel needs the 3 image channels padded to one 16-channel MFMA k-group in an NHWC copy
// (5.3x the image bytes, plus the copy kernel).  Here the whole reduction is k = (c, ky, kx) = 27, fed straight from
// the NCHW image (f16 / f32 / bf16, dataset.py:96):
//   conv3x3_first_fwd    y[B][OH][OW][16] (bf16, NHWC) = conv(images) + bias, with the BatchNorm partial statistics;
//                        LDS patch pixel-interleaved (4 bf16 per pixel), three 16x16x16 MFMAs per 16 output pixels
//   conv3x3_first_wgrad  dW[16][3][3][3] = sum over pixels of dy (x) image patch, persistent workgroups over tiles;
//                        LDS patch as 3 planes of (R+2) x (CW+2) bf16, fragment elements gathered with 2-byte LDS
//                        reads (k-major gather offsets are per-lane constants), one 16x16x32 MFMA per 32 pixels
//   ..._first_wgrad_bn   the same from (y, g): the BatchNorm-backward apply pass of layer 1 folded into the dy staging
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

constexpr int PMAX = 384;                       // host bound on (R+2)*(CW+2)
constexpr int NIV = (3 * PMAX + 255) / 256;     // image elements per thread per tile

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
struct PatchPlan {           // per thread: the NIV patch elements (c, py, px) it stages
    int rel[NIV];            // BYTES from the tile's top-left input pixel: (c*IH*IW + (py-pad)*IW + (px-pad)) * sizeof(TI)
    int pyx[NIV];            // (py-pad) << 16 | (px-pad+64); "never inside" rows for elements beyond the patch
};

template <typename TI>
__device__ __forceinline__ void plan_patch(const FirstArgs& a, int PW, int PP, PatchPlan& pl) {
    const int tid = threadIdx.x;
    const float inv_pp = 1.0f / PP;
#pragma unroll
    for (int j = 0; j < NIV; ++j) {
        const int i = tid + j * 256;
        const int c = fastdiv(i, PP, inv_pp), rem = i - c * PP;
        const int py = fastdiv(rem, PW, a.inv_pw), px = rem - py * PW;
        pl.rel[j] = (c * a.IH * a.IW + (py - a.pad) * a.IW + (a.flip ? a.pad - px : px - a.pad)) * (int)sizeof(TI);
        pl.pyx[j] = ((c < 3 ? py - a.pad : -0x4000) << 16) | (px - a.pad + 64);
    }
}

// Branch-free fetch of tile (b, oy0, ox0): all NIV loads in flight together; out-of-image elements are requested at the
// buffer's size, which the hardware answers with 0.
template <typename TI>
__device__ __forceinline__ void fetch_patch(const FirstArgs& a, __amdgpu_buffer_rsrc_t ibuf, unsigned ibytes,
                                            const PatchPlan& pl, int b, int oy0, int ox0, float (&v)[NIV]) {
    // (mirrored column: in range iff the column is)
    const unsigned org = (unsigned)((b * 3 * a.IH + oy0) * a.IW + (a.flip ? a.IW - 1 - ox0 : ox0)) * (unsigned)sizeof(TI);
#pragma unroll
    for (int j = 0; j < NIV; ++j) {
        const bool ok = (unsigned)(oy0 + (pl.pyx[j] >> 16)) < (unsigned)a.IH &&
                        (unsigned)(ox0 - 64 + (pl.pyx[j] & 0xFFFF)) < (unsigned)a.IW;
        v[j] = buffer_ld_f32<TI>(ibuf, ok ? org + (unsigned)pl.rel[j] : ibytes);
    }
}

__device__ __forceinline__ void stash_patch(unsigned char* patch, int n, const float (&v)[NIV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NIV; ++j) {
        const int i = tid + j * 256;
        const bf16 t = (bf16)v[j];
        if (i < n) reinterpret_cast<bf16*>(patch)[i] = t;
    }
}

// k = 8q + j  ->  element offset of (c, ky, kx) inside the patch; padded k (>= 27) reads element 0 (its weight is 0)
__device__ __forceinline__ int gather_off(int k, int PW, int PP) {
    if (k >= 27) return 0;
    const int c = k / 9, t = k - c * 9;
    return c * PP + (t / 3) * PW + (t % 3);
}

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
// FUSED: the BatchNorm-backward apply pass of layer 1 happens here, at the LDS write of the dy tile - layer 1 has no data
// gradient, so its dy was written (418 MB at B = 256) only to be read back by this kernel.
template <typename TI, bool FUSED>
__global__ __launch_bounds__(256) void conv3x3_first_wgrad(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int MTP = (MT + 31) / 32 * 32;
    unsigned char* patch = smem;
    unsigned char* dyl = smem + (3 * PP * 2 + 15) / 16 * 16;      // [MTP pixels][16 co] bf16, 32 bytes per pixel
    short* ptab = reinterpret_cast<short*>(dyl + MTP * 32);       // [MTP] byte offset of pixel m inside a patch plane
    float* coef = reinterpret_cast<float*>(dyl + MTP * 32 + ((MTP * 2 + 15) / 16) * 16);   // FUSED: [4][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4, rq = r >> 2, p = r & 3;
    const unsigned ibytes = (unsigned)a.B * 3 * a.IH * a.IW * (unsigned)sizeof(TI);      // (launcher: both < 4 GiB)
    const unsigned dbytes = (unsigned)a.B * a.OH * a.OW * 32u;
    const __amdgpu_buffer_rsrc_t ibuf = buffer_of(a.img, ibytes), dbuf = buffer_of(a.y, dbytes);
    const __amdgpu_buffer_rsrc_t gbuf = buffer_of(FUSED ? a.g : a.y, dbytes);
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    if ((int)blockIdx.x >= total) return;
    if (FUSED && tid < 16) {
        // dy = scale * (dz - c1 - (y - mean) * invstd * c2)  =  scale * dz - (A * y + Bc),  dz = z > 0 ? g : 0
        const float sc = a.bn_scale[tid], A = sc * a.bn_invstd[tid] * a.bn_c2[tid];
        coef[tid] = sc;
        coef[16 + tid] = a.bn_shift[tid];
        coef[32 + tid] = A;
        coef[48 + tid] = sc * a.bn_c1[tid] - A * a.bn_mean[tid];
    }

    PatchPlan pl;
    plan_patch<TI>(a, PW, PP, pl);
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
        ptab[tid] = (short)((pr * PW + (m - pr * a.CW)) * 2);
    }
    const int koff0 = gather_off(r, PW, PP) * 2, koff1 = gather_off(16 + r, PW, PP) * 2;   // k = r and k = 16 + r
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    auto tile_org = [&](int t, int& b, int& oy0, int& ox0) {
        b = fastdiv(t, tiles, a.inv_tiles);
        const int tl = t - b * tiles, ty = fastdiv(tl, a.tiles_x, a.inv_tx);
        oy0 = ty * a.R;
        ox0 = (tl - ty * a.tiles_x) * a.CW;
    };
    float pv[NIV];
    uint4 dv[2], gv[2];
    unsigned okbits = 0;
    auto fetch = [&](int t) {
        int b, oy0, ox0;
        tile_org(t, b, oy0, ox0);
        fetch_patch<TI>(a, ibuf, ibytes, pl, b, oy0, ox0, pv);
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
    // FUSED: (y, g) -> dy of 8 channels (thread's half vv = tid & 1) in packed f32x2 arithmetic; the coefficients are
    // read from LDS (the kernel has no registers to spare for 32 of them)
    auto bn_apply = [&](const uint4& yv, const uint4& gq, bool ok) {
        const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
        const f32x2* c2v = reinterpret_cast<const f32x2*>(coef + (tid & 1) * 8);
        unsigned ow[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x2 y = {__uint_as_float(yw[h] << 16), __uint_as_float(yw[h] & 0xFFFF0000u)};
            const f32x2 g = {__uint_as_float(gw[h] << 16), __uint_as_float(gw[h] & 0xFFFF0000u)};
            const f32x2 sc = c2v[h], sh = c2v[8 + h], A = c2v[16 + h], Bc = c2v[24 + h];
            const f32x2 z = __builtin_elementwise_fma(y, sc, sh);
            const f32x2 t = sc * g;
            const f32x2 lin = __builtin_elementwise_fma(A, y, Bc);
            f32x2 o = {(z.x > 0.f ? t.x : 0.f), (z.y > 0.f ? t.y : 0.f)};
            o = o - lin;
            typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t ob = {(bf16)o.x, (bf16)o.y};
            ow[h] = ok ? __builtin_bit_cast(unsigned, ob) : 0u;
        }
        return make_uint4(ow[0], ow[1], ow[2], ow[3]);
    };
    const UnitWalk uw = unit_walk(blockIdx.x, gridDim.x, total, a.rev);
    fetch(uw.unit(uw.first));
    __syncthreads();                                              // (coef)
    for (int t = uw.first; t < uw.end; t += uw.step) {
        stash_patch(patch, 3 * PP, pv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;
            const uint4 tv = FUSED ? bn_apply(dv[j], gv[j], (okbits >> j) & 1) : dv[j];
            if (i < MTP * 2) *reinterpret_cast<uint4*>(dyl + i * 16) = tv;
        }
        __syncthreads();
        if (t + uw.step < uw.end) fetch(uw.unit(t + uw.step));
        for (int ks = wave * 32; ks < MTP; ks += 128) {
            bf16x8 fa, fb0, fb1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m0 = ks + h * 16 + q * 4;               // first of the 4 pixels of K elements 4h..4h+3
                bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(dyl + (m0 + rq) * 32 + 4 * p * 2));
                const short4 po = *reinterpret_cast<const short4*>(ptab + m0);
                const int pof[4] = {po.x, po.y, po.z, po.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    fa[4 * h + e] = v[e];
                    fb0[4 * h + e] = *reinterpret_cast<const bf16*>(patch + pof[e] + koff0);
                    fb1[4 * h + e] = *reinterpret_cast<const bf16*>(patch + pof[e] + koff1);
                }
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, acc[1], 0, 0, 0);
        }
        __syncthreads();                                          // fragment reads done before the next stash
    }
    // combine the 4 waves (fixed order) and write this workgroup's partial slab [16 co][32 k]
    float* red = reinterpret_cast<float*>(smem);                  // [3 waves][2][64 lanes][4]
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(red + (((wave - 1) * 2 + n) * 64 + lane) * 4) = acc[n];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            for (int w = 0; w < 3; ++w) acc[n] += *reinterpret_cast<const f32x4*>(red + ((w * 2 + n) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)                            // D row = co = 4q+e, D col = k = 16n + r
                a.stats[((long)blockIdx.x * 16 + q * 4 + e) * 32 + n * 16 + r] = acc[n][e];
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
    if (!a.img || !a.y || !part || !dw || G < 1 || !first_args(a, B, IH, IW, OH, OW, pad, R, CW, PMAX)) return VQA_ERR_ARG;
    const int MTP = (R * CW + 31) / 32 * 32;
    if (G > a.tiles_x * a.tiles_y * B) return VQA_ERR_ARG;
    size_t lds = (size_t)(3 * (R + 2) * (CW + 2) * 2 + 15) / 16 * 16 + (size_t)MTP * 32 + (size_t)(MTP * 2 + 15) / 16 * 16 +
                 6 * 16 * 4;
    if (lds < 3 * 2 * 64 * 16) lds = 3 * 2 * 64 * 16;
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
    if (!g || !scale || !shift || !mean || !invstd || !c1 || !c2) return VQA_ERR_ARG;
    FirstArgs a{images, nullptr, static_cast<bf16*>(const_cast<void*>(y)), nullptr, part};
    a.wscale = nullptr;
    a.relu = 0;
    a.flip = flip;
    a.rev = vqa_walk_rev(32);
    a.g = static_cast<const bf16*>(g);
    a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = mean; a.bn_invstd = invstd; a.bn_c1 = c1; a.bn_c2 = c2;
    return first_wgrad_launch(stream, in_dtype, a, part, dw, G, B, IH, IW, OH, OW, pad, R, CW, true);
}
