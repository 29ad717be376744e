// First conv layer of the image encoder (reference cnn.py:23, Conv2d(3, 16, 3, padding)), bf16 path.
// The generic implicit-GEMM kernel needs the 3 image channels padded to one 16-channel MFMA k-group in an NHWC copy
// (5.3x the image bytes, plus the copy kernel).  Here the whole reduction is k = (c, ky, kx) = 27 <= 32 = ONE
// v_mfma_f32_16x16x32_bf16 per 16 output pixels, fed straight from the NCHW image (f16 / f32 / bf16, dataset.py:96):
//   conv3x3_first_fwd    y[B][OH][OW][16] (bf16, NHWC) = conv(images) + bias, with the BatchNorm partial statistics
//   conv3x3_first_wgrad  dW[16][3][3][3] = sum over pixels of dy (x) image patch, persistent workgroups over tiles
// LDS image patch: 3 planes of (R+2) x (CW+2) bf16; fragment elements are gathered with 2-byte LDS reads
// (k-major gather offsets are per-lane constants).  HBM-bound: bytes = |images| + |y|.
#include "common.h"

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

template <typename TI>
__global__ __launch_bounds__(256) void conv3x3_first_fwd(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    const unsigned ibytes = (unsigned)a.B * 3 * a.IH * a.IW * (unsigned)sizeof(TI);      // (launcher: < 4 GiB)
    const __amdgpu_buffer_rsrc_t ibuf = buffer_of(a.img, ibytes);
    if ((int)blockIdx.x >= total) return;

    PatchPlan pl;
    plan_patch<TI>(a, PW, PP, pl);
    // weight fragment (MFMA A operand: row = output channel r, k = 8q..8q+7), straight from the f32 master
    const float wsc = a.wscale ? a.wscale[r] : 1.f;
    bf16x8 fw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const float wk = a.w[r * 27 + (k < 27 ? k : 26)];     // unconditional load, zeroed by the multiply
        fw[j] = (bf16)(wk * (k < 27 ? wsc : 0.f));
    }
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = gather_off(8 * q + j, PW, PP) * 2;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = a.bias ? a.bias[q * 4 + e] : 0.f;
    // fragment pixel decode: fragment f of this wave = tile pixels (wave*4+f)*16 + r
    int pbase[4], frc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int m = (wave * 4 + f) * 16 + r;
        int pr = 0, pc = 0;
        if (m < MT) { pr = fastdiv(m, a.CW, a.inv_cw); pc = m - pr * a.CW; }
        pbase[f] = (pr * PW + pc) * 2;
        frc[f] = ((m < MT ? pr : 0x4000) << 16) | pc;
    }
    f32x2 s1[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    const f32x4 bias4 = {bv[0], bv[1], bv[2], bv[3]};

    float pv[NIV];
    auto tile_org = [&](int t, int& b, int& oy0, int& ox0) {
        b = fastdiv(t, tiles, a.inv_tiles);
        const int tl = t - b * tiles, ty = fastdiv(tl, a.tiles_x, a.inv_tx);
        oy0 = ty * a.R;
        ox0 = (tl - ty * a.tiles_x) * a.CW;
    };
    int fb, foy0, fox0;
    const bool relu = a.relu != 0;
    tile_org(blockIdx.x, fb, foy0, fox0);
    fetch_patch<TI>(a, ibuf, ibytes, pl, fb, foy0, fox0, pv);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        stash_patch(smem, 3 * PP, pv);
        __syncthreads();
        const int b = fb, oy0 = foy0, ox0 = fox0;
        if (t + (int)gridDim.x < total) {
            tile_org(t + gridDim.x, fb, foy0, fox0);
            fetch_patch<TI>(a, ibuf, ibytes, pl, fb, foy0, fox0, pv);
        }
        bf16* yt = a.y + ((long)(b * a.OH + oy0) * a.OW + ox0) * 16 + (q >> 1) * 8;
        // Fragments in pairs: after the MFMA a lane holds 4 channels (8 bytes) of one pixel per fragment.  One
        // v_permlane16_swap per dword between the fragments of a pair leaves every lane with 8 consecutive channels
        // (16 bytes) of ONE pixel - even lane groups the pair's first fragment, odd groups the second - so the tile
        // is written with half as many, twice as wide stores.
#pragma unroll
        for (int fp2 = 0; fp2 < 2; ++fp2) {
            uint2 pk[2];
            int orel[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f = fp2 * 2 + h;
                const int pr = frc[f] >> 16, pc = frc[f] & 0xFFFF;
                const bool ok = oy0 + pr < a.OH && ox0 + pc < a.OW;
                orel[h] = ok ? (pr * a.OW + pc) * 16 : -1;
                const unsigned char* base = smem + pbase[f];
                bf16x8 fp;
#pragma unroll
                for (int j = 0; j < 8; ++j) fp[j] = *reinterpret_cast<const bf16*>(base + koff[j]);
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fp, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const float okf = ok ? 1.f : 0.f;
                const Packed4<bf16> o = relu ? epilogue4<bf16, true>(acc, bias4, okf, s1, s2)
                                             : epilogue4<bf16, false>(acc, bias4, okf, s1, s2);
                pk[h] = make_uint2(o.lo, o.hi);
            }
            auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
            auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
            // lane group q now holds channels (q>>1)*8 .. +7 of the pixel of fragment (q&1) of the pair
            const int op = (q & 1) ? orel[1] : orel[0];
            if (op >= 0) *reinterpret_cast<uint4*>(yt + op) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
        __syncthreads();                                      // fragment reads done before the next stash
    }
    if (a.stats) {                                            // one row of partial sums per workgroup
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
template <typename TI>
__global__ __launch_bounds__(256) void conv3x3_first_wgrad(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int MTP = (MT + 31) / 32 * 32;
    unsigned char* patch = smem;
    unsigned char* dyl = smem + (3 * PP * 2 + 15) / 16 * 16;      // [MTP pixels][16 co] bf16, 32 bytes per pixel
    short* ptab = reinterpret_cast<short*>(dyl + MTP * 32);       // [MTP] byte offset of pixel m inside a patch plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4, rq = r >> 2, p = r & 3;
    const unsigned ibytes = (unsigned)a.B * 3 * a.IH * a.IW * (unsigned)sizeof(TI);      // (launcher: both < 4 GiB)
    const unsigned dbytes = (unsigned)a.B * a.OH * a.OW * 32u;
    const __amdgpu_buffer_rsrc_t ibuf = buffer_of(a.img, ibytes), dbuf = buffer_of(a.y, dbytes);
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    if ((int)blockIdx.x >= total) return;

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
    uint4 dv[2];
    auto fetch = [&](int t) {
        int b, oy0, ox0;
        tile_org(t, b, oy0, ox0);
        fetch_patch<TI>(a, ibuf, ibytes, pl, b, oy0, ox0, pv);
        const unsigned dorg = (unsigned)((b * a.OH + oy0) * a.OW + ox0) * 32u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = oy0 + (drc[j] >> 16) < a.OH && ox0 + (drc[j] & 0xFFFF) < a.OW;
            dv[j] = buffer_load16(dbuf, ok ? dorg + (unsigned)drel[j] : dbytes);
        }
    };
    fetch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        stash_patch(patch, 3 * PP, pv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;
            const uint4 tv = dv[j];
            if (i < MTP * 2) *reinterpret_cast<uint4*>(dyl + i * 16) = tv;
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x);
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

bool first_args(FirstArgs& a, int B, int IH, int IW, int OH, int OW, int pad, int R, int CW) {
    if (OH != IH + 2 * pad - 2 || OW != IW + 2 * pad - 2 || R < 1 || CW < 1 || R * CW > 256 ||
        (R + 2) * (CW + 2) > PMAX || B < 1)
        return false;
    // the kernels address images and y through 32-bit buffer offsets (out-of-range offset = zero padding)
    if ((size_t)B * 3 * IH * IW * 4 >= (1ull << 32) || (size_t)B * OH * OW * 32 >= (1ull << 32)) return false;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.pad = pad; a.R = R; a.CW = CW;
    a.tiles_x = (OW + CW - 1) / CW; a.tiles_y = (OH + R - 1) / R;
    a.inv_cw = 1.0f / CW; a.inv_pw = 1.0f / (CW + 2); a.inv_tx = 1.0f / a.tiles_x;
    a.inv_tiles = 1.0f / (a.tiles_x * a.tiles_y);
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
    if (!images || !w || !y || !first_args(a, B, IH, IW, OH, OW, pad, R, CW)) return VQA_ERR_ARG;
    if (G < 1 || G > a.tiles_x * a.tiles_y * B) return VQA_ERR_ARG;
    size_t lds = (size_t)3 * (R + 2) * (CW + 2) * 2;
    if (lds < 512) lds = 512;
    dim3 grid(G);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype == VQA_F16) hipLaunchKernelGGL(conv3x3_first_fwd<_Float16>, grid, dim3(256), lds, s, a);
    else if (in_dtype == VQA_F32) hipLaunchKernelGGL(conv3x3_first_fwd<float>, grid, dim3(256), lds, s, a);
    else if (in_dtype == VQA_BF16) hipLaunchKernelGGL(conv3x3_first_fwd<bf16>, grid, dim3(256), lds, s, a);
    else return VQA_ERR_ARG;
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
    if (!images || !dy || !part || !dw || G < 1 || !first_args(a, B, IH, IW, OH, OW, pad, R, CW)) return VQA_ERR_ARG;
    const int MTP = (R * CW + 31) / 32 * 32;
    if (G > a.tiles_x * a.tiles_y * B) return VQA_ERR_ARG;
    size_t lds = (size_t)(3 * (R + 2) * (CW + 2) * 2 + 15) / 16 * 16 + (size_t)MTP * 32 + (size_t)MTP * 2;
    if (lds < 3 * 2 * 64 * 16) lds = 3 * 2 * 64 * 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype == VQA_F16) hipLaunchKernelGGL(conv3x3_first_wgrad<_Float16>, dim3(G), dim3(256), lds, s, a);
    else if (in_dtype == VQA_F32) hipLaunchKernelGGL(conv3x3_first_wgrad<float>, dim3(G), dim3(256), lds, s, a);
    else if (in_dtype == VQA_BF16) hipLaunchKernelGGL(conv3x3_first_wgrad<bf16>, dim3(G), dim3(256), lds, s, a);
    else return VQA_ERR_ARG;
    hipLaunchKernelGGL(first_wgrad_reduce, dim3(108), dim3(256), 0, s, part, dw, G);
    CHECK_LAUNCH();
    return VQA_OK;
}
