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
};

constexpr int PMAX = 384;                       // host bound on (R+2)*(CW+2)
constexpr int NIV = (3 * PMAX + 255) / 256;     // image elements per thread per tile

template <typename TI> __device__ __forceinline__ float ld_f32(const TI* p) { return (float)*p; }

// Branch-free image patch fetch: element i = (c, py, px) of the 3 x (R+2) x (CW+2) patch of tile (b, oy0, ox0).
template <typename TI>
__device__ __forceinline__ void fetch_patch(const FirstArgs& a, const TI* img, int b, int oy0, int ox0, int PW, int PP,
                                            float inv_pp, bf16 (&v)[NIV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NIV; ++j) {
        const int i = tid + j * 256;
        const int c = fastdiv(i, PP, inv_pp), rem = i - c * PP;
        const int py = fastdiv(rem, PW, a.inv_pw), px = rem - py * PW;
        const int iy = oy0 + py - a.pad, ix = ox0 + px - a.pad;
        const bool ok = c < 3 && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        const long off = ok ? ((long)(b * 3 + c) * a.IH + iy) * a.IW + ix : 0;
        const float f = ld_f32(img + off);
        v[j] = (bf16)(ok ? f : 0.f);
    }
}

__device__ __forceinline__ void stash_patch(unsigned char* patch, int n, const bf16 (&v)[NIV]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NIV; ++j) {
        const int i = tid + j * 256;
        const bf16 t = v[j];
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
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = fastdiv(tile, a.tiles_x, a.inv_tx), tx = tile - ty * a.tiles_x;
    const int oy0 = ty * a.R, ox0 = tx * a.CW;
    const TI* img = static_cast<const TI*>(a.img);

    bf16 pv[NIV];
    fetch_patch<TI>(a, img, b, oy0, ox0, PW, PP, 1.0f / PP, pv);

    // weight fragment (MFMA A operand: row = output channel r, k = 8q..8q+7), straight from the f32 master
    bf16x8 fw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const float wk = a.w[r * 27 + (k < 27 ? k : 26)];     // unconditional load, zeroed by the multiply
        fw[j] = (bf16)(wk * (k < 27 ? 1.f : 0.f));
    }
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = gather_off(8 * q + j, PW, PP) * 2;

    stash_patch(smem, 3 * PP, pv);
    __syncthreads();

    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = a.bias[q * 4 + e];
    }
    // Fragments in pairs: after the MFMA a lane holds 4 channels (8 bytes) of one pixel per fragment.  One
    // v_permlane16_swap per dword between the fragments of a pair leaves every lane with 8 consecutive channels
    // (16 bytes) of ONE pixel - even lane groups the pair's first fragment, odd groups the second - so the tile is
    // written with half as many, twice as wide stores.
#pragma unroll
    for (int fp2 = 0; fp2 < 2; ++fp2) {
        uint2 pk[2];
        long opix[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = (wave * 4 + fp2 * 2 + h) * 16 + r;
            int pr = 0, pc = 0;
            bool ok = m < MT;
            if (ok) { pr = fastdiv(m, a.CW, a.inv_cw); pc = m - pr * a.CW; }
            ok = ok && (oy0 + pr < a.OH) && (ox0 + pc < a.OW);
            opix[h] = ok ? ((long)(b * a.OH + oy0 + pr) * a.OW + ox0 + pc) : -1;
            const unsigned char* base = smem + (pr * PW + pc) * 2;
            bf16x8 fp;
#pragma unroll
            for (int j = 0; j < 8; ++j) fp[j] = *reinterpret_cast<const bf16*>(base + koff[j]);
            f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fp, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (bf16)(acc[e] + bv[e]);
                const float v = ok ? (float)o[e] : 0.f;          // statistics of what is stored
                s1[e] += v;
                s2[e] += v * v;
            }
            pk[h] = __builtin_bit_cast(uint2, o);
        }
        auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
        auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
        // lane group q now holds channels (q>>1)*8 .. +7 of the pixel of fragment (q&1) of the pair
        const long op = (q & 1) ? opix[1] : opix[0];
        if (op >= 0)
            *reinterpret_cast<uint4*>(a.y + op * 16 + (q >> 1) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
    if (a.stats) {
        __syncthreads();                                  // patch reads done: reuse LDS as [4 waves][16][2]
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t1 = sum16(s1[e]), t2 = sum16(s2[e]);
            if (r == 0) {
                red[(wave * 16 + q * 4 + e) * 2 + 0] = t1;
                red[(wave * 16 + q * 4 + e) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const float t = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
            const long blk = (long)b * gridDim.x + tile;
            a.stats[blk * 32 + tid] = t;
        }
    }
}

// Persistent workgroups over (image, tile); the 4 waves split the pixel reduction of a tile.  D rows = output channel
// (dy fragment by the transposing LDS read), D cols = k index (two 16-column tiles cover k = 0..31), MFMA K = 32
// pixels with the k <-> pixel map of conv.hip's wgrad: lane group q, element 4h+e <-> pixel ks + 16h + 4q + e.
template <typename TI>
__global__ __launch_bounds__(256) void conv3x3_first_wgrad(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PP = (a.R + 2) * PW, MT = a.R * a.CW;
    const int MTP = (MT + 31) / 32 * 32;
    unsigned char* patch = smem;
    unsigned char* dyl = smem + (3 * PP * 2 + 15) / 16 * 16;      // [MTP pixels][16 co] bf16, 32 bytes per pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4, rq = r >> 2, p = r & 3;
    const TI* img = static_cast<const TI*>(a.img);
    const bf16* dy = a.y;
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    const float inv_pp = 1.0f / PP;

    int koff[2];                                                  // gather offsets of k = r and k = 16 + r
    koff[0] = gather_off(r, PW, PP) * 2;
    koff[1] = gather_off(16 + r, PW, PP) * 2;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = fastdiv(t, tiles, a.inv_tiles), tl = t - b * tiles;
        const int tyy = fastdiv(tl, a.tiles_x, a.inv_tx);
        const int oy0 = tyy * a.R, ox0 = (tl - tyy * a.tiles_x) * a.CW;
        bf16 pv[NIV];
        fetch_patch<TI>(a, img, b, oy0, ox0, PW, PP, inv_pp, pv);
        uint4 dv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256, m = i >> 1, vv = i & 1;
            const int pr = fastdiv(m, a.CW, a.inv_cw), pc = m - pr * a.CW;
            const bool ok = m < MT && oy0 + pr < a.OH && ox0 + pc < a.OW;
            const long off = ok ? ((long)(b * a.OH + oy0 + pr) * a.OW + ox0 + pc) * 16 + vv * 8 : 0;
            const uint4 v = *reinterpret_cast<const uint4*>(dy + off);
            const unsigned msk = ok ? 0xFFFFFFFFu : 0u;
            dv[j] = make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
        }
        __syncthreads();                                          // previous tile's fragment reads are done
        stash_patch(patch, 3 * PP, pv);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;
            const uint4 tv = dv[j];
            if (i < MTP * 2) *reinterpret_cast<uint4*>(dyl + i * 16) = tv;
        }
        __syncthreads();
        for (int ks = wave * 32; ks < MTP; ks += 128) {
            bf16x8 fa, fb0, fb1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = ks + h * 16 + q * 4 + rq;           // pixel whose dy row this lane addresses
                bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(dyl + m * 32 + 4 * p * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[4 * h + e] = v[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int mm = ks + h * 16 + q * 4 + e;             // pixel of K element 4h+e of this lane group
                    if (mm >= MT) mm = 0;                         // dy row is zero there: any finite image value
                    const int pr = fastdiv(mm, a.CW, a.inv_cw);
                    const unsigned char* base = patch + (pr * PW + (mm - pr * a.CW)) * 2;
                    fb0[4 * h + e] = *reinterpret_cast<const bf16*>(base + koff[0]);
                    fb1[4 * h + e] = *reinterpret_cast<const bf16*>(base + koff[1]);
                }
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, acc[1], 0, 0, 0);
        }
    }
    // combine the 4 waves (fixed order) and write this workgroup's partial slab [16 co][32 k]
    __syncthreads();
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
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.pad = pad; a.R = R; a.CW = CW;
    a.tiles_x = (OW + CW - 1) / CW; a.tiles_y = (OH + R - 1) / R;
    a.inv_cw = 1.0f / CW; a.inv_pw = 1.0f / (CW + 2); a.inv_tx = 1.0f / a.tiles_x;
    a.inv_tiles = 1.0f / (a.tiles_x * a.tiles_y);
    return true;
}

}  // namespace

extern "C" int vqa_conv3x3_first_fwd(void* stream, int in_dtype, const void* images, const float* w, void* y,
                                     const float* bias, float* stats, int B, int IH, int IW, int OH, int OW, int pad,
                                     int R, int CW) {
    FirstArgs a{images, w, static_cast<bf16*>(y), bias, stats};
    if (!images || !w || !y || !first_args(a, B, IH, IW, OH, OW, pad, R, CW)) return VQA_ERR_ARG;
    size_t lds = (size_t)3 * (R + 2) * (CW + 2) * 2;
    if (lds < 512) lds = 512;
    dim3 grid(a.tiles_x * a.tiles_y, B);
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
                                       int CW) {
    FirstArgs a{images, nullptr, static_cast<bf16*>(const_cast<void*>(dy)), nullptr, part};
    if (!images || !dy || !part || !dw || G < 1 || !first_args(a, B, IH, IW, OH, OW, pad, R, CW)) return VQA_ERR_ARG;
    const int MTP = (R * CW + 31) / 32 * 32;
    size_t lds = (size_t)(3 * (R + 2) * (CW + 2) * 2 + 15) / 16 * 16 + (size_t)MTP * 32;
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
