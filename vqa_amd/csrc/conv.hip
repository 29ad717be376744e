// 3x3 stride-1 convolution for the image encoder (reference cnn.py:23-40), NHWC, as implicit GEMM on MFMA.
//
//   conv3x3_igemm   forward (y = conv(x, W) + bias, with per-channel sum / sum-of-squares partials for the
//                   following BatchNorm) and, with rotated/transposed packed weights and pad' = 2 - pad, the
//                   data gradient (dgrad).  GEMM view: D[co][pixel] = sum_k Wp[co][k] * patch[k][pixel],
//                   k = (tap, ci).  One workgroup = R x CW output pixels (<= 256) of one image x <= 64 output
//                   channels; the input patch (R+2)x(CW+2) and the weight slab are staged in LDS per
//                   input-channel chunk of 64 bytes, so every input byte is read from L2/HBM once per tile and
//                   the 9 taps re-read it from LDS.
//   conv3x3_wgrad   weight gradient dW[co][tap][ci] = sum_pixels dy[pixel][co] * x[pixel+tap][ci]: the GEMM
//                   reduction runs over pixels, which is the slow axis of both NHWC operands, so bf16 fragments
//                   are fetched with the transposing LDS read (ds_read_b64_tr_b16) and f32 ones with plain
//                   4-byte reads.  Partials per workgroup, summed in fixed order by conv3x3_wgrad_reduce.
//   pack kernels    torch-layout f32 weights [Cout][Cin][3][3] -> k-contiguous slabs for fwd and dgrad.
#include "common.h"
#include <type_traits>

namespace {

// LDS strides chosen so that the ds_read_b128 fragment reads (lane = 16*q + r reads 16 bytes at r*stride + 16*q) and
// the ds_read_b64_tr_b16 reads are bank-conflict-free on gfx950: in 16-byte slots the stride must be 2 (mod 4) for
// b128 (brute-forced over the hardware's lane groups), and a multiple of 32 bytes with an odd quotient for tr_b16.
__host__ __device__ constexpr int pix_stride(int data_bytes) { return data_bytes == 32 ? 32 : data_bytes + 32; }
__host__ __device__ constexpr int row_stride(int data_bytes) { return data_bytes + 32; }   // 576->608, 320->352

struct ConvArgs {
    const void* x;      // [B][IH][IW][Cin]
    const void* w;      // packed [Cin/CK][Cout][KP]
    void* y;            // [B][OH][OW][Cout]
    const float* bias;  // [Cout] or null
    float* stats;       // [tiles*B][Cout][2] partial (sum, sumsq) or null
    int B, IH, IW, Cin, OH, OW, Cout, pad;
    int R, CW, tiles_x, tiles_y;
    float inv_cw, inv_pw, inv_tx, inv_tiles;
    int relu;           // epilogue: bit 0: y = max(conv + bias, 0)  (eval mode with the BatchNorm folded into w / bias);
                        // bit 1 (with bit 0; bf16, even tiles on an even map, 16 columns wide in the single-chunk
                        // kernel): the 2x2 max pool behind it as well - y is then [B][OH/2][OW/2][Cout]
    // XF kernels: x is the RAW output of the previous conv and the input really is relu(in_scale * x + in_shift) - the
    // previous layer's BatchNorm + ReLU, applied per 16-byte vector on the way into LDS ([Cin] each).  Only for
    // pad == 0 (every conv that follows an un-pooled block, cnn.py:24,28,32,36,40): no zero padding to preserve.
    const float* in_scale;
    const float* in_shift;
    // BST kernels (data gradient): y = d is the gradient w.r.t. the BatchNorm + ReLU output of the layer below (no pool
    // in between), bn_y that layer's raw conv output (same shape as y) and bn_coef its [4][Cout] scale, shift, mean,
    // invstd: `stats` then receives that layer's BatchNorm-BACKWARD partial sums (sum dz, sum dz*xhat) instead of the
    // forward statistics - its reduce pass (bn.hip) never runs, d is not re-read.
    const void* bn_y;
    const float* bn_coef;
    int rev;            // persistent kernels: walk the units back to front (unit_walk, common.h)
    // BST + weight gradient (persistent bf16 kernels, VAR 3): the conv this call is the data gradient OF also gets its
    // weight gradient here - x = relu(scale * bn_y + shift) at the tile's pixels (already loaded for the sums) against
    // the 9 shifted views of the staged dy patch - as one partial slab per workgroup, [G][Cin][9][Cout] f32 in THIS
    // call's channel names (= vqa_conv3x3_wgrad's [G][Cout][9][Cin] in the forward conv's).
    float* wpart;
    // Two images per tile (multi-chunk kernel, pad 0, one tile per image: the 10x10 maps fill 7 of a tile's 16 fragment
    // slots): consecutive images are contiguous in NHWC, so a PAIR is one image of 2 IH rows whose valid convolution has
    // 2 OH + 2 rows - `seam` = the real OH; output rows seam, seam + 1 straddle the two images and are dropped, rows from
    // seam + 2 on are rows 0.. of the second image.  B, IH, OH in here are then those of the pairs.  0 = off.
    int seam;
};

template <typename T, int CK> struct ConvCfg {
    static constexpr int VEC = Mma<T>::VEC;
    static constexpr int KC = 4 * VEC;                       // k per MFMA chunk
    static constexpr int KP = (9 * CK + KC - 1) / KC * KC;   // padded k per channel chunk
    static constexpr int NS = KP / KC;                       // chunks per channel chunk
    static constexpr int PXB = pix_stride(CK * (int)sizeof(T));   // LDS bytes per patch pixel
    // Extra bytes per patch ROW.  A fragment of 16 consecutive tile pixels wraps to the next tile row whenever CW is not
    // a multiple of 16 (every layer but the 16x16-tiled ones); the wrap must advance the LDS address like one more
    // pixel step modulo the 256-byte bank period, or the fragment's ds_read_b128 costs 8 LDS cycles instead of 4
    // (brute-forced over the hardware's lane groups for every tile shape in use): (PW - CW) * PXB + RPAD = 0 mod 256.
    static constexpr int RPAD = (256 - (2 * PXB) % 256) % 256;
    static constexpr int WROWB = row_stride(KP * (int)sizeof(T)); // LDS bytes per weight row
    static constexpr int VPP = CK * (int)sizeof(T) / 16;     // 16-byte vectors per patch pixel
};

#ifndef VQA_PLAN_RESERVE_DEFAULT
#define VQA_PLAN_RESERVE_DEFAULT 1
#endif
#ifndef VQA_MFMA_STATS
#define VQA_MFMA_STATS 1
#endif
#ifdef VQA_STAMPS
__device__ unsigned long long g_stamps[16384][12];
#define STAMP(i) do { if (threadIdx.x == 0) { const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
                      if (wg < 16384) g_stamps[wg][i] = __builtin_readcyclecounter(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------- igemm
// One workgroup per (tile, image, channel block): the multi-chunk (Cin > CK) layers, which are MFMA-bound and
// want their registers for the accumulators and the chunk prefetch rather than for persistent-loop state.
template <typename T, int CK, int NBF, int VAR>   // VAR: 0 plain, 1 XF (input BatchNorm + ReLU), 2 BST (see ConvArgs),
                                                  // 3 POOL (inference: ReLU + 2x2 max pool in the epilogue, relu = 3)
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void conv3x3_igemm_tile(ConvArgs a) {   // (two waves per SIMD)
    constexpr bool XF = VAR == 1, BST = VAR == 2, POOL = VAR == 3;
    static_assert(!POOL || sizeof(T) == 2, "pooled epilogue: bf16");
    using Cfg = ConvCfg<T, CK>;
    constexpr int VEC = Cfg::VEC, KC = Cfg::KC, KP = Cfg::KP, NS = Cfg::NS, PXB = Cfg::PXB, WROWB = Cfg::WROWB,
                  VPP = Cfg::VPP;
    constexpr int NB = NBF * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PPIX = (a.R + 2) * PW, PROWB = PW * PXB + Cfg::RPAD;
    unsigned char* patch = smem;
    unsigned char* wl = smem + (a.R + 2) * PROWB;
    float* coefl = reinterpret_cast<float*>(wl + NB * WROWB);   // XF: [2][Cin] input scale / shift
    // byte offset (from `patch`) of the 4 KB dump area behind the coefficient rows (see the staging plan)
    const int dump = (a.R + 2) * PROWB + NB * WROWB + NB * 4 + 2 * a.Cin * 4 + 4 * NB * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    STAMP(0);
    // 1-D grid of 8 * ceil(N / 8) workgroups over the N = tiles * B * (Cout / NB) units, channel block fastest: XCD k
    // (workgroups k, k + 8, ...) takes the k-th contiguous eighth of the list, so the channel blocks of one tile and
    // the tiles of one image - which read the same input rows - run side by side on ONE L2 (see unit_walk, common.h)
    const int tiles_img = a.tiles_x * a.tiles_y, zb = a.Cout / NB, nunits = tiles_img * a.B * zb;
    int unit = blockIdx.x;
    if (VQA_XCD_WALK) {
        const int k = blockIdx.x & 7;
        unit = (int)(((long)nunits * k) >> 3) + (blockIdx.x >> 3);
        if (unit >= (int)(((long)nunits * (k + 1)) >> 3)) return;
    } else if (unit >= nunits) return;
    const int ut = unit / zb;
    const int co0 = (unit - ut * zb) * NB;
    int tile, tx;
    const int b = fastdivmod(ut, tiles_img, a.inv_tiles, tile);
    // the coefficient rows go to LDS (one element per thread: Cin <= 256, 4 * NB <= 256, checked by the launcher): the
    // loads leave here and are written behind the first chunk's loads, so that nothing waits for them alone
    // (threads beyond the row re-write element 0 with the same value: a conditional write lets the compiler sink the
    // loads into its block, behind the chunk's loads, and the write then waits for all of them)
    float cf0 = 0.f, cf1 = 0.f;
    const int cfi = XF ? (tid < a.Cin ? tid : 0) : (tid < 4 * NB ? tid : 0);
    if constexpr (XF) {                                         // (read by the first stash: barrier below)
        cf0 = a.in_scale[cfi];
        cf1 = a.in_shift[cfi];
    }
    if constexpr (BST) cf0 = a.bn_coef[(cfi / NB) * a.Cout + co0 + cfi % NB];   // [4][NB] of this block (epilogue)
    const int ty = fastdivmod(tile, a.tiles_x, a.inv_tx, tx);
    const int oy0 = ty * a.R, ox0 = tx * a.CW;
    const int MT = a.R * a.CW;
    const T* w = static_cast<const T*>(a.w);

    // ---- staging plan.  Every thread moves up to NPV patch vectors and NWV weight vectors per channel chunk.  The
    // loads are branch-free (out-of-image / out-of-range lanes read a valid dummy address and are zeroed by a
    // select) so that all of a chunk's loads are in flight together, and the loads of chunk ch+1 are issued before
    // the MFMA phase of chunk ch and written to LDS after it (register prefetch, single LDS buffer).
    constexpr int PMAX = 384;                                   // host bound on (R+2)*(CW+2)
    constexpr int NPV = (PMAX * VPP + 255) / 256;
    constexpr int WV = KP / VEC;                                // vectors per weight row
    constexpr int NWV = (NB * WV + 255) / 256;
    // Out-of-image vectors get a byte offset >= 2^31, beyond the buffer (the launcher keeps x below 2 GiB): the
    // hardware answers such a buffer load with zeros - the zero padding needs no mask at the LDS write.
    uint4 pv[NPV], wv[NWV];
    auto fetch_w = [&](int ch) {
        const T* wsrc = w + ((long)ch * a.Cout + co0) * KP;     // the chunk's NB rows are contiguous
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int i = tid + j * 256;
            wv[j] = *reinterpret_cast<const uint4*>(wsrc + (i < NB * WV ? i : 0) * VEC);
        }
    };
    // the first chunk's loads leave before anything else is computed - the weights, which need no plan, first: the
    // staging plan, the fragment decode and the bias loads below run under their latency (in-kernel stamps: the
    // un-overlapped first fetch was 9-16 % of a workgroup's life)
    fetch_w(0);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned xbytes = (unsigned)a.B * a.IH * a.IW * a.Cin * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xbuf = buffer_of(a.x, xbytes);
    const unsigned img0 = (unsigned)b * a.IH * a.IW * a.Cin * (unsigned)sizeof(T);
    unsigned psrc[NPV];                                         // byte offset inside the image, or 0x80000000
    int pdst[NPV];                                              // byte offset inside the LDS patch
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        const int i = tid + j * 256;
        const int px = i / VPP, vv = i % VPP;
        int pxx;
        const int py = fastdivmod(px, PW, a.inv_pw, pxx);
        const int iy = oy0 + py - a.pad, ix = ox0 + pxx - a.pad;
        // (unsigned compares fold the two-sided range tests; everything is computed for every lane and selected -
        // the compiler turned the conditional forms into a branch per vector; 24-bit multiplies: the launcher keeps
        // IH * IW and Cin below 2^23: __mul24 is signed)
        const bool ok = i < PPIX * VPP && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const unsigned src = (unsigned)(__mul24(__mul24(iy, a.IW) + ix, a.Cin) + vv * VEC) * (unsigned)sizeof(T);
        psrc[j] = ok ? src : 0x80000000u;
        // vectors beyond the patch go to a per-thread dump slot: the stash is straight-line code (a branch per vector
        // cut it into 15 basic blocks per chunk, each with its own wait)
        const int dst = __mul24(py, PROWB) + __mul24(pxx, PXB) + vv * 16;
        pdst[j] = i < PPIX * VPP ? dst : dump + tid * 16;
    }
    auto fetch_p = [&](int ch) {
        const unsigned base = img0 + (unsigned)(ch * CK) * (unsigned)sizeof(T);
#pragma unroll
        for (int j = 0; j < NPV; ++j) pv[j] = buffer_load16(xbuf, psrc[j] + base);
    };
    auto fetch = [&](int ch) { fetch_p(ch); fetch_w(ch); };
    auto stash = [&](int ch) {
        float isc[VEC], ish[VEC];                               // XF: this thread's channel vector (256 % VPP == 0)
        if constexpr (XF) {
            const float* cs = coefl + ch * CK + (tid % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                *reinterpret_cast<float4*>(isc + e) = *reinterpret_cast<const float4*>(cs + e);
                *reinterpret_cast<float4*>(ish + e) = *reinterpret_cast<const float4*>(cs + a.Cin + e);
            }
        }
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            uint4 t = pv[j];
            if constexpr (XF) t = bn_relu_vec<T>(t, isc, ish);
            *reinterpret_cast<uint4*>(patch + pdst[j]) = t;
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
            const int i = tid + j * 256;
            const uint4 t = wv[j];
            if (NB * WV % 256 == 0 || i < NB * WV) *reinterpret_cast<uint4*>(wl + (i / WV) * WROWB + (i % WV) * 16) = t;
        }
    };
    fetch_p(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XF) { coefl[cfi] = cf0; coefl[a.Cin + cfi] = cf1; }
    if constexpr (BST) coefl[cfi] = cf0;
    // per-fragment pixel decode.  The tile's fragments of 16 consecutive pixels are dealt to the waves round-robin
    // (fragment f of wave w = fragment 4f + w of the tile) and a wave skips the MFMAs of fragments beyond the tile:
    // the small deep maps fill only part of the 16 slots (10x10: 7, 12x12: 9, 14x14: 13), and dealt in blocks of four
    // the live ones would all land on the first waves.
    const int nfw = ((MT + 15) / 16 - wave + 3) / 4;        // live fragments of this wave (wave-uniform)
    int pbase[4], opix[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int m = (f * 4 + wave) * 16 + r;
        int pc;
        int pr = fastdivmod(m < MT ? m : 0, a.CW, a.inv_cw, pc);
        bool ok = m < MT && (oy0 + pr < a.OH) && (ox0 + pc < a.OW);
        pbase[f] = __mul24(pr, PROWB) + __mul24(pc, PXB);
        int orow = b * a.OH + oy0 + pr;                                  // row of the [B * OH] x OW output
        if (a.seam) {                                                       // (see ConvArgs::seam; oy0 = 0 there)
            ok = ok && (unsigned)(pr - a.seam) >= 2u;
            orow = __mul24(b, 2 * a.seam) + pr - (pr >= a.seam + 2 ? 2 : 0);
        }
        const int op = __mul24(orow, a.OW) + ox0 + pc;
        opix[f] = ok ? op : -1;
    }

    f32x4 acc[4][NBF];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int n = 0; n < NBF; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    float bv[NBF][4];                                       // bias of this lane's channels, loaded up front
    {
        const bool hasb = !BST && a.bias;                   // (one 16-byte load per block from a valid address, selected)
        const float* bsrc = hasb ? a.bias + co0 + q * 4 : reinterpret_cast<const float*>(a.w);
#pragma unroll
        for (int n = 0; n < NBF; ++n) {
            const float4 t = *reinterpret_cast<const float4*>(bsrc + n * 16);
            bv[n][0] = hasb ? t.x : 0.f; bv[n][1] = hasb ? t.y : 0.f;
            bv[n][2] = hasb ? t.z : 0.f; bv[n][3] = hasb ? t.w : 0.f;
        }
    }
    // BST: this lane's 4 channels of the lower layer's y at each of its output pixels, loaded during the LAST chunk's
    // MFMA phase (out-of-tile pixels read zeros beyond the buffer): they take the registers the chunk prefetch no
    // longer needs there - loaded up front they are live through the whole loop, and the 64-channel form then needs
    // 264 registers: one wave per SIMD instead of two
    Raw4<T> yv[BST ? 4 : 1][BST ? NBF : 1];
    auto load_y = [&]() {
        const unsigned ybytes = (unsigned)a.B * a.OH * a.OW * a.Cout * (unsigned)sizeof(T);
        const __amdgpu_buffer_rsrc_t bybuf = buffer_of(a.bn_y, ybytes);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int n = 0; n < NBF; ++n)
                yv[f][n] = buffer_load_raw4<T>(bybuf, opix[f] >= 0 ? (unsigned)(opix[f] * a.Cout + co0 + n * 16 + q * 4) *
                                                                       (unsigned)sizeof(T) : ybytes);
    };
    if constexpr (XF) __syncthreads();                          // coefl is read by the first stash

    const int nchunks = a.Cin / CK;
    // the channel-chunk loop, specialised on the number of live fragments (a run-time test per fragment inside the
    // unrolled k-steps costs the full tiles 10 %: it cuts the MFMA / LDS-read schedule into small blocks)
    auto chunks = [&](auto nf_tag) {
        constexpr int NF = decltype(nf_tag)::value;
        STAMP(1);
        auto chunk = [&](int ch, auto last_tag) {
            stash(ch);
            if (ch == 0) STAMP(2);
            __syncthreads();
            if (ch == 0) STAMP(3);
            constexpr int LAST = decltype(last_tag)::value;     // 0 no, 1 yes, 2 run-time test
            if constexpr (LAST == 0) fetch(ch + 1);
            else if constexpr (LAST == 2) { if (ch + 1 < nchunks) fetch(ch + 1); }
            else if constexpr (BST) load_y();
            // ---- MFMA over 9 taps x CK channels
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k0 = s * KC + q * VEC;
                int tap = k0 / CK;
                const int c = k0 % CK;
                tap = tap > 8 ? 8 : tap;  // padded k: weights are zero there, read any finite data
                const int toff = (tap / 3) * PROWB + (tap % 3) * PXB + c * (int)sizeof(T);
                uint4 fw[NBF], fp[NF > 0 ? NF : 1];
                if (NF > 0) {
#pragma unroll
                    for (int n = 0; n < NBF; ++n)
                        fw[n] = *reinterpret_cast<const uint4*>(wl + (n * 16 + r) * WROWB + k0 * (int)sizeof(T));
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) fp[f] = *reinterpret_cast<const uint4*>(patch + pbase[f] + toff);
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int n = 0; n < NBF; ++n) Mma<T>::mma(acc[f][n], fw[n], fp[f]);
            }
            if (ch == 0) STAMP(4);
            __syncthreads();
            if (ch == 0) STAMP(5);
        };
        if constexpr (BST) {                                    // (peeled only where the last chunk differs: the plain
            for (int ch = 0; ch + 1 < nchunks; ++ch) chunk(ch, std::integral_constant<int, 0>{});   // forms need more
            chunk(nchunks - 1, std::integral_constant<int, 1>{});                                    // registers peeled)
        } else {
            for (int ch = 0; ch < nchunks; ++ch) chunk(ch, std::integral_constant<int, 2>{});
        }
        STAMP(6);
    };
    if (nfw >= 4) chunks(std::integral_constant<int, 4>{});
    else if (nfw == 3) chunks(std::integral_constant<int, 3>{});
    else if (nfw == 2) chunks(std::integral_constant<int, 2>{});
    else if (nfw == 1) chunks(std::integral_constant<int, 1>{});
    else chunks(std::integral_constant<int, 0>{});

    // ---- epilogue.  After the MFMAs a lane owns 4 channels of one pixel per (fragment, channel block): stored as
    // they are, every store instruction scatters 8-byte pieces over 16 pixel rows (measured: the epilogue took 29 %
    // of the workgroup's lifetime).  Instead each wave transposes its 64 pixels x NB channels through a private LDS
    // region (the staging LDS is free after the last barrier) and writes whole pixel rows, 16 bytes per lane.
    T* y = static_cast<T*>(a.y);
    constexpr int EROW = NB * (int)sizeof(T) + 16;          // LDS bytes per pixel row of the transpose
    constexpr int VPR = NB * (int)sizeof(T) / 16;           // 16-byte vectors per pixel row
    unsigned char* ew = smem + wave * 64 * EROW;
    int* epix = reinterpret_cast<int*>(smem + 4 * 64 * EROW) + wave * 64;
    f32x2 s1[NBF][2], s2[NBF][2];
#pragma unroll
    for (int n = 0; n < NBF; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) s1[n][h] = s2[n][h] = f32x2{0.f, 0.f};
    // bf16: the BatchNorm statistics are taken by MFMA from the transposed tile (below), not per element here - the
    // rows of pixels outside the image are then written as zeros
    constexpr bool MS = VQA_MFMA_STATS && sizeof(T) == 2 && !BST;
    auto finish = [&](auto mode_tag) {
        constexpr bool RELU = (decltype(mode_tag)::value & 1) != 0, STATS = (decltype(mode_tag)::value & 2) != 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float okf = opix[f] >= 0 ? 1.f : 0.f;
            if (q == 0) epix[f * 16 + r] = opix[f];
#pragma unroll
            for (int n = 0; n < NBF; ++n) {
                const f32x4 bias4 = {bv[n][0], bv[n][1], bv[n][2], bv[n][3]};
                Packed4<T> o = epilogue4<T, RELU, STATS && !MS>(acc[f][n], bias4, okf, s1[n], s2[n]);
                if constexpr (MS && STATS) {
                    if (opix[f] < 0) o = Packed4<T>{};
                }
                // (POOL: rows in TILE order - fragment f of this wave is fragment 4 f + wave of the tile - for the
                // windows that the pooled stores below read across the waves' fragments)
                if constexpr (POOL)
                    store_packed4<T>(reinterpret_cast<T*>(smem + ((f * 4 + wave) * 16 + r) * EROW) + n * 16 + q * 4, o);
                else
                    store_packed4<T>(reinterpret_cast<T*>(ew + (f * 16 + r) * EROW) + n * 16 + q * 4, o);
            }
        }
    };
    if constexpr (BST) {
        // the coefficients sit beyond the transpose regions' reach?  No: ew / epix / red overlay the staging LDS from
        // offset 0 and coefl lies behind patch + weights - copy this lane's coefficients to registers first
        f32x4 cf[NBF][4];
#pragma unroll
        for (int n = 0; n < NBF; ++n)
#pragma unroll
            for (int k = 0; k < 4; ++k) cf[n][k] = *reinterpret_cast<const f32x4*>(coefl + k * NB + n * 16 + q * 4);
        __syncthreads();                                    // every wave has its coefficients before LDS is reused
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float okf = opix[f] >= 0 ? 1.f : 0.f;
            if (q == 0) epix[f * 16 + r] = opix[f];
#pragma unroll
            for (int n = 0; n < NBF; ++n) {
                const Packed4<T> o = epilogue4_bnbwd<T>(acc[f][n], okf, yv[f][n], cf[n], s1[n], s2[n]);
                store_packed4<T>(reinterpret_cast<T*>(ew + (f * 16 + r) * EROW) + n * 16 + q * 4, o);
            }
        }
    } else {
        switch (((a.relu & 1) ? 1 : 0) | (a.stats ? 2 : 0)) {
            case 0: finish(std::integral_constant<int, 0>{}); break;
            case 1: finish(std::integral_constant<int, 1>{}); break;
            case 2: finish(std::integral_constant<int, 2>{}); break;
            default: finish(std::integral_constant<int, 3>{}); break;
        }
    }
    STAMP(7);
    if constexpr (POOL) {
        // The tile (R x CW, both even, on an even map at an even origin: whole windows) sits in LDS as bf16 rows of NB
        // channels; a thread takes one 16-byte channel vector of one window: four row reads, three packed signed
        // 16-bit maxima per dword (the values are relu'd: >= 0), one store to the pooled map.  Same bits as the
        // pool-only pass over the stored activation.
        __syncthreads();
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        auto pmax = [](unsigned u, unsigned v) {
            return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u),
                                                                          __builtin_bit_cast(s16x2, v)));
        };
        auto vmax = [&](const uint4& u, const uint4& v) {
            return make_uint4(pmax(u.x, v.x), pmax(u.y, v.y), pmax(u.z, v.z), pmax(u.w, v.w));
        };
        const int PC = a.CW >> 1, PHh = a.OH >> 1, PWh = a.OW >> 1;
        T* yp = y + ((long)(b * PHh + (oy0 >> 1)) * PWh + (ox0 >> 1)) * a.Cout + co0;
        for (int idx = tid; idx < (a.R >> 1) * PC * VPR; idx += 256) {
            const int pp = idx / VPR, c = idx - pp * VPR;
            const int ppr = pp / PC, ppc = pp - ppr * PC;
            const unsigned char* p0 = smem + (2 * ppr * a.CW + 2 * ppc) * EROW + c * 16;
            const uint4 top = vmax(*reinterpret_cast<const uint4*>(p0), *reinterpret_cast<const uint4*>(p0 + EROW));
            const uint4 bot = vmax(*reinterpret_cast<const uint4*>(p0 + a.CW * EROW),
                                   *reinterpret_cast<const uint4*>(p0 + (a.CW + 1) * EROW));
            if (oy0 + 2 * ppr + 1 < a.OH && ox0 + 2 * ppc + 1 < a.OW)
                *reinterpret_cast<uint4*>(yp + ((long)ppr * PWh + ppc) * a.Cout + c * 8) = vmax(top, bot);
        }
        return;
    }
    if (a.stats) {
        // [4 waves][NB][2] behind the transpose regions: the sums are ready before any global store is issued, so the
        // barrier below does not wait for stores
        float* red = reinterpret_cast<float*>(smem + 4 * 64 * EROW + 4 * 64 * 4);
        if constexpr (MS) {
            // Sum y and sum y^2 over this wave's 64 pixels per channel as two MFMAs per (16-channel block, 32 pixels):
            // the fragment F (k = pixel, 16 channels; transposing LDS reads of the tile just written) is BOTH operands
            // of the Gram product F^T F, whose diagonal is sum y^2, and the B operand under an all-ones A for sum y.
            // ~3.5 VALU per accumulator element (unpack, mask, add, fma: ~900 instructions per wave) become ~60.
            if constexpr (sizeof(T) == 2) {
                const int rq = r >> 2, p4 = r & 3;
                const bf16 one = (bf16)1.0f;
                const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
                for (int n = 0; n < NBF; ++n) {
                    f32x4 gram = {0.f, 0.f, 0.f, 0.f}, ssum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 64; ks += 32) {
                        const unsigned char* ad = ew + (n * 16 + 4 * p4) * 2 + (ks + q * 4 + rq) * EROW;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ad));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(ad + 16 * EROW));
                        const bf16x8 fr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        gram = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, fr, gram, 0, 0, 0);
                        ssum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fr, ssum, 0, 0, 0);
                    }
                    // D[row 4q + e][col r]: the diagonal entry of channel r sits in lane group q = r >> 2, register r & 3
                    const float d01 = (r & 1) ? gram[1] : gram[0], d23 = (r & 1) ? gram[3] : gram[2];
                    const float dg = (r & 2) ? d23 : d01;
                    float* dst = red + (wave * NB + n * 16 + r) * 2;
                    if (q == 0) dst[0] = ssum[0];
                    if (q == rq) dst[1] = dg;
                }
            }
        } else {
#pragma unroll
        for (int n = 0; n < NBF; ++n) {
            float t1[4] = {s1[n][0].x, s1[n][0].y, s1[n][1].x, s1[n][1].y};
            float t2[4] = {s2[n][0].x, s2[n][0].y, s2[n][1].x, s2[n][1].y};
            sum16_x4(t1[0], t1[1], t1[2], t1[3]);
            sum16_x4(t2[0], t2[1], t2[2], t2[3]);
            if (r == 0) {
                float* dst = red + (wave * NB + n * 16 + q * 4) * 2;
                *reinterpret_cast<float4*>(dst) = make_float4(t1[0], t2[0], t1[1], t2[1]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(t1[2], t2[2], t1[3], t2[3]);
            }
        }
        }
        __syncthreads();
        if (tid < NB * 2) {
            const float t = red[tid] + red[NB * 2 + tid] + red[2 * NB * 2 + tid] + red[3 * NB * 2 + tid];
            const long blk = (long)b * tiles_img + tile;
            a.stats[(blk * a.Cout + co0) * 2 + tid] = t;
        }
    }
#pragma unroll
    for (int it = 0; it < VPR; ++it) {
        const int idx = it * 64 + lane, p = idx / VPR, c = idx % VPR;
        const uint4 vv = *reinterpret_cast<const uint4*>(ew + p * EROW + c * 16);
        const int op = epix[p];
        if (op >= 0) *reinterpret_cast<uint4*>(y + (long)op * a.Cout + co0 + c * (16 / (int)sizeof(T))) = vv;
    }
    STAMP(8);
}


// Persistent single-chunk variant (Cin == CK: the HBM-bound early layers and their data gradients).
// grid = (G, Cout / NB); workgroup g walks the (image, tile) pairs t = g, g + G, ...  A wave64 VALU instruction
// occupies its SIMD for 4 cycles, and at one workgroup per tile the per-tile index arithmetic alone (~370 VALU per
// wave for 64 output pixels) was what held these layers at ~3.5 TB/s.  Here everything that does not depend on the
// tile is computed once per workgroup (fragment pixel decode, patch vector coordinates), the weight slab and the bias
// stay resident in LDS, the next tile's patch is prefetched into registers during the MFMAs, and the BatchNorm
// partial sums stay in registers across tiles: ONE statistics row per workgroup.
// (the BST form of the 32-channel kernel: held to three waves per SIMD - 168 registers, what its plain form has - it
// spills 52 bytes per lane; at two waves per SIMD it does not, and the 114x114 32->32 data gradient is 4 % faster)
// VAR 3 = BST + the weight gradient of the same conv (ConvArgs::wpart).  Both gradients of a conv read dy and the lower
// layer's y; as two kernels each tensor crossed the fabric twice (418 MB each in block 1).  Here the tile's x is formed
// in the BST epilogue from the y values it loads anyway, parked in LDS as [pixel][channel] (the transposing reads want
// the reduction axis - pixels - as rows), and dW[co][8 - t][ci] += sum_pixels dy_patch[pixel + t][co] * x[pixel][ci]
// over the 9 rotated taps t of the patch that the data gradient staged.
//   <16, 1>: one 16x16 tile per tap; the 4 waves split the pixels (each reduces the 64 pixels whose x it wrote itself:
//            no barrier), 36 accumulator registers, combined through LDS after the last tile
//   <32, 2>: four 16x16 tiles per tap, one (co fragment, ci fragment) role per wave over all 256 pixels (one more
//            barrier per tile: the x of the other waves)
template <typename T, int CK, int NBF, int VAR>
#ifndef VQA_BST_WAVES
#define VQA_BST_WAVES 2
#endif
#ifndef VQA_DW_WAVES
#define VQA_DW_WAVES 3
#endif
__global__ __launch_bounds__(256, ((VAR == 2 || VAR == 3) && NBF == 2) ? VQA_BST_WAVES : ((VAR == 3 && NBF == 1) ? VQA_DW_WAVES : 1)) void conv3x3_igemm_p(ConvArgs a) {
    constexpr bool XF = VAR == 1, BST = VAR == 2 || VAR == 3, WG = VAR == 3, POOL = VAR == 4;
    static_assert(!POOL || sizeof(T) == 2, "pooled epilogue: bf16");
    static_assert(!WG || (sizeof(T) == 2 && ((CK == 16 && NBF == 1) || (CK == 32 && NBF == 2))), "weight-gradient forms");
    constexpr bool WG_PSPLIT = WG && CK == 16;                  // waves split the pixels (else: the tiles)
    constexpr int XSTR = NBF == 1 ? 32 : NBF * 32 + 32;         // LDS bytes per x pixel (32 * odd: dy_stride)
    using Cfg = ConvCfg<T, CK>;
    constexpr int VEC = Cfg::VEC, KC = Cfg::KC, KP = Cfg::KP, NS = Cfg::NS, PXB = Cfg::PXB, WROWB = Cfg::WROWB,
                  VPP = Cfg::VPP;
    constexpr int NB = NBF * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PPIX = (a.R + 2) * PW, PROWB = PW * PXB + Cfg::RPAD;
    unsigned char* patch = smem;
    unsigned char* wl = smem + (a.R + 2) * PROWB;
    float* biasl = reinterpret_cast<float*>(wl + NB * WROWB);   // [NB]
    float* coefl = biasl + NB;                                  // XF: [2][CK] input scale / shift
    unsigned char* xt = reinterpret_cast<unsigned char*>(coefl + 4 * NB);   // WG: [256][XSTR] activations of the tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * NB;
    const int MT = a.R * a.CW;
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    const T* w = static_cast<const T*>(a.w);
    if ((int)blockIdx.x >= total) return;                       // (host launches G <= total)

    // ---- once per workgroup: fragment pixel decode (4 fragments of 16 consecutive tile pixels per wave);
    // (row, col) packed as row << 16 | col, rows of pixels beyond the tile are "never inside the image"
    int pbase[4], frc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int m = (wave * 4 + f) * 16 + r;
        int pr = 0, pc = 0;
        if (m < MT) { pr = fastdiv(m, a.CW, a.inv_cw); pc = m - pr * a.CW; }
        pbase[f] = pr * PROWB + pc * PXB;
        frc[f] = ((m < MT ? pr : 0x4000) << 16) | pc;
    }
    // ---- once per workgroup: coordinates of the patch vectors this thread stages, (py - pad) << 16 | (px - pad + 64)
    constexpr int PMAX = 384;                                   // host bound on (R+2)*(CW+2)
    constexpr int NPV = (PMAX * VPP + 255) / 256;
    constexpr int WV = KP / VEC;                                // vectors per weight row
    int prel[NPV], pyx[NPV], pdst[NPV];
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        const int i = tid + j * 256;
        const int px = i / VPP, vv = i % VPP;
        const int py = fastdiv(px, PW, a.inv_pw), pxx = px - py * PW;
        pyx[j] = ((i < PPIX * VPP ? py - a.pad : -0x4000) << 16) | (pxx - a.pad + 64);
        prel[j] = (((py - a.pad) * a.IW + (pxx - a.pad)) * a.Cin + vv * VEC) * (int)sizeof(T);   // bytes from the tile origin
        pdst[j] = py * PROWB + pxx * PXB + vv * 16;            // byte offset inside the LDS patch
    }
    const unsigned xbytes = (unsigned)a.B * a.IH * a.IW * a.Cin * (unsigned)sizeof(T);   // (host: both < 2^32)
    const unsigned ybytes = (unsigned)a.B * a.OH * a.OW * a.Cout * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xbuf = buffer_of(a.x, xbytes), ybuf = buffer_of(a.y, ybytes);
    // ---- once per workgroup: weight slab and bias into LDS
    {
        const T* wsrc = w + (long)co0 * KP;
        for (int i = tid; i < NB * WV; i += 256)
            *reinterpret_cast<uint4*>(wl + (i / WV) * WROWB + (i % WV) * 16) =
                *reinterpret_cast<const uint4*>(wsrc + (long)i * VEC);
        if (tid < NB) biasl[tid] = a.bias ? a.bias[co0 + tid] : 0.f;
        if constexpr (XF) {
            if (tid < CK) { coefl[tid] = a.in_scale[tid]; coefl[CK + tid] = a.in_shift[tid]; }
            __syncthreads();                                    // read by the first stash
        }
        if constexpr (BST) {                                    // [4][NB] (read in the epilogue, behind the loop's barrier)
            for (int i = tid; i < 4 * NB; i += 256) coefl[i] = a.bn_coef[(i / NB) * a.Cout + co0 + i % NB];
        }
    }
    const __amdgpu_buffer_rsrc_t bybuf = buffer_of(BST ? a.bn_y : a.y, ybytes);

    f32x4 acc[4][NBF];
    f32x2 s1[NBF][2], s2[NBF][2];
#pragma unroll
    for (int n = 0; n < NBF; ++n)
#pragma unroll
        for (int h = 0; h < 2; ++h) s1[n][h] = s2[n][h] = f32x2{0.f, 0.f};

    // ---- WG: this wave's tiles of the weight gradient and the (tile-independent) LDS offsets of its k rows.
    // k -> pixel map of a 32-pixel step as in conv3x3_wgrad: lane 16q + 4rq + p addresses pixel ks + 4q + rq (and + 16)
    constexpr int WSTEPS = WG_PSPLIT ? 2 : 8;                   // 32-pixel steps per wave and tile
    const int wcf = WG_PSPLIT ? 0 : (wave & 1), wnf = WG_PSPLIT ? 0 : (wave >> 1);   // role: dy / x channel fragment
    f32x4 wacc[WG ? 9 : 1];
    int* wptab = reinterpret_cast<int*>(xt + 256 * XSTR);       // [256] patch byte offset of tile pixel m
    if constexpr (WG) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int mm = tid < MT ? tid : 0;                  // x is zero beyond the tile: any finite dy
            const int pr = fastdiv(mm, a.CW, a.inv_cw);
            wptab[tid] = pr * PROWB + (mm - pr * a.CW) * PXB;   // (read behind the tile loop's first barrier)
        }
    }

    // ---- staging.  Out-of-image vectors are fetched at the buffer's size, which the hardware answers with zeros (the
    // zero padding costs one select per vector: no clamped address, no mask at the LDS write, no branch); the next
    // tile's loads are issued before the MFMA phase of this one and written to LDS after it (register prefetch,
    // single LDS buffer).
    uint4 pv[NPV];
    int fb = 0, foy0 = 0, fox0 = 0;                             // image / output origin of the fetched tile
    auto fetch = [&](int t) {
        int tl, tx;
        const int b = fastdivmod(t, tiles, a.inv_tiles, tl);
        const int ty = fastdivmod(tl, a.tiles_x, a.inv_tx, tx);
        const int oy0 = ty * a.R, ox0 = tx * a.CW;
        fb = b; foy0 = oy0; fox0 = ox0;
        const unsigned org = (unsigned)((b * a.IH + oy0) * a.IW + ox0) * a.Cin * (unsigned)sizeof(T);
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            const bool ok = (unsigned)(oy0 + (pyx[j] >> 16)) < (unsigned)a.IH &&
                            (unsigned)(ox0 - 64 + (pyx[j] & 0xFFFF)) < (unsigned)a.IW;
            pv[j] = buffer_load16(xbuf, ok ? org + (unsigned)prel[j] : xbytes);
        }
    };
    auto stash = [&]() {
        float isc[VEC], ish[VEC];                               // XF: this thread's channel vector (256 % VPP == 0)
        if constexpr (XF) {
            const float* cs = coefl + (tid % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                *reinterpret_cast<float4*>(isc + e) = *reinterpret_cast<const float4*>(cs + e);
                *reinterpret_cast<float4*>(ish + e) = *reinterpret_cast<const float4*>(cs + CK + e);
            }
        }
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            const int i = tid + j * 256;
            uint4 t = pv[j];   // by value first: a conditional store straight from the array keeps it in scratch
            if constexpr (XF) t = bn_relu_vec<T>(t, isc, ish);
            if (i < PPIX * VPP) *reinterpret_cast<uint4*>(patch + pdst[j]) = t;
        }
    };

    const UnitWalk uw = unit_walk(blockIdx.x, gridDim.x, total, a.rev);
    fetch(uw.unit(uw.first));
    for (int t = uw.first; t < uw.end; t += uw.step) {
        stash();
        __syncthreads();
        const int cb = fb, coy0 = foy0, cox0 = fox0;            // the tile being computed
        // BST: byte offsets of this lane's output pieces (also those of the lower layer's y), and the y loads - issued
        // BEFORE the next tile's patch loads so that the epilogue's wait does not include those
        unsigned yoff[BST ? 4 : 1][BST ? NBF : 1];
        Raw4<T> yv[BST ? 4 : 1][BST ? NBF : 1];
        if constexpr (BST) {
            const unsigned yt0 = ((unsigned)((cb * a.OH + coy0) * a.OW + cox0) * a.Cout + co0 + q * 4) * (unsigned)sizeof(T);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int pr = frc[f] >> 16, pc = frc[f] & 0xFFFF;
                const bool ok = coy0 + pr < a.OH && cox0 + pc < a.OW;
#pragma unroll
                for (int n = 0; n < NBF; ++n) {
                    yoff[f][n] = ok ? yt0 + (unsigned)(__mul24(__mul24(pr, a.OW) + pc, a.Cout) + n * 16) * (unsigned)sizeof(T)
                                    : ybytes;
                    yv[f][n] = buffer_load_raw4<T>(bybuf, yoff[f][n]);
                }
            }
        }
        if (t + uw.step < uw.end) fetch(uw.unit(t + uw.step));
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int n = 0; n < NBF; ++n) acc[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- MFMA over 9 taps x CK channels
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k0 = s * KC + q * VEC;
            int tap = k0 / CK;
            const int c = k0 % CK;
            tap = tap > 8 ? 8 : tap;  // padded k: weights are zero there, read any finite data
            const int toff = (tap / 3) * PROWB + (tap % 3) * PXB + c * (int)sizeof(T);
            uint4 fw[NBF], fp[4];
#pragma unroll
            for (int n = 0; n < NBF; ++n)
                fw[n] = *reinterpret_cast<const uint4*>(wl + (n * 16 + r) * WROWB + k0 * (int)sizeof(T));
#pragma unroll
            for (int f = 0; f < 4; ++f) fp[f] = *reinterpret_cast<const uint4*>(patch + pbase[f] + toff);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int n = 0; n < NBF; ++n) Mma<T>::mma(acc[f][n], fw[n], fp[f]);
        }
        // ---- epilogue: + bias, store (lane owns 4 consecutive channels of one pixel), BN partial statistics
        const unsigned yt = ((unsigned)((cb * a.OH + coy0) * a.OW + cox0) * a.Cout + co0 + q * 4) * (unsigned)sizeof(T);
        auto finish = [&](auto mode_tag) {
            constexpr bool RELU = (decltype(mode_tag)::value & 1) != 0, STATS = (decltype(mode_tag)::value & 2) != 0;
#pragma unroll
            for (int n = 0; n < NBF; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(biasl + n * 16 + q * 4);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int pr = frc[f] >> 16, pc = frc[f] & 0xFFFF;
                    const bool ok = coy0 + pr < a.OH && cox0 + pc < a.OW;
                    const Packed4<T> o = epilogue4<T, RELU, STATS>(acc[f][n], bv, ok ? 1.f : 0.f, s1[n], s2[n]);
                    buffer_store_packed4<T>(
                        ybuf, ok ? yt + (unsigned)(__mul24(__mul24(pr, a.OW) + pc, a.Cout) + n * 16) * (unsigned)sizeof(T)
                                 : ybytes, o);
                }
            }
        };
        if constexpr (BST) {
#pragma unroll
            for (int n = 0; n < NBF; ++n) {
                f32x4 cf[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) cf[k] = *reinterpret_cast<const f32x4*>(coefl + k * NB + n * 16 + q * 4);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const float okf = yoff[f][n] != ybytes ? 1.f : 0.f;
                    unsigned xa[2];
                    const Packed4<T> o = epilogue4_bnbwd<T>(acc[f][n], okf, yv[f][n], cf, s1[n], s2[n], WG ? xa : nullptr);
                    buffer_store_packed4<T>(ybuf, yoff[f][n], o);
                    if constexpr (WG)
                        *reinterpret_cast<uint2*>(xt + ((wave * 4 + f) * 16 + r) * XSTR + (n * 16 + q * 4) * 2) =
                            make_uint2(xa[0], xa[1]);
                }
            }
            if constexpr (WG) {
                if constexpr (WG_PSPLIT) asm volatile("" ::: "memory");   // own rows only: LDS runs a wave's ops in order
                else __syncthreads();
                typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4;
                // (recomputed per tile from an opaque copy of the thread id: held across the loop these three values
                // were what pushed the <16, 1> form over its 168 registers - and the spill reload sat between the y
                // loads and the next tile's prefetch, which then waited for both)
                int tl = tid;
                asm volatile("" : "+v"(tl));
                const int lr = tl & 15, lq = (tl >> 4) & 3;
                const int wm0 = (WG_PSPLIT ? (tl >> 6) * 64 : 0) + lq * 4 + (lr >> 2);     // this lane's k row of step 0
                const int wpc = (wcf * 16 + 4 * (lr & 3)) * 2;                             // its channel bytes in a patch pixel
                const unsigned char* wxb = xt + wm0 * XSTR + (wnf * 16 + 4 * (lr & 3)) * 2;   // its address in the x tile
#pragma unroll
                for (int st = 0; st < WSTEPS; ++st) {
                    if ((WG_PSPLIT ? wave * 64 : 0) + st * 32 >= MT) break;          // (uniform per wave)
                    const bf16x4 xl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(wxb + st * 32 * XSTR));
                    const bf16x4 xh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(wxb + (st * 32 + 16) * XSTR));
                    const bf16x8 fx = {xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
                    const unsigned char* d0 = patch + wptab[wm0 + st * 32] + wpc;
                    const unsigned char* d1 = patch + wptab[wm0 + st * 32 + 16] + wpc;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int toff = (tp / 3) * PROWB + (tp % 3) * PXB;
                        const bf16x4 dl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(d0 + toff));
                        const bf16x4 dh = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(d1 + toff));
                        const bf16x8 fd = {dl[0], dl[1], dl[2], dl[3], dh[0], dh[1], dh[2], dh[3]};
                        Mma<T>::mma(wacc[tp], __builtin_bit_cast(uint4, fd), __builtin_bit_cast(uint4, fx));
                    }
                }
            }
        } else if constexpr (POOL) {
            // ReLU + 2x2 max pool (inference, BatchNorm folded): with 16-column tiles fragment f of this wave IS tile row
            // 4 wave + f and lane r its column, so a window is fragments (2 f2, 2 f2 + 1) x lanes (r, r ^ 1).  The
            // values are rounded to bf16 first - what the separate pool pass read - and compared as signed 16-bit
            // integers (they are >= 0; a -0 loses to +0): the same bits as conv -> pool, a quarter of the stores.
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            auto pmax = [](unsigned u, unsigned v) {
                return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u),
                                                                              __builtin_bit_cast(s16x2, v)));
            };
            const int PHh = a.OH >> 1, PWh = a.OW >> 1;
            const unsigned pbytes = (unsigned)a.B * PHh * PWh * a.Cout * (unsigned)sizeof(T);
            const __amdgpu_buffer_rsrc_t pbuf = buffer_of(a.y, pbytes);
            // (tile origins are even: R is, and CW = 16)
            const unsigned pt = ((unsigned)((cb * PHh + (coy0 >> 1)) * PWh + (cox0 >> 1)) * a.Cout + co0 + q * 4) *
                                (unsigned)sizeof(T);
#pragma unroll
            for (int n = 0; n < NBF; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(biasl + n * 16 + q * 4);
#pragma unroll
                for (int f2 = 0; f2 < 2; ++f2) {
                    const Packed4<T> t0 = epilogue4<T, true, false>(acc[2 * f2][n], bv, 1.f, s1[n], s2[n]);
                    const Packed4<T> t1 = epilogue4<T, true, false>(acc[2 * f2 + 1][n], bv, 1.f, s1[n], s2[n]);
                    unsigned lo = pmax(t0.lo, t1.lo), hi = pmax(t0.hi, t1.hi);
                    // the window's other column: lane r ^ 1 (DPP quad_perm [1, 0, 3, 2])
                    lo = pmax(lo, (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, true));
                    hi = pmax(hi, (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, true));
                    const int pr = frc[2 * f2] >> 16, pr1 = frc[2 * f2 + 1] >> 16, pc = frc[2 * f2] & 0xFFFF;
                    const bool ok = (r & 1) == 0 && coy0 + pr1 < a.OH && cox0 + pc + 1 < a.OW;
                    const unsigned off = pt + (unsigned)(__mul24(__mul24(pr >> 1, PWh) + (pc >> 1), a.Cout) + n * 16) *
                                              (unsigned)sizeof(T);
                    Packed4<T> o;
                    o.lo = lo; o.hi = hi;
                    buffer_store_packed4<T>(pbuf, ok ? off : pbytes, o);
                }
            }
        } else {
            switch (((a.relu & 1) ? 1 : 0) | (a.stats ? 2 : 0)) {
                case 0: finish(std::integral_constant<int, 0>{}); break;
                case 1: finish(std::integral_constant<int, 1>{}); break;
                case 2: finish(std::integral_constant<int, 2>{}); break;
                default: finish(std::integral_constant<int, 3>{}); break;
            }
        }
        __syncthreads();
    }

    if (a.stats) {
        float* red = reinterpret_cast<float*>(smem);  // [4 waves][NB][2]; staging LDS is free after the last sync
#pragma unroll
        for (int n = 0; n < NBF; ++n) {
            float t1[4] = {s1[n][0].x, s1[n][0].y, s1[n][1].x, s1[n][1].y};
            float t2[4] = {s2[n][0].x, s2[n][0].y, s2[n][1].x, s2[n][1].y};
            sum16_x4(t1[0], t1[1], t1[2], t1[3]);
            sum16_x4(t2[0], t2[1], t2[2], t2[3]);
            if (r == 0) {
                float* dst = red + (wave * NB + n * 16 + q * 4) * 2;
                *reinterpret_cast<float4*>(dst) = make_float4(t1[0], t2[0], t1[1], t2[1]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(t1[2], t2[2], t1[3], t2[3]);
            }
        }
        __syncthreads();
        if (tid < NB * 2) {
            const float tt = red[tid] + red[NB * 2 + tid] + red[2 * NB * 2 + tid] + red[3 * NB * 2 + tid];
            a.stats[((long)blockIdx.x * a.Cout + co0) * 2 + tid] = tt;
        }
    }
    if constexpr (WG) {
        // D rows (4q + e) = dy channel, D cols (r) = x channel; tap t of the rotated patch is tap 8 - t of the conv
        float* slab = a.wpart + (long)blockIdx.x * a.Cin * 9 * a.Cout;
        if constexpr (WG_PSPLIT) {
            __syncthreads();                                    // (the statistics rows above)
            float* red = reinterpret_cast<float*>(smem);        // [4 waves][9][64 lanes][4]
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) *reinterpret_cast<f32x4*>(red + ((wave * 9 + tp) * 64 + lane) * 4) = wacc[tp];
            __syncthreads();
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {                    // thread = (lane tid / 4, element tid % 4) of tile tp
                const int i = tp * 256 + tid;
                const float v = ((red[i] + red[2304 + i]) + red[2 * 2304 + i]) + red[3 * 2304 + i];
                const int l = tid >> 2, e = tid & 3;
                const int cd = (l >> 4) * 4 + e, cx = co0 + (l & 15);
                slab[(cd * 9 + (8 - tp)) * a.Cout + cx] = v;
            }
        } else {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cd = wcf * 16 + q * 4 + e, cx = co0 + wnf * 16 + r;
                    slab[(cd * 9 + (8 - tp)) * a.Cout + cx] = wacc[tp][e];
                }
        }
    }
}

// LDS of either variant (the persistent one adds the resident bias row)
template <typename T, int CK, int NBF>
size_t igemm_lds(const ConvArgs& a) {
    using Cfg = ConvCfg<T, CK>;
    size_t lds = (size_t)(a.R + 2) * ((a.CW + 2) * Cfg::PXB + Cfg::RPAD) + (size_t)NBF * 16 * Cfg::WROWB;
    const size_t epi = (size_t)4 * 64 * (NBF * 16 * sizeof(T) + 16) + 4 * 64 * 4 +   // tile kernels' store transpose
                       (size_t)4 * NBF * 16 * 2 * 4;                                   // + their statistics rows
    if (lds < epi) lds = epi;
    lds += (size_t)NBF * 16 * 4;                                                       // resident bias row
    lds += (size_t)2 * a.Cin * 4 + (size_t)4 * NBF * 16 * 4;                           // XF / BST coefficient rows
    lds += 4096;                                                                       // tile kernel: stash dump slots
    return lds;
}

// plan_only: returns G = rows of the statistics buffer = workgroups per output-channel block.
//   single chunk (Cin == CK): persistent kernel, G = resident workgroups (occupancy x 256 CUs) / channel blocks
//   multi chunk:              one workgroup per (tile, image), G = tiles * B
template <typename T, int CK, int NBF>
int launch_igemm(hipStream_t s, const ConvArgs& a, int G, bool plan_only) {
    size_t lds = igemm_lds<T, CK, NBF>(a);
    const int total = a.tiles_x * a.tiles_y * a.B, zb = a.Cout / (NBF * 16);
    const bool persistent = a.Cin == CK;
    const int var = a.in_scale ? 1 : (a.bn_y ? (a.wpart ? 3 : 2) : 0);
    auto kp = var == 1 ? conv3x3_igemm_p<T, CK, NBF, 1> : (var >= 2 ? conv3x3_igemm_p<T, CK, NBF, 2> : conv3x3_igemm_p<T, CK, NBF, 0>);
    if (a.relu & 2) {                                           // (conv_fwd_entry: bf16, single chunk, plain variant)
        if constexpr (sizeof(T) == 2) kp = conv3x3_igemm_p<T, CK, NBF, 4>;
        else return plan_only ? -1 : VQA_ERR_ARG;
    }
    if (var == 3) {
        // the forms the kernel has (bf16, all of the conv's channels in one workgroup), + the x tile; the <16, 1>
        // form's combine buffer of the last tile is the floor
        constexpr bool have = sizeof(T) == 2 && ((CK == 16 && NBF == 1) || (CK == 32 && NBF == 2));
        if constexpr (have) {
            if (!persistent || zb != 1) return plan_only ? -1 : VQA_ERR_ARG;
            kp = conv3x3_igemm_p<T, CK, NBF, 3>;
            lds += 256 * (NBF == 1 ? 32 : NBF * 32 + 32) + 1024;
            if (CK == 16 && lds < 4 * 9 * 1024) lds = 4 * 9 * 1024;
        } else {
            return plan_only ? -1 : VQA_ERR_ARG;
        }
    }
    if (lds > 160 * 1024) return plan_only ? -1 : VQA_ERR_ARG;
    auto kt = var == 1 ? conv3x3_igemm_tile<T, CK, NBF, 1>
                       : (var >= 2 ? conv3x3_igemm_tile<T, CK, NBF, 2> : conv3x3_igemm_tile<T, CK, NBF, 0>);
    if constexpr (sizeof(T) == 2) {
        if (a.relu & 2) kt = conv3x3_igemm_tile<T, CK, NBF, 3>;
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(persistent ? (const void*)kp : (const void*)kt,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (plan_only) {
        if (!persistent) return total;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kp, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        // VQA_PLAN_RESERVE (read once): workgroup slots per CU left free for the kernels of the other streams, as long
        // as VQA_PLAN_RESERVE_KEEP (3) remain: 4 -> 3 costs the kernel nothing alone, 3 -> 2 costs it 4 % and gains the
        // step nothing more
        static const int reserve = [] { const char* e = getenv("VQA_PLAN_RESERVE"); return e ? atoi(e) : VQA_PLAN_RESERVE_DEFAULT; }();
        static const int keep = [] { const char* e = getenv("VQA_PLAN_RESERVE_KEEP"); return e ? atoi(e) : 3; }();
        if (per_cu - reserve >= keep) per_cu -= reserve;
        int g = (256 * per_cu) / zb;
        if (g < 1) g = 1;
        return g < total ? g : total;
    }
    if (persistent) {
        if (G < 1 || G > total) return VQA_ERR_ARG;
        if (a.OW >= (1 << 22) || a.Cout >= (1 << 22) || total >= (1 << 22)) return VQA_ERR_ARG;   // signed 24-bit index products
        hipLaunchKernelGGL(kp, dim3(G, zb), dim3(256), lds, s, a);
    } else {
        if (G != total) return VQA_ERR_ARG;
        if ((double)a.B * a.IH * a.IW * a.Cin * sizeof(T) >= 2147483648.0) return VQA_ERR_ARG;   // (offset sentinel 2^31)
        // the kernel's 24-bit index products (__mul24 is the SIGNED multiply: it sign-extends from bit 23, so every
        // operand must stay below 2^23, and fastdivmod's float quotient wants i < 2^22), its one-coefficient-per-thread
        // rows and its 16-byte bias loads
        if ((long)a.IH * a.IW >= (1 << 22) || (long)a.B * a.OH >= (1 << 22) || a.OW >= (1 << 22) || a.Cin > 256 ||
            (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0)
            return VQA_ERR_ARG;
        hipLaunchKernelGGL(kt, dim3(8 * ((total * zb + 7) / 8)), dim3(256), lds, s, a);
    }
    CHECK_LAUNCH();
    return VQA_OK;
}

template <typename T, int CK>
int dispatch_nb(hipStream_t s, const ConvArgs& a, int G, bool plan_only) {
    // 64-channel output blocks for the multi-chunk kernel (32-channel ones re-stage every patch twice: 56x56 64->64
    // 0.116 -> 0.150 ms); the single-chunk persistent kernel prefers 32 (165 registers = three waves per SIMD where the
    // 64-channel form has 238 = two: 58x58 32->64 0.083 -> 0.070 ms)
    if (a.Cout % 64 == 0 && a.Cin != CK) return launch_igemm<T, CK, 4>(s, a, G, plan_only);
    if (a.Cout % 32 == 0) return launch_igemm<T, CK, 2>(s, a, G, plan_only);
    if (a.Cout == 16) return launch_igemm<T, CK, 1>(s, a, G, plan_only);
    return plan_only ? -1 : VQA_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------- wgrad
struct WgradArgs {
    const void* x;   // [B][IH][IW][Cin]
    const void* dy;  // [B][OH][OW][Cout]
    float* part;     // [G][Cout][9][Cin] partial sums
    int B, IH, IW, Cin, OH, OW, Cout, pad;
    int R, CW, tiles_x, tiles_y;
    float inv_cw, inv_pw, inv_tx, inv_tiles;
    const float* in_scale;   // XF kernels: x is the previous conv's raw output, the input is relu(in_scale*x + in_shift)
    const float* in_shift;   // (see ConvArgs; pad == 0 only)
    int rev;                 // walk the pixel tiles back to front (unit_walk, common.h)
    int seam;                // two images per tile (see ConvArgs::seam): the real OH; B, IH, OH above are the pairs'
};

// One workgroup: NB = 64 output channels x CK input channels x 9 taps, looping over pixel tiles
// t = blockIdx.x, blockIdx.x + gridDim.x, ...  The 4 waves split as (4/nwk output-channel fragments) x (nwk slices
// of the pixel reduction), so layers with 16 or 32 output channels still use every wave; the slices are combined
// through LDS at the end and each workgroup writes one partial slab.
// LDS bytes per dy-tile pixel (COW channels): 32 * odd for the bf16 transposing reads; for f32 (lane group q reads
// pixel q, 16 lanes x 4 bytes each) 64 or 192 (mod 256) puts the 4 pixels of one read on disjoint banks.
__host__ __device__ constexpr int dy_stride(int esz, int cow) {
    return esz == 2 ? (cow == 16 ? 32 : cow * 2 + 32) : (cow == 16 ? 64 : cow * 4 + 64);
}

// Wave roles.  A workgroup owns COW output channels x CK input channels x 9 taps = (COW/16) x 9 x (CK/16) MFMA
// tiles.  Every wave takes ONE 16-input-channel fragment (wcf) and COF = min(2, COW/16) output-channel fragments
// (wcg), i.e. 9 x COF accumulator tiles; waves left over split the pixel reduction (wk).  Per 32-pixel step a wave
// then reads 9 patch fragments + COF dy fragments for 9 x COF MFMAs - the patch fragments, which dominate the LDS
// traffic, are no longer re-read by every output-channel wave.
// COFX = 4 (the MFMA-bound <32, 64> instantiations): FOUR output-channel fragments per wave - 36 accumulator tiles, the
// two wave pairs split the pixel reduction.  Per 32-pixel step a wave then reads 18 + 8 transposing 8-byte fragments for
// 36 MFMAs (0.72 LDS reads per MFMA) where the two-fragment form reads 18 + 4 for 18 (1.22): that form was bound by the
// LDS pipe (352 cycles against 288 of MFMA per step of a workgroup).  MEASURED SLOWER and off by default
// (VQA_WGRAD_COF4=1 turns it on): 144 accumulator + 216 other registers = 360 per lane, ONE wave per SIMD where the
// two-fragment form (224) has two, and with one wave nothing hides the LDS latency of the fragment reads -
// 56x56 64->64 0.088 -> 0.113 ms, 28x28 128->128 0.088 -> 0.114, 10x10 256->256 0.066 -> 0.085; step 4.056 -> 4.119 ms.
template <typename T, int CK, int COW, bool XF, int COFX = 0>
__global__ __launch_bounds__(256) void conv3x3_wgrad(WgradArgs a) {
    constexpr int VEC = Mma<T>::VEC, KC = 4 * VEC;
    constexpr int PXB = pix_stride(CK * (int)sizeof(T));
    constexpr int DYB = dy_stride((int)sizeof(T), COW);
    constexpr int VPP = CK * (int)sizeof(T) / 16;
    constexpr int CF = CK / 16;                         // input-channel fragments of the workgroup
    constexpr int COF = COFX ? COFX : (COW >= 32 ? 2 : 1);   // output-channel fragments per wave
    constexpr int NCG = COW / 16 / COF;                 // output-channel fragment groups
    constexpr int nwk = 4 / (CF * NCG);                 // pixel-reduction slices
    static_assert(CF * NCG * nwk == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int PW = a.CW + 2, PPIX = (a.R + 2) * PW, MT = a.R * a.CW;
    const int MTP = (MT + KC - 1) / KC * KC;
    unsigned char* patch = smem;
    unsigned char* dyl = smem + (PPIX * PXB + 15) / 16 * 16;
    int* ptab = reinterpret_cast<int*>(dyl + MTP * DYB);   // [MTP] byte offset of tile pixel m inside the patch
    float* coefl = reinterpret_cast<float*>(ptab + MTP);   // XF: [2][CK] input scale / shift of this workgroup's chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nco = a.Cout / COW;
    const int co0 = (blockIdx.y % nco) * COW, ci0 = (blockIdx.y / nco) * CK;
    const int wcf = wave % CF, wcg = (wave / CF) % NCG, wk = wave / (CF * NCG);
    const int tiles = a.tiles_x * a.tiles_y, total = tiles * a.B;
    if ((int)blockIdx.x >= total) return;

    f32x4 acc[9][COF];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COF; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- once per workgroup: what this thread stages per tile (tile-independent coordinates), pixel -> patch table
    constexpr int PMAX = sizeof(T) == 2 ? 384 : 320;               // host bound on (R+2)*(CW+2)
    constexpr int dvp = COW / VEC;                                 // 16-byte vectors per dy pixel
    constexpr int dsh = dvp == 16 ? 4 : (dvp == 8 ? 3 : (dvp == 4 ? 2 : 1));
    constexpr int NPV = (PMAX * VPP + 255) / 256;                  // 16-byte vectors per thread: patch,
    constexpr int NDV = (sizeof(T) == 2 ? 256 : 128) * dvp / 256;  // dy tile (MT <= 256 / 128 pixels)
    int prel[NPV], pyx[NPV], drel[NDV], drc[NDV];
#pragma unroll
    for (int j = 0; j < NPV; ++j) {
        const int i = tid + j * 256;
        const int px = i / VPP, vv = i % VPP;
        const int py = fastdiv(px, PW, a.inv_pw), pxx = px - py * PW;
        pyx[j] = ((i < PPIX * VPP ? py - a.pad : -0x4000) << 16) | (pxx - a.pad + 64);
        prel[j] = (((py - a.pad) * a.IW + (pxx - a.pad)) * a.Cin + ci0 + vv * VEC) * (int)sizeof(T);   // bytes
    }
#pragma unroll
    for (int j = 0; j < NDV; ++j) {
        const int i = tid + j * 256, m = i >> dsh, vv = i & (dvp - 1);
        const int pr = fastdiv(m, a.CW, a.inv_cw), pc = m - pr * a.CW;
        // (two images per tile: dy has no rows for the two output rows that straddle the images - they are fetched as
        // zeros - and the second image's rows sit two rows further up in memory than in the tile)
        const bool gap = a.seam && (unsigned)(pr - a.seam) < 2u;
        const int prm = pr - ((a.seam && pr >= a.seam + 2) ? 2 : 0);
        drc[j] = ((m < MT && !gap ? pr : 0x4000) << 16) | pc;
        drel[j] = ((prm * a.OW + pc) * a.Cout + co0 + vv * VEC) * (int)sizeof(T);                       // bytes
    }
    if constexpr (XF) {
        if (tid < CK) { coefl[tid] = a.in_scale[ci0 + tid]; coefl[CK + tid] = a.in_shift[ci0 + tid]; }
        __syncthreads();                                           // read by the first stash
    }
    for (int m = tid; m < MTP; m += 256) {
        const int mm = m < MT ? m : 0;                             // dy rows beyond the tile are zero: any finite x
        const int pr = fastdiv(mm, a.CW, a.inv_cw);
        ptab[m] = (pr * PW + (mm - pr * a.CW)) * PXB;
    }

    // ---- software pipeline over pixel tiles: the (branch-free) global loads of tile t+1 are issued into registers
    // before the MFMA loop of tile t and written to LDS after it
    // out-of-image patch vectors and dy rows beyond the tile / image are fetched at the buffers' sizes: the hardware
    // returns zeros for them (no clamped address, no mask at the LDS write)
    const unsigned xbytes = (unsigned)a.B * a.IH * a.IW * a.Cin * (unsigned)sizeof(T);      // (launcher: both < 4 GiB)
    const int ohm = a.seam ? 2 * a.seam : a.OH;                    // dy rows per image (pair) in memory
    const unsigned dbytes = (unsigned)a.B * ohm * a.OW * a.Cout * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xbuf = buffer_of(a.x, xbytes), dbuf = buffer_of(a.dy, dbytes);
    uint4 pv[NPV], dv[NDV];
    auto fetch = [&](int t) {
        const int b = fastdiv(t, tiles, a.inv_tiles), tl = t - b * tiles;
        const int tyy = fastdiv(tl, a.tiles_x, a.inv_tx);
        const int oy0 = tyy * a.R, ox0 = (tl - tyy * a.tiles_x) * a.CW;
        const unsigned xorg = (unsigned)((b * a.IH + oy0) * a.IW + ox0) * a.Cin * (unsigned)sizeof(T);
        const unsigned dorg = (unsigned)((b * ohm + oy0) * a.OW + ox0) * a.Cout * (unsigned)sizeof(T);
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            const bool ok = (unsigned)(oy0 + (pyx[j] >> 16)) < (unsigned)a.IH &&
                            (unsigned)(ox0 - 64 + (pyx[j] & 0xFFFF)) < (unsigned)a.IW;
            pv[j] = buffer_load16(xbuf, ok ? xorg + (unsigned)prel[j] : xbytes);
        }
#pragma unroll
        for (int j = 0; j < NDV; ++j) {
            const bool ok = oy0 + (drc[j] >> 16) < a.OH && ox0 + (drc[j] & 0xFFFF) < a.OW;
            dv[j] = buffer_load16(dbuf, ok ? dorg + (unsigned)drel[j] : dbytes);
        }
    };
    auto stash = [&]() {
        float isc[VEC], ish[VEC];                                  // XF: this thread's channel vector (256 % VPP == 0)
        if constexpr (XF) {
            const float* cs = coefl + (tid % VPP) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                *reinterpret_cast<float4*>(isc + e) = *reinterpret_cast<const float4*>(cs + e);
                *reinterpret_cast<float4*>(ish + e) = *reinterpret_cast<const float4*>(cs + CK + e);
            }
        }
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            const int i = tid + j * 256;
            uint4 t = pv[j];   // by value first: a conditional store straight from the array keeps it in scratch
            if constexpr (XF) t = bn_relu_vec<T>(t, isc, ish);
            if (i < PPIX * VPP) *reinterpret_cast<uint4*>(patch + (i / VPP) * PXB + (i % VPP) * 16) = t;
        }
#pragma unroll
        for (int j = 0; j < NDV; ++j) {
            const int i = tid + j * 256;
            const uint4 t = dv[j];
            if (i < (MTP << dsh)) *reinterpret_cast<uint4*>(dyl + (i >> dsh) * DYB + (i & (dvp - 1)) * 16) = t;
        }
    };
    const UnitWalk uw = unit_walk(blockIdx.x, gridDim.x, total, a.rev);
    fetch(uw.unit(uw.first));
    for (int t = uw.first; t < uw.end; t += uw.step) {
        stash();
        __syncthreads();
        if (t + uw.step < uw.end) fetch(uw.unit(t + uw.step));
        for (int ks = wk * KC; ks < MTP; ks += nwk * KC) {
            if constexpr (sizeof(T) == 2) {
                // bf16: lane i = 4*rq + p of a 16-lane group supplies the address of k-row rq, channels 4p..4p+3;
                // it receives channel i of 4 k-rows.  Two reads = the 8 k (pixels) of lane group q.
                // k -> pixel map of the chunk: lane group q, element 4h+e  <->  pixel ks + 16h + 4q + e, so one
                // read instruction's 32-lane half touches 8 CONSECUTIVE pixels (bank-conflict-free strides)
                const int rq = r >> 2, p = r & 3;
                const int m0 = ks + q * 4 + rq, m1 = m0 + 16;
                uint4 fa[COF];  // dy fragments: rows = co
#pragma unroll
                for (int c = 0; c < COF; ++c) {
                    const unsigned char* ad = dyl + ((wcg * COF + c) * 16 + 4 * p) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(ad + m0 * DYB));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(ad + m1 * DYB));
                    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fa[c] = __builtin_bit_cast(uint4, f);
                }
                const unsigned char* b0 = patch + ptab[m0] + 4 * p * 2 + wcf * 32;
                const unsigned char* b1 = patch + ptab[m1] + 4 * p * 2 + wcf * 32;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int toff = ((tp / 3) * PW + (tp % 3)) * PXB;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(b0 + toff));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(b1 + toff));
                    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int c = 0; c < COF; ++c) Mma<T>::mma(acc[tp][c], fa[c], __builtin_bit_cast(uint4, f));
                }
            } else {
                // f32: MFMA j of the chunk takes pixel k = ks + 4j + q
                float fa[COF][4];
                const unsigned char* bp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = ks + 4 * j + q;
#pragma unroll
                    for (int c = 0; c < COF; ++c)
                        fa[c][j] = *reinterpret_cast<const float*>(dyl + m * DYB + ((wcg * COF + c) * 16 + r) * 4);
                    bp[j] = patch + ptab[m] + r * 4 + wcf * 64;
                }
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int toff = ((tp / 3) * PW + (tp % 3)) * PXB;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float fb = *reinterpret_cast<const float*>(bp[j] + toff);
#pragma unroll
                        for (int c = 0; c < COF; ++c)
                            acc[tp][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][j], fb, acc[tp][c], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // combine the pixel slices of the block in LDS (fixed order), then one partial slab per workgroup
    if constexpr (nwk > 1) {
        float* red = reinterpret_cast<float*>(smem);   // staging LDS is free after the loop's last barrier
        const int wl = wave % (CF * NCG);              // this wave's role inside its slice
        if (wk > 0) {
            float* dst = red + ((long)((wk - 1) * (CF * NCG) + wl) * 9 * COF) * 256 + lane * 4;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int c = 0; c < COF; ++c)
                    *reinterpret_cast<f32x4*>(dst + (tp * COF + c) * 256) = acc[tp][c];
        }
        __syncthreads();
        if (wk == 0) {
            for (int k = 1; k < nwk; ++k) {
                const float* src = red + ((long)((k - 1) * (CF * NCG) + wl) * 9 * COF) * 256 + lane * 4;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int c = 0; c < COF; ++c) acc[tp][c] += *reinterpret_cast<const f32x4*>(src + (tp * COF + c) * 256);
            }
        }
    }
    // D rows = co (4q+e), D cols = ci (r)
    if (wk == 0) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int c = 0; c < COF; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = co0 + (wcg * COF + c) * 16 + q * 4 + e, ci = ci0 + wcf * 16 + r;
                    a.part[(((long)blockIdx.x * a.Cout + co) * 9 + tp) * a.Cin + ci] = acc[tp][c][e];
                }
    }
}

// dW[co][ci][tap] (torch layout, ci < Cin_real) = sum_g part[g][co][tap][ci], fixed summation order.
// 256 threads = (OPB/4 lanes, each 4 consecutive outputs = one 16-byte load per partial slab) x TPO partial-lanes;
// TPO grows as the output count shrinks so that small layers (2304 outputs, hundreds of partials) still fill the
// chip while every slab row is read in contiguous 16-byte pieces.
template <int TPO>
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce(const float* part, float* dw, int G, int Cout, int Cin,
                                                            int Cin_real) {
    constexpr int OL = 256 / TPO, OPB = OL * 4;     // output lanes, outputs per block
    __shared__ float4 red[256];
    const int n = Cout * 9 * Cin;                   // multiple of 16
    const int ol = threadIdx.x % OL, gl = threadIdx.x / OL;
    const int i = blockIdx.x * OPB + ol * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const float* p = part + i;
        int g = gl;
        for (; g + 3 * TPO < G; g += 4 * TPO) {     // 4 independent loads in flight
            const float4 a0 = *reinterpret_cast<const float4*>(p + (long)g * n);
            const float4 a1 = *reinterpret_cast<const float4*>(p + (long)(g + TPO) * n);
            const float4 a2 = *reinterpret_cast<const float4*>(p + (long)(g + 2 * TPO) * n);
            const float4 a3 = *reinterpret_cast<const float4*>(p + (long)(g + 3 * TPO) * n);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
            s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
            s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
            s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
        }
        for (; g < G; g += TPO) {
            const float4 a0 = *reinterpret_cast<const float4*>(p + (long)g * n);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (gl == 0 && i < n) {
        for (int k = 1; k < TPO; ++k) {
            const float4 t = red[k * OL + ol];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float v[4] = {s.x, s.y, s.z, s.w};
        const int ci0 = i % Cin, tp = (i / Cin) % 9, co = i / (9 * Cin);   // 4 consecutive ci of one (co, tap)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ci0 + e < Cin_real) dw[((long)co * Cin_real + ci0 + e) * 9 + tp] = v[e];
    }
}

// The same sums for SEVERAL layers in one launch (blockIdx.y = job): every layer's weight gradient leaves its partial
// slabs behind and one launch at the end of the backward reduces them all - nine dependent little launches (6-12 us
// each, 98 us per step) become one.  Partial lanes per output group (TPO) are a per-job run-time value here.
constexpr int RED_MAX_JOBS = 16;
struct ReduceJobs {
    const float* part[RED_MAX_JOBS];
    float* dw[RED_MAX_JOBS];
    int G[RED_MAX_JOBS], Cout[RED_MAX_JOBS], Cin[RED_MAX_JOBS], Cin_real[RED_MAX_JOBS], tpo[RED_MAX_JOBS];
};
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_all(ReduceJobs j) {
    __shared__ float4 red[256];
    const int b = blockIdx.y;
    const float* part = j.part[b];
    const int G = j.G[b], Cout = j.Cout[b], Cin = j.Cin[b], Cin_real = j.Cin_real[b], TPO = j.tpo[b];
    const int OL = 256 / TPO, OPB = OL * 4;
    const int n = Cout * 9 * Cin;
    if ((long)blockIdx.x * OPB >= n) return;                // (uniform: before any barrier)
    const int ol = threadIdx.x % OL, gl = threadIdx.x / OL;
    const int i = blockIdx.x * OPB + ol * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const float* p = part + i;
        int g = gl;
        for (; g + 3 * TPO < G; g += 4 * TPO) {             // 4 independent loads in flight
            const float4 a0 = *reinterpret_cast<const float4*>(p + (long)g * n);
            const float4 a1 = *reinterpret_cast<const float4*>(p + (long)(g + TPO) * n);
            const float4 a2 = *reinterpret_cast<const float4*>(p + (long)(g + 2 * TPO) * n);
            const float4 a3 = *reinterpret_cast<const float4*>(p + (long)(g + 3 * TPO) * n);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
            s.x += a1.x; s.y += a1.y; s.z += a1.z; s.w += a1.w;
            s.x += a2.x; s.y += a2.y; s.z += a2.z; s.w += a2.w;
            s.x += a3.x; s.y += a3.y; s.z += a3.z; s.w += a3.w;
        }
        for (; g < G; g += TPO) {
            const float4 a0 = *reinterpret_cast<const float4*>(p + (long)g * n);
            s.x += a0.x; s.y += a0.y; s.z += a0.z; s.w += a0.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (gl == 0 && i < n) {
        for (int k = 1; k < TPO; ++k) {
            const float4 t = red[k * OL + ol];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const float v[4] = {s.x, s.y, s.z, s.w};
        const int ci0 = i % Cin, tp = (i / Cin) % 9, co = i / (9 * Cin);
        float* dw = j.dw[b];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ci0 + e < Cin_real) dw[((long)co * Cin_real + ci0 + e) * 9 + tp] = v[e];
    }
}

// ---------------------------------------------------------------------------------------------- packing
// w [Cout][Cin][3][3] f32 -> fwd slab  [CinP/CK][Cout][KP],  k = tap*CK + c, ci = chunk*CK + c
//                         -> dgrad slab [Cout/CKd][CinOut][KPd], k = tap'*CKd + c, co = chunk*CKd + c,
//                            tap' = 8 - tap (180-degree rotation), output channel = ci
template <typename T>
__global__ void conv_pack(const float* w, T* out, int Cout, int Cin, int CinP, int CK, int KP, int dgrad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = dgrad ? CinP : Cout;       // output-channel rows of the slab
    const int red = dgrad ? Cout : CinP;        // reduction channels
    const long n = (long)(red / CK) * rows * KP;
    if (i >= n) return;
    const int k = i % KP;
    const int row = (i / KP) % rows;
    const int ch = i / ((long)KP * rows);
    float v = 0.f;
    if (k < 9 * CK) {
        const int tap = k / CK, c = ch * CK + k % CK;
        if (!dgrad) {
            if (c < Cin) v = w[((long)row * Cin + c) * 9 + tap];
        } else {
            if (row < Cin) v = w[((long)c * Cin + row) * 9 + (8 - tap)];
        }
    }
    out[i] = from_f32<T>(v);
}

// All conv layers' slabs (forward and data-gradient) in ONE launch: blockIdx.y = job.
constexpr int PACK_MAX_JOBS = 24;
struct PackJobs {
    const float* w[PACK_MAX_JOBS];
    const float* scale[PACK_MAX_JOBS];   // per output channel (forward slabs only) or null: W' = W * scale[co]
    void* out[PACK_MAX_JOBS];
    int Cout[PACK_MAX_JOBS], Cin[PACK_MAX_JOBS], CinP[PACK_MAX_JOBS], CK[PACK_MAX_JOBS], KP[PACK_MAX_JOBS],
        dgrad[PACK_MAX_JOBS];
};
template <typename T>
__global__ void conv_pack_all(PackJobs j) {
    const int b = blockIdx.y;
    const float* w = j.w[b];
    const float* scale = j.scale[b];
    T* out = static_cast<T*>(j.out[b]);
    const int Cout = j.Cout[b], Cin = j.Cin[b], CinP = j.CinP[b], CK = j.CK[b], KP = j.KP[b], dgrad = j.dgrad[b];
    const int rows = dgrad ? CinP : Cout, red = dgrad ? Cout : CinP;
    const long n = (long)(red / CK) * rows * KP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = i % KP;
        const int row = (i / KP) % rows;
        const int ch = i / ((long)KP * rows);
        float v = 0.f;
        if (k < 9 * CK) {
            const int tap = k / CK, c = ch * CK + k % CK;
            if (!dgrad) {
                if (c < Cin) v = w[((long)row * Cin + c) * 9 + tap];
                if (scale) v *= scale[row];
            } else {
                if (row < Cin) v = w[((long)c * Cin + row) * 9 + (8 - tap)];
            }
        }
        out[i] = from_f32<T>(v);
    }
}

// images [B][C][H][W] (f32 / f16 / bf16) -> NHWC [B][H][W][CP] in T, channels >= C zero.  One thread per pixel:
// coalesced plane reads (consecutive lanes = consecutive pixels), 16-byte stores of the CP-channel vector.
template <typename TI, typename T>
__global__ void nchw_to_nhwc(const TI* in, T* out, int B, int C, int H, int W, int CP) {
    constexpr int VEC = 16 / sizeof(T);
    const long n = (long)B * H * W, hw = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, p = i % hw;
        for (int c0 = 0; c0 < CP; c0 += VEC) {
            T v[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = c0 + j;
                v[j] = from_f32<T>(c < C ? to_f32(in[(b * C + c) * hw + p]) : 0.f);
            }
            *reinterpret_cast<uint4*>(out + i * CP + c0) = *reinterpret_cast<uint4*>(v);
        }
    }
}

}  // namespace

static int conv_fwd_entry(hipStream_t s, int dtype, const void* x, const void* wpacked, void* y, const float* bias,
                          float* stats, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int pad, int R, int CW,
                          int G, bool plan_only, int relu = 0, const float* in_scale = nullptr,
                          const float* in_shift = nullptr, const void* bn_y = nullptr, const float* bn_coef = nullptr,
                          float* wpart = nullptr) {
    const int bad = plan_only ? -1 : VQA_ERR_ARG;
    if (wpart && (!bn_y || pad != 2)) return bad;   // (pad 2 = the data gradient of a pad-0 conv: x has no zero border)
    if ((in_scale != nullptr) != (in_shift != nullptr) || (in_scale && pad != 0)) return bad;
    if ((bn_y != nullptr) != (bn_coef != nullptr) || (bn_y && (in_scale || relu || bias || (!plan_only && !stats))))
        return bad;
    if (OH != IH + 2 * pad - 2 || OW != IW + 2 * pad - 2 || R < 1 || CW < 1 || R * CW > 256 ||
        (R + 2) * (CW + 2) > 384 || B < 1)
        return bad;
    // ReLU + 2x2 max pool in the epilogue (relu == 3): the single-chunk bf16 kernel on 16-column tiles, whole windows
    if (relu & ~3) return bad;
    // (single chunk, Cin 16 / 32: 16-column tiles - a window is two fragments of a lane and its neighbour; multi chunk:
    // any even tile, the windows are read back from the epilogue's LDS transpose)
    if ((relu & 2) && (!(relu & 1) || dtype != VQA_BF16 || (R & 1) || (CW & 1) || (OH & 1) || (OW & 1) || stats ||
                       in_scale || bn_y || !((Cin == 16 || Cin == 32) ? CW == 16 : Cin % 32 == 0)))
        return bad;
    // the kernels address x and y through 32-bit byte offsets (buffer resources)
    const int esz_ = dtype == VQA_BF16 ? 2 : 4;
    if ((double)B * IH * IW * Cin * esz_ >= 4294967040.0 || (double)B * OH * OW * Cout * esz_ >= 4294967040.0) return bad;
    // two images per tile (ConvArgs::seam): the multi-chunk kernel's one-tile-per-image maps with room for a second
    // image in the tile's 256 pixels and the patch's 384 - the 10x10 map of the last conv (100 of 256 pixels: 7 of 16
    // fragment slots, and a 37 KB weight slab staged per chunk for them)
    static const int pair_on = [] { const char* e = getenv("VQA_CONV_PAIR"); return e ? atoi(e) : 1; }();
    int seam = 0;
    const int ck_ = dtype == VQA_BF16 ? 32 : 16;
    if (pair_on && !(relu & 2) && pad == 0 && !bn_y && Cin > ck_ && R >= OH && CW >= OW && B % 2 == 0 && (2 * OH + 2) * CW <= 256 &&
        (2 * IH) * (CW + 2) <= 384) {
        seam = OH;
        B /= 2; IH *= 2; OH = 2 * OH + 2; R = OH;
    }
    const int tx_ = (OW + CW - 1) / CW, ty_ = (OH + R - 1) / R;
    ConvArgs a{x, wpacked, y, bias, stats, B, IH, IW, Cin, OH, OW, Cout, pad, R, CW, tx_, ty_,
               1.0f / CW, 1.0f / (CW + 2), 1.0f / tx_, 1.0f / (tx_ * ty_), relu, in_scale, in_shift, bn_y, bn_coef,
               vqa_walk_rev(in_scale ? 2 : (bn_y ? 4 : 1)), wpart, seam};
    if (dtype == VQA_BF16) {
        if (Cin % 32 == 0) return dispatch_nb<bf16, 32>(s, a, G, plan_only);
        if (Cin == 16) return dispatch_nb<bf16, 16>(s, a, G, plan_only);
    } else if (dtype == VQA_F32) {
        if (Cin % 16 == 0) return dispatch_nb<float, 16>(s, a, G, plan_only);
    }
    return bad;
}

extern "C" int vqa_conv3x3_fwd(void* stream, int dtype, const void* x, const void* wpacked, void* y,
                               const float* bias, float* stats, int B, int IH, int IW, int Cin, int OH, int OW,
                               int Cout, int pad, int R, int CW, int G, int relu, const float* in_scale,
                               const float* in_shift, const void* bn_y, const float* bn_coef) {
    return conv_fwd_entry(static_cast<hipStream_t>(stream), dtype, x, wpacked, y, bias, stats, B, IH, IW, Cin, OH, OW,
                          Cout, pad, R, CW, G, false, relu, in_scale, in_shift, bn_y, bn_coef);
}

// G for vqa_conv3x3_fwd: persistent workgroups per output-channel block = rows of `stats`; -1 = unsupported shape
#ifdef VQA_STAMPS
extern "C" int dbg_conv_stamps(void* out) {     // host buffer of 16384 * 12 u64
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int vqa_conv3x3_plan(int dtype, int B, int IH, int IW, int Cin, int OH, int OW, int Cout, int pad, int R,
                                int CW, int variant) {
    // the variants are different instantiations with their own register counts, hence their own resident grids
    static const float dummy = 0.f;
    const float* xf = variant == 1 ? &dummy : nullptr;
    const float* bst = variant >= 2 ? &dummy : nullptr;
    if (variant < 0 || variant > 4) return -1;       // (4: the pooled ReLU epilogue, vqa_conv3x3_fwd with relu = 3)
    if (variant == 4)
        return conv_fwd_entry(nullptr, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, B, IH, IW, Cin, OH, OW, Cout,
                              pad, R, CW, 0, true, 3);
    return conv_fwd_entry(nullptr, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, B, IH, IW, Cin, OH, OW, Cout,
                          pad, R, CW, 0, true, 0, xf, xf, bst, bst, variant == 3 ? const_cast<float*>(&dummy) : nullptr);
}

// The data gradient of a pad-0 conv (vqa_conv3x3_fwd with bn_y / bn_coef, pad = 2) that ALSO leaves that conv's weight
// gradient behind: wpart [G][Cin][9][Cout] f32 partial slabs (Cin = dy's channels = the conv's output channels), to
// be summed by vqa_conv3x3_wgrad_reduce_all (desc: G, Cin, Cout, real input channels).  G from vqa_conv3x3_plan variant 3.
extern "C" int vqa_conv3x3_dgrad_wgrad(void* stream, int dtype, const void* dy, const void* wpacked, void* dx,
                                       float* stats, float* wpart, int B, int IH, int IW, int Cin, int OH, int OW,
                                       int Cout, int pad, int R, int CW, int G, const void* bn_y, const float* bn_coef) {
    if (!wpart) return VQA_ERR_ARG;
    return conv_fwd_entry(static_cast<hipStream_t>(stream), dtype, dy, wpacked, dx, nullptr, stats, B, IH, IW, Cin, OH,
                          OW, Cout, pad, R, CW, G, false, 0, nullptr, nullptr, bn_y, bn_coef, wpart);
}

extern "C" int vqa_conv3x3_wgrad(void* stream, int dtype, const void* x, const void* dy, float* part, float* dw,
                                 int G, int B, int IH, int IW, int Cin, int Cin_real, int OH, int OW, int Cout,
                                 int pad, int R, int CW, const float* in_scale, const float* in_shift, int seam) {
    if (OH != IH + 2 * pad - 2 || OW != IW + 2 * pad - 2 || R < 1 || CW < 1 || G < 1) return VQA_ERR_ARG;
    // two images per tile: B, IH, OH are those of the pairs; one tile per pair, no zero border between the images
    if (seam < 0 || (seam && (pad != 0 || OH != 2 * seam + 2 || R < OH || CW < OW))) return VQA_ERR_ARG;
    if ((in_scale != nullptr) != (in_shift != nullptr) || (in_scale && pad != 0)) return VQA_ERR_ARG;
    const bool xf = in_scale != nullptr;
    {   // x and dy are addressed through 32-bit byte offsets (buffer resources)
        const int esz_ = dtype == VQA_BF16 ? 2 : 4;
        if ((double)B * IH * IW * Cin * esz_ >= 4294967040.0 || (double)B * OH * OW * Cout * esz_ >= 4294967040.0)
            return VQA_ERR_ARG;
    }
    const int tx_ = (OW + CW - 1) / CW, ty_ = (OH + R - 1) / R;
    const int COW = Cout >= 64 ? 64 : Cout;      // output channels per workgroup
    if ((COW != 16 && COW != 32 && COW != 64) || Cout % COW) return VQA_ERR_ARG;
    // fastdiv (common.h): float quotient + signed 24-bit products - tile numbers below 2^22
    if ((long)B * tx_ * ty_ >= (1 << 22)) return VQA_ERR_ARG;
    WgradArgs a{x, dy, part, B, IH, IW, Cin, OH, OW, Cout, pad, R, CW, tx_, ty_,
                1.0f / CW, 1.0f / (CW + 2), 1.0f / tx_, 1.0f / (tx_ * ty_), in_scale, in_shift, vqa_walk_rev(8), seam};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int MT = R * CW, PPIX = (R + 2) * (CW + 2);
    int status = VQA_ERR_ARG;
    auto go = [&](auto kern, int CK, int esz, int cofx = 0) {
        const int KC = 64 / esz;
        const int MTP = (MT + KC - 1) / KC * KC;
        size_t lds = (size_t)(PPIX * pix_stride(CK * esz) + 15) / 16 * 16 + (size_t)MTP * dy_stride(esz, COW) +
                     (size_t)MTP * 4 + (size_t)2 * CK * 4;                         // + pixel -> patch table, XF coeffs
        const int cof = cofx ? cofx : (COW >= 32 ? 2 : 1), roles = (CK / 16) * (COW / 16 / cof), nwk = 4 / roles;
        const size_t red = (size_t)(nwk - 1) * roles * 9 * cof * 1024;              // cross-slice combine buffer
        if (lds < red) lds = red;
        if (lds > 160 * 1024) return VQA_ERR_ARG;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        dim3 grid(G, (Cout / COW) * (Cin / CK));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError() == hipSuccess ? VQA_OK : VQA_ERR_LAUNCH;
    };
    if (dtype == VQA_BF16) {
        if (MT > 256 || PPIX > 384) return VQA_ERR_ARG;
#define GO_XF(T, CK, COW_, ESZ) \
    (xf ? go(conv3x3_wgrad<T, CK, COW_, true>, CK, ESZ) : go(conv3x3_wgrad<T, CK, COW_, false>, CK, ESZ))
#define GO_COW(T, CK, ESZ) \
    (COW == 64 ? GO_XF(T, CK, 64, ESZ) : (COW == 32 ? GO_XF(T, CK, 32, ESZ) : GO_XF(T, CK, 16, ESZ)))
        // VQA_WGRAD_COF4 (read once; default off: slower, see the kernel): the four-fragment form of <32, 64>
        static const bool cof4 = [] { const char* e = getenv("VQA_WGRAD_COF4"); return e ? atoi(e) != 0 : false; }();
        if (Cin % 32 == 0 && COW == 64 && cof4)
            status = xf ? go(conv3x3_wgrad<bf16, 32, 64, true, 4>, 32, 2, 4) : go(conv3x3_wgrad<bf16, 32, 64, false, 4>, 32, 2, 4);
        else if (Cin % 32 == 0) status = GO_COW(bf16, 32, 2);
        else if (Cin == 16) status = GO_COW(bf16, 16, 2);
    } else if (dtype == VQA_F32) {
        if (MT > 128 || PPIX > 320) return VQA_ERR_ARG;
        if (Cin % 16 == 0) status = GO_COW(float, 16, 4);
#undef GO_COW
#undef GO_XF
    }
    if (status != VQA_OK) return status;
    if (!dw) return VQA_OK;                     // the caller reduces the slabs later (vqa_conv3x3_wgrad_reduce_all)
    const int n = Cout * 9 * Cin;
    if (n >= 524288 || G <= 4)
        hipLaunchKernelGGL(conv3x3_wgrad_reduce<4>, dim3((n + 255) / 256), dim3(256), 0, s, part, dw, G, Cout, Cin, Cin_real);
    else if (n >= 65536 || G <= 16)
        hipLaunchKernelGGL(conv3x3_wgrad_reduce<16>, dim3((n + 63) / 64), dim3(256), 0, s, part, dw, G, Cout, Cin, Cin_real);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_reduce<64>, dim3((n + 15) / 16), dim3(256), 0, s, part, dw, G, Cout, Cin, Cin_real);
    CHECK_LAUNCH();
    return VQA_OK;
}

// n jobs; part / dw are HOST arrays of device pointers, desc a host array of n x 4 ints (G, Cout, Cin, Cin_real): the
// reductions n vqa_conv3x3_wgrad calls with dw == NULL left undone, in one launch (same sums, same order per output).
extern "C" int vqa_conv3x3_wgrad_reduce_all(void* stream, int n, const void* const* part, void* const* dw,
                                            const int* desc) {
    if (n < 1 || n > RED_MAX_JOBS || !part || !dw || !desc) return VQA_ERR_ARG;
    ReduceJobs j;
    int gx = 1;
    for (int i = 0; i < n; ++i) {
        const int* d = desc + 4 * i;
        if (!part[i] || !dw[i] || d[0] < 1 || d[1] < 1 || d[2] < 1 || d[3] < 1 || d[3] > d[2] || d[2] % 4) return VQA_ERR_ARG;
        j.part[i] = static_cast<const float*>(part[i]);
        j.dw[i] = static_cast<float*>(dw[i]);
        j.G[i] = d[0]; j.Cout[i] = d[1]; j.Cin[i] = d[2]; j.Cin_real[i] = d[3];
        const int nn = d[1] * 9 * d[2];
        j.tpo[i] = (nn >= 524288 || d[0] <= 4) ? 4 : ((nn >= 65536 || d[0] <= 16) ? 16 : 64);   // (as the single form)
        const int opb = 256 / j.tpo[i] * 4, blocks = (nn + opb - 1) / opb;
        if (blocks > gx) gx = blocks;
    }
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_all, dim3(gx, n), dim3(256), 0, static_cast<hipStream_t>(stream), j);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_conv_pack(void* stream, int dtype, const float* w, void* out, int Cout, int Cin, int CinP,
                             int CK, int KP, int dgrad) {
    const int rows = dgrad ? CinP : Cout, red = dgrad ? Cout : CinP;
    if (red % CK) return VQA_ERR_ARG;
    const long n = (long)(red / CK) * rows * KP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == VQA_BF16)
        hipLaunchKernelGGL(conv_pack<bf16>, grid, dim3(256), 0, s, w, static_cast<bf16*>(out), Cout, Cin, CinP, CK, KP, dgrad);
    else if (dtype == VQA_F32)
        hipLaunchKernelGGL(conv_pack<float>, grid, dim3(256), 0, s, w, static_cast<float*>(out), Cout, Cin, CinP, CK, KP, dgrad);
    else
        return VQA_ERR_ARG;
    CHECK_LAUNCH();
    return VQA_OK;
}

// n jobs; w / out are HOST arrays of device pointers, desc is a host array of n x 6 ints
// (Cout, Cin, CinP, CK, KP, dgrad): the arguments of n vqa_conv_pack calls, done by one launch.  scale (may be NULL, or
// hold NULLs) = per-output-channel factors folded into the forward slabs: eval-mode BatchNorm, W' = W * gamma/sigma.
extern "C" int vqa_conv_pack_all(void* stream, int dtype, int n, const void* const* w, void* const* out,
                                 const int* desc, const void* const* scale) {
    if (n < 1 || n > PACK_MAX_JOBS || !w || !out || !desc) return VQA_ERR_ARG;
    PackJobs j;
    long nmax = 0;
    for (int i = 0; i < n; ++i) {
        const int* d = desc + 6 * i;
        const int rows = d[5] ? d[2] : d[0], red = d[5] ? d[0] : d[2];
        if (d[3] < 1 || red % d[3]) return VQA_ERR_ARG;
        j.w[i] = static_cast<const float*>(w[i]);
        j.scale[i] = scale ? static_cast<const float*>(scale[i]) : nullptr;
        if (j.scale[i] && d[5]) return VQA_ERR_ARG;                 // folding applies to forward slabs
        j.out[i] = out[i];
        j.Cout[i] = d[0]; j.Cin[i] = d[1]; j.CinP[i] = d[2]; j.CK[i] = d[3]; j.KP[i] = d[4]; j.dgrad[i] = d[5];
        const long ne = (long)(red / d[3]) * rows * d[4];
        if (ne > nmax) nmax = ne;
    }
    long gb = (nmax + 255) / 256;
    dim3 grid((unsigned)(gb > 256 ? 256 : gb), n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == VQA_BF16) hipLaunchKernelGGL(conv_pack_all<bf16>, grid, dim3(256), 0, s, j);
    else if (dtype == VQA_F32) hipLaunchKernelGGL(conv_pack_all<float>, grid, dim3(256), 0, s, j);
    else return VQA_ERR_ARG;
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_nchw_to_nhwc(void* stream, int in_dtype, int out_dtype, const void* in, void* out, int B, int C,
                                int H, int W, int CP) {
    const long n = (long)B * H * W;
    if (CP % 8) return VQA_ERR_ARG;
    long gb = (n + 255) / 256;
    dim3 grid((unsigned)(gb > 8192 ? 8192 : gb));
    hipStream_t s = static_cast<hipStream_t>(stream);
#define GO(TI, TO) hipLaunchKernelGGL((nchw_to_nhwc<TI, TO>), grid, dim3(256), 0, s, static_cast<const TI*>(in), \
                                      static_cast<TO*>(out), B, C, H, W, CP)
    if (out_dtype == VQA_F32) {
        if (in_dtype == VQA_F32) GO(float, float);
        else if (in_dtype == VQA_F16) GO(_Float16, float);
        else if (in_dtype == VQA_BF16) GO(bf16, float);
        else return VQA_ERR_ARG;
    } else if (out_dtype == VQA_BF16) {
        if (in_dtype == VQA_F32) GO(float, bf16);
        else if (in_dtype == VQA_F16) GO(_Float16, bf16);
        else if (in_dtype == VQA_BF16) GO(bf16, bf16);
        else return VQA_ERR_ARG;
    } else {
        return VQA_ERR_ARG;
    }
#undef GO
    CHECK_LAUNCH();
    return VQA_OK;
}
