// LDS-tiled MFMA GEMM for the GRU / fusion-head / classifier layers.
//   NT:  C[M,N] = A[M,K] * B[N,K]^T      (forward linears:  x * W^T;  input grads with a pre-transposed W)
//   TN:  C[M,N] = A[K,M]^T * B[K,N]      (weight grads:  dY^T * X, reduction over the batch dimension)
// Epilogue (fused): + bias[n], + rowadd[m / group][n], ReLU, accumulate into C, optional second bf16 copy.
// 64x64 tile, 4 waves (2x2), each wave 2x2 MFMA tiles; operands staged through LDS k-contiguous so that both
// modes feed the MFMA with one ds_read_b128 per lane per fragment.  The MFMA's D rows are the N index, so a lane
// owns 4 consecutive n of one m: 16-byte stores.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64;
constexpr int ROWB = 96;  // LDS row: 64 bytes of k + 32 pad: stride of 6 sixteen-byte slots is conflict-free for the
                          // ds_read_b128 fragment reads (lane 16q+r reads slot 6r+q; brute-forced over the lane groups)

struct GemmArgs {
    int M, N, K;
    const void* A; int lda;
    const void* B; int ldb;
    void* C; int ldc;
    const float* bias;
    const float* rowadd; int rowadd_group;
    int act, accumulate;
    bf16* C2; int ldc2;
    float* ws; int kper;   // split-K: grid.z slices of `kper` k each write raw sums to ws[z][M][N]
    // act == 2: the classifier GEMM's sigmoid-BCE epilogue (vqa_model.py:103 + :215).  C = logits as usual; besides,
    // dlogits[m][n] = sigmoid(x) - t and one partial of sum max(x,0) - x t + log1p(exp(-|x|)) per workgroup
    // (loss_part[blockIdx.y * gridDim.x + blockIdx.x], summed in fixed order by the caller).  target may be null
    // (t = 0: sparse targets, whose few (label, score) terms are applied afterwards).
    const float* target; int ldt;
    float* dlogits; int ldd;
    double* loss_part;
    // epilogue multiplier emul [M][ldm] f32 (the head's elementwise kernels between its M = 256 GEMMs are launch latency
    // on the one phase of the step where nothing else runs: ~9 us each).  emode 1: the bf16 copy C2 is x * emul while C
    // stays x (vqa_model.py:97-99: C = relu(W_q q + b), C2 = its Hadamard product with the image projection);
    // emode 16 + act: C = x * act'(emul), act' from the activation's OUTPUT (the backward of vqa_model.py:101: the
    // gradient w.r.t. relu's input from the GEMM that forms the gradient w.r.t. its output).
    // emode 32 + act (emul2 required, same ldm): the backward of the Hadamard product of two activated projections
    // (vqa_model.py:95-99) in the GEMM that forms its incoming gradient x: C = x * emul2 * act'(emul),
    // C2 = bf16(x * emul * act'(emul2)) - vqa_hadamard_bwd without a mask (emul = fv, emul2 = fq: C = dpre_v, C2 = dpre_q).
    const float* emul; int ldm; int emode;
    const float* emul2;
};

template <typename T>
__device__ __forceinline__ uint4 load_row_chunk(const T* base, long row, int ld, int k, int K, bool row_ok) {
    // 16 bytes of k starting at k (k-contiguous source); zero outside the matrix.
    constexpr int VEC = Mma<T>::VEC;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!row_ok) return v;
    const T* p = base + row * ld + k;
    if (k + VEC <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        v = *reinterpret_cast<const uint4*>(p);
    } else {
        T tmp[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) tmp[j] = (k + j < K) ? p[j] : from_f32<T>(0.f);
        v = *reinterpret_cast<uint4*>(tmp);
    }
    return v;
}


// bias / row-broadcast add / ReLU / accumulate, then a 4-wide store of C (and the optional bf16 copy C2)
// RICH: the SELU (act 3) and sigmoid-BCE (act 2) epilogues are compiled into their own instantiations - carried by
// every kernel they cost the 128 x 128 tile 272 bytes of scratch per lane (3072 x 1024 x 3584 TN: 76 -> 150 us)
template <typename TC, bool RICH = true>
__device__ __forceinline__ float epilogue_store(const GemmArgs& g, TC* C, int m, int n, const f32x4& a, bool vec_ok) {
    float v[4], loss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = a[e];
        const int ne = n + e;
        if (ne < g.N) {
            if (g.bias) x += g.bias[ne];
            if (g.rowadd) x += g.rowadd[(long)(m / g.rowadd_group) * g.N + ne];
            if (RICH) { if (g.act == 1 || g.act == 3) x = act_apply(x, g.act); }
            else if (g.act == 1) x = fmaxf(x, 0.f);
            if (g.accumulate) x += to_f32(C[(long)m * g.ldc + ne]);
            if (g.emode >= 16 && g.emode < 32) x *= act_grad(g.emul[(long)m * g.ldm + ne], g.emode - 16);
            if (RICH && g.act == 2) {              // the arithmetic of bce_partial (misc.hip), element for element
                const float t = g.target ? g.target[(long)m * g.ldt + ne] : 0.f;
                const float ex = __expf(-fabsf(x));
                loss += fmaxf(x, 0.f) - x * t + log1pf(ex);
                g.dlogits[(long)m * g.ldd + ne] = (x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex)) - t;
            }
        }
        v[e] = x;
    }
    float v2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v2[e] = v[e];
        if (g.emode == 1 && n + e < g.N) v2[e] = v[e] * g.emul[(long)m * g.ldm + n + e];
        if (g.emode >= 32 && n + e < g.N) {
            const float e1 = g.emul[(long)m * g.ldm + n + e], e2 = g.emul2[(long)m * g.ldm + n + e];
            v2[e] = v[e] * e1 * act_grad(e2, g.emode - 32);
            v[e] = v[e] * e2 * act_grad(e1, g.emode - 32);
        }
    }
    if (vec_ok && n + 3 < g.N) {
        store_vec4<TC>(C + (long)m * g.ldc + n, v);
        if (g.C2) store_vec4<bf16>(g.C2 + (long)m * g.ldc2 + n, v2);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < g.N) {
                C[(long)m * g.ldc + n + e] = from_f32<TC>(v[e]);
                if (g.C2) g.C2[(long)m * g.ldc2 + n + e] = (bf16)v2[e];
            }
    }
    return loss;
}

// act == 2: the workgroup's loss partial (fixed order: lanes by shuffle, the four waves in order); red = 4 doubles of LDS
__device__ __forceinline__ void bce_block_partial(const GemmArgs& g, float lsum, double* red) {
    double s = wave_sum_d((double)lsum);
    __syncthreads();                               // (the staging LDS may still be read by a slower wave)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g.loss_part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T, typename TC, bool TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int VEC = Mma<T>::VEC;
    constexpr int KC = 4 * VEC;  // k per stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * ROWB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 64 * ROWB;
    const T* A = static_cast<const T*>(g.A);
    const T* B = static_cast<const T*>(g.B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 15, q = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kbeg = g.ws ? blockIdx.z * g.kper : 0;
    const int kend = g.ws ? (kbeg + g.kper < g.K ? kbeg + g.kper : g.K) : g.K;
    for (int k0 = kbeg; k0 < kend; k0 += KC) {
        if (!TN) {
            const int row = tid >> 2, vc = tid & 3;
            uint4 va = load_row_chunk<T>(A, m0 + row, g.lda, k0 + vc * VEC, g.K, m0 + row < g.M);
            uint4 vb = load_row_chunk<T>(B, n0 + row, g.ldb, k0 + vc * VEC, g.K, n0 + row < g.N);
            *reinterpret_cast<uint4*>(As + row * ROWB + vc * 16) = va;
            *reinterpret_cast<uint4*>(Bs + row * ROWB + vc * 16) = vb;
        } else {
            // source rows are k, contiguous along m / n: load 16 B along m, scatter transposed into LDS
            constexpr int VPR = 64 / VEC;  // vectors per k-row of the tile
            const int kk = tid / VPR, mv = tid % VPR;
            const bool k_ok = k0 + kk < g.K;
            uint4 va = load_row_chunk<T>(A, k0 + kk, g.lda, m0 + mv * VEC, g.M, k_ok);
            uint4 vb = load_row_chunk<T>(B, k0 + kk, g.ldb, n0 + mv * VEC, g.N, k_ok);
            const T* ea = reinterpret_cast<const T*>(&va);
            const T* eb = reinterpret_cast<const T*>(&vb);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                *reinterpret_cast<T*>(As + (mv * VEC + j) * ROWB + kk * sizeof(T)) = ea[j];
                *reinterpret_cast<T*>(Bs + (mv * VEC + j) * ROWB + kk * sizeof(T)) = eb[j];
            }
        }
        __syncthreads();
        uint4 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = *reinterpret_cast<const uint4*>(As + (wm * 32 + i * 16 + r) * ROWB + q * 16);
            fb[i] = *reinterpret_cast<const uint4*>(Bs + (wn * 32 + i * 16 + r) * ROWB + q * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);  // D rows = n, D cols = m
        __syncthreads();
    }

    if (g.ws) {   // split-K slice: raw partial sums, epilogue applied by gemm_splitk_reduce
        float* W = g.ws + (long)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 32 + i * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wn * 32 + j * 16 + q * 4 + e;
                    if (n < g.N) W[(long)m * g.N + n] = acc[i][j][e];
                }
        }
        return;
    }
    TC* C = static_cast<TC*>(g.C);
    const bool vec_ok = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (g.C2 == nullptr || g.ldc2 % 4 == 0);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + q * 4;
            if (n >= g.N) continue;
            lsum += epilogue_store<TC>(g, C, m, n, acc[i][j], vec_ok);
        }
    }
    if (g.act == 2) bce_block_partial(g, lsum, reinterpret_cast<double*>(lds));
}


// ---- bf16 pipelined kernel -------------------------------------------------------------------------------------------
// Same contract as gemm_kernel, bf16 operands only.  Differences: BK = 64 or 128 k per stage, two LDS buffers and a
// register-prefetch pipeline (global loads of stage s+1 are in flight during the MFMAs of stage s; ONE barrier per
// stage), 64x64 or 128x128 tiles, and the TN mode keeps the source's k-major rows in LDS and builds fragments with
// the transposing read ds_read_b64_tr_b16 instead of scattering 2-byte LDS writes.
//   NT LDS row = one m (n), BK k's:      BK*2 + 32 bytes -> 10 / 18 sixteen-byte slots (== 2 mod 4: conflict-free b128)
//   TN LDS row = one k, BM (BN) m's:     BM*2 + 32 bytes -> 32 * odd (conflict-free transposing reads)
// TN k map of a 32-k chunk: lane group q, element 4h+e  <->  k = 16h + 4q + e  (same for both operands).
// NPRE > 1 (small-M problems: a few hundred workgroups, each a short chain of stages): the loads of NPRE stages are
// issued into registers at once and consumed stage by stage - one memory latency per NPRE stages instead of one per
// stage, which is what a 256 x 1024 x 1024 GEMM spends its 13 us on.
template <bool TN, int TM, int TNT, int BK, typename TC, int NPRE = 1, bool RICH = false>
__global__ __launch_bounds__(256) void gemm_bf16(GemmArgs g) {
    constexpr int BM_ = 32 * TM, BN_ = 32 * TNT;
    constexpr int RA = TN ? BM_ * 2 + 32 : BK * 2 + 32, RB = TN ? BN_ * 2 + 32 : BK * 2 + 32;
    constexpr int ROWS_A = TN ? BK : BM_, ROWS_B = TN ? BK : BN_;
    constexpr int SZA = ROWS_A * RA, SZB = ROWS_B * RB;
    constexpr int VPA = (RA - 32) / 16, VPB = (RB - 32) / 16;          // 16-byte vectors per LDS row
    constexpr int NVA = ROWS_A * VPA / 256, NVB = ROWS_B * VPB / 256;  // vectors per thread per stage
    static_assert(ROWS_A * VPA % 256 == 0 && ROWS_B * VPB % 256 == 0, "staging must divide over 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char dl[];
    const bf16* A = static_cast<const bf16*>(g.A);
    const bf16* B = static_cast<const bf16*>(g.B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM_, n0 = blockIdx.x * BN_;
    const int r = lane & 15, q = lane >> 4;

    f32x4 acc[TM][TNT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kbeg = g.ws ? blockIdx.z * g.kper : 0;
    const int kend = g.ws ? (kbeg + g.kper < g.K ? kbeg + g.kper : g.K) : g.K;

    auto fetch_to = [&](int k0, uint4 (&ra)[NVA], uint4 (&rb)[NVB]) {
#pragma unroll
        for (int j = 0; j < NVA; ++j) {
            const int i = tid + j * 256, row = i / VPA, sl = i % VPA;
            ra[j] = TN ? load_row_chunk<bf16>(A, k0 + row, g.lda, m0 + sl * 8, g.M, k0 + row < kend)
                       : load_row_chunk<bf16>(A, m0 + row, g.lda, k0 + sl * 8, kend, m0 + row < g.M);
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            const int i = tid + j * 256, row = i / VPB, sl = i % VPB;
            rb[j] = TN ? load_row_chunk<bf16>(B, k0 + row, g.ldb, n0 + sl * 8, g.N, k0 + row < kend)
                       : load_row_chunk<bf16>(B, n0 + row, g.ldb, k0 + sl * 8, kend, n0 + row < g.N);
        }
    };
    auto stash_from = [&](int buf, const uint4 (&ra)[NVA], const uint4 (&rb)[NVB]) {
        unsigned char* As = dl + buf * (SZA + SZB);
        unsigned char* Bs = As + SZA;
#pragma unroll
        for (int j = 0; j < NVA; ++j) {
            const int i = tid + j * 256;
            *reinterpret_cast<uint4*>(As + (i / VPA) * RA + (i % VPA) * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NVB; ++j) {
            const int i = tid + j * 256;
            *reinterpret_cast<uint4*>(Bs + (i / VPB) * RB + (i % VPB) * 16) = rb[j];
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* As = dl + buf * (SZA + SZB);
        const unsigned char* Bs = As + SZA;
#pragma unroll
        for (int c = 0; c < BK / 32; ++c) {
            uint4 fa[TM], fb[TNT];
            if constexpr (!TN) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const uint4*>(As + ((wm * TM + i) * 16 + r) * RA + (4 * c + q) * 16);
#pragma unroll
                for (int j = 0; j < TNT; ++j)
                    fb[j] = *reinterpret_cast<const uint4*>(Bs + ((wn * TNT + j) * 16 + r) * RB + (4 * c + q) * 16);
            } else {
                const int rq = r >> 2, p = r & 3;
                const int kk = c * 32 + q * 4 + rq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned char* ad = As + kk * RA + ((wm * TM + i) * 16 + 4 * p) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ad));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(ad + 16 * RA));
                    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fa[i] = __builtin_bit_cast(uint4, f);
                }
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    const unsigned char* ad = Bs + kk * RB + ((wn * TNT + j) * 16 + 4 * p) * 2;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ad));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(ad + 16 * RB));
                    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fb[j] = __builtin_bit_cast(uint4, f);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j) Mma<bf16>::mma(acc[i][j], fb[j], fa[i]);  // D rows = n, D cols = m
        }
    };
    if constexpr (NPRE == 1) {
        uint4 ra[NVA], rb[NVB];
        fetch_to(kbeg, ra, rb);
        stash_from(0, ra, rb);
        __syncthreads();
        int buf = 0;
        for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
            const bool more = k0 + BK < kend;
            if (more) fetch_to(k0 + BK, ra, rb);
            compute(buf);
            if (more) stash_from(buf ^ 1, ra, rb);
            __syncthreads();
        }
    } else {
        uint4 pa[NPRE][NVA], pb[NPRE][NVB];
        int buf = 0;
        for (int kr = kbeg; kr < kend; kr += NPRE * BK) {
#pragma unroll
            for (int st = 0; st < NPRE; ++st)
                if (kr + st * BK < kend) fetch_to(kr + st * BK, pa[st], pb[st]);     // (uniform)
#pragma unroll
            for (int st = 0; st < NPRE; ++st) {
                if (kr + st * BK < kend) {
                    // buffer `buf` was last read two stages ago; every wave has passed a barrier since
                    stash_from(buf, pa[st], pb[st]);
                    __syncthreads();
                    compute(buf);
                    buf ^= 1;
                }
            }
        }
        __syncthreads();                                  // the epilogue reuses the staging LDS
    }

    if (g.ws) {
        float* W = g.ws + (long)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 16 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const int n = n0 + (wn * TNT + j) * 16 + q * 4;
                if (n + 3 < g.N && (g.N & 3) == 0) {
                    *reinterpret_cast<f32x4*>(W + (long)m * g.N + n) = acc[i][j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < g.N) W[(long)m * g.N + n + e] = acc[i][j][e];
                }
            }
        }
        return;
    }
    TC* C = static_cast<TC*>(g.C);
    const bool vec_ok = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                        (g.C2 == nullptr || (g.ldc2 % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C2) & 7) == 0));
    if (vec_ok && n0 + BN_ <= g.N && (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0 && !(RICH && g.act == 2)) {
        // Whole-tile case: after the MFMAs a lane owns 4 consecutive n of ONE m, so a store instruction would scatter
        // 16 rows x 32- or 64-byte pieces.  Each wave passes its 16 m x (TNT*16) n block through LDS (free after the
        // loop's last barrier) and writes whole row segments: 128 (bf16) or 256 (f32) contiguous bytes per row.
        constexpr int EROW = TNT * 16 * 4 + 16, VPR = TNT * 4;      // LDS row bytes (f32), 4-element vectors per row
        unsigned char* ew = dl + wave * 16 * EROW;
        const int nb = n0 + wn * TNT * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 16;
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const int m = mb + r, n = nb + j * 16 + q * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                if (m < g.M) {
                    if (g.bias) {
                        const float4 t = *reinterpret_cast<const float4*>(g.bias + n);
                        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                    }
                    if (g.rowadd) {
                        const float* ra = g.rowadd + (long)(m / g.rowadd_group) * g.N + n;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ra[e];
                    }
                    if (g.act == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (RICH && g.act == 3) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = selu_f(v[e]);
                    }
                    if (g.accumulate) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += to_f32(C[(long)m * g.ldc + n + e]);
                    }
                    if (g.emode >= 16 && g.emode < 32) {
                        const float* em = g.emul + (long)m * g.ldm + n;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= act_grad(em[e], g.emode - 16);
                    }
                }
                *reinterpret_cast<float4*>(ew + r * EROW + (j * 16 + q * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int it = 0; it < 16 * VPR / 64; ++it) {
                const int idx = it * 64 + lane, row = idx / VPR, c = idx % VPR;
                const float4 t = *reinterpret_cast<const float4*>(ew + row * EROW + c * 16);
                const int m = mb + row;
                if (m < g.M) {
                    float o[4] = {t.x, t.y, t.z, t.w};
                    if (g.emode >= 32) {        // (the raw x went through LDS: both products are formed here)
                        const float* e1 = g.emul + (long)m * g.ldm + nb + c * 4;
                        const float* e2 = g.emul2 + (long)m * g.ldm + nb + c * 4;
                        float o2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o2[e] = o[e] * e1[e] * act_grad(e2[e], g.emode - 32);
                            o[e] = o[e] * e2[e] * act_grad(e1[e], g.emode - 32);
                        }
                        store_vec4<TC>(C + (long)m * g.ldc + nb + c * 4, o);
                        store_vec4<bf16>(g.C2 + (long)m * g.ldc2 + nb + c * 4, o2);
                        continue;
                    }
                    store_vec4<TC>(C + (long)m * g.ldc + nb + c * 4, o);
                    if (g.C2) {
                        if (g.emode == 1) {
                            const float* em = g.emul + (long)m * g.ldm + nb + c * 4;
                            const float o2[4] = {o[0] * em[0], o[1] * em[1], o[2] * em[2], o[3] * em[3]};
                            store_vec4<bf16>(g.C2 + (long)m * g.ldc2 + nb + c * 4, o2);
                        } else {
                            store_vec4<bf16>(g.C2 + (long)m * g.ldc2 + nb + c * 4, o);
                        }
                    }
                }
            }
        }
        return;
    }
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 16 + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            const int n = n0 + (wn * TNT + j) * 16 + q * 4;
            if (n >= g.N) continue;
            lsum += epilogue_store<TC, RICH>(g, C, m, n, acc[i][j], vec_ok);
        }
    }
    if (RICH && g.act == 2) bce_block_partial(g, lsum, reinterpret_cast<double*>(dl));
}

// C = epilogue(sum_z ws[z]) in fixed slice order.  VEC4: four consecutive columns per thread (N, ldc, ldc2 multiples
// of 4: the launcher checks), one element otherwise.
template <typename TC, bool VEC4>
__global__ void gemm_splitk_reduce(GemmArgs g, int splits) {
    constexpr int V = VEC4 ? 4 : 1;
    TC* C = static_cast<TC*>(g.C);
    const long total = (long)g.M * g.N;
    const int nv = g.N / V;
    for (long iv = (long)blockIdx.x * blockDim.x + threadIdx.x; iv < total / V; iv += (long)gridDim.x * blockDim.x) {
        const int n = (int)(iv % nv) * V;
        const long m = iv / nv;
        const long i = m * g.N + n;
        float x[V];
#pragma unroll
        for (int e = 0; e < V; ++e) x[e] = 0.f;
        for (int z = 0; z < splits; ++z) {
            if constexpr (VEC4) {
                const float4 t = *reinterpret_cast<const float4*>(g.ws + (long)z * total + i);
                x[0] += t.x; x[1] += t.y; x[2] += t.z; x[3] += t.w;
            } else {
                x[0] += g.ws[(long)z * total + i];
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (g.bias) x[e] += g.bias[n + e];
            if (g.rowadd) x[e] += g.rowadd[(m / g.rowadd_group) * g.N + n + e];
            if (g.act == 1 || g.act == 3) x[e] = act_apply(x[e], g.act);
            if (g.accumulate) x[e] += to_f32(C[m * g.ldc + n + e]);
            if (g.emode >= 16 && g.emode < 32) x[e] *= act_grad(g.emul[m * g.ldm + n + e], g.emode - 16);
        }
        float x2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            x2[e] = g.emode == 1 ? x[e] * g.emul[m * g.ldm + n + e] : x[e];
            if (g.emode >= 32) {
                const float e1 = g.emul[m * g.ldm + n + e], e2 = g.emul2[m * g.ldm + n + e];
                x2[e] = x[e] * e1 * act_grad(e2, g.emode - 32);
                x[e] = x[e] * e2 * act_grad(e1, g.emode - 32);
            }
        }
        if constexpr (VEC4) {
            if constexpr (sizeof(TC) == 4) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(C) + m * g.ldc + n) = make_float4(x[0], x[1], x[2], x[3]);
            } else {
                bf16x4 o = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(C) + m * g.ldc + n) = o;
            }
            if (g.C2) {
                bf16x4 o = {(bf16)x2[0], (bf16)x2[1], (bf16)x2[2], (bf16)x2[3]};
                *reinterpret_cast<bf16x4*>(g.C2 + m * g.ldc2 + n) = o;
            }
        } else {
            C[m * g.ldc + n] = from_f32<TC>(x[0]);
            if (g.C2) g.C2[m * g.ldc2 + n] = (bf16)x2[0];
        }
    }
}

template <bool TN, int TM, int TNT, int BK, typename TC, int NPRE = 1, bool RICH = false>
void launch_bf16(hipStream_t s, const GemmArgs& g, int splits) {
    constexpr int BM_ = 32 * TM, BN_ = 32 * TNT;
    constexpr int lds = 2 * ((TN ? BK : BM_) * (TN ? BM_ * 2 + 32 : BK * 2 + 32) +
                             (TN ? BK : BN_) * (TN ? BN_ * 2 + 32 : BK * 2 + 32));
    auto k = gemm_bf16<TN, TM, TNT, BK, TC, NPRE, RICH>;
    if constexpr (lds > 64 * 1024) {
        static bool attr_set[64] = {};   // per device (the attribute lives on the device's copy of the function);
        int dev = 0;                     // benign race: setting it twice is idempotent
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    dim3 grid((g.N + BN_ - 1) / BN_, (g.M + BM_ - 1) / BM_, splits);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, g);
}

template <typename T, typename TC>
int launch(hipStream_t s, int trans, const GemmArgs& g, int splits) {
    if constexpr (sizeof(T) == 2) {
        // 128x128 tiles once they alone fill most of the chip, 64x64 (with the caller's split-K) otherwise
        const long big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        // small-M NT problems (the head's and the GRU's M = 256 GEMMs): 32x32 tiles, eight stages of loads in flight
        const long small = (long)((g.M + 31) / 32) * ((g.N + 31) / 32);
        const bool rich = g.act >= 2;                                 // SELU / sigmoid-BCE epilogue (NT only)
        if (rich && trans) return VQA_ERR_ARG;
        if (!trans && splits == 1 && small <= 256 && g.K <= 1024) {   // (one workgroup per CU, one round of loads)
            if (rich) launch_bf16<false, 1, 1, 128, TC, 8, true>(s, g, splits);
            else launch_bf16<false, 1, 1, 128, TC, 8>(s, g, splits);
        } else if (big >= 160) {      // (also with the caller's split-K: 3072 x 1024 x 3584 TN, 192 tiles, 75 -> 57 us in two slices)
            if (trans) launch_bf16<true, 4, 4, 64, TC>(s, g, splits);
            else if (rich) launch_bf16<false, 4, 4, 64, TC, 1, true>(s, g, splits);
            else launch_bf16<false, 4, 4, 64, TC>(s, g, splits);
        } else {
            if (trans) launch_bf16<true, 2, 2, 128, TC>(s, g, splits);
            else if (rich) launch_bf16<false, 2, 2, 128, TC, 1, true>(s, g, splits);
            else launch_bf16<false, 2, 2, 128, TC>(s, g, splits);
        }
    } else {
        dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
        if (trans) hipLaunchKernelGGL((gemm_kernel<T, TC, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_kernel<T, TC, false>), grid, dim3(256), 0, s, g);
    }
    if (g.ws) {
        const bool v4 = g.N % 4 == 0 && g.ldc % 4 == 0 && (!g.C2 || g.ldc2 % 4 == 0) &&
                        (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.C2) & 7) == 0 &&
                        (reinterpret_cast<uintptr_t>(g.ws) & 15) == 0;
        const long blocks = ((long)g.M * g.N / (v4 ? 4 : 1) + 255) / 256;      // one item per thread
        if (v4) hipLaunchKernelGGL((gemm_splitk_reduce<TC, true>), dim3((unsigned)blocks), dim3(256), 0, s, g, splits);
        else hipLaunchKernelGGL((gemm_splitk_reduce<TC, false>), dim3((unsigned)blocks), dim3(256), 0, s, g, splits);
    }
    CHECK_LAUNCH();
    return VQA_OK;
}

}  // namespace

extern "C" int vqa_gemm(void* stream, int dtype, int trans, int M, int N, int K, const void* A, int lda,
                        const void* B, int ldb, void* C, int ldc, int c_dtype, const float* bias,
                        const float* rowadd, int rowadd_group, int act, int accumulate, void* C2, int ldc2,
                        float* splitk_ws, int splits, const float* emul, int ldm, int emode, const float* emul2) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return VQA_ERR_ARG;
    if (emode != 0 && emode != 1 && emode != 16 + VQA_ACT_RELU && emode != 16 + VQA_ACT_SELU &&
        emode != 32 + VQA_ACT_RELU && emode != 32 + VQA_ACT_SELU)
        return VQA_ERR_ARG;
    if ((emode != 0) != (emul != nullptr) || (emode && ldm < N) || ((emode == 1 || emode >= 32) && !C2) ||
        (emode >= 32) != (emul2 != nullptr))
        return VQA_ERR_ARG;
    if (rowadd && rowadd_group <= 0) return VQA_ERR_ARG;
    if (splits < 1 || (splits > 1 && !splitk_ws)) return VQA_ERR_ARG;
    const int kc = dtype == VQA_BF16 ? 128 : 16;   // split boundaries on whole pipeline stages
    const int kper = ((K + splits - 1) / splits + kc - 1) / kc * kc;
    splits = splits > 1 ? (K + kper - 1) / kper : 1;
    if (act != 0 && act != 1 && act != 3) return VQA_ERR_ARG;
    GemmArgs g{M, N, K, A, lda, B, ldb, C, ldc, bias, rowadd, rowadd_group, act, accumulate,
               static_cast<bf16*>(C2), ldc2, splits > 1 ? splitk_ws : nullptr, kper, nullptr, 0, nullptr, 0, nullptr,
               emul, ldm, emode, emul2};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == VQA_F32 && c_dtype == VQA_F32) return launch<float, float>(s, trans, g, splits);
    if (dtype == VQA_BF16 && c_dtype == VQA_F32) return launch<bf16, float>(s, trans, g, splits);
    if (dtype == VQA_BF16 && c_dtype == VQA_BF16) return launch<bf16, bf16>(s, trans, g, splits);
    return VQA_ERR_ARG;
}

// ---- classifier GEMM with the sigmoid-BCE epilogue ---------------------------------------------------------------------
namespace {
// sparse soft-score targets: the (label, score) slots of each row after the target-free epilogue pass
__global__ void fc_bce_sparse_fix(const float* x, int ldx, const long* labels, const float* scores, int M, int N, int L,
                                  float* dl, int ldd, double* rowfix) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= M) return;
    double s = 0.0;
    for (int l = lane; l < L; l += 64) {
        const long lb = labels[(long)b * L + l];
        if (lb >= 0 && lb < N) {                           // labels are distinct within a row: no two lanes share dl[lb]
            const float sc = scores[(long)b * L + l];
            s -= (double)(x[(long)b * ldx + lb] * sc);
            dl[(long)b * ldd + lb] -= sc;
        }
    }
    s = wave_sum_d(s);
    if (lane == 0) rowfix[b] = s;
}
__global__ void fc_bce_final(const double* part, int n, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}
}  // namespace

// logits = A[M,K] * W[N,K]^T (the weight-normed fc of vqa_model.py:103, no bias) with the loss of vqa_model.py:215
// formed in the GEMM's epilogue: loss[0] = BCEWithLogits(sum), dlogits = sigmoid(logits) - target; the logits are
// written once and never re-read.  target: dense f32 [M][ldt], or NULL with labels / scores [M][L] (sparse form, see
// vqa_bce_sum_sparse), or all NULL (t = 0).  part: vqa_fc_bce_part(M, N) doubles of scratch.
extern "C" int vqa_fc_bce_part(int M, int N) { return ((M + 31) / 32) * ((N + 31) / 32) + M; }
extern "C" int vqa_fc_bce(void* stream, int dtype, int M, int N, int K, const void* A, int lda, const void* W, int ldw,
                          float* logits, int ldc, const float* target, int ldt, const long* labels, const float* scores,
                          int L, float* dlogits, int ldd, double* part, float* loss) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !logits || !dlogits || !part || !loss) return VQA_ERR_ARG;
    if (target && labels) return VQA_ERR_ARG;
    if (labels && (!scores || L < 1)) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // workgroups (= loss partials) of the GEMM that launch<> will pick: only those slots, and the M row corrections of
    // the sparse form behind them, are summed - no fill of the scratch on the critical chain (it was a memset)
    long tiles;
    if (dtype == VQA_BF16) {
        const long big = (long)((M + 127) / 128) * ((N + 127) / 128), small = (long)((M + 31) / 32) * ((N + 31) / 32);
        tiles = (small <= 256 && K <= 1024) ? small : (big >= 160 ? big : (long)((M + 63) / 64) * ((N + 63) / 64));
    } else {
        tiles = (long)((N + BN - 1) / BN) * ((M + BM - 1) / BM);
    }
    if (tiles + M > vqa_fc_bce_part(M, N)) return VQA_ERR_ARG;
    GemmArgs g{M, N, K, A, lda, W, ldw, logits, ldc, nullptr, nullptr, 1, 2, 0, nullptr, 0, nullptr, K,
               target, ldt, dlogits, ldd, part, nullptr, 0, 0, nullptr};
    int st;
    if (dtype == VQA_F32) st = launch<float, float>(s, 0, g, 1);
    else if (dtype == VQA_BF16) st = launch<bf16, float>(s, 0, g, 1);
    else return VQA_ERR_ARG;
    if (st != VQA_OK) return st;
    if (labels)
        hipLaunchKernelGGL(fc_bce_sparse_fix, dim3((M + 3) / 4), dim3(256), 0, s, logits, ldc, labels, scores, M, N, L,
                           dlogits, ldd, part + tiles);
    hipLaunchKernelGGL(fc_bce_final, dim3(1), dim3(64), 0, s, part, (int)(tiles + (labels ? M : 0)), loss);
    CHECK_LAUNCH();
    return VQA_OK;
}
