// BatchNorm2d + ReLU + 2x2 pool around the convolutions (reference cnn.py:23-41), NHWC, HBM-bound kernels.
//   bn_finalize        per-channel batch statistics from the conv epilogue's (sum, sumsq) partials -> mean, invstd,
//                      scale = gamma*invstd, shift = beta - mean*scale; running-stat update (momentum, unbiased var)
//   bn_eval_coeffs     scale/shift from the running statistics (eval mode)
//   bn_act_pool_fwd    a = pool(relu(scale*y + shift)),  pool in {none, max 2x2, avg 2x2}
//   bn_pool_rows_fwd   the pooled cases of the same, one workgroup per pooled row, input rows staged through LDS
//   bn_bwd_reduce      recompute dz = relu'/pool' routed gradient from (y, dout); partial sums of dz and dz*xhat
//   bn_bwd_finalize    dgamma, dbeta, c1 = dbeta/N, c2 = dgamma/N
//   bn_bwd_apply       dy = scale * (dz - c1 - xhat*c2)
// Every thread moves 16-byte channel vectors; reductions are shuffle + fixed-order partials (deterministic).
#include "common.h"

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// Sums of the two columns (c, 0) and (c, 1) of part[rows][C][2] over the rows, fp64, fixed order; 256 threads; the
// result is valid in thread 0.  Four rows per thread are in flight at a time.
__device__ __forceinline__ void column_sums(const float* part, int rows, int C, int c, double& s1, double& s2) {
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = threadIdx.x; i0 < rows; i0 += 4 * 256) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            v[u] = i < rows ? *reinterpret_cast<const float2*>(part + ((long)i * C + c) * 2) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a1[u] += (double)v[u].x; a2[u] += (double)v[u].y; }
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    __shared__ double r1[4], r2[4];
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    s1 = r1[0] + r1[1] + r1[2] + r1[3];
    s2 = r2[0] + r2[1] + r2[2] + r2[3];
}

// One workgroup per channel: sums the conv epilogue's partial rows (a few hundred to a few thousand: one per
// persistent conv workgroup) in fp64 in a fixed order, then produces the BN coefficients and running statistics.
__global__ __launch_bounds__(256) void bn_finalize(const float* part, int rows, int C, double count,
                                                   const float* gamma, const float* beta, float* running_mean,
                                                   float* running_var, float momentum, float eps, float* mean,
                                                   float* invstd, float* scale, float* shift, long* nbt) {
    const int c = blockIdx.x;
    if (nbt && c == 0 && threadIdx.x == 0) nbt[0] += 1;        // nn.BatchNorm2d.num_batches_tracked (one writer)
    // (this kernel sits between every conv and its BatchNorm pass: the parameter loads are issued first and the rows
    // are read four at a time, so the whole kernel is two memory round trips instead of rows/256 + 2)
    float ga = 0.f, be = 0.f, rm = 0.f, rv = 0.f;
    if (threadIdx.x == 0) {
        ga = gamma[c]; be = beta[c];
        if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    }
    double s1, s2;
    column_sums(part, rows, C, c, s1, s2);
    if (threadIdx.x != 0) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = ga * is;
    scale[c] = sc;
    shift[c] = be - (float)m * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * rm + momentum * (float)m;
        running_var[c] = (1.f - momentum) * rv + momentum * (float)(var * count / (count - 1.0));
    }
}

__global__ void bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(running_var[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
}

// All layers of the encoder in one launch (inference): scale, shift and the bias the folded conv adds,
// b' = conv_bias * scale + shift (W' = W * scale is folded at packing time: conv + BatchNorm + ReLU = one kernel).
constexpr int BN_MAX_LAYERS = 16;
struct EvalJobs {
    const float* gamma[BN_MAX_LAYERS];
    const float* beta[BN_MAX_LAYERS];
    const float* rm[BN_MAX_LAYERS];
    const float* rv[BN_MAX_LAYERS];
    const float* bias[BN_MAX_LAYERS];
    int C[BN_MAX_LAYERS];
    float* coef;            // [n][3][ldc]: scale, shift, folded bias
    int ldc;
    float eps;
};
__global__ void bn_eval_coeffs_all(EvalJobs j) {
    const int li = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= j.C[li]) return;
    const float sc = j.gamma[li][c] / sqrtf(j.rv[li][c] + j.eps);
    const float sh = j.beta[li][c] - j.rm[li][c] * sc;
    float* o = j.coef + (long)li * 3 * j.ldc;
    o[c] = sc;
    o[j.ldc + c] = sh;
    o[2 * j.ldc + c] = fmaf(j.bias[li] ? j.bias[li][c] : 0.f, sc, sh);
}

// ---- work distribution of the streaming kernels.  An item = one OUTPUT pixel x one 16-byte channel vector; item i
// lives at element offset i*N of an un-pooled tensor.  Workgroup b owns the contiguous items [b*CH, (b+1)*CH),
// CH = 256*U*passes: `passes` rounds of U independent items per thread (thread t takes base + u*256 + t), all loads of
// a round issued before its first store.  Measured on MI355X with 418 MB tensors (scripts/stream_micro.hip): contiguous
// chunks per workgroup copy at 6.1-6.2 TB/s and read at 6.2 (6.8 with non-temporal loads), where the usual "cap the
// grid at 2048 and grid-stride" loop gets 4.2-5.0 (copy) and 5.5 (read).  256 % CV == 0, so a thread keeps one channel
// vector and its coefficients for its whole chunk.
constexpr int kFwdU = 4, kFwdPoolU = 2, kBwdU = 2, kBwdPoolU = 1;

template <typename T, bool NT> struct Stream {
    static constexpr int N = 16 / sizeof(T);
    static __device__ __forceinline__ void load(const T* p, float* f) {
        const v4u* q = reinterpret_cast<const v4u*>(p);
        const v4u v = NT ? __builtin_nontemporal_load(q) : *q;
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = to_f32(e[i]);
    }
    static __device__ __forceinline__ void store(T* p, const float* f) {
        T e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = from_f32<T>(f[i]);
        const v4u v = *reinterpret_cast<const v4u*>(e);
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(p)); else *reinterpret_cast<v4u*>(p) = v;
    }
};
template <int N> __device__ __forceinline__ void coeffs(const float* src, int cv, float* dst) {
#pragma unroll
    for (int e = 0; e < N; e += 4) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + cv * N + e);
}
// top-left element offset of the 2x2 window of pooled pixel `px` (32-bit divisions: B*PH*PW < 2^32)
__device__ __forceinline__ long window_base(unsigned px, int PH, int PW, int H, int W, int C) {
    const unsigned rowi = px / (unsigned)PW;
    const int pw = (int)(px - rowi * (unsigned)PW);
    const long b = rowi / (unsigned)PH;
    const int ph = (int)(rowi - (unsigned)b * (unsigned)PH);
    return ((b * H + 2 * ph) * W + 2 * pw) * C;
}

// a = pool(relu(scale*y + shift)).  pool: 0 none, 1 max, 2 avg.
template <typename T, int POOL, bool NT>
__global__ __launch_bounds__(256) void bn_act_pool_fwd(const T* __restrict__ y, T* __restrict__ a,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int B, int H, int W, int C, int passes, int rev) {
    // (rev: the workgroups take their chunks from the tensor's end - see unit_walk, common.h)
    const unsigned bx = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    using S = Stream<T, NT>;
    constexpr int N = S::N, U = POOL ? kFwdPoolU : kFwdU;
    const int CV = C / N;
    const int PH = POOL ? H / 2 : H, PW = POOL ? W / 2 : W;
    const long total = (long)B * PH * PW * CV;
    const int cv = threadIdx.x % CV;
    float sc[N], sh[N];
    coeffs<N>(scale, cv, sc);
    coeffs<N>(shift, cv, sh);
    long i = (long)bx * (256 * U) * passes + threadIdx.x;
    for (int p = 0; p < passes; ++p, i += 256 * U) {
        if (i >= total) break;
        if (POOL == 0) {
            float v[U][N];
#pragma unroll
            for (int u = 0; u < U; ++u) {           // out-of-range items re-read the last one (branch-free loads)
                const long j = i + u * 256 < total ? i + u * 256 : total - 1;
                S::load(y + j * N, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int e = 0; e < N; ++e) v[u][e] = fmaxf(fmaf(v[u][e], sc[e], sh[e]), 0.f);
                if (i + u * 256 < total) S::store(a + (i + u * 256) * N, v[u]);
            }
        } else {
            float v[U][4][N];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long j = i + u * 256 < total ? i + u * 256 : total - 1;
                const T* base = y + window_base((unsigned)(j / CV), PH, PW, H, W, C) + cv * N;
#pragma unroll
                for (int k = 0; k < 4; ++k) S::load(base + ((long)(k >> 1) * W + (k & 1)) * C, v[u][k]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float o[N];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    float acc = 0.f;                // relu floor; for avg: running sum
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float z = fmaxf(fmaf(v[u][k][e], sc[e], sh[e]), 0.f);
                        acc = POOL == 1 ? fmaxf(acc, z) : acc + z;
                    }
                    o[e] = POOL == 2 ? acc * 0.25f : acc;
                }
                if (i + u * 256 < total) S::store(a + (i + u * 256) * N, o);
            }
        }
    }
}

// Pooled forward, row-staged: one workgroup per (image, pooled row).  The two input rows of a pooled row are one
// contiguous stretch of y (2*W*C elements): it is read with lane-contiguous 16-byte loads into LDS, and every output
// vector takes its 2x2 window from there.  In bn_act_pool_fwd<POOL != 0> each load instruction picks the even or the
// odd pixels of a row - half of every line it touches.  224x224x16: 0.103 -> 0.086 ms (6.0 TB/s), 112x112x32: 0.053 ->
// 0.040.  (The same staging for the backward reduce pass measured slower: its workgroups loop over several rows to
// keep the number of partial-sum rows down, and the barriers of that loop cost more than the access pattern gains; for
// the backward apply pass, load -> barrier -> dy in place -> barrier -> store per row pair: 224x224x16 0.168 -> 0.189.)
// YS (max pool): also writes ysel = the RAW conv output at the position the pool selects (the first maximum of z in
// window order - the position bn_bwd routes the gradient to), pooled resolution.  With it the BatchNorm-backward sums of
// this layer, dz = d * 1[scale * ysel + shift > 0] and dz * xhat(ysel), need only pooled-resolution tensors: the data
// gradient that produces d forms them in its epilogue (conv.hip BST with bn_y = ysel) and the reduce pass over the
// full-resolution y (4x the bytes) never runs.
template <typename T, int POOL, bool NT, bool YS = false>
__global__ __launch_bounds__(256) void bn_pool_rows_fwd(const T* __restrict__ y, T* __restrict__ a,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int PH, int H, int W, int C, int rev,
                                                        T* __restrict__ ysel = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using S = Stream<T, NT>;
    constexpr int N = S::N;
    const int CV = C / N, PW = W / 2;
    const int bx = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int b = bx / PH, ph = bx - b * PH;
    const int nv = 2 * W * CV;                                  // 16-byte vectors of the row pair
    const v4u* src = reinterpret_cast<const v4u*>(y + ((long)(b * H + 2 * ph) * W) * C);
    v4u* lds = reinterpret_cast<v4u*>(smem);
    constexpr int LU = 4;                                       // loads in flight per thread
    for (int i0 = 0; i0 < nv; i0 += 256 * LU) {
        v4u v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            const v4u* q = src + (i < nv ? i : nv - 1);
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            if (i < nv) lds[i] = v[u];
        }
    }
    __syncthreads();
    const int cv = threadIdx.x % CV;
    float sc[N], sh[N];
    coeffs<N>(scale, cv, sc);
    coeffs<N>(shift, cv, sh);
    T* dst = a + ((long)(b * PH + ph) * PW) * C;
    for (int o = threadIdx.x; o < PW * CV; o += 256) {          // 256 % CV == 0: o % CV == cv
        const int pw = o / CV;
        float acc[N];
        if constexpr (YS) {
            static_assert(!YS || POOL == 1, "ysel: max pool");
            float bz[N], by[N];                                 // best z so far and the y it came from
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4u raw = lds[((k >> 1) * W + 2 * pw + (k & 1)) * CV + cv];
                const T* el = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const float yv = to_f32(el[e]), z = fmaf(yv, sc[e], sh[e]);
                    const bool better = k == 0 || z > bz[e];    // strict: the FIRST maximum keeps its place
                    bz[e] = better ? z : bz[e];
                    by[e] = better ? yv : by[e];
                }
            }
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] = fmaxf(bz[e], 0.f);
            S::store(ysel + ((long)(b * PH + ph) * PW) * C + (long)o * N, by);
        } else {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = 0.f;               // relu floor; for avg: running sum
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v4u raw = lds[((k >> 1) * W + 2 * pw + (k & 1)) * CV + cv];
            const T* el = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float z = fmaxf(fmaf(to_f32(el[e]), sc[e], sh[e]), 0.f);
                acc[e] = POOL == 1 ? fmaxf(acc[e], z) : acc[e] + z;
            }
        }
        }
        if (POOL == 2) {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] *= 0.25f;
        }
        S::store(dst + (long)o * N, acc);
    }
}

// per-thread partial sums (a1, a2 for the thread's channel vector cv = tid % CV) -> one row of `part` per workgroup
template <int N>
__device__ __forceinline__ void reduce_rows(float (&a1)[N], float (&a2)[N], int CV, int C, float* __restrict__ part,
                                            unsigned bx /* the workgroup's row: its chunk number */) {
    __shared__ float red[4][512];   // [wave][channel][2], C <= 256
    // lanes with equal cv inside a wave: offsets CV, 2CV, ... (CV <= 64, power of two)
#pragma unroll
    for (int e = 0; e < N; ++e) {
        for (int o = CV; o < 64; o <<= 1) {
            a1[e] += __shfl_xor(a1[e], o);
            a2[e] += __shfl_xor(a2[e], o);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* rw = &red[0][0];
    const int per_wave = (CV < 64 ? CV : 64) * N * 2;
    if (lane < CV) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            rw[wave * per_wave + (lane * N + e) * 2] = a1[e];
            rw[wave * per_wave + (lane * N + e) * 2 + 1] = a2[e];
        }
    }
    __syncthreads();
    // waves hold the same channels when CV <= 64 (CV divides 64): sum the 4 waves in fixed order
    for (int t = threadIdx.x; t < C * 2; t += blockDim.x) {
        const float s4 = rw[t] + rw[per_wave + t] + rw[2 * per_wave + t] + rw[3 * per_wave + t];
        part[(long)bx * C * 2 + t] = s4;
    }
}

// BatchNorm backward.  dz for the (up to) 4 positions of one output pixel: max pool sends the gradient to the first
// maximum of relu(z) in window order (ATen's choice); a window whose maximum is 0 passes nothing (relu' = 0 there).
//   APPLY = false: partial sums of dz and dz*xhat, one row of `part` per workgroup
//   APPLY = true : dy = scale * (dz - c1 - xhat*c2)
template <typename T, bool APPLY, int pool, bool NT>
__global__ __launch_bounds__(256) void bn_bwd(const T* __restrict__ y, const T* __restrict__ dout, T* __restrict__ dy,
                                              const float* scale, const float* shift, const float* mean,
                                              const float* invstd, const float* c1, const float* c2,
                                              float* __restrict__ part, int B, int H, int W, int C, int passes,
                                              int rev) {
    // (rev: chunks from the tensor's end; the partial rows keep their chunk's number, so the sums are the same bits)
    const unsigned bx = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    using S = Stream<T, NT>;
    constexpr int N = S::N, U = pool ? kBwdPoolU : kBwdU;
    const int CV = C / N;                       // channel vectors; power of two <= 64
    const int PH = pool ? H / 2 : H, PW = pool ? W / 2 : W;
    const long total = (long)B * PH * PW * CV;
    const int cv = threadIdx.x % CV;
    float sc[N], sh[N], mu[N], is[N], k1[N], k2[N], a1[N], a2[N];
    coeffs<N>(scale, cv, sc);
    coeffs<N>(shift, cv, sh);
    coeffs<N>(mean, cv, mu);
    coeffs<N>(invstd, cv, is);
    if (APPLY) { coeffs<N>(c1, cv, k1); coeffs<N>(c2, cv, k2); }
#pragma unroll
    for (int e = 0; e < N; ++e) a1[e] = a2[e] = 0.f;
    constexpr int np = pool ? 4 : 1;
    long i = (long)bx * (256 * U) * passes + threadIdx.x;
    for (int p = 0; p < passes; ++p, i += 256 * U) {
        if (i >= total) break;
        float g[U][N], v[U][np][N];
        long off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * 256 < total ? i + u * 256 : total - 1;
            S::load(dout + j * N, g[u]);
            off[u] = pool ? window_base((unsigned)(j / CV), PH, PW, H, W, C) + cv * N : j * N;
#pragma unroll
            for (int k = 0; k < np; ++k) S::load(y + off[u] + ((long)(k >> 1) * W + (k & 1)) * C, v[u][k]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = i + u * 256 < total;
            float z[np][N];
#pragma unroll
            for (int k = 0; k < np; ++k)
#pragma unroll
                for (int e = 0; e < N; ++e) z[k][e] = fmaf(v[u][k][e], sc[e], sh[e]);
#pragma unroll
            for (int k = 0; k < np; ++k) {
                float dz[N];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    float d;
                    if (pool == 0) {
                        d = z[0][e] > 0.f ? g[u][e] : 0.f;
                    } else if (pool == 2) {
                        d = z[k][e] > 0.f ? 0.25f * g[u][e] : 0.f;
                    } else {
                        const float m = fmaxf(fmaxf(z[0][e], z[1 % np][e]), fmaxf(z[2 % np][e], z[3 % np][e]));
                        bool first = z[k][e] == m;
#pragma unroll
                        for (int j = 0; j < k; ++j) first = first && (z[j][e] != m);
                        d = (m > 0.f && first) ? g[u][e] : 0.f;
                    }
                    dz[e] = live ? d : 0.f;
                }
                if (APPLY) {
                    float o[N];
#pragma unroll
                    for (int e = 0; e < N; ++e) {
                        const float xh = (v[u][k][e] - mu[e]) * is[e];
                        o[e] = sc[e] * (dz[e] - k1[e] - xh * k2[e]);
                    }
                    if (live) S::store(dy + off[u] + ((long)(k >> 1) * W + (k & 1)) * C, o);
                } else {
#pragma unroll
                    for (int e = 0; e < N; ++e) {
                        a1[e] += dz[e];
                        a2[e] += dz[e] * ((v[u][k][e] - mu[e]) * is[e]);
                    }
                }
            }
        }
    }
    if (APPLY && pool) {
        // floor-mode pooling drops the last row / column of an odd map: dz = 0 there, dy still gets the mean terms
        const int nslot = 256 / CV, slot = threadIdx.x / CV;
        const int lw = (W & 1) ? H : 0, lh = (H & 1) ? (W - (W & 1)) : 0, nl = lw + lh;
        for (long t = (long)blockIdx.x * nslot + slot; t < (long)B * nl; t += (long)gridDim.x * nslot) {
            const long b = t / nl;
            const int j = t % nl;
            const int h = j < lw ? j : H - 1, w = j < lw ? W - 1 : j - lw;
            const long o1 = ((b * H + h) * W + w) * C + cv * N;
            float vv[N], o[N];
            S::load(y + o1, vv);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = sc[e] * (-k1[e] - (vv[e] - mu[e]) * is[e] * k2[e]);
            S::store(dy + o1, o);
        }
    }
    if (!APPLY) {
        reduce_rows<N>(a1, a2, CV, C, part, bx);
    }
}

__global__ void bn_bwd_finalize(const float* part, int rows, int C, double count, float* dgamma, float* dbeta,
                                float* c1, float* c2) {
    const int c = blockIdx.x;
    double s1, s2;
    column_sums(part, rows, C, c, s1, s2);
    if (threadIdx.x == 0) {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        c1[c] = (float)(s1 / count);
        c2[c] = (float)(s2 / count);
    }
}

}  // namespace

extern "C" int vqa_bn_finalize(void* stream, const float* part, int rows, int C, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* mean, float* invstd, float* scale, float* shift, long* num_batches_tracked) {
    if (rows < 1 || C < 1) return VQA_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), part, rows, C, count, gamma,
                       beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, num_batches_tracked);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_bn_eval_coeffs(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift) {
    hipLaunchKernelGGL(bn_eval_coeffs, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), C, gamma,
                       beta, running_mean, running_var, eps, scale, shift);
    CHECK_LAUNCH();
    return VQA_OK;
}

// n (<= 16) layers at once; gamma .. bias are HOST arrays of n device pointers (bias[i] may be NULL), C a host array of
// n channel counts (<= ldc); coef [n][3][ldc] f32 receives scale, shift and the folded bias bias*scale + shift.
extern "C" int vqa_bn_eval_coeffs_all(void* stream, int n, const void* const* gamma, const void* const* beta,
                                      const void* const* running_mean, const void* const* running_var,
                                      const void* const* bias, const int* C, float eps, float* coef, int ldc) {
    if (n < 1 || n > BN_MAX_LAYERS || !gamma || !beta || !running_mean || !running_var || !bias || !C || !coef)
        return VQA_ERR_ARG;
    EvalJobs j{};
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        if (C[i] < 1 || C[i] > ldc) return VQA_ERR_ARG;
        j.gamma[i] = static_cast<const float*>(gamma[i]); j.beta[i] = static_cast<const float*>(beta[i]);
        j.rm[i] = static_cast<const float*>(running_mean[i]); j.rv[i] = static_cast<const float*>(running_var[i]);
        j.bias[i] = static_cast<const float*>(bias[i]); j.C[i] = C[i];
        if (C[i] > cmax) cmax = C[i];
    }
    j.coef = coef; j.ldc = ldc; j.eps = eps;
    hipLaunchKernelGGL(bn_eval_coeffs_all, dim3((cmax + 255) / 256, n), dim3(256), 0, static_cast<hipStream_t>(stream), j);
    CHECK_LAUNCH();
    return VQA_OK;
}

// ---- launch geometry of the streaming kernels (see the comment above bn_act_pool_fwd)
static int grid_of(long total, int U, int passes) {
    const long per = 256L * U * passes;
    return (int)((total + per - 1) / per);
}
// rounds per workgroup that keep the grid (= rows of partial sums for the reduce pass) at <= max_grid
static int passes_for(long total, int U, int max_grid) {
    const long g = (total + 256L * U - 1) / (256L * U);
    const long p = (g + max_grid - 1) / max_grid;
    return (int)(p < 1 ? 1 : p);
}
// Non-temporal loads / stores once a launch moves more than the 256 MB Infinity Cache can keep for the next kernel
// anyway; below that, plain accesses leave the tensors on die for their consumer.  Measured per layer (B=256, bf16):
// forward 226x226x16 0.155 -> 0.138 ms (6.1 TB/s), reduce 114x114x32 0.103 -> 0.077 ms; the POOLED apply pass is the
// exception (224x224x16: 0.160 plain, 0.205 non-temporal): each of its store instructions fills half of every 64-byte
// line (the even or the odd pixels of a row), which a write-back L2 merges and a streaming store cannot.
// (VQA_NT_MIN_MB in the environment, read once.  Same-box A/B of the step, 6 alternations of 60 steps: never
// non-temporal 4.37 ms against 4.33 at 256 MB; 256 / 100 / 0 MB: 4.214 / 4.201 / 4.203.)
#ifndef VQA_NT_MIN_MB_DEFAULT
#define VQA_NT_MIN_MB_DEFAULT 100.0
#endif
static bool use_nt(double bytes) {
    static const double min_mb = [] { const char* e = getenv("VQA_NT_MIN_MB"); return e ? atof(e) : VQA_NT_MIN_MB_DEFAULT; }();
    return bytes > min_mb * 1e6;
}
constexpr int kReduceRows = 4096;   // 2048: reduce pass 3 % slower, 16384: 8 % slower (more partial rows to finalize)

extern "C" int vqa_bn_act_pool_fwd(void* stream, int dtype, const void* y, void* a, const float* scale,
                                   const float* shift, int B, int H, int W, int C, int pool, void* ysel) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pool < 0 || pool > 2 || (dtype != VQA_BF16 && dtype != VQA_F32)) return VQA_ERR_ARG;
    const int N = dtype == VQA_BF16 ? 8 : 4, esz = 16 / N;
    if (C % N || 256 % (C / N)) return VQA_ERR_ARG;         // a thread keeps one channel vector (C/N divides 256)
    const long out_px = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const long total = out_px * (C / N);
    if (total <= 0) return VQA_ERR_ARG;
    const int U = pool ? kFwdPoolU : kFwdU, passes = 1;
    const bool nt = use_nt(((double)B * H * W * C + (double)out_px * C) * esz);
    const size_t row_pair = (size_t)2 * W * C * esz;
    if (ysel && !(pool == 1 && row_pair <= 64 * 1024 && H >= 2 && W >= 2)) return VQA_ERR_ARG;   // (rows kernel only)
    if (ysel) {
        const dim3 rgrid((unsigned)(B * (H / 2)));
#define GY(T, NT) hipLaunchKernelGGL((bn_pool_rows_fwd<T, 1, NT, true>), rgrid, dim3(256), row_pair, s,          \
                                     static_cast<const T*>(y), static_cast<T*>(a), scale, shift, H / 2, H, W, C, \
                                     vqa_walk_rev(64), static_cast<T*>(ysel))
        if (dtype == VQA_BF16) { if (nt) GY(bf16, true); else GY(bf16, false); }
        else { if (nt) GY(float, true); else GY(float, false); }
#undef GY
        CHECK_LAUNCH();
        return VQA_OK;
    }
    if (pool && row_pair <= 64 * 1024 && H >= 2 && W >= 2) {
        const dim3 rgrid((unsigned)(B * (H / 2)));
#define GR(T, P, NT) hipLaunchKernelGGL((bn_pool_rows_fwd<T, P, NT>), rgrid, dim3(256), row_pair, s, \
                                        static_cast<const T*>(y), static_cast<T*>(a), scale, shift, H / 2, H, W, C, \
                                        vqa_walk_rev(64))
#define GRN(T, P) do { if (nt) GR(T, P, true); else GR(T, P, false); } while (0)
#define GRP(T) do { if (pool == 1) GRN(T, 1); else GRN(T, 2); } while (0)
        if (dtype == VQA_BF16) GRP(bf16); else GRP(float);
#undef GRP
#undef GRN
#undef GR
        CHECK_LAUNCH();
        return VQA_OK;
    }
    const dim3 grid(grid_of(total, U, passes));
#define GO(T, P, NT) hipLaunchKernelGGL((bn_act_pool_fwd<T, P, NT>), grid, dim3(256), 0, s, static_cast<const T*>(y), \
                                        static_cast<T*>(a), scale, shift, B, H, W, C, passes, vqa_walk_rev(64))
#define GN(T, P) do { if (nt) GO(T, P, true); else GO(T, P, false); } while (0)
#define GP(T) do { if (pool == 0) GN(T, 0); else if (pool == 1) GN(T, 1); else GN(T, 2); } while (0)
    if (dtype == VQA_BF16) GP(bf16); else GP(float);
#undef GP
#undef GN
#undef GO
    CHECK_LAUNCH();
    return VQA_OK;
}

static int bwd_geometry(int dtype, int B, int H, int W, int C, int pool, int apply, int* passes) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    const long total = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / N);
    const int U = pool ? kBwdPoolU : kBwdU;
    *passes = apply ? 1 : passes_for(total, U, kReduceRows);
    return grid_of(total, U, *passes);
}

// rows of `part` the reduce pass writes (= its grid size)
extern "C" int vqa_bn_bwd_rows(int dtype, int B, int H, int W, int C, int pool) {
    int passes;
    return bwd_geometry(dtype, B, H, W, C, pool, 0, &passes);
}

extern "C" int vqa_bn_bwd(void* stream, int dtype, int apply, const void* y, const void* dout, void* dy,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          const float* c1, const float* c2, float* part, int B, int H, int W, int C, int pool) {
    if (dtype != VQA_BF16 && dtype != VQA_F32) return VQA_ERR_ARG;
    const int N = dtype == VQA_BF16 ? 8 : 4, esz = 16 / N;
    const int CV = C / N;
    if (C % N || CV > 64 || (CV & (CV - 1)) || pool < 0 || pool > 2 || C > 256) return VQA_ERR_ARG;
    if (apply ? (!dy || !c1 || !c2) : !part) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int passes;
    const dim3 grid(bwd_geometry(dtype, B, H, W, C, pool, apply, &passes));
    const double ny = (double)B * H * W * C, nd = pool ? ny / 4 : ny;
    const bool nt = use_nt((ny + nd + (apply ? ny : 0.0)) * esz) && !(apply && pool);
#define GO(T, AP, P, NT) hipLaunchKernelGGL((bn_bwd<T, AP, P, NT>), grid, dim3(256), 0, s, static_cast<const T*>(y), \
                                            static_cast<const T*>(dout), static_cast<T*>(dy), scale, shift, mean, \
                                            invstd, c1, c2, part, B, H, W, C, passes, vqa_walk_rev(apply ? 128 : 256))
#define GN(T, AP, P) do { if (nt) GO(T, AP, P, true); else GO(T, AP, P, false); } while (0)
#define GP(T, AP) do { if (pool == 0) GN(T, AP, 0); else if (pool == 1) GN(T, AP, 1); else GN(T, AP, 2); } while (0)
    if (dtype == VQA_BF16) { if (apply) GP(bf16, true); else GP(bf16, false); }
    else { if (apply) GP(float, true); else GP(float, false); }
#undef GP
#undef GN
#undef GO
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_bn_bwd_finalize(void* stream, const float* part, int rows, int C, double count, float* dgamma,
                                   float* dbeta, float* c1, float* c2) {
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), part, rows, C, count,
                       dgamma, dbeta, c1, c2);
    CHECK_LAUNCH();
    return VQA_OK;
}
