// BatchNorm2d + ReLU + 2x2 pool around the convolutions (reference cnn.py:23-41), NHWC, HBM-bound kernels.
//   bn_finalize        per-channel batch statistics from the conv epilogue's (sum, sumsq) partials -> mean, invstd,
//                      scale = gamma*invstd, shift = beta - mean*scale; running-stat update (momentum, unbiased var)
//   bn_eval_coeffs     scale/shift from the running statistics (eval mode)
//   bn_act_pool_fwd    a = pool(relu(scale*y + shift)),  pool in {none, max 2x2, avg 2x2}
//   bn_bwd_reduce      recompute dz = relu'/pool' routed gradient from (y, dout); partial sums of dz and dz*xhat
//   bn_bwd_finalize    dgamma, dbeta, c1 = dbeta/N, c2 = dgamma/N
//   bn_bwd_apply       dy = scale * (dz - c1 - xhat*c2)
// Every thread moves 16-byte channel vectors; reductions are shuffle + fixed-order partials (deterministic).
#include "common.h"

namespace {

template <typename T> struct Vec {
    static constexpr int N = 16 / sizeof(T);
    static __device__ __forceinline__ void load(const T* p, float* f) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = to_f32(e[i]);
    }
    static __device__ __forceinline__ void store(T* p, const float* f) {
        T e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = from_f32<T>(f[i]);
        *reinterpret_cast<uint4*>(p) = *reinterpret_cast<uint4*>(e);
    }
};

// One workgroup per channel: sums the conv epilogue's partial rows (a few hundred to a few thousand: one per
// persistent conv workgroup) in fp64 in a fixed order, then produces the BN coefficients and running statistics.
__global__ __launch_bounds__(256) void bn_finalize(const float* part, int rows, int C, double count,
                                                   const float* gamma, const float* beta, float* running_mean,
                                                   float* running_var, float momentum, float eps, float* mean,
                                                   float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((long)i * C + c) * 2);
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    __shared__ double r1[4], r2[4];
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = r1[0] + r1[1] + r1[2] + r1[3];
    s2 = r2[0] + r2[1] + r2[2] + r2[3];
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * count / (count - 1.0));
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

// pool: 0 none, 1 max, 2 avg.  One thread = one OUTPUT pixel x one 16-byte channel vector.
template <typename T>
__global__ void bn_act_pool_fwd(const T* y, T* a, const float* scale, const float* shift, int B, int H, int W, int C,
                                int pool) {
    constexpr int N = Vec<T>::N;
    const int CV = C / N;
    const int PH = pool ? H / 2 : H, PW = pool ? W / 2 : W;
    const long total = (long)B * PH * PW * CV;
    // CV is a power of two <= 32 and divides the grid stride: a thread keeps ONE channel vector, its 2N coefficients
    // are loaded once (inside the loop they were 16 extra vector-memory instructions per 16 bytes of data)
    const int cv = threadIdx.x % CV;
    float sc[N], sh[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { sc[e] = scale[cv * N + e]; sh[e] = shift[cv * N + e]; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / CV;
        float o[N];
        if (!pool) {
            float v[N];
            Vec<T>::load(y + px * C + cv * N, v);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
        } else {
            // 32-bit unsigned division (B*PH*PW < 2^32): a third of the instructions of the 64-bit one
            const unsigned pxu = (unsigned)px, rowi = pxu / (unsigned)PW;
            const int pw = (int)(pxu - rowi * (unsigned)PW);
            const long b = rowi / (unsigned)PH;
            const int ph = (int)(rowi - (unsigned)b * (unsigned)PH);
            const T* base = y + ((b * H + 2 * ph) * W + 2 * pw) * C + cv * N;
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = 0.f;   // relu floor; for avg: running sum
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[N];
                Vec<T>::load(base + ((long)(k >> 1) * W + (k & 1)) * C, v);
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const float z = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
                    o[e] = pool == 1 ? fmaxf(o[e], z) : o[e] + z;
                }
            }
            if (pool == 2) {
#pragma unroll
                for (int e = 0; e < N; ++e) o[e] *= 0.25f;
            }
        }
        Vec<T>::store(a + px * C + cv * N, o);
    }
}

// dz for the (up to) 4 positions of one output pixel.  Max pool: the gradient goes to the first maximum of
// relu(z) in window order (ATen's choice); a window whose maximum is 0 passes nothing (relu' = 0 there).
template <typename T, bool APPLY, int pool>
__global__ __launch_bounds__(256) void bn_bwd(const T* y, const T* dout, T* dy, const float* scale, const float* shift,
                                              const float* mean, const float* invstd, const float* c1, const float* c2,
                                              float* part, int B, int H, int W, int C) {
    constexpr int N = Vec<T>::N;
    const int CV = C / N;                       // channel vectors; power of two <= 64
    const int PH = pool ? H / 2 : H, PW = pool ? W / 2 : W;
    const long npx = (long)B * PH * PW;
    const int cv = threadIdx.x % CV;
    const int slot = threadIdx.x / CV, nslot = blockDim.x / CV;
    float sc[N], sh[N], mu[N], is[N], k1[N], k2[N], a1[N], a2[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        sc[e] = scale[cv * N + e]; sh[e] = shift[cv * N + e]; mu[e] = mean[cv * N + e]; is[e] = invstd[cv * N + e];
        if (APPLY) { k1[e] = c1[cv * N + e]; k2[e] = c2[cv * N + e]; }
        a1[e] = a2[e] = 0.f;
    }
    constexpr int np = pool ? 4 : 1;
    for (long px = (long)blockIdx.x * nslot + slot; px < npx; px += (long)gridDim.x * nslot) {
        float g[N];
        Vec<T>::load(dout + px * C + cv * N, g);
        const T* ybase;
        if (pool) {
            // 32-bit unsigned division (B*PH*PW < 2^32): a third of the instructions of the 64-bit one
            const unsigned pxu = (unsigned)px, rowi = pxu / (unsigned)PW;
            const int pw = (int)(pxu - rowi * (unsigned)PW);
            const long b = rowi / (unsigned)PH;
            const int ph = (int)(rowi - (unsigned)b * (unsigned)PH);
            ybase = y + ((b * H + 2 * ph) * W + 2 * pw) * C + cv * N;
        } else {
            ybase = y + px * C + cv * N;
        }
        float v[np][N], z[np][N];
#pragma unroll
        for (int k = 0; k < np; ++k) {
            Vec<T>::load(ybase + ((long)(k >> 1) * W + (k & 1)) * C, v[k]);
#pragma unroll
            for (int e = 0; e < N; ++e) z[k][e] = fmaf(v[k][e], sc[e], sh[e]);
        }
#pragma unroll
        for (int k = 0; k < np; ++k) {
            float dz[N];
#pragma unroll
            for (int e = 0; e < N; ++e) {
                float d;
                if (pool == 0) {
                    d = z[0][e] > 0.f ? g[e] : 0.f;
                } else if (pool == 2) {
                    d = z[k][e] > 0.f ? 0.25f * g[e] : 0.f;
                } else {
                    const float m = fmaxf(fmaxf(z[0][e], z[1 % np][e]), fmaxf(z[2 % np][e], z[3 % np][e]));
                    bool first = z[k][e] == m;
#pragma unroll
                    for (int j = 0; j < k; ++j) first = first && (z[j][e] != m);
                    d = (m > 0.f && first) ? g[e] : 0.f;
                }
                dz[e] = d;
            }
            if (APPLY) {
                float o[N];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const float xh = (v[k][e] - mu[e]) * is[e];
                    o[e] = sc[e] * (dz[e] - k1[e] - xh * k2[e]);
                }
                Vec<T>::store(dy + (ybase - y) + ((long)(k >> 1) * W + (k & 1)) * C, o);
            } else {
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    a1[e] += dz[e];
                    a2[e] += dz[e] * ((v[k][e] - mu[e]) * is[e]);
                }
            }
        }
    }
    if (APPLY && pool) {
        // floor-mode pooling drops the last row / column of an odd map: dz = 0 there, dy still gets the mean terms
        const int lw = (W & 1) ? H : 0, lh = (H & 1) ? (W - (W & 1)) : 0, nl = lw + lh;
        for (long i = (long)blockIdx.x * nslot + slot; i < (long)B * nl; i += (long)gridDim.x * nslot) {
            const long b = i / nl;
            const int j = i % nl;
            const int h = j < lw ? j : H - 1, w = j < lw ? W - 1 : j - lw;
            const long off = ((b * H + h) * W + w) * C + cv * N;
            float v[N], o[N];
            Vec<T>::load(y + off, v);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = sc[e] * (-k1[e] - (v[e] - mu[e]) * is[e] * k2[e]);
            Vec<T>::store(dy + off, o);
        }
    }
    if (!APPLY) {
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
        for (int i = threadIdx.x; i < C * 2; i += blockDim.x) {
            const float t = rw[i] + rw[per_wave + i] + rw[2 * per_wave + i] + rw[3 * per_wave + i];
            part[(long)blockIdx.x * C * 2 + i] = t;
        }
    }
}

__global__ void bn_bwd_finalize(const float* part, int rows, int C, double count, float* dgamma, float* dbeta,
                                float* c1, float* c2) {
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
        s1 += (double)part[((long)i * C + c) * 2];
        s2 += (double)part[((long)i * C + c) * 2 + 1];
    }
    __shared__ double r1[4], r2[4];
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = r1[0] + r1[1] + r1[2] + r1[3];
        s2 = r2[0] + r2[1] + r2[2] + r2[3];
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        c1[c] = (float)(s1 / count);
        c2[c] = (float)(s2 / count);
    }
}

int grid_for(long items, int per_block) {
    long g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int vqa_bn_finalize(void* stream, const float* part, int rows, int C, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* mean, float* invstd, float* scale, float* shift) {
    if (rows < 1 || C < 1) return VQA_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), part, rows, C, count, gamma,
                       beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
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

extern "C" int vqa_bn_act_pool_fwd(void* stream, int dtype, const void* y, void* a, const float* scale,
                                   const float* shift, int B, int H, int W, int C, int pool) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long out_px = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    if (dtype == VQA_BF16) {
        if (C % 8 || 256 % (C / 8)) return VQA_ERR_ARG;     // a thread keeps one channel vector (C/8 divides 256)
        hipLaunchKernelGGL(bn_act_pool_fwd<bf16>, dim3(grid_for(out_px * (C / 8), 256)), dim3(256), 0, s,
                           static_cast<const bf16*>(y), static_cast<bf16*>(a), scale, shift, B, H, W, C, pool);
    } else if (dtype == VQA_F32) {
        if (C % 4 || 256 % (C / 4)) return VQA_ERR_ARG;
        hipLaunchKernelGGL(bn_act_pool_fwd<float>, dim3(grid_for(out_px * (C / 4), 256)), dim3(256), 0, s,
                           static_cast<const float*>(y), static_cast<float*>(a), scale, shift, B, H, W, C, pool);
    } else {
        return VQA_ERR_ARG;
    }
    CHECK_LAUNCH();
    return VQA_OK;
}

// rows of `part` the reduce pass writes (= its grid size)
extern "C" int vqa_bn_bwd_rows(int dtype, int B, int H, int W, int C, int pool) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    const int CV = C / N;
    const long npx = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    return grid_for(npx, (256 / CV) * 4);
}

extern "C" int vqa_bn_bwd(void* stream, int dtype, int apply, const void* y, const void* dout, void* dy,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          const float* c1, const float* c2, float* part, int B, int H, int W, int C, int pool) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    const int CV = C / N;
    if (C % N || CV > 64 || (CV & (CV - 1))) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = vqa_bn_bwd_rows(dtype, B, H, W, C, pool);
#define GO(T, AP, P) hipLaunchKernelGGL((bn_bwd<T, AP, P>), dim3(grid), dim3(256), 0, s, static_cast<const T*>(y), \
                                        static_cast<const T*>(dout), static_cast<T*>(dy), scale, shift, mean, invstd, \
                                        c1, c2, part, B, H, W, C)
#define GP(T, AP) do { if (pool == 0) GO(T, AP, 0); else if (pool == 1) GO(T, AP, 1); else GO(T, AP, 2); } while (0)
    if (pool < 0 || pool > 2 || C > 256) return VQA_ERR_ARG;
    if (dtype == VQA_BF16) { if (apply) GP(bf16, true); else GP(bf16, false); }
    else if (dtype == VQA_F32) { if (apply) GP(float, true); else GP(float, false); }
    else return VQA_ERR_ARG;
#undef GP
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
