// Small HBM-bound kernels around the MFMA GEMMs: weight_norm, embedding, GRU gates, attention softmax/pooling,
// head elementwise pieces, soft-score BCE, soft accuracy, grad-norm clip + Adamax.
// Reference call sites: gru.py:48-52,101-106; vqa_model.py:59-75 (weight_norm), :83-103 (head), :215 (loss),
// :127-132 (accuracy), :259/:218/:266 (clip + Adamax).
#include "common.h"
#include <atomic>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T> __device__ __forceinline__ float ld(const T* p, long i) { return to_f32(p[i]); }

// ---------------------------------------------------------------------------------------- reductions
// out[0] = sum_i a[i]*b[i]   (two launches: partials in double, then a fixed-order final sum)
__global__ void dot_partial(const float* a, const float* b, long n, double* part) {
    double s = 0.0;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) {
        const long n4 = n >> 2;
        for (long i = tid; i < n4; i += nth) {
            const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
            s += (double)(x.x * y.x + x.y * y.y) + (double)(x.z * y.z + x.w * y.w);
        }
        for (long i = (n4 << 2) + tid; i < n; i += nth) s += (double)a[i] * (double)b[i];
    } else {
        for (long i = tid; i < n; i += nth) s += (double)a[i] * (double)b[i];
    }
    __shared__ double r[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = r[0] + r[1] + r[2] + r[3];
}
__global__ void dot_final(const double* part, int n, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}

// out[n] = sum_m X[m][n].  Block = 64 columns x 4 row-lanes; grid.y row slabs are combined in a second launch.
template <typename T>
__global__ void colsum(const T* x, int M, int N, int ldx, float* out, int rows_per_slab) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;   // 4 row lanes
    const int m0 = blockIdx.y * rows_per_slab;
    const int m1 = m0 + rows_per_slab < M ? m0 + rows_per_slab : M;
    float s = 0.f;
    if (n < N)
        for (int m = m0 + part; m < m1; m += 4) s += to_f32(x[(long)m * ldx + n]);
    __shared__ float r[4][64];
    r[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && n < N)
        out[(long)blockIdx.y * N + n] = r[0][threadIdx.x] + r[1][threadIdx.x] + r[2][threadIdx.x] + r[3][threadIdx.x];
}

// The same with 16-byte loads: a block is CL = 256 / RL column lanes (each 16 bytes of one row: 8 bf16 / 4 f32 columns)
// x RL row lanes.  The scalar form above moves 2 bytes per lane (128 bytes per wave load): 22 MB of bf16 gate gradients
// took 23 us.  RL = 4 sums every column in the scalar form's order (row lane = m mod 4, lanes combined left to right):
// same bits; RL = 16 is for short matrices summed by ONE launch (M <= 1024: no slab pass).
template <typename T, int RL>
__global__ __launch_bounds__(256) void colsum_vec(const T* x, int M, int N, int ldx, float* out, int rows_per_slab) {
    constexpr int VEC = 16 / (int)sizeof(T), CL = 256 / RL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int n0 = (blockIdx.x * CL + cl) * VEC;
    const int m0 = blockIdx.y * rows_per_slab;
    const int m1 = m0 + rows_per_slab < M ? m0 + rows_per_slab : M;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (n0 < N) {
        const T* p = x + n0;
#pragma unroll 4
        for (int m = m0 + rl; m < m1; m += RL) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + (long)m * ldx);
            const T* t = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += to_f32(t[e]);
        }
    }
    __shared__ float r[RL][CL * VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) r[rl][cl * VEC + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < CL * VEC; c += 256) {
        const int n = blockIdx.x * CL * VEC + c;
        if (n < N) {
            float t = r[0][c];
#pragma unroll
            for (int k = 1; k < RL; ++k) t += r[k][c];
            out[(long)blockIdx.y * N + n] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------- weight norm
// W = v * (g / sqrt(sumsq)); writes an MFMA-dtype copy [out][ldw] and a transposed copy [in][ldt]
template <typename T>
__global__ void wn_apply(const float* v, const float* g, const float* sumsq, int out, int in, T* w, int ldw, T* wt,
                         int ldt) {
    __shared__ float tile[32][33];
    const float s = g[0] / sqrtf(sumsq[0]);
    const int o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + tx;
        float x = 0.f;
        if (o < out && i < in) {
            x = v[(long)o * in + i] * s;
            w[(long)o * ldw + i] = from_f32<T>(x);
        }
        tile[r][tx] = x;
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, o = o0 + tx;
            if (i < in && o < out) wt[(long)i * ldt + o] = from_f32<T>(tile[tx][r]);
        }
    }
}
// dv = (g/n) * (dW - (dot/n^2) v),  dg = dot / n,  n = sqrt(sumsq), dot = <dW, v>
__global__ void wn_bwd(const float* dw, const float* v, const float* g, const float* sumsq, const float* dot, long n,
                       float* dv, float* dg) {
    const float nrm = sqrtf(sumsq[0]);
    const float a = g[0] / nrm, b = dot[0] / sumsq[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dv[i] = a * (dw[i] - b * v[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) dg[0] = dot[0] / nrm;
}

// ---- all weight-normed layers of the head in two launches per direction (was 3 per layer per direction) --------
// Launch 1 (wn_dots): grid (WN_SLABS, n): fixed-order fp64 partials of <v,v> (forward) or <dW,v> (backward).
// Launch 2: forward wn_apply_all - every block re-sums its layer's WN_SLABS partials in one fixed order, then scales
//           its 32x32 tile (and the transposed copy); backward wn_bwd_all - the same for dv / dg.
constexpr int WN_MAX = 8, WN_SLABS = 256;
struct WnJobs {
    const float* v[WN_MAX];
    const float* g[WN_MAX];
    const float* dw[WN_MAX];      // backward: dW
    void* w[WN_MAX];              // forward: W [out][ldw]
    void* wt[WN_MAX];             // forward: W^T [in][ldt] or null
    float* dv[WN_MAX];
    float* dg[WN_MAX];
    int out[WN_MAX], in[WN_MAX], ldw[WN_MAX], ldt[WN_MAX], tile0[WN_MAX + 1];
    float* sumsq;                 // [n] forward: written; backward: read
    double* part;                 // [n][WN_SLABS]
    int n;
};

template <bool BWD>
__global__ __launch_bounds__(256) void wn_dots(WnJobs j) {
    const int job = blockIdx.y;
    const float* a = BWD ? j.dw[job] : j.v[job];
    const float* b = j.v[job];
    const long n = (long)j.out[job] * j.in[job];
    double s = 0.0;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) {
        const long n4 = n >> 2;
        for (long i = tid; i < n4; i += nth) {
            const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
            s += (double)(x.x * y.x + x.y * y.y) + (double)(x.z * y.z + x.w * y.w);
        }
        for (long i = (n4 << 2) + tid; i < n; i += nth) s += (double)a[i] * (double)b[i];
    } else {
        for (long i = tid; i < n; i += nth) s += (double)a[i] * (double)b[i];
    }
    __shared__ double r[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) j.part[job * WN_SLABS + blockIdx.x] = r[0] + r[1] + r[2] + r[3];
}

// every thread of the block gets the layer's fixed-order sum of partials
__device__ __forceinline__ float wn_total(const double* part, int job) {
    __shared__ float tot;
    if (threadIdx.x < 64) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < WN_SLABS / 64; ++i) s += part[job * WN_SLABS + i * 64 + threadIdx.x];
        s = wave_sum_d(s);
        if (threadIdx.x == 0) tot = (float)s;
    }
    __syncthreads();
    return tot;
}

template <typename T>
__global__ __launch_bounds__(256) void wn_apply_all(WnJobs j) {
    int job = 0;
    while (job + 1 < j.n && (int)blockIdx.x >= j.tile0[job + 1]) ++job;
    const int tile = blockIdx.x - j.tile0[job];
    const int out = j.out[job], in = j.in[job], ldw = j.ldw[job], ldt = j.ldt[job];
    const int tiles_x = (in + 31) / 32;
    const float sumsq = wn_total(j.part, job);
    if (tile == 0 && threadIdx.x == 0) j.sumsq[job] = sumsq;
    const float* v = j.v[job];
    T* w = static_cast<T*>(j.w[job]);
    T* wt = static_cast<T*>(j.wt[job]);
    __shared__ float tl[32][33];
    const float s = j.g[job][0] / sqrtf(sumsq);
    const int o0 = (tile / tiles_x) * 32, i0 = (tile % tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, i = i0 + tx;
        float x = 0.f;
        if (o < out && i < in) {
            x = v[(long)o * in + i] * s;
            w[(long)o * ldw + i] = from_f32<T>(x);
        }
        tl[r][tx] = x;
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, o = o0 + tx;
            if (i < in && o < out) wt[(long)i * ldt + o] = from_f32<T>(tl[tx][r]);
        }
    }
}

__global__ __launch_bounds__(256) void wn_bwd_all(WnJobs j) {
    const int job = blockIdx.y;
    const float dot = wn_total(j.part, job);
    const float sumsq = j.sumsq[job];
    const float nrm = sqrtf(sumsq);
    const float a = j.g[job][0] / nrm, b = dot / sumsq;
    const long n = (long)j.out[job] * j.in[job];
    const float* dw = j.dw[job];
    const float* v = j.v[job];
    float* dv = j.dv[job];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dv[i] = a * (dw[i] - b * v[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) j.dg[job][0] = dot / nrm;
}

// cast / pad: dst[m][0..ldd) = src[m][0..N) then zeros.  VEC4: four consecutive elements per thread (N, lds, ldd
// multiples of 4 and 16-byte aligned bases: the entry checks), else one.
template <typename TS, typename TD, bool VEC4>
__global__ void cast_pad(const TS* __restrict__ src, int M, int N, int lds, TD* __restrict__ dst, int ldd,
                         const float* __restrict__ scale) {
    constexpr int V = VEC4 ? 4 : 1;
    const float sc = scale ? scale[0] : 1.f;
    const int ldv = ldd / V;
    const long total = (long)M * ldv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % ldv) * V;
        const long m = i / ldv;
        if constexpr (VEC4) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
                if constexpr (sizeof(TS) == 4) {
                    load4(reinterpret_cast<const float*>(src) + m * lds + n, v);
                } else {
                    const bf16x4 t = *reinterpret_cast<const bf16x4*>(src + m * lds + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
                }
            }
            if (scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= sc;
            }
            store4<TD>(dst + m * ldd + n, v);
        } else {
            dst[m * ldd + n] = from_f32<TD>(n < N ? to_f32(src[m * lds + n]) * sc : 0.f);
        }
    }
}

// dst[n][m] = src[m][n] in the destination dtype, columns m in [M, ldd) zero: the transposed, cast copy of an f32 master
// weight (the GRU's W_ih^T for its input gradient) through a 32 x 33 LDS tile - both sides move whole row segments.
template <typename TD>
__global__ __launch_bounds__(256) void cast_transpose(const float* __restrict__ src, int M, int N, int lds,
                                                      TD* __restrict__ dst, int ldd) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, n = n0 + tx;
        tile[r][tx] = (m < M && n < N) ? src[(long)m * lds + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, m = m0 + tx;
        if (n < N && m < ldd) dst[(long)n * ldd + m] = from_f32<TD>(tile[tx][r]);
    }
}

// nbytes (a multiple of 4) of zeros from a 4-byte aligned address: the step's own fill (embedding gradient, unwritten
// gradient slots) - 16-byte stores from the first 16-byte boundary on
__global__ void zero_fill(unsigned* __restrict__ p, long nwords) {
    const long head = (4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3;       // words before the boundary
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long h = head < nwords ? head : nwords;
    if (i < h) p[i] = 0u;
    const long nvec = (nwords - h) / 4;
    uint4* v = reinterpret_cast<uint4*>(p + h);
    for (long k = i; k < nvec; k += (long)gridDim.x * blockDim.x) v[k] = make_uint4(0u, 0u, 0u, 0u);
    const long tail0 = h + nvec * 4;
    if (i < nwords - tail0) p[tail0 + i] = 0u;
}

// ---------------------------------------------------------------------------------------- embedding
// X[t][b][0..ldx) = emb[q[b][t]][0..E), zero padded   (time-major so every GRU step reads a contiguous slab)
// A token id outside [0, V) (nn.Embedding raises on it) reads nothing: its row is filled with NaN, which reaches the
// loss and fails the step visibly instead of faulting the GPU; the backward skips it.
template <typename T>
__global__ void embed_fwd(const long* q, const float* emb, int B, int Tn, int E, int V, T* x, int ldx) {
    const int row = blockIdx.x;          // t*B + b
    const int t = row / B, b = row % B;
    const long id = q[(long)b * Tn + t];
    const bool ok = id >= 0 && id < V;
    for (int e = threadIdx.x; e < ldx; e += blockDim.x)
        x[(long)row * ldx + e] = from_f32<T>(e < E ? (ok ? emb[id * E + e] : NAN) : 0.f);
}
// ---------------------------------------------------------------------------------------- GRU gates
// gi, gh: [B][3H] (gate rows r,z,n; biases already added).  h' = (1-z) n + z h.   saved: r, z, n, hn = gh_n.
// One thread = 4 consecutive hidden units of one sample (float4 loads, 8- or 16-byte stores; H % 4 == 0): with one
// unit per thread and 2-byte scattered stores the backward kernel took 16 us per time step for ~10 MB.
template <typename T>
__global__ void gru_gate_fwd(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hprev,
                             int B, int H, float* __restrict__ hnew, T* __restrict__ hnew_t,
                             float* __restrict__ save /* [B][4H] */) {
    const int HV = H / 4;
    const long total = (long)B * HV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % HV) * 4;
        const long b = i / HV;
        const float* gib = gi + b * 3 * H + j;
        const float* ghb = gh + b * 3 * H + j;
        float ir[4], iz[4], in[4], hr[4], hz[4], hn[4], hp[4], r[4], z[4], n[4], h[4];
        load4(gib, ir); load4(gib + H, iz); load4(gib + 2 * H, in);
        load4(ghb, hr); load4(ghb + H, hz); load4(ghb + 2 * H, hn);
        load4(hprev + b * H + j, hp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[e] = sigmoidf_(ir[e] + hr[e]);
            z[e] = sigmoidf_(iz[e] + hz[e]);
            n[e] = tanhf(in[e] + r[e] * hn[e]);
            h[e] = (1.f - z[e]) * n[e] + z[e] * hp[e];
        }
        store4<float>(hnew + b * H + j, h);
        store4<T>(hnew_t + b * H + j, h);
        float* sv = save + b * 4 * H + j;
        store4<float>(sv, r); store4<float>(sv + H, z); store4<float>(sv + 2 * H, n); store4<float>(sv + 3 * H, hn);
    }
}
// dgi = [dr, dz, dn], dgh = [dr, dz, dn*r]  (MFMA dtype, [B][3H]);  dhprev = dh * z  (the W_hh term is added by a GEMM)
template <typename T>
__global__ void gru_gate_bwd(const float* __restrict__ dh, const float* __restrict__ save,
                             const float* __restrict__ hprev, int B, int H, T* __restrict__ dgi, T* __restrict__ dgh,
                             float* __restrict__ dhprev, const float* __restrict__ dh_add) {
    const int HV = H / 4;
    const long total = (long)B * HV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % HV) * 4;
        const long b = i / HV;
        const float* sv = save + b * 4 * H + j;
        float r[4], z[4], n[4], hn[4], g[4], hp[4], dr[4], dz[4], dn[4], dnr[4], dp[4];
        load4(sv, r); load4(sv + H, z); load4(sv + 2 * H, n); load4(sv + 3 * H, hn);
        load4(dh + b * H + j, g);
        load4(hprev + b * H + j, hp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dn[e] = g[e] * (1.f - z[e]) * (1.f - n[e] * n[e]);
            dz[e] = g[e] * (hp[e] - n[e]) * z[e] * (1.f - z[e]);
            dr[e] = dn[e] * hn[e] * r[e] * (1.f - r[e]);
            dnr[e] = dn[e] * r[e];
            dp[e] = g[e] * z[e];
        }
        if (dh_add) {      // multi-layer GRU: the layer above also feeds on h_{t-1} (its input-projection gradient)
            float ad[4];
            load4(dh_add + b * H + j, ad);
#pragma unroll
            for (int e = 0; e < 4; ++e) dp[e] += ad[e];
        }
        T* a = dgi + b * 3 * H + j;
        T* c = dgh + b * 3 * H + j;
        store4<T>(a, dr); store4<T>(a + H, dz); store4<T>(a + 2 * H, dn);
        store4<T>(c, dr); store4<T>(c + H, dz); store4<T>(c + 2 * H, dnr);
        store4<float>(dhprev + b * H + j, dp);
    }
}

// ---------------------------------------------------------------------------------------- attention
// One workgroup (256 threads) per batch element: s_k = a[b,k,:].w_s ; p = softmax_k(s) ; vhat = sum_k p_k v[b,k,:]
// Every access to `a` is a 16-byte vector per lane (H % (16 / sizeof(T)) == 0, D % 4 == 0: checked by the entry):
// with 2-byte lane accesses these two kernels took 40 + 44 us for 13 MB.
template <typename T> struct Vec16 {
    static constexpr int N = 16 / sizeof(T);
    static __device__ __forceinline__ void load(const T* p, float* f) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = to_f32(e[i]);
    }
    static __device__ __forceinline__ void store(T* p, const float* f) {
        T e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = from_f32<T>(f[i]);
        *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(e);
    }
};
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd(const T* __restrict__ a, const T* __restrict__ ws,
                                                const float* __restrict__ v, int K, int H, int D, float* p, float* vhat,
                                                T* vhat_t) {
    constexpr int N = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // s[K], then the weighted-sum combine
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HV = H / N;
    // scores: wave w takes rows w, w+4, ..., two rows per round so that four or more loads are in flight per lane
    for (int k = wave; k < K; k += 8) {
        const bool two = k + 4 < K;
        const T* row0 = a + ((long)b * K + k) * H;
        const T* row1 = a + ((long)b * K + (two ? k + 4 : k)) * H;
        float s0 = 0.f, s1 = 0.f;
        for (int jv = lane; jv < HV; jv += 64) {
            float x0[N], x1[N], w[N];
            Vec16<T>::load(row0 + jv * N, x0);
            Vec16<T>::load(row1 + jv * N, x1);
            Vec16<T>::load(ws + jv * N, w);
#pragma unroll
            for (int e = 0; e < N; ++e) { s0 += x0[e] * w[e]; s1 += x1[e] * w[e]; }
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) { sm[k] = s0; if (two) sm[k + 4] = s1; }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, sm[k]);
    float den = 0.f;
    for (int k = 0; k < K; ++k) den += __expf(sm[k] - mx);
    const float inv = 1.f / den;
    // weighted sum: thread = (4-channel vector dv, row residue kr): K rows split over 256 / (D/4) residues, then
    // combined through LDS in a fixed order
    const int DV = D / 4, nres = 256 / DV > 0 ? 256 / DV : 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int dv = threadIdx.x % DV, kr = threadIdx.x / DV;
    if (kr < nres) {
        for (int k = kr; k < K; k += nres) {
            const float pk = __expf(sm[k] - mx) * inv;
            const float4 x = *reinterpret_cast<const float4*>(v + ((long)b * K + k) * D + dv * 4);
            acc.x += pk * x.x; acc.y += pk * x.y; acc.z += pk * x.z; acc.w += pk * x.w;
        }
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) p[(long)b * K + k] = __expf(sm[k] - mx) * inv;
    __syncthreads();                                            // sm[] is reused for the combine
    float4* part = reinterpret_cast<float4*>(sm);               // [nres][DV]   (entry sizes LDS for it)
    if (kr < nres) part[kr * DV + dv] = acc;
    __syncthreads();
    if (threadIdx.x < DV) {
        float4 t = part[threadIdx.x];
        for (int i = 1; i < nres; ++i) {
            const float4 u = part[i * DV + threadIdx.x];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(vhat + (long)b * D + threadIdx.x * 4) = t;
        const float o[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) vhat_t[(long)b * D + threadIdx.x * 4 + e] = from_f32<T>(o[e]);
    }
}
// dp_k = v[b,k,:].dvhat[b];  ds_k = p_k (dp_k - sum_j p_j dp_j);  dv[b,k,:] = p_k dvhat[b];
// dpre[b,k,:] = ds_k * w_s * 1[a>0];  dws_part[b][:] = sum_k ds_k a[b,k,:]
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd(const T* __restrict__ a, const T* __restrict__ ws,
                                                const float* __restrict__ v, const float* __restrict__ p,
                                                const float* __restrict__ dvhat, int K, int H, int D,
                                                float* __restrict__ dv, T* __restrict__ dpre,
                                                float* __restrict__ dws_part, int act) {
    constexpr int N = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // ds[K], then the dws combine
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int DV = D / 4;
    for (int k = wave; k < K; k += 4) {
        float s = 0.f;
        for (int d4 = lane; d4 < DV; d4 += 64) {
            const float4 x = *reinterpret_cast<const float4*>(v + ((long)b * K + k) * D + d4 * 4);
            const float4 g = *reinterpret_cast<const float4*>(dvhat + (long)b * D + d4 * 4);
            s += x.x * g.x + x.y * g.y + x.z * g.z + x.w * g.w;
        }
        s = wave_sum(s);
        if (lane == 0) sm[k] = s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += p[(long)b * K + k] * sm[k];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) sm[k] = p[(long)b * K + k] * (sm[k] - dot);
    __syncthreads();
    // dv rows: one float4 per thread and step
    for (int i = threadIdx.x; i < K * DV; i += blockDim.x) {
        const int k = i / DV, d4 = i - k * DV;
        const float pk = p[(long)b * K + k];
        const float4 g = *reinterpret_cast<const float4*>(dvhat + (long)b * D + d4 * 4);
        *reinterpret_cast<float4*>(dv + ((long)b * K + k) * D + d4 * 4) = make_float4(pk * g.x, pk * g.y, pk * g.z, pk * g.w);
    }
    // dpre / dws: thread = (16-byte channel vector jv, row residue kr); each 16-byte vector of `a` is read once
    const int HV = H / N;
    float* comb = sm + ((K + 3) / 4) * 4;                       // [nres][H] partial dws (entry sizes LDS for it)
    for (int j0 = 0; j0 < HV; j0 += 256) {                      // (one round for H <= 256 * N)
        const int span = HV - j0 < 256 ? HV - j0 : 256;         // vectors handled in this round
        const int nres = 256 / span > 0 ? 256 / span : 1;
        const int jl = threadIdx.x % span, kr = threadIdx.x / span, jv = j0 + jl;
        float acc[N], w[N];
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = 0.f;
        if (kr < nres) {
            Vec16<T>::load(ws + jv * N, w);
            for (int k = kr; k < K; k += nres) {
                float av[N], o[N];
                Vec16<T>::load(a + ((long)b * K + k) * H + jv * N, av);
                const float dsk = sm[k];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    acc[e] += dsk * av[e];
                    o[e] = dsk * w[e] * act_grad(av[e], act);
                }
                Vec16<T>::store(dpre + ((long)b * K + k) * H + jv * N, o);
            }
#pragma unroll
            for (int e = 0; e < N; ++e) comb[(kr * span + jl) * N + e] = acc[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < span * N; i += blockDim.x) {
            float t = comb[i];
            for (int r2 = 1; r2 < nres; ++r2) t += comb[r2 * span * N + i];
            dws_part[(long)b * H + j0 * N + i] = t;
        }
        __syncthreads();
    }
}
// out[b][:] = sum_k x[b*K + k][:]
// The same with one 16-byte channel vector per thread and five rows in flight per round (the scalar form below read
// K rows of 2 bytes one after the other: K dependent round trips); rows are added in the same order k = 0 .. K-1.
template <typename T>
__global__ __launch_bounds__(256) void group_sum_vec(const T* __restrict__ x, int B, int K, int H, float* __restrict__ out,
                                                     T* __restrict__ out_t) {
    constexpr int N = Vec16<T>::N, U = 5;
    const int HV = H / N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * HV) return;
    const int b = i / HV, jv = i - b * HV;
    const T* base = x + ((long)b * K) * H + jv * N;
    float s[N];
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += U) {
        float v[U][N];
#pragma unroll
        for (int u = 0; u < U; ++u) Vec16<T>::load(base + (long)(k0 + u < K ? k0 + u : K - 1) * H, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u < K) {
#pragma unroll
                for (int e = 0; e < N; ++e) s[e] += v[u][e];
            }
    }
    float* o = out + (long)b * H + jv * N;
#pragma unroll
    for (int e = 0; e < N; e += 4) *reinterpret_cast<float4*>(o + e) = make_float4(s[e], s[e + 1], s[e + 2], s[e + 3]);
    Vec16<T>::store(out_t + (long)b * H + jv * N, s);
}
template <typename T>
__global__ void group_sum(const T* x, int B, int K, int H, float* out, T* out_t) {
    const long total = (long)B * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = i % H;
        const long b = i / H;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += to_f32(x[(b * K + k) * H + j]);
        out[i] = s;
        out_t[i] = from_f32<T>(s);
    }
}

// ---------------------------------------------------------------------------------------- head elementwise
// m0 = fv * fq * mask + add  (either may be null).  nn.Dropout: mask = keep / (1-p), no add.  nn.AlphaDropout
// (vqa_model.py:55, the selu branch): a * (x * keep + alpha' * (1 - keep)) + b = x * (a * keep) + (a * alpha' *
// (1 - keep) + b): mask = a * keep, add = the second term; the backward needs the mask only.
template <typename T>
__global__ void hadamard_fwd(const float* fv, const float* fq, const float* mask, const float* add, long n, T* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(fv[i] * fq[i] * (mask ? mask[i] : 1.f) + (add ? add[i] : 0.f));
}
// dpre_v = dm0*mask*fq*act'(fv), dpre_q = dm0*mask*fv*act'(fq)     (fv, fq are post-activation; act' from the output)
template <typename T>
__global__ void hadamard_bwd(const float* dm0, const float* fv, const float* fq, const float* mask, long n, T* dpv,
                             T* dpq, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = dm0[i] * (mask ? mask[i] : 1.f);
        dpv[i] = from_f32<T>(g * fq[i] * act_grad(fv[i], act));
        dpq[i] = from_f32<T>(g * fv[i] * act_grad(fq[i], act));
    }
}
// out = g * mask * act'(y)      (y = the activation's output)
template <typename T>
__global__ void relu_bwd(const float* g, const float* y, const float* mask, long n, T* out, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(g[i] * (mask ? mask[i] : 1.f) * act_grad(y[i], act));
}
// out = x * mask + add (MFMA-dtype copy; see hadamard_fwd)
template <typename T>
__global__ void mask_mul(const float* x, const float* mask, const float* add, long n, T* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(x[i] * (mask ? mask[i] : 1.f) + (add ? add[i] : 0.f));
}

// ---------------------------------------------------------------------------------------- loss / accuracy
// loss = sum max(x,0) - x t + log1p(exp(-|x|)) ; dx = sigmoid(x) - t     (logits [B][C] with row stride ldx)
// (between the head's forward and backward: four elements per thread and round are loaded before any is used - a round
// trip per round, not per element - and the index arithmetic is 32-bit; the entry checks B*C < 2^31)
__global__ __launch_bounds__(256) void bce_partial(const float* __restrict__ x, const float* __restrict__ t, int B, int C,
                                                   int ldx, float* __restrict__ dx, double* part) {
    double s = 0.0;
    const int total = B * C, nth = gridDim.x * 256;
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * nth) {
        float xv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nth;
            const int ic = i < total ? i : total - 1;
            const int b = ic / C, c = ic - b * C;
            xv[u] = x[(long)b * ldx + c];
            tv[u] = t[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nth;
            if (i < total) {
                const float e = __expf(-fabsf(xv[u]));
                s += (double)(fmaxf(xv[u], 0.f) - xv[u] * tv[u] + log1pf(e));
                if (dx) {
                    const float sg = xv[u] >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
                    dx[i] = sg - tv[u];
                }
            }
        }
    }
    __shared__ double r[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = r[0] + r[1] + r[2] + r[3];
}
// torch.argmax over one row by one wave: the FIRST maximum, NaN counting as larger than everything (so a row of NaN
// or of -inf - a diverged model reaching evaluate() - still yields a valid index, 0 .. C-1, as the reference's
// torch.argmax does at vqa_model.py:127, and never an out-of-range gather).
__device__ __forceinline__ bool arg_gt(float a, float b) { return a > b || (a != a && b == b); }
__device__ __forceinline__ int row_argmax(const float* xr, int C, int lane) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (lane < C) { best = xr[lane]; bi = lane; }
    for (int c = lane + 64; c < C; c += 64) {
        const float v = xr[c];
        if (arg_gt(v, best)) { best = v; bi = c; }
    }
    for (int o = 1; o < 64; o <<= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        const bool same = ob == best || (ob != ob && best != best);
        if (arg_gt(ob, best) || (same && oi < bi)) { best = ob; bi = oi; }
    }
    return (unsigned)bi < (unsigned)C ? bi : 0;
}
// acc_part[b] = target[b][argmax_c logits[b][c]]
__global__ void soft_acc_rows(const float* x, const float* t, int C, int ldx, float* out) {
    const int b = blockIdx.x, lane = threadIdx.x;   // one wave per row
    const int bi = row_argmax(x + (long)b * ldx, C, lane);
    if (lane == 0) out[b] = t[(long)b * C + bi];
}
// Sparse soft-score targets: row b has L slots (label, score), label < 0 = empty slot, labels distinct within a row -
// the (labels, scores) lists the reference densifies into a >= 99.6 %-zero [B][C] matrix (dataset.py:135-138).
// One workgroup per row: dense pass for the target-free terms, then the L slots.
__global__ __launch_bounds__(256) void bce_sparse_rows(const float* x, const long* labels, const float* scores, int C,
                                                       int L, int ldx, float* dx, double* rowloss) {
    const int b = blockIdx.x;
    const float* xr = x + (long)b * ldx;
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float xv = xr[c];
        const float e = __expf(-fabsf(xv));
        s += (double)(fmaxf(xv, 0.f) + log1pf(e));
        if (dx) dx[(long)b * C + c] = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    }
    __syncthreads();                                   // this block's dx row is written (same-block visibility)
    if ((int)threadIdx.x < L) {
        const long lb = labels[(long)b * L + threadIdx.x];
        if (lb >= 0 && lb < C) {
            const float sc = scores[(long)b * L + threadIdx.x];
            s -= (double)(xr[lb] * sc);
            if (dx) dx[(long)b * C + lb] -= sc;
        }
    }
    __shared__ double r[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowloss[b] = r[0] + r[1] + r[2] + r[3];
}
__global__ void soft_acc_rows_sparse(const float* x, const long* labels, const float* scores, int C, int L, int ldx,
                                     float* out) {
    const int b = blockIdx.x, lane = threadIdx.x;   // one wave per row
    const int bi = row_argmax(x + (long)b * ldx, C, lane);
    float sc = 0.f;
    for (int l = lane; l < L; l += 64)
        if (labels[(long)b * L + l] == bi) sc = scores[(long)b * L + l];
    sc = wave_sum(sc);                                // at most one slot matches (labels are distinct)
    if (lane == 0) out[b] = sc;
}
__global__ void sum_f32(const float* x, int n, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += (double)x[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}

// ---------------------------------------------------------------------------------------- clip + Adamax
// coef = min(1, max_norm / (sqrt(sqsum) + 1e-6))      (nn.utils.clip_grad_norm_)
// torch.optim.Adamax, weight_decay 0:  m = b1 m + (1-b1) g ; u = max(b2 u, |g| + eps) ; p -= lr/(1-b1^t) m/u
// part != null: the clip coefficient is computed here, by every workgroup alike, from the 1024 partial sums of
// |g|^2 (fixed order -> the same value everywhere); workgroup 0 publishes sq / coef / total_norm.  Two launches
// (final sum, coefficient) less at the end of every step.
__global__ __launch_bounds__(256) void adamax_step(float* p, const float* g, float* m, float* u, long n,
                                                   const double* part, float max_norm, float* sq, float* coef,
                                                   float* total_norm, float lr_t, float b1, float b2, float eps) {
    float c = 1.f;
    if (part) {
        __shared__ double r[4];
        double s = (part[threadIdx.x] + part[threadIdx.x + 256]) + (part[threadIdx.x + 512] + part[threadIdx.x + 768]);
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
        __syncthreads();
        const float sqsum = (float)(r[0] + r[1] + r[2] + r[3]);
        const float nrm = sqrtf(sqsum);
        const float cc = max_norm / (nrm + 1e-6f);
        c = cc < 1.f ? cc : 1.f;
        if (blockIdx.x == 0 && threadIdx.x == 0) { sq[0] = sqsum; coef[0] = c; total_norm[0] = nrm; }
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gv = g[i] * c;
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float uv = fmaxf(b2 * u[i], fabsf(gv) + eps);
        m[i] = mv;
        u[i] = uv;
        p[i] -= lr_t * mv / uv;
    }
}

int grid1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)
#define BY_T(dtype, CALL_BF, CALL_F32) \
    do { if ((dtype) == VQA_BF16) { CALL_BF; } else if ((dtype) == VQA_F32) { CALL_F32; } else return VQA_ERR_ARG; } while (0)

// out[0] = <a, b>; `part` holds 256 doubles of scratch
extern "C" int vqa_dot(void* stream, const float* a, const float* b, long n, double* part, float* out) {
    hipLaunchKernelGGL(dot_partial, dim3(256), dim3(256), 0, ST, a, b, n, part);
    hipLaunchKernelGGL(dot_final, dim3(1), dim3(64), 0, ST, part, 256, out);
    CHECK_LAUNCH();
    return VQA_OK;
}

// out[n] = sum_m x[m][n].  `slabs` (>= 1) row slabs are summed independently into scratch [slabs][N] and combined
// in fixed order; scratch may be NULL when slabs == 1.
extern "C" int vqa_colsum(void* stream, int dtype, const void* x, int M, int N, int ldx, float* out, float* scratch,
                          int slabs) {
    if (slabs < 1 || (slabs > 1 && !scratch)) return VQA_ERR_ARG;
    const int rps = (M + slabs - 1) / slabs;
    dim3 grid((N + 63) / 64, slabs);
    float* dst = slabs > 1 ? scratch : out;
    const int vec = dtype == VQA_BF16 ? 8 : 4;
    if ((dtype == VQA_BF16 || dtype == VQA_F32) && N % vec == 0 && ldx % vec == 0 &&
        (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // 16-byte loads: 4 row lanes per slab (the scalar form's summation order), or - one launch over a short matrix -
        // 16 row lanes
        const bool tall = slabs == 1 && M >= 64;
        const int cols = (tall ? 16 : 64) * vec;
        dim3 g((N + cols - 1) / cols, slabs);
        if (dtype == VQA_BF16) {
            if (tall) hipLaunchKernelGGL((colsum_vec<bf16, 16>), g, dim3(256), 0, ST, (const bf16*)x, M, N, ldx, dst, rps);
            else hipLaunchKernelGGL((colsum_vec<bf16, 4>), g, dim3(256), 0, ST, (const bf16*)x, M, N, ldx, dst, rps);
        } else {
            if (tall) hipLaunchKernelGGL((colsum_vec<float, 16>), g, dim3(256), 0, ST, (const float*)x, M, N, ldx, dst, rps);
            else hipLaunchKernelGGL((colsum_vec<float, 4>), g, dim3(256), 0, ST, (const float*)x, M, N, ldx, dst, rps);
        }
    } else {
    BY_T(dtype, hipLaunchKernelGGL(colsum<bf16>, grid, dim3(256), 0, ST, (const bf16*)x, M, N, ldx, dst, rps),
         hipLaunchKernelGGL(colsum<float>, grid, dim3(256), 0, ST, (const float*)x, M, N, ldx, dst, rps));
    }
    if (slabs > 1)
        hipLaunchKernelGGL(colsum<float>, dim3((N + 63) / 64, 1), dim3(256), 0, ST, scratch, slabs, N, N, out, slabs);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_wn_apply(void* stream, int dtype, const float* v, const float* g, const float* sumsq, int out,
                            int in, void* w, int ldw, void* wt, int ldt) {
    dim3 grid((in + 31) / 32, (out + 31) / 32);
    BY_T(dtype, hipLaunchKernelGGL(wn_apply<bf16>, grid, dim3(256), 0, ST, v, g, sumsq, out, in, (bf16*)w, ldw, (bf16*)wt, ldt),
         hipLaunchKernelGGL(wn_apply<float>, grid, dim3(256), 0, ST, v, g, sumsq, out, in, (float*)w, ldw, (float*)wt, ldt));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_wn_bwd(void* stream, const float* dw, const float* v, const float* g, const float* sumsq,
                          const float* dot, long n, float* dv, float* dg) {
    hipLaunchKernelGGL(wn_bwd, dim3(grid1d(n)), dim3(256), 0, ST, dw, v, g, sumsq, dot, n, dv, dg);
    CHECK_LAUNCH();
    return VQA_OK;
}

// n (<= 8) weight-normed layers at once.  All pointer / int arguments are HOST arrays of length n (device pointers
// inside); sumsq [n] f32 and part [n*256] f64 are device buffers.  dims = n x 4 ints (out, in, ldw, ldt).
extern "C" int vqa_wn_fwd_all(void* stream, int dtype, int n, const void* const* v, const void* const* g,
                              void* const* w, void* const* wt, const int* dims, float* sumsq, double* part) {
    if (n < 1 || n > WN_MAX || !v || !g || !w || !wt || !dims || !sumsq || !part) return VQA_ERR_ARG;
    WnJobs j{};
    j.n = n; j.sumsq = sumsq; j.part = part;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        j.v[i] = static_cast<const float*>(v[i]); j.g[i] = static_cast<const float*>(g[i]);
        j.w[i] = w[i]; j.wt[i] = wt[i];
        j.out[i] = dims[4 * i]; j.in[i] = dims[4 * i + 1]; j.ldw[i] = dims[4 * i + 2]; j.ldt[i] = dims[4 * i + 3];
        j.tile0[i] = tiles;
        tiles += ((j.in[i] + 31) / 32) * ((j.out[i] + 31) / 32);
    }
    j.tile0[n] = tiles;
    hipLaunchKernelGGL(wn_dots<false>, dim3(WN_SLABS, n), dim3(256), 0, ST, j);
    BY_T(dtype, hipLaunchKernelGGL(wn_apply_all<bf16>, dim3(tiles), dim3(256), 0, ST, j),
         hipLaunchKernelGGL(wn_apply_all<float>, dim3(tiles), dim3(256), 0, ST, j));
    CHECK_LAUNCH();
    return VQA_OK;
}

// backward of the same: dv[i] (f32, shape of v) and dg[i] (scalar) from dW[i]; dims = n x 2 ints (out, in)
extern "C" int vqa_wn_bwd_all(void* stream, int n, const void* const* dw, const void* const* v, const void* const* g,
                              void* const* dv, void* const* dg, const int* dims, const float* sumsq, double* part) {
    if (n < 1 || n > WN_MAX || !dw || !v || !g || !dv || !dg || !dims || !sumsq || !part) return VQA_ERR_ARG;
    WnJobs j{};
    j.n = n; j.sumsq = const_cast<float*>(sumsq); j.part = part;
    for (int i = 0; i < n; ++i) {
        j.dw[i] = static_cast<const float*>(dw[i]); j.v[i] = static_cast<const float*>(v[i]);
        j.g[i] = static_cast<const float*>(g[i]);
        j.dv[i] = static_cast<float*>(dv[i]); j.dg[i] = static_cast<float*>(dg[i]);
        j.out[i] = dims[2 * i]; j.in[i] = dims[2 * i + 1];
    }
    hipLaunchKernelGGL(wn_dots<true>, dim3(WN_SLABS, n), dim3(256), 0, ST, j);
    hipLaunchKernelGGL(wn_bwd_all, dim3(512, n), dim3(256), 0, ST, j);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_cast_pad(void* stream, int src_dtype, int dst_dtype, const void* src, int M, int N, int lds,
                            void* dst, int ldd, const float* scale) {
    const bool v4 = N % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    const long items = (long)M * (v4 ? ldd / 4 : ldd);
    dim3 grid((unsigned)((items + 255) / 256 > 0 ? (items + 255) / 256 : 1));     // one item per thread
#define GO(TS, TD)                                                                                                  \
    do {                                                                                                            \
        if (v4) hipLaunchKernelGGL((cast_pad<TS, TD, true>), grid, dim3(256), 0, ST, (const TS*)src, M, N, lds, (TD*)dst, ldd, scale); \
        else hipLaunchKernelGGL((cast_pad<TS, TD, false>), grid, dim3(256), 0, ST, (const TS*)src, M, N, lds, (TD*)dst, ldd, scale);   \
    } while (0)
    if (src_dtype == VQA_F32 && dst_dtype == VQA_BF16) GO(float, bf16);
    else if (src_dtype == VQA_F32 && dst_dtype == VQA_F32) GO(float, float);
    else if (src_dtype == VQA_BF16 && dst_dtype == VQA_F32) GO(bf16, float);
    else return VQA_ERR_ARG;
#undef GO
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_cast_transpose(void* stream, int dst_dtype, const float* src, int M, int N, int lds, void* dst,
                                  int ldd) {
    if (!src || !dst || M < 1 || N < 1 || lds < N || ldd < M) return VQA_ERR_ARG;
    const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((ldd + 31) / 32));
    BY_T(dst_dtype, hipLaunchKernelGGL(cast_transpose<bf16>, grid, dim3(256), 0, ST, src, M, N, lds, (bf16*)dst, ldd),
         hipLaunchKernelGGL(cast_transpose<float>, grid, dim3(256), 0, ST, src, M, N, lds, (float*)dst, ldd));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_zero(void* stream, void* p, long nbytes) {
    if (!p || nbytes < 0 || (nbytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3)) return VQA_ERR_ARG;
    if (nbytes == 0) return VQA_OK;
    const long nwords = nbytes / 4, nvec = nwords / 4 + 8;
    long blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill, dim3((unsigned)blocks), dim3(256), 0, ST, static_cast<unsigned*>(p), nwords);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_embed_fwd(void* stream, int dtype, const long* q, const float* emb, int B, int T, int E, int V,
                             void* x, int ldx) {
    if (B < 1 || T < 1 || E < 1 || V < 1 || ldx < E) return VQA_ERR_ARG;
    BY_T(dtype, hipLaunchKernelGGL(embed_fwd<bf16>, dim3(B * T), dim3(128), 0, ST, q, emb, B, T, E, V, (bf16*)x, ldx),
         hipLaunchKernelGGL(embed_fwd<float>, dim3(B * T), dim3(128), 0, ST, q, emb, B, T, E, V, (float*)x, ldx));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_gate_fwd(void* stream, int dtype, const float* gi, const float* gh, const float* hprev, int B,
                                int H, float* hnew, void* hnew_t, float* save) {
    if (H % 4) return VQA_ERR_ARG;
    dim3 grid((unsigned)(((long)B * (H / 4) + 255) / 256));
    BY_T(dtype, hipLaunchKernelGGL(gru_gate_fwd<bf16>, grid, dim3(256), 0, ST, gi, gh, hprev, B, H, hnew, (bf16*)hnew_t, save),
         hipLaunchKernelGGL(gru_gate_fwd<float>, grid, dim3(256), 0, ST, gi, gh, hprev, B, H, hnew, (float*)hnew_t, save));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_gate_bwd(void* stream, int dtype, const float* dh, const float* save, const float* hprev, int B,
                                int H, void* dgi, void* dgh, float* dhprev, const float* dh_add) {
    if (H % 4) return VQA_ERR_ARG;
    dim3 grid((unsigned)(((long)B * (H / 4) + 255) / 256));
    BY_T(dtype, hipLaunchKernelGGL(gru_gate_bwd<bf16>, grid, dim3(256), 0, ST, dh, save, hprev, B, H, (bf16*)dgi, (bf16*)dgh, dhprev, dh_add),
         hipLaunchKernelGGL(gru_gate_bwd<float>, grid, dim3(256), 0, ST, dh, save, hprev, B, H, (float*)dgi, (float*)dgh, dhprev, dh_add));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_attn_fwd(void* stream, int dtype, const void* a, const void* ws, const float* v, int B, int K,
                            int H, int D, float* p, float* vhat, void* vhat_t) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    if (K < 1 || H % N || D % 4 || D / 4 > 256) return VQA_ERR_ARG;
    const int nres = 256 / (D / 4);
    size_t lds = (size_t)K * sizeof(float);
    if (lds < (size_t)nres * D * sizeof(float)) lds = (size_t)nres * D * sizeof(float);
    BY_T(dtype, hipLaunchKernelGGL(attn_fwd<bf16>, dim3(B), dim3(256), lds, ST, (const bf16*)a, (const bf16*)ws, v, K, H, D, p, vhat, (bf16*)vhat_t),
         hipLaunchKernelGGL(attn_fwd<float>, dim3(B), dim3(256), lds, ST, (const float*)a, (const float*)ws, v, K, H, D, p, vhat, (float*)vhat_t));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_attn_bwd(void* stream, int dtype, const void* a, const void* ws, const float* v, const float* p,
                            const float* dvhat, int B, int K, int H, int D, float* dv, void* dpre, float* dws_part,
                            int act) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    if (K < 1 || H % N || D % 4 || (act != VQA_ACT_RELU && act != VQA_ACT_SELU)) return VQA_ERR_ARG;
    // ds[K] + the dws combine: per round min(H/N, 256) vectors x (256 / that) row residues x N floats <= 256 * N floats
    const size_t lds = ((size_t)(K + 3) / 4 * 4 + (size_t)256 * N) * sizeof(float);
    BY_T(dtype, hipLaunchKernelGGL(attn_bwd<bf16>, dim3(B), dim3(256), lds, ST, (const bf16*)a, (const bf16*)ws, v, p, dvhat, K, H, D, dv, (bf16*)dpre, dws_part, act),
         hipLaunchKernelGGL(attn_bwd<float>, dim3(B), dim3(256), lds, ST, (const float*)a, (const float*)ws, v, p, dvhat, K, H, D, dv, (float*)dpre, dws_part, act));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_group_sum(void* stream, int dtype, const void* x, int B, int K, int H, float* out, void* out_t) {
    const int N = dtype == VQA_BF16 ? 8 : 4;
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(out_t)) & 15) == 0;
    if (B >= 1 && K >= 1 && H % N == 0 && al && (long)B * (H / N) < (1L << 31)) {
        const dim3 vgrid((unsigned)(((long)B * (H / N) + 255) / 256));
        BY_T(dtype, hipLaunchKernelGGL(group_sum_vec<bf16>, vgrid, dim3(256), 0, ST, (const bf16*)x, B, K, H, out, (bf16*)out_t),
             hipLaunchKernelGGL(group_sum_vec<float>, vgrid, dim3(256), 0, ST, (const float*)x, B, K, H, out, (float*)out_t));
        CHECK_LAUNCH();
        return VQA_OK;
    }
    dim3 grid(grid1d((long)B * H));
    BY_T(dtype, hipLaunchKernelGGL(group_sum<bf16>, grid, dim3(256), 0, ST, (const bf16*)x, B, K, H, out, (bf16*)out_t),
         hipLaunchKernelGGL(group_sum<float>, grid, dim3(256), 0, ST, (const float*)x, B, K, H, out, (float*)out_t));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_hadamard_fwd(void* stream, int dtype, const float* fv, const float* fq, const float* mask, long n,
                                void* out, const float* add) {
    dim3 grid(grid1d(n));
    BY_T(dtype, hipLaunchKernelGGL(hadamard_fwd<bf16>, grid, dim3(256), 0, ST, fv, fq, mask, add, n, (bf16*)out),
         hipLaunchKernelGGL(hadamard_fwd<float>, grid, dim3(256), 0, ST, fv, fq, mask, add, n, (float*)out));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_hadamard_bwd(void* stream, int dtype, const float* dm0, const float* fv, const float* fq,
                                const float* mask, long n, void* dpv, void* dpq, int act) {
    if (act != VQA_ACT_RELU && act != VQA_ACT_SELU) return VQA_ERR_ARG;
    dim3 grid(grid1d(n));
    BY_T(dtype, hipLaunchKernelGGL(hadamard_bwd<bf16>, grid, dim3(256), 0, ST, dm0, fv, fq, mask, n, (bf16*)dpv, (bf16*)dpq, act),
         hipLaunchKernelGGL(hadamard_bwd<float>, grid, dim3(256), 0, ST, dm0, fv, fq, mask, n, (float*)dpv, (float*)dpq, act));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_relu_bwd(void* stream, int dtype, const float* g, const float* y, const float* mask, long n,
                            void* out, int act) {
    if (act != VQA_ACT_RELU && act != VQA_ACT_SELU) return VQA_ERR_ARG;
    dim3 grid(grid1d(n));
    BY_T(dtype, hipLaunchKernelGGL(relu_bwd<bf16>, grid, dim3(256), 0, ST, g, y, mask, n, (bf16*)out, act),
         hipLaunchKernelGGL(relu_bwd<float>, grid, dim3(256), 0, ST, g, y, mask, n, (float*)out, act));
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_mask_mul(void* stream, int dtype, const float* x, const float* mask, long n, void* out,
                            const float* add) {
    dim3 grid(grid1d(n));
    BY_T(dtype, hipLaunchKernelGGL(mask_mul<bf16>, grid, dim3(256), 0, ST, x, mask, add, n, (bf16*)out),
         hipLaunchKernelGGL(mask_mul<float>, grid, dim3(256), 0, ST, x, mask, add, n, (float*)out));
    CHECK_LAUNCH();
    return VQA_OK;
}

// loss[0] = BCE-with-logits sum; dlogits (may be null) = sigmoid(x) - t.   part: 256 doubles of scratch
extern "C" int vqa_bce_sum(void* stream, const float* logits, const float* target, int B, int C, int ldx,
                           float* dlogits, double* part, float* loss) {
    if (B < 1 || C < 1 || (long)B * C >= (1L << 31)) return VQA_ERR_ARG;
    hipLaunchKernelGGL(bce_partial, dim3(256), dim3(256), 0, ST, logits, target, B, C, ldx, dlogits, part);
    hipLaunchKernelGGL(dot_final, dim3(1), dim3(64), 0, ST, part, 256, loss);
    CHECK_LAUNCH();
    return VQA_OK;
}

// acc[0] = sum_b target[b][argmax logits[b]];  rows: B floats of scratch
extern "C" int vqa_soft_accuracy_sum(void* stream, const float* logits, const float* target, int B, int C, int ldx,
                                     float* rows, float* acc) {
    if (B < 1 || C < 1 || ldx < C || !logits || !target || !rows || !acc) return VQA_ERR_ARG;
    hipLaunchKernelGGL(soft_acc_rows, dim3(B), dim3(64), 0, ST, logits, target, C, ldx, rows);
    hipLaunchKernelGGL(sum_f32, dim3(1), dim3(64), 0, ST, rows, B, acc);
    CHECK_LAUNCH();
    return VQA_OK;
}

// ---- the ONE entry point that allocates nothing and takes no workspace (no counterpart in the reference).
// vqa_soft_accuracy_sum needs B floats of scratch (one gathered score per row, summed in fixed order by the second
// kernel); callers that cannot hand over device memory they allocated for it - code under HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc - use this variant, whose scratch is a file-scope device array.
// Each GPU holds its own copy of the array at its own address until the process ends; `scratch` below is only the
// host-side stand-in for it, never a device address: the address comes from hipGetSymbolAddress, once per device.
constexpr int NOALLOC_MAX_ROWS = 65536;                       // 256 KiB per GPU
static __device__ __align__(256) float scratch[NOALLOC_MAX_ROWS];
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::atomic<float*> scratch_ptr[NOALLOC_MAX_DEVICES];   // per device, filled on the first call there

// The current device's copy of `scratch` (nullptr on failure).  Set-up happens on the first call on a device, with
// that device current (the caller's: the one its stream and buffers live on); two threads racing here resolve the same
// address.
static float* scratch_of_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) return nullptr;
    float* p = scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym) return nullptr;
        p = static_cast<float*>(sym);
        scratch_ptr[dev].store(p, std::memory_order_release);
    }
    return p;
}

// acc[0] = sum_b target[b][argmax logits[b]] exactly as vqa_soft_accuracy_sum, without the `rows` workspace.
// B <= 65536 (VQA_ERR_CAPACITY beyond).  Every call on a GPU shares that GPU's scratch array: calls that overlap on one
// GPU (two streams, two threads) race on it; calls ordered on one stream do not.  Nothing needs clearing between
// calls: rows [0, B) are overwritten before the sum reads exactly those.
extern "C" int vqa_soft_accuracy_sum_noalloc(void* stream, const float* logits, const float* target, int B, int C,
                                             int ldx, float* acc) {
    if (B < 1 || C < 1 || ldx < C || !logits || !target || !acc) return VQA_ERR_ARG;
    if (B > NOALLOC_MAX_ROWS) return VQA_ERR_CAPACITY;
    float* rows = scratch_of_current_device();
    if (!rows) return VQA_ERR_LAUNCH;
    hipLaunchKernelGGL(soft_acc_rows, dim3(B), dim3(64), 0, ST, logits, target, C, ldx, rows);
    hipLaunchKernelGGL(sum_f32, dim3(1), dim3(64), 0, ST, rows, B, acc);
    CHECK_LAUNCH();
    return VQA_OK;
}

// The same loss / accuracy from sparse targets: labels int64 [B][L] (< 0 = empty), scores f32 [B][L], L <= 256.
// rowloss: B doubles of scratch; rows: B floats of scratch.
extern "C" int vqa_bce_sum_sparse(void* stream, const float* logits, const long* labels, const float* scores, int B,
                                  int C, int L, int ldx, float* dlogits, double* rowloss, float* loss) {
    if (B < 1 || C < 1 || L < 1 || L > 256) return VQA_ERR_ARG;
    hipLaunchKernelGGL(bce_sparse_rows, dim3(B), dim3(256), 0, ST, logits, labels, scores, C, L, ldx, dlogits, rowloss);
    hipLaunchKernelGGL(dot_final, dim3(1), dim3(64), 0, ST, rowloss, B, loss);
    CHECK_LAUNCH();
    return VQA_OK;
}
extern "C" int vqa_soft_accuracy_sum_sparse(void* stream, const float* logits, const long* labels, const float* scores,
                                            int B, int C, int L, int ldx, float* rows, float* acc) {
    if (B < 1 || C < 1 || L < 1) return VQA_ERR_ARG;
    hipLaunchKernelGGL(soft_acc_rows_sparse, dim3(B), dim3(64), 0, ST, logits, labels, scores, C, L, ldx, rows);
    hipLaunchKernelGGL(sum_f32, dim3(1), dim3(64), 0, ST, rows, B, acc);
    CHECK_LAUNCH();
    return VQA_OK;
}

// Global-norm clip + Adamax over ONE flat fp32 parameter / gradient buffer, no host synchronisation.
// state: m, u (n floats each); scratch: part (1024 doubles), sq/coef/total_norm (1 float each).
extern "C" int vqa_clip_adamax(void* stream, float* p, const float* g, float* m, float* u, long n, float max_norm,
                               float lr, float b1, float b2, float eps, int step, double* part, float* sq,
                               float* coef, float* total_norm) {
    if (n <= 0 || step < 1) return VQA_ERR_ARG;
    const double* pp = nullptr;
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(dot_partial, dim3(1024), dim3(256), 0, ST, g, g, n, part);
        pp = part;
    }
    const float lr_t = lr / (1.f - powf(b1, (float)step));
    hipLaunchKernelGGL(adamax_step, dim3(grid1d(n)), dim3(256), 0, ST, p, g, m, u, n, pp, max_norm, sq, coef,
                       total_norm, lr_t, b1, b2, eps);
    CHECK_LAUNCH();
    return VQA_OK;
}
