// Persistent GRU recurrence (reference gru.py:104, nn.GRU one layer, h0 = 0): ONE launch runs all T timesteps.
//
// Workgroup (jt, bt) owns 16 hidden units [16*jt, 16*jt+16) and a slab of batch rows.  Its slice of W_hh - the 48
// gate rows (r, z, n) of those units, 48 x H bf16 = 96 KB at H = 1024 - is loaded into LDS once and stays there for
// the whole sequence; per step each wave multiplies 16 batch rows of h_{t-1} (read straight from L2, bf16) against the
// three 16-unit gate tiles on the MFMA, applies the gate non-linearities in registers and writes h_t.  Steps are
// separated by one device-wide barrier (monotonic counter, agent-scope release / acquire, bounded spin).
// The backward kernel mirrors it: W_hh^T slice resident in LDS, gate derivatives in registers, one barrier per step.
//
// All workgroups must be co-resident: grid = (H/16) * NB <= resident workgroups per CU (occupancy query) x CUs, checked
// on the host.  Every spin is bounded: on expiry the kernel sets flags[1] and runs on, and every workgroup that sees
// the marker in its last step writes NaN instead of its results - a timed-out barrier poisons h_T / the gate
// gradients, reaches the loss and the gradient norm, and is never a silently wrong number (nor a hang).
#include "common.h"

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// flags[0] = arrival counter (zeroed by the host before the launch), flags[1] = timeout marker
__device__ __forceinline__ void grid_barrier(unsigned* flags, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) {
                __hip_atomic_store(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

struct GruArgs {
    const float* gi;      // [T][B][3H]  x W_ih^T + b_ih
    const bf16* w;        // fwd: W_hh [3H][H];  bwd: W_hh^T [H][ldw] (k = 3H contiguous)
    int ldw;
    const float* b_hh;    // [3H]
    float* h_f;           // [T+1][B][H] fp32, slot 0 = h_{-1} = 0
    bf16* h_t;            // [T+1][B][H] bf16 copy (MFMA operand)
    float* save;          // [T][B][4H]  r, z, n, hn
    const float* dh_last; // bwd: [B][H]
    bf16* dgi;            // bwd out: [T][B][3H]
    bf16* dgh;            // bwd out: [T][B][3H]
    int B, T, H, rows_per_wg;
    unsigned* flags;
};

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void gru_fwd_persistent(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = a.H, B = a.B;
    const int WROWB = H * 2 + 32;                       // 16-byte slots per row = 2 (mod 4): conflict-free b128 reads
    const int nj = H / 16;
    const int jt = blockIdx.x % nj, bt = blockIdx.x / nj;
    const int j0 = jt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const unsigned nwg = gridDim.x;

    // resident weight slice: LDS row g*16 + u = W_hh[g*H + j0 + u][:]
    const int vpr = H / 8;
    for (int i = tid; i < 48 * vpr; i += 256) {
        const int row = i / vpr, v = i % vpr;
        const int g = row / 16, u = row % 16;
        *reinterpret_cast<uint4*>(smem + row * WROWB + v * 16) =
            *reinterpret_cast<const uint4*>(a.w + (long)(g * H + j0 + u) * a.ldw + v * 8);
    }
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) bh[g][e] = a.b_hh[g * H + j0 + q * 4 + e];
    __syncthreads();

    const int row_base = bt * a.rows_per_wg;
    for (int t = 0; t < a.T; ++t) {
        // last step: a barrier that timed out anywhere (flags[1]) turns this workgroup's h_T into NaN
        const float poison = (t + 1 == a.T && __hip_atomic_load(a.flags + 1, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT) != 0u) ? NAN : 0.f;
        const bf16* hp_t = a.h_t + (long)t * B * H;
        const float* hp_f = a.h_f + (long)t * B * H;
        float* hn_f = a.h_f + (long)(t + 1) * B * H;
        bf16* hn_t = a.h_t + (long)(t + 1) * B * H;
        for (int rc = wave * 16; rc < a.rows_per_wg; rc += 64) {
            const int row0 = row_base + rc;
            if (row0 >= B) break;                        // wave-uniform
            const int brow = min(row0 + r, B - 1);       // clamp: padded rows compute garbage that is never stored
            f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const bf16* arow = hp_t + (long)brow * H + q * 8;
            // one latency, not H/256 of them: issue the epilogue's operands and the first batch of h fragments
            // together, then keep the next batch in flight under the MFMAs of the current one
            const int b = row0 + r;
            const int bc = min(b, B - 1);
            const int j = j0 + q * 4;
            const float* gib = a.gi + ((long)t * B + bc) * 3 * H + j;
            const float4 gr = *reinterpret_cast<const float4*>(gib);
            const float4 gz = *reinterpret_cast<const float4*>(gib + H);
            const float4 gn = *reinterpret_cast<const float4*>(gib + 2 * H);
            const float4 hp = *reinterpret_cast<const float4*>(hp_f + (long)bc * H + j);
            uint4 av[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[0][u] = *reinterpret_cast<const uint4*>(arow + u * 32);
            const int nk = H / 256;
#pragma unroll 1
            for (int kb = 0; kb < nk; kb += 2) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int k0 = (kb + half) * 256;
                    if (k0 + 256 < H) {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            av[half ^ 1][u] = *reinterpret_cast<const uint4*>(arow + k0 + 256 + u * 32);
                    }
                    if (k0 < H) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
#pragma unroll
                            for (int g = 0; g < 3; ++g) {
                                const uint4 wv = *reinterpret_cast<const uint4*>(smem + (g * 16 + r) * WROWB +
                                                                                 (k0 + u * 32 + q * 8) * 2);
                                Mma<bf16>::mma(acc[g], wv, av[half][u]);   // D rows = unit, D cols = batch row
                            }
                        }
                    }
                }
            }
            if (b < B) {
                const float gir[4] = {gr.x, gr.y, gr.z, gr.w}, giz[4] = {gz.x, gz.y, gz.z, gz.w},
                            gin[4] = {gn.x, gn.y, gn.z, gn.w}, hpv[4] = {hp.x, hp.y, hp.z, hp.w};
                float rr[4], zz[4], nn[4], hh[4], hn[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rr[e] = sigm(gir[e] + acc[0][e] + bh[0][e]);
                    zz[e] = sigm(giz[e] + acc[1][e] + bh[1][e]);
                    hn[e] = acc[2][e] + bh[2][e];
                    nn[e] = tanhf(gin[e] + rr[e] * hn[e]);
                    hh[e] = (1.f - zz[e]) * nn[e] + zz[e] * hpv[e] + poison;
                }
                store_vec4<float>(hn_f + (long)b * H + j, hh);
                store_vec4<bf16>(hn_t + (long)b * H + j, hh);
                float* sv = a.save + ((long)t * B + b) * 4 * H + j;
                store_vec4<float>(sv, rr);
                store_vec4<float>(sv + H, zz);
                store_vec4<float>(sv + 2 * H, nn);
                store_vec4<float>(sv + 3 * H, hn);
            }
        }
        if (t + 1 < a.T) grid_barrier(a.flags, (unsigned)(t + 1) * nwg);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// dh_{t-1}[b][j] = dh_t z + sum_k dgh_t[b][k] W_hh[k][j];   dgi_t = [dr, dz, dn], dgh_t = [dr, dz, dn r]
__global__ __launch_bounds__(256) void gru_bwd_persistent(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = a.H, B = a.B, K3 = 3 * H;
    const int WROWB = K3 * 2 + 32;
    const int nj = H / 16;
    const int jt = blockIdx.x % nj, bt = blockIdx.x / nj;
    const int j0 = jt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const unsigned nwg = gridDim.x;

    // resident slice: LDS row u = W_hh^T[j0 + u][0 .. 3H)
    const int vpr = K3 / 8;
    for (int i = tid; i < 16 * vpr; i += 256) {
        const int row = i / vpr, v = i % vpr;
        *reinterpret_cast<uint4*>(smem + row * WROWB + v * 16) =
            *reinterpret_cast<const uint4*>(a.w + (long)(j0 + row) * a.ldw + v * 8);
    }
    __syncthreads();

    const int row_base = bt * a.rows_per_wg;
    // dh of the current step for this lane's (row chunk, 4 units); at most 4 row chunks per wave are kept in registers
    constexpr int MAXC = 4;
    float dh[MAXC][4];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int b = row_base + wave * 16 + c * 64 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            dh[c][e] = (wave * 16 + c * 64 < a.rows_per_wg && b < B) ? a.dh_last[(long)b * H + j0 + q * 4 + e] : 0.f;
    }
    unsigned epoch = 0;
    for (int t = a.T - 1; t >= 0; --t) {
        float part[MAXC][4];
        const float poison = (t == 0 && __hip_atomic_load(a.flags + 1, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT) != 0u) ? NAN : 0.f;
        // ---- phase A: gate derivatives of step t for the owned (rows, units)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int rc = wave * 16 + c * 64;
            const int b = row_base + rc + r;
            if (rc < a.rows_per_wg && b < B) {
                const int j = j0 + q * 4;
                const float* sv = a.save + ((long)t * B + b) * 4 * H + j;
                const float4 r4 = *reinterpret_cast<const float4*>(sv), z4 = *reinterpret_cast<const float4*>(sv + H),
                             n4 = *reinterpret_cast<const float4*>(sv + 2 * H),
                             h4 = *reinterpret_cast<const float4*>(sv + 3 * H),
                             p4 = *reinterpret_cast<const float4*>(a.h_f + ((long)t * B + b) * H + j);
                const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w},
                            nn[4] = {n4.x, n4.y, n4.z, n4.w}, hn[4] = {h4.x, h4.y, h4.z, h4.w},
                            hp[4] = {p4.x, p4.y, p4.z, p4.w};
                float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = dh[c][e] + poison;
                    dn[e] = g * (1.f - zz[e]) * (1.f - nn[e] * nn[e]);
                    dz[e] = g * (hp[e] - nn[e]) * zz[e] * (1.f - zz[e]);
                    dr[e] = dn[e] * hn[e] * rr[e] * (1.f - rr[e]);
                    dnr[e] = dn[e] * rr[e];
                    part[c][e] = g * zz[e];
                }
                bf16* gi_o = a.dgi + ((long)t * B + b) * K3 + j;
                bf16* gh_o = a.dgh + ((long)t * B + b) * K3 + j;
                store_vec4<bf16>(gi_o, dr); store_vec4<bf16>(gi_o + H, dz); store_vec4<bf16>(gi_o + 2 * H, dn);
                store_vec4<bf16>(gh_o, dr); store_vec4<bf16>(gh_o + H, dz); store_vec4<bf16>(gh_o + 2 * H, dnr);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) part[c][e] = 0.f;
            }
        }
        if (t == 0) break;
        grid_barrier(a.flags, ++epoch * nwg);
        // ---- phase B: dh_{t-1} = part + dgh_t @ W_hh (the owned 16 columns)
        const bf16* dg = a.dgh + (long)t * B * K3;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int rc = wave * 16 + c * 64;
            if (rc >= a.rows_per_wg || row_base + rc >= B) continue;     // wave-uniform
            const int brow = min(row_base + rc + r, B - 1);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16* arow = dg + (long)brow * K3 + q * 8;
            for (int k0 = 0; k0 < K3; k0 += 256) {
                uint4 av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) av[u] = *reinterpret_cast<const uint4*>(arow + k0 + u * 32);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint4 wv = *reinterpret_cast<const uint4*>(smem + r * WROWB + (k0 + u * 32 + q * 8) * 2);
                    Mma<bf16>::mma(acc, wv, av[u]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[c][e] = part[c][e] + acc[e];
        }
    }
}

}  // namespace

// Largest co-resident grid of a persistent kernel with `lds` bytes of dynamic LDS: occupancy x CUs of the current
// device (queried per call: two cheap runtime calls; no state is cached, the library stays re-entrant).
template <typename K>
static int resident_workgroups(K kernel, size_t lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess) return 0;
    return cus * per_cu;
}

// batch slabs: as many as fill the resident grid (or max_wg, if the caller wants the recurrence to hold fewer CUs)
static int gru_slabs(int B, int nj, int resident, int max_wg) {
    int cap = resident;
    if (max_wg > 0 && max_wg < cap) cap = max_wg;
    if (nj > cap) return 0;
    int nb = (B + 63) / 64;
    if (nb > cap / nj) nb = cap / nj;
    return nb;
}

// gru persistent forward.  w = W_hh bf16 [3H][ldw].  flags: 2 unsigned (zeroed here).  Returns 1 if the shape does not
// fit the resident-weight scheme (caller falls back to per-step launches).
extern "C" int vqa_gru_fwd_persistent(void* stream, const float* gi, const void* w, int ldw, const float* b_hh,
                                      float* h_f, void* h_t, float* save, int B, int T, int H, unsigned* flags,
                                      int max_wg) {
    if (H % 256 || B < 1 || T < 1) return VQA_ERR_ARG;
    const size_t lds = (size_t)48 * (H * 2 + 32);
    if (lds > 160 * 1024) return VQA_ERR_ARG;
    const int nj = H / 16;
    (void)hipFuncSetAttribute((const void*)gru_fwd_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nb = gru_slabs(B, nj, resident_workgroups(gru_fwd_persistent, lds), max_wg);
    if (nb < 1) return VQA_ERR_ARG;
    const int rows_per_wg = ((B + nb - 1) / nb + 15) / 16 * 16;
    GruArgs a{gi, static_cast<const bf16*>(w), ldw, b_hh, h_f, static_cast<bf16*>(h_t), save, nullptr, nullptr, nullptr,
              B, T, H, rows_per_wg, flags};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), s) != hipSuccess) return VQA_ERR_LAUNCH;
    hipLaunchKernelGGL(gru_fwd_persistent, dim3(nj * nb), dim3(256), lds, s, a);
    CHECK_LAUNCH();
    return VQA_OK;
}

// gru persistent backward.  wt = W_hh^T bf16 [H][ldw] (3H contiguous).  dgi / dgh: bf16 [T][B][3H] outputs.
extern "C" int vqa_gru_bwd_persistent(void* stream, const void* wt, int ldw, const float* h_f, const float* save,
                                      const float* dh_last, void* dgi, void* dgh, int B, int T, int H,
                                      unsigned* flags, int max_wg) {
    if (H % 256 || B < 1 || T < 1) return VQA_ERR_ARG;
    const size_t lds = (size_t)16 * (3 * H * 2 + 32);
    if (lds > 160 * 1024) return VQA_ERR_ARG;
    const int nj = H / 16;
    (void)hipFuncSetAttribute((const void*)gru_bwd_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nb = gru_slabs(B, nj, resident_workgroups(gru_bwd_persistent, lds), max_wg);
    if (nb < 1) return VQA_ERR_ARG;
    const int rows_per_wg = ((B + nb - 1) / nb + 15) / 16 * 16;
    if (rows_per_wg > 256) return VQA_ERR_ARG;           // 4 row chunks per wave are kept in registers
    GruArgs a{nullptr, static_cast<const bf16*>(wt), ldw, nullptr, const_cast<float*>(h_f), nullptr,
              const_cast<float*>(save), dh_last, static_cast<bf16*>(dgi), static_cast<bf16*>(dgh), B, T, H, rows_per_wg,
              flags};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(flags, 0, 2 * sizeof(unsigned), s) != hipSuccess) return VQA_ERR_LAUNCH;
    hipLaunchKernelGGL(gru_bwd_persistent, dim3(nj * nb), dim3(256), lds, s, a);
    CHECK_LAUNCH();
    return VQA_OK;
}
