// One nn.GRU time step (reference gru.py:104) as ONE launch: the recurrent GEMM with the gate arithmetic in its
// epilogue, forward and BPTT, bf16 operands / f32 accumulation.
//
// The recurrence is T dependent steps of a small GEMM (B x 3H x H, 1.6 GFLOP at B = 256, H = 1024) and an elementwise
// gate pass.  As (GEMM + split-K reduce + gate kernel) a step was three dependent launches of 4-13 us each, none of them
// near any roofline: launch latency, pipeline fill and the round trip of gh / dh through memory.  Here a step is one
// kernel whose main loop has no LDS staging and no barrier:
//   * BOTH operands are kept in MFMA-fragment order in memory (vqa_gru_pack_whh for W_hh / W_hh^T once per training
//     step; the step kernels' own epilogues for h_t and dgh_t), so a fragment is one fully coalesced 1 KB wave load
//     straight into registers.  (Row-major operands - every lane its own row - cost twice the time: 16 cache lines per
//     load instruction; measured 18.9 vs 9.5 us per forward step.)
//   * a step is a latency problem (every launch finds W_hh in the Infinity Cache, not in L2), so a workgroup = 16 hidden
//     units x 64 batch rows is cut along K into 8 waves: at H = 1024 every load of a forward step is in flight at once,
//     one memory round trip per step, and every weight fragment is fetched once per workgroup for all four 16-row
//     blocks.  The K slices are combined through LDS in fixed order by the four waves that run the epilogue;
//   * workgroups with the same units sit on one XCD (one L2);
//   * D rows = hidden units, D columns = batch rows, so a lane ends with 4 consecutive units of one batch row for all
//     three gates: the epilogue is the gate kernel's float4 arithmetic on accumulators, gh / dh never exist in memory.
// forward  step t:  gh = h_{t-1} W_hh^T + b_hh;  r, z, n, h_t   (saves r, z, n, W_hn h + b_hn for the backward)
// backward step t:  dh_t = carry_{t+1} + dgh_{t+1} W_hh;  gate backward of step t -> dgi_t, dgh_t, carry_t = dh_t z_t
//
// Fragment order.  A 64-k chunk c of a 16-row block is two fragments h = 0, 1 of 64 lanes x 16 bytes; lane (r = lane &
// 15, q = lane >> 4) holds row r, k = 64 c + 16 q + 8 h + 0..7 (the same k permutation on both operands).
//   weights, forward   wf[ug][g][c][h][lane]         row = W_hh[g H + 16 ug + r][k]
//   weights, backward  wb[ug][c][h][lane]            row = W_hh[k][16 ug + r]          (k over 3H)
//   activations        xp[rg][c][rb][h][lane]        row = X[64 rg + 16 rb + r][k]     (X = h_{t-1} or dgh_{t+1})
#include "common.h"
#include "../../include/vqa_hip.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

constexpr int KS = 8;      // K slices = waves per workgroup

struct StepFwd {
    const bf16* xp;       // packed h_{t-1}, or null (t = 0: h_{-1} = 0, no product)
    const float* hp_f;    // [B][H]  h_{t-1} (f32, for the update); t = 0: not read - h_{-1} = 0 is WRITTEN there
    bf16* hp_t;           // t = 0 only: row-major h_{-1} slot of the T copy, zeroed here (the weight gradient reads it)
    const bf16* wf;       // packed W_hh
    const float* gi;      // [B][3H] input projection of step t (+ b_ih)
    const float* bhh;     // [3H]
    float* hn_f;          // [B][H]  h_t
    bf16* hn_t;           // [B][H]  (row-major copy: weight gradient, stacked layers)
    bf16* hn_p;           // packed h_t for the next step
    float* save;          // [B][4H]
    int B, H;
};

struct StepBwd {
    const bf16* xp;       // packed dgh_{t+1}, or null (the last time step: dh = carry)
    const bf16* wb;       // packed W_hh^T
    const float* carry;   // [B][H]  dh_{t+1} z_{t+1} (+ the layer above), or dh_last
    const float* save;    // [B][4H] of step t
    const float* hp_f;    // [B][H]  h_{t-1}
    const float* dh_add;  // [B][H] or null: added to the outgoing carry (multi-layer GRU)
    bf16* dgi;            // [B][3H] of step t (row-major: weight gradients)
    bf16* dgh;
    bf16* dgh_p;          // packed dgh_t for the next launch
    float* carry_out;     // [B][H]
    int B, H;
};

// block -> (unit group of 16, row group of 64): same units on one XCD (workgroups x, x + 8, ... share an L2)
__device__ __forceinline__ void decode_block(int bid, int nug, int nrg, int& ug, int& rg) {
    if ((nug & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3;
        ug = xcd + 8 * (idx / nrg);
        rg = idx % nrg;
    } else {
        ug = bid / nrg;
        rg = bid % nrg;
    }
}

// acc[rb][g] += W fragments (G row sets, `gstride` uint4 apart) x X fragments (4 row blocks) over the chunks
// [c_lo, c_hi) of this wave's K slice; NPF chunks of loads in flight.  wfrag / xfrag point at this lane's 16 bytes of
// chunk 0, fragment h = 0 (chunk stride: 128 uint4 for the weights, 512 for the activations).
// Everything that steers a branch here is wave-uniform BY CONSTRUCTION (readfirstlane'd wave index): MFMAs ignore EXEC,
// so a per-lane condition around them would only mask, not skip them.
template <int G, int NPF>
__device__ __forceinline__ void stream_mma(f32x4 (&acc)[4][G], const uint4* wfrag, long gstride, const uint4* xfrag,
                                           int c_lo, int c_hi) {
    uint4 wa[NPF][G][2], xb[NPF][4][2];
    auto load = [&](int slot, int c) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) wa[slot][g][h] = wfrag[g * gstride + (long)c * 128 + h * 64];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) xb[slot][rb][h] = xfrag[(long)c * 512 + rb * 128 + h * 64];
    };
    if (c_lo >= c_hi) return;
#pragma unroll
    for (int s = 0; s < NPF; ++s) load(s, c_lo + s < c_hi ? c_lo + s : c_hi - 1);
    for (int c0 = c_lo; c0 < c_hi; c0 += NPF) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            if (c0 + s < c_hi) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int rb = 0; rb < 4; ++rb) Mma<bf16>::mma(acc[rb][g], wa[s][g][h], xb[s][rb][h]);
            }
            if (c0 + NPF + s < c_hi) load(s, c0 + NPF + s);
        }
    }
}

// every K slice parks its accumulators in LDS; wave rb < 4 then owns row block rb and sums the slices in fixed order
template <int G>
__device__ __forceinline__ void combine_slices(f32x4 (&acc)[4][G], f32x4 (&out)[G], int wave, int lane) {
    __shared__ f32x4 red[KS][4][G][64];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int g = 0; g < G; ++g) red[wave][rb][g][lane] = acc[rb][g];
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 s = red[0][wave][g][lane];
#pragma unroll
            for (int k = 1; k < KS; ++k) s += red[k][wave][g][lane];
            out[g] = s;
        }
    }
}

// where this lane's 4 units (16 ug + 4 q .. + 3) of row block rb land in the packed activation of the next launch;
// `kbase` = first k of unit group ug in that operand (forward: 16 ug; backward: g H + 16 ug).  Byte offset.
__device__ __forceinline__ long packed_slot(int rg, int nc, int kbase, int rb, int r, int q) {
    const int c = kbase >> 6, qq = (kbase >> 4) & 3;          // chunk, lane group of the consumer
    const long frag = (((long)rg * nc + c) * 4 + rb) * 2 + (q >> 1);
    return (frag * 64 + qq * 16 + r) * 16 + (q & 1) * 8;
}

__global__ __launch_bounds__(64 * KS) void gru_step_fwd_k(StepFwd a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    const int H = a.H, nc = H >> 6;
    int ug, rg;
    decode_block(blockIdx.x, H / 16, (a.B + 63) / 64, ug, rg);
    f32x4 acc[4][3];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[rb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.xp)
        stream_mma<3, 2>(acc, reinterpret_cast<const uint4*>(a.wf) + (long)ug * 3 * nc * 128 + lane, (long)nc * 128,
                         reinterpret_cast<const uint4*>(a.xp) + (long)rg * nc * 512 + lane, wave * nc / KS,
                         (wave + 1) * nc / KS);
    // epilogue operands of the four row-block waves: issued before the combine, they arrive under the barrier
    const int m = rg * 64 + wave * 16 + r, uj = ug * 16 + q * 4;   // (wave < 4: this lane's batch row and 4 units)
    const bool ok = wave < 4 && m < a.B;
    float ir[4], iz[4], in_[4], br[4], bz[4], bn[4], hp[4];
    if (wave < 4) {
        const long mr = ok ? m : 0;
        const float* gib = a.gi + mr * 3 * H + uj;
        load4(gib, ir); load4(gib + H, iz); load4(gib + 2 * H, in_);
        load4(a.bhh + uj, br); load4(a.bhh + H + uj, bz); load4(a.bhh + 2 * H + uj, bn);
        if (a.xp) {
            load4(a.hp_f + mr * H + uj, hp);
        } else {
            // the first time step: h_{-1} = 0.  Nobody has to zero slot 0 of h_f / h_t beforehand - this launch writes
            // it (the backward's step 0 and the weight-gradient GEMM read it)
#pragma unroll
            for (int e = 0; e < 4; ++e) hp[e] = 0.f;
            if (ok) {
                store4<float>(const_cast<float*>(a.hp_f) + (long)m * H + uj, hp);
                store4<bf16>(a.hp_t + (long)m * H + uj, hp);
            }
        }
    }
    f32x4 gh[3];
    combine_slices<3>(acc, gh, wave, lane);
    if (!ok) return;
    float rr[4], zz[4], nn[4], hn[4], h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        rr[e] = sigm(ir[e] + gh[0][e] + br[e]);
        zz[e] = sigm(iz[e] + gh[1][e] + bz[e]);
        hn[e] = gh[2][e] + bn[e];
        nn[e] = tanhf(in_[e] + rr[e] * hn[e]);
        h[e] = (1.f - zz[e]) * nn[e] + zz[e] * hp[e];
    }
    store4<float>(a.hn_f + (long)m * H + uj, h);
    store4<bf16>(a.hn_t + (long)m * H + uj, h);
    store4<bf16>(reinterpret_cast<bf16*>(reinterpret_cast<char*>(a.hn_p) + packed_slot(rg, nc, ug * 16, wave, r, q)), h);
    float* sv = a.save + (long)m * 4 * H + uj;
    store4<float>(sv, rr); store4<float>(sv + H, zz); store4<float>(sv + 2 * H, nn); store4<float>(sv + 3 * H, hn);
}

__global__ __launch_bounds__(64 * KS) void gru_step_bwd_k(StepBwd a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, q = lane >> 4;
    const int H = a.H, nc = 3 * H >> 6;
    int ug, rg;
    decode_block(blockIdx.x, H / 16, (a.B + 63) / 64, ug, rg);
    f32x4 acc[4][1];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.xp)
        stream_mma<1, 3>(acc, reinterpret_cast<const uint4*>(a.wb) + (long)ug * nc * 128 + lane, 0,
                         reinterpret_cast<const uint4*>(a.xp) + (long)rg * nc * 512 + lane, wave * nc / KS,
                         (wave + 1) * nc / KS);
    const int m = rg * 64 + wave * 16 + r, uj = ug * 16 + q * 4;
    const bool ok = wave < 4 && m < a.B;
    float rr[4], zz[4], nn[4], hn[4], g[4], hp[4], ad[4] = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) {
        const long mr = ok ? m : 0;
        const float* sv = a.save + mr * 4 * H + uj;
        load4(sv, rr); load4(sv + H, zz); load4(sv + 2 * H, nn); load4(sv + 3 * H, hn);
        load4(a.carry + mr * H + uj, g);
        load4(a.hp_f + mr * H + uj, hp);
        if (a.dh_add) load4(a.dh_add + mr * H + uj, ad);
    }
    f32x4 dsum[1];
    combine_slices<1>(acc, dsum, wave, lane);
    if (!ok) return;
    float dr[4], dz[4], dn[4], dnr[4], dp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gg = g[e] + dsum[0][e];
        dn[e] = gg * (1.f - zz[e]) * (1.f - nn[e] * nn[e]);
        dz[e] = gg * (hp[e] - nn[e]) * zz[e] * (1.f - zz[e]);
        dr[e] = dn[e] * hn[e] * rr[e] * (1.f - rr[e]);
        dnr[e] = dn[e] * rr[e];
        dp[e] = gg * zz[e] + ad[e];
    }
    bf16* gi = a.dgi + (long)m * 3 * H + uj;
    bf16* gh = a.dgh + (long)m * 3 * H + uj;
    store4<bf16>(gi, dr); store4<bf16>(gi + H, dz); store4<bf16>(gi + 2 * H, dn);
    store4<bf16>(gh, dr); store4<bf16>(gh + H, dz); store4<bf16>(gh + 2 * H, dnr);
    char* pk = reinterpret_cast<char*>(a.dgh_p);
    store4<bf16>(reinterpret_cast<bf16*>(pk + packed_slot(rg, nc, ug * 16, wave, r, q)), dr);
    store4<bf16>(reinterpret_cast<bf16*>(pk + packed_slot(rg, nc, H + ug * 16, wave, r, q)), dz);
    store4<bf16>(reinterpret_cast<bf16*>(pk + packed_slot(rg, nc, 2 * H + ug * 16, wave, r, q)), dnr);
    store4<float>(a.carry_out + (long)m * H + uj, dp);
}

// f32 master W_hh [3H][H] -> the two fragment-order bf16 copies (either may be null); one thread per 16-byte fragment
// piece.  The backward copy reads W_hh transposed: the 16 lanes r of a piece are 16 consecutive floats of one row.
__global__ void gru_pack_whh_k(const float* w, bf16* wf, bf16* wb, int H) {
    const long n = (long)3 * H * H / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), r = lane & 15, q = lane >> 4, h = (int)(i >> 6) & 1;
    const long f = i >> 7;                                       // (row set, chunk) index
    bf16x8 o;
    if (wf) {
        const int nc = H >> 6, c = (int)(f % nc);
        const long rs = f / nc;                                  // ug * 3 + g
        const int g = (int)(rs % 3), ug = (int)(rs / 3);
        const float* p = w + ((long)g * H + ug * 16 + r) * H + c * 64 + q * 16 + h * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
        o = bf16x8{(bf16)v0.x, (bf16)v0.y, (bf16)v0.z, (bf16)v0.w, (bf16)v1.x, (bf16)v1.y, (bf16)v1.z, (bf16)v1.w};
        *reinterpret_cast<bf16x8*>(wf + i * 8) = o;
    }
    if (wb) {
        const int nc = 3 * H >> 6, c = (int)(f % nc), ug = (int)(f / nc);
        const float* p = w + ((long)c * 64 + q * 16 + h * 8) * H + ug * 16 + r;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)p[(long)e * H];
        *reinterpret_cast<bf16x8*>(wb + i * 8) = o;
    }
}

bool shape_ok(int B, int T, int H) { return B >= 1 && T >= 1 && H >= 64 && H % 64 == 0; }

// elements of one packed activation: rows padded to whole 64-row groups (the pad rows are never written and only feed
// D columns that are never stored)
long packed_elems(int B, int k) { return (long)((B + 63) / 64) * 64 * k; }

}  // namespace

extern "C" int vqa_gru_pack_whh(void* stream, const float* w_hh, int H, void* wf, void* wb) {
    if (!w_hh || H < 64 || H % 64 || (!wf && !wb)) return VQA_ERR_ARG;
    const long n = (long)3 * H * H / 8;
    hipLaunchKernelGGL(gru_pack_whh_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_hh, static_cast<bf16*>(wf), static_cast<bf16*>(wb), H);
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_fwd_fused(void* stream, const float* gi, const void* wf, const float* b_hh, float* h_f, void* h_t,
                                 void* h_p, float* save, int B, int T, int H) {
    if (!gi || !wf || !b_hh || !h_f || !h_t || !h_p || !save) return VQA_ERR_ARG;
    if (!shape_ok(B, T, H)) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long bh = (long)B * H, ph = packed_elems(B, H);
    const dim3 grid((unsigned)((H / 16) * ((B + 63) / 64)));
    for (int t = 0; t < T; ++t) {
        StepFwd a{t == 0 ? nullptr : static_cast<const bf16*>(h_p) + t * ph, h_f + t * bh,
                  t == 0 ? static_cast<bf16*>(h_t) : nullptr, static_cast<const bf16*>(wf),
                  gi + (long)t * B * 3 * H, b_hh, h_f + (t + 1) * bh, static_cast<bf16*>(h_t) + (t + 1) * bh,
                  static_cast<bf16*>(h_p) + (t + 1) * ph, save + (long)t * B * 4 * H, B, H};
        hipLaunchKernelGGL(gru_step_fwd_k, grid, dim3(64 * KS), 0, s, a);
    }
    CHECK_LAUNCH();
    return VQA_OK;
}

extern "C" int vqa_gru_bwd_fused(void* stream, const float* dh_last, const float* save, const float* h_f, const void* wb,
                                 void* dgi, void* dgh, void* dgh_p, float* dh_ws, int B, int T, int H,
                                 const float* dh_seq) {
    if (!dh_last || !save || !h_f || !wb || !dgi || !dgh || !dgh_p || !dh_ws) return VQA_ERR_ARG;
    if (!shape_ok(B, T, H)) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long bh = (long)B * H, b3 = (long)B * 3 * H, pg = packed_elems(B, 3 * H);
    const dim3 grid((unsigned)((H / 16) * ((B + 63) / 64)));
    const float* carry = dh_last;
    for (int t = T - 1; t >= 0; --t) {
        float* out = dh_ws + (t & 1) * bh;                 // (never the buffer `carry` points to: they alternate)
        bf16* pk = static_cast<bf16*>(dgh_p);
        StepBwd a{t == T - 1 ? nullptr : pk + ((t + 1) & 1) * pg, static_cast<const bf16*>(wb), carry,
                  save + (long)t * B * 4 * H, h_f + t * bh, (dh_seq && t > 0) ? dh_seq + (t - 1) * bh : nullptr,
                  static_cast<bf16*>(dgi) + t * b3, static_cast<bf16*>(dgh) + t * b3, pk + (t & 1) * pg, out, B, H};
        hipLaunchKernelGGL(gru_step_bwd_k, grid, dim3(64 * KS), 0, s, a);
        carry = out;
    }
    CHECK_LAUNCH();
    return VQA_OK;
}
