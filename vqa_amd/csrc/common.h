// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the VQA hot path.
// Wave = 64 lanes.  MFMA shapes used everywhere: v_mfma_f32_16x16x32_bf16 (bf16 in, f32 accumulate) and
// v_mfma_f32_16x16x4_f32 (exact f32).  Both share one C/D layout: col = lane & 15, row = 4*(lane>>4) + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define VQA_OK 0
#define VQA_ERR_ARG 1
#define VQA_ERR_LAUNCH 2
#define VQA_ERR_CAPACITY 3   // the no-allocation entry point: input larger than its static scratch

#define VQA_F32 0
#define VQA_BF16 1
#define VQA_F16 2

#define CHECK_LAUNCH()                                         \
    do {                                                       \
        hipError_t e_ = hipGetLastError();                     \
        if (e_ != hipSuccess) return VQA_ERR_LAUNCH;           \
    } while (0)

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// One "K-chunk" = 16 bytes per lane of each operand.  For bf16 that is 8 k-values per lane and ONE
// 16x16x32 MFMA (lane group g = lane>>4 holds k = 8g..8g+7).  For f32 it is 4 k-values per lane and FOUR
// 16x16x4 MFMAs (MFMA j takes element j of every lane: k = 4g + j).  Either way a chunk covers
// KCHUNK = 4 * VEC consecutive k of a row whose lane group g reads bytes [16g, 16g+16).
// `rows` feeds the MFMA A operand (D rows), `cols` the B operand (D columns).
template <typename T> struct Mma;
template <> struct Mma<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void mma(f32x4& c, const uint4& rows, const uint4& cols) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(rows.x), __uint_as_float(cols.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(rows.y), __uint_as_float(cols.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(rows.z), __uint_as_float(cols.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(rows.w), __uint_as_float(cols.w), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void mma(f32x4& c, const uint4& rows, const uint4& cols) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, rows),
                                                    __builtin_bit_cast(bf16x8, cols), c, 0, 0, 0);
    }
};

// XCD-aware walk of a persistent grid over `total` work units (image-major tiles).  Workgroups are dealt to the 8
// XCDs round-robin by linear id (x, x + 8, ... share one XCD and its 4 MiB L2; MI355X_MICROARCH.md, workgroup
// dispatch), so with the old walk u = x, x + G, ... the G tiles in flight at any moment - spatial neighbours that
// share their halo rows / columns - sat on 8 different L2s and every halo was fetched from the fabric once per
// XCD (PMC: 1.4x the algorithmic input bytes on the single-chunk conv).  Here XCD k owns the contiguous eighth
// [total*k/8, total*(k+1)/8) of the unit list and its G/8 workgroups walk it side by side: neighbours meet in one L2.
// Placement is a speed matter only: any dispatch order gives the same results (the unit -> workgroup map is fixed).
#ifndef VQA_XCD_WALK
#define VQA_XCD_WALK 1
#endif
// rev != 0: every workgroup takes ITS units in descending order (unit(t) = lo + hi - 1 - t over its range [lo, hi)):
// a consumer that walks a tensor larger than the 256 MB Infinity Cache in the producer's order finds the head evicted
// and evicts the tail before it gets there; walking it back to front it starts with what the producer wrote last
// (scripts/src/mall_micro.hip: 411 MB re-read at 3.7 TB/s front to back, 4.5 TB/s back to front).
// VQA_WALK_REV (bit mask in the environment, read once per process): which streaming kernels walk back to front -
// 1 plain single-chunk conv, 2 conv with the fused input BatchNorm, 4 data gradient with the BatchNorm-backward sums,
// 8 weight gradient, 16 first-layer forward, 32 first-layer weight gradient, 64 BatchNorm / pool forward,
// 128 BatchNorm backward apply pass, 256 BatchNorm backward reduce pass.
// Default 4, measured in the step (same box, 3 alternations of 60 steps each): the data gradient of a layer and its
// weight gradient are enqueued on two streams but each is a persistent grid that fills the chip, so they run one after
// the other over the same two tensors (dy and the lower layer's y, 411-418 MB each in block 1): walking them from
// opposite ends, the second starts with what the first read last.  4.240 -> 4.202 and 4.298 -> 4.265 ms/step; every
// other bit alone or with it is within noise (8 alone gains the same, 4 + 8 together nothing: same direction again).
#ifndef VQA_WALK_REV_DEFAULT
#define VQA_WALK_REV_DEFAULT 4
#endif
static inline int vqa_walk_rev(int bit) {
    static const int mask = [] { const char* e = getenv("VQA_WALK_REV"); return e ? atoi(e) : VQA_WALK_REV_DEFAULT; }();
    return (mask & bit) != 0;
}
struct UnitWalk {
    int first, step, end, flip;
    __device__ __forceinline__ int unit(int t) const { return flip < 0 ? t : flip - t; }
};
__device__ __forceinline__ UnitWalk unit_walk(int x, int G, int total, int rev = 0) {
    if (VQA_XCD_WALK && (G & 7) == 0 && total >= G) {
        const int k = x & 7;
        const int lo = (int)(((long)total * k) >> 3), hi = (int)(((long)total * (k + 1)) >> 3);
        return {lo + (x >> 3), G >> 3, hi, rev ? lo + hi - 1 : -1};
    }
    return {x, G, total, rev ? total - 1 : -1};
}

// Head activations (reference vqa_model.py:54: nn.ReLU() if activation == 'relu' else nn.SELU()).  act codes of the
// C-ABI: 0 none, 1 ReLU, 3 SELU (2 is the classifier's sigmoid-BCE epilogue).  The backward kernels take the
// derivative from the saved OUTPUT y: ReLU 1[y > 0]; SELU scale for y > 0, else y + scale*alpha (= scale*alpha*e^x).
#define VQA_ACT_RELU 1
#define VQA_ACT_SELU 3
__device__ __forceinline__ float selu_f(float x) {
    return x > 0.f ? 1.0507009873554805f * x : 1.7580993408473766f * expm1f(x);
}
__device__ __forceinline__ float act_apply(float x, int act) {
    return act == VQA_ACT_RELU ? fmaxf(x, 0.f) : (act == VQA_ACT_SELU ? selu_f(x) : x);
}
__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == VQA_ACT_SELU) return y > 0.f ? 1.0507009873554805f : y + 1.7580993408473766f;
    return y > 0.f ? 1.f : 0.f;
}

// i / d for 0 <= i < 2^22 and a runtime d, with inv = 1.0f / d from the host: a float multiply and two
// corrections instead of the ~40-instruction integer division sequence.
// (0 <= i < 2^23, 0 < d < 2^23 - __mul24 is the SIGNED 24-bit multiply, an operand with bit 23 set turns negative; the
// hosts' guards refuse larger shapes: the products are 24-bit multiplies - v_mul_lo_u32 is a quarter-rate instruction and
// the index arithmetic of the conv prologues and tile loops was full of it)
__device__ __forceinline__ int fastdivmod(int i, int d, float inv, int& rem) {
    int q = (int)((float)i * inv);
    int r = i - __mul24(q, d);
    const int up = r >= d, dn = r < 0;
    q += up - dn;
    r += (dn ? d : 0) - (up ? d : 0);
    rem = r;
    return q;
}
__device__ __forceinline__ int fastdiv(int i, int d, float inv) {
    int rem;
    return fastdivmod(i, d, inv, rem);
}

// Sum over the 16 lanes that share lane>>4 (the lanes holding different D columns of the same rows).
// DPP row operations (a DPP row = 16 lanes = exactly one lane>>4 group): 4 VALU ops, no LDS permute traffic.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    // mov_dpp (no "old" operand): the backend folds it into the consuming add as ONE v_add_f32_dpp; with
    // update_dpp(0, ...) every step was v_mov + v_mov_dpp + v_add (the controls used here read only valid lanes)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// four consecutive values per thread: float4 in, 8- or 16-byte out
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float* v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ void load4(const float* p, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// ---- conv epilogue arithmetic on 4 accumulator values of one lane (4 consecutive channels of one pixel).
// A wave64 VALU instruction costs its SIMD 4 cycles and the HBM-bound conv kernels are VALU-bound, so this is written
// in packed f32x2 operations (v_pk_add/mul/fma_f32) with ONE bf16 conversion per pair, reused for the statistics and
// for the store: per element 0.5 add (bias) + 0.5 cvt + 1 unpack + 0.5 mul (mask) + 0.5 add + 0.5 fma = 3.5 VALU,
// against 6-7 for the scalar form (separate convert and back-convert, select, multiply, two adds).
//   out   = round_T(acc + bias), optionally max(., 0)
//   s1/s2 += okf * out, (okf * out)^2       (okf = 1 for a pixel inside the image, 0 otherwise)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Packed4;
template <> struct Packed4<bf16> { unsigned lo, hi; };        // channels 0-1, 2-3 as bf16 pairs
template <> struct Packed4<float> { f32x2 lo, hi; };
template <typename T, bool RELU, bool STATS = true>
__device__ __forceinline__ Packed4<T> epilogue4(const f32x4& acc, const f32x4& bias, float okf, f32x2 (&s1)[2],
                                                f32x2 (&s2)[2]) {
    f32x2 p[2] = {f32x2{acc[0], acc[1]} + f32x2{bias[0], bias[1]}, f32x2{acc[2], acc[3]} + f32x2{bias[2], bias[3]}};
    const f32x2 ok2 = {okf, okf};
    Packed4<T> out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (RELU) { p[h].x = fmaxf(p[h].x, 0.f); p[h].y = fmaxf(p[h].y, 0.f); }
        f32x2 v;
        if constexpr (sizeof(T) == 2) {
            typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t t = {(bf16)p[h].x, (bf16)p[h].y};
            const unsigned bits = __builtin_bit_cast(unsigned, t);
            (h ? out.hi : out.lo) = bits;
            if (STATS) v = f32x2{__uint_as_float(bits << 16), __uint_as_float(bits & 0xFFFF0000u)};   // what is stored
        } else {
            (h ? out.hi : out.lo) = p[h];
            v = p[h];
        }
        if (STATS) {                                  // (the data gradient and the folded eval forward keep none)
            const f32x2 m = v * ok2;
            s1[h] += m;
            s2[h] = __builtin_elementwise_fma(m, m, s2[h]);
        }
    }
    return out;
}
// The same epilogue for a DATA-GRADIENT conv whose output d is the gradient w.r.t. the BatchNorm + ReLU output of the
// layer below (no pool in between): instead of the forward statistics it accumulates that layer's BatchNorm-backward
// sums (bn.hip: bn_bwd<APPLY = false, pool = 0>) from the stored d and the layer's raw conv output y at the same
// (pixel, channels):  dz = d * 1[scale*y + shift > 0],  s1 += dz,  s2 += dz * (y - mean) * invstd.
// y4 = this lane's 4 channels of y (as loaded: 8 bytes of bf16 / 16 bytes of f32), cf = {scale, shift, mean, invstd}.
template <typename T> struct Raw4;
template <> struct Raw4<bf16> { unsigned lo, hi; };
template <> struct Raw4<float> { f32x2 lo, hi; };
// xact (bf16 only, may be null): receives the layer's ACTIVATION at the same (pixel, channels), round_T(max(z, 0)) and
// zero outside the image - bit for bit what bn_relu_vec hands the forward conv - for the weight gradient formed in the
// same kernel (conv3x3_igemm_p, VAR 3).
template <typename T>
__device__ __forceinline__ Packed4<T> epilogue4_bnbwd(const f32x4& acc, float okf, const Raw4<T>& y4,
                                                      const f32x4 (&cf)[4], f32x2 (&s1)[2], f32x2 (&s2)[2],
                                                      unsigned* xact = nullptr) {
    const f32x2 p[2] = {f32x2{acc[0], acc[1]}, f32x2{acc[2], acc[3]}};
    Packed4<T> out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x2 v, yv;
        if constexpr (sizeof(T) == 2) {
            typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t t = {(bf16)p[h].x, (bf16)p[h].y};
            const unsigned bits = __builtin_bit_cast(unsigned, t);
            (h ? out.hi : out.lo) = bits;
            v = f32x2{__uint_as_float(bits << 16), __uint_as_float(bits & 0xFFFF0000u)};            // what is stored
            const unsigned yb = h ? y4.hi : y4.lo;
            yv = f32x2{__uint_as_float(yb << 16), __uint_as_float(yb & 0xFFFF0000u)};
        } else {
            (h ? out.hi : out.lo) = p[h];
            v = p[h];
            yv = h ? y4.hi : y4.lo;
        }
        const f32x2 sc = {cf[0][2 * h], cf[0][2 * h + 1]}, sh = {cf[1][2 * h], cf[1][2 * h + 1]},
                    mu = {cf[2][2 * h], cf[2][2 * h + 1]}, is = {cf[3][2 * h], cf[3][2 * h + 1]};
        const f32x2 z = __builtin_elementwise_fma(yv, sc, sh);
        if constexpr (sizeof(T) == 2) {
            if (xact) {
                typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t t = {(bf16)(okf != 0.f ? fmaxf(z.x, 0.f) : 0.f), (bf16)(okf != 0.f ? fmaxf(z.y, 0.f) : 0.f)};
                xact[h] = __builtin_bit_cast(unsigned, t);
            }
        }
        f32x2 dz = v * f32x2{okf, okf};
        dz.x = z.x > 0.f ? dz.x : 0.f;
        dz.y = z.y > 0.f ? dz.y : 0.f;
        const f32x2 xh = (yv - mu) * is;
        s1[h] += dz;
        s2[h] = __builtin_elementwise_fma(dz, xh, s2[h]);
    }
    return out;
}
template <typename T> __device__ __forceinline__ Raw4<T> buffer_load_raw4(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ Raw4<bf16> buffer_load_raw4<bf16>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return Raw4<bf16>{v.x, v.y};
}
template <> __device__ __forceinline__ Raw4<float> buffer_load_raw4<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return Raw4<float>{f32x2{__uint_as_float(v.x), __uint_as_float(v.y)}, f32x2{__uint_as_float(v.z), __uint_as_float(v.w)}};
}
template <typename T> __device__ __forceinline__ void store_packed4(T* p, const Packed4<T>& o);
template <> __device__ __forceinline__ void store_packed4<bf16>(bf16* p, const Packed4<bf16>& o) {
    *reinterpret_cast<uint2*>(p) = make_uint2(o.lo, o.hi);
}
template <> __device__ __forceinline__ void store_packed4<float>(float* p, const Packed4<float>& o) {
    *reinterpret_cast<float4*>(p) = make_float4(o.lo.x, o.lo.y, o.hi.x, o.hi.y);
}
// ---- buffer-resource addressing.  A raw buffer access whose byte offset is >= the buffer's size is dropped by the
// hardware: a load returns zeros, a store writes nothing.  The conv kernels use that for the zero padding and for
// the tile overhang: an out-of-image element gets the offset `size` (one select) instead of a clamped address, an AND
// mask over the loaded data at the LDS write and a branch around the store - and straight-line code is also what lets
// the compiler count its s_waitcnt vmcnt exactly (behind a branch that may skip the stores it waits for them too).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buffer_store16(__amdgpu_buffer_rsrc_t r, unsigned off, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, r, off, 0, 0);
}
template <typename T> __device__ __forceinline__ void buffer_store_packed4(__amdgpu_buffer_rsrc_t r, unsigned off,
                                                                           const Packed4<T>& o);
template <> __device__ __forceinline__ void buffer_store_packed4<bf16>(__amdgpu_buffer_rsrc_t r, unsigned off,
                                                                       const Packed4<bf16>& o) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.lo, o.hi}, r, off, 0, 0);
}
template <> __device__ __forceinline__ void buffer_store_packed4<float>(__amdgpu_buffer_rsrc_t r, unsigned off,
                                                                        const Packed4<float>& o) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o.lo.x), __float_as_uint(o.lo.y),
                                                 __float_as_uint(o.hi.x), __float_as_uint(o.hi.y)}, r, off, 0, 0);
}

// Sum of each of four values over the 16 lanes of a DPP row, every lane receiving the sums.  One v_add_f32_dpp per
// step and value: written through the compiler's update_dpp builtin each step was v_mov + v_mov_dpp + v_add.  The
// s_nop covers the VALU-write -> DPP-read hazard for the first row of adds (the assembler does not look inside an
// asm block); within the block three other instructions separate a value's write from its next DPP read.
__device__ __forceinline__ void sum16_x4(float& a, float& b, float& c, float& d) {
#define VQA_DPP4(CTRL)                                                            \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n"              \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n"              \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n"              \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n"
    asm("s_nop 1\n" VQA_DPP4("row_ror:8") VQA_DPP4("row_ror:4") VQA_DPP4("quad_perm:[2,3,0,1]")
        VQA_DPP4("quad_perm:[1,0,3,2]")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef VQA_DPP4
}

// BatchNorm + ReLU forward on one 16-byte vector of T, a = round_T(max(fma(y, scale, shift), 0)): exactly the
// arithmetic of bn_act_pool_fwd (pool = 0), applied by a CONSUMER of the raw conv output y while it stages its input,
// so the normalised activation is never written to HBM.  sc / sh: the 16 / sizeof(T) channels of the vector.
template <typename T> __device__ __forceinline__ uint4 bn_relu_vec(const uint4& v, const float* sc, const float* sh);
template <> __device__ __forceinline__ uint4 bn_relu_vec<bf16>(const uint4& v, const float* sc, const float* sh) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[i] << 16), sc[2 * i], sh[2 * i]), 0.f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[i] & 0xFFFF0000u), sc[2 * i + 1], sh[2 * i + 1]), 0.f);
        typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t t = {(bf16)lo, (bf16)hi};
        o[i] = __builtin_bit_cast(unsigned, t);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ uint4 bn_relu_vec<float>(const uint4& v, const float* sc, const float* sh) {
    return make_uint4(__float_as_uint(fmaxf(fmaf(__uint_as_float(v.x), sc[0], sh[0]), 0.f)),
                      __float_as_uint(fmaxf(fmaf(__uint_as_float(v.y), sc[1], sh[1]), 0.f)),
                      __float_as_uint(fmaxf(fmaf(__uint_as_float(v.z), sc[2], sh[2]), 0.f)),
                      __float_as_uint(fmaxf(fmaf(__uint_as_float(v.w), sc[3], sh[3]), 0.f)));
}

template <typename T> __device__ __forceinline__ void store_vec4(T* p, const float* v);
template <> __device__ __forceinline__ void store_vec4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_vec4<bf16>(bf16* p, const float* v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}
