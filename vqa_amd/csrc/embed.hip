// Deterministic backward of nn.Embedding (reference gru.py:48, :103; autograd's scatter-add into [V][E]).
//
// The obvious kernel - one float atomicAdd per (token, element) - is order-dependent in its last bits (the one
// non-deterministic gradient of the step in round 1) and serialises on hot rows: the <pad> row takes ~40 % of all
// tokens of a VQA batch.  Here the tokens are sorted by (id, position) once, every row's occurrences are summed in that
// fixed order in blocks of RB rows (one wave per block, whole 16-byte row pieces, four rows in flight), and rows with
// more than one block are combined from per-block partial rows in block order: no atomics, the same bits every run,
// and a hot row's work is spread over count / RB waves.
//   1. embed_sort    one workgroup: bitonic sort of (id << 32 | position) in LDS, run / block boundaries -> item list
//   2. embed_accum   one wave per item (<= RB consecutive occurrences of one id): sum -> demb row or partial row
//   3. embed_combine one wave per multi-item row: partial rows in order -> demb row
// Chunks of at most NC = 8192 tokens (96 KB of LDS); later chunks add to what earlier ones wrote.
#include "common.h"

namespace {

constexpr int RB = 16;          // occurrences per item
constexpr int NC = 8192;        // tokens per chunk (LDS: 12 bytes per token: 64-bit key + run start)

struct EmbedWs {                // layout of the caller's workspace for a chunk of n tokens (P = n rounded up to 2^k)
    int* order;                 // [n]   dx row of the s-th sorted token
    int* ids;                   // [n]   its vocabulary row
    int* item_pos;              // [n+1] first sorted index of item k; item_pos[n_items] = n
    int* n_items;               // [1]
    float* part;                // [n][E] partial rows (only items of multi-item rows are written)
};

// q int64 [B][T]; token at flat index p = b*T + t has dx row t*B + b.  Tokens with an id outside [0, V) are dropped
// (sorted to the end with key ~0: the forward made their rows NaN, see vqa_embed_fwd).
__global__ __launch_bounds__(1024) void embed_sort(const long* q, int B, int Tn, int V, int p0, int n, int P, EmbedWs w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);          // [P]
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += 1024) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const long id = q[p0 + i];
            if (id >= 0 && id < V) k = ((unsigned long long)id << 32) | (unsigned)(p0 + i);
        }
        key[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = key[i], b = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    // valid (in-vocabulary) tokens come first; items start at run starts and every RB-th occurrence of a run.
    // Each thread owns a contiguous stretch of sorted positions: per-thread counts -> exclusive scan -> compaction.
    __shared__ int wsum[16];
    __shared__ int total_valid;
    const int per = (P + 1023) / 1024, s0 = tid * per, s1 = min(s0 + per, P);
    // run-relative rank needs the run start: walk back from s0 (runs are short except hot rows: bounded by RB steps
    // is not enough for the <pad> row), so the rank is taken modulo RB from the run start found by a block-wide pass:
    // start[s] = s if run start else start[s-1]  ==  max-scan of (flag ? s : -1); done here with the per-thread
    // stretch + a scan of the stretch maxima.
    int* runstart = reinterpret_cast<int*>(smem + (size_t)P * 8);                   // [P]
    int last = -1;
    for (int s = s0; s < s1; ++s) {
        const unsigned long long ks = key[s];
        const bool valid = ks != ~0ull;
        const bool first = valid && (s == 0 || (unsigned)(key[s - 1] >> 32) != (unsigned)(ks >> 32));
        if (first) last = s;
        runstart[s] = valid ? last : -2;                   // -1: run began in an earlier stretch (fixed below)
    }
    __shared__ int stretch_last[1024];
    stretch_last[tid] = last;
    __syncthreads();
    // inclusive max-scan over the 1024 stretch values (Hillis-Steele in LDS, 10 rounds)
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? stretch_last[tid - off] : -1;
        __syncthreads();
        if (v > stretch_last[tid]) stretch_last[tid] = v;
        __syncthreads();
    }
    const int carry = tid > 0 ? stretch_last[tid - 1] : -1;
    int cnt = 0;
    for (int s = s0; s < s1; ++s) {
        if (runstart[s] == -2) continue;
        if (runstart[s] == -1) runstart[s] = carry;
        if (((s - runstart[s]) % RB) == 0) ++cnt;
    }
    // exclusive scan of cnt over threads
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < (tid >> 6); ++i) base += wsum[i];
    int k = base + incl - cnt;
    int nvalid = 0;
    for (int s = s0; s < s1; ++s) {
        if (runstart[s] == -2) continue;
        ++nvalid;
        const unsigned long long ks = key[s];
        const unsigned p = (unsigned)ks;
        const int b = p / Tn, t = p - b * Tn;
        w.order[s] = t * B + b;
        w.ids[s] = (int)(ks >> 32);
        if (((s - runstart[s]) % RB) == 0) w.item_pos[k++] = s;
    }
    if (tid == 0) total_valid = 0;
    __syncthreads();
    atomicAdd(&total_valid, nvalid);                       // integer: order-independent
    __syncthreads();
    if (tid == 1023) {
        w.n_items[0] = base + incl;
        w.item_pos[base + incl] = total_valid;
    }
}

// one wave per item; lane l owns the float4 pieces l, l + 64, ... of the row
template <bool VEC>
__global__ __launch_bounds__(256) void embed_accum(const float* __restrict__ dx, int lddx, int E, int accumulate,
                                                   EmbedWs w, float* __restrict__ demb) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= w.n_items[0]) return;
    const int s0 = w.item_pos[k], s1 = w.item_pos[k + 1];
    const int id = w.ids[s0];
    const bool head = s0 == 0 || w.ids[s0 - 1] != id;                      // first item of its row
    const bool more = s1 < w.item_pos[w.n_items[0]] && w.ids[s1] == id;    // the row continues in the next item
    float* dst = (head && !more) ? demb + (long)id * E : w.part + (long)k * E;
    const bool add = (head && !more) && accumulate;
    constexpr int W = VEC ? 4 : 1;
    for (int e0 = lane * W; e0 < E; e0 += 64 * W) {
        float acc[W];
#pragma unroll
        for (int i = 0; i < W; ++i) acc[i] = 0.f;
        for (int s = s0; s < s1; s += 4) {                                 // four rows in flight, added in order
            float v[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = w.order[s + u < s1 ? s + u : s1 - 1];
                if constexpr (VEC) load4(dx + (long)row * lddx + e0, v[u]);
                else v[u][0] = dx[(long)row * lddx + e0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s + u < s1) {
#pragma unroll
                    for (int i = 0; i < W; ++i) acc[i] += v[u][i];
                }
        }
        if (add) {
#pragma unroll
            for (int i = 0; i < W; ++i) acc[i] += dst[e0 + i];
        }
        if constexpr (VEC) store4<float>(dst + e0, acc);
        else dst[e0] = acc[0];
    }
}

// one wave per item that heads a multi-item row: its partial rows, in item order
template <bool VEC>
__global__ __launch_bounds__(256) void embed_combine(int E, int accumulate, EmbedWs w, float* __restrict__ demb) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_items = w.n_items[0];
    if (k >= n_items) return;
    const int s0 = w.item_pos[k], s1 = w.item_pos[k + 1], nvalid = w.item_pos[n_items];
    const int id = w.ids[s0];
    if (!(s0 == 0 || w.ids[s0 - 1] != id)) return;                         // not a row head
    if (!(s1 < nvalid && w.ids[s1] == id)) return;                         // single-item row: written by embed_accum
    int m = 1;
    while (k + m < n_items && w.ids[w.item_pos[k + m]] == id) ++m;
    constexpr int W = VEC ? 4 : 1;
    float* dst = demb + (long)id * E;
    for (int e0 = lane * W; e0 < E; e0 += 64 * W) {
        float acc[W];
#pragma unroll
        for (int i = 0; i < W; ++i) acc[i] = accumulate ? dst[e0 + i] : 0.f;
        for (int j = 0; j < m; j += 4) {
            float v[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* src = w.part + (long)(k + (j + u < m ? j + u : m - 1)) * E + e0;
                if constexpr (VEC) load4(src, v[u]);
                else v[u][0] = src[0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + u < m) {
#pragma unroll
                    for (int i = 0; i < W; ++i) acc[i] += v[u][i];
                }
        }
        if constexpr (VEC) store4<float>(dst + e0, acc);
        else dst[e0] = acc[0];
    }
}

}  // namespace

// bytes of workspace vqa_embed_bwd needs for B*T tokens of width E
extern "C" int vqa_embed_bwd_ws_bytes(int B, int T, int E) {
    if (B < 1 || T < 1 || E < 1) return -1;
    const long n = (long)B * T < NC ? (long)B * T : NC;
    const long bytes = ((n * E + 3) / 4 * 4) * 4 + (3 * n + 8) * 4;
    return bytes < (1L << 31) ? (int)bytes : -1;
}

extern "C" int vqa_embed_bwd(void* stream, const long* q, const float* dx, int B, int T, int E, int V, int lddx,
                             float* demb, void* ws) {
    if (B < 1 || T < 1 || E < 1 || V < 1 || !q || !dx || !demb || !ws) return VQA_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long N = (long)B * T;
    if (N >= (1L << 31)) return VQA_ERR_ARG;
    const long nmax = N < NC ? N : NC;
    EmbedWs w;
    w.part = static_cast<float*>(ws);
    int* ip = reinterpret_cast<int*>(w.part + (nmax * E + 3) / 4 * 4);
    w.order = ip; w.ids = ip + nmax; w.item_pos = ip + 2 * nmax; w.n_items = ip + 3 * nmax + 4;
    const bool vec = E % 4 == 0 && lddx % 4 == 0 && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(demb)) & 15) == 0;
    static bool attr_set[64] = {};                 // per device; benign race (idempotent)
    for (long p0 = 0; p0 < N; p0 += NC) {
        const int n = (int)(N - p0 < NC ? N - p0 : NC);
        int P = 1024;                              // (>= the block size: every thread owns a stretch)
        while (P < n) P <<= 1;
        const size_t lds = (size_t)P * 12;
        if (lds > 64 * 1024) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= 64 || !attr_set[dev]) {
                (void)hipFuncSetAttribute((const void*)embed_sort, hipFuncAttributeMaxDynamicSharedMemorySize, NC * 12);
                if (dev >= 0 && dev < 64) attr_set[dev] = true;
            }
        }
        hipLaunchKernelGGL(embed_sort, dim3(1), dim3(1024), lds, s, q, B, T, V, (int)p0, n, P, w);
        const dim3 grid((n + 3) / 4);
        const int acc = p0 > 0 ? 1 : 0;
        if (vec) {
            hipLaunchKernelGGL(embed_accum<true>, grid, dim3(256), 0, s, dx, lddx, E, acc, w, demb);
            hipLaunchKernelGGL(embed_combine<true>, grid, dim3(256), 0, s, E, acc, w, demb);
        } else {
            hipLaunchKernelGGL(embed_accum<false>, grid, dim3(256), 0, s, dx, lddx, E, acc, w, demb);
            hipLaunchKernelGGL(embed_combine<false>, grid, dim3(256), 0, s, E, acc, w, demb);
        }
    }
    CHECK_LAUNCH();
    return VQA_OK;
}
