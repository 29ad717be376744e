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
// Chunks of at most NC = 8192 tokens (<= 64 KB of LDS keys); later chunks add to what earlier ones wrote.
#include "common.h"

namespace {

constexpr int RB = 16;          // occurrences per item
constexpr int NC = 8192;        // tokens per chunk (LDS: one 4- or 8-byte key per token)

struct EmbedWs {                // layout of the caller's workspace for a chunk of n tokens (P = n rounded up to 2^k)
    int* order;                 // [n]   dx row of the s-th sorted token
    int* ids;                   // [n]   its vocabulary row
    int* item_pos;              // [n+1] first sorted index of item k; item_pos[n_items] = n
    int* n_items;               // [1]
    float* part;                // [n][E] partial rows (only items of multi-item rows are written)
};

// LDS traffic of one wave is executed in order: between wave-private LDS stages only the compiler must be held back
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// inclusive scan over the 1024 threads of the block (OP: 0 = max, 1 = sum); returns the inclusive value, *excl gets the
// exclusive one (identity for thread 0); red = 16 ints of LDS
template <int OP>
__device__ __forceinline__ int block_scan(int v, int* red, int* excl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ident = OP == 0 ? -1 : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o);
        if (lane >= o) incl = OP == 0 ? (u > incl ? u : incl) : incl + u;
    }
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    int base = ident;
    for (int i = 0; i < wave; ++i) base = OP == 0 ? (red[i] > base ? red[i] : base) : base + red[i];
    __syncthreads();                                       // red may be reused by the next scan
    int prev = __shfl_up(incl, 1);
    if (lane == 0) prev = ident;
    *excl = OP == 0 ? (prev > base ? prev : base) : prev + base;
    return OP == 0 ? (incl > base ? incl : base) : incl + base;
}

// q int64 [B][T]; token at flat index p = b*T + t has dx row t*B + b.  Tokens with an id outside [0, V) are dropped
// (sorted to the end with the all-ones key: the forward made their rows NaN, see vqa_embed_fwd).
// Key = id * P + (index inside the chunk): 32 bits when V * P < 2^32, else 64.  Bitonic sort over P = 2^k >= n keys in
// LDS; each wave owns a contiguous segment of P / 16 keys, and the compare-exchange stages whose stride stays inside a
// segment need no workgroup barrier (57 of the 78 stages at P = 4096 ... 68 with the segment at 256 keys).
template <typename KeyT>
__global__ __launch_bounds__(1024) void embed_sort(const long* q, int B, int Tn, int V, int p0, int n, int P, int logP,
                                                   EmbedWs w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KeyT* key = reinterpret_cast<KeyT*>(smem);             // [P]
    __shared__ int red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KeyT none = ~(KeyT)0;
    for (int i = tid; i < P; i += 1024) {
        KeyT k = none;
        if (i < n) {
            const long id = q[p0 + i];
            if (id >= 0 && id < V) k = ((KeyT)id << logP) | (KeyT)i;
        }
        key[i] = k;
    }
    __syncthreads();
    const int SEG = P >> 4, seg0 = wave * SEG;
    auto cmpx = [&](int i, int j, int k) {
        const int l = i + j;
        const KeyT a = key[i], b = key[l];
        if ((a > b) == ((i & k) == 0)) { key[i] = b; key[l] = a; }
    };
    for (int k = 2, lk = 1; k <= P; k <<= 1, ++lk) {
        int j = k >> 1, lj = lk - 1;                       // j = 2^lj: pair pi -> i = (pi / j) * 2j + pi % j by shifts
        for (; j >= SEG; j >>= 1, --lj) {                  // partners in other waves' segments
            for (int pi = tid; pi < (P >> 1); pi += 1024) cmpx(((pi >> lj) << (lj + 1)) + (pi & (j - 1)), j, k);
            __syncthreads();
        }
        for (; j > 0; j >>= 1, --lj) {                     // inside this wave's segment
            for (int pi = lane; pi < (SEG >> 1); pi += 64) cmpx(seg0 + ((pi >> lj) << (lj + 1)) + (pi & (j - 1)), j, k);
            wave_lds_sync();
        }
        __syncthreads();
    }
    // valid keys come first.  Items start at run starts and at every RB-th occurrence of a run.  Each thread owns the
    // EPT consecutive sorted positions [tid * EPT, ..): last run start per stretch -> max-scan -> item counts -> sum-scan
    const int EPT = P >> 10, s0 = tid * EPT;
    const KeyT idmask = ~(((KeyT)1 << logP) - 1);
    int last = -1, nvalid = 0;
    KeyT prevk = s0 > 0 ? key[s0 - 1] : none;
    for (int e = 0; e < EPT; ++e) {
        const KeyT ks = key[s0 + e];
        if (ks != none) {
            ++nvalid;
            if (s0 + e == 0 || ((prevk ^ ks) & idmask) != 0) last = s0 + e;
        }
        prevk = ks;
    }
    int carry;
    block_scan<0>(last, red, &carry);
    int cnt = 0, cur = carry;
    prevk = s0 > 0 ? key[s0 - 1] : none;
    for (int e = 0; e < EPT; ++e) {
        const KeyT ks = key[s0 + e];
        if (ks != none) {
            if (s0 + e == 0 || ((prevk ^ ks) & idmask) != 0) cur = s0 + e;
            if (((s0 + e - cur) % RB) == 0) ++cnt;
        }
        prevk = ks;
    }
    int kbase;
    const int kincl = block_scan<1>(cnt, red, &kbase);
    int vexcl;
    const int vincl = block_scan<1>(nvalid, red, &vexcl);
    int k = kbase;
    cur = carry;
    prevk = s0 > 0 ? key[s0 - 1] : none;
    for (int e = 0; e < EPT; ++e) {
        const int s = s0 + e;
        const KeyT ks = key[s];
        if (ks != none) {
            if (s == 0 || ((prevk ^ ks) & idmask) != 0) cur = s;
            const unsigned p = (unsigned)p0 + (unsigned)(ks & (((KeyT)1 << logP) - 1));
            const int b = p / Tn, t = p - b * Tn;
            w.order[s] = t * B + b;
            w.ids[s] = (int)(ks >> logP);
            if (((s - cur) % RB) == 0) w.item_pos[k++] = s;
        }
        prevk = ks;
    }
    if (tid == 1023) {
        w.n_items[0] = kincl;
        w.item_pos[kincl] = vincl;
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

// one WORKGROUP per item that heads a multi-item row (every other workgroup leaves at once): wave v sums the v-th
// quarter of the row's partial rows in order, eight in flight; the four wave sums are added in wave order
template <bool VEC>
__global__ __launch_bounds__(256) void embed_combine(int E, int accumulate, EmbedWs w, float* __restrict__ demb) {
    extern __shared__ __attribute__((aligned(16))) float comb[];           // [3][E]: the sums of waves 1..3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x;
    const int n_items = w.n_items[0];
    if (k >= n_items) return;
    const int s0 = w.item_pos[k], s1 = w.item_pos[k + 1], nvalid = w.item_pos[n_items];
    const int id = w.ids[s0];
    if (!(s0 == 0 || w.ids[s0 - 1] != id)) return;                         // not a row head
    if (!(s1 < nvalid && w.ids[s1] == id)) return;                         // single-item row: written by embed_accum
    // number of items of the row: its last occurrence by binary search over the sorted ids
    int lo = s1, hi = nvalid;                                              // first sorted index with another id
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (w.ids[mid] == id) lo = mid + 1; else hi = mid;
    }
    const int m = (lo - s0 + RB - 1) / RB;
    const int per = (m + 3) / 4, j0 = wave * per, j1 = min(j0 + per, m);
    constexpr int W = VEC ? 4 : 1, U = 8;
    float* dst = demb + (long)id * E;
    // Block-uniform trip count: every lane runs ceil(E / (64 W)) rounds and lanes whose columns lie beyond E are
    // predicated off (loads clamped to column 0, nothing stored) - the two barriers per round are then reached by all
    // 256 threads in convergent control flow (E = 300, W = 4: lanes 11..63 are idle in the second round).
    const int rounds = (E + 64 * W - 1) / (64 * W);
    for (int rd = 0; rd < rounds; ++rd) {
        const int e0 = rd * 64 * W + lane * W;
        const bool live = e0 < E;
        const int el = live ? e0 : 0;
        float acc[W];
#pragma unroll
        for (int i = 0; i < W; ++i) acc[i] = 0.f;
        for (int j = j0; j < j1; j += U) {
            float v[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* src = w.part + (long)(k + (j + u < j1 ? j + u : j1 - 1)) * E + el;
                if constexpr (VEC) load4(src, v[u]);
                else v[u][0] = src[0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j + u < j1) {
#pragma unroll
                    for (int i = 0; i < W; ++i) acc[i] += v[u][i];
                }
        }
        if (wave > 0 && live) {
#pragma unroll
            for (int i = 0; i < W; ++i) comb[(wave - 1) * E + e0 + i] = acc[i];
        }
        __syncthreads();
        if (wave == 0 && live) {
#pragma unroll
            for (int i = 0; i < W; ++i) {
                float t = acc[i] + comb[e0 + i];
                t += comb[E + e0 + i];
                t += comb[2 * E + e0 + i];
                dst[e0 + i] = accumulate ? dst[e0 + i] + t : t;
            }
        }
        __syncthreads();
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
    for (long p0 = 0; p0 < N; p0 += NC) {
        const int n = (int)(N - p0 < NC ? N - p0 : NC);
        int P = 1024;                              // (>= the block size: every thread owns a stretch)
        while (P < n) P <<= 1;
        int logP = 10;
        while ((1 << logP) < P) ++logP;
        const bool k32 = ((unsigned long long)V << logP) < 0xFFFFFFFFull;
        const size_t lds = (size_t)P * (k32 ? 4 : 8);      // <= 64 KB of keys, beside the kernel's static red[16]
        if (lds + 64 > 64 * 1024) {                        // (64-bit keys at P = 8192: over the default 64 KB limit)
            if (hipFuncSetAttribute(k32 ? (const void*)embed_sort<unsigned> : (const void*)embed_sort<unsigned long long>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return VQA_ERR_LAUNCH;
        }
        if (k32) hipLaunchKernelGGL(embed_sort<unsigned>, dim3(1), dim3(1024), lds, s, q, B, T, V, (int)p0, n, P, logP, w);
        else hipLaunchKernelGGL(embed_sort<unsigned long long>, dim3(1), dim3(1024), lds, s, q, B, T, V, (int)p0, n, P, logP, w);
        const dim3 grid((n + 3) / 4);
        const int acc = p0 > 0 ? 1 : 0;
        const size_t clds = (size_t)3 * E * sizeof(float);
        if (clds > 64 * 1024) return VQA_ERR_ARG;
        // (an upper bound on the items of a chunk: occurrences / RB + distinct ids <= n; idle workgroups leave at once)
        if (vec) {
            hipLaunchKernelGGL(embed_accum<true>, grid, dim3(256), 0, s, dx, lddx, E, acc, w, demb);
            hipLaunchKernelGGL(embed_combine<true>, dim3(n), dim3(256), clds, s, E, acc, w, demb);
        } else {
            hipLaunchKernelGGL(embed_accum<false>, grid, dim3(256), 0, s, dx, lddx, E, acc, w, demb);
            hipLaunchKernelGGL(embed_combine<false>, dim3(n), dim3(256), clds, s, E, acc, w, demb);
        }
    }
    CHECK_LAUNCH();
    return VQA_OK;
}
