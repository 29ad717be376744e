// The per-step form of the nn.GRU recurrence (reference gru.py:104) issued by ONE call: T x (recurrent GEMM + fused
// gate kernel) forward, T x (gate backward + GEMM) for the BPTT.  Same kernels, same order and same results as calling
// vqa_gemm / vqa_gru_gate_fwd / vqa_gru_gate_bwd once per time step; what it removes is the host cost of 4 T launches
// issued one by one through the binding (the training step is host-bound: ~150 launches, ~10 us each from Python
// against ~3 us from here).
#include "common.h"
#include "../../include/vqa_hip.h"

extern "C" int vqa_gru_fwd_steps(void* stream, int dtype, const float* gi, const void* w_hh, int ldw, const float* b_hh,
                                 float* h_f, void* h_t, float* save, float* gh, float* splitk_ws, int splits, int B,
                                 int T, int H) {
    if (B < 1 || T < 1 || H < 1 || !gi || !w_hh || !h_f || !h_t || !save || !gh) return VQA_ERR_ARG;
    const size_t esz = dtype == VQA_BF16 ? 2 : 4;
    const long bh = (long)B * H;
    for (int t = 0; t < T; ++t) {
        const char* hp_t = static_cast<const char*>(h_t) + (size_t)t * bh * esz;
        char* hn_t = static_cast<char*>(h_t) + (size_t)(t + 1) * bh * esz;
        int st = vqa_gemm(stream, dtype, 0, B, 3 * H, H, hp_t, H, w_hh, ldw, gh, 3 * H, VQA_F32, b_hh, nullptr, 1, 0, 0,
                          nullptr, 0, splitk_ws, splits, nullptr, 0, 0, nullptr);
        if (st != VQA_OK) return st;
        st = vqa_gru_gate_fwd(stream, dtype, gi + (long)t * B * 3 * H, gh, h_f + t * bh, B, H, h_f + (t + 1) * bh, hn_t,
                              save + (long)t * B * 4 * H);
        if (st != VQA_OK) return st;
    }
    return VQA_OK;
}

extern "C" int vqa_gru_bwd_steps(void* stream, int dtype, const float* dh_last, const float* save, const float* h_f,
                                 const void* w_hh_t, int ldw, void* dgi, void* dgh, float* dh_ws, float* splitk_ws,
                                 int splits, int B, int T, int H, const float* dh_seq) {
    if (B < 1 || T < 1 || H < 1 || !dh_last || !save || !h_f || !w_hh_t || !dgi || !dgh || !dh_ws) return VQA_ERR_ARG;
    const size_t esz = dtype == VQA_BF16 ? 2 : 4;
    const long bh = (long)B * H, b3 = (long)B * 3 * H;
    const float* dh = dh_last;
    for (int t = T - 1; t >= 0; --t) {
        float* dprev = dh_ws + (t & 1) * bh;              // (never the buffer dh points to: they alternate)
        char* gi_t = static_cast<char*>(dgi) + (size_t)t * b3 * esz;
        char* gh_t = static_cast<char*>(dgh) + (size_t)t * b3 * esz;
        // dh_seq (multi-layer GRU, may be NULL): [T][B][H] gradient of the layer above w.r.t. every step's output; the
        // caller has added dh_seq[T-1] into dh_last, step t adds dh_seq[t-1] to the gradient of h_{t-1}
        int st = vqa_gru_gate_bwd(stream, dtype, dh, save + (long)t * B * 4 * H, h_f + t * bh, B, H, gi_t, gh_t, dprev,
                                  (dh_seq && t > 0) ? dh_seq + (t - 1) * bh : nullptr);
        if (st != VQA_OK) return st;
        if (t > 0) {                                      // dh_{t-1} += dgh_t @ W_hh
            st = vqa_gemm(stream, dtype, 0, B, H, 3 * H, gh_t, 3 * H, w_hh_t, ldw, dprev, H, VQA_F32, nullptr, nullptr, 1,
                          0, 1, nullptr, 0, splitk_ws, splits, nullptr, 0, 0, nullptr);
            if (st != VQA_OK) return st;
        }
        dh = dprev;
    }
    return VQA_OK;
}
