// Plan replay: ONE C-ABI call issues a whole recorded sequence of entry-point calls (the image encoder's ten-layer forward,
// or its backward with its side-stream weight gradients) - the host side of a training step was 3.3 ms of Python /
// ctypes per 4.1 ms of GPU work, most of it the ~25 us a binding call costs against the ~4 us of the launch inside it.
//
// A plan is a POD array of 64-bit words that vqa_amd/plan.py records by running the ordinary per-launch path once:
//   [PLAN_MAGIC, n_ops] then per op [entry, stream index, n slots] + per slot [kind, value (+ inline array payload)].
// Pointers are stored RELOCATABLE, as (base index, byte offset): every replay passes the table of base addresses of
// that step (its scratch arena, the inputs, the parameters, the gradient buffer ...).  The library stays stateless and
// allocation-free: nothing is cached between calls, streams come in as handles, and the event of a cross-stream
// ordering point lives only for the two calls that use it.
// Host code only (no kernels here); the entry points it calls are the ones include/vqa_hip.h declares.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/vqa_hip.h"

namespace {

union PlanSlot {
    long i;
    float f;
    double d;
    void* p;
};
struct PlanEntry {
    const char* name;
    int (*fn)(void* stream, const PlanSlot* a);
    int nargs;
};

#include "plan_thunks.inc"

constexpr long PLAN_MAGIC = 0x56514150314cL;      // "VQAP1L"
enum { K_INT = 0, K_F32 = 1, K_F64 = 2, K_PTR = 3, K_NULL = 4, K_PTRARR = 5, K_INTARR = 6 };
constexpr long OP_SYNC = -1;                      // [OP_SYNC, waiter stream, 1, K_INT, signalling stream]
constexpr int MAX_SLOTS = 32, MAX_PTRS = 256, MAX_INTS = 1024;

inline void* resolve(long ref, void* const* bases, int n_bases, bool& ok) {
    if (ref == -1) return nullptr;
    const long b = (ref >> 48) & 0x7fff, off = ref & 0xffffffffffffL;
    if (b >= n_bases || !bases[b]) { ok = false; return nullptr; }
    return static_cast<char*>(bases[b]) + off;
}

}  // namespace

extern "C" int vqa_plan_opcode(const char* name) {
    for (int i = 0; i < kPlanEntryCount; ++i)
        if (strcmp(kPlanEntries[i].name, name) == 0) return i;
    return -1;
}

// status 0, or (status of the failing op) | (its 1-based index << 8); a malformed plan is status 1 with index 0
extern "C" int vqa_run_plan(const long* words, long n_words, void* const* bases, int n_bases, void* const* streams,
                            int n_streams, void* const* events, int n_events) {
    if (!words || n_words < 2 || words[0] != PLAN_MAGIC || !bases || !streams || n_streams < 1) return 1;
    const long n_ops = words[1];
    long w = 2;
    void* parr[MAX_PTRS];
    int iarr[MAX_INTS];
    int next_event = 0;
    for (long op = 0; op < n_ops; ++op) {
        if (w + 3 > n_words) return 1;
        const long entry = words[w], sidx = words[w + 1], nslots = words[w + 2];
        w += 3;
        if (sidx < 0 || sidx >= n_streams || nslots < 0 || nslots > MAX_SLOTS) return 1;
        PlanSlot slot[MAX_SLOTS];
        int np = 0, ni = 0;
        bool ok = true;
        for (long s = 0; s < nslots; ++s) {
            if (w + 2 > n_words) return 1;
            const long kind = words[w], val = words[w + 1];
            w += 2;
            switch (kind) {
                case K_INT: slot[s].i = val; break;
                case K_F32: { uint32_t b = (uint32_t)val; slot[s].i = 0; memcpy(&slot[s].f, &b, 4); break; }
                case K_F64: memcpy(&slot[s].d, &val, 8); break;
                case K_PTR: slot[s].p = resolve(val, bases, n_bases, ok); break;
                case K_NULL: slot[s].p = nullptr; break;
                case K_PTRARR:
                    if (val < 0 || np + val > MAX_PTRS || w + val > n_words) return 1;
                    slot[s].p = parr + np;
                    for (long k = 0; k < val; ++k) parr[np++] = resolve(words[w++], bases, n_bases, ok);
                    break;
                case K_INTARR:
                    if (val < 0 || ni + val > MAX_INTS || w + val > n_words) return 1;
                    slot[s].p = iarr + ni;
                    for (long k = 0; k < val; ++k) iarr[ni++] = (int)words[w++];
                    break;
                default: return 1;
            }
        }
        if (!ok) return 1 | (int)((op + 1) << 8);
        int status;
        if (entry == OP_SYNC) {
            // stream `sidx` waits for everything issued so far on stream slot[0].i
            if (nslots != 1 || slot[0].i < 0 || slot[0].i >= n_streams) return 1;
            hipEvent_t ev;
            status = 2;
            if (events && next_event < n_events) {
                // the caller's own events (hipEvent_t handles, one per ordering point, reusable call after call)
                ev = static_cast<hipEvent_t>(events[next_event++]);
                status = (hipEventRecord(ev, static_cast<hipStream_t>(streams[slot[0].i])) == hipSuccess &&
                          hipStreamWaitEvent(static_cast<hipStream_t>(streams[sidx]), ev, 0) == hipSuccess) ? 0 : 2;
            } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                const bool good = hipEventRecord(ev, static_cast<hipStream_t>(streams[slot[0].i])) == hipSuccess &&
                                  hipStreamWaitEvent(static_cast<hipStream_t>(streams[sidx]), ev, 0) == hipSuccess;
                (void)hipEventDestroy(ev);            // (released by the runtime once the recorded work has completed)
                status = good ? 0 : 2;
            }
        } else {
            if (entry < 0 || entry >= kPlanEntryCount || kPlanEntries[entry].nargs != nslots) return 1;
            status = kPlanEntries[entry].fn(streams[sidx], slot);
        }
        if (status != 0) return status | (int)((op + 1) << 8);
    }
    return w == n_words ? 0 : 1;
}
