// Layout probes used by tests/test_gpu_layout.py to pin the hardware maps the kernels rely on
// (ds_read_b64_tr_b16 lane/row/column map) with exact integer data.
#include "common.h"

namespace {
// tile: [rows][16] bf16 row-major in LDS (32-byte rows).  Each 16-lane group g reads the 4x16 block of rows
// 4g..4g+3: lane i = 4q+p supplies the address of row q, columns 4p..4p+3.  out[lane][e] = element e received.
__global__ void probe_tr16(const bf16* in, float* out) {
    __shared__ __attribute__((aligned(16))) bf16 tile[16 * 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) tile[i] = in[i];
    __syncthreads();
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16* addr = &tile[(4 * g + q) * 16 + 4 * p];
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(addr));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
}  // namespace

extern "C" int vqa_probe_tr16(void* stream, const void* in, float* out) {
    hipLaunchKernelGGL(probe_tr16, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(in), out);
    CHECK_LAUNCH();
    return VQA_OK;
}
