// Does the order in which a consumer walks a freshly written tensor matter (256 MB Infinity Cache)?
// writer: X[i] = f(i) front to back; reader: sums X front to back or back to front.  usage: mall_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void writer(uint4* x, long n, int nchunks) {
    // contiguous chunk per workgroup, chunks in grid order (as the BatchNorm passes walk)
    const long per = (n + nchunks - 1) / nchunks;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long lo = c * per, hi = lo + per < n ? lo + per : n;
        for (long i = lo + threadIdx.x; i < hi; i += 256) x[i] = make_uint4((unsigned)i, 1, 2, 3);
    }
}
__global__ __launch_bounds__(256) void reader(const uint4* x, long n, int nchunks, int reverse, unsigned* out) {
    const long per = (n + nchunks - 1) / nchunks;
    unsigned acc = 0;
    for (int c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
        const int c = reverse ? nchunks - 1 - c0 : c0;
        const long lo = c * per, hi = lo + per < n ? lo + per : n;
        for (long i = lo + threadIdx.x; i < hi; i += 256) { const uint4 v = x[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const long sizes_mb[] = {103, 205, 411, 822};
    unsigned* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long mb : sizes_mb) {
        const long n = mb * 1000000 / 16;
        uint4* x; CK(hipMalloc(&x, n * 16));
        const int nchunks = 8192, grid = 2048;
        for (int reverse = 0; reverse < 2; ++reverse) {
            float best = 1e9f, sum = 0.f;
            for (int it = 0; it < 6; ++it) {
                hipLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, x, n, nchunks);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(reader, dim3(grid), dim3(256), 0, 0, x, n, nchunks, reverse, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0) { sum += ms; if (ms < best) best = ms; }
            }
            printf("%4ld MB  reader %s: best %.3f ms (%.2f TB/s), mean %.3f ms\n", mb, reverse ? "back-to-front" : "front-to-back",
                   best, mb * 1e6 / best / 1e9, sum / 5);
        }
        CK(hipFree(x));
    }
    return 0;
}
