// valu_rate.hip — how many wave64 VALU instructions per cycle one gfx950 SIMD sustains, as a function of
// waves per SIMD and of ILP (independent chains per wave). hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int CHAINS, bool DIV>
__global__ void k(float* out, int iters, float seed) {
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + threadIdx.x;
    const float m = 1.0000001f, b = 1e-7f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (DIV) a[c] = 1.0f / a[c] + b;   // IEEE division sequence
            else a[c] = __builtin_fmaf(a[c], m, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool DIV>
void run(int wavesPerSimd, int iters) {
    const int blocks = 256 * 4 * wavesPerSimd;  // 64-thread blocks: one wave each
    float* d;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, DIV>), dim3(blocks), dim3(64), 0, 0, d, 16, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, DIV>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * iters * CHAINS;  // wave-level "a = f(a)" steps
    printf("%s chains=%d waves/SIMD=%d : %.3f ms  -> %.3f steps/ns/SIMD  (%.2f cycles per step at 2.4 GHz)\n", DIV ? "div" : "fma",
           CHAINS, wavesPerSimd, ms, ops / (ms * 1e6) / 1024.0, 2.4 / (ops / (ms * 1e6) / 1024.0));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 5, 8}) { run<1, false>(w, 20000); run<4, false>(w, 20000); run<8, false>(w, 20000); }
    for (int w : {1, 2, 5, 8}) { run<1, true>(w, 4000); run<4, true>(w, 4000); }
    return 0;
}
