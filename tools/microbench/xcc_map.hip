// xcc_map.hip — which XCD runs workgroup b? (shard s = b % 16 is meant to stay with XCD s % 8.)
// hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id & 0xf;
}

int main() {
    const int blocks = 16 * 1280;  // the bounce kernel's capped grid
    unsigned* d;
    hipMalloc(&d, blocks * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d);
    std::vector<unsigned> h(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    int match = 0, hist[16][8] = {{0}};
    for (int b = 0; b < blocks; ++b) {
        match += (h[b] == (unsigned)(b % 8));
        if (h[b] < 8) hist[b % 16][h[b]]++;
    }
    printf("workgroups whose XCC id == blockIdx %% 8: %d of %d\n", match, blocks);
    for (int s = 0; s < 16; ++s) {
        printf("shard %2d:", s);
        for (int x = 0; x < 8; ++x) printf(" %5d", hist[s][x]);
        printf("\n");
    }
    return 0;
}
