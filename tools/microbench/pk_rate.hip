// pk_rate.hip — issue rate of scalar vs packed FP32 VALU instructions on one gfx950 SIMD (wave64), forced with inline
// asm so the compiler cannot choose for us. hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
// Answers: does v_pk_fma_f32 (2 FMAs per lane) issue as fast as v_fma_f32 (1 FMA per lane)?
#include <hip/hip_runtime.h>
#include <stdio.h>

using f2 = float __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
    float a[8];
    float x5[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    f2 p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = seed + c + threadIdx.x;
        p[c] = f2{a[c], a[c] + 0.5f};
    }
    float m = 1.0000001f, b = 1e-7f;
    const f2 pm = f2{m, m}, pb = f2{b, b};
    unsigned long long sel = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pm), "v"(pb));
            if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pm));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pb));
            if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b) : );
            if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[c]) : "v"(b));
            if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));
            if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc");
            if (KIND == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "s"(sel));
            if (KIND == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[c]) : "v"(b), "v"(m) : "vcc");
            if (KIND == 12) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
            if (KIND == 13) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (KIND == 14) asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (KIND == 15) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[c]) : "v"(b), "v"(m));
            if (KIND == 16) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(m), "v"(b));
            if (KIND == 17) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sel) : "v"(a[c]), "v"(b));
            if (KIND == 18) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
            if (KIND == 19) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[c]));
            if (KIND == 20) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[c]));
            if (KIND == 21) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[c]));
            if (KIND == 22) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (KIND == 24) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[c]), "=&s"(sel) : "v"(b), "v"(m));
            if (KIND == 25) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b));
            if (KIND == 26) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_mul_f32_e32 %0, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[c]) : "v"(b), "v"(m) : "vcc");
            if (KIND == 27) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n v_mul_f32_e32 %0, %0, %3\n v_cndmask_b32_e64 %0, %0, %3, %1" : "+v"(a[c]), "=&s"(sel) : "v"(b), "v"(m));
            if (KIND == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %4, %4, %2, vcc\n v_cndmask_b32 %5, %5, %2, vcc" : "+v"(a[c]), "+v"(b), "+v"(m), "+v"(p[c].x), "+v"(p[c].y), "+v"(x5[c]) : : "vcc");
            if (KIND == 29) asm volatile("v_cmp_lt_f32_e64 %6, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, %6\n v_cndmask_b32_e64 %3, %3, %2, %6\n v_cndmask_b32_e64 %4, %4, %2, %6\n v_cndmask_b32_e64 %5, %5, %2, %6" : "+v"(a[c]), "+v"(b), "+v"(m), "+v"(p[c].x), "+v"(p[c].y), "+v"(x5[c]), "=&s"(sel));
            if (KIND == 23) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += a[c] + p[c].x + p[c].y;
    s += (float)(sel & 3);
    for (int c = 0; c < 8; ++c) s += x5[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int wavesPerSimd, int iters) {
    const int blocks = 256 * 4 * wavesPerSimd;  // 64-thread blocks: one wave each
    float* d;
    hipMalloc(&d, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64), 0, 0, d, 16, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ins = (double)blocks * iters * 8;  // wave-level instructions
    const double perNsPerSimd = ins / (ms * 1e6) / 1024.0;
    printf("%-14s waves/SIMD=%d : %.3f ms -> %.2f SIMD cycles per wave64 instruction (2.4 GHz)\n", name, wavesPerSimd, ms, 2.4 / perNsPerSimd);
    hipFree(d);
}

int main() {
    for (int w : {3, 5}) {
        run<10>("cndmask_e64_sgpr", w, 20000);
        run<11>("cmp+cndmask", w, 20000);
        run<12>("v_fmac_f32_e32", w, 20000);
        run<13>("v_max_f32_e32", w, 20000);
        run<14>("v_or_b32_e32", w, 20000);
        run<15>("cndmask_nodep", w, 20000);
        run<16>("v_fma acc", w, 20000);
        run<17>("v_cmp_e64 sgpr", w, 20000);
        run<18>("v_mul_f32_e64", w, 20000);
        run<19>("v_lshlrev_b32", w, 20000);
        run<20>("v_sqrt_f32", w, 20000);
        run<21>("v_rsq_f32", w, 20000);
        run<22>("v_add_u32", w, 20000);
        run<23>("v_xor_b32", w, 20000);
        run<24>("cmp_e64+cnd_e64 (x2)", w, 20000);
        run<25>("cndmask_e64 vcc", w, 20000);
        run<26>("cmp,mul,cnd vcc (x3)", w, 20000);
        run<27>("cmp,mul,cnd sgpr (x3)", w, 20000);
        run<28>("cmp + 4 cnd vcc (x5)", w, 20000);
        run<29>("cmp + 4 cnd sgpr (x5)", w, 20000);
    }
    for (int w : {1, 2, 5}) {
        run<0>("v_fma_f32", w, 20000);
        run<1>("v_pk_fma_f32", w, 20000);
        run<2>("v_mul_f32", w, 20000);
        run<3>("v_pk_mul_f32", w, 20000);
        run<4>("v_add_f32", w, 20000);
        run<5>("v_pk_add_f32", w, 20000);
        run<6>("v_cndmask_b32", w, 20000);
        run<7>("v_mov_b32", w, 20000);
        run<8>("v_rcp_f32", w, 20000);
        run<9>("v_cmp_lt_f32", w, 20000);
    }
    return 0;
}
