// vgpr_banks.hip — does a wave64 VALU instruction on gfx950 cost more when its VGPR operands share a register bank
// (register index mod 4), and does a third VGPR source cost an extra cycle by itself? Hard-coded registers in one asm
// block, 28 waves per CU (7 per SIMD, 256-thread workgroups) like the bounce kernel.
//   hipcc --offload-arch=gfx950 -O3 vgpr_banks.hip -o vgpr_banks
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256, 7) void k(float* out, int iters) {
    float r = 0;
    // v32..v35 sources (banks 0..3), v40.. accumulators
    asm volatile(
        "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 2.0\n v_mov_b32 v35, 4.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 0.5\n"
        "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
        "v_mov_b32 v48, 0\n v_mov_b32 v52, 0\n v_mov_b32 v56, 0\n v_mov_b32 v60, 0\n"
        "s_mov_b32 s20, %1\n s_mov_b32 s21, 0x3f800000\n s_mov_b64 s[22:23], exec\n"
        "1:\n"
        ".if %2 == 0\n"  // fmac, three distinct banks: dst bank 0 (v40,v44,v48,v52), srcs banks 1,2
        REP8("v_fmac_f32_e32 v40, v33, v34\n v_fmac_f32_e32 v44, v33, v34\n v_fmac_f32_e32 v48, v33, v34\n v_fmac_f32_e32 v52, v33, v34\n")
        ".endif\n"
        ".if %2 == 1\n"  // fmac, dst shares the bank of src0 (v41,v45.. bank 1 = v33's)
        REP8("v_fmac_f32_e32 v41, v33, v34\n v_fmac_f32_e32 v45, v33, v34\n v_fmac_f32_e32 v41, v33, v34\n v_fmac_f32_e32 v45, v33, v34\n")
        ".endif\n"
        ".if %2 == 2\n"  // fmac, src0 and src1 in one bank (v33, v37), dst elsewhere
        REP8("v_fmac_f32_e32 v40, v33, v37\n v_fmac_f32_e32 v44, v33, v37\n v_fmac_f32_e32 v48, v33, v37\n v_fmac_f32_e32 v52, v33, v37\n")
        ".endif\n"
        ".if %2 == 3\n"  // fmac, all three in bank 1
        REP8("v_fmac_f32_e32 v41, v33, v37\n v_fmac_f32_e32 v45, v33, v37\n v_fmac_f32_e32 v41, v33, v37\n v_fmac_f32_e32 v45, v33, v37\n")
        ".endif\n"
        ".if %2 == 4\n"  // mul, two sources in distinct banks, dst third bank
        REP8("v_mul_f32_e32 v40, v33, v34\n v_mul_f32_e32 v44, v33, v34\n v_mul_f32_e32 v48, v33, v34\n v_mul_f32_e32 v52, v33, v34\n")
        ".endif\n"
        ".if %2 == 5\n"  // mul, two sources in one bank
        REP8("v_mul_f32_e32 v40, v33, v37\n v_mul_f32_e32 v44, v33, v37\n v_mul_f32_e32 v48, v33, v37\n v_mul_f32_e32 v52, v33, v37\n")
        ".endif\n"
        ".if %2 == 6\n"  // fma VOP3 with an inline constant as third source (two VGPR reads)
        REP8("v_fma_f32 v40, v33, v34, 1.0\n v_fma_f32 v44, v33, v34, 1.0\n v_fma_f32 v48, v33, v34, 1.0\n v_fma_f32 v52, v33, v34, 1.0\n")
        ".endif\n"
        ".if %2 == 7\n"  // fmac with an SGPR source (two VGPR reads: dst + one)
        REP8("v_fmac_f32_e32 v40, s21, v34\n v_fmac_f32_e32 v44, s21, v34\n v_fmac_f32_e32 v48, s21, v34\n v_fmac_f32_e32 v52, s21, v34\n")
        ".endif\n"
        ".if %2 == 8\n"  // mul with one operand repeated (one distinct VGPR)
        REP8("v_mul_f32_e32 v40, v33, v33\n v_mul_f32_e32 v44, v33, v33\n v_mul_f32_e32 v48, v33, v33\n v_mul_f32_e32 v52, v33, v33\n")
        ".endif\n"
        ".if %2 == 9\n"  // dependent fmac chain on ONE accumulator (latency exposed per wave, hidden across 7 waves?)
        REP8("v_fmac_f32_e32 v40, v33, v34\n v_fmac_f32_e32 v40, v33, v34\n v_fmac_f32_e32 v40, v33, v34\n v_fmac_f32_e32 v40, v33, v34\n")
        ".endif\n"
        ".if %2 == 10\n"  // mul by a 32-bit literal
        REP8("v_mul_f32_e32 v40, 0x38d1b717, v34\n v_mul_f32_e32 v44, 0x38d1b717, v34\n v_mul_f32_e32 v48, 0x38d1b717, v34\n v_mul_f32_e32 v52, 0x38d1b717, v34\n")
        ".endif\n"
        ".if %2 == 11\n"  // mul by an SGPR
        REP8("v_mul_f32_e32 v40, s21, v34\n v_mul_f32_e32 v44, s21, v34\n v_mul_f32_e32 v48, s21, v34\n v_mul_f32_e32 v52, s21, v34\n")
        ".endif\n"
        ".if %2 == 12\n"  // mul by an inline constant
        REP8("v_mul_f32_e32 v40, 4.0, v34\n v_mul_f32_e32 v44, 4.0, v34\n v_mul_f32_e32 v48, 4.0, v34\n v_mul_f32_e32 v52, 4.0, v34\n")
        ".endif\n"
        ".if %2 == 13\n"  // compare into an SGPR pair
        REP8("v_cmp_lt_f32_e64 s[22:23], v33, v34\n v_cmp_lt_f32_e64 s[24:25], v33, v34\n v_cmp_lt_f32_e64 s[22:23], v33, v34\n v_cmp_lt_f32_e64 s[24:25], v33, v34\n")
        ".endif\n"
        ".if %2 == 14\n"  // compare into VCC
        REP8("v_cmp_lt_f32_e32 vcc, v33, v34\n v_cmp_lt_f32_e32 vcc, v33, v34\n v_cmp_lt_f32_e32 vcc, v33, v34\n v_cmp_lt_f32_e32 vcc, v33, v34\n")
        ".endif\n"
        ".if %2 == 15\n"  // select on an SGPR pair (VOP3)
        REP8("v_cndmask_b32_e64 v40, v33, v34, s[22:23]\n v_cndmask_b32_e64 v44, v33, v34, s[22:23]\n v_cndmask_b32_e64 v48, v33, v34, s[22:23]\n v_cndmask_b32_e64 v52, v33, v34, s[22:23]\n")
        ".endif\n"
        ".if %2 == 16\n"  // select on VCC (VOP2)
        REP8("v_cndmask_b32_e32 v40, v33, v34, vcc\n v_cndmask_b32_e32 v44, v33, v34, vcc\n v_cndmask_b32_e32 v48, v33, v34, vcc\n v_cndmask_b32_e32 v52, v33, v34, vcc\n")
        ".endif\n"
        ".if %2 == 17\n"  // v_mov from an SGPR
        REP8("v_mov_b32 v40, s21\n v_mov_b32 v44, s21\n v_mov_b32 v48, s21\n v_mov_b32 v52, s21\n")
        ".endif\n"
        ".if %2 == 18\n"  // compare with an SGPR operand into an SGPR pair
        REP8("v_cmp_lt_f32_e64 s[22:23], v33, s21\n v_cmp_lt_f32_e64 s[24:25], v33, s21\n v_cmp_lt_f32_e64 s[22:23], v33, s21\n v_cmp_lt_f32_e64 s[24:25], v33, s21\n")
        ".endif\n"
        ".if %2 == 19\n"  // rcp
        REP8("v_rcp_f32 v40, v33\n v_rcp_f32 v44, v33\n v_rcp_f32 v48, v33\n v_rcp_f32 v52, v33\n")
        ".endif\n"
        ".if %2 == 20\n"  // v_fma VOP3 with neg modifier, three banks
        REP8("v_fma_f32 v40, -v33, v34, v40\n v_fma_f32 v44, -v33, v34, v44\n v_fma_f32 v48, -v33, v34, v48\n v_fma_f32 v52, -v33, v34, v52\n")
        ".endif\n"
        ".if %2 == 21\n"  // v_add3_u32 / v_lshl_add_u32 (three sources)
        REP8("v_add3_u32 v40, v33, v34, v40\n v_lshl_add_u32 v44, v33, 2, v44\n v_add3_u32 v48, v33, v34, v48\n v_lshl_add_u32 v52, v33, 2, v52\n")
        ".endif\n"
        ".if %2 == 22\n"  // 64-bit shift-add (address arithmetic of the ray planes)
        REP8("v_lshl_add_u64 v[40:41], v[32:33], 2, v[40:41]\n v_lshl_add_u64 v[44:45], v[32:33], 2, v[44:45]\n v_lshl_add_u64 v[48:49], v[32:33], 2, v[48:49]\n v_lshl_add_u64 v[52:53], v[32:33], 2, v[52:53]\n")
        ".endif\n"
        ".if %2 == 23\n"  // 64-bit unsigned compare (the (distance, ~index) key of an order-free closest hit)
        REP8("v_cmp_lt_u64_e64 s[22:23], v[32:33], v[34:35]\n v_cmp_lt_u64_e64 s[24:25], v[32:33], v[34:35]\n v_cmp_lt_u64_e64 s[22:23], v[32:33], v[34:35]\n v_cmp_lt_u64_e64 s[24:25], v[32:33], v[34:35]\n")
        ".endif\n"
        ".if %2 == 24\n"  // v_min3_f32
        REP8("v_min3_f32 v40, v33, v34, v35\n v_min3_f32 v44, v33, v34, v35\n v_min3_f32 v48, v33, v34, v35\n v_min3_f32 v52, v33, v34, v35\n")
        ".endif\n"
        ".if %2 == 25\n"  // 32-bit unsigned compare VGPR, VGPR -> SGPR pair
        REP8("v_cmp_lt_u32_e64 s[22:23], v33, v34\n v_cmp_lt_u32_e64 s[24:25], v33, v34\n v_cmp_lt_u32_e64 s[22:23], v33, v34\n v_cmp_lt_u32_e64 s[24:25], v33, v34\n")
        ".endif\n"
        ".if %2 == 26\n"  // v_cmp_class (one instruction for "positive normal/subnormal/zero/inf ...")
        REP8("v_cmp_class_f32_e64 s[22:23], v33, v34\n v_cmp_class_f32_e64 s[24:25], v33, v34\n v_cmp_class_f32_e64 s[22:23], v33, v34\n v_cmp_class_f32_e64 s[24:25], v33, v34\n")
        ".endif\n"
        ".if %2 == 27\n"  // 32 scalar ALU instructions per trip (are they free beside the other waves' VALU? here: alone)
        REP8("s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n")
        ".endif\n"
        ".if %2 == 28\n"  // 16 plain VALU + 16 SALU interleaved: does the scalar half hide?
        REP8("v_mul_f32_e32 v40, v33, v34\n s_add_u32 s24, s24, 1\n v_mul_f32_e32 v44, v33, v34\n s_add_u32 s25, s25, 1\n")
        ".endif\n"
        ".if %2 == 29\n"  // 16 plain VALU + 16 taken scalar branches
        REP8("v_mul_f32_e32 v40, v33, v34\n s_branch 2f\n s_nop 0\n 2:\n v_mul_f32_e32 v44, v33, v34\n s_branch 3f\n s_nop 0\n 3:\n")
        ".endif\n"
        ".if %2 == 30\n"  // v_max_f32 (clamp-style helpers)
        REP8("v_max_f32_e32 v40, v33, v34\n v_max_f32_e32 v44, v33, v34\n v_max_f32_e32 v48, v33, v34\n v_max_f32_e32 v52, v33, v34\n")
        ".endif\n"
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
        "v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v44\n v_add_f32 %0, %0, v45\n v_add_f32 %0, %0, v48\n v_add_f32 %0, %0, v52\n"
        : "=v"(r)
        : "s"(iters), "n"(KIND)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v52", "v53", "v56", "v60", "s20", "s21", "s22", "s23", "s24", "s25", "scc", "vcc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name) {
    const int blocks = 256 * 7 * 2, iters = 2000;
    float* d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters * 32;
    printf("%-52s %7.3f ms  %5.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / instr);
    (void)hipFree(d);
}

int main() {
    run<0>("v_fmac dst/src0/src1 in three banks");
    run<1>("v_fmac dst shares src0's bank");
    run<2>("v_fmac src0, src1 share a bank");
    run<3>("v_fmac all three in one bank");
    run<4>("v_mul src0, src1 in two banks");
    run<5>("v_mul src0, src1 share a bank");
    run<6>("v_fma v, v, v, 1.0 (two VGPR sources)");
    run<7>("v_fmac v, s, v (SGPR source)");
    run<8>("v_mul v, v33, v33 (one distinct source)");
    run<9>("v_fmac dependent chain, one accumulator");
    run<10>("v_mul by a 32-bit literal");
    run<11>("v_mul by an SGPR");
    run<12>("v_mul by an inline constant");
    run<13>("v_cmp -> SGPR pair");
    run<14>("v_cmp -> VCC");
    run<15>("v_cndmask VOP3 on an SGPR pair");
    run<16>("v_cndmask VOP2 on VCC");
    run<17>("v_mov from an SGPR");
    run<18>("v_cmp with an SGPR operand -> SGPR pair");
    run<19>("v_rcp_f32");
    run<20>("v_fma VOP3 (neg), three banks");
    run<21>("v_add3_u32 / v_lshl_add_u32");
    run<22>("v_lshl_add_u64");
    run<23>("v_cmp_lt_u64 -> SGPR pair");
    run<24>("v_min3_f32");
    run<25>("v_cmp_lt_u32 v, v -> SGPR pair");
    run<26>("v_cmp_class_f32 -> SGPR pair");
    run<27>("32 x s_add_u32 (counted as 32 instructions)");
    run<28>("16 x v_mul + 16 x s_add interleaved (per instruction of 32)");
    run<29>("16 x v_mul + 16 taken s_branch (per instruction of 32; REP8 local labels)");
    run<30>("v_max_f32");
    run<0>("v_fmac three banks (again)");
    return 0;
}
