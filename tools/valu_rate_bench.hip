// valu_rate_bench.hip -- issue cost (cycles per wave64 instruction per SIMD, 8 waves/SIMD resident) of the VALU
// instruction forms the hot kernels use, on gfx950.  Each form runs as 8 independent chains x 16 x iters.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/valu_rate_bench.hip -o /tmp/vrb && /tmp/vrb
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define CHAIN8(OP, TAIL)                                                                                              \
    REP16(asm volatile(OP " %0, " TAIL "\n" OP " %1, " TAIL "\n" OP " %2, " TAIL "\n" OP " %3, " TAIL "\n"               \
                       OP " %4, " TAIL "\n" OP " %5, " TAIL "\n" OP " %6, " TAIL "\n" OP " %7, " TAIL "\n"               \
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)              \
                       : "v"(va), "s"(sa), "v"(vb));)
#define CHAIN8D(OP, TAIL)                                                                                             \
    REP16(asm volatile(OP " %0, " TAIL "\n" OP " %1, " TAIL "\n" OP " %2, " TAIL "\n" OP " %3, " TAIL "\n"               \
                       OP " %4, " TAIL "\n" OP " %5, " TAIL "\n" OP " %6, " TAIL "\n" OP " %7, " TAIL "\n"               \
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)              \
                       : "v"(da), "s"(ds), "v"(va));)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sa, double ds)
{
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, va = 1.0001f, vb = 0.5f;
    double d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7, da = 1.0001;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { CHAIN8("v_mul_f32", "%0, %8") }                 // vgpr, vgpr   (note %0 reused as both)
        if (MODE == 1) { CHAIN8("v_mul_f32", "%9, %0") }                 // sgpr, vgpr
        if (MODE == 2) { CHAIN8("v_mul_f32", "0.5, %0") }                // inline constant
        if (MODE == 3) { CHAIN8("v_mul_f32", "0x3f000011, %0") }         // 32-bit literal
        if (MODE == 4) { CHAIN8("v_add_u32", "%0, %8") }
        if (MODE == 5) { CHAIN8("v_add_u32", "%9, %0") }
        if (MODE == 6) { CHAIN8("v_add_u32", "3, %0") }
        if (MODE == 7) { CHAIN8("v_lshlrev_b32", "2, %0") }
        if (MODE == 8) { CHAIN8("v_mul_lo_u32", "%0, %8") }
        if (MODE == 9) { CHAIN8("v_mul_u32_u24", "%0, %8") }
        if (MODE == 10) { CHAIN8("v_mad_u32_u24", "%0, %8, %10") }
        if (MODE == 11) { CHAIN8("v_fma_f32", "%0, %8, %10") }
        if (MODE == 12) { CHAIN8("v_fma_f32", "%0, %9, %10") }           // one sgpr operand
        if (MODE == 13) { CHAIN8("v_floor_f32", "%0") }
        if (MODE == 14) { CHAIN8("v_cvt_i32_f32", "%0") }
        if (MODE == 15) { CHAIN8("v_cvt_f32_i32", "%0") }
        if (MODE == 16) { CHAIN8("v_rcp_f32", "%0") }
        if (MODE == 17) { CHAIN8("v_sqrt_f32", "%0") }
        if (MODE == 18) { CHAIN8D("v_mul_f64", "%0, %8") }
        if (MODE == 19) { CHAIN8D("v_add_f64", "%0, %8") }
        if (MODE == 20) { CHAIN8D("v_fma_f64", "%0, %8, %8") }
        if (MODE == 21) { CHAIN8D("v_mul_f64", "%0, %9") }               // sgpr pair operand
        if (MODE == 22) {                                                 // v_cvt_f64_f32 / v_cvt_f32_f64 round trip (2 instr)
            REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %2, %0\n v_cvt_f64_f32 %1, %3\n v_cvt_f32_f64 %3, %1\n"
                               "v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %2, %0\n v_cvt_f64_f32 %1, %3\n v_cvt_f32_f64 %3, %1\n"
                               : "+v"(d0), "+v"(d1), "+v"(x0), "+v"(x1));)
        }
        if (MODE == 23) { CHAIN8("v_min_f32", "%0, %8") }
        if (MODE == 24) { CHAIN8("v_max3_f32", "%0, %8, %10") }
        if (MODE == 25) {                                                 // v_cmp + v_cndmask pair (2 instr)
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                               "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va) : "vcc");)
        }
        if (MODE == 26) { CHAIN8("v_sub_f32", "%0, %8") }
        if (MODE == 27) { CHAIN8("v_mov_b32", "%9") }                    // broadcast an sgpr
        if (MODE == 28) { CHAIN8("v_and_b32", "%0, %8") }
        if (MODE == 29) { CHAIN8("v_add3_u32", "%0, %8, %10") }
        if (MODE == 30) { CHAIN8("v_lshl_add_u32", "%0, 2, %8") }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int MODE>
void run(const char *name, float *out, int per_iter)
{
    const int iters = 1000, w = 8, blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.5f, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.5f, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * per_iter * w;
    printf("%-34s %6.2f cycles @2.4GHz per wave-instruction per SIMD\n", name, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_mul_f32 v,v", out, 128);   run<1>("v_mul_f32 s,v", out, 128);   run<2>("v_mul_f32 inline,v", out, 128);
    run<3>("v_mul_f32 literal,v", out, 128); run<26>("v_sub_f32 v,v", out, 128); run<23>("v_min_f32 v,v", out, 128);
    run<11>("v_fma_f32 v,v,v", out, 128); run<12>("v_fma_f32 v,s,v", out, 128); run<24>("v_max3_f32 v,v,v", out, 128);
    run<4>("v_add_u32 v,v", out, 128);   run<5>("v_add_u32 s,v", out, 128);   run<6>("v_add_u32 inline,v", out, 128);
    run<7>("v_lshlrev_b32 imm,v", out, 128); run<28>("v_and_b32 v,v", out, 128); run<29>("v_add3_u32 v,v,v", out, 128);
    run<30>("v_lshl_add_u32 v,imm,v", out, 128);
    run<8>("v_mul_lo_u32 v,v", out, 128); run<9>("v_mul_u32_u24 v,v", out, 128); run<10>("v_mad_u32_u24 v,v,v", out, 128);
    run<13>("v_floor_f32", out, 128);     run<14>("v_cvt_i32_f32", out, 128);   run<15>("v_cvt_f32_i32", out, 128);
    run<16>("v_rcp_f32", out, 128);       run<17>("v_sqrt_f32", out, 128);      run<27>("v_mov_b32 v,s", out, 128);
    run<18>("v_mul_f64 v,v", out, 128);   run<19>("v_add_f64 v,v", out, 128);   run<20>("v_fma_f64 v,v,v", out, 128);
    run<21>("v_mul_f64 v,s", out, 128);   run<22>("v_cvt_f64_f32 + v_cvt_f32_f64", out, 128);
    run<25>("v_cmp_lt_f32 + v_cndmask (vcc)", out, 128);
    return 0;
}
