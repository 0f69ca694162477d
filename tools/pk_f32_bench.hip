// pk_f32_bench.hip -- issue rate of packed-fp32 VALU ops on gfx950, with VGPR and with SGPR (broadcast) operands.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/pk_f32_bench.hip -o /tmp/pkb && /tmp/pkb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s0, float s1)
{
    f2 a = {(float)threadIdx.x, 1.f}, b = {2.f, 3.f}, c = {4.f, 5.f}, d = {6.f, 7.f};
    f2 e = {8.f, 1.f}, f = {2.5f, 3.f}, g = {4.5f, 5.f}, h = {6.5f, 7.f};
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // v_add_f32, VGPR operands, 8 independent chains
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s0));)
        } else if (MODE == 1) {   // v_pk_add_f32, VGPR operands
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(a));)
        } else if (MODE == 2) {   // v_pk_add_f32, SGPR pair operand broadcast with op_sel_hi
            f2 sp = {s0, s1};
            REP16(asm volatile("v_pk_add_f32 %0, %8, %0 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %8, %1 op_sel_hi:[0,1]\n"
                               "v_pk_add_f32 %2, %8, %2 op_sel_hi:[0,1]\n v_pk_add_f32 %3, %8, %3 op_sel_hi:[0,1]\n"
                               "v_pk_add_f32 %4, %8, %4 op_sel_hi:[0,1]\n v_pk_add_f32 %5, %8, %5 op_sel_hi:[0,1]\n"
                               "v_pk_add_f32 %6, %8, %6 op_sel_hi:[0,1]\n v_pk_add_f32 %7, %8, %7 op_sel_hi:[0,1]\n"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(sp));)
        } else if (MODE == 3) {   // v_pk_mul_f32 VGPR
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(b));)
        } else if (MODE == 4) {   // v_pk_fma_f32 VGPR
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                               "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(b));)
        } else if (MODE == 5) {   // v_fma_f32 VGPR
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                               "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s0));)
        } else if (MODE == 6) {   // v_add_f32 with SGPR operand
            REP16(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                               "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(s0));)
        } else if (MODE == 7) {   // v_min3_f32
            REP16(asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                               "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(s0));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + b.x + c.x + d.x + e.y + f.y + g.y + h.y + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
void run(const char *name, float *out, int waves_per_simd)
{
    const int iters = 2000, blocks = 256 * waves_per_simd;       // 256-thread blocks = 1 wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 128 * waves_per_simd;       // wave-instructions issued per SIMD
    printf("%-40s waves/SIMD %d : %6.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", name, waves_per_simd,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("v_add_f32 vgpr", out, w);
        run<6>("v_add_f32 sgpr", out, w);
        run<5>("v_fma_f32 vgpr", out, w);
        run<1>("v_pk_add_f32 vgpr", out, w);
        run<2>("v_pk_add_f32 sgpr-pair op_sel_hi", out, w);
        run<3>("v_pk_mul_f32 vgpr", out, w);
        run<4>("v_pk_fma_f32 vgpr", out, w);
        run<7>("v_min3_f32 vgpr", out, w);
    }
    return 0;
}
