// Micro-benchmark: LDS atomic throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs ds_add_u64),
// conflict-free vs random addresses, plus plain ds_write as a reference.  One 256-thread block per CU x8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    __shared__ unsigned long long tile64[2048];
    float *tf = (float *)tile64; unsigned *tu = (unsigned *)tile64;
    for (int i = threadIdx.x; i < 2048; i += 256) tile64[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned idx = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (RANDOM) { s = s * 1664525u + 1013904223u; idx = (s >> 12) & 4095u; }
        else idx = (idx + 256) & 4095u;
        if (MODE == 0) unsafeAtomicAdd(&tf[idx], 1.0f);
        else if (MODE == 1) atomicAdd(&tu[idx], 1u);
        else if (MODE == 2) atomicAdd(&tile64[idx & 2047u], 1ull);
        else tf[idx] = (float)i;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tf[5];
}
template <int MODE, bool RANDOM> void run(const char *name)
{
    float *out; hipMalloc(&out, 4096 * 4);
    const int iters = 4096, blocks = 2048;
    k<MODE, RANDOM><<<blocks, 256>>>(iters, out);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<MODE, RANDOM><<<blocks, 256>>>(iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters;
    printf("%-28s %8.3f ms  %8.2f G lane-ops/s  (%.2f lane-ops/clk/CU @2.4GHz)\n", name, ms, ops / ms / 1e6,
           ops / ms / 1e6 / 256 / 2.4);
    hipFree(out);
}
int main()
{
    run<0, false>("ds_add_f32 conflict-free"); run<0, true>("ds_add_f32 random");
    run<1, false>("ds_add_u32 conflict-free"); run<1, true>("ds_add_u32 random");
    run<2, false>("ds_add_u64 conflict-free"); run<2, true>("ds_add_u64 random");
    run<3, false>("ds_write_b32 conflict-free"); run<3, true>("ds_write_b32 random");
    return 0;
}
