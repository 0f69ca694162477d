// LDS u64 atomic throughput vs number of distinct addresses per wave instruction (same-address contention).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DISTINCT, int MODE>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    __shared__ unsigned long long tile64[4096];
    unsigned *tu = (unsigned *)tile64;
    for (int i = threadIdx.x; i < 4096; i += 256) tile64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned base = (threadIdx.x >> 6) * 997u;
    for (int i = 0; i < iters; i++) {
        base = (base + 61u) & 2047u;
        const unsigned idx = (base + (lane % DISTINCT) * 33u) & 4095u;
        if (MODE == 0) atomicAdd(&tile64[idx], 1ull);
        else atomicAdd(&tu[idx], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)tile64[5];
}
template <int DISTINCT, int MODE> void run()
{
    float *out; (void)hipMalloc(&out, 4096 * 4);
    const int iters = 2048, blocks = 2048;
    k<DISTINCT, MODE><<<blocks, 256>>>(iters, out);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    k<DISTINCT, MODE><<<blocks, 256>>>(iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters;
    // 8 blocks/CU resident -> per-CU cycles per wave instruction
    printf("%s distinct=%2d : %8.3f ms  %7.1f cycles per wave-instruction per CU (@2.4GHz)\n", MODE ? "u32" : "u64",
           DISTINCT, ms, ms * 1e-3 * 2.4e9 / (winstr / 256));
    (void)hipFree(out);
}
int main()
{
    run<64, 0>(); run<32, 0>(); run<16, 0>(); run<8, 0>(); run<4, 0>(); run<2, 0>(); run<1, 0>();
    run<64, 1>(); run<8, 1>(); run<1, 1>();
    return 0;
}
