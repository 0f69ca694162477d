// fill_bench.hip -- how fast can 2 x 268 MB be filled on gfx950?  (cam_bp's only full-volume pass)
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/fill_bench.hip -o /tmp/fb && /tmp/fb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void fill2(v4f *a, v4f *b, float va, float vb, long n4)
{
    const v4f fa = {va, va, va, va}, fb = {vb, vb, vb, vb};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long j = i + u * stride;
            if (j < n4) {
                if (NT) { __builtin_nontemporal_store(fa, &a[j]); __builtin_nontemporal_store(fb, &b[j]); }
                else { a[j] = fa; b[j] = fb; }
            }
        }
    }
}
template <int U, bool NT>
void run(const char *name, v4f *a, v4f *b, long n4, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) fill2<U, NT><<<blocks, 256>>>(a, b, 1.f, 0.f, n4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) fill2<U, NT><<<blocks, 256>>>(a, b, 1.f, 0.f, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %6d : %7.1f us  %6.2f TB/s\n", name, blocks, ms * 1000 / 20, 2.0 * n4 * 16 / (ms / 20 * 1e-3) / 1e12);
}
int main()
{
    const long n4 = 32l * 128 * 128 * 128 / 4;
    v4f *a, *b; hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run<1, true>("nt, 1 pair/iter", a, b, n4, blocks);
        run<4, true>("nt, 4 pairs/iter", a, b, n4, blocks);
        run<1, false>("plain, 1 pair/iter", a, b, n4, blocks);
        run<4, false>("plain, 4 pairs/iter", a, b, n4, blocks);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) { hipMemsetD32Async((hipDeviceptr_t)a, 0x3f800000, n4 * 4, 0); hipMemsetD32Async((hipDeviceptr_t)b, 0, n4 * 4, 0); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetD32Async x2                       : %7.1f us  %6.2f TB/s\n", ms * 1000 / 20, 2.0 * n4 * 16 / (ms / 20 * 1e-3) / 1e12);
    return 0;
}
