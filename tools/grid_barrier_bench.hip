// grid_barrier_bench.hip -- what does a device-wide barrier cost inside one launch on gfx950?
// variants: 0 = no fences (waitcnt only), 1 = release(wbl2) by one wave per block, 2 = acquire(inv) by one wave,
// 3 = both, 4 = both by every wave.  Each launch: [optional 16 MiB fill] + NB barriers.
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ unsigned g_ctr[64];
template <int V>
__device__ __forceinline__ void gbar(unsigned *ctr, unsigned target)
{
    if (V == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (V == 1 || V == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (V == 2 || V == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (V == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <int V, int STORE>
__global__ __launch_bounds__(1024) void k(float *a, long n4, int nb, int slot)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
    const v4f z = {1.f, 1.f, 1.f, 1.f};
    if (STORE == 1) for (long i = tid; i < n4; i += nt) __builtin_nontemporal_store(z, &((v4f *)a)[i]);
    if (STORE == 2) for (long i = tid; i < n4; i += nt) ((v4f *)a)[i] = z;
    if (STORE == 3) for (long i = tid; i < n4 * 2; i += nt)
        __hip_atomic_store(&((unsigned long long *)a)[i], 0x3f8000003f800000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned *ctr = g_ctr + slot;
    for (int b = 0; b < nb; b++) gbar<V>(ctr, (unsigned)(b + 1) * gridDim.x);
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nb + 1) * gridDim.x - 1u)
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int V, int STORE>
void run(const char *name, float *a, long n4, int nb, int blocks, int threads)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) k<V, STORE><<<blocks, threads>>>(a, n4, nb, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 50; i++) k<V, STORE><<<blocks, threads>>>(a, n4, nb, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks %4d x %4d  barriers %d : %7.2f us/launch  (%s)\n", name, blocks, threads, nb, ms * 1000 / 50, hipGetErrorString(hipGetLastError()));
}
int main()
{
    float *a; const long n4 = 2l * 128 * 128 * 128 / 4;
    hipMalloc(&a, n4 * 16);
    for (int nb = 0; nb <= 2; nb += 2) {
        run<0, 0>("no store, no fences", a, n4, nb, 256, 1024);
        run<1, 0>("no store, release(1 wave)", a, n4, nb, 256, 1024);
        run<2, 0>("no store, acquire(1 wave)", a, n4, nb, 256, 1024);
        run<3, 0>("no store, both(1 wave)", a, n4, nb, 256, 1024);
        run<0, 1>("nt fill 16MiB, no fences", a, n4, nb, 256, 1024);
        run<3, 1>("nt fill 16MiB, both(1 wave)", a, n4, nb, 256, 1024);
        run<3, 2>("plain fill 16MiB, both(1 wave)", a, n4, nb, 256, 1024);
        run<0, 3>("sc1 8B fill 16MiB, no fences", a, n4, nb, 256, 1024);
        run<3, 1>("nt fill 16MiB, both(1 wave) 256thr", a, n4, nb, 256, 256);
        run<3, 1>("nt fill 16MiB, both(1 wave) 64 blocks", a, n4, nb, 64, 1024);
        run<0, 1>("nt fill 16MiB, no fences 2048x256", a, n4, nb == 0 ? 0 : 0, 2048, 256);
        run<4, 1>("nt fill 16MiB, both(all waves)", a, n4, nb, 256, 1024);
    }
    return 0;
}
