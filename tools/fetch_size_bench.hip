// fetch_size_bench.hip -- what does the FETCH_SIZE counter report on gfx950 for reads of KNOWN size?
// Four kernels read the same 256 MiB buffer exactly once (every 128-byte line once), differing only in the access shape:
//   probe_stream16   16 B / lane, consecutive (a wave reads 1 KiB contiguous)            -- calc_prob, fills, tile staging
//   probe_stream4     4 B / lane, consecutive (a wave reads 256 B contiguous)
//   probe_lines4      4 B / lane, a half-wave reads one 128-byte line, lines visited in a scattered order
//                     (the batch-minor renderer's pattern: 32 images of one voxel / sample / segment)
//   probe_lines16    16 B / lane, 8 lanes read one 128-byte line, lines scattered
// Run under `rocprofv3 --pmc FETCH_SIZE` (units: KB); expected 262144 per dispatch if the counter is exact.
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_size_bench.hip -o tools/fetch_size_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t kBytes = 256ull << 20;
constexpr uint32_t kLines = kBytes / 128;            // 2 M lines of 128 bytes
constexpr uint32_t kMul = 1000003u;                   // odd: i -> i * kMul mod 2^21 is a permutation of the lines

__global__ void probe_stream16(const float4 *__restrict__ src, float *__restrict__ sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = src[i];
    if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = 1.f;
}
__global__ void probe_stream4(const float *__restrict__ src, float *__restrict__ sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (src[i] == 12345.f) sink[0] = 1.f;
}
__global__ void probe_lines4(const float *__restrict__ src, float *__restrict__ sink)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;         // 32 lanes per line
    const uint32_t line = ((t >> 5) * kMul) & (kLines - 1);
    if (src[(size_t)line * 32 + (t & 31)] == 12345.f) sink[0] = 1.f;
}
__global__ void probe_lines16(const float4 *__restrict__ src, float *__restrict__ sink)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;         // 8 lanes per line
    const uint32_t line = ((t >> 3) * kMul) & (kLines - 1);
    const float4 v = src[(size_t)line * 8 + (t & 7)];
    if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = 1.f;
}

int main()
{
    float *buf, *sink;
    (void)hipMalloc(&buf, kBytes);
    (void)hipMalloc(&sink, 256);
    (void)hipMemset(buf, 0, kBytes);
    // a second buffer written in between pushes the first out of L2 / MALL (512 MiB > 256 MiB Infinity Cache)
    float *evict;
    (void)hipMalloc(&evict, 2 * kBytes);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipMemset(evict, rep, 2 * kBytes);
        hipLaunchKernelGGL(probe_stream16, dim3(kBytes / 16 / 256), dim3(256), 0, 0, (const float4 *)buf, sink);
        (void)hipMemset(evict, rep, 2 * kBytes);
        hipLaunchKernelGGL(probe_stream4, dim3(kBytes / 4 / 256), dim3(256), 0, 0, (const float *)buf, sink);
        (void)hipMemset(evict, rep, 2 * kBytes);
        hipLaunchKernelGGL(probe_lines4, dim3(kBytes / 4 / 256), dim3(256), 0, 0, (const float *)buf, sink);
        (void)hipMemset(evict, rep, 2 * kBytes);
        hipLaunchKernelGGL(probe_lines16, dim3(kBytes / 16 / 256), dim3(256), 0, 0, (const float4 *)buf, sink);
    }
    (void)hipDeviceSynchronize();
    printf("4 probes x 3 repetitions, %zu bytes each: expect FETCH_SIZE = %zu KB per dispatch\n", kBytes, kBytes / 1024);
    return 0;
}
