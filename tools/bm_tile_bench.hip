// Micro-benchmarks behind the batch-minor tile renderer (csrc/sph_render_bm.hip), gfx950:
//   A  LDS throughput in the [voxel line][32 images] pattern: a wave touches two adjacent lines (z0, z0+1) of a
//      wave-uniform pseudo-random voxel -- ds_add_f32 / ds_add_u64 / ds_add_f64 / plain read-add-write / 4 reads
//   B  global line-coherent float atomics: every workgroup flushes a 5x9x9-line tile (128 B per line) into a
//      268 MB batch-minor gradient volume, with hardware atomics vs plain stores; plus a contended variant in which
//      many workgroups flush the SAME tile
// build: hipcc --offload-arch=gfx950 -O3 -o bm_tile_bench tools/bm_tile_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kLines = 405;     // 5 x 9 x 9

template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(int iters, float *out)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tile64[kLines * 32];      // 103 680 B
    float *tf = (float *)tile64;
    double *td = (double *)tile64;
    for (int i = threadIdx.x; i < kLines * 32; i += 256) tile64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = (threadIdx.x >> 6) * 2654435761u + blockIdx.x * 40503u + 12345u;       // wave-uniform stream
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const int x = (s >> 8) & 3, y = (s >> 12) & 7, z = (s >> 16) & 7;                // base corner in a 4x8x8 brick
        const int line = (x * 9 + y) * 9 + z + (lane >> 5);                              // half-waves: z0 / z0+1
        const int e = line * 32 + (lane & 31);
        if (MODE == 0) {
            unsafeAtomicAdd(&tf[e], 1.0f); unsafeAtomicAdd(&tf[e + 81 * 32], 1.0f);
            unsafeAtomicAdd(&tf[e + 9 * 32], 1.0f); unsafeAtomicAdd(&tf[e + 90 * 32], 1.0f);
        } else if (MODE == 1) {
            atomicAdd(&tile64[e], 1ull); atomicAdd(&tile64[e + 81 * 32], 1ull);
            atomicAdd(&tile64[e + 9 * 32], 1ull); atomicAdd(&tile64[e + 90 * 32], 1ull);
        } else if (MODE == 2) {
            unsafeAtomicAdd(&td[e], 1.0); unsafeAtomicAdd(&td[e + 81 * 32], 1.0);
            unsafeAtomicAdd(&td[e + 9 * 32], 1.0); unsafeAtomicAdd(&td[e + 90 * 32], 1.0);
        } else if (MODE == 3) {
            tf[e] += 1.0f; tf[e + 81 * 32] += 1.0f; tf[e + 9 * 32] += 1.0f; tf[e + 90 * 32] += 1.0f;
        } else {
            acc += tf[e] + tf[e + 81 * 32] + tf[e + 9 * 32] + tf[e + 90 * 32];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tf[5] + acc;
}

template <int MODE> void run_lds(const char *name)
{
    float *out; hipMalloc(&out, 8192 * 4);
    const int iters = 2048, blocks = 1024;
    lds_kernel<MODE><<<blocks, 256>>>(iters, out);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    lds_kernel<MODE><<<blocks, 256>>>(iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_ops = (double)blocks * 4 * iters * 4;
    printf("A %-30s %8.3f ms  %7.2f G wave-ops/s  %6.2f clk per wave-op per CU (256 CUs @2.4 GHz, 1 WG/CU)\n", name, ms,
           wave_ops / ms / 1e6, 256 * 2.4e9 / (wave_ops / ms * 1e3));
    hipFree(out);
}

// B: flush tiles.  grid.x = bricks (order[]), each workgroup adds / stores its 405 lines.
template <int MODE>
__global__ __launch_bounds__(256) void flush_kernel(float *__restrict__ g, const int *__restrict__ order, int X, int Y, int Z)
{
    const int b = order[blockIdx.x];
    const int nby = Y / 8, nbz = Z / 8;
    const int ox = (b / (nby * nbz)) * 4, oy = ((b / nbz) % nby) * 8, oz = (b % nbz) * 8;
    const int lane = threadIdx.x & 31;
    for (int l = threadIdx.x >> 5; l < kLines; l += 8) {
        const int x = ox + l / 81, y = oy + (l / 9) % 9, z = oz + l % 9;
        if (x >= X || y >= Y || z >= Z) continue;
        float *p = g + (((size_t)x * Y + y) * Z + z) * 32 + lane;
        const float v = (float)(l + lane) * 1e-3f;
        if (MODE == 0) unsafeAtomicAdd(p, v);
        else if (MODE == 1) *p = v;
        else __builtin_nontemporal_store(v, p);
    }
}

template <int MODE> void run_flush(const char *name, float *g, const int *order, int nb)
{
    flush_kernel<MODE><<<nb, 256>>>(g, order, 128, 128, 128);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) flush_kernel<MODE><<<nb, 256>>>(g, order, 128, 128, 128);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double lines = (double)nb * kLines;
    printf("B %-34s %8.3f ms  %7.2f G lines/s  %7.1f GB/s of line payload\n", name, ms, lines / ms / 1e6, lines * 128 / ms / 1e6);
}

int main()
{
    run_lds<0>("ds_add_f32 x4 (z-pair lines)");
    run_lds<1>("ds_add_u64 x4");
    run_lds<2>("ds_add_f64 x4");
    run_lds<3>("read-add-write f32 x4");
    run_lds<4>("ds_read_b32 x4");

    float *g; hipMalloc(&g, (size_t)128 * 128 * 128 * 32 * 4);
    hipMemset(g, 0, (size_t)128 * 128 * 128 * 32 * 4);
    const int nb = 32 * 16 * 16;
    int *h = (int *)malloc(nb * 4), *d;
    hipMalloc(&d, nb * 4);
    for (int i = 0; i < nb; i++) h[i] = i;
    hipMemcpy(d, h, nb * 4, hipMemcpyHostToDevice);
    run_flush<0>("atomic add, bricks in order", g, d, nb);
    run_flush<1>("plain store, bricks in order", g, d, nb);
    run_flush<2>("nontemporal store, in order", g, d, nb);
    unsigned s = 1;
    for (int i = nb - 1; i > 0; i--) { s = s * 1664525u + 1013904223u; int j = (s >> 8) % (i + 1); int t = h[i]; h[i] = h[j]; h[j] = t; }
    hipMemcpy(d, h, nb * 4, hipMemcpyHostToDevice);
    run_flush<0>("atomic add, bricks shuffled", g, d, nb);
    run_flush<1>("plain store, bricks shuffled", g, d, nb);
    for (int i = 0; i < nb; i++) h[i] = 4000 + (i & 7);                  // 8 hot bricks, 1024 workgroups each
    hipMemcpy(d, h, nb * 4, hipMemcpyHostToDevice);
    run_flush<0>("atomic add, 8 hot bricks", g, d, nb);
    for (int i = 0; i < nb; i++) h[i] = 4000;                            // one hot brick
    hipMemcpy(d, h, nb * 4, hipMemcpyHostToDevice);
    run_flush<0>("atomic add, 1 hot brick (x8192)", g, d, nb);
    return 0;
}
