// nnd_host.hip -- the HOST entry points of nndistance: the reference's `my_lib.nnd_forward` / `my_lib.nnd_backward`
// (toolbox/nndistance/src/my_lib.h:3-5, my_lib.c:6-118), which its NNDFunction calls for CPU tensors
// (functions/nnd.py:27-28,53-54; BASELINE.json configs[0] is exactly that call).  They are part of the reference's
// operator interface, not a fallback: CUDA tensors always go to the HIP kernels of nnd.hip and never come here.
//
// Same arithmetic as my_lib.c -- float differences and float left-to-right x*x + y*y + z*z (this file is compiled
// with -ffp-contract=off), strict `<` so that the first of equal minima wins, serial accumulation of the gradients
// inside one batch item -- spread over host threads by (batch item, block of queries) instead of one core.
//
// Round 4: the search runs 16 (AVX-512) or 8 (AVX2) QUERIES per vector -- lane = query, the target broadcast -- with separate
// multiplies and adds (no FMA: an fma would round x*x + y*y once instead of twice) and a strict `<` per lane: every lane
// executes the scalar loop's operations in the scalar loop's order, so dist and idx are bit-identical to my_lib.c whatever
// the width; the path is picked at run time from the CPU's features (the library itself is built for plain x86-64).
#include "common.hpp"
#include <algorithm>
#include <cstring>
#include <immintrin.h>
#include <thread>
#include <vector>

namespace genre {
namespace {

void nearest_scalar(int n, int m, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ dist,
                    int *__restrict__ idx, int j0, int j1);

#ifndef __HIP_DEVICE_COMPILE__          // (hipcc parses this file for gfx950 too: x86 intrinsics and CPU probes are host-pass only)
// U independent groups of queries per target: the chain best -> compare -> blend -> best is ~6 cycles long, and one group per
// target runs at that latency; with U groups in flight the loop is bound by issue instead (measured on a 2.1 GHz Xeon, one
// thread, 2048 x 2048 both directions: 1.57 ms with U = 1).  The last groups of a block are padded with copies of its last query.
template <int U>
__attribute__((target("avx2"))) void nearest_avx2_u(int m, const float *__restrict__ a, const float *__restrict__ b,
                                                    float *__restrict__ dist, int *__restrict__ idx, int j0, int j1)
{
    for (int j = j0; j < j1; j += 8 * U) {
        __m256 x1[U], y1[U], z1[U], best[U];
        __m256i besti[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            alignas(32) float qx[8], qy[8], qz[8];
            for (int l = 0; l < 8; l++) {
                const int q = std::min(j + u * 8 + l, j1 - 1);
                qx[l] = a[q * 3 + 0]; qy[l] = a[q * 3 + 1]; qz[l] = a[q * 3 + 2];
            }
            x1[u] = _mm256_load_ps(qx); y1[u] = _mm256_load_ps(qy); z1[u] = _mm256_load_ps(qz);
            // k == 0 initialises (my_lib.c:16: `k == 0 || d < best`)
            const __m256 x2 = _mm256_sub_ps(_mm256_set1_ps(b[0]), x1[u]), y2 = _mm256_sub_ps(_mm256_set1_ps(b[1]), y1[u]),
                         z2 = _mm256_sub_ps(_mm256_set1_ps(b[2]), z1[u]);
            best[u] = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(x2, x2), _mm256_mul_ps(y2, y2)), _mm256_mul_ps(z2, z2));
            besti[u] = _mm256_setzero_si256();
        }
        for (int k = 1; k < m; k++) {
            const __m256 bx = _mm256_set1_ps(b[k * 3 + 0]), by = _mm256_set1_ps(b[k * 3 + 1]), bz = _mm256_set1_ps(b[k * 3 + 2]);
            const __m256i kk = _mm256_set1_epi32(k);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const __m256 x2 = _mm256_sub_ps(bx, x1[u]), y2 = _mm256_sub_ps(by, y1[u]), z2 = _mm256_sub_ps(bz, z1[u]);
                const __m256 d = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(x2, x2), _mm256_mul_ps(y2, y2)), _mm256_mul_ps(z2, z2));
                const __m256 lt = _mm256_cmp_ps(d, best[u], _CMP_LT_OQ);   // strict, false on NaN: as the scalar `d < best`
                best[u] = _mm256_blendv_ps(best[u], d, lt);
                besti[u] = _mm256_castps_si256(_mm256_blendv_ps(_mm256_castsi256_ps(besti[u]), _mm256_castsi256_ps(kk), lt));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            alignas(32) float bd[8];
            alignas(32) int bi[8];
            _mm256_store_ps(bd, best[u]);
            _mm256_store_si256(reinterpret_cast<__m256i *>(bi), besti[u]);
            for (int l = 0; l < 8 && j + u * 8 + l < j1; l++) { dist[j + u * 8 + l] = bd[l]; idx[j + u * 8 + l] = bi[l]; }
        }
    }
}

__attribute__((target("avx2"))) void nearest_avx2(int n, int m, const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ dist, int *__restrict__ idx, int j0, int j1)
{
    if (m <= 0) { nearest_scalar(n, m, a, b, dist, idx, j0, j1); return; }
    if (j1 - j0 > 8) nearest_avx2_u<2>(m, a, b, dist, idx, j0, j1);       // 16 ymm registers: two groups
    else nearest_avx2_u<1>(m, a, b, dist, idx, j0, j1);
}

template <int U>
__attribute__((target("avx512f"))) void nearest_avx512_u(int m, const float *__restrict__ a, const float *__restrict__ b,
                                                         float *__restrict__ dist, int *__restrict__ idx, int j0, int j1)
{
    for (int j = j0; j < j1; j += 16 * U) {
        __m512 x1[U], y1[U], z1[U], best[U];
        __m512i besti[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            alignas(64) float qx[16], qy[16], qz[16];
            for (int l = 0; l < 16; l++) {
                const int q = std::min(j + u * 16 + l, j1 - 1);
                qx[l] = a[q * 3 + 0]; qy[l] = a[q * 3 + 1]; qz[l] = a[q * 3 + 2];
            }
            x1[u] = _mm512_load_ps(qx); y1[u] = _mm512_load_ps(qy); z1[u] = _mm512_load_ps(qz);
            const __m512 x2 = _mm512_sub_ps(_mm512_set1_ps(b[0]), x1[u]), y2 = _mm512_sub_ps(_mm512_set1_ps(b[1]), y1[u]),
                         z2 = _mm512_sub_ps(_mm512_set1_ps(b[2]), z1[u]);
            best[u] = _mm512_add_ps(_mm512_add_ps(_mm512_mul_ps(x2, x2), _mm512_mul_ps(y2, y2)), _mm512_mul_ps(z2, z2));
            besti[u] = _mm512_setzero_si512();
        }
        for (int k = 1; k < m; k++) {
            const __m512 bx = _mm512_set1_ps(b[k * 3 + 0]), by = _mm512_set1_ps(b[k * 3 + 1]), bz = _mm512_set1_ps(b[k * 3 + 2]);
            const __m512i kk = _mm512_set1_epi32(k);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const __m512 x2 = _mm512_sub_ps(bx, x1[u]), y2 = _mm512_sub_ps(by, y1[u]), z2 = _mm512_sub_ps(bz, z1[u]);
                const __m512 d = _mm512_add_ps(_mm512_add_ps(_mm512_mul_ps(x2, x2), _mm512_mul_ps(y2, y2)), _mm512_mul_ps(z2, z2));
                const __mmask16 lt = _mm512_cmp_ps_mask(d, best[u], _CMP_LT_OQ);
                best[u] = _mm512_mask_mov_ps(best[u], lt, d);
                besti[u] = _mm512_mask_mov_epi32(besti[u], lt, kk);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            alignas(64) float bd[16];
            alignas(64) int bi[16];
            _mm512_store_ps(bd, best[u]);
            _mm512_store_si512(reinterpret_cast<void *>(bi), besti[u]);
            for (int l = 0; l < 16 && j + u * 16 + l < j1; l++) { dist[j + u * 16 + l] = bd[l]; idx[j + u * 16 + l] = bi[l]; }
        }
    }
}

__attribute__((target("avx512f"))) void nearest_avx512(int n, int m, const float *__restrict__ a, const float *__restrict__ b,
                                                       float *__restrict__ dist, int *__restrict__ idx, int j0, int j1)
{
    if (m <= 0) { nearest_scalar(n, m, a, b, dist, idx, j0, j1); return; }
    if (j1 - j0 > 32) nearest_avx512_u<4>(m, a, b, dist, idx, j0, j1);    // 32 zmm registers: four groups
    else if (j1 - j0 > 16) nearest_avx512_u<2>(m, a, b, dist, idx, j0, j1);
    else nearest_avx512_u<1>(m, a, b, dist, idx, j0, j1);
}

#endif

using nearest_fn = void (*)(int, int, const float *, const float *, float *, int *, int, int);

// which path runs: picked once per process from the CPU's features; GENRE_NND_HOST_ISA = scalar | avx2 | avx512 pins one
// (tests).  A request the CPU cannot serve is NOT silently replaced: the name reported by genre_nnd_host_isa() is then the
// path that really runs, and tests/test_nnd_host.py skips instead of passing on another width.
struct HostIsa { nearest_fn fn; const char *name; };

const HostIsa &host_isa()
{
#ifdef __HIP_DEVICE_COMPILE__
    static const HostIsa h{(nearest_fn)nearest_scalar, "scalar"};
    return h;
#else
    static const HostIsa h = [] {
        const char *e = getenv("GENRE_NND_HOST_ISA");
        __builtin_cpu_init();
        const bool a512 = __builtin_cpu_supports("avx512f"), a2 = __builtin_cpu_supports("avx2");
        const bool want_scalar = e && strcmp(e, "scalar") == 0, want_a2 = e && strcmp(e, "avx2") == 0;
        if (want_scalar) return HostIsa{(nearest_fn)nearest_scalar, "scalar"};
        if (want_a2 && a2) return HostIsa{(nearest_fn)nearest_avx2, "avx2"};
        if (a512) return HostIsa{(nearest_fn)nearest_avx512, "avx512"};
        if (a2) return HostIsa{(nearest_fn)nearest_avx2, "avx2"};
        return HostIsa{(nearest_fn)nearest_scalar, "scalar"};
    }();
    return h;
#endif
}

nearest_fn pick_nearest() { return host_isa().fn; }

void nearest_scalar(int n, int m, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ dist,
                    int *__restrict__ idx, int j0, int j1)
{
    for (int j = j0; j < j1; j++) {
        const float x1 = a[j * 3 + 0], y1 = a[j * 3 + 1], z1 = a[j * 3 + 2];
        float best = 0.f;
        int besti = 0;
        for (int k = 0; k < m; k++) {
            const float x2 = b[k * 3 + 0] - x1, y2 = b[k * 3 + 1] - y1, z2 = b[k * 3 + 2] - z1;
            const float d = x2 * x2 + y2 * y2 + z2 * z2;
            if (k == 0 || d < best) { best = d; besti = k; }
        }
        dist[j] = best;
        idx[j] = besti;
    }
}

// `want`: how many threads the work is worth (a thread costs ~50 us to start and join: measured on the MI355X host, the
// 2048 x 2048 pair of configs[0] -- 8.4 M distance evaluations, 0.4 ms of one core -- takes 0.48 ms on 16 threads, 0.23 ms on 4)
template <typename F>
void parallel_for(int64_t items, F &&fn, int64_t want = 16)
{
    unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    hw = (unsigned)std::max<int64_t>(1, std::min<int64_t>(hw, want));
    if (const char *e = getenv("GENRE_HOST_THREADS")) hw = std::max(1, atoi(e));      // (experiments / reproducible timings)
    const int nt = (int)std::min<int64_t>(hw, items);
    if (nt <= 1) { for (int64_t i = 0; i < items; i++) fn(i); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([=, &fn] { for (int64_t i = t; i < items; i += nt) fn(i); });
    for (auto &th : pool) th.join();
}

int check_clouds(const char *op, const genre_tensor *x1, const genre_tensor *x2)
{
    GENRE_REQUIRE(is_f32(x1, 3) && is_f32(x2, 3) && x1->size[2] == 3 && x2->size[2] == 3 && x1->size[0] == x2->size[0] &&
                      is_contiguous(x1) && is_contiguous(x2), "%s: clouds must be contiguous fp32 [B,n,3] / [B,m,3]", op);
    GENRE_REQUIRE(x1->size[1] < (1 << 30) && x2->size[1] < (1 << 30), "%s: cloud too large", op);
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" const char *genre_nnd_host_isa(void) { return host_isa().name; }

extern "C" int genre_nnd_forward_host(const genre_tensor *xyz1, const genre_tensor *xyz2, const genre_tensor *dist1,
                                      const genre_tensor *dist2, const genre_tensor *idx1, const genre_tensor *idx2)
{
    const char *op = "nnd_forward (host)";
    if (!check_clouds(op, xyz1, xyz2)) return 0;
    const int b = (int)xyz1->size[0], n = (int)xyz1->size[1], m = (int)xyz2->size[1];
    GENRE_REQUIRE(is_f32(dist1, 2) && is_f32(dist2, 2) && is_i32(idx1, 2) && is_i32(idx2, 2) && is_contiguous(dist1) &&
                      is_contiguous(dist2) && is_contiguous(idx1) && is_contiguous(idx2) && dist1->size[0] == b &&
                      dist1->size[1] == n && dist2->size[1] == m && idx1->size[1] == n && idx2->size[1] == m,
                  "%s: outputs must be contiguous dist [B,n] / [B,m] fp32 and idx int32", op);
    if (b == 0) return 1;
    const nearest_fn nearest = pick_nearest();
    const float *a = (const float *)xyz1->data, *c = (const float *)xyz2->data;
    constexpr int kBlock = 256;                                    // queries per work item
    const int nb1 = (n + kBlock - 1) / kBlock, nb2 = (m + kBlock - 1) / kBlock;
    parallel_for((int64_t)b * (nb1 + nb2), [&](int64_t w) {
        const int i = (int)(w / (nb1 + nb2)), blk = (int)(w % (nb1 + nb2));
        if (blk < nb1)
            nearest(n, m, a + (size_t)i * n * 3, c + (size_t)i * m * 3, (float *)dist1->data + (size_t)i * n,
                    (int *)idx1->data + (size_t)i * n, blk * kBlock, std::min(n, (blk + 1) * kBlock));
        else
            nearest(m, n, c + (size_t)i * m * 3, a + (size_t)i * n * 3, (float *)dist2->data + (size_t)i * m,
                    (int *)idx2->data + (size_t)i * m, (blk - nb1) * kBlock, std::min(m, (blk - nb1 + 1) * kBlock));
    }, ((int64_t)b * n * m * 2 + ((int64_t)1 << 21) - 1) >> 21);     // one thread per 2 M distance evaluations
    return 1;
}

extern "C" int genre_nnd_backward_host(const genre_tensor *xyz1, const genre_tensor *xyz2, const genre_tensor *gradxyz1,
                                       const genre_tensor *gradxyz2, const genre_tensor *graddist1,
                                       const genre_tensor *graddist2, const genre_tensor *idx1, const genre_tensor *idx2)
{
    const char *op = "nnd_backward (host)";
    if (!check_clouds(op, xyz1, xyz2)) return 0;
    const int b = (int)xyz1->size[0], n = (int)xyz1->size[1], m = (int)xyz2->size[1];
    GENRE_REQUIRE(is_f32(gradxyz1, 3) && is_f32(gradxyz2, 3) && same_shape(gradxyz1, xyz1) && same_shape(gradxyz2, xyz2) &&
                      is_contiguous(gradxyz1) && is_contiguous(gradxyz2), "%s: gradxyz must match the clouds", op);
    GENRE_REQUIRE(is_f32(graddist1, 2) && is_f32(graddist2, 2) && is_i32(idx1, 2) && is_i32(idx2, 2) &&
                      is_contiguous(graddist1) && is_contiguous(graddist2) && is_contiguous(idx1) && is_contiguous(idx2) &&
                      graddist1->size[1] == n && graddist2->size[1] == m && idx1->size[1] == n && idx2->size[1] == m,
                  "%s: graddist / idx must be contiguous [B,n] / [B,m]", op);
    parallel_for(b, [&](int64_t i) {
        const float *a = (const float *)xyz1->data + (size_t)i * n * 3, *c = (const float *)xyz2->data + (size_t)i * m * 3;
        float *ga = (float *)gradxyz1->data + (size_t)i * n * 3, *gc = (float *)gradxyz2->data + (size_t)i * m * 3;
        std::fill(ga, ga + (size_t)n * 3, 0.f);
        std::fill(gc, gc + (size_t)m * 3, 0.f);
        const float *g1 = (const float *)graddist1->data + (size_t)i * n, *g2 = (const float *)graddist2->data + (size_t)i * m;
        const int *i1 = (const int *)idx1->data + (size_t)i * n, *i2 = (const int *)idx2->data + (size_t)i * m;
        for (int pass = 0; pass < 2; pass++) {                      // direction 1 -> 2, then 2 -> 1 (my_lib.c:78-115)
            const float *p = pass ? c : a, *q = pass ? a : c, *g = pass ? g2 : g1;
            float *gp = pass ? gc : ga, *gq = pass ? ga : gc;
            const int *ix = pass ? i2 : i1;
            const int cnt = pass ? m : n;
            for (int j = 0; j < cnt; j++) {
                const int k = ix[j];
                const float w = g[j] * 2;
                for (int d = 0; d < 3; d++) {
                    const float t = w * (p[j * 3 + d] - q[k * 3 + d]);
                    gp[j * 3 + d] += t;
                    gq[k * 3 + d] -= t;
                }
            }
        }
    });
    return 1;
}
