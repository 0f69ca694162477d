// nnd_host.hip -- the HOST entry points of nndistance: the reference's `my_lib.nnd_forward` / `my_lib.nnd_backward`
// (toolbox/nndistance/src/my_lib.h:3-5, my_lib.c:6-118), which its NNDFunction calls for CPU tensors
// (functions/nnd.py:27-28,53-54; BASELINE.json configs[0] is exactly that call).  They are part of the reference's
// operator interface, not a fallback: CUDA tensors always go to the HIP kernels of nnd.hip and never come here.
//
// Same arithmetic as my_lib.c -- float differences and float left-to-right x*x + y*y + z*z (this file is compiled
// with -ffp-contract=off), strict `<` so that the first of equal minima wins, serial accumulation of the gradients
// inside one batch item -- spread over host threads by (batch item, block of queries) instead of one core.
#include "common.hpp"
#include <algorithm>
#include <thread>
#include <vector>

namespace genre {
namespace {

void nearest(int n, int m, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ dist,
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

template <typename F>
void parallel_for(int64_t items, F &&fn)
{
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
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
    });
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
