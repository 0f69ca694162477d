// nnd.hip -- Chamfer nearest-neighbour distance (both directions) for gfx950.
//
// Replaces toolbox/nndistance/src/nnd_cuda.cu (K9 NmDistanceKernel :6-128,
// K10 NmDistanceGradKernel :143-162, launchers :129-177) and, for values, the
// CPU path my_lib.c:6-118.  The reference launches a fixed <<<(32,16),512>>>
// grid twice on the DEFAULT stream; at batch 1 only 4 of its 512 blocks work.
//
// Here ONE launch covers both directions.  A workgroup owns 128 queries (two per
// lane) and its 16 waves each scan a different slice of the target cloud, so a
// 2048 x 2048 problem already puts a wave on every SIMD of 32 CUs and a batch
// fills the chip.  Target coordinates are wave-uniform, so they are read
// through the scalar cache (s_load) straight into SGPR operands of the VALU
// ops -- no LDS staging, no broadcast reads.  The 16 partial minima are merged
// through LDS in slice order with a strict '<', which is exactly the
// reference's "first minimum wins" (my_lib.c:20, nnd_cuda.cu:26,120).
//
// Distances use the reference's fp32 expression x*x + y*y + z*z evaluated left
// to right WITHOUT fma contraction, so dist is bit-identical to the CPU
// reference and idx is bit-exact.  (No MFMA: the |a|^2+|b|^2-2ab GEMM form
// rounds differently and would break idx exactness.)
#include "common.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

constexpr int kSlices = 16;                 // waves per workgroup
constexpr int kNndBlock = kSlices * 64;

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float x = tx - qx, y = ty - qy, z = tz - qz;     // my_lib.c:15-17
    return x * x + y * y + z * z;                          // :18
}

// Q queries per lane (register blocking: one set of scalar target loads feeds Q distance pipelines).
// Minimum tracking is done per CHUNK of 4 targets: m = min(d0..d3) (2 instructions) and one strict-less
// update of (best, chunk) instead of four compare/select pairs; the index inside the winning chunk is
// resolved once at the end as the first target whose distance equals the minimum.  Same answer as the
// reference's "if (d < best)" per target: the earliest chunk that attains the minimum wins, and inside it
// the earliest target.  (v_min_f32 ignores a NaN operand, like the always-false compare does.)
constexpr int kQ = 2;

__global__ __launch_bounds__(kNndBlock) void nnd_forward_kernel(int n, int m, int qblocks1,
                                                                 const float *__restrict__ xyz1,
                                                                 const float *__restrict__ xyz2,
                                                                 float *__restrict__ dist1, int *__restrict__ idx1,
                                                                 float *__restrict__ dist2, int *__restrict__ idx2)
{
    __shared__ float s_d[kSlices][64 * kQ];
    __shared__ int s_i[kSlices][64 * kQ];

    const int b = blockIdx.y;
    const bool dir2 = (int)blockIdx.x >= qblocks1;
    const int qb = dir2 ? blockIdx.x - qblocks1 : blockIdx.x;
    const int nq = dir2 ? m : n, nt = dir2 ? n : m;
    const float *__restrict__ Q = (dir2 ? xyz2 : xyz1) + (int64_t)b * nq * 3;
    const float *__restrict__ T = (dir2 ? xyz1 : xyz2) + (int64_t)b * nt * 3;
    float *__restrict__ dout = (dir2 ? dist2 : dist1) + (int64_t)b * nq;
    int *__restrict__ iout = (dir2 ? idx2 : idx1) + (int64_t)b * nq;

    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = (nt + kSlices - 1) / kSlices;
    const int k0 = slice * per;
    int k1 = k0 + per;
    if (k1 > nt) k1 = nt;

    float qx[kQ], qy[kQ], qz[kQ], best[kQ];
    int bchunk[kQ];                                          // first target index of the best chunk
#pragma unroll
    for (int u = 0; u < kQ; u++) {
        const int j = qb * 64 * kQ + u * 64 + lane;
        qx[u] = qy[u] = qz[u] = 0.f;
        if (j < nq) { qx[u] = Q[j * 3 + 0]; qy[u] = Q[j * 3 + 1]; qz[u] = Q[j * 3 + 2]; }
        best[u] = 0.0f; bchunk[u] = 0;
    }
    if (k0 < k1) {
        const float *__restrict__ t = T + (int64_t)k0 * 3;
        // the first chunk initialises ("k==0 ||" of my_lib.c:20); it may be shorter than 4
        const int first_len = (k1 - k0 < 4) ? k1 - k0 : 4;
#pragma unroll
        for (int u = 0; u < kQ; u++) {
            float mch = sqdist(qx[u], qy[u], qz[u], t[0], t[1], t[2]);
            for (int e = 1; e < first_len; e++) mch = fminf(mch, sqdist(qx[u], qy[u], qz[u], t[e * 3], t[e * 3 + 1], t[e * 3 + 2]));
            best[u] = mch; bchunk[u] = k0;
        }
        int k = k0 + first_len;
        const float *__restrict__ tp = t + first_len * 3;
        for (; k + 4 <= k1; k += 4, tp += 12) {
            const float t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3], t4 = tp[4], t5 = tp[5];
            const float t6 = tp[6], t7 = tp[7], t8 = tp[8], t9 = tp[9], t10 = tp[10], t11 = tp[11];
#pragma unroll
            for (int u = 0; u < kQ; u++) {
                const float d0 = sqdist(qx[u], qy[u], qz[u], t0, t1, t2);
                const float d1 = sqdist(qx[u], qy[u], qz[u], t3, t4, t5);
                const float d2 = sqdist(qx[u], qy[u], qz[u], t6, t7, t8);
                const float d3 = sqdist(qx[u], qy[u], qz[u], t9, t10, t11);
                const float mch = fminf(fminf(d0, d1), fminf(d2, d3));
                if (mch < best[u]) { best[u] = mch; bchunk[u] = k; }
            }
        }
        for (; k < k1; k++, tp += 3) {                       // tail: chunks of one
#pragma unroll
            for (int u = 0; u < kQ; u++) {
                const float d = sqdist(qx[u], qy[u], qz[u], tp[0], tp[1], tp[2]);
                if (d < best[u]) { best[u] = d; bchunk[u] = k; }
            }
        }
        // resolve the index inside the winning chunk: first target (<= 4 candidates) with d == best
#pragma unroll
        for (int u = 0; u < kQ; u++) {
            const int c0 = bchunk[u];
            int found = c0;
            bool done = false;
            for (int e = 0; e < 4; e++) {
                const int kk = c0 + e;
                if (kk < k1 && !done) {
                    const float d = sqdist(qx[u], qy[u], qz[u], T[(int64_t)kk * 3], T[(int64_t)kk * 3 + 1], T[(int64_t)kk * 3 + 2]);
                    if (d == best[u]) { found = kk; done = true; }
                }
            }
            bchunk[u] = found;
        }
    }
#pragma unroll
    for (int u = 0; u < kQ; u++) { s_d[slice][u * 64 + lane] = best[u]; s_i[slice][u * 64 + lane] = bchunk[u]; }
    __syncthreads();
    // merge the slices in order with a strict '<' (first minimum wins); kQ*64 queries by the first kQ waves
    if (slice < kQ) {
        const int u = slice;
        const int j = qb * 64 * kQ + u * 64 + lane;
        if (j < nq) {
            float bd = s_d[0][u * 64 + lane];
            int bi = s_i[0][u * 64 + lane];                  // slice 0 is never empty when nt > 0;
#pragma unroll                                               // nt == 0 leaves (0, 0) like my_lib.c:12-13
            for (int sl = 1; sl < kSlices; sl++) {
                if (sl * per < nt) {
                    const float d = s_d[sl][u * 64 + lane];
                    if (d < bd) { bd = d; bi = s_i[sl][u * 64 + lane]; }
                }
            }
            dout[j] = bd;
            iout[j] = bi;
        }
    }
}

// Backward, pass A: each point's own term (plain store, initialises the outputs;
// replaces the two cudaMemset + atomicAdd-onto-zero of nnd_cuda.cu:150-155,164-165).
__global__ __launch_bounds__(256) void nnd_backward_own_kernel(int n, int m,
                                                                const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2,
                                                                const float *__restrict__ gd1,
                                                                const float *__restrict__ gd2,
                                                                const int *__restrict__ idx1,
                                                                const int *__restrict__ idx2,
                                                                float *__restrict__ gx1, float *__restrict__ gx2)
{
    const int b = blockIdx.y;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n + m; t += gridDim.x * 256) {
        const bool second = t >= n;
        const int j = second ? t - n : t;
        const int nq = second ? m : n, nt = second ? n : m;
        const float *Q = (second ? xyz2 : xyz1) + ((int64_t)b * nq + j) * 3;
        const int j2 = (second ? idx2 : idx1)[(int64_t)b * nq + j];
        const float *T = (second ? xyz1 : xyz2) + ((int64_t)b * nt + j2) * 3;
        const float g = (second ? gd2 : gd1)[(int64_t)b * nq + j] * 2;       // my_lib.c:88
        float *o = (second ? gx2 : gx1) + ((int64_t)b * nq + j) * 3;
        o[0] = g * (Q[0] - T[0]);
        o[1] = g * (Q[1] - T[1]);
        o[2] = g * (Q[2] - T[2]);
    }
}

// Backward, pass B: the scatter onto the matched point of the other cloud (my_lib.c:95-97,112-114).
__global__ __launch_bounds__(256) void nnd_backward_scatter_kernel(int n, int m,
                                                                    const float *__restrict__ xyz1,
                                                                    const float *__restrict__ xyz2,
                                                                    const float *__restrict__ gd1,
                                                                    const float *__restrict__ gd2,
                                                                    const int *__restrict__ idx1,
                                                                    const int *__restrict__ idx2,
                                                                    float *gx1, float *gx2)
{
    const int b = blockIdx.y;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n + m; t += gridDim.x * 256) {
        const bool second = t >= n;
        const int j = second ? t - n : t;
        const int nq = second ? m : n, nt = second ? n : m;
        const float *Q = (second ? xyz2 : xyz1) + ((int64_t)b * nq + j) * 3;
        const int j2 = (second ? idx2 : idx1)[(int64_t)b * nq + j];
        const float *T = (second ? xyz1 : xyz2) + ((int64_t)b * nt + j2) * 3;
        const float g = (second ? gd2 : gd1)[(int64_t)b * nq + j] * 2;
        float *o = (second ? gx1 : gx2) + ((int64_t)b * nt + j2) * 3;
        unsafeAtomicAdd(o + 0, -(g * (Q[0] - T[0])));
        unsafeAtomicAdd(o + 1, -(g * (Q[1] - T[1])));
        unsafeAtomicAdd(o + 2, -(g * (Q[2] - T[2])));
    }
}

int check_cloud(const char *op, const char *name, const genre_tensor *t, int64_t B)
{
    GENRE_REQUIRE(is_f32(t, 3) && t->size[2] == 3, "%s: %s must be a 3-D fp32 tensor [B,n,3]", op, name);
    GENRE_REQUIRE(B < 0 || t->size[0] == B, "%s: %s batch size mismatch", op, name);
    GENRE_REQUIRE(is_contiguous(t), "%s: %s must be contiguous (functions/nnd.py:16)", op, name);
    GENRE_REQUIRE(t->size[1] * 3 < ((int64_t)1 << 31), "%s: %s too many points", op, name);
    return 1;
}
int check_per_point(const char *op, const char *name, const genre_tensor *t, bool is_int, int64_t B, int64_t n)
{
    GENRE_REQUIRE(is_int ? is_i32(t, 2) : is_f32(t, 2), "%s: %s must be a 2-D %s tensor", op, name,
                  is_int ? "int32" : "fp32");
    GENRE_REQUIRE(t->size[0] == B && t->size[1] == n && is_contiguous(t), "%s: %s must be contiguous [%lld,%lld]",
                  op, name, (long long)B, (long long)n);
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_nnd_forward(const genre_tensor *xyz1, const genre_tensor *xyz2, const genre_tensor *dist1,
                                 const genre_tensor *dist2, const genre_tensor *idx1, const genre_tensor *idx2,
                                 void *stream)
{
    const char *op = "nnd_forward";
    if (!check_cloud(op, "xyz1", xyz1, -1)) return 0;
    const int64_t B = xyz1->size[0], n = xyz1->size[1];
    if (!check_cloud(op, "xyz2", xyz2, B)) return 0;
    const int64_t m = xyz2->size[1];
    if (!check_per_point(op, "dist1", dist1, false, B, n) || !check_per_point(op, "dist2", dist2, false, B, m) ||
        !check_per_point(op, "idx1", idx1, true, B, n) || !check_per_point(op, "idx2", idx2, true, B, m))
        return 0;
    GENRE_REQUIRE(B <= 65535, "%s: batch must be <= 65535", op);
    const int qb1 = ceil_div(n, 64 * kQ), qb2 = ceil_div(m, 64 * kQ);
    if (B == 0 || qb1 + qb2 == 0) return 1;
    nnd_forward_kernel<<<dim3(qb1 + qb2, (unsigned)B), kNndBlock, 0, (hipStream_t)stream>>>(
        (int)n, (int)m, qb1, (const float *)xyz1->data, (const float *)xyz2->data, (float *)dist1->data,
        (int *)idx1->data, (float *)dist2->data, (int *)idx2->data);
    GENRE_LAUNCH_CHECK("nnd updateOutput");
    return 1;
}

extern "C" int genre_nnd_backward(const genre_tensor *xyz1, const genre_tensor *xyz2, const genre_tensor *gradxyz1,
                                  const genre_tensor *gradxyz2, const genre_tensor *graddist1,
                                  const genre_tensor *graddist2, const genre_tensor *idx1,
                                  const genre_tensor *idx2, void *stream)
{
    const char *op = "nnd_backward";
    if (!check_cloud(op, "xyz1", xyz1, -1)) return 0;
    const int64_t B = xyz1->size[0], n = xyz1->size[1];
    if (!check_cloud(op, "xyz2", xyz2, B)) return 0;
    const int64_t m = xyz2->size[1];
    if (!check_cloud(op, "gradxyz1", gradxyz1, B) || !check_cloud(op, "gradxyz2", gradxyz2, B)) return 0;
    GENRE_REQUIRE(gradxyz1->size[1] == n && gradxyz2->size[1] == m, "%s: grad shapes must match the clouds", op);
    if (!check_per_point(op, "graddist1", graddist1, false, B, n) ||
        !check_per_point(op, "graddist2", graddist2, false, B, m) ||
        !check_per_point(op, "idx1", idx1, true, B, n) || !check_per_point(op, "idx2", idx2, true, B, m))
        return 0;
    GENRE_REQUIRE(B <= 65535, "%s: batch must be <= 65535", op);
    if (B == 0 || n + m == 0) return 1;
    GENRE_REQUIRE((n == 0) == (m == 0), "%s: one cloud is empty", op);
    hipStream_t st = (hipStream_t)stream;
    int gx = ceil_div(n + m, 256);
    if (gx > 1024) gx = 1024;
    const dim3 grid(gx, (unsigned)B);
    nnd_backward_own_kernel<<<grid, 256, 0, st>>>((int)n, (int)m, (const float *)xyz1->data,
                                                  (const float *)xyz2->data, (const float *)graddist1->data,
                                                  (const float *)graddist2->data, (const int *)idx1->data,
                                                  (const int *)idx2->data, (float *)gradxyz1->data,
                                                  (float *)gradxyz2->data);
    GENRE_LAUNCH_CHECK("nnd get grad (own)");
    nnd_backward_scatter_kernel<<<grid, 256, 0, st>>>((int)n, (int)m, (const float *)xyz1->data,
                                                      (const float *)xyz2->data, (const float *)graddist1->data,
                                                      (const float *)graddist2->data, (const int *)idx1->data,
                                                      (const int *)idx2->data, (float *)gradxyz1->data,
                                                      (float *)gradxyz2->data);
    GENRE_LAUNCH_CHECK("nnd get grad (scatter)");
    return 1;
}
