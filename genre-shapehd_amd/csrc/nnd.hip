// nnd.hip -- Chamfer nearest-neighbour distance (both directions) for gfx950.
//
// Replaces toolbox/nndistance/src/nnd_cuda.cu (K9 NmDistanceKernel :6-128,
// K10 NmDistanceGradKernel :143-162, launchers :129-177) and, for values, the
// CPU path my_lib.c:6-118.  The reference launches a fixed <<<(32,16),512>>>
// grid twice on the DEFAULT stream; at batch 1 only 4 of its 512 blocks work.
//
// Here ONE launch covers both directions.  A workgroup owns 128 queries (two per
// lane, riding in the two halves of packed-fp32 ops) and its waves each scan a
// different slice of the target cloud: 16 slices for a lone cloud pair, so that
// 2048 x 2048 already puts a wave on every SIMD of 32 CUs; 8, 4 or 2 once the
// batch alone fills the chip (longer scans per wave, less prologue and merge per
// pair).  Target coordinates are wave-uniform, so they are read through the
// scalar cache (s_load, prefetched one chunk ahead) straight into SGPR operands
// of the VALU ops -- no LDS staging, no broadcast reads.  The partial minima are
// merged through LDS in slice order with a strict '<', which is exactly the
// reference's "first minimum wins" (my_lib.c:20, nnd_cuda.cu:26,120).
//
// Issue cost on gfx950 (tools/valu_rate_bench.hip, tools/pk_f32_bench.hip): a
// v_pk_{add,mul}_f32 takes ~4.2 cycles per wave for two pairs, a plain fp32 op on
// VGPRs ~2.5 for one (4.2 with an SGPR operand), v_min_f32 / v_min3_f32 ~4.3.  The
// scan is 4 packed ops + ~0.9 min + ~0.5 select per pair = ~22 cycles: a ceiling of
// ~56 TFLOP/s for this un-fused 8-flop formulation (measured: 45 at 32 x 2048^2).
//
// Distances use the reference's fp32 expression x*x + y*y + z*z evaluated left
// to right WITHOUT fma contraction, so dist is bit-identical to the CPU
// reference and idx is bit-exact.  (No MFMA: the |a|^2+|b|^2-2ab GEMM form
// rounds differently and would break idx exactness.)
#include "common.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

constexpr int kMaxSlices = 16;              // waves per workgroup: 16 for a lone cloud pair, fewer (longer scans per
                                            // wave, less prologue/merge per pair) once the batch fills the chip

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float x = tx - qx, y = ty - qy, z = tz - qz;     // my_lib.c:15-17
    return x * x + y * y + z * z;                          // :18
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 sqdist2(f2 qx, f2 qy, f2 qz, float tx, float ty, float tz)
{
    const f2 x = tx - qx, y = ty - qy, z = tz - qz;
    return x * x + y * y + z * z;
}

// Q queries per lane (register blocking: one set of scalar target loads feeds Q distance pipelines).
// Minimum tracking is done per CHUNK of kC = 8 targets: m = min(d0..d7) (v_min3 tree) and one strict-less
// update of (best, chunk) instead of eight compare/select pairs; the index inside the winning chunk is
// resolved once at the end as the first target whose distance equals the minimum.  Same answer as the
// reference's "if (d < best)" per target: the earliest chunk that attains the minimum wins, and inside it
// the earliest target.  (v_min_f32 ignores a NaN operand, like the always-false compare does.)
constexpr int kQ = 2;
constexpr int kC = 8;

template <int kSlices>
__global__ __launch_bounds__(kSlices * 64) void nnd_forward_kernel(int n, int m, int qblocks1,
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
        // the slice's first target initialises ("k==0 ||" of my_lib.c:20) -- even a NaN/inf distance, as in the
        // reference; the chunked scan then starts at that same target again (an equal distance never replaces)
#pragma unroll
        for (int u = 0; u < kQ; u++) { best[u] = sqdist(qx[u], qy[u], qz[u], t[0], t[1], t[2]); bchunk[u] = k0; }
        int k = k0;
        const float *__restrict__ tp = t;
        // main loop: the two queries of a lane ride in the two halves of packed fp32 ops (v_pk_add_f32 with a
        // negated operand, v_pk_mul_f32: IEEE results identical to the scalar forms, 2 pairs per instruction)
        static_assert(kQ == 2, "the packed main loop carries exactly two queries per lane");
        const f2 px = {qx[0], qx[1]}, py = {qy[0], qy[1]}, pz = {qz[0], qz[1]};
        // the chunk after the current one is requested before the current one is evaluated: scalar loads return
        // out of order, so their only wait is "all outstanding" -- without the prefetch every iteration starts
        // with an exposed scalar-cache round trip
        float c[3 * kC];
        if (k + kC <= k1) {
#pragma unroll
            for (int e = 0; e < 3 * kC; e++) c[e] = tp[e];              // wave-uniform: s_load into SGPR operands
        }
        for (; k + kC <= k1; k += kC, tp += 3 * kC) {
            float nx[3 * kC];
            const float *__restrict__ tn = (k + 2 * kC <= k1) ? tp + 3 * kC : tp;   // last round: harmless re-read
#pragma unroll
            for (int e = 0; e < 3 * kC; e++) nx[e] = tn[e];
            __builtin_amdgcn_sched_barrier(0);                          // keep the requests ahead of the arithmetic
            f2 d[kC];
#pragma unroll
            for (int e = 0; e < kC; e++) d[e] = sqdist2(px, py, pz, c[3 * e], c[3 * e + 1], c[3 * e + 2]);
#pragma unroll
            for (int u = 0; u < kQ; u++) {
                const float mch = fminf(fminf(fminf(d[0][u], d[1][u]), fminf(d[2][u], d[3][u])),
                                        fminf(fminf(d[4][u], d[5][u]), fminf(d[6][u], d[7][u])));
                if (mch < best[u]) { best[u] = mch; bchunk[u] = k; }
            }
#pragma unroll
            for (int e = 0; e < 3 * kC; e++) c[e] = nx[e];
        }
        if (k < k1 && k1 - k0 >= kC) {
            // ragged end: one more chunk over the slice's LAST kC targets.  It overlaps the previous chunk; the
            // re-evaluated targets cannot win (their distance is >= best, and only '<' replaces), and if a new
            // one wins, no re-evaluated target can equal the new minimum, so the first-match resolve stays right.
            k = k1 - kC;
            tp = T + (int64_t)k * 3;
            f2 d[kC];
#pragma unroll
            for (int e = 0; e < kC; e++) d[e] = sqdist2(px, py, pz, tp[3 * e], tp[3 * e + 1], tp[3 * e + 2]);
#pragma unroll
            for (int u = 0; u < kQ; u++) {
                const float mch = fminf(fminf(fminf(d[0][u], d[1][u]), fminf(d[2][u], d[3][u])),
                                        fminf(fminf(d[4][u], d[5][u]), fminf(d[6][u], d[7][u])));
                if (mch < best[u]) { best[u] = mch; bchunk[u] = k; }
            }
            k = k1;
        }
        for (; k < k1; k++, tp += 3) {                       // slices shorter than kC: chunks of one
#pragma unroll
            for (int u = 0; u < kQ; u++) {
                const float d = sqdist(qx[u], qy[u], qz[u], tp[0], tp[1], tp[2]);
                if (d < best[u]) { best[u] = d; bchunk[u] = k; }
            }
        }
        // resolve the index inside the winning chunk: first target (<= kC candidates) with d == best.  All
        // candidate coordinates are requested up front (one exposed latency, not kC dependent ones); a candidate
        // past the slice end is clamped onto the last target -- it can only repeat a distance already seen.
#pragma unroll
        for (int u = 0; u < kQ; u++) {
            const int c0 = bchunk[u];
            float cx[kC], cy[kC], cz[kC];
#pragma unroll
            for (int e = 0; e < kC; e++) {
                const int kk = (c0 + e < k1) ? c0 + e : k1 - 1;
                cx[e] = T[(int64_t)kk * 3]; cy[e] = T[(int64_t)kk * 3 + 1]; cz[e] = T[(int64_t)kk * 3 + 2];
            }
            int found = c0;
#pragma unroll
            for (int e = kC - 1; e >= 0; e--) {                          // descending: the earliest match is kept
                const int kk = (c0 + e < k1) ? c0 + e : k1 - 1;
                if (sqdist(qx[u], qy[u], qz[u], cx[e], cy[e], cz[e]) == best[u]) found = kk;
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

// ---- small problems: a lone cloud pair ----------------------------------------------------------------------
// With 128 queries per workgroup a single 2048 x 2048 pair (BASELINE.json configs[0]) is 32 workgroups = 32 of the
// 256 CUs (15.6 us).  Here the roles are turned round: a workgroup owns kSQ = 16 queries, held in SCALAR registers,
// the 256 lanes stride over the targets, and the per-lane minima are merged lexicographically on the 64-bit key
// (distance bits << 32 | index) -- a non-negative float orders like its bit pattern, and the smaller index wins among
// equal distances, which is the reference's "first strict minimum" (my_lib.c:19).  256 workgroups for that pair:
// 7.8 us instead of 15.6 (HIP-graph replay; ~2.5 us of it is the launch).
constexpr int kSQ = 16;

__global__ __launch_bounds__(256) void nnd_forward_small_kernel(int n, int m, int qblocks1, const float *__restrict__ xyz1,
                                                                 const float *__restrict__ xyz2, float *__restrict__ dist1,
                                                                 int *__restrict__ idx1, float *__restrict__ dist2,
                                                                 int *__restrict__ idx2)
{
    __shared__ unsigned long long s_key[kSQ][256];
    const int b = blockIdx.y;
    int qb = blockIdx.x;
    const bool second = qb >= qblocks1;                        // direction 2 -> 1
    if (second) qb -= qblocks1;
    const int nq = second ? m : n, nt = second ? n : m;
    const float *__restrict__ q = (second ? xyz2 : xyz1) + (size_t)b * nq * 3;
    const float *__restrict__ t = (second ? xyz1 : xyz2) + (size_t)b * nt * 3;
    float *__restrict__ dout = (second ? dist2 : dist1) + (size_t)b * nq;
    int *__restrict__ iout = (second ? idx2 : idx1) + (size_t)b * nq;
    float qx[kSQ], qy[kSQ], qz[kSQ], best[kSQ];
    int bi[kSQ];
#pragma unroll
    for (int u = 0; u < kSQ; u++) {
        const int j = min(qb * kSQ + u, nq - 1);              // uniform: scalar loads
        qx[u] = q[j * 3 + 0]; qy[u] = q[j * 3 + 1]; qz[u] = q[j * 3 + 2];
        bi[u] = -1;
        best[u] = 0.f;
    }
    // eight targets per lane in flight (all of a 2048-point cloud): their loads are issued before the first is used
    constexpr int kT = 8;
    for (int k0 = threadIdx.x; k0 < nt; k0 += 256 * kT) {
        float tx[kT], ty[kT], tz[kT];
#pragma unroll
        for (int i = 0; i < kT; i++) {
            const int k = min(k0 + i * 256, nt - 1);
            tx[i] = t[k * 3 + 0]; ty[i] = t[k * 3 + 1]; tz[i] = t[k * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < kT; i++) {
            const int k = k0 + i * 256;
            if (k < nt) {
#pragma unroll
                for (int u = 0; u < kSQ; u++) {
                    const float d = sqdist(qx[u], qy[u], qz[u], tx[i], ty[i], tz[i]);
                    if (bi[u] < 0 || d < best[u]) { best[u] = d; bi[u] = k; }  // my_lib.c:19 (k ascending per lane)
                }
            }
        }
    }
    // merge: keys through LDS, transposed -- thread (u, r) takes 16 of query u's 256 keys, then a 16-lane DPP row
    // minimum; no dependent chain of cross-lane operations
#pragma unroll
    for (int u = 0; u < kSQ; u++)
        s_key[u][threadIdx.x] = bi[u] < 0 ? ~0ull : (((unsigned long long)__float_as_uint(best[u])) << 32) | (unsigned)bi[u];
    __syncthreads();
    const int u = threadIdx.x >> 4, r = threadIdx.x & 15;
    unsigned long long key = ~0ull;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const unsigned long long o = s_key[u][i * 16 + r];                       // lanes of a row read 16 consecutive keys
        key = o < key ? o : key;
    }
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) {                                         // xor butterfly inside the 16-lane row
        const unsigned long long o = __shfl_xor(key, sh, 16);
        key = o < key ? o : key;
    }
    const int j = qb * kSQ + u;
    if (r == 0 && j < nq) {
        const bool none = key == ~0ull;                         // nt == 0 leaves (0, 0) like my_lib.c:12-13
        dout[j] = none ? 0.f : __uint_as_float((unsigned)(key >> 32));
        iout[j] = none ? 0 : (int)(unsigned)key;
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
    // slices per workgroup: as few as still put ~8 waves on each of the 1024 SIMDs
    const int64_t groups = (int64_t)B * (qb1 + qb2);
    if (groups <= 128 && n >= 1 && m >= 1) {        // too few 128-query workgroups for 256 CUs: 16-query workgroups
        const int sq1 = ceil_div(n, kSQ), sq2 = ceil_div(m, kSQ);
        nnd_forward_small_kernel<<<dim3(sq1 + sq2, (unsigned)B), 256, 0, (hipStream_t)stream>>>(
            (int)n, (int)m, sq1, (const float *)xyz1->data, (const float *)xyz2->data, (float *)dist1->data,
            (int *)idx1->data, (float *)dist2->data, (int *)idx2->data);
        GENRE_LAUNCH_CHECK("nnd updateOutput (small)");
        return 1;
    }
    int slices = kMaxSlices;
    while (slices > 2 && groups * (slices / 2) >= 8192) slices /= 2;
    const dim3 grid(qb1 + qb2, (unsigned)B);
#define GENRE_NND_LAUNCH(S)                                                                                        \
    nnd_forward_kernel<S><<<grid, S * 64, 0, (hipStream_t)stream>>>(                                               \
        (int)n, (int)m, qb1, (const float *)xyz1->data, (const float *)xyz2->data, (float *)dist1->data,           \
        (int *)idx1->data, (float *)dist2->data, (int *)idx2->data)
    switch (slices) {
    case 16: GENRE_NND_LAUNCH(16); break;
    case 8: GENRE_NND_LAUNCH(8); break;
    case 4: GENRE_NND_LAUNCH(4); break;
    default: GENRE_NND_LAUNCH(2); break;
    }
#undef GENRE_NND_LAUNCH
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
