// sph_render.hip -- fused voxel -> spherical depth map renderer for gfx950.
//
// Replaces the op sequence of render_spherical.forward (toolbox/spherical_proj.py:62-72):
// expand + permute + grid_sample (5-D trilinear, zeros padding, PyTorch-0.4.1 ==
// align_corners=True) + clamp + CalcStopProb (calc_prob_kernel.cu:113-143) +
// matmul(depth_weight) + prod(1-p) + add, and its autograd backward.  The reference moves
// ~150 MB per image forward (48 MiB grid buffer, four 16 MiB intermediates, taps); the
// forward here reads the 128^3 volume (8 MiB, L2/MALL resident) and writes the [R,R] map:
// algorithmic traffic 8.45 MB per image.
//
// Sample positions are regenerated in fp64 from the per-ray unit direction --
// grid[i,j,k] = float((2*dir_ij) * (1 - alpha_k)), the reference's own float64 expression
// (spherical_proj.py:50-56) -- bit-identical to its 48 MiB `grid` buffer without reading it.  Trilinear taps follow
// ATen's grid_sampler_3d arithmetic (what PyTorch runs for the reference).  Which samples touch which 16^3 voxel
// BRICK depends only on the geometry, so it comes from lists built once on the host (toolbox/_fused_render.py).
//
// FORWARD (brick path)  render_sample_brick_group_kernel: a workgroup stages the 18^3 tiles of one or two images in
//   LDS and evaluates the brick's listed samples (one lane per sample, 8 LDS reads each), writing the raw values
//   v[ray,k]; render_scan_fwd_kernel: a wave per ray, lane l owns samples 4l..4l+3, exclusive product scan of
//   (1-p) and depth expectation in fp64 registers + 6 DPP steps (wave_scan.hpp), optional sph_pad fan-out.
//
// BACKWARD (brick path), no global atomics:
//   A  render_scan_bwd_kernel (wave per ray) re-reads v and forms
//        dL/dp_k = g * ( T_k w_k - (sum_{j>k} s_j w_j + prod_all(1-p)) / (1 - p_k) ),
//      masked with torch.clamp's rule (pass where lo <= v <= hi), into a [rays, ZR] scratch, plus max|dL/dp|.
//   B  render_bwd_brick_kernel (workgroup per brick row) accumulates the trilinear adjoint of every listed sample
//      into a 32 KiB LDS tile and writes each voxel of grad_vox exactly once with plain stores.  The tile is 64-bit
//      FIXED POINT (ds_add_u64): measured on gfx950 ds_add_f32 sustains 0.33 lane-ops/clk/CU, ds_add_u64 6-10
//      (tools/lds_atomic_bench.hip).  The scale is 2^(44-e) with 2^e >= max|dL/dp|, leaving 18 bits of headroom for the
//      up-to-2^17 contributions a central voxel receives: 44 bits below the largest term -- finer than fp32
//      accumulation, and order-independent (deterministic).  Bricks are scheduled heaviest first; the eight central
//      ones are split over several rows (atomic flush onto pre-zeroed voxels).
//
// BATCH-MINOR volumes (image index fastest in memory, batches of >= 16) have their own renderer: sph_render_bm.hip.
//
// FALLBACKS without the tables: render_fwd_kernel / render_bwd_dp_kernel (a wave renders one ray with gathers from
//   global memory) and render_bwd_atomic_kernel (8 global fp32 atomics per sample: 1.3 ms/image, 95 % of it atomic
//   throughput -- all 16 384 rays converge on the central voxels).
#include "render_common.hpp"
#include <cstdlib>
#include "wave_scan.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

// lane-parallel store of one ray's value to all its padded positions (value is wave-uniform)
__device__ __forceinline__ void store_map(const RenderDims &D, float *oimg, const View4 &out, int q, int lane, float v)
{
    const int i = q / D.R, j = q % D.R;
    if (D.pad == 0) {
        if (lane == 0) oimg[i * out.s2 + j * out.s3] = v;
        return;
    }
    if (i > 0 && i < D.R - 1 && j >= D.pad && j < D.R - D.pad) {      // three quarters of the map: a single position
        if (lane == 0) oimg[(i + D.pad) * out.s2 + (j + D.pad) * out.s3] = v;
        return;
    }
    int r_lo, r_n, c0, c1;
    pad_span(D.R, D.pad, i, j, r_lo, r_n, c0, c1);
    const int cnt = r_n * (c1 >= 0 ? 2 : 1);
    for (int t = lane; t < cnt; t += 64) {
        const int second = t >= r_n;
        oimg[(r_lo + (second ? t - r_n : t)) * out.s2 + (second ? c1 : c0) * out.s3] = v;
    }
}

// gradient of one ray's value: sum over its padded positions
__device__ __forceinline__ float load_map_grad(const RenderDims &D, const float *gimg, const View4 &gout, int q)
{
    const int i = q / D.R, j = q % D.R;
    if (D.pad == 0) return gimg[i * gout.s2 + j * gout.s3];
    int r_lo, r_n, c0, c1;
    pad_span(D.R, D.pad, i, j, r_lo, r_n, c0, c1);
    float g = 0.f;
    for (int r = 0; r < r_n; r++) {
        g += gimg[(r_lo + r) * gout.s2 + c0 * gout.s3];
        if (c1 >= 0) g += gimg[(r_lo + r) * gout.s2 + c1 * gout.s3];
    }
    return g;
}

__device__ __forceinline__ float gather(const RenderDims &D, const float *__restrict__ base, const Cell &c)
{
    const int o = c.x0 * D.sx + c.y0 * D.sy + c.z0 * D.sz;
    float acc = 0.f;
    if (c.x0 >= 0 && c.x0 + 1 < D.X && c.y0 >= 0 && c.y0 + 1 < D.Y && c.z0 >= 0 && c.z0 + 1 < D.Z) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            acc += base[o + ((i & 1) ? D.sx : 0) + ((i & 2) ? D.sy : 0) + ((i & 4) ? D.sz : 0)] * corner_w(c, i);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int x = c.x0 + (i & 1), y = c.y0 + ((i >> 1) & 1), z = c.z0 + ((i >> 2) & 1);
            if (x >= 0 && x < D.X && y >= 0 && y < D.Y && z >= 0 && z < D.Z)
                acc += base[x * D.sx + y * D.sy + z * D.sz] * corner_w(c, i);
        }
    }
    return acc;
}

__device__ __forceinline__ double wave_sum(double v) { return wave_last(wave_incl_sum(v)); }

__device__ __forceinline__ void ray_decode(const RenderDims &D, int64_t r, int64_t &n, int &c, int &q)
{
    const int rr = D.R * D.R;
    q = (int)(r % rr);
    const int64_t nc = r / rr;
    c = (int)(nc % D.NC);
    n = nc / D.NC;
}

// Sampling with a coalescing-friendly mapping: in round c (0..3) lane l samples k = k0 + 64c + l, so
// the 64 lanes of one load instruction walk 32 voxels along the ray instead of 128 (half the
// distinct cache lines per gather).  The raw sample values go through a per-wave 1 KiB LDS row and
// come back blocked (lane l <- samples 4l..4l+3, one ds_read_b128) for the lane-local + wave scan.
// LDS traffic of one wave is processed in issue order, so wave-scope fences (compiler ordering only)
// are sufficient -- no workgroup barrier.
template <bool WITH_MASK>
__device__ __forceinline__ void lane_samples(const RenderDims &D, const float *__restrict__ base, double dx2,
                                             double dy2, double dz2, int k0, int lane, float *__restrict__ row,
                                             float (&p)[4], bool (&pass)[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int k = k0 + c * 64 + lane;
        float v = 0.f;                                               // v = 0 -> p = lo; beyond ZR masked below
        if (k < D.ZR) {
            float gx, gy, gz;
            sample_pos(D, dx2, dy2, dz2, k, gx, gy, gz);
            Cell cl;
            if (locate(D, gx, gy, gz, cl)) v = gather(D, base, cl);
        }
        row[c * 64 + lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4 v4 = *reinterpret_cast<const float4 *>(row + lane * 4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                 // row is reused by the next ray
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const bool live = k0 + lane * 4 + t < D.ZR;
        if (WITH_MASK) pass[t] = live && (vv[t] >= D.lo) && (vv[t] <= D.hi);   // torch.clamp backward mask
        p[t] = live ? fminf(fmaxf(vv[t], D.lo), D.hi) : 0.f;        // clamp(.,1e-5,1-1e-5), :66; 0 = neutral
    }
}

// ---- forward ----------------------------------------------------------------------------------
__device__ __forceinline__ double expect4(const RenderDims &D, const float (&p)[4], const float *__restrict__ dw,
                                          int kb, int lane, double &carry);

__global__ __launch_bounds__(kBlock) void render_fwd_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                             const float *__restrict__ dw, View4 out)
{
    __shared__ __attribute__((aligned(16))) float rows[kWavesPerBlock][256];
    const int lane = threadIdx.x & 63;
    float *row = rows[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t rays = (int64_t)D.N * D.NC * D.R * D.R;
    for (int64_t r = wave0; r < rays; r += nwaves) {
        int64_t n; int c, q;
        ray_decode(D, r, n, c, q);
        const float *__restrict__ base = vox.p + n * vox.s0 + c * vox.s1;
        const double dx2 = dirs[q * 3 + 0] * 2, dy2 = dirs[q * 3 + 1] * 2, dz2 = dirs[q * 3 + 2] * 2;
        double carry = 1.0, acc = 0.0;
        for (int k0 = 0; k0 < D.ZR; k0 += 256) {
            float p[4];
            bool unused[4];
            lane_samples<false>(D, base, dx2, dy2, dz2, k0, lane, row, p, unused);
            acc += expect4(D, p, dw, k0 + lane * 4, lane, carry);
        }
        const double total = wave_sum(acc) + carry;                      // + prod(1-p)  (:69-71)
        if (lane == 0) {
            const int i = q / D.R, j = q % D.R;
            out.p[n * out.s0 + c * out.s1 + i * out.s2 + j * out.s3] = (float)total;
        }
    }
}

// clamp + mask of 4 raw sample values (kb = first sample index of this lane)
__device__ __forceinline__ void clamp4(const RenderDims &D, const float (&v)[4], int kb, float (&p)[4], bool (&pass)[4])
{
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const bool live = kb + t < D.ZR;
        pass[t] = live && (v[t] >= D.lo) && (v[t] <= D.hi);           // torch.clamp backward mask
        p[t] = live ? fminf(fmaxf(v[t], D.lo), D.hi) : 0.f;           // clamp(.,1e-5,1-1e-5), :66; 0 = neutral
    }
}

// this lane's 4 depth weights (0 beyond the ray's end) -- loaded once per wave, not once per ray
__device__ __forceinline__ void load_w4(const RenderDims &D, const float *__restrict__ dw, int kb, double (&w)[4])
{
#pragma unroll
    for (int t = 0; t < 4; t++) w[t] = (kb + t < D.ZR) ? (double)dw[kb + t] : 0.0;
}

// forward scan of one ray chunk: returns this lane's share of sum_k s_k w_k, updates carry = prod(1-p)
__device__ __forceinline__ double expect4w(const float (&p)[4], const double (&w)[4], double &carry)
{
    const double q0 = 1.0 - (double)p[0], q1 = 1.0 - (double)p[1], q2 = 1.0 - (double)p[2], q3 = 1.0 - (double)p[3];
    const double e1 = q0, e2 = q0 * q1, e3 = e2 * q2, tot = e3 * q3;
    const double incl = wave_incl_prod(tot);
    const double excl = wave_prev(1.0, incl) * carry;
    // sum_k s_k * depth_weight[k]   (:68)
    const double acc = (((double)p[0] * excl) * w[0] + ((double)p[1] * (excl * e1)) * w[1]) +
                       (((double)p[2] * (excl * e2)) * w[2] + ((double)p[3] * (excl * e3)) * w[3]);
    carry *= wave_last(incl);
    return acc;
}
__device__ __forceinline__ double expect4(const RenderDims &D, const float (&p)[4], const float *__restrict__ dw,
                                          int kb, int lane, double &carry)
{
    double w[4];
    load_w4(D, dw, kb, w);
    return expect4w(p, w, carry);
}

// wave total of per-lane partial sums that are <= 1 in magnitude: the lane values are exact fp64 sums of 4
// terms; the 64-way tree runs in fp32 (v_add_f32 with a DPP operand: 6 instructions instead of 18), adding
// at most ~4e-7 relative error to a map whose tolerance is 1e-5
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowShr1, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowShr2, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowShr4, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowShr8, 0xf, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowBcast15, 0xa, 0xf, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kRowBcast31, 0xc, 0xf, false));
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// dL/dp of the lane's 4 samples (ZR <= 256: the whole ray is one chunk), clamp-masked; w = load_w4()
__device__ __forceinline__ void dp4w(const float (&p)[4], const bool (&pass)[4], const double (&w)[4], float g,
                                     float (&dp)[4])
{
    const double q0 = 1.0 - (double)p[0], q1 = 1.0 - (double)p[1], q2 = 1.0 - (double)p[2], q3 = 1.0 - (double)p[3];
    const double e2 = q0 * q1, e3 = e2 * q2, tot = e3 * q3;
    const double incl = wave_incl_prod(tot);
    const double excl = wave_prev(1.0, incl);
    const double prod_all = wave_last(incl);
    const double Tw0 = excl * w[0], Tw1 = (excl * q0) * w[1], Tw2 = (excl * e2) * w[2], Tw3 = (excl * e3) * w[3];
    const double sw0 = (double)p[0] * Tw0, sw1 = (double)p[1] * Tw1;      // s_k w_k, T_k = transmittance before k
    const double sw2 = (double)p[2] * Tw2, sw3 = (double)p[3] * Tw3;
    const double lane_sw = ((sw3 + sw2) + sw1) + sw0;
    const double incl_s = wave_incl_sum(lane_sw);                        // prefix over lanes <= lane
    const double after = (wave_last(incl_s) - incl_s) + prod_all;        // suffix; prod(1-p) joins it
    const double A3 = after, A2 = after + sw3, A1 = after + (sw3 + sw2), A0 = after + ((sw3 + sw2) + sw1);
    const double gd = (double)g;
    // A/(1-p) as A * rcp(1-p) in fp32: v_rcp_f32 is good to 1 ulp and 1-p lies in [1e-5, 1], so the quotient is
    // within ~2e-7 relative -- far inside the gradient's tolerance -- for 2 instructions instead of the ~10 of a
    // correctly rounded fp32 divide (an fp64 divide is ~30); everything feeding it is fp64
    dp[0] = pass[0] ? (float)(gd * (Tw0 - (double)((float)A0 * __builtin_amdgcn_rcpf(1.0f - p[0])))) : 0.f;
    dp[1] = pass[1] ? (float)(gd * (Tw1 - (double)((float)A1 * __builtin_amdgcn_rcpf(1.0f - p[1])))) : 0.f;
    dp[2] = pass[2] ? (float)(gd * (Tw2 - (double)((float)A2 * __builtin_amdgcn_rcpf(1.0f - p[2])))) : 0.f;
    dp[3] = pass[3] ? (float)(gd * (Tw3 - (double)((float)A3 * __builtin_amdgcn_rcpf(1.0f - p[3])))) : 0.f;
}
__device__ __forceinline__ void dp4(const RenderDims &D, const float (&p)[4], const bool (&pass)[4],
                                    const float *__restrict__ dw, float g, int lane, float (&dp)[4])
{
    double w[4];
    load_w4(D, dw, lane * 4, w);
    dp4w(p, pass, w, g, dp);
}

__device__ __forceinline__ void lane_dp(const RenderDims &D, const float *__restrict__ base, double dx2, double dy2,
                                        double dz2, const float *__restrict__ dw, float g, int lane,
                                        float *__restrict__ row, float (&dp)[4])
{
    float p[4];
    bool pass[4];
    lane_samples<true>(D, base, dx2, dy2, dz2, 0, lane, row, p, pass);
    dp4(D, p, pass, dw, g, lane, dp);
}

// A wave's max |dL/dp| as a bit pattern: patterns of non-negative floats order like their values, and Inf / NaN
// patterns sort above every finite one, so a non-finite gradient survives the maximum (the brick kernel then writes
// NaN for that image instead of pushing garbage through the fixed-point conversion).  One same-address global atomic
// costs ~10 ns, so publish only when it would raise the maximum (the racy pre-read is safe -- the value only grows).
__device__ __forceinline__ unsigned abs_bits4(const float (&dp)[4])
{
    const unsigned a = __float_as_uint(dp[0]) & 0x7fffffffu, b = __float_as_uint(dp[1]) & 0x7fffffffu;
    const unsigned c = __float_as_uint(dp[2]) & 0x7fffffffu, d = __float_as_uint(dp[3]) & 0x7fffffffu;
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ void publish_max(unsigned wmax, int lane, unsigned *dpmax_bits)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (unsigned)__shfl_xor((int)wmax, o, 64));
    if (lane == 0 && wmax > 0u &&
        wmax > __hip_atomic_load(dpmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dpmax_bits, wmax);
}

__device__ __forceinline__ void store_dp(const RenderDims &D, float *dst, int lane, const float (&dp)[4])
{
    if ((D.ZR & 3) == 0) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        if (lane * 4 < D.ZR) __builtin_nontemporal_store((v4f){dp[0], dp[1], dp[2], dp[3]}, reinterpret_cast<v4f *>(dst));
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (lane * 4 + t < D.ZR) dst[t] = dp[t];
    }
}

// ---- backward pass A: dL/dp -> scratch [rays, ZR] -----------------------------------------------
__global__ __launch_bounds__(kBlock) void render_bwd_dp_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                                const float *__restrict__ dw, View4 gout,
                                                                float *__restrict__ dpbuf,
                                                                unsigned *__restrict__ dpmax_bits)
{
    __shared__ __attribute__((aligned(16))) float rows[kWavesPerBlock][256];
    float *row = rows[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t rays = (int64_t)D.N * D.NC * D.R * D.R;
    unsigned wmax = 0u;
    int64_t cur = -1;                                                    // image the running maximum belongs to
    for (int64_t r = wave0; r < rays; r += nwaves) {
        int64_t n; int c, q;
        ray_decode(D, r, n, c, q);
        if (n * D.NC + c != cur) {
            if (cur >= 0) publish_max(wmax, lane, dpmax_bits + cur);
            cur = n * D.NC + c; wmax = 0u;
        }
        const int i = q / D.R, j = q % D.R;
        const float g = gout.p[n * gout.s0 + c * gout.s1 + i * gout.s2 + j * gout.s3];
        float dp[4] = {0.f, 0.f, 0.f, 0.f};
        if (g != 0.0f) {                                                 // wave-uniform
            const float *__restrict__ base = vox.p + n * vox.s0 + c * vox.s1;
            lane_dp(D, base, dirs[q * 3 + 0] * 2, dirs[q * 3 + 1] * 2, dirs[q * 3 + 2] * 2, dw, g, lane, row, dp);
        }
        wmax = max(wmax, abs_bits4(dp));
        store_dp(D, dpbuf + r * D.ZR + lane * 4, lane, dp);
    }
    if (cur >= 0) publish_max(wmax, lane, dpmax_bits + cur);
}

// ---- brick path, forward: LDS-staged voxel tiles ----------------------------------------------------
// fwd_table [rows,4] = (brick id, begin, end, -) into fwd_list (entries (ray << 8) | k); every in-volume
// sample appears once, under the brick that holds its base corner.  The workgroup stages the brick plus a one-voxel halo
// (18^3 floats, zeros outside the volume) with coalesced row reads -- each voxel leaves HBM/L2 once per
// row instead of once per tap -- then evaluates its samples with 8 LDS reads each and writes the raw
// value v[ray, k].

// Which cell a listed sample falls in and its eight trilinear weights depend on the geometry only, not on the
// image -- and that arithmetic (fp64 position, floor/convert, weight products: ~80 of the ~100 VALU instructions
// per sample, most of them 4-cycle forms) dominated a one-image-per-workgroup sampler.  For a batch, a workgroup
// therefore keeps the tiles of G images resident in LDS, walks the brick's sample list ONCE and evaluates every
// sample on all resident tiles: the geometry is paid once per G images, the per-image part is 8 LDS reads +
// 8 multiply-adds + one store.
// Measured at batch 32 (sample kernel alone): one image per workgroup 390 us; G = 2 x 512 threads, 3 x 512 and
// 4 x 1024 all 330 us -- of which ~120 us is tile staging and ~90 us the v stores (ablations), i.e. the kernel is
// now bound by its memory phases, not by the geometry arithmetic; G = 2 keeps three workgroups per CU.
constexpr int kGroup = 2, kGroupBlock = 512;

template <int G, int NT>
__global__ __launch_bounds__(NT) void render_sample_brick_group_kernel(RenderDims D, View5 vox,
                                                                        const double *__restrict__ dirs,
                                                                        const int *__restrict__ fwd_table,
                                                                        const int *__restrict__ fwd_list,
                                                                        float *__restrict__ vbuf, int imgs,
                                                                        int *__restrict__ live)
{
    extern __shared__ float gtile[];                                     // [G][kTile3]
    // (placing all rows of an image group on one XCD, as render_bwd_brick_kernel does, made THIS kernel 6 % slower)
    const int trow = blockIdx.x;
    const int img0 = blockIdx.y * G;
    const int ng = (imgs - img0 < G) ? imgs - img0 : G;
    const int brick = fwd_table[trow * 4 + 0];
    const int begin = fwd_table[trow * 4 + 1], end = fwd_table[trow * 4 + 2];
    const int nby = (D.Y + kBrick - 1) / kBrick, nbz = (D.Z + kBrick - 1) / kBrick;
    const int ox = (brick / (nby * nbz)) * kBrick - 1, oy = ((brick / nbz) % nby) * kBrick - 1,
              oz = (brick % nbz) * kBrick - 1;                            // tile origin (incl. halo)
    // tile element t = thread + i*NT walked incrementally, all loads of one image in flight together.  (Staging is
    // bound by the fetch itself -- 18-float rows straddle three 64-byte sectors -- not by load instructions: a
    // float4-core + scalar-halo variant measured the same.)
    constexpr int kPer = (kTile3 + NT - 1) / NT;
    constexpr int kSX = NT / (kTile * kTile), kSY = (NT % (kTile * kTile)) / kTile, kSZ = NT % kTile;
    static_assert(kSY + 1 < kTile && kSZ < kTile, "tile walk: one carry per axis");
    const int lz0 = (int)threadIdx.x % kTile, ly0 = ((int)threadIdx.x / kTile) % kTile,
              lx0 = (int)threadIdx.x / (kTile * kTile);
    const int step = kSX * D.sx + kSY * D.sy + kSZ * D.sz, wrap_z = D.sy - kTile * D.sz, wrap_y = D.sx - kTile * D.sy;
    int pass_g[G];
#pragma unroll
    for (int g = 0; g < G; g++) pass_g[g] = 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g >= ng) break;
        const int img = img0 + g;
        const float *__restrict__ base = vox.p + (img / D.NC) * vox.s0 + (img % D.NC) * vox.s1;
        float vals[kPer];
        unsigned inside = 0, own = 0;                                    // own: a voxel of the brick itself (not its halo)
        int lz = lz0, ly = ly0, x = ox + lx0, y = oy + ly0, z = oz + lz0;
        int off = x * D.sx + y * D.sy + z * D.sz;
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            vals[i] = 0.f;
            // The LOW halo planes (x == ox, y == oy, z == oz) are never fetched: a sample is listed under the brick of
            // its base corner, so inside the volume it reads tile indices 1..17 only; index 0 is reached only by base
            // corner -1, i.e. outside the volume, where grid_sample's zero padding applies.  That also makes every
            // z-row start on a 64-byte boundary (17 floats = 2 sectors instead of 18 floats straddling 3).
            if ((int)threadIdx.x + i * NT < kTile3 && x > ox && y > oy && z > oz && x < D.X && y < D.Y && z < D.Z) {
                vals[i] = base[off];
                inside |= 1u << i;
                if (x <= ox + kBrick && y <= oy + kBrick && z <= oz + kBrick) own |= 1u << i;
            }
            lz += kSZ; z += kSZ; ly += kSY; y += kSY; x += kSX; off += step;
            if (lz >= kTile) { lz -= kTile; z -= kTile; ly += 1; y += 1; off += wrap_z; }
            if (ly >= kTile) { ly -= kTile; y -= kTile; x += 1; off += wrap_y; }
        }
        int passes = 0;                                                  // some voxel of the BRICK passes the pre_scale clamp
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            if (D.pre_scale != 0.0f && (inside & (1u << i))) {           // depth_pred_with_sph_inpaint.py:124
                const float raw = vals[i] * D.pre_scale;
                vals[i] = fminf(fmaxf(raw, D.lo), D.hi);
                passes |= (vals[i] == raw && (own & (1u << i))) ? 1 : 0;  // lo <= raw <= hi: the clamp passes the gradient
            }
            if ((int)threadIdx.x + i * NT < kTile3) gtile[g * kTile3 + threadIdx.x + i * NT] = vals[i];
        }
        pass_g[g] = passes;
    }
    // WHAT THE CLAMP BLOCKS IS NOT COMPUTED (round 5; the batch-minor renderer's 3.4d for this layout).  With pre_scale the
    // backward multiplies every voxel's sum by its clamp mask.  live[img][0] = "some voxel of this image passes",
    // live[img][1 + brick] = "some voxel of this brick passes" (cleared by the host entry in front of this launch; every writer
    // stores the same 1): render_scan_bwd_kernel returns at once for a dead image, render_bwd_brick_kernel writes zeros for a
    // dead brick.  On GenRe's own chain -- clamp(proj * 50) of proj = 1 - 128 tdf, depth_pred_with_sph_inpaint.py:124: occupied
    // voxels saturate, empty ones sit below the lower bound -- that is every brick of every image.
    if (live != nullptr) {
        const int nbricks = ((D.X + kBrick - 1) / kBrick) * nby * nbz;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int any = __syncthreads_or(g < ng ? pass_g[g] : 0);
            if (threadIdx.x == 0 && g < ng && any) {
                int *lv = live + (int64_t)(img0 + g) * (nbricks + 1);
                lv[0] = 1;
                lv[1 + brick] = 1;
            }
        }
    } else {
        __syncthreads();
    }
    const int64_t img_stride = (int64_t)D.R * D.R * D.ZR;
    float *__restrict__ v0 = vbuf + (int64_t)img0 * img_stride;
    constexpr int kFwdInFlight = 2;
    for (int e0 = begin + threadIdx.x; e0 < end; e0 += kFwdInFlight * NT) {
        unsigned ent[kFwdInFlight];
        double d2[kFwdInFlight][3];
#pragma unroll
        for (int u = 0; u < kFwdInFlight; u++) {
            const int e = e0 + u * NT;
            ent[u] = (unsigned)fwd_list[e < end ? e : e0];
        }
#pragma unroll
        for (int u = 0; u < kFwdInFlight; u++) {
            const int q = (int)(ent[u] >> 8);
            d2[u][0] = dirs[q * 3 + 0]; d2[u][1] = dirs[q * 3 + 1]; d2[u][2] = dirs[q * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kFwdInFlight; u++) {
            if (e0 + u * NT >= end) break;
            const int q = (int)(ent[u] >> 8), k = (int)(ent[u] & 255u);
            float gx, gy, gz;
            sample_pos(D, d2[u][0] * 2, d2[u][1] * 2, d2[u][2] * 2, k, gx, gy, gz);
            Cell c;
            locate(D, gx, gy, gz, c);
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = corner_w(c, i);
            const float *tp = gtile + ((c.x0 - ox) * kTile + (c.y0 - oy)) * kTile + (c.z0 - oz);
            float *__restrict__ vq = v0 + ((unsigned)q * (unsigned)D.ZR + (unsigned)k);
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng) break;
                float acc = 0.f;                                          // ATen corner order, zeros outside
#pragma unroll
                for (int i = 0; i < 8; i++)
                    acc += tp[g * kTile3 + ((i & 1) ? kTile * kTile : 0) + ((i & 2) ? kTile : 0) + ((i & 4) ? 1 : 0)] * w[i];
                vq[g * img_stride] = acc;                   // plain store: L2 merges the partial lines (nontemporal: +60 us)
            }
        }
    }
}

template <int G, int NT>
int launch_sample_group(const char *op, const RenderDims &D, const genre_tensor *vox, const genre_tensor *dirs,
                        const genre_tensor *fwd_table, const genre_tensor *fwd_chunks, const genre_tensor *v_scratch,
                        int rows, int imgs, int *live, hipStream_t st)
{
    constexpr size_t lds = (size_t)G * kTile3 * sizeof(float);
    static std::atomic<uint64_t> done{0};
    if (!reserve_lds(op, reinterpret_cast<const void *>(&render_sample_brick_group_kernel<G, NT>), lds, done)) return 0;
    render_sample_brick_group_kernel<G, NT><<<dim3(rows, (imgs + G - 1) / G), NT, lds, st>>>(
        D, view5(vox), (const double *)dirs->data, (const int *)fwd_table->data, (const int *)fwd_chunks->data,
        (float *)v_scratch->data, imgs, live);
    return 1;
}

// raw sample values of this lane's 4 samples (0 before kin: outside the volume, never written)
__device__ __forceinline__ void load_v4(const float *__restrict__ vray, int kb, int k_in, int ZR, float (&v)[4])
{
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (kb + 3 >= k_in && kb < ZR) {
        const float4 v4 = *reinterpret_cast<const float4 *>(vray + kb);
        v[0] = kb + 0 >= k_in ? v4.x : 0.f; v[1] = kb + 1 >= k_in ? v4.y : 0.f;
        v[2] = kb + 2 >= k_in ? v4.z : 0.f; v[3] = kb + 3 >= k_in ? v4.w : 0.f;
    }
}

// forward scan over the raw values.  grid = (blocks, N*NC).  A wave owns rays w0, w0+nw, ... (<= 64 of them):
// their kin[] entries (which gate the v loads) are fetched up front with ONE load, one per lane, so the loop
// has no dependent load chain, and two rays are in flight per iteration.
__global__ __launch_bounds__(kBlock) void render_scan_fwd_kernel(RenderDims D, const float *__restrict__ vbuf,
                                                                  const int *__restrict__ kin,
                                                                  const float *__restrict__ dw, View4 out)
{
    const int lane = threadIdx.x & 63, kb = lane * 4;
    const int rr = D.R * D.R, img = blockIdx.y;
    const int w0 = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = gridDim.x * kWavesPerBlock;
    const float *__restrict__ vimg = vbuf + (int64_t)img * rr * D.ZR;
    float *oimg = out.p + (img / D.NC) * out.s0 + (img % D.NC) * out.s1;
    double w4[4];
    load_w4(D, dw, kb, w4);
    for (int qbase = w0; qbase < rr; qbase += 64 * nw) {
        const int myq = qbase + lane * nw;
        const int mykin = myq < rr ? kin[myq] : D.ZR;
        for (int it = 0; it < 64; it += 2) {
            const int q = qbase + it * nw;
            if (q >= rr) break;
            const int qb = q + nw;
            const bool hasb = qb < rr;
            float va[4], vb[4];
            load_v4(vimg + (int64_t)q * D.ZR, kb, __builtin_amdgcn_readlane(mykin, it), D.ZR, va);
            load_v4(vimg + (int64_t)(hasb ? qb : q) * D.ZR, kb, __builtin_amdgcn_readlane(mykin, hasb ? it + 1 : it),
                    D.ZR, vb);
            float p[4];
            bool pass[4];
            double carry = 1.0;
            clamp4(D, va, kb, p, pass);
            float total = wave_sum_f32((float)expect4w(p, w4, carry)) + (float)carry;     // + prod(1-p)  (:69-71)
            store_map(D, oimg, out, q, lane, total);
            if (hasb) {
                carry = 1.0;
                clamp4(D, vb, kb, p, pass);
                total = wave_sum_f32((float)expect4w(p, w4, carry)) + (float)carry;
                store_map(D, oimg, out, qb, lane, total);
            }
        }
    }
}

// backward scan: v -> dL/dp (same maths as render_bwd_dp_kernel without the sampling); kin[] and the rays'
// output gradients are prefetched one per lane like above
__global__ __launch_bounds__(kBlock) void render_scan_bwd_kernel(RenderDims D, const float *__restrict__ vbuf,
                                                                  const int *__restrict__ kin,
                                                                  const float *__restrict__ dw, View4 gout,
                                                                  float *__restrict__ dpbuf,
                                                                  unsigned *__restrict__ dpmax_bits,
                                                                  const int *__restrict__ live, int live_stride)
{
    const int lane = threadIdx.x & 63, kb = lane * 4;
    const int rr = D.R * D.R, img = blockIdx.y;
    // no voxel of this image passes the pre_scale clamp (forward's live word): its grad_vox is identically zero, nothing reads
    // its dL/dp (render_bwd_brick_kernel writes the zeros without looking at the list)
    if (live != nullptr && live[(int64_t)img * live_stride] == 0) return;
    const int w0 = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = gridDim.x * kWavesPerBlock;
    const float *__restrict__ vimg = vbuf + (int64_t)img * rr * D.ZR;
    float *__restrict__ dimg = dpbuf + (int64_t)img * rr * D.ZR;
    const float *gimg = gout.p + (img / D.NC) * gout.s0 + (img % D.NC) * gout.s1;
    unsigned wmax = 0u;
    double w4[4];
    load_w4(D, dw, kb, w4);
    for (int qbase = w0; qbase < rr; qbase += 64 * nw) {
        const int myq = qbase + lane * nw;
        const int mykin = myq < rr ? kin[myq] : D.ZR;
        const float myg = myq < rr ? load_map_grad(D, gimg, gout, myq) : 0.f;
        for (int it = 0; it < 64; it += 2) {
            const int q = qbase + it * nw;
            if (q >= rr) break;
            const int qb = q + nw;
            const bool hasb = qb < rr;
            const int itb = hasb ? it + 1 : it, q2 = hasb ? qb : q;
            float va[4], vb[4];
            const int kin_a = __builtin_amdgcn_readlane(mykin, it), kin_b = __builtin_amdgcn_readlane(mykin, itb);
            load_v4(vimg + (int64_t)q * D.ZR, kb, kin_a, D.ZR, va);
            load_v4(vimg + (int64_t)q2 * D.ZR, kb, kin_b, D.ZR, vb);
            const float ga = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(myg), it));
            const float gb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(myg), itb));
            float p[4], dp[4];
            bool pass[4];
            dp[0] = dp[1] = dp[2] = dp[3] = 0.f;
            if (ga != 0.0f) { clamp4(D, va, kb, p, pass); dp4w(p, pass, w4, ga, dp); }
            wmax = max(wmax, abs_bits4(dp));
            // samples before kin lie outside the volume: no brick list names them, so their dL/dp is never read
            if (kb + 3 >= kin_a) store_dp(D, dimg + (int64_t)q * D.ZR + kb, lane, dp);
            if (hasb) {
                dp[0] = dp[1] = dp[2] = dp[3] = 0.f;
                if (gb != 0.0f) { clamp4(D, vb, kb, p, pass); dp4w(p, pass, w4, gb, dp); }
                wmax = max(wmax, abs_bits4(dp));
                if (kb + 3 >= kin_b) store_dp(D, dimg + (int64_t)qb * D.ZR + kb, lane, dp);
            }
        }
    }
    publish_max(wmax, lane, dpmax_bits + img);
}

// ---- backward pass B: brick-owned accumulation ----------------------------------------------------
// brick_table [rows,4] = (brick id, begin, end, mode) into chunk_list, heaviest first; mode 0: the
// row covers the whole brick (plain stores); mode 1: the brick's list is split over several rows
// (each flushes its tile with atomics onto the brick pre-zeroed by zero_shared_bricks_kernel).
// chunk_list entries = (ray q << 8) | sample k, sorted by (brick, ray, sample): one lane per entry, so the
// VALU-bound loop runs on full waves; neighbouring lanes read consecutive dL/dp values.
__global__ __launch_bounds__(kBlock) void render_bwd_brick_kernel(RenderDims D, const double *__restrict__ dirs,
                                                                   const float *__restrict__ dpbuf,
                                                                   const int *__restrict__ brick_table,
                                                                   const int *__restrict__ chunk_list,
                                                                   const unsigned *__restrict__ dpmax_bits, View5 vox,
                                                                   View5 gvox, const int *__restrict__ live)
{
    __shared__ unsigned long long tile[kBrick * kBrick * kBrick];
    // XCD-aware order: workgroups go to the 8 XCDs round-robin by linear id; when the image count allows it, all rows
    // of an image run on ONE XCD (images xcd, xcd + 8, ... in turn), so that image's dL/dp lines and clamp mask meet in
    // that XCD's L2 instead of being fetched by all eight
    int img = blockIdx.y, trow = blockIdx.x;
    const int n_img = gridDim.y, n_row = gridDim.x;
    if ((n_img & 7) == 0) {
        const int lin = blockIdx.y * n_row + blockIdx.x;
        const int xcd = lin & 7, idx = lin >> 3;
        img = xcd + 8 * (idx / n_row);
        trow = idx % n_row;
    }
    if (live != nullptr) {
        // the clamp blocks every voxel of this brick (or of the whole image): its gradient is zero whatever arrives from above
        // (the clamp adjoint is a select -- not even a NaN survives it, as in the batch-minor renderer): stream the zeros, or
        // nothing at all where zero_shared_bricks_kernel already wrote them, before any list word is read
        const int brick_e = brick_table[trow * 4 + 0];
        const int nby_e = (D.Y + kBrick - 1) / kBrick, nbz_e = (D.Z + kBrick - 1) / kBrick;
        const int nbricks = ((D.X + kBrick - 1) / kBrick) * nby_e * nbz_e;
        const int *lv = live + (int64_t)img * (nbricks + 1);
        if (lv[0] == 0 || (lv[1 + brick_e] & 1) == 0) {
            if (brick_table[trow * 4 + 3] == 0) {
                const int oxe = (brick_e / (nby_e * nbz_e)) * kBrick, oye = ((brick_e / nbz_e) % nby_e) * kBrick,
                          oze = (brick_e % nbz_e) * kBrick;
                float *gb = gvox.p + (img / D.NC) * gvox.s0 + (img % D.NC) * gvox.s1;
                // 16-byte stores where the z rows allow it (dense volumes).  PLAIN stores: a brick's z row is 64 bytes, half of a
                // 128-byte line whose other half belongs to the next brick -- nontemporal stores (no merging in L2) measured
                // 77 -> 96 us at batch 32
                const bool v4 = gvox.s4 == 1 && ((gvox.s0 | gvox.s1 | gvox.s2 | gvox.s3) & 3) == 0 &&
                                (reinterpret_cast<uintptr_t>(gvox.p) & 15) == 0 && oze + kBrick <= D.Z;
                if (v4) {
                    typedef float v4f_ __attribute__((ext_vector_type(4)));
                    const int z4 = oze + ((int)threadIdx.x & 3) * 4, y = oye + (((int)threadIdx.x >> 2) & 15);
#pragma unroll
                    for (int i = 0; i < kBrick * kBrick * kBrick / 4 / kBlock; i++) {
                        const int x = oxe + ((int)threadIdx.x >> 6) + i * (kBlock / 64);
                        if (x < D.X && y < D.Y)
                            *reinterpret_cast<v4f_ *>(gb + x * gvox.s2 + y * gvox.s3 + z4) = (v4f_){0.f, 0.f, 0.f, 0.f};
                    }
                } else {
                    const int y = oye + (int)threadIdx.x / kBrick, z = oze + (int)threadIdx.x % kBrick;
                    if (y < D.Y && z < D.Z) {
#pragma unroll
                        for (int i = 0; i < kBrick * kBrick * kBrick / kBlock; i++)
                            if (oxe + i < D.X) gb[(oxe + i) * gvox.s2 + y * gvox.s3 + z * gvox.s4] = 0.f;
                    }
                }
            }
            return;
        }
    }
    // fixed-point scale 2^(44-e), 2^e >= THIS image's max|dL/dp| (see file header); max == 0 -> everything is 0.  Per
    // image, so that an image whose upstream gradient is many orders of magnitude below its neighbours' keeps all 44 bits.
    const unsigned maxbits = dpmax_bits[img];
    const bool nonfinite = maxbits >= 0x7f800000u;                       // an Inf / NaN dL/dp somewhere in this image
    int e = 0;
    (void)frexpf(nonfinite ? 1.0f : __uint_as_float(maxbits), &e);
    const double scale = ldexp(1.0, 44 - e), inv_scale = ldexp(1.0, e - 44);
    const int brick = brick_table[trow * 4 + 0];
    const int begin = brick_table[trow * 4 + 1], end = brick_table[trow * 4 + 2];
    const int shared = brick_table[trow * 4 + 3];
    const int nby = (D.Y + kBrick - 1) / kBrick, nbz = (D.Z + kBrick - 1) / kBrick;
    const int bx = brick / (nby * nbz), by = (brick / nbz) % nby, bz = brick % nbz;
    const int ox = bx * kBrick, oy = by * kBrick, oz = bz * kBrick;
    for (int t = threadIdx.x; t < kBrick * kBrick * kBrick; t += kBlock) tile[t] = 0ull;
    __syncthreads();
    const float *__restrict__ dpi = dpbuf + (int64_t)img * D.R * D.R * D.ZR;
    // One lane per listed sample, eight samples per thread in flight (4: +15 us, 16: +100 us): list words first
    // (coalesced), then the dependent dL/dp loads of all of them (64-byte runs: entries are sorted by ray, then
    // sample), then the work.
    constexpr int kInFlight = 8;
    for (int e0 = begin + threadIdx.x; e0 < end; e0 += kInFlight * kBlock) {
        unsigned ent[kInFlight];
        float dp[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            const int e = e0 + u * kBlock;
            ent[u] = e < end ? (unsigned)chunk_list[e] : 0xffffffffu;
        }
        double d2[kInFlight][3];
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            const bool ok = ent[u] != 0xffffffffu;
            const int q = ok ? (int)(ent[u] >> 8) : 0;
            // one image's samples fit 32 bits (R*R < 2^24, ZR <= 256): unsigned index from the image's base
            dp[u] = ok ? dpi[(unsigned)q * (unsigned)D.ZR + (ent[u] & 255u)] : 0.f;
            d2[u][0] = dirs[q * 3 + 0]; d2[u][1] = dirs[q * 3 + 1]; d2[u][2] = dirs[q * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < kInFlight; u++) {
            if (dp[u] == 0.0f) continue;
            const int k = (int)(ent[u] & 255u);
            float gx, gy, gz;
            sample_pos(D, d2[u][0] * 2, d2[u][1] * 2, d2[u][2] * 2, k, gx, gy, gz);
            Cell c;
            locate(D, gx, gy, gz, c);
            const int lx = c.x0 - ox, ly = c.y0 - oy, lz = c.z0 - oz;
            unsigned long long *tp = tile + (lx * kBrick + ly) * kBrick + lz;
            // fp32 weight products (x*y first, then *z, as ATen forms them) times dL/dp in fp32, then ONE
            // fp64 fma with the power-of-two scale and 1.5*2^52: the sum is an integer in the mantissa
            // (|value| <= 2^44) -- cvt + fma + a subtract on the high word per corner
            const float wxy[4] = {c.wx0 * c.wy0, c.wx1 * c.wy0, c.wx0 * c.wy1, c.wx1 * c.wy1};
            // dL/dp joins the z weight once per sample instead of once per corner (one fp32 rounding placed
            // differently from ATen's ((wx*wy)*wz)*g -- far below the fixed-point quantum)
            const float wzd0 = c.wz0 * dp[u], wzd1 = c.wz1 * dp[u];
            // ONE predicated path (a fast "all corners inside" branch would run in addition to the general one
            // in most waves): per-axis ownership bits, then eight exec-masked ds_add_u64
            const bool ax0 = (unsigned)lx < (unsigned)kBrick, ax1 = (unsigned)(lx + 1) < (unsigned)kBrick;
            const bool ay0 = (unsigned)ly < (unsigned)kBrick, ay1 = (unsigned)(ly + 1) < (unsigned)kBrick;
            const bool az0 = (unsigned)lz < (unsigned)kBrick, az1 = (unsigned)(lz + 1) < (unsigned)kBrick;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const bool own = ((i & 1) ? ax1 : ax0) && ((i & 2) ? ay1 : ay0) && ((i & 4) ? az1 : az0);
                if (own)
                    atomicAdd(tp + ((i & 1) ? kBrick * kBrick : 0) + ((i & 2) ? kBrick : 0) + ((i & 4) ? 1 : 0),
                              (unsigned long long)(__double_as_longlong(fma((double)(wxy[i & 3] * ((i & 4) ? wzd1 : wzd0)),
                                                                           scale, 6755399441055744.0)) -
                                                   0x4338000000000000LL));   // ds_add_u64
            }
        }
    }
    __syncthreads();
    const int n = img / D.NC, cc = img % D.NC;
    float *gbase = gvox.p + n * gvox.s0 + cc * gvox.s1;
    const float *vbase = vox.p + n * vox.s0 + cc * vox.s1;
    // flush.  Element t = thread + 256 i is voxel (lx = i, ly = thread / 16, lz = thread % 16): the clamp-mask values of
    // all 16 are requested before the first one is used (one exposed round trip instead of 16)
    constexpr int kPerT = kBrick * kBrick * kBrick / kBlock;
    const int ly = (int)threadIdx.x / kBrick, lz = (int)threadIdx.x % kBrick;
    const int y = oy + ly, z = oz + lz;
    const bool col_in = y < D.Y && z < D.Z;
    float tvs[kPerT];
#pragma unroll
    for (int i = 0; i < kPerT; i++) {
        tvs[i] = 0.f;
        if (D.pre_scale != 0.0f && col_in && ox + i < D.X) tvs[i] = vbase[(ox + i) * D.sx + y * D.sy + z * D.sz] * D.pre_scale;
    }
#pragma unroll
    for (int i = 0; i < kPerT; i++) {
        const int x = ox + i;
        if (col_in && x < D.X) {
            const unsigned long long acc = tile[threadIdx.x + i * kBlock];
            float *dst = gbase + x * gvox.s2 + y * gvox.s3 + z * gvox.s4;
            float val = (float)((double)(long long)acc * inv_scale);
            if (D.pre_scale != 0.0f)                         // adjoint of clamp(x * pre_scale, lo, hi)
                val = (tvs[i] >= D.lo && tvs[i] <= D.hi) ? val * D.pre_scale : 0.0f;
            if (nonfinite) val = __uint_as_float(0x7fc00000u);          // the reference chain would return NaN here too
            if (!shared) *dst = val;
            else if (acc != 0ull || nonfinite) unsafeAtomicAdd(dst, val);
        }
    }
}

__global__ __launch_bounds__(kBlock) void zero_shared_bricks_kernel(RenderDims D, const int *__restrict__ brick_table,
                                                                     View5 gvox)
{
    if (brick_table[blockIdx.x * 4 + 3] == 0) return;
    const int brick = brick_table[blockIdx.x * 4 + 0];
    const int nby = (D.Y + kBrick - 1) / kBrick, nbz = (D.Z + kBrick - 1) / kBrick;
    const int ox = (brick / (nby * nbz)) * kBrick, oy = ((brick / nbz) % nby) * kBrick, oz = (brick % nbz) * kBrick;
    const int img = blockIdx.y;
    float *gbase = gvox.p + (img / D.NC) * gvox.s0 + (img % D.NC) * gvox.s1;
    for (int t = threadIdx.x; t < kBrick * kBrick * kBrick; t += kBlock) {
        const int x = ox + t / (kBrick * kBrick), y = oy + (t / kBrick) % kBrick, z = oz + t % kBrick;
        if (x < D.X && y < D.Y && z < D.Z) gbase[x * gvox.s2 + y * gvox.s3 + z * gvox.s4] = 0.f;
    }
}

// ---- backward fallback: global-atomic scatter (no tables) ------------------------------------------
__global__ __launch_bounds__(kBlock) void render_bwd_atomic_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                                    const float *__restrict__ dw, View4 gout, View5 gvox)
{
    __shared__ __attribute__((aligned(16))) float rows[kWavesPerBlock][256];
    float *row = rows[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t rays = (int64_t)D.N * D.NC * D.R * D.R;
    for (int64_t r = wave0; r < rays; r += nwaves) {
        int64_t n; int c, q;
        ray_decode(D, r, n, c, q);
        const int i = q / D.R, j = q % D.R;
        const float g = gout.p[n * gout.s0 + c * gout.s1 + i * gout.s2 + j * gout.s3];
        if (g == 0.0f) continue;                                         // wave-uniform
        const float *__restrict__ base = vox.p + n * vox.s0 + c * vox.s1;
        float *gbase = gvox.p + n * gvox.s0 + c * gvox.s1;
        const double dx2 = dirs[q * 3 + 0] * 2, dy2 = dirs[q * 3 + 1] * 2, dz2 = dirs[q * 3 + 2] * 2;
        float dp[4];
        lane_dp(D, base, dx2, dy2, dz2, dw, g, lane, row, dp);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (dp[t] == 0.0f) continue;
            float gx, gy, gz;
            sample_pos(D, dx2, dy2, dz2, lane * 4 + t, gx, gy, gz);
            Cell cl;
            locate(D, gx, gy, gz, cl);
#pragma unroll
            for (int ci = 0; ci < 8; ci++) {
                const int x = cl.x0 + (ci & 1), y = cl.y0 + ((ci >> 1) & 1), z = cl.z0 + ((ci >> 2) & 1);
                if (x >= 0 && x < D.X && y >= 0 && y < D.Y && z >= 0 && z < D.Z)
                    unsafeAtomicAdd(gbase + x * gvox.s2 + y * gvox.s3 + z * gvox.s4, corner_w(cl, ci) * dp[t]);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void zero_vec4_kernel(float4 *__restrict__ a, int64_t n4)
{
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) a[i] = z;
}

inline int grid_for_rays(int64_t rays)
{
    int64_t b = (rays + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = (int64_t)kCUs * 8 * 8;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

// scan kernels: grid.y = image; grid.x sized so that a wave sees about two rays (both in flight)
inline dim3 scan_grid(const RenderDims &D)
{
    int bx = (D.R * D.R + 2 * kWavesPerBlock - 1) / (2 * kWavesPerBlock);
    const int imgs = D.N * D.NC;
    // keep the launch around 8k workgroups (measured at batch 32: 2k / 4k / 8k workgroups within 1 %, 32k +7 %,
    // 131k +13 % -- unlike the pure streaming kernels, these VALU-bound ones prefer long-lived waves)
    const int cap = (kCUs * 8 * 4 + imgs - 1) / imgs;
    if (bx > cap) bx = cap;
    return dim3(bx < 1 ? 1 : bx, imgs);
}

}  // namespace
}  // namespace genre

using namespace genre;

// Tables shared by the brick kernels (formats: include/genre_hip.h).
static int check_tables(const char *op, const RenderDims &D, const genre_tensor *table, const genre_tensor *chunks,
                        int &rows)
{
    const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
    GENRE_REQUIRE(is_i32(table, 2) && table->size[0] >= nb && table->size[1] == 4 && is_contiguous(table) &&
                      table->size[0] < (1 << 30),
                  "%s: brick table must be a contiguous int32 [rows >= %d, 4] tensor", op, nb);
    GENRE_REQUIRE(is_i32(chunks, 1) && is_contiguous(chunks), "%s: chunk list must be int32 [S]", op);
    GENRE_REQUIRE((int64_t)D.R * D.R < (1 << 24) && D.ZR <= 256, "%s: brick path needs R*R < 2^24 and ZR <= 256", op);
    GENRE_REQUIRE(D.N * D.NC <= 65535, "%s: N*NC must be <= 65535", op);
    rows = (int)table->size[0];
    return 1;
}

extern "C" int genre_render_spherical_forward(const genre_tensor *vox, const genre_tensor *dirs,
                                              const genre_tensor *depth_weight, const genre_tensor *out,
                                              const genre_tensor *v_scratch, const genre_tensor *fwd_table,
                                              const genre_tensor *fwd_chunks, const genre_tensor *kin,
                                              const genre_tensor *live, float pre_scale, void *stream)
{
    const char *op = "render_spherical_forward";
    RenderDims D{};
    if (!check_render(op, vox, dirs, depth_weight, out, D)) return 0;
    D.pre_scale = pre_scale;
    GENRE_REQUIRE(pre_scale == 0.0f || (v_scratch && fwd_table && fwd_chunks && kin),
                  "%s: pre_scale needs the brick path (pass the tables)", op);
    const int64_t rays = (int64_t)D.N * D.NC * D.R * D.R;
    if (rays == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (v_scratch && fwd_table && fwd_chunks && kin) {
        int rows = 0;
        if (!check_tables(op, D, fwd_table, fwd_chunks, rows)) return 0;
        GENRE_REQUIRE((D.ZR & 3) == 0, "%s: brick path needs ZR %% 4 == 0", op);
        GENRE_REQUIRE(is_f32(v_scratch, 1) && is_contiguous(v_scratch) && v_scratch->size[0] >= rays * D.ZR &&
                          aligned16(v_scratch->data),
                      "%s: v_scratch must be a contiguous, 16-byte aligned fp32 buffer of >= rays*ZR elements", op);
        GENRE_REQUIRE(is_i32(kin, 1) && is_contiguous(kin) && kin->size[0] == (int64_t)D.R * D.R,
                      "%s: kin must be int32 [R*R]", op);
        const int imgs = D.N * D.NC;
        int *live_p = nullptr;
        if (live != nullptr && pre_scale != 0.0f) {      // the clamp's pass words of the backward (file header; ABI 4)
            const int64_t nb = (int64_t)((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
            GENRE_REQUIRE(is_i32(live, 1) && is_contiguous(live) && live->size[0] >= (int64_t)imgs * (nb + 1),
                          "%s: live must be int32 [N*NC*(1 + bricks)] = [%lld]", op, (long long)(imgs * (nb + 1)));
            live_p = (int *)live->data;
            GENRE_REQUIRE(hipMemsetAsync(live_p, 0, (size_t)imgs * (nb + 1) * 4, st) == hipSuccess,
                          "%s: hipMemsetAsync of the live words failed", op);
        }
        // batches: kGroup images share one walk over the sample list; a lone image gets the same kernel with G = 1
        // (512 threads per workgroup: 38.9 us for the batch-1 forward chain against 42.1 with 256, 42.2 with 1024)
        const int ok = imgs >= 2
            ? launch_sample_group<kGroup, kGroupBlock>(op, D, vox, dirs, fwd_table, fwd_chunks, v_scratch, rows, imgs, live_p, st)
            : launch_sample_group<1, 512>(op, D, vox, dirs, fwd_table, fwd_chunks, v_scratch, rows, imgs, live_p, st);
        if (!ok) return 0;
        GENRE_LAUNCH_CHECK("render_spherical forward (bricks)");
        // (LDS-transposed per-lane serial scans for this layout -- 64 rays per wave, 16-sample tiles -- measured 136 us
        // forward / 243 us backward against 124-150 / 196 for the tree scans below: the tile loads serialise with the
        // recurrence; only the batch-minor layout, where lanes are images and loads need no transpose, profits)
        render_scan_fwd_kernel<<<scan_grid(D), kBlock, 0, st>>>(
            D, (const float *)v_scratch->data, (const int *)kin->data, (const float *)depth_weight->data, view4(out));
        GENRE_LAUNCH_CHECK("render_spherical forward (scan)");
        return 1;
    }
    GENRE_REQUIRE(D.pad == 0, "%s: the padded map layout needs the brick path (pass the tables)", op);
    render_fwd_kernel<<<grid_for_rays(rays), kBlock, 0, st>>>(
        D, view5(vox), (const double *)dirs->data, (const float *)depth_weight->data, view4(out));
    GENRE_LAUNCH_CHECK("render_spherical forward");
    return 1;
}

extern "C" int genre_render_spherical_backward(const genre_tensor *vox, const genre_tensor *dirs,
                                               const genre_tensor *depth_weight, const genre_tensor *grad_out,
                                               const genre_tensor *grad_vox, const genre_tensor *dp_scratch,
                                               const genre_tensor *brick_table, const genre_tensor *chunk_list,
                                               const genre_tensor *v_scratch, const genre_tensor *kin,
                                               const genre_tensor *live, float pre_scale, void *stream)
{
    const char *op = "render_spherical_backward";
    RenderDims D{};
    if (!check_render(op, vox, dirs, depth_weight, grad_out, D)) return 0;
    D.pre_scale = pre_scale;
    GENRE_REQUIRE(pre_scale == 0.0f || (brick_table && chunk_list && dp_scratch && v_scratch && kin),
                  "%s: pre_scale needs the brick path with the forward's v_scratch", op);
    GENRE_REQUIRE(is_f32(grad_vox, 5) && same_shape(grad_vox, vox), "%s: grad_vox must have the shape of vox", op);
    GENRE_REQUIRE(D.ZR <= 256, "%s: fused backward supports z_res <= 256", op);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nv = numel(grad_vox);
    const int64_t rays = (int64_t)D.N * D.NC * D.R * D.R;
    if (nv == 0) return 1;
    if (brick_table && chunk_list && dp_scratch) {
        int rows = 0;
        if (!check_tables(op, D, brick_table, chunk_list, rows)) return 0;
        const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
        const int imgs = D.N * D.NC;
        GENRE_REQUIRE(is_f32(dp_scratch, 1) && is_contiguous(dp_scratch) && dp_scratch->size[0] >= rays * D.ZR + imgs &&
                          aligned16(dp_scratch->data),
                      "%s: dp_scratch must be a contiguous, 16-byte aligned fp32 buffer of >= rays*ZR + N*NC elements", op);
        unsigned *dpmax = (unsigned *)dp_scratch->data + rays * D.ZR;       // per-image max|dL/dp| behind the samples
        const int *live_p = nullptr;                                        // the forward's clamp pass words (pre_scale only)
        if (live != nullptr && pre_scale != 0.0f && v_scratch && kin) {
            GENRE_REQUIRE(is_i32(live, 1) && is_contiguous(live) && live->size[0] >= (int64_t)imgs * (nb + 1),
                          "%s: live must be the forward's int32 [N*NC*(1 + bricks)] buffer", op);
            live_p = (const int *)live->data;
        }
        if (hipMemsetAsync(dpmax, 0, (size_t)imgs * 4, st) != hipSuccess) return fail("%s: hipMemsetAsync failed", op);
        if (rays > 0 && v_scratch && kin) {          // the per-sample forward left the raw value of EVERY sample: scan only
            GENRE_REQUIRE((D.ZR & 3) == 0, "%s: brick path needs ZR %% 4 == 0", op);
            GENRE_REQUIRE(is_f32(v_scratch, 1) && is_contiguous(v_scratch) && v_scratch->size[0] >= rays * D.ZR &&
                              aligned16(v_scratch->data),
                          "%s: v_scratch must be a contiguous, 16-byte aligned fp32 buffer of >= rays*ZR elements", op);
            GENRE_REQUIRE(is_i32(kin, 1) && is_contiguous(kin) && kin->size[0] == (int64_t)D.R * D.R,
                          "%s: kin must be int32 [R*R]", op);
            render_scan_bwd_kernel<<<scan_grid(D), kBlock, 0, st>>>(
                D, (const float *)v_scratch->data, (const int *)kin->data, (const float *)depth_weight->data,
                view4(grad_out), (float *)dp_scratch->data, dpmax, live_p, nb + 1);
            GENRE_LAUNCH_CHECK("render_spherical backward (scan)");
        } else if (rays > 0) {                       // recompute the samples from vox
            GENRE_REQUIRE(D.pad == 0, "%s: the padded map layout needs the forward's v_scratch", op);
            render_bwd_dp_kernel<<<grid_for_rays(rays), kBlock, 0, st>>>(
                D, view5(vox), (const double *)dirs->data, (const float *)depth_weight->data, view4(grad_out),
                (float *)dp_scratch->data, dpmax);
            GENRE_LAUNCH_CHECK("render_spherical backward (dL/dp)");
        }
        if (rows > nb) {        // some bricks are split over several rows: those accumulate with atomics
            zero_shared_bricks_kernel<<<dim3(rows, D.N * D.NC), kBlock, 0, st>>>(D, (const int *)brick_table->data,
                                                                                view5(grad_vox));
            GENRE_LAUNCH_CHECK("render_spherical backward (zero shared bricks)");
        }
        // (a batched variant like the forward's -- G images' u64 tiles resident, one walk over the list -- measured
        // SLOWER, 1090 us for G = 2 and 1260 us for G = 4 against 1005: this kernel is bound by the per-image part,
        // the dL/dp gathers and the LDS atomics, not by the shared geometry arithmetic)
        render_bwd_brick_kernel<<<dim3(rows, D.N * D.NC), kBlock, 0, st>>>(
            D, (const double *)dirs->data, (const float *)dp_scratch->data, (const int *)brick_table->data,
            (const int *)chunk_list->data, dpmax, view5(vox), view5(grad_vox), live_p);
        GENRE_LAUNCH_CHECK("render_spherical backward (bricks)");
        return 1;
    }
    GENRE_REQUIRE(is_contiguous(grad_vox) && aligned16(grad_vox->data) && nv % 4 == 0,
                  "%s: (atomic fallback) grad_vox must be contiguous, 16-byte aligned, numel %% 4 == 0", op);
    int64_t zb = (nv / 4 + kBlock - 1) / kBlock;
    if (zb > (1 << 20)) zb = 1 << 20;                    // one float4 per thread (see cam_bp's fill)
    zero_vec4_kernel<<<(int)zb, kBlock, 0, st>>>((float4 *)grad_vox->data, nv / 4);
    GENRE_LAUNCH_CHECK("render_spherical backward (zero)");
    if (rays == 0) return 1;
    GENRE_REQUIRE(D.pad == 0, "%s: the padded map layout needs the brick path", op);
    render_bwd_atomic_kernel<<<grid_for_rays(rays), kBlock, 0, st>>>(D, view5(vox), (const double *)dirs->data,
                                                                    (const float *)depth_weight->data,
                                                                    view4(grad_out), view5(grad_vox));
    GENRE_LAUNCH_CHECK("render_spherical backward");
    return 1;
}
