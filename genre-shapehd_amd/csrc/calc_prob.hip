// calc_prob.hip -- per-ray stop-probability scan and its adjoint for gfx950.
//
// Replaces toolbox/calc_prob/calc_prob/src/calc_prob_kernel.cu (K7 :113-143,
// K8 :146-189; wrappers :191-266).  The reference runs ONE THREAD PER RAY that
// walks 256 dependent global read-modify-write steps, with neighbouring threads
// 128 KiB apart (zero coalescing), after two redundant full-tensor zero fills.
//
// Here a ray is a wave-level scan.  With z innermost, one ray of 256 samples is
// 1 KiB: the 64 lanes of a wave each own 4 consecutive samples, so a ray is one
// perfectly coalesced 16 B/lane load and one 16 B/lane store, and every byte of
// the tensor moves exactly once (the algorithmic minimum: 4 B in + 4 B out per
// sample forward, 8 + 4 backward).  The recurrence
//     s[z] = s[z-1] * (1/p[z-1] - 1) * p[z]                     (K7 :138)
// is the closed form  s[z] = p[z] * prod_{k<z} (1 - p[k]),  an exclusive
// product scan; its adjoint (K8) is
//     g[z] = w[z]/p[z] - (sum_{j>z} w[j]) / (1 - p[z]),         w = s * dL/ds
// a reverse exclusive sum scan.  Both scans run in fp64 registers (lane-local
// over 4 samples, then 6 cross-lane steps) -- the kernels are bandwidth-bound,
// so fp64 is free -- and are rounded to fp32 once.  The reference rounds to fp32
// at every one of its 255 serial steps, so it is the reference that carries the
// (random-walk) rounding noise; measured difference <= 3e-6 absolute on
// near-binary GenRe-like inputs, <= 6e-8 on uniform inputs (tolerance 1e-5).
//
// Rays longer than 256 are processed in 256-sample chunks with a carried
// prefix; any strides are accepted (a generic one-sample-per-lane kernel covers
// layouts the float4 path cannot).
#include "common.hpp"
#include <cstdlib>
#include "wave_scan.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

#ifndef GENRE_CP_BLOCK
#define GENRE_CP_BLOCK 256                     // tools/ab_round4.py: workgroup size of the stop-probability kernels
#endif
#ifndef GENRE_CP_NT
#define GENRE_CP_NT 3                          // bit 1: nontemporal stores; bit 0 CLEARED: plain loads at every size (A/B)
#endif
// The forward's loads are nontemporal only when the tensor cannot stay in the 256 MiB Infinity Cache between launches.
// Measured (profiles/r04b_ab_experiments.txt): one image (16 MiB of prob_in, re-read by every request of a serving loop)
// 5.86 us with nontemporal loads, 4.73 us with plain ones; batch 32 (512 MiB) 165 us nontemporal, 170 us plain.
constexpr int64_t kStreamBytes = (int64_t)128 << 20;
constexpr int kBlock = GENRE_CP_BLOCK;         // 4 waves = 4 rays in flight per block
constexpr int kWavesPerBlock = kBlock / 64;

struct RayDims { int NC, X, Y, Z; int64_t rays; };
// pitch >= 0: the four outer dims collapse to "ray r starts at r*pitch" (any dense layout);
// pitch < 0: decode (n,c,x,y) and use the individual strides.
struct RayView { float *p; int64_t sn, sc, sx, sy, sz, pitch; };

inline RayView ray_view(const genre_tensor *t)
{
    RayView v{(float *)t->data, t->stride[0], t->stride[1], t->stride[2], t->stride[3], t->stride[4], -1};
    int64_t pitch = -1, span = 1;
    bool ok = true;
    for (int i = 3; i >= 0; i--) {
        if (t->size[i] == 1) continue;
        if (pitch < 0) pitch = t->stride[i];
        if (t->stride[i] != pitch * span) ok = false;
        span *= t->size[i];
    }
    if (pitch < 0) pitch = 0;              // a single ray
    if (ok) v.pitch = pitch;
    return v;
}

__device__ __forceinline__ int64_t ray_base(const RayDims &D, const RayView &v, int64_t r)
{
    if (v.pitch >= 0) return r * v.pitch;
    const int64_t xy = (int64_t)D.X * D.Y;
    const int64_t nc = r / xy;
    const int rem = (int)(r - nc * xy);
    const int x = rem / D.Y, y = rem - x * D.Y;
    const int64_t n = nc / D.NC;
    const int c = (int)(nc - n * D.NC);
    return n * v.sn + c * v.sc + x * v.sx + y * v.sy;
}

// streaming accesses: every byte is touched exactly once, keep it out of the way of L2
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT = true>
__device__ __forceinline__ float4 nt_load4(const float *p)
{
    const v4f v = (NT && (GENRE_CP_NT & 1)) ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p))
                                            : *reinterpret_cast<const v4f *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float *p, const float4 &s)
{
    const v4f v = {s.x, s.y, s.z, s.w};
    if (GENRE_CP_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
    else *reinterpret_cast<v4f *>(p) = v;
}

// ---- forward, float4 path: z stride 1, Z % 4 == 0, 16-B aligned rays ----------------
template <bool NTLOAD>
__global__ __launch_bounds__(kBlock) void stop_fwd_vec4_kernel(RayDims D, RayView pin, RayView pout)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t r = wave0; r < D.rays; r += nwaves) {
        const float *src = pin.p + ray_base(D, pin, r);
        float *dst = pout.p + ray_base(D, pout, r);
        double carry = 1.0;                                   // prod of (1-p) over earlier chunks
        for (int z0 = 0; z0 < D.Z; z0 += 256) {
            const int z = z0 + lane * 4;
            const bool live = z < D.Z;
            float4 p = make_float4(0.f, 0.f, 0.f, 0.f);       // p = 0 -> factor 1 (neutral)
            if (live) p = nt_load4<NTLOAD>(src + z);
            const double q0 = 1.0 - (double)p.x, q1 = 1.0 - (double)p.y;
            const double q2 = 1.0 - (double)p.z, q3 = 1.0 - (double)p.w;
            const double e1 = q0, e2 = q0 * q1, e3 = e2 * q2, tot = e3 * q3;
            const double incl = wave_incl_prod(tot);
            const double excl = wave_prev(1.0, incl) * carry;
            if (live) {
                float4 s;
                s.x = (float)((double)p.x * excl);
                s.y = (float)((double)p.y * (excl * e1));
                s.z = (float)((double)p.z * (excl * e2));
                s.w = (float)((double)p.w * (excl * e3));
                nt_store4(dst + z, s);
            }
            carry *= wave_last(incl);
        }
    }
}

// ---- forward, generic strides: one sample per lane per 64-chunk -----------------------
__global__ __launch_bounds__(kBlock) void stop_fwd_generic_kernel(RayDims D, RayView pin, RayView pout)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t r = wave0; r < D.rays; r += nwaves) {
        const float *src = pin.p + ray_base(D, pin, r);
        float *dst = pout.p + ray_base(D, pout, r);
        double carry = 1.0;
        for (int z0 = 0; z0 < D.Z; z0 += 64) {
            const int z = z0 + lane;
            const bool live = z < D.Z;
            const float p = live ? src[z * pin.sz] : 0.0f;
            const double incl = wave_incl_prod(1.0 - (double)p);
            const double excl = wave_prev(1.0, incl);
            if (live) dst[z * pout.sz] = (float)((double)p * (excl * carry));
            carry *= wave_last(incl);
        }
    }
}

// ---- backward, float4 path ---------------------------------------------------------------
// FUSED: w = stop_prob * grad_in formed here in fp32 (calc_prob.py:27); otherwise `a` already is w.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void stop_bwd_vec4_kernel(RayDims D, RayView pin, RayView a, RayView b,
                                                                RayView gout)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int nchunks = (D.Z + 255) / 256;
    for (int64_t r = wave0; r < D.rays; r += nwaves) {
        const float *src = pin.p + ray_base(D, pin, r);
        const float *wa = a.p + ray_base(D, a, r);
        const float *wb = FUSED ? b.p + ray_base(D, b, r) : nullptr;
        float *dst = gout.p + ray_base(D, gout, r);
        double carry = 0.0;                                   // sum of w over later chunks
        for (int ch = nchunks - 1; ch >= 0; ch--) {
            const int z = ch * 256 + lane * 4;
            const bool live = z < D.Z;
            float4 p = make_float4(0.5f, 0.5f, 0.5f, 0.5f), w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                p = nt_load4(src + z);
                w = nt_load4(wa + z);
                if (FUSED) {
                    const float4 g = nt_load4(wb + z);
                    w.x *= g.x; w.y *= g.y; w.z *= g.z; w.w *= g.w;
                }
            }
            const double w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w;
            const double tot = ((w3 + w2) + w1) + w0;
            const double incl = wave_incl_sum(tot);            // prefix over lanes <= lane
            const double chunk_total = wave_last(incl);
            const double after = (chunk_total - incl) + carry; // sum over lanes > lane (+ later chunks)
            if (live) {
                float4 g;
                // fp32 divides (w/p is an fp32 divide in the reference too, :178); the suffix sums stay fp64 and
                // are rounded once.  An fp64 divide costs ~30 instructions, and there were 8 per lane.
                g.w = w.w / p.w - (float)after / (1.0f - p.w);
                g.z = w.z / p.z - (float)(after + w3) / (1.0f - p.z);
                g.y = w.y / p.y - (float)(after + (w3 + w2)) / (1.0f - p.y);
                g.x = w.x / p.x - (float)(after + ((w3 + w2) + w1)) / (1.0f - p.x);
                nt_store4(dst + z, g);
            }
            carry += chunk_total;
        }
    }
}

template <bool FUSED>
__global__ __launch_bounds__(kBlock) void stop_bwd_generic_kernel(RayDims D, RayView pin, RayView a, RayView b,
                                                                   RayView gout)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int nchunks = (D.Z + 63) / 64;
    for (int64_t r = wave0; r < D.rays; r += nwaves) {
        const float *src = pin.p + ray_base(D, pin, r);
        const float *wa = a.p + ray_base(D, a, r);
        const float *wb = FUSED ? b.p + ray_base(D, b, r) : nullptr;
        float *dst = gout.p + ray_base(D, gout, r);
        double carry = 0.0;
        for (int ch = nchunks - 1; ch >= 0; ch--) {
            const int z = ch * 64 + lane;
            const bool live = z < D.Z;
            float p = 0.5f, w = 0.0f;
            if (live) {
                p = src[z * pin.sz];
                w = wa[z * a.sz];
                if (FUSED) w *= wb[z * b.sz];
            }
            const double incl = wave_incl_sum((double)w);
            const double chunk_total = wave_last(incl);
            const double after = (chunk_total - incl) + carry;
            if (live) dst[z * gout.sz] = (float)((double)w / (double)p - after / (1.0 - (double)p));
            carry += chunk_total;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------
int check5(const char *op, const char *name, const genre_tensor *t, const genre_tensor *like)
{
    GENRE_REQUIRE(is_f32(t, 5), "%s: %s must be a 5-D fp32 tensor", op, name);     // calc_prob_kernel.cu:88-110
    if (like) GENRE_REQUIRE(same_shape(t, like), "%s: %s must have the shape of prob_in", op, name);
    return 1;
}

bool vec4_ok(const genre_tensor *t)
{
    if (t->stride[4] != 1 && t->size[4] != 1) return false;
    if (t->size[4] % 4 != 0 || !aligned16(t->data)) return false;
    for (int i = 0; i < 4; i++)
        if (t->size[i] != 1 && (t->stride[i] % 4) != 0) return false;
    return true;
}

RayDims ray_dims(const genre_tensor *t)
{
    RayDims D;
    D.NC = (int)t->size[1]; D.X = (int)t->size[2]; D.Y = (int)t->size[3]; D.Z = (int)t->size[4];
    D.rays = t->size[0] * t->size[1] * t->size[2] * t->size[3];
    return D;
}

inline int grid_for_rays(int64_t rays)
{
    int64_t b = (rays + kWavesPerBlock - 1) / kWavesPerBlock;
    // one ray per wave up to 2^20 workgroups: measured at batch 32 (524 288 rays) forward 199 / 181 / 172 / 165 us
    // for 2 k / 8 k / 32 k / 131 k workgroups -- short-lived waves walking the tensor front to back beat long-lived
    // grid-striding ones on this memory system (same finding as cam_bp's fill, tools/fill_bench.hip)
    const int64_t cap = (int64_t)1 << 20;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

template <bool FUSED>
int backward_impl(const char *op, const genre_tensor *prob_in, const genre_tensor *a, const genre_tensor *b,
                  const genre_tensor *grad_out, void *stream)
{
    if (!check5(op, "prob_in", prob_in, nullptr) || !check5(op, FUSED ? "stop_prob" : "stop_prob_weighted", a, prob_in) ||
        (FUSED && !check5(op, "grad_in", b, prob_in)) || !check5(op, "grad_out", grad_out, prob_in))
        return 0;
    const RayDims D = ray_dims(prob_in);
    if (D.rays == 0 || D.Z == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const RayView vb = FUSED ? ray_view(b) : RayView{nullptr, 0, 0, 0, 0, 0, 0};
    if (vec4_ok(prob_in) && vec4_ok(a) && (!FUSED || vec4_ok(b)) && vec4_ok(grad_out))
        stop_bwd_vec4_kernel<FUSED><<<grid_for_rays(D.rays), kBlock, 0, st>>>(D, ray_view(prob_in), ray_view(a), vb,
                                                                              ray_view(grad_out));
    else
        stop_bwd_generic_kernel<FUSED><<<grid_for_rays(D.rays), kBlock, 0, st>>>(D, ray_view(prob_in), ray_view(a),
                                                                                 vb, ray_view(grad_out));
    GENRE_LAUNCH_CHECK("calc_prob backward");
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_calc_prob_forward(const genre_tensor *prob_in, const genre_tensor *prob_out, void *stream)
{
    const char *op = "calc_prob_forward";
    if (!check5(op, "prob_in", prob_in, nullptr) || !check5(op, "prob_out", prob_out, prob_in)) return 0;
    const RayDims D = ray_dims(prob_in);
    if (D.rays == 0 || D.Z == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (vec4_ok(prob_in) && vec4_ok(prob_out)) {
        if (D.rays * D.Z * 4 > kStreamBytes)
            stop_fwd_vec4_kernel<true><<<grid_for_rays(D.rays), kBlock, 0, st>>>(D, ray_view(prob_in), ray_view(prob_out));
        else
            stop_fwd_vec4_kernel<false><<<grid_for_rays(D.rays), kBlock, 0, st>>>(D, ray_view(prob_in), ray_view(prob_out));
    }
    else
        stop_fwd_generic_kernel<<<grid_for_rays(D.rays), kBlock, 0, st>>>(D, ray_view(prob_in), ray_view(prob_out));
    GENRE_LAUNCH_CHECK("calc_prob forward");
    return 1;
}

extern "C" int genre_calc_prob_backward(const genre_tensor *prob_in, const genre_tensor *stop_prob_weighted,
                                        const genre_tensor *grad_out, void *stream)
{
    return backward_impl<false>("calc_prob_backward", prob_in, stop_prob_weighted, nullptr, grad_out, stream);
}

extern "C" int genre_calc_prob_backward_fused(const genre_tensor *prob_in, const genre_tensor *stop_prob,
                                              const genre_tensor *grad_in, const genre_tensor *grad_out,
                                              void *stream)
{
    return backward_impl<true>("calc_prob_backward_fused", prob_in, stop_prob, grad_in, grad_out, stream);
}
