// cam_bp.hip -- depth-map / spherical-map  ->  voxel TDF back-projection for gfx950.
//
// DISPATCH of the camera / spherical FORWARD (forward_impl; which one a by-value call takes: genre_cam_forward_plan):
//   entry point                                   output layout                                   kernels
//   any camera entry (tensor or by-value camera)  dense NCXYZ, float4-aligned z rows, <= 65535    cam_brick_kernel<BYVAL, PIXELSCREEN>: ONE launch, LDS bricks,
//                                                 images (the reference's tensors)                deterministic; optional per-cell occupancy words (by value)
//   genre_back_projection_forward_const           anything else (image-minor volumes of the       fill1/fill2_vec4_kernel + cam_leader_kernel<H>: no atomics,
//   (camera by value, voxels project to <= 4 px)  batch-minor renderer, odd res)                  bit-identical to the serial reference; optional brick words,
//                                                                                                 optional sparse cnt
//   tensor cameras                                anything else                                   fill2 + scatter_tile_kernel<false> + normalise_tile_kernel<false>
//   spherical back-projection (K5)                any                                             fill2 + scatter_tile_kernel<true> + normalise_tile_kernel<true>
//   GENRE_CAMBP_MODE=scatter | gather | brick     (read once per process; tests pin each)         scatter: the three launches for dense volumes too;
//                                                                                                 gather: cam_gather_kernel (serial-order sums, opt-in)
// BACKWARD / mask: cam_backward_kernel (K4), surface_mask_kernel (K3), sph_backward_kernel (K6): one implementation each.
//
// Replaces toolbox/cam_bp/cam_bp/src/back_projection_kernel.cu of the reference
// (kernels K1-K6, wrappers :629-963).  Not a translation: the reference's
// pipeline per forward is  zero(cnt) + zero(tdf) + add 1/res (Python)  ->
// zero(cnt) again  ->  K1 scatter  ->  K2 full-volume divide with 5 div/mod per
// voxel and n-fastest (uncoalesced) indexing.  Here the camera forward onto a dense
// volume is ONE launch (cam_brick_kernel: a workgroup owns an 8x8x32 voxel brick,
// screens the depth pixels under the brick's image footprint, accumulates the points
// that land in it in an LDS tile and writes every voxel of the brick once; see there).
// Other layouts, and the spherical forward, are
//   (1) one float4 streaming fill of tdf and cnt (the only full-volume pass;
//       16 B/lane stores, the algorithmic minimum of 2 x 4 B per voxel),
//   (2) the scatter: a wave per 8x8 pixel tile, pixels of one voxel merged with
//       DPP moves, then hardware fp32 atomics at L2 (one pair per voxel and tile),
//   (3) a per-PIXEL normalise that touches only the voxels that were hit
//       (<= H*W of them) instead of re-streaming the whole volume.
// (3) needs to know whether a voxel still holds a raw sum or was already
// normalised by another pixel of the same voxel.  Raw sums are accumulated
// NEGATED (the first arriver, detected by cnt's atomic return value, also
// cancels the prefill), so "raw" == negative and "done" == non-negative, and
// the race between two pixels of one voxel is benign: both compute the same
// value from the same final (sum, cnt).  No scratch memory, no grid barrier.
//
// Index arithmetic (voxel index, centre, distance) is the reference's fp32
// sequence, compiled with contraction OFF so that the voxel a point lands in
// is bit-identical to the reference; `cnt` is therefore exact, and the
// per-point distances are bit-exact (only the order of the float atomics,
// which the reference does not define either, can differ).
#include "common.hpp"
#include <cstdlib>

#pragma clang fp contract(off)

namespace genre {
namespace {

constexpr int kBlock = 256;

// back_projection_kernel.cu:36-37,74-75 (FLOOR_I, VOXIND_TO_VOXC)
__device__ __forceinline__ int floor_i(float a) { return (a < 0.0f) ? (int)a - 1 : (int)a; }
__device__ __forceinline__ int vox_index(float g, int res) { return floor_i((g + 0.5f) * (float)res); }
// :195-196 (vec3d_norm): left-to-right sum, correctly rounded sqrt
__device__ __forceinline__ float norm3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }
// fp32 centre (:258-260) and the fp64-literal variant used by K3/K4/K6 (:336-338,:428-430,:596-598).  Dividing by a
// power-of-two resolution (the reference's 128) is an exact scaling: multiplying by 2^-k -- built from its exponent bits on
// the scalar unit -- gives the identical bits without the ~30-instruction correctly rounded division (three per point).
__device__ __forceinline__ bool pow2(int R) { return R > 0 && (R & (R - 1)) == 0; }
__device__ __forceinline__ float centre_f(int i, int R)
{
    const float n = (float)i + 0.5f;
    const float q = pow2(R) ? n * __int_as_float((127 - __builtin_ctz((unsigned)R)) << 23) : n / (float)R;
    return q - 0.5f;
}
__device__ __forceinline__ float centre_d(int i, int R)
{
    const double n = (double)(float)i + 0.5;
    const double q = pow2(R) ? n * __longlong_as_double((long long)(1023 - __builtin_ctz((unsigned)R)) << 52)
                             : n / (double)(float)R;
    return (float)(q - 0.5);
}

struct Dims { int N, NC, H, W, X, Y, Z; };

// Back-projected point of pixel (h,w): camera model of :231-242, or grid*d (:506-508).
template <bool SPH>
__device__ __forceinline__ bool pixel_point(const Dims &D, const View4 &depth, const View2 &camdist,
                                            const View2 &fl, const View5 &grid, int n, int c, int h, int w,
                                            float &d_raw, float &gx, float &gy, float &gz,
                                            float &u_h, float &u_w, float &f)
{
    d_raw = depth.p[n * depth.s0 + c * depth.s1 + h * depth.s2 + w * depth.s3];
    if (d_raw < 0.0f) return false;                                  // :225 / :501
    if (SPH) {
        const float *gp = grid.p + n * grid.s0 + c * grid.s1 + h * grid.s2 + w * grid.s3;
        gx = gp[0] * d_raw; gy = gp[grid.s4] * d_raw; gz = gp[2 * grid.s4] * d_raw;
        u_h = u_w = f = 0.0f;
    } else {
        f = fl.p[n * fl.s0 + c * fl.s1];
        float cam_dist = camdist.p[n * camdist.s0 + c * camdist.s1];
        u_h = (float)h - ((float)D.H - 1.0f) / 2.0f;
        u_w = (float)w - ((float)D.W - 1.0f) / 2.0f;
        float cos_theta = f / norm3(u_h, u_w, f);
        float d = d_raw * cos_theta;
        gy = -d * u_w / f;
        gz = -d * u_h / f;
        gx = d - cam_dist;
    }
    return true;
}

__device__ __forceinline__ bool in_grid(const Dims &D, int ix, int iy, int iz)
{
    return ix >= 0 && ix < D.X && iy >= 0 && iy < D.Y && iz >= 0 && iz < D.Z;
}

__device__ __forceinline__ void decode_pixel(const Dims &D, int64_t idx, int &n, int &c, int &h, int &w)
{
    w = (int)(idx % D.W); idx /= D.W;
    h = (int)(idx % D.H); idx /= D.H;
    c = (int)(idx % D.NC);
    n = (int)(idx / D.NC);
}

// ---- (1) fill -----------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fill2_vec4_kernel(float4 *__restrict__ a, float va,
                                                             float4 *__restrict__ b, float vb, int64_t n4)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f fa = {va, va, va, va}, fb = {vb, vb, vb, vb};
    v4f *pa = reinterpret_cast<v4f *>(a), *pb = reinterpret_cast<v4f *>(b);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        __builtin_nontemporal_store(fa, &pa[i]);       // streaming: nothing re-reads a full line soon
        __builtin_nontemporal_store(fb, &pb[i]);
    }
}

__global__ __launch_bounds__(kBlock) void fill1_vec4_kernel(float4 *__restrict__ a, float va, int64_t n4)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f fa = {va, va, va, va};
    v4f *pa = reinterpret_cast<v4f *>(a);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock)
        __builtin_nontemporal_store(fa, &pa[i]);
}

__global__ __launch_bounds__(kBlock) void fill2_strided_kernel(Dims D, View5 a, float va, View5 b, float vb)
{
    const int64_t total = (int64_t)D.N * D.NC * D.X * D.Y * D.Z;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        int64_t r = i;
        int z = (int)(r % D.Z); r /= D.Z;
        int y = (int)(r % D.Y); r /= D.Y;
        int x = (int)(r % D.X); r /= D.X;
        int c = (int)(r % D.NC);
        int n = (int)(r / D.NC);
        a.p[n * a.s0 + c * a.s1 + x * a.s2 + y * a.s3 + z * a.s4] = va;
        b.p[n * b.s0 + c * b.s1 + x * b.s2 + y * b.s3 + z * b.s4] = vb;
    }
}

// ---- (2) scatter and (3) normalise: neighbouring pixels are combined before they touch memory ---------
// The scatter is bound by the rate of global float atomics (~15-18 G/s measured): a 256^2 image puts ~2x2
// pixels into every surface voxel, i.e. four atomic pairs per voxel.  Here a wave owns an 8x8 pixel tile
// (lane = y*8 + x) and pixels that landed in the same voxel are merged with DPP moves first -- (x,x+1) pairs
// at even x, then at odd x, then (y,y+1) pairs at even y (all inside one 16-lane DPP row) -- so a voxel usually
// costs ONE atomic pair.  The survivor ("leader") carries the count and the sum of the absorbed distances.
// Single-hit voxels are untouched by this (bit-exact as before); for multi-hit voxels only the summation
// order changes, which the reference's atomics do not define either.
constexpr int kRowShl1 = 0x101, kRowShl8 = 0x108, kRowShr1c = 0x111, kRowShr8 = 0x118;

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}

// one merge step: lanes selected by `takes` absorb the lane SHL positions above them when both hold the same key;
// the absorbed lane (selected by `gives`, looking SHR positions down) retires
template <int SHL, int SHR>
__device__ __forceinline__ void merge_step(bool takes, bool gives, int &key, float &rest, float &n, float dist)
{
    const int up_key = dpp_i<SHL>(key);
    const float up_sum = dpp_f<SHL>(dist + rest), up_n = dpp_f<SHL>(n);
    const int down_key = dpp_i<SHR>(key);
    const bool absorb = takes && key >= 0 && up_key == key;
    const bool retire = gives && key >= 0 && down_key == key;
    if (absorb) { rest += up_sum; n += up_n; }
    if (retire) key = -1;
}

// pixel (tile, lane): image, row, column; false outside the image.  `tile` is wave-uniform and fits 32 bits
// (host-checked), so the decode runs on the scalar unit.
__device__ __forceinline__ bool tile_coords(const Dims &D, int tile, int lane, int &n, int &c, int &h, int &w)
{
    const int tw = (D.W + 7) >> 3, th = (D.H + 7) >> 3;
    const int tx = tile % tw; tile /= tw;
    const int ty = tile % th; tile /= th;
    c = tile % D.NC;
    n = tile / D.NC;
    h = ty * 8 + (lane >> 3); w = tx * 8 + (lane & 7);
    return h < D.H && w < D.W;
}

// element offset of voxel (ix,iy,iz) inside one image: 32-bit (the host checks that an image's volume spans
// fewer than 2^31 elements), so the per-lane part of the address is three 32-bit multiply-adds
__device__ __forceinline__ int vox_off(const View5 &v, int ix, int iy, int iz)
{
    return ix * (int)v.s2 + iy * (int)v.s3 + iz * (int)v.s4;
}

// voxel of a pixel whose depth (and, spherical path, direction) has been fetched: key = linear voxel index
// inside the image, or -1.  Same arithmetic as pixel_point.
template <bool SPH>
__device__ __forceinline__ int pixel_voxel(const Dims &D, bool valid, float d_raw, float g0, float g1, float g2, float f,
                                           float cam_dist, int h, int w, int &ix, int &iy, int &iz, float &dist)
{
    dist = 0.f; ix = iy = iz = 0;
    if (!valid || d_raw < 0.0f) return -1;                           // :225 / :501
    float gx, gy, gz;
    if (SPH) {
        gx = g0 * d_raw; gy = g1 * d_raw; gz = g2 * d_raw;           // :506-508
    } else {
        const float u_h = (float)h - ((float)D.H - 1.0f) / 2.0f;    // :231-242
        const float u_w = (float)w - ((float)D.W - 1.0f) / 2.0f;
        const float cos_theta = f / norm3(u_h, u_w, f);
        const float d = d_raw * cos_theta;
        gy = -d * u_w / f;
        gz = -d * u_h / f;
        gx = d - cam_dist;
    }
    ix = vox_index(gx, D.X); iy = vox_index(gy, D.Y); iz = vox_index(gz, D.Z);
    if (!in_grid(D, ix, iy, iz)) return -1;                          // :252
    dist = norm3(gx - centre_f(ix, D.X), gy - centre_f(iy, D.Y), gz - centre_f(iz, D.Z));
    return (ix * D.Y + iy) * D.Z + iz;
}

__device__ __forceinline__ void merge_tile(int lane, int &key, float &rest, float &n, float dist)
{
    const int x = lane & 7, y = lane >> 3;
    merge_step<kRowShl1, kRowShr1c>((x & 1) == 0, (x & 1) == 1, key, rest, n, dist);
    merge_step<kRowShl1, kRowShr1c>((x & 1) == 1 && x < 7, (x & 1) == 0 && x > 0, key, rest, n, dist);
    merge_step<kRowShl8, kRowShr8>((y & 1) == 0, (y & 1) == 1, key, rest, n, dist);
}

// A wave walks its tiles two at a time with the loads of both in flight (depth first, then -- normalise --
// the sum / count gathers): at batch 32 a wave sees four tiles, and one dependent latency chain per tile was
// most of the kernel.
constexpr int kTilesInFlight = 2;

struct TileLanes {
    int n[kTilesInFlight], c[kTilesInFlight], ix[kTilesInFlight], iy[kTilesInFlight], iz[kTilesInFlight], key[kTilesInFlight];
    float dist[kTilesInFlight], rest[kTilesInFlight], num[kTilesInFlight];
};

template <bool SPH>
__device__ __forceinline__ void load_tiles(const Dims &D, const View4 &depth, const View2 &camdist, const View2 &fl,
                                           const View5 &grid, int tile0, int stride, int tiles, int lane, TileLanes &T)
{
    int h[kTilesInFlight], w[kTilesInFlight];
    bool valid[kTilesInFlight];
    float d_raw[kTilesInFlight], g0[kTilesInFlight], g1[kTilesInFlight], g2[kTilesInFlight], f[kTilesInFlight],
        cd[kTilesInFlight];
#pragma unroll
    for (int u = 0; u < kTilesInFlight; u++) {
        const int tile = tile0 + u * stride;
        const bool in_image = tile_coords(D, tile < tiles ? tile : 0, lane, T.n[u], T.c[u], h[u], w[u]);
        valid[u] = tile < tiles && in_image;
        d_raw[u] = -1.f; g0[u] = g1[u] = g2[u] = f[u] = cd[u] = 0.f;
        if (valid[u]) {
            d_raw[u] = depth.p[(T.n[u] * depth.s0 + T.c[u] * depth.s1) + (h[u] * (int)depth.s2 + w[u] * (int)depth.s3)];
            if (SPH) {
                const float *gp = grid.p + (T.n[u] * grid.s0 + T.c[u] * grid.s1) + (h[u] * (int)grid.s2 + w[u] * (int)grid.s3);
                g0[u] = gp[0]; g1[u] = gp[grid.s4]; g2[u] = gp[2 * grid.s4];
            } else {
                f[u] = fl.p[T.n[u] * fl.s0 + T.c[u] * fl.s1];
                cd[u] = camdist.p[T.n[u] * camdist.s0 + T.c[u] * camdist.s1];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kTilesInFlight; u++) {
        T.key[u] = pixel_voxel<SPH>(D, valid[u], d_raw[u], g0[u], g1[u], g2[u], f[u], cd[u], h[u], w[u], T.ix[u], T.iy[u],
                                    T.iz[u], T.dist[u]);
        T.rest[u] = 0.f; T.num[u] = 1.f;
        merge_tile(lane, T.key[u], T.rest[u], T.num[u], T.dist[u]);
    }
}

template <bool SPH>
__global__ __launch_bounds__(kBlock) void scatter_tile_kernel(Dims D, View4 depth, View2 camdist, View2 fl, View5 grid,
                                                               View5 vox, View5 cnt, float empty_val, float fill_val)
{
    const int lane = threadIdx.x & 63;
    const int tiles = D.N * D.NC * ((D.H + 7) >> 3) * ((D.W + 7) >> 3);
    const int stride = gridDim.x * (kBlock / 64);
    for (int tile = blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); tile < tiles;
         tile += kTilesInFlight * stride) {
        TileLanes T;
        load_tiles<SPH>(D, depth, camdist, fl, grid, tile, stride, tiles, lane, T);
#pragma unroll
        for (int u = 0; u < kTilesInFlight; u++) {
            if (T.key[u] < 0) continue;
            float *pc = cnt.p + (T.n[u] * cnt.s0 + T.c[u] * cnt.s1) + vox_off(cnt, T.ix[u], T.iy[u], T.iz[u]);
            float *pv = vox.p + (T.n[u] * vox.s0 + T.c[u] * vox.s1) + vox_off(vox, T.ix[u], T.iy[u], T.iz[u]);
            // Negated accumulation (see file header).  The reference starts every sum at the prefill e = 1/res
            // (0 on the spherical path) and subtracts it again in K2 (:304): the first point contributes
            // t = fl(e + dist) - e (exact).  The first arriver -- detected by the count's atomic return value --
            // reproduces that rounding and cancels whatever the fill pass wrote; later arrivers just add, as the
            // reference's atomics do.  When there is nothing to cancel (spherical path, and the camera path with
            // the shift folded in) nobody needs to know who is first: both atomics are fire-and-forget.
            const float t = (T.dist[u] + empty_val) - empty_val;
            if (fill_val == 0.0f) {
                unsafeAtomicAdd(pc, T.num[u]);                           // :274
                unsafeAtomicAdd(pv, -(t + T.rest[u]));                   // :273
            } else {
                const float old = unsafeAtomicAdd(pc, T.num[u]);
                unsafeAtomicAdd(pv, (old == 0.0f) ? -((t + fill_val) + T.rest[u]) : -(T.dist[u] + T.rest[u]));
            }
        }
    }
}

template <bool SPH>
__global__ __launch_bounds__(kBlock) void normalise_tile_kernel(Dims D, View4 depth, View2 camdist, View2 fl, View5 grid,
                                                                 View5 vox, View5 cnt, float post_scale, float post_bias,
                                                                 int post_mode)
{
    const int lane = threadIdx.x & 63;
    const int tiles = D.N * D.NC * ((D.H + 7) >> 3) * ((D.W + 7) >> 3);
    const int stride = gridDim.x * (kBlock / 64);
    for (int tile = blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); tile < tiles;
         tile += kTilesInFlight * stride) {
        TileLanes T;
        load_tiles<SPH>(D, depth, camdist, fl, grid, tile, stride, tiles, lane, T);   // one lane per voxel and tile works
        float s[kTilesInFlight], k[kTilesInFlight];
        float *pv[kTilesInFlight];
#pragma unroll
        for (int u = 0; u < kTilesInFlight; u++) {
            pv[u] = vox.p + (T.n[u] * vox.s0 + T.c[u] * vox.s1) + vox_off(vox, T.ix[u], T.iy[u], T.iz[u]);
            s[u] = 0.f; k[u] = 1.f;
            if (T.key[u] >= 0) {
                s[u] = *pv[u];
                k[u] = cnt.p[(T.n[u] * cnt.s0 + T.c[u] * cnt.s1) + vox_off(cnt, T.ix[u], T.iy[u], T.iz[u])];
            }
        }
#pragma unroll
        for (int u = 0; u < kTilesInFlight; u++) {
            // still a raw (negated) sum?  `<=`: a point that sits exactly on its voxel's centre leaves a raw sum of
            // -0.0; finished values are > 0 in the shifted modes and recomputing a finished 0 is idempotent in identity mode
            if (T.key[u] >= 0 && s[u] <= 0.0f) {
                // :304 (mean distance); post = identity (scale 1, bias 0), the camera layer's shift 1 - res*tdf
                // (mode 0 with scale -res, bias 1), or GenRe's spherical glue (-tdf + 1/res)*res (mode 1,
                // genre_full_model.py:141: post_bias holds 1/res, post_scale holds res)
                const float mean = (0.0f - s[u]) / k[u];
                *pv[u] = post_mode == 1 ? (-mean + post_bias) * post_scale : post_bias + post_scale * mean;
            }
        }
    }
}

// ---- camera forward, single-launch GATHER formulation --------------------------------------------
// Fill + scatter + normalise above need three dependent launches (the scatter's atomics must see the
// fill, the normalise must see all atomics): ~10 us of kernels and boundaries for ONE image whose
// algorithmic traffic is 17 MB (2.1 us at 8 TB/s), and float-atomic summation order is undefined.
// This kernel inverts the mapping: every voxel looks up the few pixels whose rays cross it (a voxel
// projects to a ~2x2 pixel window), re-evaluates the reference's per-pixel arithmetic for them and keeps
// those that land in itself.  Exactly the same set of points per voxel as the scatter (every candidate
// is checked with the bit-identical index maths), summed in row-major pixel order -- the order of the
// reference's serial index loop -- so tdf is DETERMINISTIC and bit-identical to a serial CPU evaluation of the reference, and
// every output element is written exactly once with no atomics, no prefill pass and no second launch.
// A workgroup owns an 8x8x32 voxel brick; one cooperative min/max scan of the depth pixels under the
// brick's footprint rejects the ~90 % of bricks that no point can reach, which then cost only their
// coalesced stores.
constexpr int kGX = 16, kGY = 8, kGZ = 64;         // brick owned by a workgroup (32 voxels per thread)
constexpr int kFoot = 4096;                        // depth pixels of a brick footprint staged in LDS

struct Win { int h0, h1, w0, w1; float amax2; };   // inclusive pixel window, max(u_h^2 + u_w^2) over it

// pixel window that can see the world box [xlo,xhi]x[ylo,yhi]x[zlo,zhi] (conservative, `margin` pixels;
// the 1-ulp fast reciprocal is far inside the margin)
__device__ __forceinline__ Win project_box(const Dims &D, float xlo, float xhi, float ylo, float yhi, float zlo,
                                           float zhi, float cam_dist, float f, float margin)
{
    Win r;
    const float Xn = xlo + cam_dist, Xf = xhi + cam_dist;
    const float ch = ((float)D.H - 1.0f) / 2.0f, cw = ((float)D.W - 1.0f) / 2.0f;
    if (!(Xn > 1e-3f) || !(f > 0.0f)) {               // camera inside / behind the box: everything is a candidate
        r.h0 = 0; r.h1 = D.H - 1; r.w0 = 0; r.w1 = D.W - 1;
        r.amax2 = ch * ch + cw * cw + 1.0f;
        return r;
    }
    const float a = f * __frcp_rn(Xn), b = f * __frcp_rn(Xf);   // u_w = -y*f/X, u_h = -z*f/X   (:240-241 inverted)
    const float w_a = -ylo * a, w_b = -ylo * b, w_c = -yhi * a, w_d = -yhi * b;
    const float h_a = -zlo * a, h_b = -zlo * b, h_c = -zhi * a, h_d = -zhi * b;
    const float uw_lo = fminf(fminf(w_a, w_b), fminf(w_c, w_d)), uw_hi = fmaxf(fmaxf(w_a, w_b), fmaxf(w_c, w_d));
    const float uh_lo = fminf(fminf(h_a, h_b), fminf(h_c, h_d)), uh_hi = fmaxf(fmaxf(h_a, h_b), fmaxf(h_c, h_d));
    const float wl = ceilf(uw_lo + cw - margin), wh = floorf(uw_hi + cw + margin);
    const float hl = ceilf(uh_lo + ch - margin), hh = floorf(uh_hi + ch + margin);
    r.w0 = (int)fmaxf(wl, 0.0f); r.w1 = (int)fminf(wh, (float)(D.W - 1));
    r.h0 = (int)fmaxf(hl, 0.0f); r.h1 = (int)fminf(hh, (float)(D.H - 1));
    const float mw = fmaxf(fabsf(uw_lo), fabsf(uw_hi)) + margin, mh = fmaxf(fabsf(uh_lo), fabsf(uh_hi)) + margin;
    r.amax2 = mw * mw + mh * mh;
    return r;
}

struct Band { float dmin, dmax; int any_zero; };

// can any pixel of the window put a point into the slab x in [xlo, xhi]?  plane depth of a pixel's point is
// d * cos(theta) in [d * cmin, d]  (:235-237)
__device__ __forceinline__ bool slab_live(const Band &b, const Win &w, float xlo, float xhi, float cam_dist, float f,
                                          float &band_lo, float &band_hi, bool &special)
{
    const float eps = 1e-4f;
    const float cmin = f / sqrtf(f * f + w.amax2);
    band_lo = b.dmin * cmin - eps; band_hi = b.dmax + eps;
    const bool exotic = !(f > 0.0f) || !(xlo + cam_dist > 1e-3f);        // no shortcut is safe
    const bool zero_hits = b.any_zero && xlo + cam_dist - eps <= 0.0f && xhi + cam_dist + eps >= 0.0f;
    special = exotic || zero_hits;
    return special || ((b.dmax > 0.0f) && band_hi >= xlo + cam_dist && band_lo <= xhi + cam_dist);
}

__global__ __launch_bounds__(kBlock) void cam_gather_kernel(Dims D, View4 depth, View2 camdist, View2 fl, View5 vox,
                                                             View5 cnt, float prefill, float bias, float post_scale,
                                                             float post_bias, float fill_val, int vec_ok)
{
    __shared__ float s_depth[kFoot];
    __shared__ float s_min[kBlock / 64], s_max[kBlock / 64];
    __shared__ int s_any[kBlock / 64];
    const int nbz = (D.Z + kGZ - 1) / kGZ, nby = (D.Y + kGY - 1) / kGY;
    const int bz = blockIdx.x % nbz, by = (blockIdx.x / nbz) % nby, bx = blockIdx.x / (nbz * nby);
    const int img = blockIdx.y, n = img / D.NC, c = img % D.NC;
    const float f = fl.p[n * fl.s0 + c * fl.s1];
    const float cam_dist = camdist.p[n * camdist.s0 + c * camdist.s1];
    const float *dimg = depth.p + n * depth.s0 + c * depth.s1;
    float *vimg = vox.p + n * vox.s0 + c * vox.s1, *cimg = cnt.p + n * cnt.s0 + c * cnt.s1;
    const int x0 = bx * kGX, y0 = by * kGY, z0 = bz * kGZ;
    const int x1 = min(x0 + kGX, D.X), y1 = min(y0 + kGY, D.Y), z1 = min(z0 + kGZ, D.Z);
    const float rX = 1.0f / (float)D.X, rY = 1.0f / (float)D.Y, rZ = 1.0f / (float)D.Z;   // box faces: +-1 ulp is
    // ---- brick-level rejection: depth range under the brick's footprint (staged in LDS) ----  inside the margins
    const float bxlo = (float)x0 * rX - 0.5f, bxhi = (float)x1 * rX - 0.5f;
    const Win bw = project_box(D, bxlo, bxhi, (float)y0 * rY - 0.5f, (float)y1 * rY - 0.5f, (float)z0 * rZ - 0.5f,
                               (float)z1 * rZ - 0.5f, cam_dist, f, 1.0f);
    const int bww = bw.w1 - bw.w0 + 1, bwh = bw.h1 - bw.h0 + 1;
    const bool staged = bww > 0 && bwh > 0 && bww * bwh <= kFoot;
    // one pass over the footprint: stage it in LDS and reduce min / max of the positive depths.  Footprints
    // are tall and narrow (~20 x 130 px), so threads are laid out 32 wide x 8 high.
    Band bb{3.0e38f, 0.0f, 0};
    if (bww > 0 && bwh > 0) {
        // 16 pixel loads in flight per thread (a plain loop waits for each load: one exposed latency per pixel)
        constexpr int kDeep = 16;
        const int area = bww * bwh;
        for (int t0 = threadIdx.x; t0 < area; t0 += kDeep * kBlock) {
            float dv[kDeep];
#pragma unroll
            for (int u = 0; u < kDeep; u++) {
                const int t = t0 + u * kBlock;
                const int r = t / bww, q = t - r * bww;
                dv[u] = t < area ? dimg[(bw.h0 + r) * depth.s2 + (bw.w0 + q) * depth.s3] : -1.0f;
            }
#pragma unroll
            for (int u = 0; u < kDeep; u++) {
                const int t = t0 + u * kBlock;
                if (t >= area) continue;
                const float d = dv[u];
                if (staged) s_depth[t] = d;
                if (d > 0.0f) { bb.dmin = fminf(bb.dmin, d); bb.dmax = fmaxf(bb.dmax, d); }
                else if (!(d < 0.0f)) bb.any_zero = 1;  // d == 0 (or NaN): lands at x = -cam_dist
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bb.dmin = fminf(bb.dmin, __shfl_xor(bb.dmin, o, 64));
        bb.dmax = fmaxf(bb.dmax, __shfl_xor(bb.dmax, o, 64));
        bb.any_zero |= __shfl_xor(bb.any_zero, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = bb.dmin; s_max[threadIdx.x >> 6] = bb.dmax; s_any[threadIdx.x >> 6] = bb.any_zero; }
    __syncthreads();                                    // also publishes s_depth
#pragma unroll
    for (int i = 0; i < kBlock / 64; i++) { bb.dmin = fminf(bb.dmin, s_min[i]); bb.dmax = fmaxf(bb.dmax, s_max[i]); bb.any_zero |= s_any[i]; }
    float blo, bhi;
    bool bspecial;
    if (!slab_live(bb, bw, bxlo, bxhi, cam_dist, f, blo, bhi, bspecial)) {
        // ---- dead brick: nothing can land here; stream the fill values (float4 when the layout allows) ------
        if (vec_ok && z1 - z0 == kGZ) {
            const float4 fv = make_float4(fill_val, fill_val, fill_val, fill_val), zv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int z4 = (threadIdx.x & 15) * 4;
            for (int xy = threadIdx.x >> 4; xy < kGX * kGY; xy += kBlock / 16) {
                const int ix = x0 + xy / kGY, iy = y0 + xy % kGY;
                if (ix >= D.X || iy >= D.Y) continue;
                const int64_t o = ix * vox.s2 + iy * vox.s3 + (z0 + z4);
                *reinterpret_cast<float4 *>(vimg + o) = fv;
                *reinterpret_cast<float4 *>(cimg + ix * cnt.s2 + iy * cnt.s3 + (z0 + z4)) = zv;
            }
        } else {
            const int iz = z0 + (threadIdx.x & (kGZ - 1));
            for (int xy = threadIdx.x / kGZ; xy < kGX * kGY; xy += kBlock / kGZ) {
                const int ix = x0 + xy / kGY, iy = y0 + xy % kGY;
                if (ix >= D.X || iy >= D.Y || iz >= D.Z) continue;
                vimg[ix * vox.s2 + iy * vox.s3 + iz * vox.s4] = fill_val;
                cimg[ix * cnt.s2 + iy * cnt.s3 + iz * cnt.s4] = 0.0f;
            }
        }
        return;
    }
    // ---- live brick: one thread per (iy, iz) COLUMN of kGX voxels along the viewing axis ----------------------
    // The 8 voxels of a column see (almost) the same pixels, so each candidate pixel's point is evaluated
    // ONCE with the reference arithmetic and credited to whichever voxel of the column it lands in; pixels
    // are visited in row-major order, so every voxel still sums its points in the reference's serial order.
    for (int col = threadIdx.x; col < kGY * kGZ; col += kBlock) {
        const int iz = z0 + (col & (kGZ - 1)), iy = y0 + col / kGZ;
        if (iy >= D.Y || iz >= D.Z) continue;
        float sum[kGX], k[kGX];
#pragma unroll
        for (int i = 0; i < kGX; i++) { sum[i] = prefill; k[i] = 0.0f; }   // cam_back_projection.py:23-24
        const Win cw = project_box(D, bxlo - 1e-6f, bxhi + 1e-6f, (float)iy * rY - 0.5f - 1e-6f,
                                   (float)(iy + 1) * rY - 0.5f + 1e-6f, (float)iz * rZ - 0.5f - 1e-6f,
                                   (float)(iz + 1) * rZ - 0.5f + 1e-6f, cam_dist, f, 0.05f);
        const float cy = centre_f(iy, D.Y), cz = centre_f(iz, D.Z);
        const float xlo_t = bxlo - 1e-5f, xhi_t = bxhi + 1e-5f;
        for (int h = cw.h0; h <= cw.h1; h++) {
            const float u_h = (float)h - ((float)D.H - 1.0f) / 2.0f;
            for (int w = cw.w0; w <= cw.w1; w++) {
                const bool in_tile = staged && h >= bw.h0 && h <= bw.h1 && w >= bw.w0 && w <= bw.w1;
                const float d_raw = in_tile ? s_depth[(h - bw.h0) * bww + (w - bw.w0)] : dimg[h * depth.s2 + w * depth.s3];
                if (d_raw < 0.0f) continue;                              // :225
                const float u_w = (float)w - ((float)D.W - 1.0f) / 2.0f;
                // cheap plane-depth test first (1-ulp rsqrt, generous margin); exact maths only inside the brick
                const float xp = d_raw * f * __frsqrt_rn(u_h * u_h + u_w * u_w + f * f) - cam_dist;
                if (xp < xlo_t || xp > xhi_t) continue;
                const float cos_theta = f / norm3(u_h, u_w, f);          // :235
                const float d = d_raw * cos_theta;
                const float gy = -d * u_w / f, gz = -d * u_h / f, gx = d - cam_dist;   // :240-242
                if (vox_index(gy, D.Y) != iy || vox_index(gz, D.Z) != iz) continue;
                const int lx = vox_index(gx, D.X) - x0;
                if (lx < 0 || lx >= x1 - x0) continue;
                const float dist = norm3(gx - centre_f(x0 + lx, D.X), gy - cy, gz - cz);   // :266
#pragma unroll
                for (int i = 0; i < kGX; i++)
                    if (lx == i) { sum[i] = sum[i] + dist; k[i] = k[i] + 1.0f; }          // :273-274, serial order
            }
        }
#pragma unroll
        for (int i = 0; i < kGX; i++) {
            const int ix = x0 + i;
            if (ix >= D.X) break;
            const float out = k[i] > 0.0f ? post_bias + post_scale * ((sum[i] - bias) / k[i]) : fill_val;   // :304
            vimg[ix * vox.s2 + iy * vox.s3 + iz * vox.s4] = out;
            cimg[ix * cnt.s2 + iy * cnt.s3 + iz * cnt.s4] = k[i];
        }
    }
}

// ---- camera forward, single-launch BRICK formulation (the default for dense volumes) ------------------
// The three launches above are bound by their boundaries at batch 1 (17 MB of traffic, ~12 us) and by the float-atomic
// rate at batch 32.  Here a workgroup owns an 8x8x32 voxel brick (1024 workgroups per 128^3 image: four per CU, so the
// 16 MB of stores have the memory parallelism the 256-workgroup gather kernel lacks) and does all three phases for it:
//   (a) every thread fetches its share of the depth pixels under the brick's footprint (~23 x 71 px: <= 8 loads per
//       thread, all in flight together) -- the footprint's depth range (DPP wave reduction + one barrier) rejects the
//       ~77 % of bricks no point can reach, which then cost only their float4 stores;
//   (b) live bricks: every footprint pixel is evaluated ONCE, by the thread that fetched it -- a cheap plane-depth
//       screen, then the reference's arithmetic (pixel_voxel); a point that lands inside the brick is added to the
//       brick's LDS tile -- ds_add_f64 for the distance (8.7 clk per wave-instruction on gfx950 against 193 for
//       ds_add_f32, tools/bm_tile_bench.hip), ds_add_u32 for the count;
//   (c) the tile is normalised (K2, :291-305) and written with one float4 per thread and array.
// A voxel hit by one point is bit-exact: its double sum IS that distance, and fl(fl(prefill + dist) - bias) is the
// reference's sequence.  Voxels hit more than once add their (exact) distances in double instead of in the
// reference's undefined fp32 atomic order and divide by rcp(cnt): <= 2 ulp of 1/res from any of its orders, and
// deterministic.  tests/test_cam_brick_screens.py checks on the host that the three screens (footprint, depth range,
// plane depth) never drop a pixel whose point the reference puts into the brick.  Measured variants: forward_impl.

// wave-wide min / max / or with DPP (row shifts + row broadcasts, as wave_scan.hpp): the total lands in lane 63 and
// is broadcast through a scalar register -- ~20 vector instructions instead of 18 dependent ds_bpermute round trips
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int identity, int v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
template <typename F>
__device__ __forceinline__ int wave_reduce_bits(int v, int identity, F op)
{
    v = op(v, dpp_keep<0x111, 0xf>(identity, v));       // row_shr:1
    v = op(v, dpp_keep<0x112, 0xf>(identity, v));       // row_shr:2
    v = op(v, dpp_keep<0x114, 0xf>(identity, v));       // row_shr:4
    v = op(v, dpp_keep<0x118, 0xf>(identity, v));       // row_shr:8
    v = op(v, dpp_keep<0x142, 0xa>(identity, v));       // row_bcast:15 into rows 1, 3
    v = op(v, dpp_keep<0x143, 0xc>(identity, v));       // row_bcast:31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int kQX = 8, kQY = 8, kQZ = 32;
constexpr int kQVox = kQX * kQY * kQZ;
constexpr int kQDeep = 8;                          // footprint pixels a thread keeps in registers

// t = r * d + q, 0 <= q < d: a float multiply and one correction step instead of an integer division (the quotient
// estimate is within one of the truth for t < 2^22: both roundings together stay below half a unit)
__device__ __forceinline__ void divmod_px(int t, int d, float inv_d, int &r, int &q)
{
    if (t >= (1 << 22)) { r = t / d; q = t - r * d; return; }
    r = (int)((float)t * inv_d);
    q = t - r * d;
    if (q < 0) { r--; q += d; }
    else if (q >= d) { r++; q -= d; }
}

// BYVAL: focal length and camera distance are kernel arguments (one value for every image -- what
// Camera_back_projection_layer fills its [N,1] tensors with, camera_backprojection_module.py:16-21) instead of two loads
// in front of the footprint: one dependent memory round trip less before a workgroup knows its pixels.
// PIXELSCREEN (small batches): no block-wide depth-range screen.  The footprint's depth range costs a wave reduction and a
// barrier BEHIND the pixel loads, on the critical path of every brick, live or not; here the tile is cleared and the barrier
// passed while the loads are in flight, every pixel goes through the per-pixel plane-depth test the live bricks apply
// anyway (a dozen vector instructions), and "live" is whether any pixel landed -- one barrier behind the loads instead of
// two and a reduction.  Same output bit for bit: a brick the range screen calls dead has no landing pixel
// (tests/test_cam_brick_screens.py), and a live brick without hits writes the fill values from its empty tile.
// Measured (tools/ab_round4.py, profiles/r04b_ab_experiments.txt; one MI355X, HIP-graph replay at batch 1, events at batch 32):
//   variant                                   batch 1           batch 32
//   range screen, plain stores (round 3)      9.63 us           150.1 us
//   pixel screen                              9.00 us           144.5 us
//   range screen, nontemporal stores          8.13 us           143.9 us
//   pixel screen, nontemporal stores          7.49 us           143.7 us      <- the build default
// (outputs bit-identical in all four).  The 16 MB of the two volumes are written once and read by another kernel much
// later: kept out of L2 they do not have to be written back at the end of the kernel, where nothing overlaps it.
#ifndef GENRE_CAMQ_PIXELSCREEN_MAXN
#define GENRE_CAMQ_PIXELSCREEN_MAXN 0x7fffffff     // images per launch up to which the PIXELSCREEN instantiation is used
#endif
#ifndef GENRE_CAMQ_NT
#define GENRE_CAMQ_NT 1                            // nontemporal stores of the brick (0: plain stores, for the A/B)
#endif
template <typename T>
__device__ __forceinline__ void camq_store4(float *p, const T &v)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ q = {v.x, v.y, v.z, v.w};
    if (GENRE_CAMQ_NT) __builtin_nontemporal_store(q, reinterpret_cast<v4f_ *>(p));
    else *reinterpret_cast<v4f_ *>(p) = q;
}

template <bool BYVAL, bool PIXELSCREEN = false>
__global__ __launch_bounds__(kBlock) void cam_brick_kernel(Dims D, View4 depth, View2 camdist, View2 fl, View5 vox,
                                                            View5 cnt, float prefill, float bias, float post_scale,
                                                            float post_bias, float fill_val, int vec_ok, float fl_val,
                                                            float cd_val, int *__restrict__ cell_live)
{
    __shared__ double s_sum[kQVox];
    __shared__ unsigned s_cnt[kQVox];
    __shared__ float s_min[kBlock / 64], s_max[kBlock / 64];
    __shared__ int s_any[kBlock / 64];
    __shared__ int s_hit;
    // (no runtime integer division: gfx950 has no divide instruction, ~40 instructions each, and five of them stood in front of
    // every workgroup's first load -- the image is (blockIdx.y, blockIdx.z) = (n, c), the brick's coordinates come from two
    // float-reciprocal quotients, exact for these sizes: csrc/sph_render_seg.hip)
    const int nbz = (D.Z + kQZ - 1) / kQZ, nby = (D.Y + kQY - 1) / kQY;       // (constant divisors: shifts)
    int t1, bx;
    if (gridDim.x < (1u << 20)) {
        t1 = (int)(((float)blockIdx.x + 0.5f) * __builtin_amdgcn_rcpf((float)nbz));
        bx = (int)(((float)t1 + 0.5f) * __builtin_amdgcn_rcpf((float)nby));
    } else {                                                             // (volumes of more than 2^20 bricks: the quotients as such)
        t1 = (int)(blockIdx.x / (unsigned)nbz);
        bx = t1 / nby;
    }
    const int bz = (int)blockIdx.x - t1 * nbz, by = t1 - bx * nby;
    const int n = blockIdx.y, c = blockIdx.z, img = n * D.NC + c;
    const float f = BYVAL ? fl_val : fl.p[n * fl.s0 + c * fl.s1];
    const float cam_dist = BYVAL ? cd_val : camdist.p[n * camdist.s0 + c * camdist.s1];
    const float *dimg = depth.p + n * depth.s0 + c * depth.s1;
    float *vimg = vox.p + n * vox.s0 + c * vox.s1, *cimg = cnt.p + n * cnt.s0 + c * cnt.s1;
    const int x0 = bx * kQX, y0 = by * kQY, z0 = bz * kQZ;
    const int x1 = min(x0 + kQX, D.X), y1 = min(y0 + kQY, D.Y), z1 = min(z0 + kQZ, D.Z);
    const float rX = 1.0f / (float)D.X, rY = 1.0f / (float)D.Y, rZ = 1.0f / (float)D.Z;
    const float bxlo = (float)x0 * rX - 0.5f, bxhi = (float)x1 * rX - 0.5f;
    const Win bw = project_box(D, bxlo, bxhi, (float)y0 * rY - 0.5f, (float)y1 * rY - 0.5f, (float)z0 * rZ - 0.5f,
                               (float)z1 * rZ - 0.5f, cam_dist, f, 1.0f);
    const int bww = bw.w1 - bw.w0 + 1, bwh = bw.h1 - bw.h0 + 1;
    const int area = (bww > 0 && bwh > 0) ? bww * bwh : 0;
    // ---- (a) footprint: the first kQDeep x 256 pixels stay in registers for (b); larger footprints (camera inside
    // the grid) are screened here and fetched again there
    Band bb{3.0e38f, 0.0f, 0};
    const float inv_bww = 1.0f / (float)(bww > 0 ? bww : 1);
    auto fetch = [&](int t) {
        int r, q;
        divmod_px(t, bww, inv_bww, r, q);
        return dimg[(bw.h0 + r) * depth.s2 + (bw.w0 + q) * depth.s3];
    };
    float dv[kQDeep];
#pragma unroll
    for (int u = 0; u < kQDeep; u++) {
        const int t = threadIdx.x + u * kBlock;
        dv[u] = t < area ? fetch(t) : -1.0f;
    }
    auto screen = [&](float d) {
        if (d > 0.0f) { bb.dmin = fminf(bb.dmin, d); bb.dmax = fmaxf(bb.dmax, d); }
        else if (!(d < 0.0f)) bb.any_zero = 1;          // d == 0 (or NaN): lands at x = -cam_dist
    };
    float blo, bhi;
    bool bspecial, live;
    if (PIXELSCREEN) {
        // the tile is cleared and the barrier passed while the pixel loads are in flight (a workgroup barrier does not wait
        // for outstanding loads); "special" from the geometry alone: the slab contains the plane x = -cam_dist, or the
        // camera is inside the grid -- then no per-pixel shortcut is safe (slab_live)
        for (int e = threadIdx.x; e < kQVox; e += kBlock) { s_sum[e] = 0.0; s_cnt[e] = 0u; }
        if (threadIdx.x == 0) s_hit = 0;
        __syncthreads();
        const float eps = 1e-4f;
        bspecial = !(f > 0.0f) || !(bxlo + cam_dist > 1e-3f) ||
                   (bxlo + cam_dist - eps <= 0.0f && bxhi + cam_dist + eps >= 0.0f);
        live = true;
    } else {
#pragma unroll
        for (int u = 0; u < kQDeep; u++)
            if ((int)threadIdx.x + u * kBlock < area) screen(dv[u]);
        for (int t = threadIdx.x + kQDeep * kBlock; t < area; t += kBlock) screen(fetch(t));
        // positive floats order like their bit patterns: min / max of the depths as integers (dmin starts at 3e38, dmax at 0)
        bb.dmin = __int_as_float(wave_reduce_bits(__float_as_int(bb.dmin), 0x7f7fffff, [](int a, int b) { return a < b ? a : b; }));
        bb.dmax = __int_as_float(wave_reduce_bits(__float_as_int(bb.dmax), 0, [](int a, int b) { return a > b ? a : b; }));
        bb.any_zero = wave_reduce_bits(bb.any_zero, 0, [](int a, int b) { return a | b; });
        if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = bb.dmin; s_max[threadIdx.x >> 6] = bb.dmax; s_any[threadIdx.x >> 6] = bb.any_zero; }
        // the tile is cleared before the brick is known to be live: the stores ride under the reduction's barrier
        for (int e = threadIdx.x; e < kQVox; e += kBlock) { s_sum[e] = 0.0; s_cnt[e] = 0u; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kBlock / 64; i++) { bb.dmin = fminf(bb.dmin, s_min[i]); bb.dmax = fmaxf(bb.dmax, s_max[i]); bb.any_zero |= s_any[i]; }
        live = slab_live(bb, bw, bxlo, bxhi, cam_dist, f, blo, bhi, bspecial);
    }
    bool hit = false;
    if (live) {
        // ---- (b) every footprint pixel once, by the thread that fetched it: a cheap plane-depth test first (1-ulp
        // rsqrt, generous margin, as in cam_gather_kernel -- only ~1/16 of the footprint's points lie in this brick's x
        // range), the reference's arithmetic (pixel_voxel) for the survivors
        const float xlo_t = bxlo - 1e-5f, xhi_t = bxhi + 1e-5f;
        auto pixel = [&](int t, float d_raw) {
            if (d_raw < 0.0f) return;                                       // :225
            int r, q;
            divmod_px(t, bww, inv_bww, r, q);
            const int h = bw.h0 + r, w = bw.w0 + q;
            if (!bspecial) {
                const float u_h = (float)h - ((float)D.H - 1.0f) / 2.0f, u_w = (float)w - ((float)D.W - 1.0f) / 2.0f;
                const float xp = d_raw * f * __frsqrt_rn(u_h * u_h + u_w * u_w + f * f) - cam_dist;
                if (xp < xlo_t || xp > xhi_t) return;
            }
            int ix, iy, iz;
            float dist;
            if (pixel_voxel<false>(D, true, d_raw, 0.f, 0.f, 0.f, f, cam_dist, h, w, ix, iy, iz, dist) < 0) return;
            if (ix < x0 || ix >= x1 || iy < y0 || iy >= y1 || iz < z0 || iz >= z1) return;
            const int l = ((ix - x0) * kQY + (iy - y0)) * kQZ + (iz - z0);
            unsafeAtomicAdd(&s_sum[l], (double)dist);                       // :273
            atomicAdd(&s_cnt[l], 1u);                                       // :274
            hit = true;
        };
#pragma unroll
        for (int u = 0; u < kQDeep; u++)
            if ((int)threadIdx.x + u * kBlock < area) pixel(threadIdx.x + u * kBlock, dv[u]);
        for (int t = threadIdx.x + kQDeep * kBlock; t < area; t += kBlock) pixel(t, fetch(t));
        if (PIXELSCREEN && __ballot(hit) != 0ull && (threadIdx.x & 63) == 0) s_hit = 1;
        __syncthreads();
        if (PIXELSCREEN) live = s_hit != 0;
    }
    // occupancy for the consumer (the segment renderer, csrc/sph_render_seg.hip, does not read tiles none of whose cells
    // received a point): one word per image and cell = this workgroup's brick, written by its owner -- no clearing pass, no
    // atomics.  0 => every voxel of the cell holds fill_val.
    if (cell_live != nullptr && threadIdx.x == 0) cell_live[(size_t)img * gridDim.x + blockIdx.x] = live ? 1 : 0;
    // ---- (c) normalise (:291-305) and write the brick; a dead brick streams the fill values ---------------------
    auto value = [&](int l, float &k) {
        k = (float)s_cnt[l];
        // (sum - bias) / k  (:304): a correctly rounded division, as the reference's -- once per voxel, not on the hot path
        return k > 0.0f ? post_bias + post_scale * (((prefill + (float)s_sum[l]) - bias) / k) : fill_val;
    };
    if (vec_ok && z1 - z0 == kQZ) {
        const int z4 = (threadIdx.x & (kQZ / 4 - 1)) * 4;
        for (int xy = threadIdx.x / (kQZ / 4); xy < kQX * kQY; xy += kBlock / (kQZ / 4)) {
            const int ix = x0 + xy / kQY, iy = y0 + xy % kQY;
            if (ix >= D.X || iy >= D.Y) continue;
            float4 tv = make_float4(fill_val, fill_val, fill_val, fill_val), kv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                const int l = xy * kQZ + z4;
                tv.x = value(l, kv.x); tv.y = value(l + 1, kv.y); tv.z = value(l + 2, kv.z); tv.w = value(l + 3, kv.w);
            }
            camq_store4(vimg + ix * vox.s2 + iy * vox.s3 + (z0 + z4), tv);
            camq_store4(cimg + ix * cnt.s2 + iy * cnt.s3 + (z0 + z4), kv);
        }
    } else {
        const int lz = threadIdx.x & (kQZ - 1), iz = z0 + lz;
        for (int xy = threadIdx.x / kQZ; xy < kQX * kQY; xy += kBlock / kQZ) {
            const int ix = x0 + xy / kQY, iy = y0 + xy % kQY;
            if (ix >= D.X || iy >= D.Y || iz >= D.Z) continue;
            float k = 0.0f;
            const float tval = live ? value(xy * kQZ + lz, k) : fill_val;
            vimg[ix * vox.s2 + iy * vox.s3 + iz * vox.s4] = tval;
            cimg[ix * cnt.s2 + iy * cnt.s3 + iz * cnt.s4] = k;
        }
    }
}
// ---- camera forward for volumes WITHOUT contiguous z rows (image-minor, strided): fill + LEADER pass, no atomics (round 5) ----
// The batch-minor renderer wants the volume image-minor (element (n,x,y,z) at x*sx + y*sy + z*sz + n): no z rows for the brick
// kernel above, and until round 5 such outputs took fill + scatter_tile_kernel + normalise_tile_kernel -- hardware float
// atomics in undefined order, like the reference.  When the camera is passed BY VALUE (one camera for the whole batch: what
// Camera_back_projection_layer does, camera_backprojection_module.py:12-21) the host can bound how far apart, in pixels, two
// points of one voxel can lie: |du| <= f (dy / x_min + |y|_max dx / x_min^2) with dy, dx <= one voxel, |y| <= 1/2 and
// x_min = cam_dist - 1/2 the nearest in-grid plane -- 2.49 px for f = 418.3, cam_dist = 2.2, res = 128, so contributors of one
// voxel lie within +-2 pixels of one another.  A wave then owns an 8x8 pixel tile: it evaluates the reference's per-pixel
// arithmetic (pixel_voxel) once for every pixel of the tile AND of a halo of HALO pixels, parks (voxel key, distance) in LDS,
// and every pixel scans its (2 HALO + 1)^2 window IN ROW-MAJOR ORDER for pixels of the same voxel -- summing their distances in
// fp32 from the prefill, exactly the reference's serial index order (back_projection_kernel.cu:215-275: what a
// serial host evaluation of the reference does).  The first contributor in that order -- the LEADER -- writes the voxel's normalised value (:291-305, with the layer's
// shift folded in) and its count; nobody else touches it.  No atomics, no second pass over the pixels, run-to-run
// deterministic, and tdf / cnt BIT-IDENTICAL to a serial evaluation of the reference on every voxel
// (tests/test_gpu_cam_bp.py::test_image_minor_camera_forward_is_deterministic_and_bit_identical_to_the_serial_reference).
// Replaces round 4's opt-in cam_bm_brick_kernel (LDS bricks over 32 images: 335 us at batch 32 against 149 for the atomics).
struct BrickFlags { int *p; int per_group, nbx, nby, nbz, bx, by, bz, sg, sx, sy, sz; };   // int32 [groups, nbx, nby, nbz]; p == nullptr: none
                                                                    // (sg, sx, sy, sz: log2 of per_group, bx, by, bz, or -1)

template <int HALO>
__global__ __launch_bounds__(kBlock) void cam_leader_kernel(Dims D, View4 depth, View5 vox, View5 cnt, float cam_dist, float f,
                                                            float prefill, float bias, float post_scale, float post_bias,
                                                            BrickFlags flags, float d_screen, int fast_idx)
{
    // fast_idx: fewer than 2^20 tiles and a grid of <= 1024 voxels per axis.  Then nothing in this kernel divides integers at run
    // time (gfx950 has no divide instruction: ~40 instructions each; four per tile to take its number apart, three per leader to
    // take its voxel's index apart and four more for its brick were a third of a tile's instructions): tile -> (tx, ty, c, n) by
    // float-reciprocal quotients (exact for these sizes), the window's keys are the voxel's coordinates PACKED (x | y << 10 |
    // z << 20: only equality is ever asked of a key), bricks and image groups of power-of-two size by shifts
    constexpr int TW = 8 + 2 * HALO, TN = TW * TW, kWaves = kBlock / 64;
    __shared__ int s_key[kWaves][TN];
    __shared__ float s_dist[kWaves][TN];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tw = (D.W + 7) >> 3, th = (D.H + 7) >> 3;
    const int tiles = D.N * D.NC * th * tw;
    int *key_w = s_key[wave];
    float *dist_w = s_dist[wave];
    const int my = (HALO + (lane >> 3)) * TW + HALO + (lane & 7);
    for (int tile = blockIdx.x * kWaves + wave; tile < tiles; tile += gridDim.x * kWaves) {
        int tx, ty, c, n;
        if (fast_idx) {
            const int t1 = (int)(((float)tile + 0.5f) * __builtin_amdgcn_rcpf((float)tw));
            tx = tile - t1 * tw;
            const int t2 = (int)(((float)t1 + 0.5f) * __builtin_amdgcn_rcpf((float)th));
            ty = t1 - t2 * th;
            n = (int)(((float)t2 + 0.5f) * __builtin_amdgcn_rcpf((float)D.NC));
            c = t2 - n * D.NC;
        } else {
            int t = tile;
            tx = t % tw; t /= tw;
            ty = t % th; t /= th;
            c = t % D.NC; n = t / D.NC;
        }
        const int h0 = ty * 8 - HALO, w0 = tx * 8 - HALO;
        const float *dimg = depth.p + n * depth.s0 + c * depth.s1;
        // phase 1: the window's pixels, all depth loads of a lane in flight together
        constexpr int kRounds = (TN + 63) / 64;
        float d_raw[kRounds];
        int hh[kRounds], ww[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const int e = lane + r * 64;
            hh[r] = h0 + e / TW; ww[r] = w0 + e % TW;
            d_raw[r] = -1.0f;
            if (e < TN && hh[r] >= 0 && hh[r] < D.H && ww[r] >= 0 && ww[r] < D.W)
                d_raw[r] = dimg[hh[r] * (int)depth.s2 + ww[r] * (int)depth.s3];
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const int e = lane + r * 64;
            int ix, iy, iz, key = -1;
            float dist = 0.f;
            // depth screen: a point lands in the grid only if its depth along the optical axis, d_raw cos(theta) <= d_raw, reaches
            // the grid's near plane cam_dist - 1/2 (d_screen, a hair below it): background pixels (depth 0) and everything in
            // front of the cube skip the arithmetic -- two thirds of a GenRe depth map, whole rounds of a wave at a time
            if (d_raw[r] >= d_screen)
                key = pixel_voxel<false>(D, e < TN, d_raw[r], 0.f, 0.f, 0.f, f, cam_dist, hh[r], ww[r], ix, iy, iz, dist);
            if (fast_idx && key >= 0) key = ix | (iy << 10) | (iz << 20);
            if (e < TN) { key_w[e] = key; dist_w[e] = dist; }
            any |= key >= 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // (LDS operations of one wave execute in order)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__ballot(any) != 0ull) {
            // phase 2: this lane's pixel against its window, row-major = the reference's serial order
            const int key = key_w[my];
            if (key >= 0) {
                bool leader = true;
                float sum = prefill, k = 0.0f;                           // cam_back_projection.py:23-24
#pragma unroll
                for (int dy = -HALO; dy <= HALO; dy++) {
#pragma unroll
                    for (int dx = -HALO; dx <= HALO; dx++) {
                        const int e = my + dy * TW + dx;
                        if (key_w[e] == key) {
                            if (dy < 0 || (dy == 0 && dx < 0)) leader = false;
                            sum = sum + dist_w[e];                       // :273, serial order
                            k = k + 1.0f;                                // :274
                        }
                    }
                }
                if (leader) {
                    int ix, iy, iz;
                    if (fast_idx) { ix = key & 1023; iy = (key >> 10) & 1023; iz = key >> 20; }
                    else { iz = key % D.Z; iy = (key / D.Z) % D.Y; ix = key / (D.Z * D.Y); }
                    vox.p[n * vox.s0 + c * vox.s1 + vox_off(vox, ix, iy, iz)] = post_bias + post_scale * ((sum - bias) / k);   // :304
                    cnt.p[n * cnt.s0 + c * cnt.s1 + vox_off(cnt, ix, iy, iz)] = k;
                    // occupancy for the consumer (the batch-minor renderer skips tiles -- a brick plus the voxels one step beyond
                    // its HIGH faces -- that hold the fill value in every image of the group): one word per brick and image group,
                    // cleared by the host entry, set for the voxel's brick and its <= 7 neighbours on the LOW side (whose tiles
                    // may reach this voxel); every writer stores 1
                    if (flags.p) {
                        int *fg = flags.p + (size_t)(flags.sg >= 0 ? n >> flags.sg : n / flags.per_group) * flags.nbx * flags.nby * flags.nbz;
                        const int b0 = flags.sx >= 0 ? ix >> flags.sx : ix / flags.bx, b1 = flags.sy >= 0 ? iy >> flags.sy : iy / flags.by,
                                  b2 = flags.sz >= 0 ? iz >> flags.sz : iz / flags.bz;
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const int x = b0 - (q & 1), y = b1 - ((q >> 1) & 1), z = b2 - (q >> 2);
                            if (x >= 0 && y >= 0 && z >= 0) fg[(x * flags.nby + y) * flags.nbz + z] = 1;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the next tile overwrites the window
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// how many pixels apart two points of ONE voxel can project (see cam_leader_kernel): -1 if the camera is too close for the bound
inline int leader_halo(const Dims &D, float f, float cam_dist)
{
    const double x_min = (double)cam_dist - 0.5;
    if (!(x_min > 0.05) || !(f > 0.0f)) return -1;
    const double dx = 1.0 / D.X, dy = 1.0 / D.Y, dz = 1.0 / D.Z;
    const double bw = (double)f * (dy / x_min + 0.5 * dx / (x_min * x_min));     // columns:  u_w = -f y / x
    const double bh = (double)f * (dz / x_min + 0.5 * dx / (x_min * x_min));     // rows:     u_h = -f z / x
    const double b = (bw > bh ? bw : bh) * (1.0 + 1e-4) + 1e-3;                   // fp32 rounding of the coordinates
    return (int)b;                                                                // integer pixel distances are <= floor(b)
}

// ---- K3: surface mask (:324-357), one lane per voxel, z fastest -----------------
__device__ __forceinline__ int floor_i_d(double a) { return (a < 0) ? (int)a - 1 : (int)a; }
__device__ __forceinline__ int round_i_d(double a)
{   // ROUND_I (:42-43): FLOOR_F is (float)FLOOR_I; ties go down
    const double ff = (double)(float)floor_i_d(a);
    return (a - ff > ff + 1.0 - a) ? floor_i_d(a) + 1 : floor_i_d(a);
}

__global__ __launch_bounds__(kBlock) void surface_mask_kernel(Dims D, View4 depth, View2 camdist, View2 fl,
                                                               View5 cnt, View5 mask)
{
    const int64_t total = (int64_t)D.N * D.NC * D.X * D.Y * D.Z;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        int64_t r = i;
        const int iz = (int)(r % D.Z); r /= D.Z;
        const int iy = (int)(r % D.Y); r /= D.Y;
        const int ix = (int)(r % D.X); r /= D.X;
        const int c = (int)(r % D.NC);
        const int n = (int)(r / D.NC);
        float m = 1.0f;                                                // :853 fill
        const float ptnum = cnt.p[n * cnt.s0 + c * cnt.s1 + ix * cnt.s2 + iy * cnt.s3 + iz * cnt.s4];
        if (!((double)ptnum > 1e-5)) {                                 // :333
            const float f = fl.p[n * fl.s0 + c * fl.s1];
            const float cam_dist = camdist.p[n * camdist.s0 + c * camdist.s1];
            const float cx = centre_d(ix, D.X), cy = centre_d(iy, D.Y), cz = centre_d(iz, D.Z);
            const float im_h = -cz * f / (cx + cam_dist);              // :339
            const float im_w = -cy * f / (cx + cam_dist);              // :340
            const int ih = round_i_d(0.5 * ((double)(float)D.H - 1.0) + (double)im_h);
            const int iw = round_i_d(0.5 * ((double)(float)D.W - 1.0) + (double)im_w);
            if (ih >= 0 && ih < D.H && iw >= 0 && iw < D.W) {
                const float d = depth.p[n * depth.s0 + c * depth.s1 + ih * depth.s2 + iw * depth.s3];
                if (!(d < 0.0f)) {
                    const float ray = norm3(cx + cam_dist, cy, cz);    // :353
                    if (d < ray) m = 0.0f;
                }
            }
        }
        mask.p[n * mask.s0 + c * mask.s1 + ix * mask.s2 + iy * mask.s3 + iz * mask.s4] = m;
    }
}

// ---- K4: camera backward (:387-470) ----------------------------------------------
// grid = (blocks per image, N*NC).  grad_depth is written for every pixel (0 where
// the reference leaves its zero fill).  The two per-image scalars are reduced in
// fp64 inside the wave (DPP/bpermute shuffles) and the block (LDS), then ONE fp32
// atomic per block -- the reference does 2 same-address atomics per pixel.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Four pixels per thread are in flight (depth loads, then the dependent cnt / grad_in gathers, then the
// arithmetic), and the launch keeps the TOTAL number of workgroups near 512: grad_fl / grad_camdist of all
// images share one cache line each, and same-line float atomics retire at ~10 ns apiece -- 8192 workgroups
// (256 per image at batch 32) made this kernel 85 us of atomic queueing around 3 us of work.
constexpr int kBwdUnroll = 4;

// Round 6: pixels whose depth cannot reach the grid's near plane (background: two thirds of a GenRe depth map) skip the arithmetic,
// and the pixel -> (row, column) split is a float multiply + correction (divmod_px), not an integer division: 34.8 -> 28.5 us at
// batch 32, back to back, either layout.  (1024-thread workgroups that take their pixels in ONE pass instead of four: 30.0 us --
// the kernel is bound by its correctly rounded divisions and square roots at two waves per SIMD, not by the passes; the random
// 4-byte gathers do not show either: the image-minor and the NCXYZ volume, a random and an all-zero gradient time the same.)
// "this image's incoming gradient is identically zero": word [(image / group) * stride + offset] == 0 (genre_hip.h:
// genre_back_projection_backward_hinted); words == nullptr: no hint
struct ZeroHint { const int *words; int64_t stride, offset; int group; };

template <int NT>
__global__ __launch_bounds__(NT) void cam_backward_kernel(Dims D, View4 depth, View2 fl, View2 camdist,
                                                           View5 cnt, View5 gin, View4 gdepth,
                                                           View2 gcam, View2 gfl, float gscale, ZeroHint zh)
{
    __shared__ double red[2][NT / 64];
    const int n = blockIdx.y, c = blockIdx.z;
    const int npix = D.H * D.W;
    if (zh.words != nullptr && zh.words[(int64_t)((n * D.NC + c) / zh.group) * zh.stride + zh.offset] == 0) {
        // the producer of grad_in says this image's incoming gradient is identically zero (the renderer's backward where the clamp
        // in front of it blocks every voxel: GenRe's own chain): grad_depth = 0, nothing for grad_fl / grad_camdist
        float *gz = gdepth.p + n * gdepth.s0 + c * gdepth.s1;
        const float inv = 1.0f / (float)D.W;
        for (int p = blockIdx.x * NT + threadIdx.x; p < npix; p += gridDim.x * NT) {
            int h, w;
            divmod_px(p, D.W, inv, h, w);
            gz[h * gdepth.s2 + w * gdepth.s3] = 0.0f;
        }
        return;
    }
    const float inv_w = 1.0f / (float)D.W;
    double acc_fl = 0.0, acc_cd = 0.0;
    const float f = fl.p[n * fl.s0 + c * fl.s1];
    const float cam_dist = camdist.p[n * camdist.s0 + c * camdist.s1];
    // a point lands in the grid only if its depth along the optical axis, d cos(theta) <= d, reaches the near plane cam_dist - 1/2
    // (a hair below it; cameras inside the cube: no screen)
    const float d_screen = cam_dist > 0.5f ? (cam_dist - 0.5f) * (1.0f - 1e-5f) : -1.0f;
    const float *dimg = depth.p + n * depth.s0 + c * depth.s1;
    float *gdimg = gdepth.p + n * gdepth.s0 + c * gdepth.s1;
    const float *cimg = cnt.p + n * cnt.s0 + c * cnt.s1;
    const float *gimg = gin.p + n * gin.s0 + c * gin.s1;
    for (int p0 = blockIdx.x * (NT * kBwdUnroll) + threadIdx.x; p0 < npix; p0 += gridDim.x * (NT * kBwdUnroll)) {
        float d_i[kBwdUnroll], gx[kBwdUnroll], gy[kBwdUnroll], gz[kBwdUnroll], u_h[kBwdUnroll], u_w[kBwdUnroll];
        float ptnum[kBwdUnroll], gd[kBwdUnroll];
        int ix[kBwdUnroll], iy[kBwdUnroll], iz[kBwdUnroll], hh[kBwdUnroll], ww[kBwdUnroll];
        bool live[kBwdUnroll];
#pragma unroll
        for (int u = 0; u < kBwdUnroll; u++) {
            const int p = p0 + u * NT;
            divmod_px(p < npix ? p : 0, D.W, inv_w, hh[u], ww[u]);
            d_i[u] = -1.0f;
            if (p < npix) d_i[u] = dimg[hh[u] * depth.s2 + ww[u] * depth.s3];
        }
#pragma unroll
        for (int u = 0; u < kBwdUnroll; u++) {
            const int h = hh[u], w = ww[u];
            live[u] = d_i[u] >= d_screen && !(d_i[u] < 0.0f);           // :225 (and the screen: such a point is outside the grid)
            ix[u] = iy[u] = iz[u] = 0;
            ptnum[u] = 1.0f; gd[u] = 0.0f;
            if (live[u]) {
                // camera model of :231-242 (same sequence as pixel_point<false>)
                u_h[u] = (float)h - ((float)D.H - 1.0f) / 2.0f;
                u_w[u] = (float)w - ((float)D.W - 1.0f) / 2.0f;
                const float cos_theta = f / norm3(u_h[u], u_w[u], f);
                const float d = d_i[u] * cos_theta;
                gy[u] = -d * u_w[u] / f;
                gz[u] = -d * u_h[u] / f;
                gx[u] = d - cam_dist;
                ix[u] = vox_index(gx[u], D.X); iy[u] = vox_index(gy[u], D.Y); iz[u] = vox_index(gz[u], D.Z);
                live[u] = in_grid(D, ix[u], iy[u], iz[u]);
            }
            if (live[u]) {
                ptnum[u] = cimg[ix[u] * cnt.s2 + iy[u] * cnt.s3 + iz[u] * cnt.s4];
                gd[u] = gimg[ix[u] * gin.s2 + iy[u] * gin.s3 + iz[u] * gin.s4];
                // a voxel whose incoming gradient is exactly zero gives its pixel exactly zero (every other factor below is finite:
                // L, Dn >= 1e-5, k >= 1) -- on GenRe's own chain (the x50 clamp blocks every voxel) that is every pixel, and the
                // divisions and square roots below are this kernel's duration.  (A NaN is not zero and goes through.)
                live[u] = gd[u] != 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < kBwdUnroll; u++) {
            const int p = p0 + u * NT;
            float gd_out = 0.0f;
            if (live[u]) {
                const float cx = centre_d(ix[u], D.X), cy = centre_d(iy[u], D.Y), cz = centre_d(iz[u], D.Z);
                float L = norm3(u_h[u], u_w[u], f);                     // :432
                if ((double)L < 1e-5) L = (float)1e-5;
                const float rx = -f / L, ry = u_w[u] / L, rz = u_h[u] / L;   // :436-438
                float Dn = norm3(gx[u] - cx, gy[u] - cy, gz[u] - cz);   // :440
                if ((double)Dn < 1e-5) Dn = (float)1e-5;
                const float qx = (gx[u] - cx) / Dn, qy = (gy[u] - cy) / Dn, qz = (gz[u] - cz) / Dn;
                const float cos_cc = (rx * qx) + (ry * qy) + (rz * qz); // :448
                float k = ptnum[u];
                if (!(k >= 1.0f)) k = 1.0f;                             // :452 max(cnt, 1); also a NaN (a sparse cnt is only defined where a point landed)
                // gscale = 1, or -res when the incoming gradient is w.r.t. the shifted output 1 - res*tdf
                const float g = gd[u] * gscale;
                gd_out = -g * cos_cc / k;                               // :455
                const float L3 = L * L * L;
                const float gfx = ((gx[u] - cx) / Dn) * (u_w[u] * u_w[u] + u_h[u] * u_h[u]) / L3;   // :459
                const float gfy = ((gy[u] - cy) / Dn) * (u_w[u] * f) / L3;                          // :460
                const float gfz = ((gz[u] - cz) / Dn) * (u_h[u] * f) / L3;                          // :461
                acc_fl += (double)((gfx + gfy + gfz) * g * d_i[u] / k);                             // :462
                acc_cd += (double)(-qx * g / k);                                                    // :469
            }
            if (p < npix) gdimg[hh[u] * gdepth.s2 + ww[u] * gdepth.s3] = gd_out;
        }
    }
    acc_fl = wave_sum(acc_fl);
    acc_cd = wave_sum(acc_cd);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = acc_fl; red[1][wv] = acc_cd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < NT / 64; i++) { a += red[0][i]; b += red[1][i]; }
        unsafeAtomicAdd(gfl.p + n * gfl.s0 + c * gfl.s1, (float)a);
        unsafeAtomicAdd(gcam.p + n * gcam.s0 + c * gcam.s1, (float)b);
    }
}

__global__ void zero2_kernel(View2 a, View2 b, int N, int NC)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * NC) {
        const int n = i / NC, c = i % NC;
        a.p[n * a.s0 + c * a.s1] = 0.0f;
        b.p[n * b.s0 + c * b.s1] = 0.0f;
    }
}

// ---- K6: spherical backward (:560-626) ---------------------------------------------
__global__ __launch_bounds__(kBlock) void sph_backward_kernel(Dims D, View4 depth, View5 grid, View5 cnt,
                                                               View5 gin, View4 gdepth, float gscale)
{
    const int64_t total = (int64_t)D.N * D.NC * D.H * D.W;
    const View2 none = {nullptr, 0, 0};
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * kBlock) {
        int n, c, h, w;
        decode_pixel(D, idx, n, c, h, w);
        float out = 0.0f;
        float d, gx, gy, gz, u_h, u_w, f;
        if (pixel_point<true>(D, depth, none, none, grid, n, c, h, w, d, gx, gy, gz, u_h, u_w, f)) {
            const int ix = vox_index(gx, D.X), iy = vox_index(gy, D.Y), iz = vox_index(gz, D.Z);
            if (in_grid(D, ix, iy, iz)) {
                const float cx = centre_d(ix, D.X), cy = centre_d(iy, D.Y), cz = centre_d(iz, D.Z);
                float L = norm3(gx, gy, gz);                            // :600
                if ((double)L < 1e-5) L = (float)1e-5;
                const float rx = gx / L, ry = gy / L, rz = gz / L;
                const float cos_cc = (rx * cx) + (ry * cy) + (rz * cz); // :608
                float dist = norm3(gx - cx, gy - cy, gz - cz);          // :609
                float ptnum = cnt.p[n * cnt.s0 + c * cnt.s1 + ix * cnt.s2 + iy * cnt.s3 + iz * cnt.s4];
                if (ptnum < 1.0f) ptnum = 1.0f;
                if ((double)dist < 1e-5) dist = (float)1e-5;
                const float gd = gin.p[n * gin.s0 + c * gin.s1 + ix * gin.s2 + iy * gin.s3 + iz * gin.s4] * gscale;
                out = gd * (d - cos_cc) / (ptnum * dist);               // :621
            }
        }
        gdepth.p[n * gdepth.s0 + c * gdepth.s1 + h * gdepth.s2 + w * gdepth.s3] = out;
    }
}

// ---- host side ---------------------------------------------------------------------
inline int grid_for(int64_t work_items, int cap_blocks = kCUs * 8)
{
    int64_t b = (work_items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return (int)(b > cap_blocks ? cap_blocks : b);
}

// shape checks of cambp_shapecheck (:105-144 / :146-184)
int check_image(const char *op, const genre_tensor *depth, Dims &D)
{
    GENRE_REQUIRE(is_f32(depth, 4), "%s: depth must be a 4-D fp32 tensor [N,NC,H,W]", op);
    D.N = (int)depth->size[0]; D.NC = (int)depth->size[1]; D.H = (int)depth->size[2]; D.W = (int)depth->size[3];
    GENRE_REQUIRE((int64_t)D.N * D.NC * D.H * D.W < (int64_t)1 << 31, "%s: depth too large", op);
    return 1;
}
int check_scalar(const char *op, const char *name, const genre_tensor *t, const Dims &D)
{
    GENRE_REQUIRE(is_f32(t, 2) && t->size[0] == D.N && t->size[1] == D.NC,
                  "%s: %s must be a 2-D fp32 tensor [N=%d,NC=%d]", op, name, D.N, D.NC);
    return 1;
}
int check_volume(const char *op, const char *name, const genre_tensor *t, Dims &D, bool set)
{
    GENRE_REQUIRE(is_f32(t, 5) && t->size[0] == D.N && t->size[1] == D.NC,
                  "%s: %s must be a 5-D fp32 tensor [N=%d,NC=%d,X,Y,Z]", op, name, D.N, D.NC);
    if (set) { D.X = (int)t->size[2]; D.Y = (int)t->size[3]; D.Z = (int)t->size[4]; }
    GENRE_REQUIRE(t->size[2] == D.X && t->size[3] == D.Y && t->size[4] == D.Z,
                  "%s: %s spatial size must be [%d,%d,%d]", op, name, D.X, D.Y, D.Z);
    return 1;
}
int check_map(const char *op, const char *name, const genre_tensor *t, const Dims &D)
{
    GENRE_REQUIRE(is_f32(t, 4) && t->size[0] == D.N && t->size[1] == D.NC && t->size[2] == D.H &&
                      t->size[3] == D.W,
                  "%s: %s must be a 4-D fp32 tensor [%d,%d,%d,%d]", op, name, D.N, D.NC, D.H, D.W);
    return 1;
}

int launch_fill2(const Dims &D, const genre_tensor *a, float va, const genre_tensor *b, float vb,
                 hipStream_t st)
{
    const int64_t total = (int64_t)D.N * D.NC * D.X * D.Y * D.Z;
    if (total == 0) return 1;
    if (is_dense(a) && is_dense(b) && (total % 4) == 0 && aligned16(a->data) && aligned16(b->data)) {
        // one float4 pair per thread where possible: tools/fill_bench.hip measures 2 x 268 MB at 126 us with 2048
        // grid-striding workgroups, 94 us with 16 384 and 81 us (6.6 TB/s) with 65 536 and more
        fill2_vec4_kernel<<<grid_for(total / 4, 1 << 20), kBlock, 0, st>>>((float4 *)a->data, va, (float4 *)b->data, vb,
                                                                  total / 4);
    } else {
        fill2_strided_kernel<<<grid_for(total, 1 << 20), kBlock, 0, st>>>(D, view5(a), va, view5(b), vb);
    }
    GENRE_LAUNCH_CHECK("fill");
    return 1;
}

// The camera forward has three implementations.  By default the output layout picks one: the single-launch brick kernel
// when both volumes have contiguous, 16-byte aligned z rows (the reference's dense NCXYZ tensors), fill + scatter +
// normalise otherwise (e.g. the image-minor volumes of the batch-minor renderer).  The environment variable
// GENRE_CAMBP_MODE = scatter | brick | gather pins one of them for every call.  Measured on MI355X (HIP-graph replay,
// tools/ab_round2.py, per image at batch 1 / 8 / 32):
//  brick    cam_brick_kernel: one launch, 8x8x32 bricks accumulated in LDS (fp64 sums), deterministic:
//           9.7-10.1 / 4.7-4.9 / 4.0-4.2 us.  Timestamps inside the kernel put ~3 us of the batch-1 figure into the
//           dependent chain every workgroup runs before it knows whether it is live (scalar loads -> footprint -> pixel
//           loads -> reduction -> barrier) and ~3 us into the live bricks' pixel and normalise phases; queueing the
//           screened pixels (in LDS or in-wave), pairing pixels for instruction-level parallelism, 8x8x16 bricks, 512
//           threads and a speculative fill ahead of the screen were all measured within +-0.4 us of this version.
//  scatter  fill + scatter + normalise: three launches, hardware float atomics: 11.9 / 5.5-5.7 / 4.3-4.5 us; sums of
//           voxels hit more than once depend on atomic order, as in the reference.
//  gather   cam_gather_kernel: one launch, deterministic and bit-identical to the serial reference order;
//           6.6 us/image at batch 32, ~42 us at batch 1 (256 workgroups per image: too few stores in flight, and in a
//           live brick the exact arithmetic runs on almost every candidate iteration of a wave).
// (Also measured and dropped: a slab-owned single-launch variant -- one x-plane quarter per workgroup accumulated in
// LDS, whose footprint is a quarter of the IMAGE -- at 9-11 us/image at batch 32 and 18 us at batch 1; and the three
// phases in ONE launch of co-resident workgroups separated by two device-wide barriers, 44 us at batch 1:
// tools/grid_barrier_bench.hip measures 3.9 us per barrier on 256 workgroups before any fence (256 same-address
// arrivals at ~10 ns each + the polling), +1.9 us for the L2 write-back and +1.7 us for the invalidate each side
// needs so that the XCDs' L2s agree; a kernel boundary inside a HIP graph costs ~1 us.)
// The spherical path always scatters.
enum CamMode { kAuto, kScatter, kGather, kBrick };
// The only process-level setting the library reads: the environment variable GENRE_CAMBP_MODE, looked up ONCE at the first
// camera forward and constant afterwards -- a read-only configuration value, not mutable state: calls stay re-entrant
// and independent of one another (include/genre_hip.h: "keeps no global state").
inline CamMode cam_mode()
{
    static const CamMode m = [] {
        const char *e = getenv("GENRE_CAMBP_MODE");
        if (!e) return kAuto;
        return e[0] == 'g' ? kGather : e[0] == 'b' ? kBrick : e[0] == 's' ? kScatter : kAuto;
    }();
    return m;
}

// float4 rows (brick kernel; dead bricks of the gather kernel) need unit z stride and 16-byte aligned z-rows
inline bool rows_aligned(const Dims &D, const genre_tensor *t)
{
    if (t->stride[4] != 1 || !aligned16(t->data) || (D.Z % 4) != 0) return false;
    for (int i = 0; i < 4; i++)
        if (t->size[i] != 1 && (t->stride[i] % 4) != 0) return false;
    return true;
}

// which implementation genre_back_projection_forward_const takes for these outputs and this camera (genre_cam_forward_plan)
enum { kPlanNone = 0, kPlanBrick = 1, kPlanLeader = 2 };
inline int byval_plan(const Dims &D, const genre_tensor *voxel, const genre_tensor *cnt, float f, float cam_dist)
{
    const CamMode m = cam_mode();
    if (rows_aligned(D, voxel) && rows_aligned(D, cnt) && D.N * D.NC <= 65535 && (m == kBrick || m == kAuto)) return kPlanBrick;
    const int halo = leader_halo(D, f, cam_dist);
    return (m != kGather && m != kBrick && halo >= 0 && halo <= 4) ? kPlanLeader : kPlanNone;
}

template <bool SPH>
int forward_impl(const char *op, const genre_tensor *depth, const genre_tensor *camdist, const genre_tensor *fl,
                 const genre_tensor *grid, const genre_tensor *voxel, const genre_tensor *cnt, void *stream,
                 bool shifted = false, const float *byval = nullptr, const genre_tensor *tile_live = nullptr,
                 bool sparse_cnt = false)
{
    Dims D{};
    if (!check_image(op, depth, D)) return 0;
    View2 vcd{nullptr, 0, 0}, vfl{nullptr, 0, 0};
    View5 vgrid{nullptr, 0, 0, 0, 0, 0};
    if (SPH) {
        GENRE_REQUIRE(is_f32(grid, 5) && grid->size[0] == D.N && grid->size[1] == D.NC &&
                          grid->size[2] == D.H && grid->size[3] == D.W && grid->size[4] == 3,
                      "%s: grid must be a 5-D fp32 tensor [%d,%d,%d,%d,3]", op, D.N, D.NC, D.H, D.W);
        vgrid = view5(grid);
    } else if (!byval) {
        if (!check_scalar(op, "camdist", camdist, D) || !check_scalar(op, "fl", fl, D)) return 0;
        vcd = view2(camdist); vfl = view2(fl);
    }
    if (!check_volume(op, "voxel", voxel, D, true) || !check_volume(op, "cnt", cnt, D, false)) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int mx = D.X > D.Y ? (D.X > D.Z ? D.X : D.Z) : (D.Y > D.Z ? D.Y : D.Z);
    // camera path: prefill 1/res (cam_back_projection.py:23-24) and bias 1/max(res) (:304,:829) are the
    // same number for the cubic grids the reference builds; spherical path: prefill 0, bias 0 (:695).
    const float empty_val = SPH ? 0.0f : (float)(1.0 / (double)mx);
    // optional fused epilogue of Camera_back_projection_layer.shift_tdf (camera_backprojection_module.py:25-28):
    // out = 1 - res*tdf.  The "negative == raw sum" marker of the normalise pass needs out >= 0, true for
    // cubic grids (mean distance <= sqrt(3)/2 voxel).
    float post_scale = 1.0f, post_bias = 0.0f, fill_val = empty_val;
    int post_mode = 0;
    if (shifted) {
        GENRE_REQUIRE(D.X == D.Y && D.Y == D.Z, "%s: the fused shift needs a cubic grid", op);
        if (SPH) {          // (-tdf + 1/res) * res * clamp(cnt,0,1)  (genre_full_model.py:139-142); empty -> 0
            post_mode = 1; post_scale = (float)mx; post_bias = (float)(1.0 / (double)mx);
            fill_val = 0.0f;
        } else {            // 1 - res * tdf  (camera_backprojection_module.py:25-28)
            post_scale = -(float)mx; post_bias = 1.0f;
            fill_val = 1.0f - (float)mx * empty_val;
        }
    }
    const int vec_ok = rows_aligned(D, voxel) && rows_aligned(D, cnt);
    CamMode mode = SPH ? kScatter : cam_mode();
    if (mode == kAuto) mode = (vec_ok && D.N * D.NC <= 65535) ? kBrick : kScatter;
    if (byval && (SPH || byval_plan(D, voxel, cnt, byval[0], byval[1]) != kPlanBrick)) {
        // by value, but no contiguous z rows (image-minor / strided volumes): fill + the deterministic leader pass
        const int halo = SPH ? -1 : leader_halo(D, byval[0], byval[1]);
        GENRE_REQUIRE(!SPH && cam_mode() != kGather && cam_mode() != kBrick && halo >= 0 && halo <= 4,
                      "%s: the by-value entry serves dense NCXYZ outputs (single-launch brick kernel) or, for other layouts, "
                      "cameras whose voxels project to at most 4 pixels (fill + leader pass; this one: %d, GENRE_CAMBP_MODE must "
                      "not pin gather / brick); pass fl / camdist tensors otherwise", op, halo);
        const int64_t npix = (int64_t)D.N * D.NC * D.H * D.W;
        // sparse_cnt: the caller reads cnt only where a point landed (the layer keeps cnt for ITS backward, which reads it at
        // the voxel of each in-grid pixel and nowhere else) -- the leaders write exactly those elements, and without atomics no
        // voxel needs a zero to start from: half of the fill (268 of 537 MB per 32 images) is not written
        const int64_t total = (int64_t)D.N * D.NC * D.X * D.Y * D.Z;
        if (sparse_cnt && total > 0 && is_dense(voxel) && (total % 4) == 0 && aligned16(voxel->data)) {
            fill1_vec4_kernel<<<grid_for(total / 4, 1 << 20), kBlock, 0, st>>>((float4 *)voxel->data, fill_val, total / 4);
            GENRE_LAUNCH_CHECK("fill (volume only)");
        } else if (!launch_fill2(D, voxel, fill_val, cnt, 0.0f, st)) return 0;
        if (npix == 0 || (int64_t)D.X * D.Y * D.Z == 0) return 1;
        const int64_t tiles = (int64_t)D.N * D.NC * ((D.H + 7) / 8) * ((D.W + 7) / 8);
        auto span31 = [](const genre_tensor *t) {                    // per-image extent in elements < 2^31 ?
            int64_t span = 1;
            for (int i = 2; i < t->ndim; i++) span += (t->size[i] - 1) * (t->stride[i] < 0 ? -t->stride[i] : t->stride[i]);
            return span < ((int64_t)1 << 31);
        };
        GENRE_REQUIRE(tiles < ((int64_t)1 << 30) && span31(voxel) && span31(cnt) && span31(depth),
                      "%s: one image (map or volume) must span fewer than 2^31 elements", op);
        const float prefill = (float)(1.0 / (double)D.X);               // cam_back_projection.py:23-24 (res = X)
        const float bias = 1.0f / (float)mx;                             // K2: dist_bias / max(res)  (:304,:829)
        const int g = grid_for(tiles * 64, 1 << 16);
        BrickFlags flags{nullptr, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};
        if (tile_live) {         // int32 [groups, nbx, nby, nbz]: word (g, b) <- 1 iff brick b or a high-side neighbour received a point
            GENRE_REQUIRE(is_i32(tile_live, 4) && is_contiguous(tile_live) && tile_live->size[0] >= 1 && tile_live->size[1] >= 1 &&
                              tile_live->size[2] >= 1 && tile_live->size[3] >= 1 && D.NC == 1 && D.N >= 1,
                          "%s: tile_live must be a contiguous int32 [groups, nbx, nby, nbz] tensor (single-channel volumes)", op);
            flags.p = (int *)tile_live->data;
            constexpr int kGroupImgs = 32;                               // the batch-minor renderer's image groups (sph_render_bm.hip: kImgs)
            GENRE_REQUIRE(tile_live->size[0] == (D.N + kGroupImgs - 1) / kGroupImgs,
                          "%s: tile_live needs one slab per group of 32 consecutive images: size[0] == %d", op,
                          (D.N + kGroupImgs - 1) / kGroupImgs);
            flags.per_group = kGroupImgs;
            flags.nbx = (int)tile_live->size[1]; flags.nby = (int)tile_live->size[2]; flags.nbz = (int)tile_live->size[3];
            flags.bx = (D.X + flags.nbx - 1) / flags.nbx; flags.by = (D.Y + flags.nby - 1) / flags.nby;
            flags.bz = (D.Z + flags.nbz - 1) / flags.nbz;
            auto log2_or = [](int v) { int l = 0; while ((1 << l) < v) l++; return (1 << l) == v ? l : -1; };
            flags.sg = log2_or(flags.per_group); flags.sx = log2_or(flags.bx); flags.sy = log2_or(flags.by); flags.sz = log2_or(flags.bz);
            GENRE_REQUIRE(hipMemsetAsync(flags.p, 0, (size_t)numel(tile_live) * 4, st) == hipSuccess,
                          "%s: hipMemsetAsync of the brick flags failed", op);
        }
#define GENRE_CAM_LEADER(HV)                                                                                              \
        cam_leader_kernel<HV><<<g, kBlock, 0, st>>>(D, view4(depth), view5(voxel), view5(cnt), byval[1], byval[0], prefill, bias, \
                                                    post_scale, post_bias, flags, (byval[1] - 0.5f) * (1.0f - 1e-5f),     \
                                                    (tiles < (1 << 20) && D.X <= 1024 && D.Y <= 1024 && D.Z <= 1024) ? 1 : 0)
        switch (halo < 1 ? 1 : halo) {
            case 1: GENRE_CAM_LEADER(1); break;
            case 2: GENRE_CAM_LEADER(2); break;
            case 3: GENRE_CAM_LEADER(3); break;
            default: GENRE_CAM_LEADER(4); break;
        }
#undef GENRE_CAM_LEADER
        GENRE_LAUNCH_CHECK("projection forward (leader pass)");
        return 1;
    }
    // (sparse_cnt is a permission, not a request: the brick kernel writes cnt densely)
    if (byval) mode = kBrick;
    int *cell_live = nullptr;
    if (tile_live != nullptr) {
        // dense outputs: the words are per image and per cell of the brick kernel (genre_cam_cell()), each written by its owner
        const int ncx = (D.X + kQX - 1) / kQX, ncy = (D.Y + kQY - 1) / kQY, ncz = (D.Z + kQZ - 1) / kQZ;
        GENRE_REQUIRE(mode == kBrick, "%s: tile_live is produced by the brick kernel (dense outputs) or the leader pass (by-value "
                                      "camera, other layouts) only", op);
        GENRE_REQUIRE(is_i32(tile_live, 4) && is_contiguous(tile_live) && tile_live->size[0] == (int64_t)D.N * D.NC &&
                          tile_live->size[1] == ncx && tile_live->size[2] == ncy && tile_live->size[3] == ncz,
                      "%s: for dense outputs tile_live must be a contiguous int32 [N*NC, %d, %d, %d] tensor (one word per image "
                      "and %dx%dx%d-voxel cell)", op, ncx, ncy, ncz, kQX, kQY, kQZ);
        cell_live = (int *)tile_live->data;
    }
    if (mode != kScatter) {
        const int64_t nvox = (int64_t)D.X * D.Y * D.Z;
        if (nvox == 0 || D.N * D.NC == 0) return 1;
        GENRE_REQUIRE(D.N * D.NC <= 65535, "%s: N*NC must be <= 65535", op);
        const float prefill = (float)(1.0 / (double)D.X);               // cam_back_projection.py:23-24 (res = X)
        const float bias = 1.0f / (float)mx;                             // K2: dist_bias / max(res)  (:304,:829)
        if (mode == kBrick) {
            const int64_t bricks = (int64_t)((D.X + kQX - 1) / kQX) * ((D.Y + kQY - 1) / kQY) * ((D.Z + kQZ - 1) / kQZ);
            GENRE_REQUIRE(bricks < ((int64_t)1 << 31), "%s: volume too large", op);
            const dim3 bgrid((unsigned)bricks, D.N, D.NC);
            const bool pixelscreen = D.N * D.NC <= GENRE_CAMQ_PIXELSCREEN_MAXN;
#define GENRE_CAMQ_LAUNCH(BV, PXS, A, B_)                                                                                 \
            cam_brick_kernel<BV, PXS><<<bgrid, kBlock, 0, st>>>(D, view4(depth), vcd, vfl, view5(voxel), view5(cnt), prefill, bias, \
                                                                post_scale, post_bias, fill_val, vec_ok, A, B_, cell_live)
            if (byval) { if (pixelscreen) GENRE_CAMQ_LAUNCH(true, true, byval[0], byval[1]); else GENRE_CAMQ_LAUNCH(true, false, byval[0], byval[1]); }
            else { if (pixelscreen) GENRE_CAMQ_LAUNCH(false, true, 0.0f, 0.0f); else GENRE_CAMQ_LAUNCH(false, false, 0.0f, 0.0f); }
#undef GENRE_CAMQ_LAUNCH
            GENRE_LAUNCH_CHECK("projection forward (bricks)");
            return 1;
        }
        const int bricks = ((D.X + kGX - 1) / kGX) * ((D.Y + kGY - 1) / kGY) * ((D.Z + kGZ - 1) / kGZ);
        cam_gather_kernel<<<dim3(bricks, D.N * D.NC), kBlock, 0, st>>>(D, view4(depth), vcd, vfl, view5(voxel),
                                                                      view5(cnt), prefill, bias, post_scale, post_bias,
                                                                      fill_val, vec_ok);
        GENRE_LAUNCH_CHECK("projection forward (gather)");
        return 1;
    }
    const int64_t npix = (int64_t)D.N * D.NC * D.H * D.W;
    if (!launch_fill2(D, voxel, fill_val, cnt, 0.0f, st)) return 0;
    if (npix == 0 || (int64_t)D.X * D.Y * D.Z == 0) return 1;
    const int64_t tiles = (int64_t)D.N * D.NC * ((D.H + 7) / 8) * ((D.W + 7) / 8);
    auto span32 = [](const genre_tensor *t, int first) {          // per-image extent in elements < 2^31 ?
        int64_t span = 1;
        for (int i = first; i < t->ndim; i++) span += (t->size[i] - 1) * (t->stride[i] < 0 ? -t->stride[i] : t->stride[i]);
        return span < ((int64_t)1 << 31);
    };
    GENRE_REQUIRE(tiles < ((int64_t)1 << 30) && span32(voxel, 2) && span32(cnt, 2) && span32(depth, 2) &&
                      (!SPH || span32(grid, 2)),
                  "%s: one image (map or volume) must span fewer than 2^31 elements", op);
    const int g = grid_for(tiles * 64, 1 << 16);                 // one tile pair per wave (143 -> 136 us vs 2048 workgroups)
    scatter_tile_kernel<SPH><<<g, kBlock, 0, st>>>(D, view4(depth), vcd, vfl, vgrid, view5(voxel), view5(cnt), empty_val,
                                                   fill_val);
    GENRE_LAUNCH_CHECK("projection forward");
    normalise_tile_kernel<SPH><<<g, kBlock, 0, st>>>(D, view4(depth), vcd, vfl, vgrid, view5(voxel), view5(cnt),
                                                     post_scale, post_bias, post_mode);
    GENRE_LAUNCH_CHECK("safe divide");
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_back_projection_forward(const genre_tensor *depth, const genre_tensor *camdist,
                                             const genre_tensor *fl, const genre_tensor *voxel,
                                             const genre_tensor *cnt, void *stream)
{
    return forward_impl<false>("back_projection_forward", depth, camdist, fl, nullptr, voxel, cnt, stream);
}

extern "C" int genre_spherical_back_proj_forward(const genre_tensor *depth, const genre_tensor *grid_in,
                                                 const genre_tensor *voxel, const genre_tensor *cnt,
                                                 void *stream)
{
    return forward_impl<true>("spherical_back_proj_forward", depth, nullptr, nullptr, grid_in, voxel, cnt, stream);
}

static int backward_impl(const char *op, const genre_tensor *depth, const genre_tensor *fl,
                         const genre_tensor *camdist, const genre_tensor *cnt, const genre_tensor *grad_in,
                         const genre_tensor *grad_depth, const genre_tensor *grad_camdist,
                         const genre_tensor *grad_fl, void *stream, bool shifted, const genre_tensor *zero_words = nullptr,
                         int64_t word_stride = 0, int64_t word_offset = 0, int group = 1)
{
    Dims D{};
    if (!check_image(op, depth, D) || !check_scalar(op, "fl", fl, D) || !check_scalar(op, "camdist", camdist, D) ||
        !check_volume(op, "cnt", cnt, D, true) || !check_volume(op, "grad_in", grad_in, D, false) ||
        !check_map(op, "grad_depth", grad_depth, D) || !check_scalar(op, "grad_camdist", grad_camdist, D) ||
        !check_scalar(op, "grad_fl", grad_fl, D))
        return 0;
    hipStream_t st = (hipStream_t)stream;
    const int imgs = D.N * D.NC;
    if (imgs == 0) return 1;
    zero2_kernel<<<ceil_div(imgs, 256), 256, 0, st>>>(view2(grad_camdist), view2(grad_fl), D.N, D.NC);
    GENRE_LAUNCH_CHECK("zero grads");
    const int npix = D.H * D.W;
    if (npix == 0) return 1;
    GENRE_REQUIRE(imgs <= 65535, "%s: N*NC must be <= 65535", op);
    GENRE_REQUIRE(D.N <= 65535 && D.NC <= 65535, "%s: N and NC must be <= 65535", op);
    int bx = ceil_div(npix, kBlock * kBwdUnroll);                       // ~512 workgroups in total (see the kernel)
    const int cap = imgs >= 512 ? 1 : 512 / imgs;
    if (bx > cap) bx = cap;
    float gscale = 1.0f;
    if (shifted) {
        GENRE_REQUIRE(D.X == D.Y && D.Y == D.Z, "%s: the fused shift needs a cubic grid", op);
        gscale = -(float)D.X;
    }
    ZeroHint zh{nullptr, 0, 0, 1};
    if (zero_words != nullptr) {
        GENRE_REQUIRE(group >= 1 && word_stride >= 0 && word_offset >= 0 && is_i32(zero_words, 1) && is_contiguous(zero_words) &&
                          zero_words->size[0] > (int64_t)((imgs - 1) / group) * word_stride + word_offset,
                      "%s: zero_words must be a contiguous int32 buffer holding word (image / group) * stride + offset of every image", op);
        zh = ZeroHint{(const int *)zero_words->data, word_stride, word_offset, group};
    }
    cam_backward_kernel<kBlock><<<dim3(bx, D.N, D.NC), kBlock, 0, st>>>(D, view4(depth), view2(fl), view2(camdist), view5(cnt),
                                                                        view5(grad_in), view4(grad_depth), view2(grad_camdist),
                                                                        view2(grad_fl), gscale, zh);
    GENRE_LAUNCH_CHECK("projection backward");
    return 1;
}

extern "C" int genre_back_projection_backward(const genre_tensor *depth, const genre_tensor *fl,
                                              const genre_tensor *camdist, const genre_tensor *cnt,
                                              const genre_tensor *grad_in, const genre_tensor *grad_depth,
                                              const genre_tensor *grad_camdist, const genre_tensor *grad_fl,
                                              void *stream)
{
    return backward_impl("back_projection_backward", depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist,
                         grad_fl, stream, false);
}

extern "C" int genre_back_projection_forward_shifted(const genre_tensor *depth, const genre_tensor *camdist,
                                                     const genre_tensor *fl, const genre_tensor *voxel,
                                                     const genre_tensor *cnt, void *stream)
{
    return forward_impl<false>("back_projection_forward_shifted", depth, camdist, fl, nullptr, voxel, cnt, stream, true);
}

extern "C" int genre_back_projection_forward_const(const genre_tensor *depth, const genre_tensor *voxel,
                                                   const genre_tensor *cnt, const genre_tensor *tile_live, float camdist,
                                                   float fl, int shifted, void *stream)
{
    const float byval[2] = {fl, camdist};
    return forward_impl<false>("back_projection_forward_const", depth, nullptr, nullptr, nullptr, voxel, cnt, stream,
                               (shifted & 1) != 0, byval, tile_live, (shifted & 2) != 0);
}

extern "C" int genre_cam_cell(void) { return kQX * 10000 + kQY * 100 + kQZ; }

extern "C" int genre_cam_forward_plan(const genre_tensor *voxel, const genre_tensor *cnt, float camdist, float fl)
{
    const char *op = "cam_forward_plan";
    Dims D{};
    GENRE_REQUIRE(is_f32(voxel, 5) && is_f32(cnt, 5) && same_shape(voxel, cnt), "%s: voxel and cnt must be 5-D fp32 tensors of one shape", op);
    D.N = (int)voxel->size[0]; D.NC = (int)voxel->size[1];
    D.X = (int)voxel->size[2]; D.Y = (int)voxel->size[3]; D.Z = (int)voxel->size[4];
    return byval_plan(D, voxel, cnt, fl, camdist);
}

extern "C" int genre_back_projection_backward_shifted(const genre_tensor *depth, const genre_tensor *fl,
                                                      const genre_tensor *camdist, const genre_tensor *cnt,
                                                      const genre_tensor *grad_in, const genre_tensor *grad_depth,
                                                      const genre_tensor *grad_camdist,
                                                      const genre_tensor *grad_fl, void *stream)
{
    return backward_impl("back_projection_backward_shifted", depth, fl, camdist, cnt, grad_in, grad_depth,
                         grad_camdist, grad_fl, stream, true);
}

extern "C" int genre_back_projection_backward_hinted(const genre_tensor *depth, const genre_tensor *fl,
                                                     const genre_tensor *camdist, const genre_tensor *cnt,
                                                     const genre_tensor *grad_in, const genre_tensor *grad_depth,
                                                     const genre_tensor *grad_camdist, const genre_tensor *grad_fl,
                                                     const genre_tensor *zero_words, int64_t word_stride, int64_t word_offset,
                                                     int group, int shifted, void *stream)
{
    return backward_impl("back_projection_backward_hinted", depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist, grad_fl,
                         stream, shifted != 0, zero_words, word_stride, word_offset, group);
}

extern "C" int genre_get_surface_mask(const genre_tensor *depth, const genre_tensor *camdist,
                                      const genre_tensor *fl, const genre_tensor *cnt,
                                      const genre_tensor *mask, void *stream)
{
    const char *op = "get_surface_mask";
    Dims D{};
    if (!check_image(op, depth, D) || !check_scalar(op, "camdist", camdist, D) || !check_scalar(op, "fl", fl, D) ||
        !check_volume(op, "mask", mask, D, true) || !check_volume(op, "cnt", cnt, D, false))
        return 0;
    const int64_t total = (int64_t)D.N * D.NC * D.X * D.Y * D.Z;
    if (total == 0) return 1;
    surface_mask_kernel<<<grid_for(total, 1 << 20), kBlock, 0, (hipStream_t)stream>>>(D, view4(depth), view2(camdist),
                                                                              view2(fl), view5(cnt), view5(mask));
    GENRE_LAUNCH_CHECK("surface mask");
    return 1;
}

static int sph_backward_impl(const char *op, const genre_tensor *depth, const genre_tensor *grid_in,
                             const genre_tensor *cnt, const genre_tensor *grad_in, const genre_tensor *grad_depth,
                             void *stream, bool shifted)
{
    Dims D{};
    if (!check_image(op, depth, D)) return 0;
    GENRE_REQUIRE(is_f32(grid_in, 5) && grid_in->size[0] == D.N && grid_in->size[1] == D.NC &&
                      grid_in->size[2] == D.H && grid_in->size[3] == D.W && grid_in->size[4] == 3,
                  "%s: grid must be a 5-D fp32 tensor [%d,%d,%d,%d,3]", op, D.N, D.NC, D.H, D.W);
    if (!check_volume(op, "cnt", cnt, D, true) || !check_volume(op, "grad_in", grad_in, D, false) ||
        !check_map(op, "grad_depth", grad_depth, D))
        return 0;
    const int64_t npix = (int64_t)D.N * D.NC * D.H * D.W;
    if (npix == 0) return 1;
    float gscale = 1.0f;
    if (shifted) {
        GENRE_REQUIRE(D.X == D.Y && D.Y == D.Z, "%s: the fused glue needs a cubic grid", op);
        gscale = -(float)D.X;
    }
    sph_backward_kernel<<<grid_for(npix), kBlock, 0, (hipStream_t)stream>>>(D, view4(depth), view5(grid_in),
                                                                            view5(cnt), view5(grad_in),
                                                                            view4(grad_depth), gscale);
    GENRE_LAUNCH_CHECK("spherical projection backward");
    return 1;
}

extern "C" int genre_spherical_back_proj_backward(const genre_tensor *depth, const genre_tensor *grid_in,
                                                  const genre_tensor *cnt, const genre_tensor *grad_in,
                                                  const genre_tensor *grad_depth, void *stream)
{
    return sph_backward_impl("spherical_back_proj_backward", depth, grid_in, cnt, grad_in, grad_depth, stream, false);
}

extern "C" int genre_spherical_back_proj_forward_shifted(const genre_tensor *depth, const genre_tensor *grid_in,
                                                         const genre_tensor *voxel, const genre_tensor *cnt,
                                                         void *stream)
{
    return forward_impl<true>("spherical_back_proj_forward_shifted", depth, nullptr, nullptr, grid_in, voxel, cnt, stream,
                              true);
}

extern "C" int genre_spherical_back_proj_backward_shifted(const genre_tensor *depth, const genre_tensor *grid_in,
                                                          const genre_tensor *cnt, const genre_tensor *grad_in,
                                                          const genre_tensor *grad_depth, void *stream)
{
    return sph_backward_impl("spherical_back_proj_backward_shifted", depth, grid_in, cnt, grad_in, grad_depth, stream,
                             true);
}
