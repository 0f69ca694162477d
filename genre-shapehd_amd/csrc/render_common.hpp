// render_common.hpp -- what the renderers of the standard (NCXYZ) layout share: dimensions, the sample position and cell
// arithmetic of the reference (toolbox/spherical_proj.py:50-56; ATen grid_sampler_3d, align_corners=True == PyTorch 0.4.1),
// the sph_pad fan-out (spherical_proj.py:21-28) and the argument checks.  Included by sph_render.hip (per-sample scratch, the
// backward) and sph_render_seg.hip (per-segment forward).
#pragma once
#include "common.hpp"

#pragma clang fp contract(off)

namespace genre {

namespace {

constexpr int kBrick = 16;                       // brick edge (voxels)
constexpr int kTile = kBrick + 2;                 // + one voxel of halo on either side
constexpr int kTile3 = kTile * kTile * kTile;

struct RenderDims {
    int N, NC, X, Y, Z, R, ZR;
    int sx, sy, sz;                              // element strides of one image's volume (fit in int)
    double step;                                 // 1/(ZR-1)
    float lo, hi;                                // clamp bounds of spherical_proj.py:66
    float pre_scale;                             // != 0: the volume is clamp(vox * pre_scale, lo, hi), formed on the
                                                 //       fly (the caller's `clamp(proj * 50, 1e-5, 1 - 1e-5)` folded in)
    int pad;                                     // > 0: the map is written / read as sph_pad(map, pad) would lay it out
};

// sph_pad (spherical_proj.py:21-28) as a fan-out of map pixel (i, j): replicate padding repeats the first / last
// row pad more times (:23); the left margin is then overwritten by the last pad interior columns and the right
// margin by the first pad ones (:25-26, azimuth wraps around), rows included.  Output rows r_lo .. r_lo+r_n-1,
// column c0 and (if >= 0) c1.  Needs 2*pad <= R.
__device__ __forceinline__ void pad_span(int R, int pm, int i, int j, int &r_lo, int &r_n, int &c0, int &c1)
{
    r_lo = (i == 0) ? 0 : i + pm;
    r_n = ((i == R - 1) ? R - 1 + 2 * pm : i + pm) - r_lo + 1;
    c0 = j + pm;
    c1 = (j >= R - pm) ? j - (R - pm) : (j < pm ? j + R + pm : -1);
}

// sample k of the ray with doubled direction 2*dir (fp64): spherical_proj.py:50-56
__device__ __forceinline__ void sample_pos(const RenderDims &D, double dx2, double dy2, double dz2, int k,
                                           float &gx, float &gy, float &gz)
{
    const double alpha = (k == D.ZR - 1) ? 1.0 : (double)k * D.step;       // numpy.linspace(0,1,ZR)[k]
    const double a = 1.0 - alpha;
    gx = (float)(dx2 * a); gy = (float)(dy2 * a); gz = (float)(dz2 * a);
}

// ATen grid_sampler_3d coordinates (align_corners=True): base corner + weights of the two corners
// per axis.  x -> X axis, y -> Y, z -> Z (vox.permute(0,1,4,3,2) in spherical_proj.py:64).
struct Cell { int x0, y0, z0; float wx0, wx1, wy0, wy1, wz0, wz1; };

__device__ __forceinline__ bool locate(const RenderDims &D, float gx, float gy, float gz, Cell &c)
{
    const float ix = ((gx + 1.f) / 2) * (D.X - 1);
    const float iy = ((gy + 1.f) / 2) * (D.Y - 1);
    const float iz = ((gz + 1.f) / 2) * (D.Z - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    c.x0 = (int)fx; c.y0 = (int)fy; c.z0 = (int)fz;
    c.wx1 = ix - fx; c.wy1 = iy - fy; c.wz1 = iz - fz;                       // weight of the +1 corner
    c.wx0 = (fx + 1) - ix; c.wy0 = (fy + 1) - iy; c.wz0 = (fz + 1) - iz;
    // at least one of the 8 corners inside the volume?
    return c.x0 >= -1 && c.x0 < D.X && c.y0 >= -1 && c.y0 < D.Y && c.z0 >= -1 && c.z0 < D.Z;
}

// ATen corner order: tnw, tne, tsw, tse, bnw, bne, bsw, bse (t/b: z, n/s: y, w/e: x); weight
// products evaluated left to right as ATen does.
__device__ __forceinline__ float corner_w(const Cell &c, int i)
{
    const float wx = (i & 1) ? c.wx1 : c.wx0, wy = (i & 2) ? c.wy1 : c.wy0, wz = (i & 4) ? c.wz1 : c.wz0;
    return wx * wy * wz;
}

inline int check_render(const char *op, const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *dw,
                 const genre_tensor *map, RenderDims &D)
{
    GENRE_REQUIRE(is_f32(vox, 5), "%s: vox must be a 5-D fp32 tensor [N,NC,X,Y,Z]", op);
    GENRE_REQUIRE(dirs && dirs->ndim == 3 && dirs->size[0] >= 0, "%s: dirs must be a 3-D tensor", op);
    D.N = (int)vox->size[0]; D.NC = (int)vox->size[1];
    D.X = (int)vox->size[2]; D.Y = (int)vox->size[3]; D.Z = (int)vox->size[4];
    D.R = (int)dirs->size[0];
    // the map is [N,NC,R,R], or [N,NC,R+2p,R+2p] laid out as sph_pad(map, p) (spherical_proj.py:21-28)
    GENRE_REQUIRE(is_f32(map, 4) && map->size[0] == vox->size[0] && map->size[1] == vox->size[1] &&
                      map->size[2] == map->size[3] && map->size[2] >= D.R && ((map->size[2] - D.R) & 1) == 0 &&
                      (map->size[2] - D.R) <= D.R,
                  "%s: the spherical map must be a 4-D fp32 tensor [N,NC,R+2p,R+2p] with 0 <= 2p <= R = %d", op, D.R);
    D.pad = (int)(map->size[2] - D.R) / 2;
    int64_t span = 1;
    for (int i = 2; i < 5; i++) {
        GENRE_REQUIRE(vox->stride[i] >= 0, "%s: negative vox strides are not supported", op);
        span += (vox->size[i] - 1) * vox->stride[i];
    }
    GENRE_REQUIRE(span < ((int64_t)1 << 31), "%s: one image's volume must span < 2^31 elements", op);
    D.sx = (int)vox->stride[2]; D.sy = (int)vox->stride[3]; D.sz = (int)vox->stride[4];
    // dirs: [R,R,6] fp32 words = [R,R,3] float64 unit directions (the caller passes the raw storage)
    GENRE_REQUIRE(dirs && dirs->data && dirs->ndim == 3 && dirs->size[0] == D.R && dirs->size[1] == D.R &&
                      dirs->size[2] == 6 && is_contiguous(dirs) && ((uintptr_t)dirs->data & 7u) == 0,
                  "%s: dirs must be the contiguous float64 [R,R,3] direction table viewed as fp32 [R,R,6]", op);
    GENRE_REQUIRE(is_f32(dw, 1) && is_contiguous(dw) && dw->size[0] >= 1, "%s: depth_weight must be a 1-D fp32 tensor", op);
    D.ZR = (int)dw->size[0];
    D.step = D.ZR > 1 ? 1.0 / (double)(D.ZR - 1) : 0.0;
    D.lo = 1e-5f; D.hi = (float)(1 - 1e-5);                              // spherical_proj.py:66
    return 1;
}

}  // namespace
}  // namespace genre
