// Driver around the reference's cam_bp kernel bodies (sliced into
// _ref/cam_bp_slice.inc by build_ref.py; the slice is reference code and is
// never committed).  The reference's host wrappers use THC and <<<>>> and are
// not buildable here; this driver reproduces ONLY their fills and launch
// arguments (back_projection_kernel.cu:629-963) for dense row-major tensors.
// Test infrastructure only.
#include <cstdint>
#include <cstddef>
#include "cuda_host_shim.h"
#include "cam_bp_slice.inc"

namespace {
struct S5 { int n, c, x, y, z; };
inline S5 dense5(int NC, int X, int Y, int Z) { return {NC * X * Y * Z, X * Y * Z, Y * Z, Z, 1}; }
}

extern "C" {

// wrapper :760-838 (+ Python prefill cam_back_projection.py:22-24)
void ref_back_projection_forward(float *depth, int N, int NC, int H, int W,
                                 float *camdist, float *fl, float *voxel, float *cnt,
                                 int X, int Y, int Z)
{
    size_t tot = (size_t)N * NC * X * Y * Z;
    float prefill = (float)(1.0 / (double)X);
    for (size_t i = 0; i < tot; i++) { voxel[i] = 0.0f + prefill; cnt[i] = 0.0f; }
    S5 s = dense5(NC, X, Y, Z);
    back_projection_forward_kernel(depth, N, NC, H, W, NC * H * W, H * W, W, 1,
                                   camdist, NC, 1, fl, NC, 1,
                                   voxel, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                   cnt, s.n, s.c, s.x, s.y, s.z, N * NC * H * W);
    inplace_safe_divide(voxel, N, NC, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                        cnt, s.n, s.c, s.x, s.y, s.z, 1.0f, (int)tot);
}

// wrapper :897-963
void ref_back_projection_backward(float *depth, int N, int NC, int H, int W,
                                  float *fl, float *camdist, float *cnt, float *grad_in,
                                  int X, int Y, int Z,
                                  float *grad_depth, float *grad_camdist, float *grad_fl)
{
    for (size_t i = 0; i < (size_t)N * NC * H * W; i++) grad_depth[i] = 0.0f;
    for (int i = 0; i < N * NC; i++) { grad_camdist[i] = 0.0f; grad_fl[i] = 0.0f; }
    S5 s = dense5(NC, X, Y, Z);
    back_projection_backward_kernel(depth, N, NC, H, W, NC * H * W, H * W, W, 1,
                                    fl, NC, 1, camdist, NC, 1,
                                    cnt, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                    grad_in, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                    grad_depth, NC * H * W, H * W, W, 1,
                                    grad_camdist, NC, 1, grad_fl, NC, 1, N * NC * H * W);
}

// wrapper :840-891
void ref_get_surface_mask(float *depth, int N, int NC, int H, int W,
                          float *camdist, float *fl, float *cnt, float *mask,
                          int X, int Y, int Z)
{
    size_t tot = (size_t)N * NC * X * Y * Z;
    for (size_t i = 0; i < tot; i++) mask[i] = 1.0f;
    S5 s = dense5(NC, X, Y, Z);
    get_surface_mask_kernel(depth, N, NC, H, W, NC * H * W, H * W, W, 1,
                            camdist, NC, 1, fl, NC, 1,
                            cnt, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                            mask, s.n, s.c, s.x, s.y, s.z, (int)tot);
}

// wrapper :629-703 (+ Python fills sperical_to_tdf.py:23-25); grid strides explicit
void ref_spherical_back_proj_forward(float *depth, int N, int NC, int H, int W,
                                     float *grid, const int64_t *gs,
                                     float *voxel, float *cnt, int X, int Y, int Z)
{
    size_t tot = (size_t)N * NC * X * Y * Z;
    for (size_t i = 0; i < tot; i++) { voxel[i] = 0.0f; cnt[i] = 0.0f; }
    S5 s = dense5(NC, X, Y, Z);
    spherical_back_projection_forward_kernel(depth, N, NC, H, W, NC * H * W, H * W, W, 1,
                                             grid, (int)gs[0], (int)gs[1], (int)gs[2], (int)gs[3], (int)gs[4],
                                             voxel, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                             cnt, s.n, s.c, s.x, s.y, s.z, N * NC * H * W);
    inplace_safe_divide(voxel, N, NC, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                        cnt, s.n, s.c, s.x, s.y, s.z, 0.0f, (int)tot);
}

// wrapper :704-757 (+ Python zero fill sperical_to_tdf.py:39)
void ref_spherical_back_proj_backward(float *depth, int N, int NC, int H, int W,
                                      float *grid, const int64_t *gs, float *cnt, float *grad_in,
                                      int X, int Y, int Z, float *grad_depth)
{
    for (size_t i = 0; i < (size_t)N * NC * H * W; i++) grad_depth[i] = 0.0f;
    S5 s = dense5(NC, X, Y, Z);
    spherical_back_projection_backward_kernel(depth, N, NC, H, W, NC * H * W, H * W, W, 1,
                                              grid, (int)gs[0], (int)gs[1], (int)gs[2], (int)gs[3], (int)gs[4],
                                              cnt, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                              grad_in, X, Y, Z, s.n, s.c, s.x, s.y, s.z,
                                              grad_depth, NC * H * W, H * W, W, 1, N * NC * H * W);
}

}  // extern "C"
