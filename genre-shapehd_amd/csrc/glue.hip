// glue.hip -- GenRe caller glue folded into single passes.
//
// get_abs_depth of models/depth_pred_with_sph_inpaint.py:131-142 is, in the reference, six PyTorch
// kernels over a [N,1,256,256] map (divide, 1-x, range multiply-add, silhouette divide + compare,
// masked assign, permute + flip copies): at batch 1 that is more launch time than the whole
// geometric chain that follows.  Here it is one pass whose output is already the transposed,
// row-flipped depth map cam_bp consumes.  Arithmetic is the reference's, op by op, un-fused fp32.
#include "common.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

constexpr int kTileDim = 32;

// out[n,c,i,j] = f(pred[n,c,j,W-1-i]); a 32x32 tile goes through LDS so that both the reads (along w)
// and the writes (along j) are coalesced
__global__ __launch_bounds__(kTileDim * 8) void abs_depth_forward_kernel(int NC, int H, int W, View4 pred, View2 mm,
                                                                          View4 sil, View4 out, float scale)
{
    __shared__ float tile[kTileDim][kTileDim + 1];
    const int img = blockIdx.z, n = img / NC, c = img % NC;
    const float dmin = mm.p[n * mm.s0], dmax = mm.p[n * mm.s0 + mm.s1];
    const float range = (dmax - dmin) + 1e-4f;                               // marrnetbase.py:150
    const int h0 = blockIdx.y * kTileDim, w0 = blockIdx.x * kTileDim;
    const int tx = threadIdx.x % kTileDim, ty = threadIdx.x / kTileDim;
    for (int r = ty; r < kTileDim; r += 8) {
        const int h = h0 + r, w = w0 + tx;
        float v = 0.f;
        if (h < H && w < W) {
            const float p = pred.p[n * pred.s0 + c * pred.s1 + h * pred.s2 + w * pred.s3];
            const float s = sil.p[n * sil.s0 + c * sil.s1 + h * sil.s2 + w * sil.s3];
            const float rel = 1.0f - p / scale;                              // :133-134, marrnetbase.py:139
            v = rel * range + dmin;                                          // marrnetbase.py:150
            if (s / scale < 0.5f) v = 0.f;                                   // :137-138
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    // out[i = W-1-w][j = h]
    for (int r = ty; r < kTileDim; r += 8) {
        const int w = w0 + r, h = h0 + tx;
        if (h < H && w < W) out.p[n * out.s0 + c * out.s1 + (W - 1 - w) * out.s2 + h * out.s3] = tile[tx][r];
    }
}

__global__ __launch_bounds__(kTileDim * 8) void abs_depth_backward_kernel(int NC, int H, int W, View4 gout, View2 mm,
                                                                           View4 sil, View4 gpred, float scale)
{
    __shared__ float tile[kTileDim][kTileDim + 1];
    const int img = blockIdx.z, n = img / NC, c = img % NC;
    const float dmin = mm.p[n * mm.s0], dmax = mm.p[n * mm.s0 + mm.s1];
    const float range = (dmax - dmin) + 1e-4f;
    const int h0 = blockIdx.y * kTileDim, w0 = blockIdx.x * kTileDim;
    const int tx = threadIdx.x % kTileDim, ty = threadIdx.x / kTileDim;
    for (int r = ty; r < kTileDim; r += 8) {                                 // coalesced along j = h
        const int w = w0 + r, h = h0 + tx;
        tile[r][tx] = (h < H && w < W) ? gout.p[n * gout.s0 + c * gout.s1 + (W - 1 - w) * gout.s2 + h * gout.s3] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < kTileDim; r += 8) {
        const int h = h0 + r, w = w0 + tx;
        if (h < H && w < W) {
            const float s = sil.p[n * sil.s0 + c * sil.s1 + h * sil.s2 + w * sil.s3];
            const float g = tile[tx][r];
            // autograd of the reference's ops: mul by range, negate (1 - x), divide by scale; masked -> 0
            const float v = (s / scale < 0.5f) ? 0.f : (0.0f - g * range) / scale;
            gpred.p[n * gpred.s0 + c * gpred.s1 + h * gpred.s2 + w * gpred.s3] = v;
        }
    }
}

int check_maps(const char *op, const genre_tensor *a, const genre_tensor *mm, const genre_tensor *sil,
               const genre_tensor *t, const char *tname, int &N, int &NC, int &H, int &W)
{
    GENRE_REQUIRE(is_f32(a, 4), "%s: the map must be a 4-D fp32 tensor [N,NC,H,W]", op);
    N = (int)a->size[0]; NC = (int)a->size[1]; H = (int)a->size[2]; W = (int)a->size[3];
    GENRE_REQUIRE(is_f32(sil, 4) && sil->size[0] == N && sil->size[1] == NC && sil->size[2] == H && sil->size[3] == W,
                  "%s: silhou must be a 4-D fp32 tensor [%d,%d,%d,%d]", op, N, NC, H, W);
    GENRE_REQUIRE(is_f32(mm, 2) && mm->size[0] == N && mm->size[1] == 2, "%s: depth_minmax must be fp32 [%d,2]", op, N);
    GENRE_REQUIRE(is_f32(t, 4) && t->size[0] == N && t->size[1] == NC && t->size[2] == W && t->size[3] == H,
                  "%s: %s must be a 4-D fp32 tensor [%d,%d,%d,%d] (transposed map)", op, tname, N, NC, W, H);
    GENRE_REQUIRE((int64_t)N * NC <= 65535, "%s: N*NC must be <= 65535", op);
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_abs_depth_forward(const genre_tensor *pred_depth, const genre_tensor *depth_minmax,
                                       const genre_tensor *silhou, const genre_tensor *out, float scale_25d,
                                       void *stream)
{
    const char *op = "abs_depth_forward";
    int N, NC, H, W;
    if (!check_maps(op, pred_depth, depth_minmax, silhou, out, "out", N, NC, H, W)) return 0;
    GENRE_REQUIRE(scale_25d != 0.0f, "%s: scale_25d must be non-zero", op);
    if ((int64_t)N * NC * H * W == 0) return 1;
    const dim3 grid((W + kTileDim - 1) / kTileDim, (H + kTileDim - 1) / kTileDim, N * NC);
    abs_depth_forward_kernel<<<grid, kTileDim * 8, 0, (hipStream_t)stream>>>(NC, H, W, view4(pred_depth), view2(depth_minmax),
                                                                            view4(silhou), view4(out), scale_25d);
    GENRE_LAUNCH_CHECK("abs depth");
    return 1;
}

extern "C" int genre_abs_depth_backward(const genre_tensor *grad_out, const genre_tensor *depth_minmax,
                                        const genre_tensor *silhou, const genre_tensor *grad_pred, float scale_25d,
                                        void *stream)
{
    const char *op = "abs_depth_backward";
    int N, NC, H, W;
    if (!check_maps(op, grad_pred, depth_minmax, silhou, grad_out, "grad_out", N, NC, H, W)) return 0;
    GENRE_REQUIRE(scale_25d != 0.0f, "%s: scale_25d must be non-zero", op);
    if ((int64_t)N * NC * H * W == 0) return 1;
    const dim3 grid((W + kTileDim - 1) / kTileDim, (H + kTileDim - 1) / kTileDim, N * NC);
    abs_depth_backward_kernel<<<grid, kTileDim * 8, 0, (hipStream_t)stream>>>(NC, H, W, view4(grad_out), view2(depth_minmax),
                                                                             view4(silhou), view4(grad_pred), scale_25d);
    GENRE_LAUNCH_CHECK("abs depth backward");
    return 1;
}
