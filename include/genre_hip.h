/*
 * genre_hip.h -- C ABI of libgenre_hip.so: MI355X (gfx950) kernels for the
 * geometric hot path of GenRe / ShapeHD.
 *
 * This is the drop-in boundary.  Each entry point replaces one cffi-exported C
 * function of the reference (xiumingzhang/GenRe-ShapeHD); argument ORDER is the
 * reference's, `THCudaTensor*` becomes `const genre_tensor*` (a plain
 * pointer + sizes + element strides descriptor -- no torch types), and a
 * `hipStream_t` (passed as void*) is appended.  Conventions kept from the
 * reference:
 *   - the caller allocates every output (cam_back_projection.py:22-25,39-45);
 *     the library never allocates device memory and keeps no mutable global state (one
 *     read-only environment setting, GENRE_CAMBP_MODE=scatter|brick|gather, is looked up once);
 *   - return 1 on success, 0 on failure (back_projection.c:13-15 turns 0 into
 *     THError("aborting")); on 0, genre_last_error() returns a thread-local
 *     message -- shape/dtype violations that THArgCheck / THCUNN_check_dim_size
 *     (back_projection_kernel.cu:85-97,105-184) would have raised, or the HIP
 *     launch error string;
 *   - all work is enqueued asynchronously on `stream` (the reference used THC's
 *     current stream, back_projection_kernel.cu:647; its nndistance launcher
 *     ignored the stream, nnd_cuda.cu:130-131 -- here every op honours it);
 *     no host synchronisation inside any call; re-entrant and thread-safe.
 *   - cam_bp / calc_prob tensors may have arbitrary element strides (the
 *     reference kernels are stride-generic; an expand()ed `grid` with batch
 *     stride 0 is legal); nndistance tensors must be contiguous (nnd.py:16).
 * Outputs are fully defined by the call: every element of every output tensor
 * is written (the fills the reference did in Python / THCudaTensor_zero are
 * part of the op), so callers may pass uninitialised memory.
 *
 * All data is fp32 (GENRE_F32) except nndistance indices (GENRE_I32).
 */
#ifndef GENRE_HIP_H
#define GENRE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENRE_ABI_VERSION 5
#define GENRE_MAX_DIMS 5

enum { GENRE_F32 = 0, GENRE_I32 = 1 };

/* Tensor view: device pointer, sizes and ELEMENT strides (not bytes). */
typedef struct genre_tensor {
    void   *data;
    int32_t ndim;
    int32_t dtype;                   /* GENRE_F32 | GENRE_I32 */
    int64_t size[GENRE_MAX_DIMS];
    int64_t stride[GENRE_MAX_DIMS];
} genre_tensor;

/* ABI version of the loaded library (== GENRE_ABI_VERSION it was built with). */
int genre_abi_version(void);

/* Thread-local description of the last failure on this thread ("" if none). */
const char *genre_last_error(void);

/* ---- cam_bp : toolbox/cam_bp/cam_bp/src/back_projection.h:1-5 ------------- */

/* Replaces back_projection_forward (back_projection.c:9-17 ->
 * back_projection_forward_wrap, back_projection_kernel.cu:760-838; kernels
 * :200-276 and :282-306) INCLUDING the Python prefill of
 * cam_back_projection.py:22-24.
 *   depth [N,NC,H,W]  camdist [N,NC]  fl [N,NC]  ->  voxel, cnt [N,NC,X,Y,Z]
 * voxel = mean distance of the back-projected points of a voxel to its centre,
 * 1/max(X,Y,Z) where no point fell; cnt = number of points (integer-valued).
 * Volumes with contiguous, 16-byte aligned z rows (the reference's dense tensors) take one launch of the
 * LDS-brick kernel (deterministic); other layouts take fill + scatter + normalise (float atomics).  cnt is exact
 * and voxels hit by one point are bit-identical to the reference either way. */
int genre_back_projection_forward(const genre_tensor *depth, const genre_tensor *camdist,
                                  const genre_tensor *fl, const genre_tensor *voxel,
                                  const genre_tensor *cnt, void *stream);

/* Replaces back_projection_backward (back_projection.c:20-28 -> :897-963,
 * kernel :366-471).  NOTE fl/camdist swap places relative to forward, as in the
 * reference.  grad_depth [N,NC,H,W], grad_camdist / grad_fl [N,NC] are fully
 * written (zero where the reference leaves its zero fill).  camdist is read
 * with its own strides (the reference's :401 uses the cnt strides -- an
 * out-of-bounds read for n>0 that is not reproduced). */
int genre_back_projection_backward(const genre_tensor *depth, const genre_tensor *fl,
                                   const genre_tensor *camdist, const genre_tensor *cnt,
                                   const genre_tensor *grad_in, const genre_tensor *grad_depth,
                                   const genre_tensor *grad_camdist, const genre_tensor *grad_fl,
                                   void *stream);

/* Extensions: the same two ops with Camera_back_projection_layer.shift_tdf
 * (camera_backprojection_module.py:25-28) folded in: forward writes
 * voxel = 1 - res*tdf (res = X = Y = Z; 0 where empty), backward takes the gradient
 * w.r.t. that shifted output.  Saves one full-volume elementwise pass each way. */
int genre_back_projection_forward_shifted(const genre_tensor *depth, const genre_tensor *camdist,
                                          const genre_tensor *fl, const genre_tensor *voxel,
                                          const genre_tensor *cnt, void *stream);
int genre_back_projection_backward_shifted(const genre_tensor *depth, const genre_tensor *fl,
                                           const genre_tensor *camdist, const genre_tensor *cnt,
                                           const genre_tensor *grad_in, const genre_tensor *grad_depth,
                                           const genre_tensor *grad_camdist, const genre_tensor *grad_fl,
                                           void *stream);

/* Extension: genre_back_projection_backward[_shifted] (shifted != 0) for a grad_in whose PRODUCER knows that the gradient of some
 * images is identically zero -- the renderer's backward where the caller's clamp in front of it blocks every voxel (GenRe's own
 * chain, depth_pred_with_sph_inpaint.py:124): it still writes those zeros to grad_in, and says so in words it owns anyway.
 * zero_words int32, contiguous: word [(image / group) * word_stride + word_offset] == 0 <=> the gradient of that image (N*NC
 * order) is all zeros: its grad_depth is written as zeros without reading grad_in, cnt or depth.  Other images: as without the
 * words.  (toolbox/_fused_render.py hangs the words on the gradient tensor it returns, guarded by the tensor's version.) */
int genre_back_projection_backward_hinted(const genre_tensor *depth, const genre_tensor *fl, const genre_tensor *camdist,
                                          const genre_tensor *cnt, const genre_tensor *grad_in,
                                          const genre_tensor *grad_depth, const genre_tensor *grad_camdist,
                                          const genre_tensor *grad_fl, const genre_tensor *zero_words, int64_t word_stride,
                                          int64_t word_offset, int group, int shifted, void *stream);

/* Extension: the camera forward with ONE focal length and ONE camera distance for every image, passed by value --
 * exactly what Camera_back_projection_layer fills its [N,NC] tensors with when it is called with Python floats
 * (camera_backprojection_module.py:16-21).  The kernel then has the camera in its arguments instead of behind two
 * loads (one dependent memory round trip less in front of the brick screen; batch-1 latency).  `shifted` is a bit set: bit 0 =
 * output as genre_back_projection_forward_shifted; bit 1 (leader pass only) = cnt is written ONLY where a point landed and is
 * undefined elsewhere -- for a caller that, like Camera_back_projection_layer, keeps cnt for genre_back_projection_backward*
 * alone (which reads it at the voxel of each in-grid pixel and nowhere else): half of the fill is then not written.  Two implementations, picked by the output layout:
 *  - dense NCXYZ outputs (unit z stride, 16-byte aligned rows, Z % 4 == 0): the single-launch brick kernel; results are
 *    identical to the tensor entry points called with constant-filled tensors;
 *  - any other layout (the image-minor volumes of the batch-minor renderer; rows that are not float4-aligned): fill + a LEADER
 *    pass without atomics (csrc/cam_bp.hip: cam_leader_kernel) -- every pixel sums the distances of the pixels of its
 *    (2H+1)^2 window that fall into its voxel, in row-major order, and the first of them writes: run-to-run deterministic,
 *    tdf and cnt bit-identical to a serial evaluation of the reference (back_projection_kernel.cu:215-305).  H = how many pixels
 *    apart two points of one voxel can project, bounded on the host from (fl, camdist, res); cameras with H > 4 (or closer than
 *    0.55 to the grid centre) return 0 -- pass tensors there (fill + scatter with float atomics + normalise).
 * (bit 1 of `shifted` is a permission: the brick kernel writes cnt densely whatever it says.)
 * tile_live (optional occupancy words for the renderer that consumes the volume; NULL: none), by implementation --
 * genre_cam_forward_plan() says which one a call will take:
 *  - leader pass: int32 [ceil(N/32), nbx, nby, nbz], contiguous.  The volume's images are cut into groups of 32 consecutive
 *    images (the batch-minor renderer's) and its voxels into nbx x nby x nbz bricks of ceil(X / nbx) x ... voxels; the op clears
 *    the tensor and sets word (g, b) = 1 iff some image of group g has a point in brick b OR in one of b's <= 7 neighbours on the
 *    high side (b + {0,1}^3) -- i.e. in any brick the TILE of b (the brick plus the voxels one step beyond its high faces: what a
 *    trilinear sampler of b's cells reads) can reach.  Consumer: genre_render_bm_forward.
 *  - brick kernel (ABI 5): int32 [N*NC, ceil(X/cx), ceil(Y/cy), ceil(Z/cz)], contiguous, (cx, cy, cz) = genre_cam_cell(): one
 *    word per image and cell, written by the workgroup that owns the cell (no clearing pass): 1 iff a point landed in the cell
 *    (or could have: a superset is legal).  Consumer: genre_render_seg_forward.
 * Every voxel of a tile / cell whose word is 0 holds the fill value (tdf: 1/res; shifted: 1 - res/res). */
int genre_back_projection_forward_const(const genre_tensor *depth, const genre_tensor *voxel,
                                        const genre_tensor *cnt, const genre_tensor *tile_live, float camdist, float fl,
                                        int shifted, void *stream);

/* cells of the brick kernel's occupancy words, as cx*10000 + cy*100 + cz voxels (80832 = 8 x 8 x 32) */
int genre_cam_cell(void);

/* which implementation genre_back_projection_forward_const takes for these outputs and this camera: 1 = brick kernel (dense
 * NCXYZ, float4-aligned z rows), 2 = fill + leader pass, 0 = the call would be refused (pass fl / camdist tensors) */
int genre_cam_forward_plan(const genre_tensor *voxel, const genre_tensor *cnt, float camdist, float fl);

/* Replaces get_surface_mask (back_projection.c:30-38 -> :840-891, kernel
 * :310-358).  mask [N,NC,X,Y,Z] := 1, except 0 for empty voxels (cnt <= 1e-5)
 * that lie behind the observed surface. */
int genre_get_surface_mask(const genre_tensor *depth, const genre_tensor *camdist,
                           const genre_tensor *fl, const genre_tensor *cnt,
                           const genre_tensor *mask, void *stream);

/* Replaces spherical_back_proj_forward (back_projection.c:40-48 -> :629-703,
 * kernels :475-542 and :282-306 with bias 0) INCLUDING the zero fills of
 * sperical_to_tdf.py:23-25.
 *   depth [N,NC,H,W]  grid_in [N,NC,H,W,3] (any strides)  ->  voxel, cnt */
int genre_spherical_back_proj_forward(const genre_tensor *depth, const genre_tensor *grid_in,
                                      const genre_tensor *voxel, const genre_tensor *cnt,
                                      void *stream);

/* Replaces spherical_back_proj_backward (back_projection.c:49-57 -> :704-757,
 * kernel :545-627).  grad_depth [N,NC,H,W] fully written. */
int genre_spherical_back_proj_backward(const genre_tensor *depth, const genre_tensor *grid_in,
                                       const genre_tensor *cnt, const genre_tensor *grad_in,
                                       const genre_tensor *grad_depth, void *stream);

/* Extensions: the same two ops with GenRe's caller-side glue (genre_full_model.py:139-142)
 * folded in: forward writes voxel = (-tdf + 1/res) * res * clamp(cnt, 0, 1) (0 where empty),
 * backward takes the gradient w.r.t. that.  voxel may be a strided view, e.g. channel 0 of the
 * [N,2,R,R,R] refiner input. */
int genre_spherical_back_proj_forward_shifted(const genre_tensor *depth, const genre_tensor *grid_in,
                                              const genre_tensor *voxel, const genre_tensor *cnt,
                                              void *stream);
int genre_spherical_back_proj_backward_shifted(const genre_tensor *depth, const genre_tensor *grid_in,
                                               const genre_tensor *cnt, const genre_tensor *grad_in,
                                               const genre_tensor *grad_depth, void *stream);

/* ---- calc_prob : toolbox/calc_prob/calc_prob/src/calc_prob.h:1-2 ---------- */

/* Extension: the glue of models/depth_pred_with_sph_inpaint.py:131-142 (`get_abs_depth`) in one pass --
 *   abs = (1 - pred_depth / scale_25d) * (dmax - dmin + 1e-4) + dmin      marrnetbase.py:138-151
 *   abs[silhou / scale_25d < 0.5] = 0                                      :138-139
 *   out = flip(abs.permute(0,1,3,2), [2])  ==  out[n,c,i,j] = abs[n,c,j,W-1-i]        :140-141
 * pred_depth, silhou [N,NC,H,W] (any strides); depth_minmax [N,2] = (min, max) per sample (:145-149);
 * out [N,NC,W,H], fully written: the depth map genre_back_projection_forward consumes.
 * Replaces six elementwise / copy kernels of the reference's PyTorch glue.  Un-fused fp32
 * arithmetic in the reference's order (true division by scale_25d): bit-identical to the
 * same lines run on CPU torch. */
int genre_abs_depth_forward(const genre_tensor *pred_depth, const genre_tensor *depth_minmax,
                            const genre_tensor *silhou, const genre_tensor *out, float scale_25d,
                            void *stream);

/* Adjoint of the above w.r.t. pred_depth (depth_minmax and silhou are detached in the
 * reference, :135,137): grad_pred[n,c,h,w] = masked ? 0 : (-(g * (dmax - dmin + 1e-4))) / scale_25d
 * with g = grad_out[n,c,W-1-w,h].  grad_pred [N,NC,H,W] fully written. */
int genre_abs_depth_backward(const genre_tensor *grad_out, const genre_tensor *depth_minmax,
                             const genre_tensor *silhou, const genre_tensor *grad_pred, float scale_25d,
                             void *stream);

/* Replaces calc_prob_forward (calc_prob.c:9-17 -> calc_prob_kernel.cu:191-226,
 * kernel :113-143).  prob_in, prob_out [N,NC,X,Y,Z]; rays run along Z:
 *   out[z] = in[z] * prod_{k<z} (1 - in[k])                                   */
int genre_calc_prob_forward(const genre_tensor *prob_in, const genre_tensor *prob_out,
                            void *stream);

/* Replaces calc_prob_backward (calc_prob.c:18-26 -> :227-266, kernel
 * :146-189).  stop_prob_weighted = stop_prob * grad (formed by the caller,
 * calc_prob.py:27):  grad_out[z] = w[z]/p[z] - (sum_{j>z} w[j]) / (1 - p[z])  */
int genre_calc_prob_backward(const genre_tensor *prob_in, const genre_tensor *stop_prob_weighted,
                             const genre_tensor *grad_out, void *stream);

/* Extension (no reference counterpart): same as genre_calc_prob_backward but
 * forms w = stop_prob * grad_in inside the kernel (fp32 product, as
 * calc_prob.py:27 does), saving one 3-tensor elementwise pass. */
int genre_calc_prob_backward_fused(const genre_tensor *prob_in, const genre_tensor *stop_prob,
                                   const genre_tensor *grad_in, const genre_tensor *grad_out,
                                   void *stream);

/* ---- nndistance : toolbox/nndistance/src/my_lib_cuda.h:1-4 ---------------- */

/* Replaces nnd_forward_cuda (my_lib_cuda.c:9-27 -> NmDistanceKernelLauncher,
 * nnd_cuda.cu:129-141; kernel :6-128) and, for values, the CPU nnd_forward
 * (my_lib.c:30-49).  xyz1 [B,n,3], xyz2 [B,m,3] contiguous fp32 ->
 * dist1 [B,n], dist2 [B,m] squared L2 to the nearest neighbour in the other
 * cloud; idx1, idx2 int32 its index (lowest index among equal distances). */
int genre_nnd_forward(const genre_tensor *xyz1, const genre_tensor *xyz2,
                      const genre_tensor *dist1, const genre_tensor *dist2,
                      const genre_tensor *idx1, const genre_tensor *idx2, void *stream);

/* Replaces nnd_backward_cuda (my_lib_cuda.c:30-54 -> NmDistanceGradKernelLauncher,
 * nnd_cuda.cu:163-177; kernel :143-162).  gradxyz1 [B,n,3], gradxyz2 [B,m,3]
 * fully written (the reference memsets them, :164-165). */
int genre_nnd_backward(const genre_tensor *xyz1, const genre_tensor *xyz2,
                       const genre_tensor *gradxyz1, const genre_tensor *gradxyz2,
                       const genre_tensor *graddist1, const genre_tensor *graddist2,
                       const genre_tensor *idx1, const genre_tensor *idx2, void *stream);

/* The vector width the host search runs at in this process: "scalar", "avx2" or "avx512" (picked from the CPU's features;
 * GENRE_NND_HOST_ISA pins a narrower one).  Results are bit-identical at every width. */
const char *genre_nnd_host_isa(void);

/* HOST entry points: the reference's my_lib.nnd_forward / nnd_backward (toolbox/nndistance/src/my_lib.h:3-5,
 * my_lib.c:30-118), which its NNDFunction takes for CPU tensors (functions/nnd.py:27-28,53-54).  All pointers are
 * HOST memory; same shapes as above; synchronous; multi-threaded over (batch item, block of queries).  Values and
 * indices equal the reference's single-threaded loop bit for bit (same float expression, first minimum wins,
 * serial gradient accumulation inside a batch item). */
int genre_nnd_forward_host(const genre_tensor *xyz1, const genre_tensor *xyz2,
                           const genre_tensor *dist1, const genre_tensor *dist2,
                           const genre_tensor *idx1, const genre_tensor *idx2);
int genre_nnd_backward_host(const genre_tensor *xyz1, const genre_tensor *xyz2,
                            const genre_tensor *gradxyz1, const genre_tensor *gradxyz2,
                            const genre_tensor *graddist1, const genre_tensor *graddist2,
                            const genre_tensor *idx1, const genre_tensor *idx2);

/* ---- render_spherical : toolbox/spherical_proj.py:31-72 (extension) ---------- */

/* No native counterpart in the reference: fuses the PyTorch op sequence of
 * render_spherical.forward (spherical_proj.py:62-72: expand, permute,
 * grid_sample [0.4.1 semantics == align_corners=True, zeros padding], clamp to
 * [1e-5, 1-1e-5], CalcStopProb, matmul(depth_weight), prod(1-p), add).
 * vox [N,NC,X,Y,Z] (any non-negative strides) -> out [N,NC,R,R] (or padded, see below).
 *   dirs         : the float64 [R,R,3] unit-direction table of spherical_proj.py:43-49,
 *                  passed as its raw storage viewed as fp32 [R,R,6] (contiguous);
 *                  sample k of ray (i,j) sits at float((2*dirs[i,j]) * (1 - k/(ZR-1))),
 *                  bit-identical to the reference's `grid` buffer (:50-56)
 *   depth_weight : [ZR] fp32, the reference's buffer of the same name (:57)
 * Optional geometry tables (all NULL = one wave-per-ray gather kernel):
 *   v_scratch  : fp32 [>= N*NC*R*R*ZR], 16-byte aligned; receives the raw (un-clamped)
 *                trilinear value of every in-volume sample, layout [ray][k]
 *   fwd_table  : int32 [rows,4] = (brick id, begin, end, 0): rows of fwd_chunks handled by
 *                one workgroup; brick id = (bx*nby+by)*nbz+bz over 16^3-voxel bricks
 *   fwd_chunks : int32 [S], entry (ray i*R+j) << 8 | k; every sample with at least one
 *                trilinear corner inside the volume appears exactly once, under the brick
 *                of its (clamped) base corner, sorted by (brick, ray, k)
 *   kin        : int32 [R*R], first such sample of each ray (they form a suffix)
 * pre_scale != 0 (brick path only): the rendered volume is
 * clamp(vox * pre_scale, 1e-5, 1-1e-5) formed on the fly -- the caller-side
 * `torch.clamp(proj * 50, 1e-5, 1 - 1e-5)` of depth_pred_with_sph_inpaint.py:124 folded
 * in (two full-volume elementwise passes less each way); backward then returns the
 * gradient w.r.t. the un-scaled, un-clamped vox.
 * Padded map (brick path only): `out` may be [N,NC,R+2p,R+2p] with 0 <= 2p <= R (R is taken
 * from dirs); it is then written as sph_pad(map, p) of spherical_proj.py:21-28 lays it out
 * (first/last row replicated, azimuth columns wrapped around) -- the `sph_pad(sph_in, margin)`
 * of depth_pred_with_sph_inpaint.py:126 folded in; backward accepts grad_out of that shape.
 * live (ABI 4; optional, used with pre_scale != 0 on the brick path): int32 [N*NC*(1 + bricks)], bricks = ceil(X/16) *
 * ceil(Y/16) * ceil(Z/16).  The forward clears it and sets live[img*(1+bricks)] = 1 if ANY voxel of image img passes the
 * pre_scale clamp (lo <= vox*pre_scale <= hi) and live[img*(1+bricks) + 1 + b] = 1 if any voxel of brick b does.  Handed to
 * genre_render_spherical_backward it lets that op skip what the clamp blocks: a dead image's scan pass returns at once, a
 * dead brick's voxels are written as zeros without reading the sample list.  On GenRe's own chain (clamp(proj * 50) of a
 * saturated-or-empty volume, depth_pred_with_sph_inpaint.py:124) every image is dead: the gradient is identically zero.
 * Reference builder of the tables: genre-shapehd_amd/toolbox/_fused_render.py. */
int genre_render_spherical_forward(const genre_tensor *vox, const genre_tensor *dirs,
                                   const genre_tensor *depth_weight, const genre_tensor *out,
                                   const genre_tensor *v_scratch, const genre_tensor *fwd_table,
                                   const genre_tensor *fwd_chunks, const genre_tensor *kin,
                                   const genre_tensor *live, float pre_scale, void *stream);

/* Adjoint of the above w.r.t. vox (what autograd derives for the reference's
 * op chain).  grad_out [N,NC,R,R] -> grad_vox [N,NC,X,Y,Z], fully written.
 * Requires ZR <= 256.  Modes:
 *  - dp_scratch, brick_table, chunk_list given: brick-owned accumulation without
 *    global atomics.  dp_scratch: fp32 [>= N*NC*R*R*ZR + N*NC], 16-byte aligned (receives
 *    dL/dp per sample and, behind them, each image's max|dL/dp| for its fixed-point scale;
 *    an image with a non-finite dL/dp gets NaN gradients, as the reference chain would).
 *    brick_table: int32 [rows,4] = (brick id, begin, end, mode), every brick in >= 1 row;
 *    mode 0: the row is the brick's only one, mode 1: the brick's samples are split over
 *    several rows.  chunk_list: as fwd_chunks, but a brick's rows cover every sample with
 *    at least one trilinear corner inside that brick (samples near faces appear in
 *    several bricks).  If v_scratch (as written by the forward) and kin are given too,
 *    the samples are not recomputed from vox.
 *    live (optional; needs v_scratch + kin and pre_scale != 0): the forward's pass words, see above -- where the clamp
 *    blocks everything the adjoint is a select of zeros, so a non-finite upstream gradient does not reach a dead image.
 *  - those pointers NULL: global-atomic scatter fallback (grad_vox must be
 *    contiguous, 16-byte aligned, numel % 4 == 0). */
int genre_render_spherical_backward(const genre_tensor *vox, const genre_tensor *dirs,
                                    const genre_tensor *depth_weight, const genre_tensor *grad_out,
                                    const genre_tensor *grad_vox, const genre_tensor *dp_scratch,
                                    const genre_tensor *brick_table, const genre_tensor *chunk_list,
                                    const genre_tensor *v_scratch, const genre_tensor *kin,
                                    const genre_tensor *live, float pre_scale, void *stream);

/* ---- segment renderer: the forward for the standard (NCXYZ) layout (ABI 5; csrc/sph_render_seg.hip) ----------------
 *
 * Same operator and arguments as genre_render_spherical_forward (vox, dirs, depth_weight, out, pre_scale, padded `out`, `live`),
 * but nothing per SAMPLE goes through memory: a lane marches one SEGMENT -- a run of consecutive samples of one ray whose base
 * voxel lies in one 16^3-voxel brick -- through the brick's tile in LDS and leaves the pair (P, S) = (prod(1-p),
 * sum T p (w - w_first)), the sum relative to the depth weight of the segment's first sample; a per-ray pass chains the pairs in
 * fp64 (sum T p w = S + w_first (1 - P)).  Tables (genre-shapehd_amd/toolbox/_seg_tables.py: build_seg_tables):
 *   seg_rows  int32 [rows,4]    (brick, seg begin, seg end, bx | by << 10 | bz << 20): one workgroup each; every brick in >= 1 row (an empty row stages
 *                               its tile for the live words only)
 *   segs      int32 [nseg,4]    (ray, k0 | L << 8, scratch line, bx | by << 10 | bz << 20 of its brick); inside a row sorted by L
 *                               descending -- the first of every 64 consecutive segments is the longest
 *   ray_nseg  int32 [R*R]       segments per ray; segment s (sample order) of ray q owns scratch line s*R*R + q
 *   ray_pre   float64 [R*R,2] viewed as fp32 [R*R,4]: (transmittance, partial sum) of the samples before the volume
 *   line_w    fp32 [smax*R*R,2] per scratch line: depth_weight of its segment's first and of its last sample
 *   ps_scratch fp32 [N*NC * smax*R*R * 2], 8-byte aligned, smax = max(ray_nseg): receives the pairs
 * Occupancy hint (both or neither): occ int32 [N*NC, ceil(X/cx), ceil(Y/cy), ceil(Z/cz)] with occ_cell = cx*10000 + cy*100 + cz (powers of two) --
 *   word 0 <=> every voxel of that cx x cy x cz cell of that image holds the producer's fill value c (what
 *   genre_back_projection_forward_const writes for dense volumes: genre_cam_cell()) -- and ps_empty fp32 [nseg,2], table order:
 *   the pairs of every segment on the CONSTANT volume vox == c (this op's own ps_scratch on such a volume).  A tile none of whose
 *   cells is occupied is not read: its segments get the constants.  The caller guarantees that vox still is what the producer
 *   wrote and, when it passes `live`, that c * pre_scale does not pass the clamp.
 * v_scratch (optional; with pre_scale != 0 also pass `live`): fp32 [>= N*NC * ceil(nseg/64)*64 * 16], 16-byte aligned, 16 floats
 *   per image and segment (table order; segments in groups of 64, inside a group quarter q of segment l at float4 index
 *   q*64 + l) -- a backward will follow: receives the raw (un-clamped) values of the samples of every segment of
 *   every tile some voxel of which passes the pre_scale clamp (every tile when pre_scale == 0): what
 *   genre_render_seg_backward reads.  ps_scratch is the other half of the saved state.  (`live` then
 *   carries two bits per brick: bit 0 a voxel of the brick passes, bit 1 a voxel of its tile does = its values were saved.) */
int genre_render_seg_forward(const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                             const genre_tensor *out, const genre_tensor *seg_rows, const genre_tensor *segs,
                             const genre_tensor *ray_nseg, const genre_tensor *ray_pre, const genre_tensor *line_w,
                             const genre_tensor *ps_scratch, const genre_tensor *live, const genre_tensor *occ,
                             const genre_tensor *ps_empty,
                             const genre_tensor *v_scratch, float pre_scale, int occ_cell, void *stream);

/* The backward of genre_render_seg_forward (render_spherical.backward of spherical_proj.py:62-72, with the caller's clamp and
 * sph_pad folded in as in the forward): grad_vox (contiguous, every voxel written) from grad_out and the state the forward left --
 * ps_scratch (the pairs), v_scratch (the saved sample values), live (with pre_scale: which tiles' values exist) -- with the
 * forward's tables plus
 *   bwd_rows  int32 [rows,4]    (brick, seg begin, seg end, bx | by << 10 | bz << 20 | split << 30 | first << 31): one workgroup each,
 *                               heaviest first; every brick in >= 1 row; split = the brick's segments are divided over several
 *                               rows, first = the first of them (toolbox/_seg_tables.py)
 *   halo_scratch fp32 [>= N*NC * rows * 832]: per image and row the sums of the 817 tile cells beyond the brick's high faces
 *   tr_scratch fp32 [>= numel(ps_scratch) + N*NC * ceil(R*R / 64)], 8-byte aligned: receives (g T in front, w_last - R behind) of
 *                               every segment and, behind them, per image the per-block maxima of |g T| (the fixed-point scale).
 * Per-ray chains in fp64, then one pass over the segments: dL/dp of every sample and its trilinear adjoint into the brick's tile
 * (64-bit fixed point in LDS, scaled per image), flushed through the clamp's adjoint: a brick's row writes the brick's voxels, the
 * cells beyond the high faces go through halo_scratch and are added onto the neighbours' low faces behind those stores with fp32
 * atomics, as are all rows of a split brick (the order of at most eight addends is not fixed: the last bit of a voxel on a
 * brick's low face can differ between runs). */
int genre_render_seg_backward(const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                              const genre_tensor *grad_out, const genre_tensor *grad_vox, const genre_tensor *bwd_rows,
                              const genre_tensor *segs, const genre_tensor *ray_nseg, const genre_tensor *ray_pre,
                              const genre_tensor *line_w, const genre_tensor *ps_scratch, const genre_tensor *tr_scratch,
                              const genre_tensor *v_scratch, const genre_tensor *halo_scratch, const genre_tensor *live,
                              float pre_scale, void *stream);

/* ---- batch-minor tile renderer (extension; csrc/sph_render_bm.hip) ------------------------------
 *
 * The same operator for BATCHES whose volume is laid out with the image index fastest in memory:
 * vox [N,1,X,Y,Z] with stride[0] == 1 (element (n,x,y,z) at x*sx + y*sy + z*sz + n; e.g. memory order
 * (X,Y,Z,N)).  32 lanes = 32 images of one sample; all geometry comes from tables built once per geometry
 * (genre-shapehd_amd/toolbox/_bm_tables.py: build_bm_tables -- formats are documented there and repeated here):
 *   segs      int32 [nseg,4]  (the segment's line in ps_scratch / tr_scratch = its position in RAY order, first sample k0,
 *                             length L, slot of the first sample); a segment = a run of
 *                             consecutive samples of one ray whose base voxel lies in one 4x8x8-voxel brick (L <= 16);
 *                             sorted by (brick, ray, k0); slots number the in-volume samples in that order
 *   rec_f     int32 [S,12]    per slot: byte offset of the base voxel line in the brick's 5x9x9 x 32-image fp32 tile,
 *                             depth_weight[k] (fp32 bits), 0, 0, the 8 trilinear weights (x fastest, then y, then z);
 *                             S = in-volume samples + 24 unused trailing slots: the kernels issue a fixed number of loads
 *                             per segment (all 64 lanes load 16 bytes of records, all 16 sample slots are fetched) so
 *                             that the hardware's in-order load counter can be waited on exactly; what lies past a
 *                             segment's own records / samples is read and never used
 *   fwd_rows  int32 [rows,4]  (brick, seg begin, seg end, 0): one workgroup each; every brick in >= 1 row
 *   ray_ptr   int32 [R*R+1], ray_seg int32 [nseg]: the segments of each ray in sample order (ray q owns the scratch
 *                             lines ray_ptr[q] .. ray_ptr[q+1]-1: the per-ray passes stream contiguous memory; ray_seg -- the
 *                             segs row of each line -- is part of the table contract, the kernels do not read it)
 *   ray_pre   float64 [R*R,2] viewed as fp32 [R*R,4]: (transmittance, partial sum) of the samples before the ray
 *                             enters the volume (p = 1e-5 each)
 *   ent       int32 [E,4]     (the segment's scratch line, slot of its first sample, i0 | i1<<6 | L<<12 | k0<<18, rec_b slot of sample i0):
 *                             samples i0..i1-1 of the segment have a
 *                             corner inside the brick of the row that lists the entry
 *   rec_b     int32 [SB,12]   byte offset in the brick's own 4x8x8 x 32-image fp64 tile, ownership bits
 *                             (bit c + 4h: corner (x,y) = c of z half h belongs to this brick), 0, 0, 8 weights;
 *                             SB = listed samples + 22 unused trailing records (same reason)
 *   bwd_rows  int32 [rows,4]  (pull brick, ent begin, ent end, shared): shared = 1 rows add onto pre-zeroed voxels;
 *                             the backward's bricks are pull_brick = 488 (4x8x8 voxels) or 888 (8x8x8), as the tables were built
 * Scratch (caller-allocated, groups = ceil(N/32)):
 *   ps_scratch fp32 [groups*nseg*64]: per segment (in ray order) and image (prod(1-p), sum T p w) -- forward output,
 *                             backward input
 *   p_stash    fp32 [groups*S*32] or NULL: clamped sample values (negated where the clamp blocks the gradient);
 *              pass it when a backward will follow;  mask int32 [groups*X*Y*Z + groups] (with p_stash and pre_scale != 0):
 *              bit i = image i passes clamp(vox*pre_scale); the trailing `groups` words (ABI 3) say whether ANY voxel of
 *              the group passes it -- cleared and set by the forward, read by the backward: a group (or a brick) whose
 *              masks are all zero has an identically zero gradient and the backward only writes its zeros
 *   tr_scratch fp32 [groups*nseg*64]: backward only
 * Occupancy hint of the forward (both tensors or both NULL): tile_live int32 [groups, ceil(X/4), ceil(Y/8), ceil(Z/8)] as
 *   genre_back_projection_forward_const writes it -- word 0: every voxel of that brick's tile (brick + high halo) holds the
 *   producer's fill value c in every image of the group -- and ps_empty fp32 [nseg, 4], one row per segment in TABLE order (the
 *   order of segs): (prod(1-p), sum T p w) of the segment on the CONSTANT volume vox == c -- the forward's own ps_scratch on such a
 *   volume, one image's column --, the segment's scratch line (= segs[.,0]) as int32 bits, 0.  A dead tile is then not read: its
 *   segments get the constants.  The caller guarantees that vox still is what the producer wrote.
 * out / grad_out [N,1,R+2p,R+2p] (p = padding margin as above, any strides); pre_scale as above.
 * grad_vox must be batch-minor too (stride[0] == 1); every element is written exactly once. */
/* the forward's brick the library was built for, as X*100 + Y*10 + Z (488 = 4x8x8 voxels): the tables must be built for it */
int genre_bm_brick(void);

int genre_render_bm_forward(const genre_tensor *vox, const genre_tensor *out, const genre_tensor *segs,
                            const genre_tensor *rec_f, const genre_tensor *fwd_rows, const genre_tensor *ray_ptr,
                            const genre_tensor *ray_seg, const genre_tensor *ray_pre,
                            const genre_tensor *ps_scratch, const genre_tensor *p_stash, const genre_tensor *mask,
                            const genre_tensor *tile_live, const genre_tensor *ps_empty, float pre_scale, void *stream);

int genre_render_bm_backward(const genre_tensor *grad_out, const genre_tensor *grad_vox, const genre_tensor *segs,
                             const genre_tensor *ray_ptr, const genre_tensor *ray_seg, const genre_tensor *ray_pre,
                             const genre_tensor *ent, const genre_tensor *rec_b, const genre_tensor *bwd_rows,
                             const genre_tensor *depth_weight, const genre_tensor *ps_scratch,
                             const genre_tensor *tr_scratch, const genre_tensor *p_stash, const genre_tensor *mask,
                             float pre_scale, int pull_brick, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GENRE_HIP_H */
