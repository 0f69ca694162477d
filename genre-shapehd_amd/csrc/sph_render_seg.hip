// sph_render_seg.hip -- the forward of the fused renderer for the standard (NCXYZ) layout, per-SEGMENT form.
//
// Same operator as sph_render.hip: render_spherical.forward of the reference (toolbox/spherical_proj.py:62-72: grid_sample
// [PyTorch 0.4.1 == align_corners=True], clamp, CalcStopProb [calc_prob_kernel.cu:113-143], matmul(depth_weight), prod(1-p), add).
// sph_render.hip's forward wrote the raw value of every sample to a [ray][k] scratch (16 MiB per image) and a second kernel read
// it back to run the per-ray scan: 6.4x the operator's algorithmic traffic.  The ray integral is associative -- a run of
// consecutive samples contributes (P, S) = (prod(1-p_k), sum_k T_k p_k w_k with T = 1 at its start) and the runs chain as
// S += T S_run, T *= P_run -- so here nothing per SAMPLE goes through HBM:
//
//  * seg_sample_kernel: a workgroup stages a 16^3-voxel brick plus one voxel beyond its high faces (the tile a trilinear tap of a
//    sample based in the brick can reach) for G images in LDS, with the caller's clamp(vox * pre_scale) folded in.  ONE LANE then
//    marches ONE SEGMENT -- a run of <= 16 consecutive samples of one ray whose base voxel lies in this brick
//    (toolbox/_seg_tables.py) -- serially: the sample's position and cell from the reference's own fp64 -> fp32 sequence
//    (render_common.hpp: locate), 8 LDS taps per image, the clamp, T and S in registers; 8 bytes per segment and image leave the
//    kernel.  The 64 segments of a wave are neighbours in the table's (length, ray) order: one loop count, no divergence.
//  * seg_combine_kernel: lane = ray, chains the ray's segments in fp64 from the closed-form prefix of the samples before the
//    volume (p = clamp(0) = 1e-5) and writes the map -- optionally laid out as sph_pad(map, pad) (spherical_proj.py:21-28).
//    Segment s of ray q owns scratch line s * R*R + q: a wave's 64 loads are 512 contiguous bytes.
//  * OCCUPANCY.  The volumes this renderer sees are surfaces (a depth map back-projected into 128^3 voxels occupies ~0.5 % of
//    them).  The producer -- the camera forward's brick kernel, csrc/cam_bp.hip -- knows which of its 8x8x32-voxel cells
//    received a point and says so in one word per cell (`occ`); everything else holds its fill value.  A tile none of whose cells
//    is occupied is not read: on the constant tile every segment's (P, S) is a constant of the geometry (`ps_empty`, this
//    kernel's own output on the constant volume, built once per geometry by the caller -- bit-identical to what the march
//    would compute), which the workgroup copies to its segments' lines.
//  * What the backward needs is NOT saved: genre_render_spherical_backward recomputes the raw sample values from the volume
//    (images in which no voxel passes the pre_scale clamp -- every image of GenRe's own chain -- are skipped there: `live` words).
#include "render_common.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

// occupancy cells of the producer: word [img][ncx][ncy][ncz] != 0 <=> cell (cx x cy x cz voxels) may hold anything but the fill
struct Occ { const int *p; int cx, cy, cz, ncx, ncy, ncz; };

constexpr int kMaxZR = 256;

template <int G, int NT>
__global__ __launch_bounds__(NT) void seg_sample_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                         const float *__restrict__ dw, const int4 *__restrict__ rows,
                                                         const int4 *__restrict__ segs, float2 *__restrict__ ps, int lines,
                                                         int imgs, int *__restrict__ live, Occ occ,
                                                         const float2 *__restrict__ ps_empty)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *gtile = lds_f;                                                // [G][kTile3]
    double *a_tab = reinterpret_cast<double *>(lds_f + G * kTile3);      // [ZR]  1 - alpha_k  (spherical_proj.py:52-56)
    float *w_tab = reinterpret_cast<float *>(a_tab + kMaxZR);            // [ZR]  depth_weight
    const int4 row = rows[blockIdx.x];
    const int img0 = blockIdx.y * G;
    const int ng = (imgs - img0 < G) ? imgs - img0 : G;
    const int brick = row.x;
    if (row.y >= row.z && live == nullptr) return;                       // a brick no sample is based in (the cube's corners)
    const int nby = (D.Y + kBrick - 1) / kBrick, nbz = (D.Z + kBrick - 1) / kBrick;
    const int ox = (brick / (nby * nbz)) * kBrick - 1, oy = ((brick / nbz) % nby) * kBrick - 1,
              oz = (brick % nbz) * kBrick - 1;                            // tile origin (incl. the never-fetched low halo)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // ---- occupancy: which of this workgroup's images have anything but the fill value in the tile -----------------------------
    unsigned alive = (1u << ng) - 1u;
    if (occ.p != nullptr) {
        const int xa = (ox + 1) / occ.cx, xb = min(ox + kBrick + 1, D.X - 1) / occ.cx;
        const int ya = (oy + 1) / occ.cy, yb = min(oy + kBrick + 1, D.Y - 1) / occ.cy;
        const int za = (oz + 1) / occ.cz, zb = min(oz + kBrick + 1, D.Z - 1) / occ.cz;
        const int ny = yb - ya + 1, nz = zb - za + 1, ncell = (xb - xa + 1) * ny * nz;
        alive = 0u;
#pragma unroll
        for (int g = 0; g < G; g++) {
            int any = 0;
            if (g < ng) {
                const int *w = occ.p + (size_t)(img0 + g) * occ.ncx * occ.ncy * occ.ncz;
                for (int t = threadIdx.x; t < ncell; t += NT) {
                    const int z = za + t % nz, y = ya + (t / nz) % ny, x = xa + t / (nz * ny);
                    any |= w[(x * occ.ncy + y) * occ.ncz + z];
                }
            }
            if (__syncthreads_or(any)) alive |= 1u << g;
        }
        // a dead tile: every segment's (P, S) is the geometry's constant (the caller guarantees that the fill value does not
        // pass the pre_scale clamp when it asks for the live words -- they stay 0)
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g >= ng || (alive >> g & 1u)) continue;
            float2 *pg = ps + (size_t)(img0 + g) * lines;
            for (int s = row.y + threadIdx.x; s < row.z; s += NT) pg[segs[s].z] = ps_empty[s];
        }
        if (alive == 0u) return;
    }

    // ---- the first chunk's segments and directions are requested in front of the tile ---------------------------------------------
    int c0 = row.y + wave * 64;
    int4 e = make_int4(0, 0, 0, 0);
    double d2x = 0.0, d2y = 0.0, d2z = 0.0;
    if (c0 < row.z) {
        e = segs[min(c0 + lane, row.z - 1)];
        d2x = dirs[e.x * 3 + 0]; d2y = dirs[e.x * 3 + 1]; d2z = dirs[e.x * 3 + 2];
    }

    // ---- stage the tiles: element t = thread + i*NT walked incrementally, all loads of one image in flight together ---------------
    constexpr int kPer = (kTile3 + NT - 1) / NT;
    constexpr int kSX = NT / (kTile * kTile), kSY = (NT % (kTile * kTile)) / kTile, kSZ = NT % kTile;
    static_assert(kSY + 1 < kTile && kSZ < kTile, "tile walk: one carry per axis");
    const int lz0 = (int)threadIdx.x % kTile, ly0 = ((int)threadIdx.x / kTile) % kTile, lx0 = (int)threadIdx.x / (kTile * kTile);
    const int step = kSX * D.sx + kSY * D.sy + kSZ * D.sz, wrap_z = D.sy - kTile * D.sz, wrap_y = D.sx - kTile * D.sy;
    int pass_g[G];
#pragma unroll
    for (int g = 0; g < G; g++) pass_g[g] = 0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g >= ng || !(alive >> g & 1u)) continue;
        const int img = img0 + g;
        const float *__restrict__ base = vox.p + (img / D.NC) * vox.s0 + (img % D.NC) * vox.s1;
        float vals[kPer];
        unsigned inside = 0, own = 0;                                    // own: a voxel of the brick itself (not its halo)
        int lz = lz0, ly = ly0, x = ox + lx0, y = oy + ly0, z = oz + lz0;
        int off = x * D.sx + y * D.sy + z * D.sz;
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            vals[i] = 0.f;
            // the LOW halo planes are never fetched: a sample is listed under the brick of its base corner, so inside the
            // volume it reads tile indices 1..17 only; index 0 is reached by base corner -1 alone -- grid_sample's zero padding
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
    for (int k = threadIdx.x; k < D.ZR; k += NT) {
        a_tab[k] = 1.0 - ((k == D.ZR - 1) ? 1.0 : (double)k * D.step);   // numpy.linspace(0,1,ZR)[k], render_common.hpp: sample_pos
        w_tab[k] = dw[k];
    }
    // live[img][0] = "some voxel of this image passes the pre_scale clamp", live[img][1 + brick] = "some voxel of this brick
    // does" (cleared by the host entry; every writer stores the same 1): what the backward skips (sph_render.hip)
    if (live != nullptr) {
        const int nbricks = ((D.X + kBrick - 1) / kBrick) * nby * nbz;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int any = __syncthreads_or(pass_g[g]);
            if (threadIdx.x == 0 && g < ng && any) {
                int *lv = live + (int64_t)(img0 + g) * (nbricks + 1);
                lv[0] = 1;
                lv[1 + brick] = 1;
            }
        }
    } else {
        __syncthreads();
    }

    // ---- the march: lane = segment ----------------------------------------------------------------------------------------------------
    for (; c0 < row.z; c0 += NT) {
        const int s = c0 + lane;
        const bool act = s < row.z;
        const int k0 = e.y & 255, L = act ? (e.y >> 8) : 0;
        const int Lmax = __builtin_amdgcn_readfirstlane(e.y >> 8);       // lane 0 holds the chunk's longest segment
        const double dx2 = d2x * 2, dy2 = d2y * 2, dz2 = d2z * 2;
        const int line = e.z;
        // the next chunk of this wave
        const int cn = c0 + NT;
        if (cn < row.z) {
            e = segs[min(cn + lane, row.z - 1)];
            d2x = dirs[e.x * 3 + 0]; d2y = dirs[e.x * 3 + 1]; d2z = dirs[e.x * 3 + 2];
        }
        float T[G], S[G];
#pragma unroll
        for (int g = 0; g < G; g++) { T[g] = 1.f; S[g] = 0.f; }
        for (int i = 0; i < Lmax; i++) {
            if (i < L) {
                const int k = k0 + i;
                const double a = a_tab[k];
                const float gx = (float)(dx2 * a), gy = (float)(dy2 * a), gz = (float)(dz2 * a);
                Cell c;
                locate(D, gx, gy, gz, c);
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = corner_w(c, j);
                const float *tp = gtile + ((c.x0 - ox) * kTile + (c.y0 - oy)) * kTile + (c.z0 - oz);
                const float wk = w_tab[k];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (g >= ng || !(alive >> g & 1u)) continue;
                    float acc = 0.f;                                      // ATen corner order, zeros outside
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        acc += tp[g * kTile3 + ((j & 1) ? kTile * kTile : 0) + ((j & 2) ? kTile : 0) + ((j & 4) ? 1 : 0)] * w[j];
                    const float p = fminf(fmaxf(acc, D.lo), D.hi);        // clamp(., 1e-5, 1 - 1e-5)  (spherical_proj.py:66)
                    S[g] = __builtin_fmaf(T[g] * p, wk, S[g]);            // + s_k w_k  (:68)
                    T[g] *= 1.0f - p;
                }
            }
        }
        if (act) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (g >= ng || !(alive >> g & 1u)) continue;
                ps[(size_t)(img0 + g) * lines + line] = make_float2(T[g], S[g]);
            }
        }
    }
}

// ---- chain the segments of a ray: lane = ray ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void seg_combine_kernel(RenderDims D, const float2 *__restrict__ ps,
                                                          const int *__restrict__ ray_nseg,
                                                          const double2 *__restrict__ ray_pre, int lines, View4 out)
{
    const int rr = D.R * D.R;
    const int q = blockIdx.x * NT + threadIdx.x, img = blockIdx.y;
    if (q >= rr) return;
    const int n = ray_nseg[q];
    const double2 pre = ray_pre[q];
    const float2 *__restrict__ b = ps + (size_t)img * lines + q;
    double T = pre.x, S = pre.y;
    for (int s0 = 0; s0 < n; s0 += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = b[(size_t)min(s0 + u, n - 1) * rr];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (s0 + u < n) {
                S += T * (double)v[u].y;
                T *= (double)v[u].x;
            }
        }
    }
    const float val = (float)(S + T);                                   // + prod(1-p)  (:69-71)
    float *o = out.p + (img / D.NC) * out.s0 + (img % D.NC) * out.s1;
    const int i = q / D.R, j = q % D.R;
    if (D.pad == 0) { o[i * out.s2 + j * out.s3] = val; return; }
    int r_lo, r_n, c0, c1;
    pad_span(D.R, D.pad, i, j, r_lo, r_n, c0, c1);
    for (int r = 0; r < r_n; r++) {
        o[(r_lo + r) * out.s2 + c0 * out.s3] = val;
        if (c1 >= 0) o[(r_lo + r) * out.s2 + c1 * out.s3] = val;
    }
}

template <int G, int NT>
void launch_seg_sample(const RenderDims &D, const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *dw,
                       const genre_tensor *rows, const genre_tensor *segs, const genre_tensor *ps, int lines, int imgs, int *live,
                       const Occ &occ, const genre_tensor *ps_empty, hipStream_t st)
{
    constexpr size_t lds = (size_t)G * kTile3 * sizeof(float) + kMaxZR * (sizeof(double) + sizeof(float));
    static_assert(lds <= 64 * 1024, "dynamic LDS beyond 64 KB needs reserve_lds");
    seg_sample_kernel<G, NT><<<dim3((unsigned)rows->size[0], (imgs + G - 1) / G), NT, lds, st>>>(
        D, view5(vox), (const double *)dirs->data, (const float *)dw->data, (const int4 *)rows->data, (const int4 *)segs->data,
        (float2 *)ps->data, lines, imgs, live, occ, ps_empty ? (const float2 *)ps_empty->data : nullptr);
}

// GENRE_SEG_CFG = "G,NT" (1|2, 256|512): A/B switch of the sampler's geometry, read once per process
inline int seg_cfg()
{
    static const int cfg = [] {
        const char *s = getenv("GENRE_SEG_CFG");
        int g = 0, nt = 0;
        if (s && sscanf(s, "%d,%d", &g, &nt) == 2 && (g == 1 || g == 2) && (nt == 256 || nt == 512)) return g * 1000 + nt;
        return 0;
    }();
    return cfg;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_render_seg_forward(const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                                        const genre_tensor *out, const genre_tensor *seg_rows, const genre_tensor *segs,
                                        const genre_tensor *ray_nseg, const genre_tensor *ray_pre,
                                        const genre_tensor *ps_scratch, const genre_tensor *live, const genre_tensor *occ,
                                        const genre_tensor *ps_empty, float pre_scale, int occ_cell, void *stream)
{
    const char *op = "render_seg_forward";
    RenderDims D{};
    if (!check_render(op, vox, dirs, depth_weight, out, D)) return 0;
    D.pre_scale = pre_scale;
    const int imgs = D.N * D.NC, rr = D.R * D.R;
    if ((int64_t)imgs * rr == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    GENRE_REQUIRE(D.ZR <= kMaxZR && (int64_t)rr < (1 << 24), "%s: needs ZR <= 256 and R*R < 2^24", op);
    GENRE_REQUIRE(imgs <= 65535, "%s: N*NC must be <= 65535", op);
    const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
    GENRE_REQUIRE(is_i32(seg_rows, 2) && seg_rows->size[1] == 4 && is_contiguous(seg_rows) && seg_rows->size[0] >= nb &&
                      seg_rows->size[0] < (1 << 30),
                  "%s: seg_rows must be a contiguous int32 [rows >= %d, 4] tensor (every brick in at least one row)", op, nb);
    GENRE_REQUIRE(is_i32(segs, 2) && segs->size[1] == 4 && is_contiguous(segs) && aligned16(segs->data),
                  "%s: segs must be a contiguous, 16-byte aligned int32 [nseg, 4] tensor", op);
    GENRE_REQUIRE(is_i32(ray_nseg, 1) && is_contiguous(ray_nseg) && ray_nseg->size[0] == rr, "%s: ray_nseg must be int32 [R*R]", op);
    GENRE_REQUIRE(is_f32(ray_pre, 2) && is_contiguous(ray_pre) && ray_pre->size[0] == rr && ray_pre->size[1] == 4 &&
                      aligned16(ray_pre->data),
                  "%s: ray_pre must be the float64 [R*R, 2] prefix table viewed as fp32 [R*R, 4]", op);
    GENRE_REQUIRE(is_f32(ps_scratch, 1) && is_contiguous(ps_scratch) && ((uintptr_t)ps_scratch->data & 7u) == 0 &&
                      ps_scratch->size[0] % ((int64_t)2 * imgs * rr) == 0 && ps_scratch->size[0] > 0 &&
                      ps_scratch->size[0] / (2 * imgs) < ((int64_t)1 << 31),
                  "%s: ps_scratch must be a contiguous, 8-byte aligned fp32 buffer of N*NC * smax*R*R * 2 elements", op);
    const int lines = (int)(ps_scratch->size[0] / (2 * imgs));
    int *live_p = nullptr;
    if (live != nullptr && pre_scale != 0.0f) {
        GENRE_REQUIRE(is_i32(live, 1) && is_contiguous(live) && live->size[0] >= (int64_t)imgs * (nb + 1),
                      "%s: live must be int32 [N*NC*(1 + bricks)] = [%lld]", op, (long long)imgs * (nb + 1));
        live_p = (int *)live->data;
        GENRE_REQUIRE(hipMemsetAsync(live_p, 0, (size_t)imgs * (nb + 1) * 4, st) == hipSuccess,
                      "%s: hipMemsetAsync of the live words failed", op);
    }
    Occ oc{};
    GENRE_REQUIRE((occ == nullptr) == (ps_empty == nullptr), "%s: occ and ps_empty come together", op);
    if (occ != nullptr) {
        oc.cx = occ_cell / 10000; oc.cy = (occ_cell / 100) % 100; oc.cz = occ_cell % 100;
        GENRE_REQUIRE(oc.cx >= 1 && oc.cy >= 1 && oc.cz >= 1, "%s: occ_cell must be cx*10000 + cy*100 + cz (voxels per cell)", op);
        oc.ncx = (D.X + oc.cx - 1) / oc.cx; oc.ncy = (D.Y + oc.cy - 1) / oc.cy; oc.ncz = (D.Z + oc.cz - 1) / oc.cz;
        GENRE_REQUIRE(is_i32(occ, 4) && is_contiguous(occ) && occ->size[0] == imgs && occ->size[1] == oc.ncx &&
                          occ->size[2] == oc.ncy && occ->size[3] == oc.ncz,
                      "%s: occ must be a contiguous int32 [N*NC, %d, %d, %d] tensor (cells of %dx%dx%d voxels)", op, oc.ncx, oc.ncy,
                      oc.ncz, oc.cx, oc.cy, oc.cz);
        GENRE_REQUIRE(is_f32(ps_empty, 2) && is_contiguous(ps_empty) && ps_empty->size[0] == segs->size[0] &&
                          ps_empty->size[1] == 2 && ((uintptr_t)ps_empty->data & 7u) == 0,
                      "%s: ps_empty must be a contiguous fp32 [nseg, 2] tensor (table order)", op);
        oc.p = (const int *)occ->data;
    }
    // images per workgroup / threads: two images share one walk over the geometry; a lone image (or few) gets more, smaller
    // workgroups (the tables are cut finer for small batches, toolbox/_seg_tables.py)
    int cfg = seg_cfg();
    if (cfg == 0) cfg = imgs >= 4 ? 2512 : 1256;
    switch (cfg) {
    case 1256: launch_seg_sample<1, 256>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, st); break;
    case 1512: launch_seg_sample<1, 512>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, st); break;
    case 2256: launch_seg_sample<2, 256>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, st); break;
    default:   launch_seg_sample<2, 512>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, st); break;
    }
    GENRE_LAUNCH_CHECK("render_seg forward (sampler)");
    if (imgs * (int64_t)rr >= 65536 * 4)
        seg_combine_kernel<256><<<dim3((rr + 255) / 256, imgs), 256, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out));
    else
        seg_combine_kernel<64><<<dim3((rr + 63) / 64, imgs), 64, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out));
    GENRE_LAUNCH_CHECK("render_seg forward (combine)");
    return 1;
}
