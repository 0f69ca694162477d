// sph_render_seg.hip -- the forward of the fused renderer for the standard (NCXYZ) layout, per-SEGMENT form.
//
// Same operator as sph_render.hip: render_spherical.forward of the reference (toolbox/spherical_proj.py:62-72: grid_sample
// [PyTorch 0.4.1 == align_corners=True], clamp, CalcStopProb [calc_prob_kernel.cu:113-143], matmul(depth_weight), prod(1-p), add).
// sph_render.hip's forward wrote the raw value of every sample to a [ray][k] scratch (16 MiB per image) and a second kernel read
// it back to run the per-ray scan: 6.4x the operator's algorithmic traffic.  The ray integral is associative -- a run of
// consecutive samples contributes (P, S) = (prod(1-p_k), sum_k T_k p_k w_k with T = 1 at its start) and the runs chain as
// S += T S_run, T *= P_run -- so here nothing per SAMPLE goes through HBM:
//
//  * seg_sample_kernel: a workgroup (256 threads, one image) stages a 16^3-voxel brick plus one voxel beyond its high faces (the
//    tile a trilinear tap of a sample based in the brick can reach) in LDS, with the caller's clamp(vox * pre_scale) folded in.
//    ONE LANE then marches ONE SEGMENT -- a run of <= 16 consecutive samples of one ray whose base voxel lies in this brick
//    (toolbox/_seg_tables.py) -- serially: the sample's position and cell from the reference's own fp64 -> fp32 sequence
//    (render_common.hpp: locate), 8 LDS taps, the clamp, T and S in registers; 8 bytes per segment leave the kernel.  The 64
//    segments of a wave are neighbours in the table's (length, ray) order: one loop count, no divergence.
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
//
// What the kernel is bound by (per-workgroup timelines, tools/seg_timeline.py; profiles/r06_ab_experiments.txt): a workgroup lives
// 9-13 us -- kernel arguments 0.9, occupancy words 1.4, segment entries 1.0, tile + directions 1.6 (each one dependent memory
// round trip), the march 4-5 (16 steps of ~90 instructions; one wave alone issues ~one instruction per 4.5 cycles) -- and a
// launch is that latency times (workgroups / resident workgroups): LATENCY x RESIDENCY, not bandwidth and not issue slots.  Hence
// one image and 256 threads per workgroup (28 KB of LDS: five per CU; two images sharing the geometry arithmetic in 55 KB
// measured slower at every batch size), loads that never sit behind a branch, and two samples in flight per march iteration.
#include "render_common.hpp"

#pragma clang fp contract(off)

namespace genre {
namespace {

// occupancy cells of the producer: word [img][ncx][ncy][ncz] != 0 <=> cell (cx x cy x cz voxels) may hold anything but the fill
struct Occ { const int *p; int cx, cy, cz, ncx, ncy, ncz; };     // (cx, cy, cz: log2 of the cell edges inside the kernel)

// Per-workgroup timeline (variant build -DGENRE_SEG_TIMELINE, tools/seg_timeline.py; never in the shipped library): thread 0 of
// every workgroup stamps s_memrealtime (100 MHz) at the marks below into a buffer the host entry dumps to $GENRE_SEG_TIMELINE.
#ifdef GENRE_SEG_TIMELINE
#define GENRE_TL_PARAM , unsigned long long *tl
#define GENRE_TL_ARG , tl_buf
#define GENRE_TL(i) do { if (tl && threadIdx.x == 0) tl[((size_t)(blockIdx.y * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GENRE_TL_PARAM
#define GENRE_TL_ARG
#define GENRE_TL(i) do {} while (0)
#endif

constexpr int kMaxZR = 256;
constexpr int kNT = 256;                                                 // threads per workgroup
constexpr int kRowS = 20;                                                // floats per z-row of the LDS tile (see the staging)
constexpr int kTileF = kTile * kTile * kRowS;                            // floats of the tile

// VEC: the volume's z rows allow 16-byte loads.  SPEC (small batches): the segment entries and the tile are requested BEFORE the
// occupancy words have answered -- one dependent round trip (1.4 us of a batch-1 forward) less for a live tile; a dead tile's
// loads are wasted, which is what the words are there to avoid when bandwidth matters (large batches: SPEC off).
template <bool VEC, bool SPEC>
__global__ __launch_bounds__(kNT) void seg_sample_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                          const float *__restrict__ dw, const int4 *__restrict__ rows,
                                                          const int4 *__restrict__ segs, float2 *__restrict__ ps, int lines,
                                                          int *__restrict__ live, Occ occ,
                                                          const float2 *__restrict__ ps_empty GENRE_TL_PARAM)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *gtile_base = lds_f;                                           // 4 zero floats in front of ...
    float *gtile = lds_f + 4;                                            // ... [kTileF]: tile index -1 is a zero
    double *a_tab = reinterpret_cast<double *>(lds_f + 4 + kTileF);      // [ZR]  1 - alpha_k  (spherical_proj.py:52-56)
    // (No runtime integer division anywhere in this kernel: gfx950 has no divide instruction -- ~40 instructions each -- and a
    // first version spent 1.5 us of every workgroup's 9 on fifteen of them: the brick's coordinates come packed in the row, the
    // image is (blockIdx.y, blockIdx.z) = (n, c), the occupancy cells are powers of two.)
    const int4 row = rows[blockIdx.x];
    const int img = blockIdx.y * D.NC + blockIdx.z;
    const int brick = row.x;
    GENRE_TL(0);
    if (row.y >= row.z && live == nullptr) return;                       // a brick no sample is based in (the cube's corners)
    const int nbricks = ((D.X + kBrick - 1) >> 4) * ((D.Y + kBrick - 1) >> 4) * ((D.Z + kBrick - 1) >> 4);
    static_assert(kBrick == 16, "brick coordinates are packed and shifted for 16^3 bricks");
    const int ox = (row.w & 1023) * kBrick - 1, oy = ((row.w >> 10) & 1023) * kBrick - 1,
              oz = (row.w >> 20) * kBrick - 1;                            // tile origin (incl. the never-fetched low halo)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // ---- occupancy words of the cells this tile overlaps: one per lane (<= 18 for 8x8x32 cells), requested first ---------------
    // Every WAVE decides for itself with a ballot -- one round trip, no LDS, no barrier in front of the loads that matter.
    // (Measured at batch 1, row -> decision: per-thread loads reduced with __syncthreads_or 2.3 us; wave-uniform addresses, i.e.
    // scalar loads the compiler waits for one by one: 3.8 us.)
    int occ_word = 1;
    if (occ.p != nullptr) {
        // (occ.cx / cy / cz hold log2 of the cell edge here)
        const int xa = (ox + 1) >> occ.cx, xb = min(ox + kBrick + 1, D.X - 1) >> occ.cx;
        const int ya = (oy + 1) >> occ.cy, yb = min(oy + kBrick + 1, D.Y - 1) >> occ.cy;
        const int za = (oz + 1) >> occ.cz, zb = min(oz + kBrick + 1, D.Z - 1) >> occ.cz;
        const int ny = yb - ya + 1, nz = zb - za + 1, ncell = (xb - xa + 1) * ny * nz;
        const int *__restrict__ w = occ.p + (size_t)img * occ.ncx * occ.ncy * occ.ncz;
        // ONE unconditional load per lane (the host entry refuses cells so small that a tile overlaps more than 64 of them; lanes
        // beyond the last cell re-read it): a loop around the load would wait for it on the spot.  t -> (x, y, z) with the
        // float-reciprocal quotient (exact for t < 64, divisors <= 17)
        const int t = min(lane, ncell - 1);
        const int tz = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)nz));
        const int ty = (int)(((float)tz + 0.5f) * __builtin_amdgcn_rcpf((float)ny));
        const int z = za + (t - tz * nz), y = ya + (tz - ty * ny), x = xa + ty;
        occ_word = w[(x * occ.ncy + y) * occ.ncz + z];
    }
    // a dead tile: every segment's (P, S) is the geometry's constant (the caller guarantees that the fill value does not pass the
    // pre_scale clamp when it asks for the live words -- the brick's word is 0)
    auto dead_tile = [&]() {
        if (__ballot(occ_word != 0) != 0ull) return false;
        float2 *pg = ps + (size_t)img * lines;
        for (int s = row.y + threadIdx.x; s < row.z; s += kNT) pg[segs[s].z] = ps_empty[s];
        if (live != nullptr && threadIdx.x == 0) live[(int64_t)img * (nbricks + 1) + 1 + brick] = 0;
        return true;
    };
    if (!SPEC && dead_tile()) return;

    GENRE_TL(1);
    // ---- loads, in the order the in-order load counter wants them (gfx950 counts loads and stores in ONE in-order counter, and the
    // compiler places the waits: a load behind a branch or an execution mask "may be pending" on the other path and is waited
    // for at once -- a first version of this staging, with its loads under `if (inside the volume)`, ran them ONE AT A TIME,
    // s_waitcnt vmcnt(1) after each; batch 1: 4.5 us of tile loads for one round trip's worth of work).  Every load here is
    // UNCONDITIONAL, from an address clamped into the volume, and masked when it is used:
    //   the first chunk's segments -> the tile (all of a thread's loads) -> the segments' directions (they need the ray index:
    //   one wait that leaves the tile loads in flight) -> LDS stores -> barrier -> march.
    int c0 = row.y + wave * 64;
    const int s_hi = row.z > row.y ? row.z - 1 : row.y;                   // (an empty row re-reads a neighbour's entry, unused)
    int4 e = segs[min(c0 + lane, s_hi)];

    // LDS layout: a z-row of the tile (18 voxels: low halo, 16 of the brick, high halo) occupies kRowS = 20 floats: the brick's 16
    // at slots 0..15 (four aligned float4), the high halo at slot 16, zeros at 17..19 -- and the LOW halo is slot 19 of the row
    // in front (tile index lz - 1 = -1), a zero: the low planes are never fetched, a sample is listed under the brick of its
    // base corner, so inside the volume it reads tile indices 1..17 only; index 0 is reached by base corner -1 alone, where
    // grid_sample's zero padding applies.  Work item = (row, piece): pieces 0..3 the four float4 of the brick (one 16-byte load
    // when the volume's z rows allow it: VEC), piece 4 = (high halo, 0, 0, 0); 18 x 18 x 5 items, each ONE 16-byte LDS store.
    // (Round 5's element-wise walk spent ~50 instructions per 4-byte load.)
    constexpr int kItems = kTile * kTile * 5, kPer = (kItems + kNT - 1) / kNT;
    float4 vals[kPer];
    unsigned okm = 0u;                                                   // bit 4 i + c: element c of item i is a voxel of the volume
    const float *__restrict__ base = vox.p + blockIdx.y * vox.s0 + blockIdx.z * vox.s1;
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const int it = (int)threadIdx.x + i * kNT;
        const int row_i = it / 5, piece = it - row_i * 5;
        const int lx = row_i / kTile, ly = row_i - lx * kTile;
        const int x = ox + lx, y = oy + ly, z = oz + 1 + piece * 4;
        const bool rok = it < kItems && lx > 0 && ly > 0 && x < D.X && y < D.Y;
        const int nel = piece == 4 ? 1 : 4;
        const int off = rok ? x * D.sx + y * D.sy : 0;
        if (VEC) {                                                        // Z % 4 == 0 and z % 4 == 0: all four elements or none
            const bool ok = rok && z < D.Z;
            vals[i] = *reinterpret_cast<const float4 *>(base + (ok ? off + z : 0));
            okm |= (ok ? (piece == 4 ? 1u : 15u) : 0u) << (4 * i);
        } else {
            float ev[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool ok = rok && c < nel && z + c < D.Z;
                ev[c] = base[ok ? off + (z + c) * D.sz : 0];
                okm |= (ok ? 1u : 0u) << (4 * i + c);
            }
            vals[i] = make_float4(ev[0], ev[1], ev[2], ev[3]);
        }
    }
    if (SPEC && dead_tile()) return;                                     // (the occupancy words were requested first: they answer first)
    // (VEC is a template parameter, not a branch: behind the merge of two paths the wait for the segment entry would be vmcnt(0))
    double d2x = dirs[e.x * 3 + 0], d2y = dirs[e.x * 3 + 1], d2z = dirs[e.x * 3 + 2];

    GENRE_TL(2);
    if (threadIdx.x < 4) gtile_base[threadIdx.x] = 0.f;                  // (index -1 of the first row)
    int passes = 0;                                                      // some voxel of the BRICK passes the pre_scale clamp
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const int it = (int)threadIdx.x + i * kNT;
        if (it >= kItems) break;
        const int row_i = it / 5, piece = it - row_i * 5;
        const int lx = row_i / kTile, ly = row_i - lx * kTile;
        float ev[4] = {vals[i].x, vals[i].y, vals[i].z, vals[i].w};
        const bool own = lx <= kBrick && ly <= kBrick && piece < 4;       // a voxel of the brick itself, not of its halo
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const bool ok = okm >> (4 * i + c) & 1u;
            float val = ok ? ev[c] : 0.f;
            if (D.pre_scale != 0.0f && ok) {                              // depth_pred_with_sph_inpaint.py:124
                const float raw = val * D.pre_scale;
                val = fminf(fmaxf(raw, D.lo), D.hi);
                passes |= (val == raw && own) ? 1 : 0;                   // lo <= raw <= hi: the clamp passes the gradient
            }
            ev[c] = val;
        }
        *reinterpret_cast<float4 *>(gtile + row_i * kRowS + piece * 4) = make_float4(ev[0], ev[1], ev[2], ev[3]);
    }
    GENRE_TL(3);
    for (int k = threadIdx.x; k < D.ZR; k += kNT)
        a_tab[k] = 1.0 - ((k == D.ZR - 1) ? 1.0 : (double)k * D.step);   // numpy.linspace(0,1,ZR)[k], render_common.hpp: sample_pos
    // live[img][1 + brick] = "some voxel of this brick passes the pre_scale clamp": what the backward skips (sph_render.hip).
    // Written by EVERY workgroup of the brick (every row stages the same tile: the same value), so nothing has to be cleared in
    // front of this launch -- a memset node costs a batch-1 forward 3.4 us; live[img][0], "some voxel of this image passes",
    // is the OR of the image's brick words, formed by the per-ray pass behind this kernel.
    if (live != nullptr) {
        const int any = __syncthreads_or(passes);
        if (threadIdx.x == 0) live[(int64_t)img * (nbricks + 1) + 1 + brick] = any ? 1 : 0;
    } else {
        __syncthreads();
    }

    GENRE_TL(4);
    // ---- the march: lane = segment ----------------------------------------------------------------------------------------------------
    for (; c0 < row.z; c0 += kNT) {
        const bool act = c0 + lane < row.z;
        const int k0 = e.y & 255, L = act ? (e.y >> 8) : 0;
        const int Lmax = __builtin_amdgcn_readfirstlane(e.y >> 8);       // lane 0 holds the chunk's longest segment
        const double dx2 = d2x * 2, dy2 = d2y * 2, dz2 = d2z * 2;
        const int line = e.z;
        // the next chunk of this wave (unconditional: the last chunk re-reads the row's last entry)
        e = segs[min(c0 + kNT + lane, s_hi)];
        d2x = dirs[e.x * 3 + 0]; d2y = dirs[e.x * 3 + 1]; d2z = dirs[e.x * 3 + 2];
        // Two samples per iteration, no branch: the geometry and the taps of sample i + 1 do not depend on sample i (only the
        // T / S recurrence does), and a wave executes in order.  A lane beyond its segment's end re-evaluates its last sample
        // (valid addresses) and multiplies the result away (p := 0: T *= 1, S += 0).
        const int klast = k0 + (L > 0 ? L - 1 : 0);
        auto sample = [&](const int k, float &wk) {
            const double a = a_tab[k];
            const float gx = (float)(dx2 * a), gy = (float)(dy2 * a), gz = (float)(dz2 * a);
            Cell c;
            locate(D, gx, gy, gz, c);
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = corner_w(c, j);
            const float *tp = gtile + ((c.x0 - ox) * kTile + (c.y0 - oy)) * kRowS + (c.z0 - oz) - 1;
            wk = dw[k];                                                   // (1 KB table: L1)
            float acc = 0.f;                                              // ATen corner order, zeros outside
#pragma unroll
            for (int j = 0; j < 8; j++)
                acc += tp[((j & 1) ? kTile * kRowS : 0) + ((j & 2) ? kRowS : 0) + ((j & 4) ? 1 : 0)] * w[j];
            return fminf(fmaxf(acc, D.lo), D.hi);                        // clamp(., 1e-5, 1 - 1e-5)  (spherical_proj.py:66)
        };
        float T = 1.f, S = 0.f;
        for (int i = 0; i < Lmax; i += 2) {
            float wa, wb;
            const float pa = sample(min(k0 + i, klast), wa);
            const float pb = sample(min(k0 + i + 1, klast), wb);
            const float qa = i < L ? pa : 0.f, qb = i + 1 < L ? pb : 0.f;
            S = __builtin_fmaf(T * qa, wa, S);                            // + s_k w_k  (:68)
            T *= 1.0f - qa;
            S = __builtin_fmaf(T * qb, wb, S);
            T *= 1.0f - qb;
        }
        GENRE_TL(5);
        if (act) ps[(size_t)img * lines + line] = make_float2(T, S);
    }
    GENRE_TL(6);
}

// ---- chain the segments of a ray: lane = ray ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void seg_combine_kernel(RenderDims D, const float2 *__restrict__ ps,
                                                          const int *__restrict__ ray_nseg,
                                                          const double2 *__restrict__ ray_pre, int lines, View4 out,
                                                          int *__restrict__ live, int nbricks)
{
    const int rr = D.R * D.R;
    const int q = blockIdx.x * NT + threadIdx.x, img = blockIdx.y * D.NC + blockIdx.z;
    if (live != nullptr && blockIdx.x == 0) {                            // live[img][0] = OR of the image's brick words (sampler)
        int *lv = live + (int64_t)img * (nbricks + 1);
        int any = 0;
        for (int b = threadIdx.x; b < nbricks; b += NT) any |= lv[1 + b];
        any = __syncthreads_or(any);
        if (threadIdx.x == 0) lv[0] = any ? 1 : 0;
    }
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
    float *o = out.p + blockIdx.y * out.s0 + blockIdx.z * out.s1;
    int i = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)D.R));   // q / R by reciprocal + one correction (q < 2^24)
    int j = q - i * D.R;
    if (j < 0) { i--; j += D.R; } else if (j >= D.R) { i++; j -= D.R; }
    if (D.pad == 0) { o[i * out.s2 + j * out.s3] = val; return; }
    int r_lo, r_n, c0, c1;
    pad_span(D.R, D.pad, i, j, r_lo, r_n, c0, c1);
    for (int r = 0; r < r_n; r++) {
        o[(r_lo + r) * out.s2 + c0 * out.s3] = val;
        if (c1 >= 0) o[(r_lo + r) * out.s2 + c1 * out.s3] = val;
    }
}

template <bool VEC, bool SPEC>
void launch_seg_sample(const RenderDims &D, const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *dw,
                       const genre_tensor *rows, const genre_tensor *segs, const genre_tensor *ps, int lines, int imgs, int *live,
                       const Occ &occ, const genre_tensor *ps_empty, hipStream_t st)
{
    constexpr size_t lds = (size_t)(4 + kTileF) * sizeof(float) + kMaxZR * sizeof(double);
    static_assert(lds <= 64 * 1024, "dynamic LDS beyond 64 KB needs reserve_lds");
    const dim3 grid((unsigned)rows->size[0], D.N, D.NC);
#ifdef GENRE_SEG_TIMELINE
    static unsigned long long *tl_buf = nullptr;
    const size_t tl_n = (size_t)grid.x * grid.y * grid.z * 8;
    if (getenv("GENRE_SEG_TIMELINE")) {
        if (tl_buf) { (void)hipFree(tl_buf); tl_buf = nullptr; }
        (void)hipMalloc(&tl_buf, tl_n * 8);
        (void)hipMemsetAsync(tl_buf, 0, tl_n * 8, st);
    }
#endif
    seg_sample_kernel<VEC, SPEC><<<grid, kNT, lds, st>>>(D, view5(vox), (const double *)dirs->data, (const float *)dw->data,
                                                         (const int4 *)rows->data, (const int4 *)segs->data, (float2 *)ps->data,
                                                         lines, live, occ,
                                                         ps_empty ? (const float2 *)ps_empty->data : nullptr GENRE_TL_ARG);
#ifdef GENRE_SEG_TIMELINE
    if (tl_buf && getenv("GENRE_SEG_TIMELINE")) {            // dump: [grid.y][grid.x][8] stamps of the launch just made
        (void)hipStreamSynchronize(st);
        unsigned long long *h = (unsigned long long *)malloc(tl_n * 8);
        (void)hipMemcpy(h, tl_buf, tl_n * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("GENRE_SEG_TIMELINE"), "wb");
        if (f) { int hdr[4] = {(int)grid.x, (int)(grid.y * grid.z), 1, kNT}; fwrite(hdr, 4, 4, f); fwrite(h, 8, tl_n, f); fclose(f); }
        free(h);
    }
#endif
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
    GENRE_REQUIRE(D.N <= 65535 && D.NC <= 65535, "%s: N and NC must be <= 65535", op);
    const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
    GENRE_REQUIRE(is_i32(seg_rows, 2) && seg_rows->size[1] == 4 && is_contiguous(seg_rows) && seg_rows->size[0] >= nb &&
                      seg_rows->size[0] < (1 << 30),
                  "%s: seg_rows must be a contiguous int32 [rows >= %d, 4] tensor (every brick in at least one row)", op, nb);
    GENRE_REQUIRE(is_i32(segs, 2) && segs->size[1] == 4 && segs->size[0] >= 1 && is_contiguous(segs) && aligned16(segs->data),
                  "%s: segs must be a contiguous, 16-byte aligned int32 [nseg >= 1, 4] tensor", op);
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
        live_p = (int *)live->data;                                      // (every word is written: nothing to clear)
    }
    Occ oc{};
    GENRE_REQUIRE((occ == nullptr) == (ps_empty == nullptr), "%s: occ and ps_empty come together", op);
    if (occ != nullptr) {
        oc.cx = occ_cell / 10000; oc.cy = (occ_cell / 100) % 100; oc.cz = occ_cell % 100;
        auto pow2 = [](int v) { return v >= 1 && (v & (v - 1)) == 0; };
        GENRE_REQUIRE(pow2(oc.cx) && pow2(oc.cy) && pow2(oc.cz),
                      "%s: occ_cell must be cx*10000 + cy*100 + cz with power-of-two cell edges (voxels per cell)", op);
        GENRE_REQUIRE(((kBrick + oc.cx - 1) / oc.cx + 1) * ((kBrick + oc.cy - 1) / oc.cy + 1) * ((kBrick + oc.cz - 1) / oc.cz + 1) <= 64,
                      "%s: occupancy cells of %dx%dx%d voxels are too small (a 17^3 tile must overlap at most 64 of them)", op,
                      oc.cx, oc.cy, oc.cz);
        oc.ncx = (D.X + oc.cx - 1) / oc.cx; oc.ncy = (D.Y + oc.cy - 1) / oc.cy; oc.ncz = (D.Z + oc.cz - 1) / oc.cz;
        GENRE_REQUIRE(is_i32(occ, 4) && is_contiguous(occ) && occ->size[0] == imgs && occ->size[1] == oc.ncx &&
                          occ->size[2] == oc.ncy && occ->size[3] == oc.ncz,
                      "%s: occ must be a contiguous int32 [N*NC, %d, %d, %d] tensor (cells of %dx%dx%d voxels)", op, oc.ncx, oc.ncy,
                      oc.ncz, oc.cx, oc.cy, oc.cz);
        GENRE_REQUIRE(is_f32(ps_empty, 2) && is_contiguous(ps_empty) && ps_empty->size[0] == segs->size[0] &&
                          ps_empty->size[1] == 2 && ((uintptr_t)ps_empty->data & 7u) == 0,
                      "%s: ps_empty must be a contiguous fp32 [nseg, 2] tensor (table order)", op);
        oc.p = (const int *)occ->data;
        auto lg = [](int v) { int l = 0; while ((1 << l) < v) l++; return l; };
        oc.cx = lg(oc.cx); oc.cy = lg(oc.cy); oc.cz = lg(oc.cz);          // the kernel shifts
    }
    // float4 loads of the brick's z rows: unit z stride, every row start 16-byte aligned (bricks start at multiples of 16)
    const bool vec = vox->stride[4] == 1 && aligned16(vox->data) && vox->stride[0] % 4 == 0 && vox->stride[1] % 4 == 0 &&
                     vox->stride[2] % 4 == 0 && vox->stride[3] % 4 == 0 && D.Z % 4 == 0;
    // speculative tile loads (in front of the occupancy answer): only where the launch is latency-, not bandwidth-bound
    const bool spec = oc.p != nullptr && imgs < 4;
#define GENRE_SEG_LAUNCH(V_, S_) launch_seg_sample<V_, S_>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, st)
    if (vec) { if (spec) GENRE_SEG_LAUNCH(true, true); else GENRE_SEG_LAUNCH(true, false); }
    else { if (spec) GENRE_SEG_LAUNCH(false, true); else GENRE_SEG_LAUNCH(false, false); }
#undef GENRE_SEG_LAUNCH
    GENRE_LAUNCH_CHECK("render_seg forward (sampler)");
    if (imgs * (int64_t)rr >= 65536 * 4)
        seg_combine_kernel<256><<<dim3((rr + 255) / 256, D.N, D.NC), 256, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out),
            live_p, nb);
    else
        seg_combine_kernel<64><<<dim3((rr + 63) / 64, D.N, D.NC), 64, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out),
            live_p, nb);
    GENRE_LAUNCH_CHECK("render_seg forward (combine)");
    return 1;
}
