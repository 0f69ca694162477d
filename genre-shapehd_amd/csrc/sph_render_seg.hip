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
constexpr int kSegSlot = 16;                                             // samples per segment at most (toolbox/_seg_tables.py: MAX_SEG)
                                                                         //  = floats per segment slot of the saved sample values

// VEC: the volume's z rows allow 16-byte loads.  SPEC (small batches): the segment entries and the tile are requested BEFORE the
// occupancy words have answered -- one dependent round trip (1.4 us of a batch-1 forward) less for a live tile; a dead tile's
// loads are wasted, which is what the words are there to avoid when bandwidth matters (large batches: SPEC off).
// SAVE_V (a gradient is wanted): the raw value of every sample of a tile through which a gradient CAN come back -- some voxel of the
// tile passes the pre_scale clamp, or there is no pre_scale -- is also written to v[ray][k], for seg_dp_kernel.  On GenRe's own
// chain no tile qualifies and nothing is written.
template <bool VEC, bool SPEC, bool SAVE_V>
__global__ __launch_bounds__(kNT) void seg_sample_kernel(RenderDims D, View5 vox, const double *__restrict__ dirs,
                                                          const float *__restrict__ dw, const int4 *__restrict__ rows,
                                                          const int4 *__restrict__ segs, float2 *__restrict__ ps, int lines,
                                                          int *__restrict__ live, Occ occ,
                                                          const float2 *__restrict__ ps_empty, float *__restrict__ vbuf,
                                                          int nseg_total GENRE_TL_PARAM)
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
                passes |= (val == raw) ? (own ? 3 : 2) : 0;              // lo <= raw <= hi: the clamp passes the gradient (bit 0: a voxel of the brick, bit 1: of the tile)
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
    bool save_v = SAVE_V;                                                // (workgroup-uniform)
    if (live != nullptr) {
        // bit 0: a voxel of the BRICK passes (what render_bwd_brick_kernel skips by); bit 1: a voxel of the TILE passes = this
        // workgroup saves its samples' values (what seg_dp_kernel skips by: exactly the tiles whose values exist).
        // (__syncthreads_or returns a predicate, not the bitwise OR: one call per bit)
        const int own_any = __syncthreads_or(passes & 1) ? 1 : 0;
        const int tile_any = (SAVE_V && __syncthreads_or(passes & 2)) ? 2 : 0;
        if (threadIdx.x == 0) live[(int64_t)img * (nbricks + 1) + 1 + brick] = own_any | tile_any;
        save_v = SAVE_V && tile_any != 0;
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
        auto sample = [&](const int k, float &wk, float &raw) {
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
            raw = acc;
            return fminf(fmaxf(acc, D.lo), D.hi);                        // clamp(., 1e-5, 1 - 1e-5)  (spherical_proj.py:66)
        };
        float T = 1.f, S = 0.f, Tlo = 0.f;
        const float wc = dw[k0];                                          // S is kept relative to the segment's first depth weight
        // With a gradient wanted the segment's product is carried as T + Tlo (the rounding error of every product, recovered by
        // one fma, follows along) and rounded ONCE at the end: dL/dp is proportional to the transmittance in front of the sample,
        // a product over every sample before it -- a hundred fp32 roundings showed as 2e-6 of dL/dp on the worst of 3.4 M samples
        // (one per segment: 5e-7).  Two fma per sample; the map alone does not need them (1e-6 of it either way).
        auto tmul = [&](const float f) {
            const float t = T * f;
            if (SAVE_V) Tlo = __builtin_fmaf(Tlo, f, __builtin_fmaf(T, f, -t));
            T = t;
        };
        float rawv[kSegSlot];                                             // SAVE_V: the segment's raw sample values
        auto pair = [&](const int i) {
            float wa, wb, ra, rb;
            const float pa = sample(min(k0 + i, klast), wa, ra);
            const float pb = sample(min(k0 + i + 1, klast), wb, rb);
            const float qa = i < L ? pa : 0.f, qb = i + 1 < L ? pb : 0.f;
            S = __builtin_fmaf(T * qa, wa - wc, S);                       // + s_k (w_k - w_k0)  (:68; see seg_combine_kernel)
            tmul(1.0f - qa);
            S = __builtin_fmaf(T * qb, wb - wc, S);
            tmul(1.0f - qb);
            if (SAVE_V) { rawv[i] = ra; rawv[i + 1] = rb; }               // (i is a compile-time constant there: unrolled)
        };
        if (SAVE_V) {
#pragma unroll
            for (int i = 0; i < kSegSlot; i += 2) {
                rawv[i] = rawv[i + 1] = 0.f;
                if (i < Lmax) pair(i);
            }
            // the values leave as ONE aligned 64-byte slot per segment (slot = the segment's position in the table: a wave's
            // 64 slots are 4 KB of contiguous memory) -- written [ray][k], 4 bytes at a time from 64 different rays per store
            // instruction, they cost the forward 250 us at batch 32
            if (save_v && act) {
                float4 *slot = reinterpret_cast<float4 *>(vbuf + ((size_t)img * nseg_total + (unsigned)(c0 + lane)) * kSegSlot);
#pragma unroll
                for (int j = 0; j < kSegSlot / 4; j++) slot[j] = make_float4(rawv[4 * j], rawv[4 * j + 1], rawv[4 * j + 2], rawv[4 * j + 3]);
            }
        } else {
            for (int i = 0; i < Lmax; i += 2) pair(i);
        }
        GENRE_TL(5);
        if (act) ps[(size_t)img * lines + line] = make_float2(SAVE_V ? T + Tlo : T, S);
    }
    GENRE_TL(6);
}

// ---- chain the segments of a ray: lane = ray ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void seg_combine_kernel(RenderDims D, const float2 *__restrict__ ps,
                                                          const int *__restrict__ ray_nseg,
                                                          const double2 *__restrict__ ray_pre, int lines, View4 out,
                                                          int *__restrict__ live, int nbricks, const float2 *__restrict__ line_w)
{
    const int rr = D.R * D.R;
    const int q = blockIdx.x * NT + threadIdx.x, img = blockIdx.y * D.NC + blockIdx.z;
    if (live != nullptr && blockIdx.x == 0) {                            // live[img][0] = OR of the image's brick words (sampler)
        int *lv = live + (int64_t)img * (nbricks + 1);
        int any = 0;
        for (int b = threadIdx.x; b < nbricks; b += NT) any |= lv[1 + b] & 1;
        any = __syncthreads_or(any);
        if (threadIdx.x == 0) lv[0] = any ? 1 : 0;
    }
    if (q >= rr) return;
    const int n = ray_nseg[q];
    const double2 pre = ray_pre[q];
    const float2 *__restrict__ b = ps + (size_t)img * lines + q;
    double T = pre.x, S = pre.y;
    // A segment's S is its sum of s_k (w_k - w_first), w_first = the depth weight of its first sample (line_w): with
    // sum s_k = 1 - P the segment's sum s_k w_k = S + w_first (1 - P).  The fp32 sum then rounds relative to the depth range of
    // sixteen samples instead of to a depth, and stays consistent with P whatever P's own rounding: the backward's dL/dp are
    // DIFFERENCES of depths (w_k - R), and a ray that meets the surface where w_k - R is a few percent of a depth showed the
    // fp32 rounding of a depth-sized S as 1e-6 of its dL/dp, coherent along the ray (tests/test_gpu_render_genre.py, pole rays)
    const float2 *__restrict__ lw = line_w + q;
    for (int s0 = 0; s0 < n; s0 += 8) {
        float2 v[8];
        float wf[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v[u] = b[(size_t)min(s0 + u, n - 1) * rr];
            wf[u] = lw[(size_t)min(s0 + u, n - 1) * rr].x;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (s0 + u < n) {
                S += T * ((double)v[u].y + (double)wf[u] * (1.0 - (double)v[u].x));
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

template <bool VEC, bool SPEC, bool SAVE_V>
void launch_seg_sample(const RenderDims &D, const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *dw,
                       const genre_tensor *rows, const genre_tensor *segs, const genre_tensor *ps, int lines, int imgs, int *live,
                       const Occ &occ, const genre_tensor *ps_empty, float *vbuf, hipStream_t st)
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
    seg_sample_kernel<VEC, SPEC, SAVE_V><<<grid, kNT, lds, st>>>(D, view5(vox), (const double *)dirs->data, (const float *)dw->data,
                                                                 (const int4 *)rows->data, (const int4 *)segs->data,
                                                                 (float2 *)ps->data, lines, live, occ,
                                                                 ps_empty ? (const float2 *)ps_empty->data : nullptr, vbuf,
                                                                 (int)segs->size[0] GENRE_TL_ARG);
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


// ---- backward, segment form: dL/dp of every sample from per-segment state ------------------------------------------------------------
//   dL/dp_k = g T_k (w_k - R_{k+1}),   R_k = p_k w_k + (1 - p_k) R_{k+1},   R behind the ray's last sample = 1      (no division, no
// cancellation: the form of csrc/sph_render_bm.hip).  T_k = transmittance in front of sample k, g = dL/d(map value of the ray).
// seg_combine_bwd_kernel (lane = ray) chains the forward's (P, S) pairs once forwards (g T in front of every segment) and once
// backwards (R behind every segment's end: R in front of a segment = S + P R behind it) in fp64; seg_dp_kernel (lane = segment)
// re-runs the segment's two short scans from the saved sample values and writes dL/dp[ray][k] for render_bwd_brick_kernel
// (csrc/sph_render.hip), plus each image's max |dL/dp| for its fixed-point scale.  Replaces round 5's render_scan_bwd_kernel (a
// wave per ray over ALL samples, fp64 DPP scans) and needs the sample values only where a gradient can come back.
constexpr int kMaxRaySegs = 32;

template <int NT>
__global__ __launch_bounds__(NT) void seg_combine_bwd_kernel(RenderDims D, const float2 *__restrict__ ps,
                                                              const int *__restrict__ ray_nseg,
                                                              const double2 *__restrict__ ray_pre, int lines, View4 gout,
                                                              float2 *__restrict__ tr, const int *__restrict__ live, int nbricks,
                                                              const float2 *__restrict__ line_w)
{
    const int rr = D.R * D.R;
    const int q = blockIdx.x * NT + threadIdx.x, img = blockIdx.y * D.NC + blockIdx.z;
    if (live != nullptr && live[(int64_t)img * (nbricks + 1)] == 0) return;      // no voxel of this image passes the clamp: nothing reads tr
    if (q >= rr) return;
    const int n = min(ray_nseg[q], kMaxRaySegs);
    const float2 *__restrict__ b = ps + (size_t)img * lines + q;
    float2 *__restrict__ t = tr + (size_t)img * lines + q;
    float2 v[kMaxRaySegs], lw[kMaxRaySegs];                              // (P, S), (first, last depth weight) of the ray's segments
#pragma unroll
    for (int u = 0; u < kMaxRaySegs; u++) {
        v[u] = b[(size_t)min(u, max(n - 1, 0)) * rr];
        lw[u] = line_w[(size_t)min(u, max(n - 1, 0)) * rr + q];
    }
    // gradient of the ray's value: the sum over its padded positions (sph_pad, spherical_proj.py:21-28)
    const float *gi = gout.p + blockIdx.y * gout.s0 + blockIdx.z * gout.s1;
    int i = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)D.R));
    int j = q - i * D.R;
    if (j < 0) { i--; j += D.R; } else if (j >= D.R) { i++; j -= D.R; }
    double g = 0.0;
    if (D.pad == 0) g = (double)gi[i * gout.s2 + j * gout.s3];
    else {
        int r_lo, r_n, c0, c1;
        pad_span(D.R, D.pad, i, j, r_lo, r_n, c0, c1);
        for (int r = 0; r < r_n; r++) {
            g += (double)gi[(r_lo + r) * gout.s2 + c0 * gout.s3];
            if (c1 >= 0) g += (double)gi[(r_lo + r) * gout.s2 + c1 * gout.s3];
        }
    }
    double T = ray_pre[q].x;
    float gT[kMaxRaySegs];
#pragma unroll
    for (int u = 0; u < kMaxRaySegs; u++) {
        gT[u] = (float)(g * T);
        if (u < n) T *= (double)v[u].x;
    }
    // R in fp64; what leaves is the DIFFERENCE seg_dp_kernel starts from, w_last - R behind the segment (rounded to fp32 relative
    // to itself, not to a depth).  In front of a segment: R = w_first + S + P (R behind - w_first)  (seg_combine_kernel: S is relative)
    double R = 1.0;                                                      // behind the last sample: prod(1-p) * 1  (:69-71)
#pragma unroll
    for (int u = kMaxRaySegs - 1; u >= 0; u--) {
        if (u < n) {
            t[(size_t)u * rr] = make_float2(gT[u], (float)((double)lw[u].y - R));
            R = (double)lw[u].x + (double)v[u].y + (double)v[u].x * (R - (double)lw[u].x);
        }
    }
}

__device__ __forceinline__ void publish_max_bits(unsigned wmax, int lane, unsigned *dpmax_bits)
{
    // a wave's max |dL/dp| as a bit pattern (non-negative floats order like their bits; Inf / NaN sort above every finite value, so a
    // non-finite gradient survives: csrc/sph_render.hip: publish_max)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (unsigned)__shfl_xor((int)wmax, o, 64));
    if (lane == 0 && wmax > 0u &&
        wmax > __hip_atomic_load(dpmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dpmax_bits, wmax);
}

constexpr int kDpSeg = kSegSlot;

__global__ __launch_bounds__(kNT) void seg_dp_kernel(RenderDims D, const int4 *__restrict__ segs, int nseg,
                                                      const float2 *__restrict__ tr, int lines, const float *__restrict__ vbuf,
                                                      const float *__restrict__ dw, float *__restrict__ dpbuf,
                                                      unsigned *__restrict__ dpmax_bits, const int *__restrict__ live, int nbricks)
{
    __shared__ float dw_s[kMaxZR];
    const int img = blockIdx.y * D.NC + blockIdx.z, lane = threadIdx.x & 63;
    const int *lv = live ? live + (int64_t)img * (nbricks + 1) : nullptr;
    if (lv != nullptr && lv[0] == 0) return;                             // no voxel of this image passes the clamp: grad_vox = 0
    const int s = min((int)(blockIdx.x * kNT + threadIdx.x), nseg - 1);
    const bool act = (int)(blockIdx.x * kNT + threadIdx.x) < nseg;
    // Everything whose address is known now is requested now -- the table entry, the segment's saved values (one aligned 64-byte slot,
    // found by the segment's index alone), the depth weights for LDS -- and only the two loads that need the entry (the brick's word,
    // the segment's chain line) wait for it: two dependent round trips per wave instead of four (the kernel is bound by them: 113 000
    // waves at batch 32).  Slots of tiles without saved values hold whatever the buffer held: read, never used (selects below).
    const int4 e = segs[s];
    float p[kDpSeg], w[kDpSeg], c[kDpSeg];
    const float4 *slot = reinterpret_cast<const float4 *>(vbuf + ((size_t)img * nseg + (unsigned)s) * kSegSlot);
#pragma unroll
    for (int j = 0; j < kDpSeg / 4; j++) {
        const float4 t4 = slot[j];
        p[4 * j] = t4.x; p[4 * j + 1] = t4.y; p[4 * j + 2] = t4.z; p[4 * j + 3] = t4.w;
    }
    for (int i = threadIdx.x; i < kMaxZR; i += kNT) dw_s[i] = dw[min(i, D.ZR - 1)];
    const int k0 = e.y & 255, L = min(e.y >> 8, kDpSeg);
    // a gradient can come back through this segment's tile -- its brick and the voxels one step beyond the high faces -- only if
    // one of the tile's voxels passes the pre_scale clamp: bit 1 of the brick's word, set by the very workgroup that then saved the
    // tile's sample values.  Elsewhere no values exist and nobody's gradient depends on dL/dp (whatever render_bwd_brick_kernel
    // accumulates from such samples lands on voxels the clamp blocks: a select)
    bool tile_live = true;
    const float2 st = tr[(size_t)img * lines + e.z];                       // (g T in front of the segment, w_last - R behind its end)
    if (lv != nullptr) {
        const int nby = (D.Y + kBrick - 1) >> 4, nbz = (D.Z + kBrick - 1) >> 4;
        const int bx = e.w & 1023, by = (e.w >> 10) & 1023, bz = e.w >> 20;
        tile_live = (lv[1 + (bx * nby + by) * nbz + bz] & 2) != 0;
    }
    __syncthreads();
    unsigned wmax = 0u;
    if (__ballot(act && tile_live) != 0ull) {
        unsigned pass = 0u;
#pragma unroll
        for (int i = 0; i < kDpSeg; i++) w[i] = dw_s[k0 + min(i, L - 1)];
        // Both short scans in fp64 (the kernel waits for memory, not for its ALUs): forwards g T_k; backwards
        // dL/dp_k = g T_k (w_k - R_{k+1}) with the DIFFERENCE d_k = w_k - R_{k+1} carried instead of R -- from R_k = R_{k+1} + p_k d_k
        // follows d_{k-1} = (w_{k-1} - w_k) + (1 - p_k) d_k.  In fp32 with R carried, sixteen roundings of a number ~2 under
        // differences ~0.3 showed as 2e-5 of the gradient's scale on sharp volumes (tests/test_gpu_render_genre.py asks for 1e-5)
        double Tg = (double)st.x, gt[kDpSeg];
#pragma unroll
        for (int i = 0; i < kDpSeg; i++) {
            const float raw = p[i];
            p[i] = fminf(fmaxf(raw, D.lo), D.hi);                         // clamp(., 1e-5, 1 - 1e-5)  (spherical_proj.py:66)
            pass |= (raw >= D.lo && raw <= D.hi) ? 1u << i : 0u;          // torch.clamp's backward mask
            gt[i] = Tg;
            if (i < L) Tg *= (double)(1.0f - p[i]);
        }
        double d = 0.0;
#pragma unroll
        for (int i = kDpSeg - 1; i >= 0; i--) {
            c[i] = 0.f;
            if (i < L) {
                double dn = (double)st.y;                                  // the segment's last sample: w - R behind it, from the fp64 chain
                if (i + 1 < kDpSeg) {
                    if (i + 1 < L) dn = fma((double)(1.0f - p[min(i + 1, kDpSeg - 1)]), d, (double)(w[i] - w[min(i + 1, kDpSeg - 1)]));
                }
                d = dn;
                c[i] = (pass >> i & 1u) ? (float)(gt[i] * d) : 0.f;
                if (act && tile_live) wmax = max(wmax, __float_as_uint(c[i]) & 0x7fffffffu);
            }
        }
        if (act && tile_live) {       // one aligned 64-byte slot per segment, like the saved values (render_bwd_brick_kernel finds a
                                      // listed sample's slot through chunk_slot): written [ray][k], 4 bytes at a time to 64 different
                                      // rays per store instruction, the stores alone took 400 us at batch 32
            float4 *dslot = reinterpret_cast<float4 *>(dpbuf + ((size_t)img * nseg + (unsigned)s) * kSegSlot);
#pragma unroll
            for (int j = 0; j < kDpSeg / 4; j++)
                dslot[j] = make_float4(4 * j < L ? c[4 * j] : 0.f, 4 * j + 1 < L ? c[4 * j + 1] : 0.f, 4 * j + 2 < L ? c[4 * j + 2] : 0.f,
                                       4 * j + 3 < L ? c[4 * j + 3] : 0.f);
        }
    }
    publish_max_bits(wmax, lane, dpmax_bits + img);
}

}  // namespace

// the dL/dp phase of genre_render_spherical_backward in segment form (called from csrc/sph_render.hip; arguments checked here)
int seg_backward_dlp(const char *op, const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                     const genre_tensor *grad_out, const genre_tensor *segs, const genre_tensor *ray_nseg,
                     const genre_tensor *ray_pre, const genre_tensor *line_w, const genre_tensor *ps_scratch,
                     const genre_tensor *tr_scratch,
                     const genre_tensor *v_scratch, float *dp, unsigned *dpmax, const int *live, float pre_scale, hipStream_t st)
{
    RenderDims D{};
    if (!check_render(op, vox, dirs, depth_weight, grad_out, D)) return 0;
    D.pre_scale = pre_scale;
    const int imgs = D.N * D.NC, rr = D.R * D.R;
    const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
    GENRE_REQUIRE(D.ZR <= kMaxZR && D.N <= 65535 && D.NC <= 65535, "%s: needs ZR <= 256, N and NC <= 65535", op);
    GENRE_REQUIRE(is_i32(segs, 2) && segs->size[1] == 4 && segs->size[0] >= 1 && is_contiguous(segs) && aligned16(segs->data),
                  "%s: segs must be a contiguous, 16-byte aligned int32 [nseg >= 1, 4] tensor", op);
    GENRE_REQUIRE(is_i32(ray_nseg, 1) && is_contiguous(ray_nseg) && ray_nseg->size[0] == rr, "%s: ray_nseg must be int32 [R*R]", op);
    GENRE_REQUIRE(is_f32(ray_pre, 2) && is_contiguous(ray_pre) && ray_pre->size[0] == rr && ray_pre->size[1] == 4 &&
                      aligned16(ray_pre->data), "%s: ray_pre must be the float64 [R*R, 2] prefix table viewed as fp32 [R*R, 4]", op);
    GENRE_REQUIRE(line_w != nullptr && is_f32(line_w, 2) && is_contiguous(line_w) && line_w->size[1] == 2 &&
                      ((uintptr_t)line_w->data & 7u) == 0 && line_w->size[0] * (int64_t)2 * imgs == ps_scratch->size[0],
                  "%s: line_w must be fp32 [smax*R*R, 2], one pair per scratch line of an image", op);
    GENRE_REQUIRE(is_f32(ps_scratch, 1) && is_contiguous(ps_scratch) && ((uintptr_t)ps_scratch->data & 7u) == 0 &&
                      ps_scratch->size[0] % ((int64_t)2 * imgs * rr) == 0 && ps_scratch->size[0] > 0 &&
                      ps_scratch->size[0] / (2 * imgs) < ((int64_t)1 << 31) && ps_scratch->size[0] / ((int64_t)2 * imgs * rr) <= kMaxRaySegs,
                  "%s: ps_scratch must be the forward's fp32 [N*NC * smax*R*R * 2] buffer (smax <= %d)", op, kMaxRaySegs);
    GENRE_REQUIRE(is_f32(tr_scratch, 1) && is_contiguous(tr_scratch) && ((uintptr_t)tr_scratch->data & 7u) == 0 &&
                      tr_scratch->size[0] >= ps_scratch->size[0], "%s: tr_scratch must hold as many floats as ps_scratch", op);
    GENRE_REQUIRE(is_f32(v_scratch, 1) && is_contiguous(v_scratch) && aligned16(v_scratch->data) &&
                      v_scratch->size[0] >= (int64_t)imgs * segs->size[0] * kSegSlot,
                  "%s: v_scratch must be the forward's fp32 [N*NC*nseg*%d] buffer (one slot per segment)", op, kSegSlot);
    const int lines = (int)(ps_scratch->size[0] / (2 * imgs));
    const int nseg = (int)segs->size[0];
    if (imgs * (int64_t)rr >= 65536 * 4)
        seg_combine_bwd_kernel<256><<<dim3((rr + 255) / 256, D.N, D.NC), 256, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines,
            view4(grad_out), (float2 *)tr_scratch->data, live, nb, (const float2 *)line_w->data);
    else
        seg_combine_bwd_kernel<64><<<dim3((rr + 63) / 64, D.N, D.NC), 64, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines,
            view4(grad_out), (float2 *)tr_scratch->data, live, nb, (const float2 *)line_w->data);
    GENRE_LAUNCH_CHECK("render_spherical backward (segment chains)");
    seg_dp_kernel<<<dim3((nseg + kNT - 1) / kNT, D.N, D.NC), kNT, 0, st>>>(
        D, (const int4 *)segs->data, nseg, (const float2 *)tr_scratch->data, lines, (const float *)v_scratch->data,
        (const float *)depth_weight->data, dp, dpmax, live, nb);
    GENRE_LAUNCH_CHECK("render_spherical backward (dL/dp per segment)");
    return 1;
}

}  // namespace genre

using namespace genre;

extern "C" int genre_render_seg_forward(const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                                        const genre_tensor *out, const genre_tensor *seg_rows, const genre_tensor *segs,
                                        const genre_tensor *ray_nseg, const genre_tensor *ray_pre, const genre_tensor *line_w,
                                        const genre_tensor *ps_scratch, const genre_tensor *live, const genre_tensor *occ,
                                        const genre_tensor *ps_empty, const genre_tensor *v_scratch, float pre_scale,
                                        int occ_cell, void *stream)
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
    GENRE_REQUIRE(is_f32(line_w, 2) && is_contiguous(line_w) && line_w->size[1] == 2 && ((uintptr_t)line_w->data & 7u) == 0 &&
                      line_w->size[0] * (int64_t)2 * imgs == ps_scratch->size[0],
                  "%s: line_w must be fp32 [smax*R*R, 2], one pair per scratch line of an image", op);
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
    float *vbuf = nullptr;
    if (v_scratch != nullptr) {          // a gradient is wanted: the raw sample values of the tiles a gradient can come back through
        GENRE_REQUIRE(is_f32(v_scratch, 1) && is_contiguous(v_scratch) && aligned16(v_scratch->data) &&
                          v_scratch->size[0] >= (int64_t)imgs * segs->size[0] * kSegSlot,
                      "%s: v_scratch must be a contiguous, 16-byte aligned fp32 buffer of >= N*NC*nseg*%d elements", op, kSegSlot);
        GENRE_REQUIRE(pre_scale == 0.0f || live_p != nullptr, "%s: v_scratch with pre_scale needs the live words too", op);
        vbuf = (float *)v_scratch->data;
    }
#define GENRE_SEG_LAUNCH(V_, S_)                                                                                                       \
    do {                                                                                                                              \
        if (vbuf) launch_seg_sample<V_, S_, true>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, vbuf, st); \
        else launch_seg_sample<V_, S_, false>(D, vox, dirs, depth_weight, seg_rows, segs, ps_scratch, lines, imgs, live_p, oc, ps_empty, nullptr, st); \
    } while (0)
    if (vec) { if (spec) GENRE_SEG_LAUNCH(true, true); else GENRE_SEG_LAUNCH(true, false); }
    else { if (spec) GENRE_SEG_LAUNCH(false, true); else GENRE_SEG_LAUNCH(false, false); }
#undef GENRE_SEG_LAUNCH
    GENRE_LAUNCH_CHECK("render_seg forward (sampler)");
    if (imgs * (int64_t)rr >= 65536 * 4)
        seg_combine_kernel<256><<<dim3((rr + 255) / 256, D.N, D.NC), 256, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out),
            live_p, nb, (const float2 *)line_w->data);
    else
        seg_combine_kernel<64><<<dim3((rr + 63) / 64, D.N, D.NC), 64, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines, view4(out),
            live_p, nb, (const float2 *)line_w->data);
    GENRE_LAUNCH_CHECK("render_seg forward (combine)");
    return 1;
}
