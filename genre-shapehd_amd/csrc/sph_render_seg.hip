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
//  * With a gradient wanted (v_scratch) the raw sample values are saved too, one 64-byte slot per segment, in the tiles a gradient
//    can come back through (none on GenRe's own chain: `live` words); genre_render_seg_backward (below) needs nothing else.
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
// Saved sample values: 16 floats per image and segment, laid out so that the lanes of a wave -- 64 consecutive segments -- read or
// write 1 KB of contiguous memory per instruction: segments in groups of 64 (4 KB), inside a group quarter q (samples 4q .. 4q+3)
// of segment l at float4 index q * 64 + l.  (One 64-byte slot per segment, a lane's four float4 behind one another, cost the
// forward's stores and the backward's loads four passes over the same lines.)
__device__ __forceinline__ size_t seg_groups(int nseg) { return (size_t)((nseg + 63) >> 6); }
__device__ __forceinline__ const float4 *slot_q0(const float *vbuf, int img, int nseg, int s)
{
    return reinterpret_cast<const float4 *>(vbuf) + ((size_t)img * seg_groups(nseg) + ((unsigned)s >> 6)) * 256 + (s & 63);
}
                                                                         //  = floats per segment slot of the saved sample values

// VEC: the volume's z rows allow 16-byte loads.  SPEC (small batches): the segment entries and the tile are requested BEFORE the
// occupancy words have answered -- one dependent round trip (1.4 us of a batch-1 forward) less for a live tile; a dead tile's
// loads are wasted, which is what the words are there to avoid when bandwidth matters (large batches: SPEC off).
// SAVE_V (a gradient is wanted): the raw value of every sample of a tile through which a gradient CAN come back -- some voxel of the
// tile passes the pre_scale clamp, or there is no pre_scale -- is also saved (16 floats per segment), for seg_scatter_kernel.  On GenRe's own
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
        // workgroup saves its samples' values (what seg_scatter_kernel skips by: exactly the tiles whose values exist).
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
                float4 *slot = const_cast<float4 *>(slot_q0(vbuf, img, nseg_total, c0 + lane));
#pragma unroll
                for (int j = 0; j < kSegSlot / 4; j++) slot[j * 64] = make_float4(rawv[4 * j], rawv[4 * j + 1], rawv[4 * j + 2], rawv[4 * j + 3]);
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


// ---- backward: who writes what of grad_vox --------------------------------------------------------------------------------------
// A row's workgroup (seg_scatter_kernel) accumulates the tile of its brick: the brick's 16^3 voxels, which it WRITES (plain
// stores; every voxel of grad_vox is written by its brick's row, zeros where nothing comes back), and the 817 cells one step
// beyond the high faces, voxels of other bricks, which it leaves in its halo record; seg_halo_kernel adds the records onto the
// voxels behind the first kernel's stores.  Bricks whose segments are divided over several rows (bit 30 of the row's last column;
// bit 31: the first of them) are zeroed first (by seg_combine_bwd_kernel, on its way) and added to with atomics by all their rows.
constexpr int kHF = kBrick + 1;                                          // cells per tile edge that exist as voxels: locals 0 .. 16
constexpr int kHaloX = 0;                                                // x face: local x = 16, [y 0..16][z 0..16]
constexpr int kHaloY = kHF * kHF;                                        // y face: local y = 16, [x 0..15][z 0..16]
constexpr int kHaloZ = kHaloY + kBrick * kHF;                            // z face: local z = 16, [x 0..15][y 0..15]
constexpr int kHaloN = kHaloZ + kBrick * kBrick;                         // 817
constexpr int kHaloRec = 832;                                            // floats per record

struct RowBits { int bx0, by0, bz0; bool split, first; };
__device__ __forceinline__ RowBits row_bits(const int4 &row)
{
    const unsigned w = (unsigned)row.w;
    return {(int)(w & 1023u) * kBrick, (int)((w >> 10) & 1023u) * kBrick, (int)((w >> 20) & 1023u) * kBrick, ((w >> 30) & 1u) != 0,
            (w >> 31) != 0};
}

// the row's tile holds nothing that comes back (or the row is empty): with live words, no voxel of the tile passes the pre_scale clamp
__device__ __forceinline__ bool row_dead(const RenderDims &D, const int4 &row, const int *__restrict__ live, int img)
{
    if (row.y >= row.z) return true;                                     // a brick no sample is based in (the cube's corners)
    if (live == nullptr) return false;
    const int nbricks = ((D.X + kBrick - 1) >> 4) * ((D.Y + kBrick - 1) >> 4) * ((D.Z + kBrick - 1) >> 4);
    const int *lv = live + (int64_t)img * (nbricks + 1);
    return lv[0] == 0 || (lv[1 + row.x] & 2) == 0;
}

__device__ __forceinline__ void zero_brick(const RenderDims &D, const View5 &gvox, float *gb, const RowBits &b, int nthreads)
{
    // 16-byte stores where the z rows allow it.  PLAIN stores: a brick's z row is 64 bytes, half of a 128-byte line whose other
    // half belongs to the next brick -- nontemporal stores (no merging in L2) measured slower (csrc/sph_render.hip)
    const bool v4 = gvox.s4 == 1 && ((gvox.s2 | gvox.s3) & 3) == 0 && (reinterpret_cast<uintptr_t>(gb) & 15) == 0 && b.bz0 + kBrick <= D.Z;
    for (int it = threadIdx.x; it < kBrick * kBrick * 4; it += nthreads) {
        const int x = b.bx0 + (it >> 6), y = b.by0 + ((it >> 2) & 15), z = b.bz0 + (it & 3) * 4;
        if (x >= D.X || y >= D.Y) continue;
        float *dst = gb + x * gvox.s2 + y * gvox.s3 + z * gvox.s4;
        if (v4) *reinterpret_cast<float4 *>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
        else
            for (int c = 0; c < 4; c++)
                if (z + c < D.Z) dst[c * gvox.s4] = 0.f;
    }
}

// ---- backward, segment form: dL/dp of every sample from per-segment state ------------------------------------------------------------
//   dL/dp_k = g T_k (w_k - R_{k+1}),   R_k = p_k w_k + (1 - p_k) R_{k+1},   R behind the ray's last sample = 1      (no division, no
// cancellation: the form of csrc/sph_render_bm.hip).  T_k = transmittance in front of sample k, g = dL/d(map value of the ray).
// seg_combine_bwd_kernel (lane = ray) chains the forward's (P, S) pairs once forwards (g T in front of every segment) and once
// backwards (R behind every segment's end) in fp64; seg_scatter_kernel (lane = segment) re-runs the segment's two short scans from
// the saved sample values and scatters dL/dp times the trilinear weights into its brick's tile.  Replaces round 5's
// render_scan_bwd_kernel + render_bwd_brick_kernel (a wave per ray over ALL samples, dL/dp[ray][k] through memory, a lane per
// (sample, brick it touches)) for forwards by genre_render_seg_forward; needs the sample values only where a gradient can come back.
constexpr int kMaxRaySegs = 32;

// (NB: lines requested together: 8 for large batches -- registers, occupancy --, all 32 for small ones, where a ray's dependent round
// trips are the kernel's duration)
template <int NT, int NB>
__global__ __launch_bounds__(NT) void seg_combine_bwd_kernel(RenderDims D, const float2 *__restrict__ ps,
                                                              const int *__restrict__ ray_nseg,
                                                              const double2 *__restrict__ ray_pre, int lines, View4 gout,
                                                              float2 *__restrict__ tr, const int *__restrict__ live, int nbricks,
                                                              const float2 *__restrict__ line_w, unsigned *__restrict__ bmax,
                                                              const int4 *__restrict__ rows, int nrows, View5 gvox)
{
    __shared__ unsigned red[NT / 64];
    const int rr = D.R * D.R;
    const int img = blockIdx.y * D.NC + blockIdx.z;
    if (live != nullptr && live[(int64_t)img * (nbricks + 1)] == 0) return;      // no voxel of this image passes the clamp: nothing reads tr
    // on the way: the bricks whose segments are divided over several rows are zeroed here, in front of seg_scatter_kernel whose
    // rows all ADD to them (in an image nothing comes back through, the first row of such a brick writes its zeros there: all rows
    // of a brick are dead together).  The rows of split bricks are the head of the table: a block stops at the first other row
    for (int i = blockIdx.x; i < nrows; i += gridDim.x) {
        const RowBits rb = row_bits(rows[i]);
        if (!rb.split) break;
        if (rb.first) zero_brick(D, gvox, gvox.p + blockIdx.y * gvox.s0 + blockIdx.z * gvox.s1, rb, NT);
    }
    const bool ray = (int)(blockIdx.x * NT + threadIdx.x) < rr;
    const int q = min((int)(blockIdx.x * NT + threadIdx.x), rr - 1);              // (lanes beyond the last ray repeat it, store nothing)
    const int n = ray ? min(ray_nseg[q], kMaxRaySegs) : 0;
    const float2 *__restrict__ b = ps + (size_t)img * lines + q;
    float2 *__restrict__ t = tr + (size_t)img * lines + q;
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
        // (eight rows in flight, both columns, loads unconditional: a pole row's 17 x 2 positions one dependent load at a time were
        // most of this kernel's duration at batch 1)
        const int c1c = c1 >= 0 ? c1 : c0;
        for (int r = 0; r < r_n; r += 8) {
            float ga[8], gb2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int rw = r_lo + min(r + u, r_n - 1);
                ga[u] = gi[rw * gout.s2 + c0 * gout.s3];
                gb2[u] = gi[rw * gout.s2 + c1c * gout.s3];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r + u < r_n) {
                    g += (double)ga[u];
                    if (c1 >= 0) g += (double)gb2[u];
                }
            }
        }
    }
    // Two passes over the ray's lines, eight loads in flight each (the (P, S) pairs of all 32 possible segments plus their depth
    // weights held in registers -- 160 of them -- ran this kernel at a third of the speed)
    const int nl = max(n - 1, 0);
    double T = ray_pre[q].x;
    float gT[kMaxRaySegs];
#pragma unroll
    for (int s0 = 0; s0 < kMaxRaySegs; s0 += NB) {
        if (s0 < n || s0 == 0) {
            float P[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) P[u] = b[(size_t)min(s0 + u, nl) * rr].x;
#pragma unroll
            for (int u = 0; u < NB; u++) {
                gT[s0 + u] = (float)(g * T);
                if (s0 + u < n) T *= (double)P[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < NB; u++) gT[s0 + u] = 0.f;
        }
    }
    // R in fp64; what leaves is the DIFFERENCE seg_scatter_kernel starts from, w_last - R behind the segment (rounded to fp32 relative
    // to itself, not to a depth).  In front of a segment: R = w_first + S + P (R behind - w_first)  (seg_combine_kernel: S is relative)
    double R = 1.0;                                                      // behind the last sample: prod(1-p) * 1  (:69-71)
#pragma unroll
    for (int s0 = kMaxRaySegs - NB; s0 >= 0; s0 -= NB) {
        if (s0 < n) {
            float2 v[NB], lw[NB];                                          // (P, S), (first, last depth weight)
#pragma unroll
            for (int u = 0; u < NB; u++) {
                v[u] = b[(size_t)min(s0 + u, nl) * rr];
                lw[u] = line_w[(size_t)min(s0 + u, nl) * rr + q];
            }
#pragma unroll
            for (int u = NB - 1; u >= 0; u--) {
                if (s0 + u < n) {
                    t[(size_t)(s0 + u) * rr] = make_float2(gT[s0 + u], (float)((double)lw[u].y - R));
                    R = (double)lw[u].x + (double)v[u].y + (double)v[u].x * (R - (double)lw[u].x);
                }
            }
        }
    }
    // this block's max |g T| in front of a ray's first segment (T only falls along a ray), as a bit pattern -- non-negative floats
    // order like their bits, Inf / NaN sort above every finite value, so a non-finite gradient survives --: seg_scatter_kernel
    // derives its image's fixed-point scale from the blocks' maxima (no atomics, nothing to clear)
    unsigned m = n > 0 ? (__float_as_uint(gT[0]) & 0x7fffffffu) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < NT / 64; w++) m = max(m, red[w]);
        bmax[(size_t)img * gridDim.x + blockIdx.x] = m;
    }
}

// adds every live row's halo record onto the voxels it belongs to (other bricks' low faces): behind seg_scatter_kernel's stores.
// A wave per row, four rows per workgroup (a quarter of the workgroups to dispatch when nothing comes back through an image)
__global__ __launch_bounds__(kNT) void seg_halo_kernel(RenderDims D, View5 gvox, const int4 *__restrict__ rows, int nrows,
                                                        const float *__restrict__ halo, const int *__restrict__ live)
{
    const int r = blockIdx.x * (kNT / 64) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nrows) return;
    const int img = blockIdx.y * D.NC + blockIdx.z;
    if (live != nullptr) {                                               // (the image's word before the row: one round trip for a dead image)
        const int nbricks = ((D.X + kBrick - 1) >> 4) * ((D.Y + kBrick - 1) >> 4) * ((D.Z + kBrick - 1) >> 4);
        if (live[(int64_t)img * (nbricks + 1)] == 0) return;
    }
    const int4 row = rows[r];
    if (row_dead(D, row, live, img)) return;
    const RowBits b = row_bits(row);
    const float *rec = halo + ((size_t)img * nrows + r) * kHaloRec;
    float *gb = gvox.p + blockIdx.y * gvox.s0 + blockIdx.z * gvox.s1;
    constexpr int kPer = (kHaloN + 63) / 64;
    float v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; i++) v[i] = rec[min(lane + i * 64, kHaloN - 1)];
#pragma unroll
    for (int i = 0; i < kPer; i++) {
        const int idx = lane + i * 64;
        if (idx >= kHaloN || v[i] == 0.0f) continue;                      // (cells outside the volume were recorded as zeros)
        int lx, ly, lz;
        if (idx < kHaloY) { lx = kBrick; ly = (int)(((float)idx + 0.5f) * (1.0f / kHF)); lz = idx - ly * kHF; }
        else if (idx < kHaloZ) { const int j = idx - kHaloY; lx = (int)(((float)j + 0.5f) * (1.0f / kHF)); ly = kBrick; lz = j - lx * kHF; }
        else { const int j = idx - kHaloZ; lx = j >> 4; ly = j & 15; lz = kBrick; }
        unsafeAtomicAdd(gb + (b.bx0 + lx) * gvox.s2 + (b.by0 + ly) * gvox.s3 + (b.bz0 + lz) * gvox.s4, v[i]);
    }
}

// ---- backward, accumulation: lane = segment, the brick's tile as 64-bit fixed point in LDS -------------------------------------------
// One workgroup per row of bwd_rows (a brick, or a piece of a heavy one), one image.  A lane takes one segment: the saved values of
// its <= 16 samples (one 64-byte slot), (g T in front, w_last - R behind) from seg_combine_bwd_kernel; forwards g T_k, backwards
//   dL/dp_k = g T_k d_k,   d_k = w_k - R_{k+1}:   d_{k-1} = (w_{k-1} - w_k) + (1 - p_k) d_k        (R_k = R_{k+1} + p_k d_k)
// (the DIFFERENCE carried, not R: roundings relative to a fraction of the depth range), and per sample the position, cell and
// ATen's eight trilinear weights exactly as the forward formed them, accumulated into the tile -- 18^3 cells: the brick, the
// voxels one step beyond its high faces (a sample is listed under the brick of its base corner) and one plane in front of the low
// faces (base corner -1: zero padding, never flushed): no ownership tests.  Every sample is visited ONCE (round 5's brick-owned
// kernel listed a sample under every brick it touches, 1.37 visits, at one lane per visit with the position recomputed in fp64;
// pulling whole segments would be 1.8) and nothing per sample goes through memory between the scan and the scatter.
//   * The tile is 64-bit FIXED POINT: LDS integer atomics cost nothing next to the arithmetic here whatever the lanes' addresses
//     (64 lanes adding to near-by, often identical cells), floating-point ones do -- measured on this kernel at batch 32:
//     ds_add_u64 914 us (no atomics at all: 930), ds_add_f64 1564, ds_add_f32 2324.  Scale 2^(44-e) with 2^e above a BOUND of the
//     image's |dL/dp|, max |g T| in front of a ray (seg_combine_bwd_kernel's per-block maxima) times the span of the depth
//     weights and 1 (|w_k - R| cannot exceed it): a few bits of the 44 unused, no pass over the samples.  Sums of up to 2^17
//     contributions stay below 2^63; integer sums do not depend on the order.  A non-finite bound: the image's gradient is NaN.
//   * Consecutive samples of a ray are a quarter to half a voxel apart; the eight corner sums of a cell are formed in registers
//     (fp32, a handful of terms) and go to the tile when the cell changes.
//   * The next chunk's entry and slot are requested before the current chunk's arithmetic, its chain line and direction after.
// The flush applies the adjoint of clamp(vox * pre_scale) -- a per-voxel select and scale, so it distributes over partial sums --
// and adds the tile into grad_vox, which the host entry zeroes first: voxels only this workgroup can reach (inside the brick,
// away from its low faces, brick not split over rows) with plain stores, the rest (its low faces, which the tiles of the bricks
// in front reach too; the halo, = other bricks' low faces; split bricks) with fp32 atomics, skipping zeros.
#ifndef GENRE_SCATTER_AB
#define GENRE_SCATTER_AB 0
#endif
constexpr int kAT = kBrick + 2;                                          // tile edge: voxel locals -1 .. 16
constexpr int kATn = kAT * kAT * kAT;
// Threads per workgroup (template): 512 = two workgroups (99 KB of LDS) per CU for small batches, where a row's eight waves finish
// it in one chunk each (batch 1: 50.6 us against 52.9); 256 = three per CU for batches of 16 images and more, where the smaller
// workgroups turn over better (batch 32: 792 -> 740 us).  Both at <= 128 VGPRs (four waves per SIMD); 128 and 384 threads: slower.
template <int kNTs>
__global__ __launch_bounds__(kNTs) __attribute__((amdgpu_waves_per_eu(4, 4))) void seg_scatter_kernel(RenderDims D, View5 vox, View5 gvox, const double *__restrict__ dirs,
                                                           const float *__restrict__ dw, const int4 *__restrict__ rows,
                                                           const int4 *__restrict__ segs, int nseg,
                                                           const float2 *__restrict__ tr, int lines,
                                                           const float *__restrict__ vbuf, const int *__restrict__ live,
                                                           const unsigned *__restrict__ bmax, int nblk, float *__restrict__ halo)
{
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    unsigned long long *tile = reinterpret_cast<unsigned long long *>(lds_d);   // [kATn]
    float *dw_s = reinterpret_cast<float *>(lds_d + kATn);               // [kMaxZR]
    __shared__ unsigned red_m[kNTs / 64];
    __shared__ float red_lo[kNTs / 64], red_hi[kNTs / 64];
    const int4 row = rows[blockIdx.x];
    const int img = blockIdx.y * D.NC + blockIdx.z;
    const RowBits rb = row_bits(row);
    const bool split = rb.split;                                         // other rows accumulate into this brick too
    const int bx0 = rb.bx0, by0 = rb.by0, bz0 = rb.bz0;
    float *gb = gvox.p + blockIdx.y * gvox.s0 + blockIdx.z * gvox.s1;
    if (row_dead(D, row, live, img)) {                                   // nothing comes back: this brick's voxels are zeros
        if (!split || rb.first) zero_brick(D, gvox, gb, rb, kNTs);       // (a split brick: its first row; all its rows are dead together)
        return;
    }
    const int ox = bx0 - 1, oy = by0 - 1, oz = bz0 - 1;                  // tile origin
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float2 *__restrict__ trimg = tr + (size_t)img * lines;
    // the first chunk's entry and slot (addresses known now), in flight across the set-up below
    int c0 = row.y + wave * 64;
    int sc = min(c0 + lane, row.z - 1);
    int4 e = segs[sc];
    float4 pv[kSegSlot / 4];
    {
        const float4 *slot = slot_q0(vbuf, img, nseg, sc);
#pragma unroll
        for (int j = 0; j < kSegSlot / 4; j++) pv[j] = slot[j * 64];
    }
    unsigned mb = 0u;
    for (int i = threadIdx.x; i < nblk; i += kNTs) mb = max(mb, bmax[(size_t)img * nblk + i]);
    float wlo = 1.0f, whi = 1.0f;                                        // (R behind a ray's last sample is 1)
    for (int k = threadIdx.x; k < kMaxZR; k += kNTs) {
        const int kk = min(k, D.ZR - 1);
        const float w = dw[kk];
        dw_s[k] = w;
        wlo = fminf(wlo, w); whi = fmaxf(whi, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
        wlo = fminf(wlo, __shfl_xor(wlo, o, 64)); whi = fmaxf(whi, __shfl_xor(whi, o, 64));
    }
    if (lane == 0) { red_m[wave] = mb; red_lo[wave] = wlo; red_hi[wave] = whi; }
    for (int i = threadIdx.x; i < kATn; i += kNTs) tile[i] = 0ull;
    float2 st = trimg[e.z];                                               // (g T in front of the segment, w_last - R behind its end)
    double d2x = dirs[e.x * 3 + 0], d2y = dirs[e.x * 3 + 1], d2z = dirs[e.x * 3 + 2];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kNTs / 64; w++) { mb = max(mb, red_m[w]); wlo = fminf(wlo, red_lo[w]); whi = fmaxf(whi, red_hi[w]); }
    const float bound = __uint_as_float(mb) * (whi - wlo);               // >= |dL/dp| of every sample of this image
    const bool nonfinite = !(bound <= 3.0e38f);                          // an Inf / NaN upstream gradient (or depth weight)
    int ex = 0;
    (void)frexpf(nonfinite ? 1.0f : bound, &ex);                         // bound < 2^ex
    const double scale = ldexp(1.0, 44 - ex), inv_scale = ldexp(1.0, ex - 44);

    for (; c0 < row.z && !nonfinite; c0 += kNTs) {                       // (waves run on their own: no barrier in here)
        const bool act = c0 + lane < row.z;
        const int k0 = e.y & 255, L = act ? min(e.y >> 8, kSegSlot) : 0;
        const double dx2 = d2x * 2, dy2 = d2y * 2, dz2 = d2z * 2;
        float p[kSegSlot], gt[kSegSlot];
#pragma unroll
        for (int j = 0; j < kSegSlot / 4; j++) { p[4 * j] = pv[j].x; p[4 * j + 1] = pv[j].y; p[4 * j + 2] = pv[j].z; p[4 * j + 3] = pv[j].w; }
        float Tg = st.x, d = st.y;
        // the next chunk's entry and slot (the last chunk re-reads its own)
        sc = min(c0 + kNTs + lane, row.z - 1);
        e = segs[sc];
        {
            const float4 *slot = slot_q0(vbuf, img, nseg, sc);
#pragma unroll
            for (int j = 0; j < kSegSlot / 4; j++) pv[j] = slot[j * 64];
        }
        unsigned pass = 0u;
        float Tlo = 0.f;
#pragma unroll
        for (int i = 0; i < kSegSlot; i++) {                              // forwards: g T_k, carried as T + Tlo (see the sampler)
            const float raw = p[i];
            p[i] = fminf(fmaxf(raw, D.lo), D.hi);                         // clamp(., 1e-5, 1 - 1e-5)  (spherical_proj.py:66)
            pass |= (raw >= D.lo && raw <= D.hi && i < L) ? 1u << i : 0u; // torch.clamp's backward mask
            gt[i] = Tg + Tlo;
            if (i < L) {
                const float f = 1.0f - p[i], t = Tg * f;
                Tlo = __builtin_fmaf(Tlo, f, __builtin_fmaf(Tg, f, -t));
                Tg = t;
            }
        }
        // backwards: d_k, and dL/dp_k in the place of g T_k.  ALL of this chunk's LDS reads happen here, in front of its atomics: LDS
        // operations complete in order, and a read behind a spill would wait for the spill's eight conflict-laden atomics
#pragma unroll
        for (int i = kSegSlot - 1; i >= 0; i--) {
            const int k = min(k0 + i, kMaxZR - 1);
            if (i + 1 < kSegSlot) {
                const float dn = __builtin_fmaf(1.0f - p[min(i + 1, kSegSlot - 1)], d, dw_s[k] - dw_s[min(k + 1, kMaxZR - 1)]);
                d = i + 1 < L ? dn : d;
            }
            gt[i] = (pass >> i & 1u) ? gt[i] * d : 0.f;                   // (pass: inside the segment and through the clamp)
        }
        // the scatter: position, cell and weights of every sample as the forward formed them (spherical_proj.py:50-56; 1 - alpha_k
        // in fp64 arithmetic here, not from a table in LDS: see above)
        int cur = -1;                                                     // tile index of the cell whose sums are in acc[]
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.f;
        auto spill = [&]() {
            unsigned long long *tp = tile + cur;
#pragma unroll
            for (int j = 0; j < 8; j++) {
#if GENRE_SCATTER_AB == 1                                                 // (A/B: no LDS atomics)
                if (acc[j] == 12345.678f)
#endif
                // one fp64 fma with the power-of-two scale and 1.5 * 2^52: the rounded product is an integer in the mantissa
                atomicAdd(tp + ((j & 1) ? kAT * kAT : 0) + ((j & 2) ? kAT : 0) + ((j & 4) ? 1 : 0),
                          (unsigned long long)(__double_as_longlong(fma((double)acc[j], scale, 6755399441055744.0)) -
                                               0x4338000000000000LL));    // ds_add_u64
                acc[j] = 0.f;
            }
        };
#pragma unroll
        for (int i = 0; i < kSegSlot; i++) {
            if (__ballot(i < L) == 0ull) break;                           // (a wave's segments are neighbours in length)
            const int k = min(k0 + i, D.ZR - 1);
            const double a = 1.0 - ((k == D.ZR - 1) ? 1.0 : (double)k * D.step);   // numpy.linspace(0,1,ZR)[k]  (render_common.hpp: sample_pos)
            const float gx = (float)(dx2 * a), gy = (float)(dy2 * a), gz = (float)(dz2 * a);
            Cell c;
            locate(D, gx, gy, gz, c);
            const int idx = ((c.x0 - ox) * kAT + (c.y0 - oy)) * kAT + (c.z0 - oz);
            const float dp = gt[i];
            const bool on = dp != 0.0f;                                   // (NaN != 0: a non-finite gradient goes through)
#if GENRE_SCATTER_AB == 5                                                 // (A/B: one spill per segment)
            if (on) cur = idx;
#elif GENRE_SCATTER_AB == 9                                               // (A/B: eight atomics per sample, no register sums)
            if (on) {
                unsigned long long *tp = tile + idx;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    atomicAdd(tp + ((j & 1) ? kAT * kAT : 0) + ((j & 2) ? kAT : 0) + ((j & 4) ? 1 : 0),
                              (unsigned long long)(__double_as_longlong(fma((double)(corner_w(c, j) * dp), scale, 6755399441055744.0)) -
                                                   0x4338000000000000LL));
            }
            continue;
#else
            if (on && idx != cur) {
                if (cur >= 0) spill();
                cur = idx;
            }
#endif
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += corner_w(c, j) * dp;    // ATen's weight (fp32, its product order) times dL/dp (0 off the segment)
        }
        if (cur >= 0) spill();
        // what needs the next entry
        st = trimg[e.z];
        d2x = dirs[e.x * 3 + 0]; d2y = dirs[e.x * 3 + 1]; d2z = dirs[e.x * 3 + 2];
    }
    // the clamp-mask values of the flush below: requested before the barrier (unconditional loads, addresses clamped into the volume)
    const float *vb = vox.p + blockIdx.y * vox.s0 + blockIdx.z * vox.s1;
    const bool v4 = D.sz == 1 && ((D.sx | D.sy) & 3) == 0 && (reinterpret_cast<uintptr_t>(vb) & 15) == 0 && gvox.s4 == 1 &&
                    ((gvox.s2 | gvox.s3) & 3) == 0 && (reinterpret_cast<uintptr_t>(gb) & 15) == 0 && bz0 + kBrick <= D.Z;
    constexpr int kFlushIt = (kHF * kHF * 4 + kNTs - 1) / kNTs;
    constexpr int kZIt = (kHF * kHF + kNTs - 1) / kNTs;
    float4 mvv[kFlushIt];
    float mvz[kZIt];
#pragma unroll
    for (int i = 0; i < kZIt; i++) mvz[i] = 0.f;
#pragma unroll
    for (int i = 0; i < kFlushIt; i++) mvv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (D.pre_scale != 0.0f) {
#pragma unroll
        for (int i = 0; i < kFlushIt; i++) {
            const int it = (int)threadIdx.x + i * kNTs;
            const int r = it >> 2, lz0 = (it & 3) * 4;
            const int lx = (int)(((float)r + 0.5f) * (1.0f / kHF)), ly = r - lx * kHF;
            const int x = bx0 + lx, y = by0 + ly, z = bz0 + lz0;
            const bool ok = it < kHF * kHF * 4 && x < D.X && y < D.Y;
            const int base = ok ? x * D.sx + y * D.sy : 0;
            if (v4) mvv[i] = *reinterpret_cast<const float4 *>(vb + base + (ok ? z : 0));
            else {
                float m[4];
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = vb[base + ((ok && z + c < D.Z) ? (z + c) * D.sz : 0)];
                mvv[i] = make_float4(m[0], m[1], m[2], m[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < kZIt; i++) {
            const int r = min((int)threadIdx.x + i * kNTs, kHF * kHF - 1);
            const int lx = (int)(((float)r + 0.5f) * (1.0f / kHF)), ly = r - lx * kHF;
            const int x = bx0 + lx, y = by0 + ly, z = bz0 + kBrick;
            mvz[i] = vb[(x < D.X && y < D.Y && z < D.Z) ? x * D.sx + y * D.sy + z * D.sz : 0];
        }
    }
    __syncthreads();
#if GENRE_SCATTER_AB == 6                                                 // (A/B: no flush)
    if (tile[threadIdx.x] != 0x123456789ull) return;
#endif
    // ---- flush: the tile's 17 x 17 z rows in four pieces of four cells (+ the cell at z = 16), through the adjoint of
    // clamp(vox * pre_scale) -- a per-voxel select and scale, so it distributes over the partial sums of a voxel that several tiles
    // reach.  The brick's own voxels go to grad_vox (16-byte stores when the rows allow it; atomics in a split brick), the cells
    // beyond its high faces to the row's halo record (zeros for cells outside the volume) ----------------------------------------------
    float *rec = halo + ((size_t)img * gridDim.x + blockIdx.x) * kHaloRec;
    auto cell = [&](const unsigned long long raw, const float mv, const bool ok) {
        float val = (float)((double)(long long)raw * inv_scale);
        if (nonfinite) val = __uint_as_float(0x7fc00000u);                // the reference chain would return NaN here too
        if (D.pre_scale != 0.0f) {                                        // adjoint of clamp(x * pre_scale, lo, hi): a select
            const float t = mv * D.pre_scale;
            val = (t >= D.lo && t <= D.hi) ? val * D.pre_scale : 0.0f;
        }
        return ok ? val : 0.0f;
    };
#pragma unroll
    for (int i = 0; i < kFlushIt; i++) {
        const int it = (int)threadIdx.x + i * kNTs;
        if (it >= kHF * kHF * 4) break;
        const int r = it >> 2, lz0 = (it & 3) * 4;
        const int lx = (int)(((float)r + 0.5f) * (1.0f / kHF)), ly = r - lx * kHF;
        const int x = bx0 + lx, y = by0 + ly, z = bz0 + lz0;
        const bool in_xy = x < D.X && y < D.Y;
        const float mv[4] = {mvv[i].x, mvv[i].y, mvv[i].z, mvv[i].w};
        const unsigned long long *tp = tile + ((lx + 1) * kAT + ly + 1) * kAT + lz0 + 1;
        float val[4];
#pragma unroll
        for (int c = 0; c < 4; c++) val[c] = cell(tp[c], mv[c], in_xy && z + c < D.Z);
        if (lx < kBrick && ly < kBrick) {                                 // the brick's own voxels
            if (!in_xy) continue;
            float *dst = gb + x * gvox.s2 + y * gvox.s3 + z * gvox.s4;
            if (!split) {
                if (v4) *reinterpret_cast<float4 *>(dst) = make_float4(val[0], val[1], val[2], val[3]);
                else {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (z + c < D.Z) dst[c * gvox.s4] = val[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (val[c] != 0.0f) unsafeAtomicAdd(dst + c * gvox.s4, val[c]);
            }
        } else {                                                          // x face (local x = 16), else y face
            float *h = rec + (lx == kBrick ? kHaloX + ly * kHF + lz0 : kHaloY + lx * kHF + lz0);
#pragma unroll
            for (int c = 0; c < 4; c++) h[c] = val[c];
        }
    }
#pragma unroll
    for (int i = 0; i < kZIt; i++) {                                      // the cells at local z = 16
        const int r = (int)threadIdx.x + i * kNTs;
        if (r >= kHF * kHF) break;
        const int lx = (int)(((float)r + 0.5f) * (1.0f / kHF)), ly = r - lx * kHF;
        const int x = bx0 + lx, y = by0 + ly, z = bz0 + kBrick;
        const bool ok = x < D.X && y < D.Y && z < D.Z;
        const float val = cell(tile[((lx + 1) * kAT + ly + 1) * kAT + kBrick + 1], mvz[i], ok);
        rec[lx == kBrick ? kHaloX + ly * kHF + kBrick : (ly == kBrick ? kHaloY + lx * kHF + kBrick : kHaloZ + lx * kBrick + ly)] = val;
    }
}

}  // namespace

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
                          v_scratch->size[0] >= (int64_t)imgs * ((segs->size[0] + 63) / 64) * 64 * kSegSlot,
                      "%s: v_scratch must be a contiguous, 16-byte aligned fp32 buffer of >= N*NC * ceil(nseg/64)*64 * %d elements", op, kSegSlot);
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


// The backward of genre_render_seg_forward (include/genre_hip.h): per-ray chains over the forward's (P, S) pairs, then one pass over
// the segments -- dL/dp of every sample from the values the forward saved, scattered straight into the brick's tile
extern "C" int genre_render_seg_backward(const genre_tensor *vox, const genre_tensor *dirs, const genre_tensor *depth_weight,
                                         const genre_tensor *grad_out, const genre_tensor *grad_vox, const genre_tensor *bwd_rows,
                                         const genre_tensor *segs, const genre_tensor *ray_nseg, const genre_tensor *ray_pre,
                                         const genre_tensor *line_w, const genre_tensor *ps_scratch,
                                         const genre_tensor *tr_scratch, const genre_tensor *v_scratch,
                                         const genre_tensor *halo_scratch, const genre_tensor *live, float pre_scale,
                                         void *stream)
{
    using namespace genre;
    const char *op = "render_seg_backward";
    RenderDims D{};
    if (!check_render(op, vox, dirs, depth_weight, grad_out, D)) return 0;
    D.pre_scale = pre_scale;
    const int imgs = D.N * D.NC, rr = D.R * D.R;
    hipStream_t st = (hipStream_t)stream;
    GENRE_REQUIRE(is_f32(grad_vox, 5) && same_shape(grad_vox, vox) && is_contiguous(grad_vox),
                  "%s: grad_vox must be a contiguous fp32 tensor of the shape of vox", op);
    const int64_t nv = numel(grad_vox);
    if (nv == 0) return 1;
    if ((int64_t)imgs * rr == 0)                                         // no rays: the gradient of nothing
        return hipMemsetAsync(grad_vox->data, 0, (size_t)nv * sizeof(float), st) == hipSuccess ? 1 : fail("%s: hipMemsetAsync failed", op);
    const int nb = ((D.X + kBrick - 1) / kBrick) * ((D.Y + kBrick - 1) / kBrick) * ((D.Z + kBrick - 1) / kBrick);
    GENRE_REQUIRE(D.ZR <= kMaxZR && (int64_t)rr < (1 << 24) && D.N <= 65535 && D.NC <= 65535,
                  "%s: needs ZR <= 256, R*R < 2^24, N and NC <= 65535", op);
    GENRE_REQUIRE(is_i32(bwd_rows, 2) && bwd_rows->size[1] == 4 && is_contiguous(bwd_rows) && bwd_rows->size[0] >= 1 &&
                      bwd_rows->size[0] < (1 << 30) && aligned16(bwd_rows->data),
                  "%s: bwd_rows must be a contiguous, 16-byte aligned int32 [rows >= 1, 4] tensor", op);
    GENRE_REQUIRE(is_i32(segs, 2) && segs->size[1] == 4 && segs->size[0] >= 1 && is_contiguous(segs) && aligned16(segs->data),
                  "%s: segs must be a contiguous, 16-byte aligned int32 [nseg >= 1, 4] tensor", op);
    GENRE_REQUIRE(is_i32(ray_nseg, 1) && is_contiguous(ray_nseg) && ray_nseg->size[0] == rr, "%s: ray_nseg must be int32 [R*R]", op);
    GENRE_REQUIRE(is_f32(ray_pre, 2) && is_contiguous(ray_pre) && ray_pre->size[0] == rr && ray_pre->size[1] == 4 &&
                      aligned16(ray_pre->data), "%s: ray_pre must be the float64 [R*R, 2] prefix table viewed as fp32 [R*R, 4]", op);
    GENRE_REQUIRE(is_f32(ps_scratch, 1) && is_contiguous(ps_scratch) && ((uintptr_t)ps_scratch->data & 7u) == 0 &&
                      ps_scratch->size[0] % ((int64_t)2 * imgs * rr) == 0 && ps_scratch->size[0] > 0 &&
                      ps_scratch->size[0] / (2 * imgs) < ((int64_t)1 << 31) && ps_scratch->size[0] / ((int64_t)2 * imgs * rr) <= kMaxRaySegs,
                  "%s: ps_scratch must be the forward's fp32 [N*NC * smax*R*R * 2] buffer (smax <= %d)", op, kMaxRaySegs);
    GENRE_REQUIRE(is_f32(line_w, 2) && is_contiguous(line_w) && line_w->size[1] == 2 && ((uintptr_t)line_w->data & 7u) == 0 &&
                      line_w->size[0] * (int64_t)2 * imgs == ps_scratch->size[0],
                  "%s: line_w must be fp32 [smax*R*R, 2], one pair per scratch line of an image", op);
    GENRE_REQUIRE(is_f32(tr_scratch, 1) && is_contiguous(tr_scratch) && ((uintptr_t)tr_scratch->data & 7u) == 0,
                  "%s: tr_scratch must be a contiguous, 8-byte aligned fp32 buffer", op);
    GENRE_REQUIRE(is_f32(v_scratch, 1) && is_contiguous(v_scratch) && aligned16(v_scratch->data) &&
                      v_scratch->size[0] >= (int64_t)imgs * ((segs->size[0] + 63) / 64) * 64 * kSegSlot,
                  "%s: v_scratch must be the forward's fp32 [N*NC * ceil(nseg/64)*64 * %d] buffer", op, kSegSlot);
    const int *live_p = nullptr;                                        // the forward's clamp pass words (pre_scale only)
    if (pre_scale != 0.0f) {
        GENRE_REQUIRE(is_i32(live, 1) && is_contiguous(live) && live->size[0] >= (int64_t)imgs * (nb + 1),
                      "%s: with pre_scale, live must be the forward's int32 [N*NC*(1 + bricks)] buffer (it says which slots of "
                      "v_scratch hold values)", op);
        live_p = (const int *)live->data;
    }
    const int lines = (int)(ps_scratch->size[0] / (2 * imgs));
    const bool big = imgs * (int64_t)rr >= 65536 * 4;
    const int nblk = big ? (rr + 255) / 256 : (rr + 63) / 64;               // blocks of seg_combine_bwd_kernel per image
    GENRE_REQUIRE(tr_scratch->size[0] >= ps_scratch->size[0] + (int64_t)imgs * nblk,
                  "%s: tr_scratch must hold numel(ps_scratch) + N*NC * ceil(R*R / 64) floats", op);
    unsigned *bmax = (unsigned *)tr_scratch->data + ps_scratch->size[0];   // per image and block: max |g T|, behind the lines
    GENRE_REQUIRE(is_f32(halo_scratch, 1) && is_contiguous(halo_scratch) &&
                      halo_scratch->size[0] >= (int64_t)imgs * bwd_rows->size[0] * kHaloRec,
                  "%s: halo_scratch must be a contiguous fp32 buffer of >= N*NC * rows * %d elements", op, kHaloRec);
    const dim3 rgrid((unsigned)bwd_rows->size[0], D.N, D.NC);
    if (big)
        seg_combine_bwd_kernel<256, 8><<<dim3((rr + 255) / 256, D.N, D.NC), 256, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines,
            view4(grad_out), (float2 *)tr_scratch->data, live_p, nb, (const float2 *)line_w->data, bmax,
            (const int4 *)bwd_rows->data, (int)bwd_rows->size[0], view5(grad_vox));
    else
        seg_combine_bwd_kernel<64, 32><<<dim3((rr + 63) / 64, D.N, D.NC), 64, 0, st>>>(
            D, (const float2 *)ps_scratch->data, (const int *)ray_nseg->data, (const double2 *)ray_pre->data, lines,
            view4(grad_out), (float2 *)tr_scratch->data, live_p, nb, (const float2 *)line_w->data, bmax,
            (const int4 *)bwd_rows->data, (int)bwd_rows->size[0], view5(grad_vox));
    GENRE_LAUNCH_CHECK("render_seg backward (segment chains)");
    constexpr size_t lds = (size_t)kATn * sizeof(double) + kMaxZR * sizeof(float);
    static_assert(lds <= 64 * 1024, "dynamic LDS beyond 64 KB needs reserve_lds");
    const auto scatter = imgs >= 16 ? &seg_scatter_kernel<256> : &seg_scatter_kernel<512>;
    scatter<<<rgrid, imgs >= 16 ? 256 : 512, lds, st>>>(
        D, view5(vox), view5(grad_vox), (const double *)dirs->data, (const float *)depth_weight->data,
        (const int4 *)bwd_rows->data, (const int4 *)segs->data, (int)segs->size[0], (const float2 *)tr_scratch->data, lines,
        (const float *)v_scratch->data, live_p, bmax, nblk, (float *)halo_scratch->data);
    GENRE_LAUNCH_CHECK("render_seg backward (scatter)");
    seg_halo_kernel<<<dim3((rgrid.x + kNT / 64 - 1) / (kNT / 64), D.N, D.NC), kNT, 0, st>>>(
        D, view5(grad_vox), (const int4 *)bwd_rows->data, (int)bwd_rows->size[0], (const float *)halo_scratch->data, live_p);
    GENRE_LAUNCH_CHECK("render_seg backward (halo records)");
    return 1;
}
