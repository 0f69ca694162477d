// sph_render_bm.hip -- batch-minor TILE renderer: fused voxel -> spherical depth map for batches, gfx950.
//
// Same operator as sph_render.hip -- render_spherical.forward of the reference (toolbox/spherical_proj.py:62-72:
// grid_sample (PyTorch 0.4.1 == align_corners=True), clamp, CalcStopProb (calc_prob_kernel.cu:113-189),
// matmul(depth_weight), prod(1-p), add) and its autograd backward -- for volumes whose IMAGE index is fastest in
// memory (element (n,x,y,z) at x*sx + y*sy + z*sz + n).  Then 32 lanes can be 32 images of ONE sample, and everything
// that depends on the geometry only (where a sample falls, its eight trilinear weights, which brick holds it, how
// a ray is cut into per-brick segments) comes from tables built once per geometry on the host
// (toolbox/_bm_tables.py); the kernels execute no geometry arithmetic at all.
//
//  * Bricks of 4 x 8 x 8 voxels.  A SEGMENT = a run of consecutive samples of one ray whose base voxel lies in one
//    brick (<= 32 samples, 8.7 on average at 128^3 / 128^2 rays / 256 samples).
//  * The ray integral is associative: a segment contributes (P, S) = (prod(1-p_k), sum_k T_k p_k w_k with T = 1 at
//    its start); a per-ray pass chains the segments.  Nothing per SAMPLE ever goes through HBM in the forward
//    (sph_render.hip writes and re-reads 16 MiB of raw sample values per image).
//  * bm_sample_kernel: a workgroup stages its brick + the high halo (5 x 9 x 9 voxel lines of 32 images, 51 KB) in
//    LDS; a WAVE marches one segment serially.  Lane = (image, z half): the two half-waves read the z0 / z0+1
//    corner lines of the same sample -- adjacent in LDS, 256 contiguous bytes, conflict-free by construction -- and
//    exchange their partial sums with one v_permlane32_swap.  Per sample: 2 reads of the 48-byte record,
//    4 ds_read_b32, 4 multiply-adds, the clamp, 3 scan operations.
//  * WAITS.  gfx950 counts vector loads AND stores in one in-order counter (vmcnt), and the compiler places the waits:
//    to wait for an older load while younger operations stay in flight it must know their number at compile time, on
//    every path.  Both kernels are written to that rule -- a fixed number of loads per prefetch, on every path (the
//    tables are padded for what is read past a segment's own data), unconditional use of what was loaded, stores issued
//    BEFORE the next prefetch and not after it -- or their software pipelines overlap nothing (see the kernels).
//  * Backward: dL/dp_k = g T_k (w_k - R_{k+1}),  R_k = p_k w_k + (1 - p_k) R_{k+1},  R_end = 1 -- no division, no
//    cancellation.  bm_combine_bwd_kernel leaves g*T at the start and R behind the end of every segment;
//    bm_scatter_kernel: every brick PULLS the segments that touch one of its voxels, re-runs their (cheap) serial
//    scans from the saved clamped samples p (the only per-sample stream: 4 B per sample and image, written by the
//    forward when a gradient is wanted; ownership of a corner is a scalar-unit execution mask, not a vector compare),
//    and accumulates the trilinear adjoint of the samples it owns corners of in an
//    fp64 LDS tile it alone owns -- ds_add_f64 runs at 8.7 clk per 64-lane instruction on gfx950 against 193 for
//    ds_add_f32 (tools/bm_tile_bench.hip) -- then writes every voxel of grad_vox exactly once with plain stores.
//    fp64 accumulation needs no scale (sph_render.hip's 64-bit fixed point needs a batch-global max |dL/dp| pass),
//    propagates NaN / Inf gradients, and is order-independent to ~1e-16.
//  * The caller-side clamp(proj * 50, 1e-5, 1 - 1e-5) (depth_pred_with_sph_inpaint.py:124) folds into the tile load
//    (pre_scale); its adjoint mask is one bit per voxel and image, written by the forward (the brick that owns the
//    voxel), read by the backward's flush -- the volume itself is not read again.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace genre {
namespace {

#ifndef GENRE_BM_BY
#define GENRE_BM_BY 8                                    // 4: half bricks, 512-thread workgroups, four per CU (A/B: tools/build_variants.sh)
#endif
// Per-workgroup timeline of bm_sample_kernel (variant build -DGENRE_BM_TIMELINE, tools/bm_timeline.py; never in the shipped library)
#ifdef GENRE_BM_TIMELINE
#define GENRE_BTL_PARAM , unsigned long long *tlbuf_
#define GENRE_BTL_ARG , tl_buf
#define GENRE_BTL(i) do { if (tlbuf_ && threadIdx.x == 0) tlbuf_[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GENRE_BTL_PARAM
#define GENRE_BTL_ARG
#define GENRE_BTL(i) do {} while (0)
#endif
constexpr int kBX = 4, kBY = GENRE_BM_BY, kBZ = 8;       // brick (must match toolbox/_bm_tables.py: genre_bm_brick())
constexpr int kTX = kBX + 1, kTY = kBY + 1, kTZ = kBZ + 1;
constexpr int kLinesF = kTX * kTY * kTZ;                 // 405 voxel lines in the forward tile
constexpr int kImgs = 32;                                // images per group = lanes of a half-wave
constexpr int kMaxSeg = 16;
constexpr int kRec = 12;                                 // words per sample record
constexpr int kThreads = 512;
constexpr int kRecL = 16;                                // words per record slot in LDS (sampler and scatter kernel)
constexpr int kRayRegs = 31;                             // segments of a ray the per-ray backward pass keeps in registers

struct BmDims {
    int N, X, Y, Z, R, pad;
    int nseg, groups, ZR, nbricks;                       // nbricks: bricks of the forward's tiling (tile_live words per group)
    int64_t nslot;
    int64_t sx, sy, sz;                                  // element strides of vox (image stride == 1)
    int64_t gx, gy, gz;                                  // ... of grad_vox
    float pre_scale, lo, hi;
};

__device__ __forceinline__ float other_half_sum(float v)
{
    // lanes 0-31 and 32-63 hold the z0 / z0+1 partial sums of the same (sample, image): v_permlane32_swap
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order: only the compiler has to be kept from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A row's fourth word = flag (0 plain, 1 shared, 2 padding) | bx << 8 | by << 16 | bz << 24: the brick's coordinates come with the
// row (toolbox/_bm_tables.py: _pack_coords) -- gfx950 has no integer divide, taking a brick id apart cost every workgroup three
// ~40-instruction divisions in front of its first load
__device__ __forceinline__ int row_flag(const int4 &row) { return row.w & 255; }
template <int BX = kBX, int BY = kBY, int BZ = kBZ>
__device__ __forceinline__ void brick_origin(const int4 &row, int &ox, int &oy, int &oz)
{
    ox = ((row.w >> 8) & 255) * BX; oy = ((row.w >> 16) & 255) * BY; oz = ((row.w >> 24) & 255) * BZ;
}

// ---- forward: brick sampler ---------------------------------------------------------------------------
// grid = (rows, groups).  ps [group][segment][2][32] <- (P, S); stash [group][slot][32] <- clamped sample, negated
// where the clamp does not pass the gradient; mask [group][voxel] <- bit i: image i passes the pre_scale clamp.
template <bool PS, bool SAVE, bool HINT, int NT>
__global__ __launch_bounds__(NT) void bm_sample_kernel(BmDims D, const float *__restrict__ vox,
                                                             const int4 *__restrict__ segs, const int *__restrict__ rec_f,
                                                             const int4 *__restrict__ rows, float *__restrict__ ps,
                                                             float *__restrict__ stash, unsigned *__restrict__ mask,
                                                             const int *__restrict__ tile_live,
                                                             const float4 *__restrict__ ps_empty GENRE_BTL_PARAM)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float *tile = lds_f;                                               // [kLinesF][32]
    int *recs = reinterpret_cast<int *>(lds_f + kLinesF * kImgs);      // [(NT / 64)][64 lanes x 16 bytes]
    const int4 row = rows[blockIdx.x];
    GENRE_BTL(0);
    if (row_flag(row) == 2) return;                                    // padding row of the XCD interleave
    const int g = blockIdx.y, n0 = g * kImgs;
    int ox, oy, oz;
    brick_origin(row, ox, oy, oz);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int NW = NT / 64;
    // EVERYTHING THE ROW ALONE DETERMINES IS REQUESTED AT ONCE (round 6): the occupancy word of the tile, the constants of this
    // wave's first segment (what a dead tile copies) and the headers of its first two segments (what a live tile marches) -- the
    // workgroup's life is a chain of dependent round trips (row -> word -> headers / constants -> records, tile -> march), ~1.2 us
    // each, and a launch is that chain times (workgroups / resident workgroups: two per CU); one link less for either kind of
    // tile.  All of them unconditional (indices clamped into the row; HINT is a template parameter): a load that is only
    // conditionally outstanding would cost the march loop its exact waits (below).
    int s = row.y + wave;
    const bool has_seg = s < row.z;
    const int s_last = has_seg ? s + ((row.z - 1 - s) / NW) * NW : max(min(s, row.z - 1), 0);   // this wave's last segment: indices are clamped to it
    int word = 1;
    float4 pe0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HINT) {
        word = tile_live[(size_t)g * D.nbricks + row.x];
        // (through a VECTOR register: as a scalar load the four words of the constants keep four more SGPRs live across the
        // kernel -- 84 instead of 80 -- and at 81-96 SGPRs gfx950 admits 7 waves per SIMD, not 8: ONE 16-wave workgroup per CU
        // instead of two; measured 134 -> 167 us)
        int pe_idx = has_seg ? s : s_last;
        asm volatile("" : "+v"(pe_idx));
        pe0 = ps_empty[pe_idx];
    }
    int4 sg = segs[has_seg ? s : s_last];
    int4 sg1 = segs[has_seg ? min(s + NW, s_last) : s_last], rq;
    if (HINT) {
        // (the two headers wait in VECTOR registers while the occupancy word is decided: eight scalar registers less across the
        // dead path -- the kernel must stay at <= 80 SGPRs, or gfx950 admits one 16-wave workgroup per CU instead of two)
        asm volatile("" : "+v"(sg.x), "+v"(sg.y), "+v"(sg.z), "+v"(sg.w), "+v"(sg1.x), "+v"(sg1.y), "+v"(sg1.z), "+v"(sg1.w));
    }
    // WHAT THE PRODUCER KNOWS TO BE EMPTY IS NOT READ (round 5).  The volumes this renderer sees are surfaces: a depth map
    // back-projected into 128^3 voxels leaves ~0.5 % of them occupied, and over a group of 32 images 61 % of the tiles (brick +
    // high halo) hold nothing but the producer's fill value.  tile_live [groups][bricks] (written by the camera forward's leader
    // pass, csrc/cam_bp.hip; include/genre_hip.h) says which tiles hold anything else; on the constant tile the (P, S) pair of
    // every segment is a constant of the geometry: ps_empty [segment] = (P, S, the segment's scratch line), the sampler's own
    // output on the constant volume (built once per geometry by the caller: bit-identical to what the march below computes).
    // A dead workgroup copies the constants to its segments' scratch lines, clears its brick's clamp masks and is done: no
    // tile loads (268 MB per group, mostly fill values), no records, no march.
    if (HINT && word == 0) {
        GENRE_BTL(1);
        if (PS && SAVE) {
            for (int e = threadIdx.x; e < kBX * kBY * kBZ; e += NT) {
                const int x = ox + e / (kBY * kBZ), y = oy + (e / kBZ) % kBY, z = oz + e % kBZ;
                if (x < D.X && y < D.Y && z < D.Z) mask[(size_t)g * D.X * D.Y * D.Z + ((size_t)x * D.Y + y) * D.Z + z] = 0u;
            }
        }
        float4 pe = pe0;
        for (int sd = s; sd < row.z; sd += NW) {
            const int nxt = sd + NW;
            const float4 pn = ps_empty[min(nxt, s_last)];               // (the next one is in flight while this one is stored)
            ps[(size_t)(g * D.nseg + __float_as_int(pe.z)) * 2 * kImgs + lane] = lane < kImgs ? pe.x : pe.y;
            pe = pn;
        }
        GENRE_BTL(7);
        return;
    }
    GENRE_BTL(1);
    // THE FIRST SEGMENT'S RECORDS ARE REQUESTED BEFORE THE TILE (round 5).  A wave marches ~1.2 segments of a brick on average
    // (315 k segments over 16 384 bricks x 16 waves), so the software pipeline of the march loop below rarely gets past its
    // prologue; the records first, so that the ONE wait the tile values need also covers them and every later wait stays exact.
    rq = reinterpret_cast<const int4 *>(rec_f + (int64_t)__builtin_amdgcn_readfirstlane(sg.w) * kRec)[lane];   // lane's 16 bytes of the records
    const bool vec = (D.N & 3) == 0 && (D.sx & 3) == 0 && (D.sy & 3) == 0 && (D.sz & 3) == 0;
    // Stage the tile: thread t + 512 j takes 16 bytes (4 images) of voxel line (t + 512 j) / 8.  ALL loads of a thread are
    // issued before the first one is used -- one exposed HBM round trip per tile instead of seven.
    constexpr int kIter = (kLinesF * 8 + NT - 1) / NT;
    float4 q[kIter];
    int xyz[kIter];                                                    // x | y << 10 | z << 20, or -1 outside the volume
#pragma unroll
    for (int j = 0; j < kIter; j++) {
        const int t = threadIdx.x + j * NT;
        const int line = t >> 3, piece = t & 7;
        const int lz = line % kTZ, ly = (line / kTZ) % kTY, lx = line / (kTZ * kTY);
        const int x = ox + lx, y = oy + ly, z = oz + lz;
        q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        xyz[j] = -1;
        if (t < kLinesF * 8 && x < D.X && y < D.Y && z < D.Z) {
            xyz[j] = x | (y << 10) | (z << 20);
            const int n = n0 + piece * 4;
            const float *src = vox + x * D.sx + y * D.sy + z * D.sz + n;
            if (vec && n + 3 < D.N) q[j] = *reinterpret_cast<const float4 *>(src);
            else {
                if (n + 0 < D.N) q[j].x = src[0];
                if (n + 1 < D.N) q[j].y = src[1];
                if (n + 2 < D.N) q[j].z = src[2];
                if (n + 3 < D.N) q[j].w = src[3];
            }
        }
    }
    GENRE_BTL(2);
    unsigned tile_bits = 0u;                                           // OR of every mask bit this thread staged (halo included)
#pragma unroll
    for (int j = 0; j < kIter; j++) {
        const int t = threadIdx.x + j * NT;
        const int line = t >> 3, piece = t & 7;
        float v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
        unsigned bits = 0;
        const bool in = xyz[j] >= 0;
        if (PS && in) {                                                // depth_pred_with_sph_inpaint.py:124
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float raw = v[c] * D.pre_scale;
                v[c] = fminf(fmaxf(raw, D.lo), D.hi);
                bits |= (v[c] == raw) ? (1u << c) : 0u;                // lo <= raw <= hi: the clamp passes the gradient
            }
        }
        if (t < kLinesF * 8) *reinterpret_cast<float4 *>(tile + line * kImgs + piece * 4) = make_float4(v[0], v[1], v[2], v[3]);
        tile_bits |= bits;
        if (PS && SAVE) {
            // the 8 threads of a line sit in 8 consecutive lanes: OR their nibbles together with DPP
            bits <<= piece * 4;
            bits |= __builtin_amdgcn_update_dpp(0u, bits, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
            bits |= __builtin_amdgcn_update_dpp(0u, bits, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
            bits |= __builtin_amdgcn_update_dpp(0u, bits, 0x141, 0xf, 0xf, true);     // row_half_mirror
            const int x = xyz[j] & 1023, y = (xyz[j] >> 10) & 1023, z = xyz[j] >> 20;
            if (piece == 0 && in && x - ox < kBX && y - oy < kBY && z - oz < kBZ) {
                mask[(size_t)g * D.X * D.Y * D.Z + ((size_t)x * D.Y + y) * D.Z + z] = bits;
                // one word per image group behind the masks: "some voxel of this group passes the clamp" (cleared by the
                // host entry in front of this launch; every writer stores the same 1).  The backward of a group whose word
                // is still 0 -- GenRe's own chain: every occupied voxel saturates the x50 clamp, every empty one sits below
                // its lower bound (depth_pred_with_sph_inpaint.py:124) -- is identically zero and skips all of its work.
                if (bits != 0u) mask[(size_t)D.groups * D.X * D.Y * D.Z + g] = 1u;
            }
        }
    }
    // SAVED STATE ONLY WHERE A GRADIENT CAN COME BACK.  With pre_scale the backward multiplies every voxel's sum by its clamp
    // mask; a segment of this brick touches voxels of this tile (brick + high halo) only, so if NO voxel of the staged tile
    // passes the clamp in any image, nothing that is computed from these segments' saved samples survives the backward's
    // flush -- their 4 B per sample and image (the forward's only per-sample stream: 117 us of 277 at batch 32) are not
    // written.  (A neighbouring brick that is live may still pull such a segment and scan whatever the buffer holds: its
    // contributions land on voxels of THIS tile, all of which the mask zeroes -- by a select, so not even a NaN survives.)
    // On GenRe's own volume (x50 of a saturated or empty voxel, depth_pred_with_sph_inpaint.py:124) that is every tile.
    unsigned *wflags = reinterpret_cast<unsigned *>(recs + (NT / 64) * 64 * 4);      // one word per wave
    if (PS && SAVE) {
        const unsigned long long hit = __ballot(tile_bits != 0u);
        if ((threadIdx.x & 63) == 0) wflags[threadIdx.x >> 6] = hit ? 1u : 0u;
    }
    GENRE_BTL(3);
    __syncthreads();
    bool save_rt = SAVE;
    if (PS && SAVE) {
        unsigned any = 0u;
#pragma unroll
        for (int i = 0; i < NT / 64; i += 4) {
            const uint4 w = *reinterpret_cast<const uint4 *>(wflags + i);
            any |= w.x | w.y | w.z | w.w;
        }
        save_rt = __builtin_amdgcn_readfirstlane((int)any) != 0;
    }
    const int half = lane >> 5, l = lane & 31;
    // A wave's records in LDS: the 48-byte table records of the segment, as they come -- every lane parks its 16 bytes
    // (1 KB per wave; the table is padded so that the lanes beyond the segment's L * 3 read real memory), sample i at
    // myrec + 12 i words: header (tile byte offset, w_k) read by all lanes, corner weights per half-wave.  The march
    // below is unrolled over the <= 16 samples of a segment, so all record offsets are compile-time constants.
    int *myrec = recs + wave * (64 * 4);
    const int *myw = myrec + 4 + half * 4;
    const char *tl = reinterpret_cast<const char *>(tile + half * kImgs + l);
    constexpr int kXS = kTY * kTZ * kImgs, kYS = kTZ * kImgs;          // floats between x / y neighbours
    const unsigned lo_st = (unsigned)(half * kImgs + l);
    // Software pipeline over this wave's segments s, s + NT/64, ...: while segment s is marched, the records of the next
    // one are in flight to registers and the header of the one after that is being fetched.  Two rules keep the waits the
    // compiler inserts from undoing it (gfx950 counts loads AND stores in one in-order counter, vmcnt):
    //  * the record load and its use are UNCONDITIONAL (all 64 lanes, every segment; a wave's last segment loads its own
    //    records again): a load behind an execution mask or a branch "may still be pending" on the other path, and the
    //    next load into the same registers then waits for everything in flight;
    //  * STORES ARE DELAYED BY ONE SEGMENT: the wait for the next segment's records would also wait for every store
    //    issued after that load -- the saved samples of the segment just marched, a store round trip per segment and wave.
    //    A segment's saved samples (and its (P, S) pair) stay in registers and leave at the top of the NEXT segment, after
    //    the wait, so that at every wait the only operations outstanding were issued a whole march earlier.
    GENRE_BTL(4);
    if (!has_seg) return;                                              // (header and records of the first segment: requested at the top)
    // the march, compiled twice where SAVE: with and without the stores of the saved samples (the choice is per workgroup and
    // made once, outside the loop: every wait inside stays exact)
    auto march = [&](auto save_c) {
    constexpr bool SV = decltype(save_c)::value;
    float sp[kMaxSeg / 2];                                             // this half-wave's saved sample of pair j
    float ps_val = 0.f;
    int Lprev = 0;
    float *stp_prev = nullptr, *ps_prev = nullptr;
    auto flush_prev = [&]() {                                          // the previous segment's stores
        if (Lprev == 0) return;
        if (SV) {
#pragma unroll
            for (int j = 0; j < kMaxSeg / 2; j++)      // samples 2j (lower half-wave) and 2j + 1 (upper): one 256-byte store
                if (2 * j < Lprev && (2 * j + 1 < Lprev || half == 0)) (stp_prev + (2 * j) * kImgs)[lo_st] = sp[j];
        }
        ps_prev[lo_st] = ps_val;
    };
    for (; s < row.z; s += NW) {
        const int L = __builtin_amdgcn_readfirstlane(sg.z);
        const int64_t slot0 = __builtin_amdgcn_readfirstlane(sg.w);
        reinterpret_cast<int4 *>(myrec)[lane] = rq;
        wave_lds_fence();
        flush_prev();
        const int4 sg2 = segs[min(s + 2 * NW, s_last)];
        rq = reinterpret_cast<const int4 *>(rec_f + (int64_t)__builtin_amdgcn_readfirstlane(sg1.w) * kRec)[lane];
        float T = 1.f, S = 0.f;
        auto one = [&](const int i) {                                  // sample i: returns the saved (sign-coded) value
            const int2 h = *reinterpret_cast<const int2 *>(myrec + i * kRec);        // (tile byte offset, w_k)
            const float4 w = *reinterpret_cast<const float4 *>(myw + i * kRec);
            const float *a = reinterpret_cast<const float *>(tl + h.x);
            float v = a[0] * w.x;
            v = __builtin_fmaf(a[kXS], w.y, v);
            v = __builtin_fmaf(a[kYS], w.z, v);
            v = __builtin_fmaf(a[kXS + kYS], w.w, v);
            v = other_half_sum(v);
            const float p = __builtin_amdgcn_fmed3f(v, D.lo, D.hi);                   // clamp (spherical_proj.py:66)
            S = __builtin_fmaf(T * p, __int_as_float(h.y), S);                        // + s_k w_k  (:68)
            T *= 1.0f - p;
            return (p == v) ? p : -p;                                  // negated where the clamp does not pass the gradient
        };
#pragma unroll
        for (int j = 0; j < kMaxSeg / 2; j++) {
            if (2 * j < L) {
                const float spa = one(2 * j);
                float spb = 0.f;
                if (2 * j + 1 < L) spb = one(2 * j + 1);
                sp[j] = half ? spb : spa;
            }
        }
        ps_val = half ? S : T;
        Lprev = L;
        stp_prev = SV ? stash + ((size_t)g * D.nslot + slot0) * kImgs : nullptr;
        ps_prev = ps + (size_t)(g * D.nseg + __builtin_amdgcn_readfirstlane(sg.x)) * 2 * kImgs;   // its line: ray order
        wave_lds_fence();                                                             // records are overwritten next
        sg = sg1; sg1 = sg2;
    }
    flush_prev();
    };
    if (SAVE && save_rt) march(std::true_type{});
    else march(std::false_type{});
    GENRE_BTL(6);
}

// sph_pad (spherical_proj.py:21-28) as a fan-out of map pixel (i, j): see sph_render.hip: pad_span
__device__ __forceinline__ void pad_span(int R, int pm, int i, int j, int &r_lo, int &r_n, int &c0, int &c1)
{
    r_lo = (i == 0) ? 0 : i + pm;
    r_n = ((i == R - 1) ? R - 1 + 2 * pm : i + pm) - r_lo + 1;
    c0 = j + pm;
    c1 = (j >= R - pm) ? j - (R - pm) : (j < pm ? j + R + pm : -1);
}

// ---- forward: chain the segments of a ray --------------------------------------------------------------
// a half-wave per ray, lane = image; fp64 (18 segments per ray on average: nothing to save here)
__global__ __launch_bounds__(256) void bm_combine_fwd_kernel(BmDims D, const float *__restrict__ ps,
                                                             const int *__restrict__ ray_ptr,
                                                             const int *__restrict__ ray_seg,
                                                             const double2 *__restrict__ ray_pre, View4 out)
{
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (q >= D.R * D.R) return;
    const int g = blockIdx.y, n = g * kImgs + l;
    const int j0 = ray_ptr[q], j1 = ray_ptr[q + 1];
    const double2 pre = ray_pre[q];
    double T = pre.x, S = pre.y;
    const float *base = ps + (size_t)g * D.nseg * 2 * kImgs + l;
    // the ray's segments are neighbours in the scratch buffer (line = ray-order position): contiguous 256-byte records,
    // eight in flight together
    for (int jb = j0; jb < j1; jb += 8) {
        float P[8], Sg[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = min(jb + u, j1 - 1);
            P[u] = base[(size_t)j * 2 * kImgs];
            Sg[u] = base[(size_t)j * 2 * kImgs + kImgs];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (jb + u < j1) {
                S += T * (double)Sg[u];
                T *= (double)P[u];
            }
        }
    }
    if (n >= D.N) return;
    const float val = (float)(S + T);                                   // + prod(1-p)  (:69-71)
    float *o = out.p + n * out.s0;
    const int i = q / D.R, jj = q % D.R;
    if (D.pad == 0) { o[i * out.s2 + jj * out.s3] = val; return; }
    int r_lo, r_n, c0, c1;
    pad_span(D.R, D.pad, i, jj, r_lo, r_n, c0, c1);
    for (int r = 0; r < r_n; r++) {
        o[(r_lo + r) * out.s2 + c0 * out.s3] = val;
        if (c1 >= 0) o[(r_lo + r) * out.s2 + c1 * out.s3] = val;
    }
}

// ---- backward: per-ray pass ------------------------------------------------------------------------------
// tr [group][segment][2][32] <- (g * T at the segment's start, R behind its end)
__global__ __launch_bounds__(256, 4) void bm_combine_bwd_kernel(BmDims D, const float *__restrict__ ps,
                                                             const int *__restrict__ ray_ptr,
                                                             const int *__restrict__ ray_seg,
                                                             const double2 *__restrict__ ray_pre, View4 gout,
                                                             float *__restrict__ tr, const unsigned *__restrict__ group_any)
{
    if (group_any && group_any[blockIdx.y] == 0u) return;              // no voxel of this group passes the clamp: nothing reads tr
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (q >= D.R * D.R) return;
    const int g = blockIdx.y, n = g * kImgs + l;
    float gv = 0.f;
    if (n < D.N) {                                                       // gradient of the ray's value: its padded positions
        const float *gi = gout.p + n * gout.s0;
        const int i = q / D.R, jj = q % D.R;
        if (D.pad == 0) gv = gi[i * gout.s2 + jj * gout.s3];
        else {
            int r_lo, r_n, c0, c1;
            pad_span(D.R, D.pad, i, jj, r_lo, r_n, c0, c1);
            for (int r = 0; r < r_n; r++) {
                gv += gi[(r_lo + r) * gout.s2 + c0 * gout.s3];
                if (c1 >= 0) gv += gi[(r_lo + r) * gout.s2 + c1 * gout.s3];
            }
        }
    }
    const int j0 = ray_ptr[q], j1 = ray_ptr[q + 1];
    const size_t gb = (size_t)g * D.nseg * 2 * kImgs + l;
    if (j1 - j0 <= kRayRegs && (int64_t)D.nseg * (2 * kImgs * 4) < ((int64_t)1 << 31)) {
        // The usual case (<= 32 segments per ray; 19 on average at 128^3): ALL of the ray's (P, S) lines are requested at
        // once -- the kernel is three dependent round trips (ray_ptr -> segment ids -> lines) instead of ten -- and both
        // chains run from registers: g T at every segment's start (prefix products), R behind its end (suffix recursion).
        // Buffer addressing (wave-uniform descriptor of this group's slice + a 32-bit byte offset per line): one
        // register per line address instead of two.
        const int cnt = j1 - j0;
        const unsigned bytes = (unsigned)D.nseg * (2 * kImgs * 4);
        const auto r_ps = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ps) + (size_t)g * D.nseg * 2 * kImgs, 0, bytes, 0x00020000);
        const auto r_tr = __builtin_amdgcn_make_buffer_rsrc(tr + (size_t)g * D.nseg * 2 * kImgs, 0, bytes, 0x00020000);
        int o[kRayRegs];
        float P[kRayRegs], Sg[kRayRegs];
#pragma unroll
        for (int u = 0; u < kRayRegs; u++) {
            o[u] = ((j0 + (u < cnt ? u : 0)) * (2 * kImgs) + l) * 4;      // (an index beyond the ray re-reads its first line)
            P[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_ps, o[u], 0, 0));
            Sg[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_ps, o[u] + kImgs * 4, 0, 0));
        }
        double T = ray_pre[q].x;
#pragma unroll
        for (int u = 0; u < kRayRegs; u++) {
            if (u < cnt) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)((double)gv * T)), r_tr, o[u], 0, 0);
                T *= (double)P[u];
            }
        }
        double Rr = 1.0;                                                  // behind the last sample: prod(1-p) * 1
#pragma unroll
        for (int u = kRayRegs - 1; u >= 0; u--) {
            if (u < cnt) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)Rr), r_tr, o[u] + kImgs * 4, 0, 0);
                Rr = (double)Sg[u] + (double)P[u] * Rr;
            }
        }
        return;
    }
    double T = ray_pre[q].x;
    for (int jb = j0; jb < j1; jb += 32) {                               // forward: g T at every segment's start
        const int cnt = (j1 - jb < 32) ? j1 - jb : 32;
        for (int t = 0; t < cnt; t += 4) {
            float P[4];
            size_t o[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                o[u] = gb + (size_t)(jb + min(t + u, cnt - 1)) * 2 * kImgs;
                P[u] = t + u < cnt ? ps[o[u]] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (t + u < cnt) tr[o[u]] = (float)((double)gv * T);
                T *= (double)P[u];
            }
        }
    }
    double Rr = 1.0;                                                      // behind the last sample: prod(1-p) * 1
    for (int je = j1; je > j0; je -= 32) {                                // reverse: R behind every segment's end
        const int cnt = (je - j0 < 32) ? je - j0 : 32;
        for (int t = 0; t < cnt; t += 4) {
            float P[4], Sg[4];
            size_t o[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                o[u] = gb + (size_t)(je - 1 - min(t + u, cnt - 1)) * 2 * kImgs;
                const bool on = t + u < cnt;
                P[u] = on ? ps[o[u]] : 0.f;
                Sg[u] = on ? ps[o[u] + kImgs] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (t + u < cnt) {
                    tr[o[u] + kImgs] = (float)Rr;
                    Rr = (double)Sg[u] + (double)P[u] * Rr;
                }
            }
        }
    }
}

// ---- backward: brick-owned pull scatter ----------------------------------------------------------------
// The backward's bricks ("pull bricks", PX x PY x PZ voxels) need not be the forward's: a bigger one lowers the number of
// bricks a segment touches (every touching brick re-reads the segment's saved samples) at the price of LDS.
template <int PX, int PY, int PZ>
__global__ __launch_bounds__(kThreads) void bm_zero_shared_kernel(BmDims D, const int4 *__restrict__ rows, int nrows,
                                                                 float *__restrict__ gvox)
{
    // the rows of split bricks are the head of the table (toolbox/_bm_tables.py: _split_rows): a workgroup stops at the first row that
    // is not one.  (Launched over all 16 k rows to find ~200, this kernel took 14 us of the step's zero backward.)
    for (int i = blockIdx.x; i < nrows; i += gridDim.x) {
        const int4 row = rows[i];
        if (row_flag(row) != 1) break;
        int ox, oy, oz;
        brick_origin<PX, PY, PZ>(row, ox, oy, oz);
        const int n0 = blockIdx.y * kImgs;
        for (int e = threadIdx.x; e < PX * PY * PZ * kImgs; e += kThreads) {
            const int line = e >> 5, n = n0 + (e & 31);
            const int x = ox + line / (PY * PZ), y = oy + (line / PZ) % PY, z = oz + line % PZ;
            if (x < D.X && y < D.Y && z < D.Z && n < D.N) gvox[x * D.gx + y * D.gy + z * D.gz + n] = 0.f;
        }
    }
}

// swap the two half-waves' values: returns (value held by the lower half, value held by the upper half) in every lane
__device__ __forceinline__ void both_halves(float v, float &lower, float &upper)
{
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lower = __uint_as_float(r[0]); upper = __uint_as_float(r[1]);
}

constexpr int kHalfSeg = kMaxSeg / 2;
#ifndef GENRE_BM_SCATTER_NT
#define GENRE_BM_SCATTER_NT 768                // threads of the 4x8x8 pull-brick scatter kernel (tools/ab_round4.py: 1024)
#endif
#ifndef GENRE_BM_SCATTER_WAVES
#define GENRE_BM_SCATTER_WAVES 8               // waves per SIMD the 1024-thread variant is compiled for (<= 64 VGPRs)
#endif

// A lane-uniform ownership bit pair (lower half-wave, upper half-wave) as an execution mask: two s_bfe_i32 and the
// s_and_saveexec of the `if` -- no vector instruction and no branch, so the compiler counts the LDS operations in flight
// exactly and the next sample's record read can stay queued behind this sample's atomics.
__device__ __forceinline__ bool owned(unsigned own, int c)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_sbfe((int)own, c, 1);
    const unsigned hi = (unsigned)__builtin_amdgcn_sbfe((int)own, c + 4, 1);
    return __builtin_amdgcn_inverse_ballot_w64((unsigned long long)lo | ((unsigned long long)hi << 32));
}

// What a wave holds of one entry before it works on it: the saved samples (split over the half-waves: the lower half keeps
// samples 0, 2, 4, ..., the upper half 1, 3, 5, ...), the two ray scalars of the segment, the depth weight of sample
// `lane` (lanes 0-15) and the lane's 16 bytes of the entry's records.
struct BmEntryRegs {
    float p[kHalfSeg];
    float T, R, wk;
    int4 rq;
};

// 768 threads: two workgroups per CU = 6 waves per SIMD, which the register allocation must respect (<= 80 VGPRs)
template <bool PS, int PX, int PY, int PZ, int kThreadsB>
__global__ __launch_bounds__(kThreadsB, (PX == 4 ? (kThreadsB <= 768 ? 6 : GENRE_BM_SCATTER_WAVES) : 4)) void bm_scatter_kernel(BmDims D, const int4 *__restrict__ ents, const int *__restrict__ rec_b,
                                                               const int4 *__restrict__ rows, const float *__restrict__ dw,
                                                               const float *__restrict__ tr, const float *__restrict__ stash,
                                                               const unsigned *__restrict__ mask, float *__restrict__ gvox)
{
    constexpr int QX = PX, QY = PY, QZ = PZ;                            // tile dimensions in lines: the brick, no halo
    constexpr int kLinesB = QX * QY * QZ, kWavesB = kThreadsB / 64;
    constexpr int kMaskThreads = (kLinesB + 255) / 256 * 256;           // threads that carry a mask word (whole waves)
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    double *tile = lds_d;                                               // [kLinesB][32]
    int *recs = reinterpret_cast<int *>(lds_d + kLinesB * kImgs);       // [kWavesB][kMaxSeg * kRecL]
    unsigned *mlds = reinterpret_cast<unsigned *>(recs + kWavesB * kMaxSeg * kRecL);    // [kMaskThreads] clamp masks + summary words
    // (the group word is requested beside the row, not behind it: on GenRe's own chain every group is dead, the launch is then
    // 16 k workgroups that each wait for their first loads, write 32 KB of zeros and leave -- two per CU (LDS): the launch is
    // bound by that turnover, i.e. by the dependent round trips in front of the stores)
    const int g = blockIdx.y, n0 = g * kImgs;
    const unsigned group_word = PS ? mask[(size_t)D.groups * D.X * D.Y * D.Z + g] : 1u;
    const int4 row = rows[blockIdx.x];
    if (row_flag(row) == 2) return;                                     // padding row of the XCD interleave
    const bool shared_row = row_flag(row) == 1;
    int ox, oy, oz;
    brick_origin<PX, PY, PZ>(row, ox, oy, oz);
    if (PS) {
        // The clamp adjoint decides first.  A brick none of whose voxels passes clamp(x * pre_scale, lo, hi) in any of the 32
        // images has an identically zero gradient whatever the samples say (the flush multiplies by the mask): it writes
        // its zeros and is done -- no entries, no scans, no atomics.  On GenRe's own chain (x50 of a saturated or empty
        // voxel) that is every brick; the group word behind the masks says so without reading them.
        static_assert(kMaskThreads <= kThreadsB, "one mask word per thread");
        auto write_zeros = [&]() {                                       // this brick's voxels of all 32 images <- 0 (streaming stores)
            const bool v4 = (D.N & 3) == 0 && (D.gx & 3) == 0 && (D.gy & 3) == 0 && (D.gz & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(gvox) & 15) == 0;
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            for (int q = threadIdx.x; q < PX * PY * PZ * (kImgs / 4); q += kThreadsB) {
                const int line = q >> 3, n = n0 + (q & 7) * 4;
                const int x = ox + line / (PY * PZ), y = oy + (line / PZ) % PY, z = oz + line % PZ;
                if (x >= D.X || y >= D.Y || z >= D.Z) continue;
                float *dst = gvox + x * D.gx + y * D.gy + z * D.gz + n;
                if (v4 && n + 3 < D.N) __builtin_nontemporal_store((v4f_){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f_ *>(dst));
                else
                    for (int c = 0; c < 4; c++) if (n + c < D.N) dst[c] = 0.f;
            }
        };
        const bool group_live = group_word != 0u;
        if (!group_live) {          // nothing of this group passes the clamp: the zeros at once -- no tile to clear, no masks, no barrier
            if (!shared_row) write_zeros();                             // (a shared brick: bm_zero_shared_kernel wrote its zeros)
            return;
        }
        unsigned m = 0u;
        if ((int)threadIdx.x < kLinesB) {
            const int line = threadIdx.x;
            const int x = ox + line / (QY * QZ), y = oy + (line / QZ) % QY, z = oz + line % QZ;
            if (x < D.X && y < D.Y && z < D.Z) m = mask[(size_t)g * D.X * D.Y * D.Z + ((size_t)x * D.Y + y) * D.Z + z];
        }
        for (int e = threadIdx.x; e < kLinesB * kImgs / 2; e += kThreadsB) reinterpret_cast<double2 *>(tile)[e] = make_double2(0.0, 0.0);
        if ((int)threadIdx.x < kMaskThreads) {
            mlds[threadIdx.x] = m;
            const unsigned long long hit = __ballot(m != 0u);          // the waves that carry mask words: one summary word each
            if ((threadIdx.x & 63) == 0) mlds[kMaskThreads + (threadIdx.x >> 6)] = hit ? 1u : 0u;
        }
        __syncthreads();
        unsigned brick_live = 0u;
#pragma unroll
        for (int i = 0; i < kMaskThreads / 64; i += 4) {
            const uint4 w = *reinterpret_cast<const uint4 *>(mlds + kMaskThreads + i);
            brick_live |= w.x | w.y | w.z | w.w;
        }
        if (!brick_live) {
            if (!shared_row) write_zeros();                             // (a shared brick: bm_zero_shared_kernel wrote its zeros)
            return;
        }
    } else {
        for (int e = threadIdx.x; e < kLinesB * kImgs / 2; e += kThreadsB) reinterpret_cast<double2 *>(tile)[e] = make_double2(0.0, 0.0);
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = lane >> 5, l = lane & 31;
    // A wave's records in LDS: slot = sample index inside the segment, 64 bytes per slot = one 32-byte part per half-wave,
    //   (tile byte offset, ownership bits, depth weight w_k, -)  (this half's four corner weights)
    // so that a lane reads its sample with two 16-byte reads at compile-time offsets from ONE per-lane base (no address
    // arithmetic per sample).  The 48-byte table record (header, weights z0, weights z0+1) is spread out when it is staged:
    // the lanes holding a header write it into both parts.
    int *myrec = recs + wave * (kMaxSeg * kRecL);
    const int *myh = myrec + half * 8;
    const int stage_rec = lane / 3, stage_part = lane - stage_rec * 3;   // staging: lane -> (record, 16-byte chunk)
    int *stage_to = myrec + stage_rec * kRecL + (stage_part == 0 ? 0 : stage_part == 1 ? 4 : 12);
    char *tl = reinterpret_cast<char *>(tile + half * kImgs + l);
    constexpr int kXS = QY * QZ * kImgs, kYS = QZ * kImgs;           // doubles between x / y neighbours
    // entry = (segment, stash slot of its first sample, i0 | i1 << 6 | L << 12 | k0 << 18, rec_b slot of sample i0).
    // Software pipeline: while entry e is scattered, the saved samples, the two ray scalars, the depth weights and the
    // records of entry e + kWavesB are in flight to registers (two register sets, used alternately: no copies) and the
    // header of e + 2 kWavesB is being fetched and decoded.
    // Global addresses = wave-uniform 64-bit base (scalar registers) + a 32-bit per-lane offset: three persistent
    // vector registers instead of three address pairs.
    const unsigned lo_st = (unsigned)(half * kImgs + l), lo_tr = (unsigned)l, lo_rq = (unsigned)lane * 4u;
    const float *stash_g = stash + (size_t)g * D.nslot * kImgs;
    const float *tr_g = tr + (size_t)g * D.nseg * 2 * kImgs;
    // A decoded header: everything the prefetch of an entry needs, wave-uniform, on the scalar unit.  (Passed BY VALUE:
    // a captured reference makes the compiler keep it in vector registers.)
    struct Hdr { const float *st, *tp; const int *rq; int pk; };
    auto decode = [&](const int4 h) {
        Hdr r;
        r.st = stash_g + (size_t)h.y * kImgs;
        r.tp = tr_g + (size_t)h.x * 2 * kImgs;
        r.rq = rec_b + (size_t)h.w * kRec;
        r.pk = h.z;
        return r;
    };
    // EVERY fetch issues the same twelve vector loads, unconditionally: the compiler can then wait for "all but the
    // twelve youngest" when the current entry's registers are first used -- with loads behind branches or execution masks
    // it has to assume the fewest, waits for (almost) everything, and the prefetch it was meant to overlap is stalled on at
    // once.  Samples and records read past the entry's own (the slot and record spaces are padded) are never used.
    auto fetch = [&](const Hdr h, BmEntryRegs &o) {
#pragma unroll
        for (int j = 0; j < kHalfSeg; j++) o.p[j] = (h.st + (2 * j) * kImgs)[lo_st];
        o.T = h.tp[lo_tr]; o.R = (h.tp + kImgs)[lo_tr];
        o.rq = *reinterpret_cast<const int4 *>(h.rq + lo_rq);
        o.wk = dw[min(((h.pk >> 18) & 255) + (lane & (kMaxSeg - 1)), D.ZR - 1)];     // depth weight of sample `lane` (lanes 0-15)
    };
    // One entry.  `h2raw` is the raw header of the entry after next, requested by the caller just before: it is decoded
    // between the two scans, so that the scalar load is retired BEFORE the reverse loop -- a scalar load still in flight
    // would force every wait of that loop down to "everything", atomics included (scalar loads return out of order).
    auto entry = [&](const int pk, const Hdr h1, const bool has_next2, const int4 h2raw, Hdr &h2,
                     BmEntryRegs &cur, BmEntryRegs &nxt) {
        const int i0 = pk & 63, i1 = (pk >> 6) & 63, L = (pk >> 12) & 63;
        if (lane < (i1 - i0) * 3) {
            int4 *to = reinterpret_cast<int4 *>(stage_to + i0 * kRecL);
            const int4 rv = cur.rq;
            *to = rv;
            if (stage_part == 0) to[2] = rv;                            // the header, again, for the upper half-wave
        }
        wave_lds_fence();                                               // (headers first: they carry zeros where w_k goes)
        if (lane < kMaxSeg) {
            myrec[lane * kRecL + 2] = __float_as_int(cur.wk);
            myrec[lane * kRecL + 10] = __float_as_int(cur.wk);
        }
        wave_lds_fence();
        fetch(h1, nxt);                                                 // (the last entry: itself again, see the caller)
        float Tg = cur.T, Rr = cur.R;
        float ce[kHalfSeg], co[kHalfSeg];                               // g T at samples 2j and 2j + 1
#pragma unroll
        for (int j = 0; j < kHalfSeg; j++) {                            // forward: g T_k
            if (2 * j < L) {
                float pe, po;
                both_halves(cur.p[j], pe, po);                          // samples 2j, 2j + 1 (po unused beyond the end)
                ce[j] = Tg;
                Tg *= 1.0f - fabsf(pe);
                co[j] = Tg;
                Tg *= 1.0f - fabsf(po);
            }
        }
        if (has_next2) h2 = decode(h2raw);
        // Records: two register sets used by sample parity (compile-time after unrolling: no copies).  That of the sample
        // the reverse loop meets first is the only one read with a run-time offset (into both sets).
        int4 hd0 = *reinterpret_cast<const int4 *>(myh + (L - 1) * kRecL), hd1 = hd0;    // (tile byte offset, ownership, w_k, -)
        float4 w0 = *reinterpret_cast<const float4 *>(myh + (L - 1) * kRecL + 4), w1 = w0;
        auto sample = [&](const int i, float ps, float c) {            // reverse: R_k = p w + (1-p) R_{k+1}; dL/dp_k; scatter
            const int4 hc = (i & 1) ? hd1 : hd0;
            const float4 wc = (i & 1) ? w1 : w0;
            const float d = __int_as_float(hc.z) - Rr;
            Rr = __builtin_fmaf(fabsf(ps), d, Rr);
            if (i > 0) {                                                // the next sample's record: queued before this one's atomics
                const int4 hn = *reinterpret_cast<const int4 *>(myh + (i - 1) * kRecL);
                const float4 wn = *reinterpret_cast<const float4 *>(myh + (i - 1) * kRecL + 4);
                if (i & 1) { hd0 = hn; w0 = wn; } else { hd1 = hn; w1 = wn; }
            }
            if (i < i1) {
                const float dp = ps > 0.f ? c * d : 0.f;                // the clamp passes the gradient where the saved sample is > 0
                double *a = reinterpret_cast<double *>(tl + hc.x);
                const unsigned own = (unsigned)__builtin_amdgcn_readfirstlane(hc.y);
                if (owned(own, 0)) unsafeAtomicAdd(a, (double)(wc.x * dp));                    // ds_add_f64
                if (owned(own, 1)) unsafeAtomicAdd(a + kXS, (double)(wc.y * dp));
                if (owned(own, 2)) unsafeAtomicAdd(a + kYS, (double)(wc.z * dp));
                if (owned(own, 3)) unsafeAtomicAdd(a + kXS + kYS, (double)(wc.w * dp));
            }
        };
#pragma unroll
        for (int j = kHalfSeg - 1; j >= 0; j--) {
            if (2 * j < L && 2 * j + 1 >= i0) {
                float pe, po;
                both_halves(cur.p[j], pe, po);
                if (2 * j + 1 < L) sample(2 * j + 1, po, co[j]);
                if (2 * j >= i0) sample(2 * j, pe, ce[j]);
            }
        }
        wave_lds_fence();
    };
    int e = row.y + wave;
    int pk0 = 0;
    Hdr h1{nullptr, nullptr, nullptr, 0}, h2{nullptr, nullptr, nullptr, 0};
    BmEntryRegs A, B;
    A.rq = make_int4(0, 0, 0, 0); B.rq = make_int4(0, 0, 0, 0);
    if (e < row.z) {
        const Hdr h = decode(ents[e]);
        pk0 = h.pk;
        fetch(h, A);
        h1 = h;                                                         // a wave's last entry prefetches a valid entry: itself
    }
    if (e + kWavesB < row.z) h1 = decode(ents[e + kWavesB]);
#define GENRE_BM_STEP(CUR, NXT)                                                                                         \
    {                                                                                                                   \
        const bool n2_ = e + 2 * kWavesB < row.z;                                                                        \
        int4 raw_ = make_int4(0, 0, 0, 0);                                                                               \
        if (n2_) raw_ = ents[e + 2 * kWavesB];                                                                           \
        entry(__builtin_amdgcn_readfirstlane(pk0), h1, n2_, raw_, h2, CUR, NXT);                                         \
        pk0 = h1.pk; if (n2_) h1 = h2;                                                                                   \
        e += kWavesB;                                                                                                   \
    }
    while (e < row.z) {
        GENRE_BM_STEP(A, B)
        if (e >= row.z) break;
        GENRE_BM_STEP(B, A)
    }
#undef GENRE_BM_STEP
    __syncthreads();
    // Flush: every voxel of the brick exactly once.  Four images per thread (two 16-byte LDS reads, one 16-byte store) when
    // the layout allows; rows that share their brick with other rows add atomically onto pre-zeroed voxels.
    const bool vec4 = !shared_row && (D.N & 3) == 0 && (D.gx & 3) == 0 && (D.gy & 3) == 0 && (D.gz & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(gvox) & 15) == 0;
    if (vec4) {
        for (int q = threadIdx.x; q < kLinesB * (kImgs / 4); q += kThreadsB) {
            const int line = q >> 3, piece = (q & 7) * 4, n = n0 + piece;
            const int x = ox + line / (PY * PZ), y = oy + (line / PZ) % PY, z = oz + line % PZ;
            if (x < D.X && y < D.Y && z < D.Z && n < D.N) {
                const double2 t01 = *reinterpret_cast<const double2 *>(tile + line * kImgs + piece);
                const double2 t23 = *reinterpret_cast<const double2 *>(tile + line * kImgs + piece + 2);
                float4 v = make_float4((float)t01.x, (float)t01.y, (float)t23.x, (float)t23.y);
                if (PS) {                                               // adjoint of clamp(x * pre_scale, lo, hi)
                    const unsigned m = mlds[line] >> piece;
                    v.x = (m & 1u) ? v.x * D.pre_scale : 0.f;
                    v.y = (m & 2u) ? v.y * D.pre_scale : 0.f;
                    v.z = (m & 4u) ? v.z * D.pre_scale : 0.f;
                    v.w = (m & 8u) ? v.w * D.pre_scale : 0.f;
                }
                *reinterpret_cast<float4 *>(gvox + x * D.gx + y * D.gy + z * D.gz + n) = v;
            }
        }
        return;
    }
#pragma unroll 4
    for (int e2 = threadIdx.x; e2 < kLinesB * kImgs; e2 += kThreadsB) {
        const int line = e2 >> 5, n = n0 + (e2 & 31);
        const int x = ox + line / (PY * PZ), y = oy + (line / PZ) % PY, z = oz + line % PZ;
        if (x < D.X && y < D.Y && z < D.Z && n < D.N) {
            float val = (float)tile[e2];
            if (PS) val = ((mlds[line] >> (e2 & 31)) & 1u) ? val * D.pre_scale : 0.f;  // adjoint of clamp(x * pre_scale, lo, hi)
            float *dst = gvox + x * D.gx + y * D.gy + z * D.gz + n;
            if (!shared_row) *dst = val;
            else if (val != 0.f) unsafeAtomicAdd(dst, val);
        }
    }
}

int check_bm(const char *op, const genre_tensor *vox, const genre_tensor *map, const genre_tensor *segs,
             const genre_tensor *ray_ptr, const genre_tensor *ray_seg, const genre_tensor *ray_pre, BmDims &D, bool grad)
{
    GENRE_REQUIRE(is_f32(vox, 5) && vox->size[1] == 1 && vox->stride[0] == 1, "%s: %s must be a batch-minor fp32 tensor "
                  "[N,1,X,Y,Z] (stride[0] == 1)", op, grad ? "grad_vox" : "vox");
    D.N = (int)vox->size[0]; D.X = (int)vox->size[2]; D.Y = (int)vox->size[3]; D.Z = (int)vox->size[4];
    int64_t span = 1;
    for (int i = 0; i < 5; i++) {
        GENRE_REQUIRE(vox->stride[i] >= 0, "%s: negative strides are not supported", op);
        span += (vox->size[i] - 1) * vox->stride[i];
    }
    GENRE_REQUIRE(span < ((int64_t)1 << 40) && numel(vox) > 0, "%s: empty or oversized volume", op);
    GENRE_REQUIRE(is_i32(ray_ptr, 1) && is_contiguous(ray_ptr) && ray_ptr->size[0] >= 2, "%s: ray_ptr must be int32 [R*R+1]", op);
    const int64_t rr = ray_ptr->size[0] - 1;
    int R = 1;
    while ((int64_t)R * R < rr) R++;
    GENRE_REQUIRE((int64_t)R * R == rr, "%s: ray_ptr must have R*R+1 entries", op);
    D.R = R;
    GENRE_REQUIRE(is_f32(map, 4) && map->size[0] == vox->size[0] && map->size[1] == 1 && map->size[2] == map->size[3] &&
                      map->size[2] >= R && ((map->size[2] - R) & 1) == 0 && (map->size[2] - R) <= R,
                  "%s: the spherical map must be fp32 [N,1,R+2p,R+2p] with 0 <= 2p <= R = %d", op, R);
    D.pad = (int)(map->size[2] - R) / 2;
    GENRE_REQUIRE(is_i32(segs, 2) && segs->size[1] == 4 && is_contiguous(segs) && aligned16(segs->data) &&
                      segs->size[0] < ((int64_t)1 << 26), "%s: segs must be a contiguous int32 [nseg,4] tensor", op);
    D.nseg = (int)segs->size[0];
    GENRE_REQUIRE(is_i32(ray_seg, 1) && is_contiguous(ray_seg) && ray_seg->size[0] == D.nseg, "%s: ray_seg must be int32 [nseg]", op);
    GENRE_REQUIRE(is_f32(ray_pre, 2) && ray_pre->size[0] == rr && ray_pre->size[1] == 4 && is_contiguous(ray_pre) &&
                      aligned16(ray_pre->data), "%s: ray_pre must be the float64 [R*R,2] table viewed as fp32 [R*R,4]", op);
    D.groups = (D.N + kImgs - 1) / kImgs;
    GENRE_REQUIRE(D.groups <= 65535 && (int64_t)D.groups * D.nseg < ((int64_t)1 << 31), "%s: batch too large", op);
    D.lo = 1e-5f; D.hi = (float)(1 - 1e-5);                              // spherical_proj.py:66
    return 1;
}

int check_rows(const char *op, const BmDims &D, const genre_tensor *rows, int bx = kBX, int by = kBY, int bz = kBZ)
{
    const int nb = ((D.X + bx - 1) / bx) * ((D.Y + by - 1) / by) * ((D.Z + bz - 1) / bz);
    GENRE_REQUIRE(is_i32(rows, 2) && rows->size[1] == 4 && is_contiguous(rows) && aligned16(rows->data) &&
                      rows->size[0] >= nb && rows->size[0] < ((int64_t)1 << 30),
                  "%s: row table must be a contiguous int32 [rows >= %d, 4] tensor", op, nb);
    return 1;
}

}  // namespace
}  // namespace genre

using namespace genre;

extern "C" int genre_bm_brick(void) { return kBX * 100 + kBY * 10 + kBZ; }

extern "C" int genre_render_bm_forward(const genre_tensor *vox, const genre_tensor *out, const genre_tensor *segs,
                                       const genre_tensor *rec_f, const genre_tensor *fwd_rows,
                                       const genre_tensor *ray_ptr, const genre_tensor *ray_seg,
                                       const genre_tensor *ray_pre, const genre_tensor *ps_scratch,
                                       const genre_tensor *p_stash, const genre_tensor *mask,
                                       const genre_tensor *tile_live, const genre_tensor *ps_empty, float pre_scale,
                                       void *stream)
{
    const char *op = "render_bm_forward";
    BmDims D{};
    if (!check_bm(op, vox, out, segs, ray_ptr, ray_seg, ray_pre, D, false)) return 0;
    if (!check_rows(op, D, fwd_rows)) return 0;
    D.sx = vox->stride[2]; D.sy = vox->stride[3]; D.sz = vox->stride[4];
    D.pre_scale = pre_scale;
    GENRE_REQUIRE(is_i32(rec_f, 2) && rec_f->size[1] == kRec && is_contiguous(rec_f) && aligned16(rec_f->data),
                  "%s: rec_f must be a contiguous int32 [S,12] tensor", op);
    D.nslot = rec_f->size[0];
    GENRE_REQUIRE(is_f32(ps_scratch, 1) && is_contiguous(ps_scratch) && ps_scratch->size[0] >= (int64_t)D.groups * D.nseg * 2 * kImgs,
                  "%s: ps_scratch must hold groups*nseg*64 floats", op);
    const bool save = p_stash != nullptr;
    if (save) {
        GENRE_REQUIRE(is_f32(p_stash, 1) && is_contiguous(p_stash) && p_stash->size[0] >= (int64_t)D.groups * D.nslot * kImgs,
                      "%s: p_stash must hold groups*S*32 floats", op);
        GENRE_REQUIRE(pre_scale == 0.0f || (is_i32(mask, 1) && is_contiguous(mask) &&
                                            mask->size[0] >= (int64_t)D.groups * D.X * D.Y * D.Z + D.groups),
                      "%s: pre_scale with a saved state needs mask int32 [groups*X*Y*Z + groups] (one word per image group behind the voxel masks)", op);
    }
    const int *live_p = nullptr;
    const float4 *empty_p = nullptr;
    if (tile_live != nullptr || ps_empty != nullptr) {     // the producer's occupancy words + this geometry's constants (both or none)
        GENRE_REQUIRE(tile_live && ps_empty && is_i32(tile_live, 4) && is_contiguous(tile_live) &&
                          tile_live->size[0] == D.groups && tile_live->size[1] == (D.X + kBX - 1) / kBX &&
                          tile_live->size[2] == (D.Y + kBY - 1) / kBY && tile_live->size[3] == (D.Z + kBZ - 1) / kBZ,
                      "%s: tile_live must be int32 [groups = %d, %d, %d, %d] (one word per %dx%dx%d-voxel brick and image group)",
                      op, D.groups, (D.X + kBX - 1) / kBX, (D.Y + kBY - 1) / kBY, (D.Z + kBZ - 1) / kBZ, kBX, kBY, kBZ);
        GENRE_REQUIRE(is_f32(ps_empty, 2) && is_contiguous(ps_empty) && ps_empty->size[0] == D.nseg && ps_empty->size[1] == 4 &&
                          aligned16(ps_empty->data),
                      "%s: ps_empty must be fp32 [nseg, 4]: (P, S, scratch line as int32 bits, 0) of every segment on the "
                      "constant volume", op);
        live_p = (const int *)tile_live->data;
        empty_p = (const float4 *)ps_empty->data;
        D.nbricks = (int)(tile_live->size[1] * tile_live->size[2] * tile_live->size[3]);
    }
    GENRE_REQUIRE(D.nseg >= 1, "%s: the segment table is empty", op);
    hipStream_t st = (hipStream_t)stream;
    if (save && pre_scale != 0.0f) {          // the groups' "some voxel passes the clamp" words behind the masks (set by the sampler)
        GENRE_REQUIRE(hipMemsetAsync((unsigned *)mask->data + (int64_t)D.groups * D.X * D.Y * D.Z, 0, (size_t)D.groups * 4, st) == hipSuccess,
                      "%s: hipMemsetAsync of the group words failed", op);
    }
    // 1024 threads: 16 waves share one tile, two workgroups per CU = 8 waves per SIMD (measured 257 us against 318 with 512);
    // with half bricks (kBY == 4: a 29 KB tile) 512 threads, four workgroups per CU
    constexpr int nt = kBY == 4 ? 512 : 1024;
    const size_t lds = (size_t)kLinesF * kImgs * 4 + (size_t)(nt / 64) * 64 * 16 + (size_t)(nt / 64) * 4;   // tile, records, wave flags
    const dim3 grid((unsigned)fwd_rows->size[0], (unsigned)D.groups);
#ifdef GENRE_BM_TIMELINE
    static unsigned long long *tl_buf = nullptr;
    const size_t tl_n = (size_t)grid.x * grid.y * 8;
    if (getenv("GENRE_BM_TIMELINE")) {
        if (tl_buf) { (void)hipFree(tl_buf); tl_buf = nullptr; }
        (void)hipMalloc(&tl_buf, tl_n * 8);
        (void)hipMemsetAsync(tl_buf, 0, tl_n * 8, st);
    }
#endif
#define GENRE_BM_SAMPLE_NT(PSV, SV, HV, NTV)                                                                              \
    do {                                                                                                                  \
        static std::atomic<uint64_t> done_{0};                                                                            \
        if (!reserve_lds(op, reinterpret_cast<const void *>(&bm_sample_kernel<PSV, SV, HV, NTV>), lds, done_)) return 0;  \
        bm_sample_kernel<PSV, SV, HV, NTV><<<grid, NTV, lds, st>>>(                                                       \
            D, (const float *)vox->data, (const int4 *)segs->data, (const int *)rec_f->data, (const int4 *)fwd_rows->data, \
            (float *)ps_scratch->data, save ? (float *)p_stash->data : nullptr,                                           \
            (save && pre_scale != 0.0f) ? (unsigned *)mask->data : nullptr, live_p, empty_p GENRE_BTL_ARG);               \
    } while (0)
#define GENRE_BM_SAMPLE(PSV, SV) do { if (live_p) GENRE_BM_SAMPLE_NT(PSV, SV, true, nt); else GENRE_BM_SAMPLE_NT(PSV, SV, false, nt); } while (0)
    if (pre_scale != 0.0f) { if (save) GENRE_BM_SAMPLE(true, true); else GENRE_BM_SAMPLE(true, false); }
    else { if (save) GENRE_BM_SAMPLE(false, true); else GENRE_BM_SAMPLE(false, false); }
#undef GENRE_BM_SAMPLE
#undef GENRE_BM_SAMPLE_NT
    GENRE_LAUNCH_CHECK("render_bm forward (bricks)");
#ifdef GENRE_BM_TIMELINE
    if (tl_buf && getenv("GENRE_BM_TIMELINE")) {            // dump: [grid.y][grid.x][8] stamps of the launch just made
        (void)hipStreamSynchronize(st);
        unsigned long long *h = (unsigned long long *)malloc(tl_n * 8);
        (void)hipMemcpy(h, tl_buf, tl_n * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("GENRE_BM_TIMELINE"), "wb");
        if (f) { int hdr[4] = {(int)grid.x, (int)grid.y, 32, nt}; fwrite(hdr, 4, 4, f); fwrite(h, 8, tl_n, f); fclose(f); }
        free(h);
    }
#endif
    bm_combine_fwd_kernel<<<dim3((unsigned)((D.R * D.R + 7) / 8), (unsigned)D.groups), 256, 0, st>>>(
        D, (const float *)ps_scratch->data, (const int *)ray_ptr->data, (const int *)ray_seg->data,
        (const double2 *)ray_pre->data, view4(out));
    GENRE_LAUNCH_CHECK("render_bm forward (rays)");
    return 1;
}

extern "C" int genre_render_bm_backward(const genre_tensor *grad_out, const genre_tensor *grad_vox,
                                        const genre_tensor *segs, const genre_tensor *ray_ptr,
                                        const genre_tensor *ray_seg, const genre_tensor *ray_pre,
                                        const genre_tensor *ent, const genre_tensor *rec_b,
                                        const genre_tensor *bwd_rows, const genre_tensor *depth_weight,
                                        const genre_tensor *ps_scratch, const genre_tensor *tr_scratch,
                                        const genre_tensor *p_stash, const genre_tensor *mask, float pre_scale,
                                        int pull_brick, void *stream)
{
    const char *op = "render_bm_backward";
    BmDims D{};
    if (!check_bm(op, grad_vox, grad_out, segs, ray_ptr, ray_seg, ray_pre, D, true)) return 0;
    GENRE_REQUIRE(pull_brick == 488 || pull_brick == 888, "%s: pull_brick must be 488 (4x8x8 voxels) or 888 (8x8x8)", op);
    const int px = pull_brick / 100;
    if (!check_rows(op, D, bwd_rows, px, 8, 8)) return 0;
    D.gx = grad_vox->stride[2]; D.gy = grad_vox->stride[3]; D.gz = grad_vox->stride[4];
    D.pre_scale = pre_scale;
    GENRE_REQUIRE(is_i32(ent, 2) && ent->size[1] == 4 && is_contiguous(ent) && aligned16(ent->data), "%s: ent must be int32 [E,4]", op);
    GENRE_REQUIRE(is_i32(rec_b, 2) && rec_b->size[1] == kRec && is_contiguous(rec_b) && aligned16(rec_b->data),
                  "%s: rec_b must be a contiguous int32 [SB,12] tensor", op);
    GENRE_REQUIRE(is_f32(depth_weight, 1) && is_contiguous(depth_weight) && depth_weight->size[0] >= 1 &&
                      depth_weight->size[0] <= 256, "%s: depth_weight must be fp32 [ZR], 1 <= ZR <= 256", op);
    D.ZR = (int)depth_weight->size[0];
    const int64_t per = (int64_t)D.groups * D.nseg * 2 * kImgs;
    GENRE_REQUIRE(is_f32(ps_scratch, 1) && is_contiguous(ps_scratch) && ps_scratch->size[0] >= per &&
                      is_f32(tr_scratch, 1) && is_contiguous(tr_scratch) && tr_scratch->size[0] >= per,
                  "%s: ps_scratch / tr_scratch must hold groups*nseg*64 floats", op);
    GENRE_REQUIRE(is_f32(p_stash, 1) && is_contiguous(p_stash) && p_stash->size[0] % kImgs == 0 && D.groups > 0 &&
                      p_stash->size[0] / kImgs % D.groups == 0, "%s: p_stash must be the forward's [groups*S*32] buffer", op);
    D.nslot = p_stash->size[0] / kImgs / D.groups;
    GENRE_REQUIRE(pre_scale == 0.0f || (is_i32(mask, 1) && is_contiguous(mask) && mask->size[0] >= (int64_t)D.groups * D.X * D.Y * D.Z + D.groups),
                  "%s: pre_scale needs the forward's mask int32 [groups*X*Y*Z + groups]", op);
    hipStream_t st = (hipStream_t)stream;
    bm_combine_bwd_kernel<<<dim3((unsigned)((D.R * D.R + 7) / 8), (unsigned)D.groups), 256, 0, st>>>(
        D, (const float *)ps_scratch->data, (const int *)ray_ptr->data, (const int *)ray_seg->data,
        (const double2 *)ray_pre->data, view4(grad_out), (float *)tr_scratch->data,
        pre_scale != 0.0f ? (const unsigned *)mask->data + (int64_t)D.groups * D.X * D.Y * D.Z : nullptr);
    GENRE_LAUNCH_CHECK("render_bm backward (rays)");
    const int nb = ((D.X + px - 1) / px) * ((D.Y + 7) / 8) * ((D.Z + 7) / 8);
    const dim3 grid((unsigned)bwd_rows->size[0], (unsigned)D.groups);
    const bool split = bwd_rows->size[0] > nb;     // some bricks are split over several rows: those add atomically
    // 4x8x8: 64 KB tile, 768 threads, two workgroups per CU (72 VGPRs: 6 of the 7 possible waves per SIMD);
    // 8x8x8: 128 KB tile, 1024 threads, one workgroup per CU
#define GENRE_BM_SCATTER(PSV, PXV, NTV)                                                                                   \
    do {                                                                                                                  \
        if (split) {                                                                                                      \
            bm_zero_shared_kernel<PXV, 8, 8><<<dim3(grid.x < 1024u ? grid.x : 1024u, grid.y), kThreads, 0, st>>>(        \
                D, (const int4 *)bwd_rows->data, (int)bwd_rows->size[0], (float *)grad_vox->data);                        \
            GENRE_LAUNCH_CHECK("render_bm backward (zero shared bricks)");                                                \
        }                                                                                                                 \
        constexpr size_t lds = (size_t)PXV * 64 * kImgs * 8 + (size_t)(NTV / 64) * kMaxSeg * kRecL * 4 + (size_t)(PXV * 64 + PXV + 4) * 4; \
        static std::atomic<uint64_t> done_{0};                                                                            \
        if (!reserve_lds(op, reinterpret_cast<const void *>(&bm_scatter_kernel<PSV, PXV, 8, 8, NTV>), lds, done_)) return 0; \
        bm_scatter_kernel<PSV, PXV, 8, 8, NTV><<<grid, NTV, lds, st>>>(                                                   \
            D, (const int4 *)ent->data, (const int *)rec_b->data, (const int4 *)bwd_rows->data,                           \
            (const float *)depth_weight->data, (const float *)tr_scratch->data, (const float *)p_stash->data,             \
            pre_scale != 0.0f ? (const unsigned *)mask->data : nullptr, (float *)grad_vox->data);                         \
    } while (0)
    if (px == 4) { if (pre_scale != 0.0f) GENRE_BM_SCATTER(true, 4, GENRE_BM_SCATTER_NT); else GENRE_BM_SCATTER(false, 4, GENRE_BM_SCATTER_NT); }
    else { if (pre_scale != 0.0f) GENRE_BM_SCATTER(true, 8, 1024); else GENRE_BM_SCATTER(false, 8, 1024); }
#undef GENRE_BM_SCATTER
    GENRE_LAUNCH_CHECK("render_bm backward (bricks)");
    return 1;
}
