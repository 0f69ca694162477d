"""Geometry-only tables of the batch-minor tile renderer (csrc/sph_render_bm.hip).

render_spherical (toolbox/spherical_proj.py:31-72 of the reference) samples the volume at 2*dir*(1 - k/(z_res-1)),
k = 0 .. z_res-1, along res*res rays that all end at the centre of the cube.  Where each sample falls -- its base
voxel, its eight trilinear weights (ATen grid_sampler_3d, align_corners=True == PyTorch 0.4.1), the brick that
holds it -- depends on the geometry only, never on the batch.  It is worked out here once per geometry, with
the exact fp64 -> fp32 operation sequence of the reference's `grid` buffer and of ATen, and shipped to the GPU as
flat tables; the kernels then spend their instructions on the 32 images of a voxel line, not on geometry.

Decomposition: bricks of BX x BY x BZ voxels.  A SEGMENT is a maximal run of consecutive samples of one ray whose base
voxel lies in one brick (<= MAXSEG = 16 samples).  Forward: a workgroup stages its brick (+1 voxel on the high sides)
for 32 images in LDS and a wave marches a segment serially, lanes = images, producing the segment's transmittance
P = prod(1-p) and partial expectation S = sum T_k p_k w_k; a per-ray pass combines them (the scan is associative).
Backward: dL/dp_k = g T_k (w_k - R_{k+1}) needs T at the segment's start and R behind its end (per-ray pass), then
every brick PULLS the samples that touch one of its voxels and accumulates them into an LDS tile it alone owns
(plain stores, every voxel of grad_vox written exactly once).

Formats (all little-endian 32-bit words unless noted) -- see include/genre_hip.h, "batch-minor tile renderer":
  segs      int32 [nseg,4]   (line of the segment in the per-segment scratch buffers = its position in RAY order, first
                              sample k0, length L, slot of the first sample)   sorted by (brick, ray, k0)
  rec_f     int32 [S,12]     per sample slot: (tile byte offset, depth_weight[k] bits, 0, 0,
                              w(x0y0z0), w(x1y0z0), w(x0y1z0), w(x1y1z0), w(x0y0z1), w(x1y0z1), w(x0y1z1), w(x1y1z1));
                              S = samples + SLOT_PAD: the last SLOT_PAD slots belong to no sample (all zero) -- the kernels
                              issue a FIXED number of loads per segment (every lane of a wave loads 16 bytes of records =
                              21.3 records from the segment's first; all 16 sample slots of a segment are fetched) so
                              that their waits on the in-order load counter can be exact; the excess is never used
  fwd_rows  int32 [rows,4]   (brick, seg begin, seg end, flag | bx << 8 | by << 16 | bz << 24); flag 2 = padding row (skipped);
                              (bx, by, bz) = the brick's coordinates in bricks, so that the kernels divide nothing; order: see _xcd_order
  ray_ptr   int32 [RR+1], ray_seg int32 [nseg]   the segments of every ray in sample order
  ray_pre   float64 [RR,2]   (P0, S0) of the samples before the ray enters the volume (p = 1e-5 each)
  ent       int32 [E,4]      backward listing: (the segment's scratch line = ray-order position, slot of its first sample,
                              i0 | i1 << 6 | L << 12 | k0 << 18, rec_b slot of sample i0): samples i0..i1-1 of the
                              segment touch the brick of the row
  rec_b     int32 [SB,12]    (tile byte offset in the brick's own fp64 tile -- may point outside it for corners the
                              brick does not own --, ownership bits (corner c, z half h) -> bit c + 4h, 0, 0, 8 weights);
                              SB = listed samples + REC_PAD zero records (same reason as SLOT_PAD)
  bwd_rows  int32 [rows,4]   (pull brick, ent begin, ent end, shared | bx << 8 | by << 16 | bz << 24); shared = 1: the brick is split
                              over several rows, which add their tiles atomically onto pre-zeroed voxels.  The backward's
                              ("pull") bricks are PULL = 4x8x8 voxels like the forward's; 8x8x8 (a segment then touches fewer bricks,
                              each of which re-reads its saved samples, but only one workgroup fits a CU) measured slower
"""
import numpy as np

import os

# the forward's bricks; must match csrc/sph_render_bm.hip (kBX, kBY, kBZ: toolbox/_fused_render.py checks it against the loaded
# library's genre_bm_brick()).  GENRE_BM_BY (4 | 8) is the A/B switch of both sides (tools/build_variants.sh -DGENRE_BM_BY=4)
BX, BY, BZ = 4, int(os.environ.get("GENRE_BM_BY", "8")), 8
TX, TY, TZ = BX + 1, BY + 1, BZ + 1
MAXSEG = 16
SLOT_PAD = 24                   # unused slots behind the last sample (see rec_f): >= 22, every lane of a wave loads 16 bytes of rec_f
REC_PAD = 22                    # unused records behind rec_b: every lane of a wave loads 16 bytes from an entry's first record
LINE_F = 128                    # bytes of one voxel line in the forward tile (32 images x fp32)
LINE_B = 256                    # ... in the backward tile (32 images x fp64)
SPLIT_F = 4096                  # samples per forward row
SPLIT_B = 2048                  # listed samples per backward row
LO = np.float32(1e-5)           # spherical_proj.py:66


def _axis(d2a, a, size):
    """one axis of ATen's grid_sampler_3d (align_corners=True): base index and the two corner weights, fp32"""
    one, two = np.float32(1), np.float32(2)
    g = (d2a[:, None] * a[None, :]).astype(np.float32)                       # float(grid) of spherical_proj.py:56
    ix = ((g + one) / two) * np.float32(size - 1)
    f = np.floor(ix)
    i0 = f.astype(np.int32)
    w1 = ix - f
    w0 = (f + one) - ix
    return i0, w0.astype(np.float32), w1.astype(np.float32)


PULL = (4, 8, 8)                # the backward's bricks (csrc/sph_render_bm.hip: pull_brick 488 or 888; 8x8x8 measured slower)


def build_bm_tables(X, Y, Z, dirs64, z_res, depth_weight, split_f=SPLIT_F, split_b=SPLIT_B, pull=PULL):
    R = dirs64.shape[0]
    RR = R * R
    assert z_res <= 256 and RR < (1 << 22)
    dw = np.asarray(depth_weight, np.float32).reshape(-1)
    assert dw.shape[0] == z_res
    d2 = dirs64.reshape(-1, 3).astype(np.float64) * 2.0
    step = 1.0 / (z_res - 1) if z_res > 1 else 0.0
    alpha = np.arange(z_res, dtype=np.float64) * step
    alpha[-1] = 1.0                                                          # numpy.linspace(0, 1, z_res)
    a = 1.0 - alpha
    sizes, bsz = (X, Y, Z), (BX, BY, BZ)
    nbr = [-(-s // b) for s, b in zip(sizes, bsz)]
    nb = nbr[0] * nbr[1] * nbr[2]

    base, w0s, w1s, v1s = [], [], [], []
    inside = None
    for ax in range(3):
        i0, w0, w1 = _axis(d2[:, ax], a, sizes[ax])
        ins = (i0 >= -1) & (i0 < sizes[ax])
        inside = ins if inside is None else (inside & ins)
        low = i0 == -1                                    # the -1 corner is zero padding: re-base on voxel 0, weight of the
        w0 = np.where(low, w1, w0)                        # +1 corner set to 0 (same value, same order of the non-zero terms)
        w1 = np.where(low, np.float32(0), w1)
        i0 = np.where(low, 0, i0)
        base.append(i0); w0s.append(w0); w1s.append(w1)
        v1s.append((i0 + 1 < sizes[ax]) & ~low)           # the +1 corner is a real voxel that receives a weight
    kin = np.where(inside.any(1), inside.argmax(1), z_res).astype(np.int32)
    assert (inside == (np.arange(z_res)[None, :] >= kin[:, None])).all(), "inside samples are not a suffix of the ray"

    # ---- samples before the volume: p = clamp(0) = 1e-5 (toolbox/spherical_proj.py:66) ----
    q1 = 1.0 - float(LO)
    pw = q1 ** np.arange(z_res + 1, dtype=np.float64)                       # T before sample k
    s_pre = np.concatenate(([0.0], np.cumsum(pw[:-1] * float(LO) * dw.astype(np.float64))))
    ray_pre = np.stack([pw[kin], s_pre[kin]], 1)

    # ---- segments ----
    qq, kk = np.nonzero(inside)                                             # ray-major, k ascending
    bxyz = [base[ax][qq, kk] for ax in range(3)]
    brick = ((bxyz[0] // BX).astype(np.int64) * nbr[1] + bxyz[1] // BY) * nbr[2] + bxyz[2] // BZ
    ns = qq.shape[0]
    new = np.ones(ns, bool)
    if ns > 1:
        new[1:] = (qq[1:] != qq[:-1]) | (brick[1:] != brick[:-1])
    start = np.nonzero(new)[0]
    run_id = np.cumsum(new) - 1
    pos = np.arange(ns) - start[run_id]                                     # position inside the run
    new |= (pos % MAXSEG) == 0                                              # cut runs longer than MAXSEG
    start = np.nonzero(new)[0]
    seg_of = np.cumsum(new) - 1                                             # ray-order segment id of every sample
    nseg = start.shape[0]
    seg_len = np.diff(np.concatenate((start, [ns]))).astype(np.int32)
    seg_q, seg_k0, seg_brick = qq[start].astype(np.int32), kk[start].astype(np.int32), brick[start]
    order = np.lexsort((seg_k0, seg_q, seg_brick))                          # forward order: (brick, ray, k0)
    rank = np.empty(nseg, np.int64)
    rank[order] = np.arange(nseg)
    slot0 = np.concatenate(([0], np.cumsum(seg_len[order])))[:-1]           # by forward position
    # column 0: the segment's position in RAY order (rank[.] inverted) = its line in the per-segment scratch buffers
    # (P, S) / (g T, R): a ray's segments are neighbours there, so the per-ray passes stream contiguous memory
    rpos = order.astype(np.int32)                                            # forward position -> ray-order position
    segs = np.stack([rpos, seg_k0[order], seg_len[order], slot0.astype(np.int32)], 1).astype(np.int32)
    # per-ray segment lists, in sample order (ray-order segments are already sorted by (q, k0))
    ray_seg = rank.astype(np.int32)
    ray_ptr = np.searchsorted(seg_q, np.arange(RR + 1), side="left").astype(np.int32)
    # sample -> slot
    samp_seg = rank[seg_of]                                                 # forward segment id
    samp_i = (np.arange(ns) - start[seg_of]).astype(np.int32)
    samp_slot = slot0[samp_seg] + samp_i

    # ---- per-sample weights ----
    wx = (w0s[0][qq, kk], w1s[0][qq, kk])
    wy = (w0s[1][qq, kk], w1s[1][qq, kk])
    wz = (w0s[2][qq, kk], w1s[2][qq, kk])
    wts = np.empty((ns, 8), np.float32)
    for c in range(8):                                                      # ATen: (wx * wy) * wz, corner bit 0 = x
        wts[:, c] = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1]
    lx, ly, lz = bxyz[0] % BX, bxyz[1] % BY, bxyz[2] % BZ
    rec_f = np.zeros((ns + SLOT_PAD, 12), np.int32)
    rec_f[samp_slot, 0] = ((lx * TY + ly) * TZ + lz) * LINE_F
    rec_f[samp_slot, 1] = dw[kk].view(np.int32)
    rec_f[samp_slot, 4:12] = wts.view(np.int32)

    # ---- forward rows ----
    seg_brick_f = seg_brick[order]
    sb = np.searchsorted(seg_brick_f, np.arange(nb), side="left")
    se = np.searchsorted(seg_brick_f, np.arange(nb), side="right")
    cum = np.concatenate(([0], np.cumsum(segs[:, 2].astype(np.int64))))
    fwd_rows = _pack_coords(_split_rows(sb, se, cum, split_f, 0), nbr)

    # ---- backward listing: every PULL brick lists the samples that touch one of its voxels ----
    assert tuple(pull) in ((4, 8, 8), (8, 8, 8))
    pnbr = [-(-sz // b) for sz, b in zip(sizes, pull)]
    pnb = pnbr[0] * pnbr[1] * pnbr[2]
    pb = [bxyz[ax] // pull[ax] for ax in range(3)]                          # pull brick of the base corner
    flags = []
    for ax in range(3):
        v1 = v1s[ax][qq, kk]
        flags.append(v1 & ((bxyz[ax] + 1) // pull[ax] != pb[ax]))           # the +1 corner lies in the next pull brick
    keys, whos = [], []
    sid = np.arange(ns, dtype=np.int64)
    for d in range(8):
        m = np.ones(ns, bool)
        for ax in range(3):
            if (d >> ax) & 1:
                m &= flags[ax]
        if not m.any():
            continue
        dxyz = [(d >> ax) & 1 for ax in range(3)]
        b2 = ((pb[0][m] + dxyz[0]).astype(np.int64) * pnbr[1] + (pb[1][m] + dxyz[1])) * pnbr[2] + (pb[2][m] + dxyz[2])
        keys.append((b2 << 40) | (samp_seg[m] << 8) | samp_i[m])
        whos.append(np.stack([sid[m], np.full(int(m.sum()), d, np.int64)], 1))
    keys = np.concatenate(keys)
    whos = np.concatenate(whos)
    o = np.argsort(keys, kind="stable")
    keys, whos = keys[o], whos[o]
    nl = keys.shape[0]
    grp = keys >> 8                                                          # (brick, segment)
    gnew = np.ones(nl, bool)
    gnew[1:] = grp[1:] != grp[:-1]
    gstart = np.nonzero(gnew)[0]
    gcount = np.diff(np.concatenate((gstart, [nl])))
    i_first = (keys[gstart] & 255).astype(np.int32)
    i_last = (keys[gstart + gcount - 1] & 255).astype(np.int32)
    assert ((i_last - i_first + 1) == gcount).all(), "a brick's share of a segment is not contiguous"
    ent_seg = ((keys[gstart] >> 8) & 0xFFFFFFFF).astype(np.int64)
    pack = i_first | ((i_last + 1) << 6) | (segs[ent_seg, 2] << 12) | (segs[ent_seg, 1] << 18)
    ent = np.stack([segs[ent_seg, 0], segs[ent_seg, 3], pack.astype(np.int32), gstart.astype(np.int32)], 1).astype(np.int32)
    ent_brick = (keys[gstart] >> 40).astype(np.int64)
    # rec_b of every listed sample, relative to the pulling brick
    s_id, dcode = whos[:, 0], whos[:, 1]
    rel = []
    for ax in range(3):                                                     # base corner relative to the pulling brick
        rel.append(bxyz[ax][s_id] - (pb[ax][s_id] + ((dcode >> ax) & 1)) * pull[ax])
    rec_b = np.zeros((nl + REC_PAD, 12), np.int32)
    rec_b[:nl, 0] = ((rel[0] * pull[1] + rel[1]) * pull[2] + rel[2]) * LINE_B
    own = np.zeros(nl, np.int32)
    valid = [(np.ones(ns, bool), v1s[ax][qq, kk]) for ax in range(3)]
    for c in range(8):
        okc = np.ones(nl, bool)
        for ax in range(3):
            bit = (c >> ax) & 1
            coord = rel[ax] + bit
            okc &= (coord >= 0) & (coord < pull[ax]) & valid[ax][bit][s_id]
        h, cxy = c >> 2, c & 3
        own |= okc.astype(np.int32) << (cxy + 4 * h)
    rec_b[:nl, 1] = own
    rec_b[:nl, 4:12] = wts[s_id].view(np.int32)
    assert (own != 0).all()
    eb = np.searchsorted(ent_brick, np.arange(pnb), side="left")
    ee = np.searchsorted(ent_brick, np.arange(pnb), side="right")
    cumb = np.concatenate(([0], np.cumsum((i_last + 1 - i_first).astype(np.int64))))
    bwd_rows = _pack_coords(_split_rows(eb, ee, cumb, split_b, 1), pnbr)

    out = dict(segs=segs, rec_f=rec_f, fwd_rows=fwd_rows, ray_ptr=ray_ptr, ray_seg=ray_seg, ray_pre=ray_pre,
               ent=ent, rec_b=rec_b, bwd_rows=bwd_rows, kin=kin, pull=np.asarray(pull, np.int32))
    return out


def _pack_coords(rows, nbr):
    """flag | bx << 8 | by << 16 | bz << 24 in column 3: the brick's coordinates (gfx950 has no integer divide: ~40 instructions
    each, three per workgroup to take a brick id apart)"""
    assert max(nbr) <= 255
    b = rows[:, 0].astype(np.int64)
    bx, by, bz = b // (nbr[1] * nbr[2]), (b // nbr[2]) % nbr[1], b % nbr[2]
    rows = rows.copy()
    rows[:, 3] = (rows[:, 3].astype(np.int64) | (bx << 8) | (by << 16) | (bz << 24)).astype(np.int32)
    return rows


def row_flag(w):
    """the flag of a row's fourth word (0 plain, 1 shared, SKIP padding)"""
    return int(w) & 255


def _split_rows(begin, end, cum, split, shared_mode):
    """one row per brick, more when it holds over `split` samples (cut at item boundaries); heaviest first"""
    rows = []
    for b in range(begin.shape[0]):
        b0, b1 = int(begin[b]), int(end[b])
        total = int(cum[b1] - cum[b0])
        if total <= split:
            rows.append((b, b0, b1, 0, total))
            continue
        parts = -(-total // split)
        target = -(-total // parts)
        cuts = [b0]
        while cuts[-1] < b1:
            nxt = int(np.searchsorted(cum, cum[cuts[-1]] + target, side="right")) - 1
            nxt = min(max(nxt, cuts[-1] + 1), b1)
            cuts.append(nxt)
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            rows.append((b, c0, c1, shared_mode, int(cum[c1] - cum[c0])))
    # the rows of split bricks (flag 1: they add to their brick atomically, which bm_zero_shared_kernel zeroes first) form the HEAD of
    # the table in every order, heaviest first: the zero kernel then stops at the first row that is not one instead of being
    # launched over all 16 k rows to find ~200
    head = sorted((r for r in rows if r[3] == 1), key=lambda r: -r[4])
    rest = [r for r in rows if r[3] != 1]
    if ROW_ORDER == "heaviest":
        rest.sort(key=lambda r: -r[4])
        return np.asarray([r[:4] for r in head + rest], np.int32).reshape(-1, 4)
    tail = _xcd_order(rest)
    return np.concatenate([np.asarray([r[:4] for r in head], np.int32).reshape(-1, 4), tail]) if head else tail


# Row order.  "heaviest": rows sorted by weight, heaviest first (best tail).  "xcd": weight classes, brick order inside a
# class, interleaved so that each XCD walks a contiguous slab (_xcd_order) -- measured SLOWER on MI355X at batch 32
# (forward 221 -> 236 us, backward 630 -> 820 us; a pure brick-order interleave: 244 / 857 us), so it is not the default.
ROW_ORDER = "heaviest"
SKIP = 2                        # row flag: padding, the workgroup returns at once
FIXED_COST = 256                # weight of a row beyond its samples (tile staging / zeroing / flush), in samples


def _xcd_order(rows):
    """MI355X deals the workgroups of a launch to its 8 XCDs round-robin by linear id, and each XCD has its own L2.
    Neighbouring bricks share voxel lines (the forward's halo) and saved samples (a segment is pulled by every brick it
    touches).  Rows are grouped into weight classes (powers of two), heaviest class first -- so the launch still ends
    on its lightest rows -- and inside a class they stay in BRICK order, cut into 8 consecutive parts that are
    interleaved: within a class, row 8 i + x is the i-th row of part x, i.e. XCD x walks a contiguous slab of the
    volume and meets the shared lines in its own L2.  Classes are padded to a multiple of 8 rows with SKIP rows."""
    classes = {}
    for r in rows:
        classes.setdefault(int(np.log2(r[4] + 1)), []).append(r)
    out = []
    for c in sorted(classes, reverse=True):
        part = classes[c]                                                  # already in brick order
        depth = -(-len(part) // 8)
        block = np.zeros((depth * 8, 4), np.int32)
        block[:, 3] = SKIP
        for j, r in enumerate(part):
            x, i = divmod(j, depth)
            block[i * 8 + x] = r[:4]
        out.append(block)
    return np.concatenate(out) if out else np.zeros((0, 4), np.int32)
