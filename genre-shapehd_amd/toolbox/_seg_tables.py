"""Geometry-only tables of the standard-layout (NCXYZ) SEGMENT renderer (csrc/sph_render_seg.hip).

render_spherical (toolbox/spherical_proj.py:31-72 of the reference) samples the volume at 2*dir*(1 - k/(z_res-1)),
k = 0 .. z_res-1, along res*res rays that end at the centre of the cube.  The ray integral is associative: a run of
consecutive samples contributes (P, S) = (prod(1-p_k), sum_k T_k p_k w_k with T = 1 at its start), and a per-ray pass
chains the runs.  So the sampler never has to put anything per SAMPLE through HBM (csrc/sph_render.hip wrote and re-read
16 MiB of raw sample values per image): a workgroup stages a 16^3-voxel brick (+ one voxel on the high sides) of one or
two images in LDS, ONE LANE marches ONE SEGMENT -- a run of <= max_seg consecutive samples of one ray whose base voxel
lies in that brick -- and writes 8 bytes.

Which sample falls into which brick depends on the geometry only; it is worked out here with exactly the fp64 -> fp32
operation sequence of the reference's `grid` buffer (spherical_proj.py:50-56) and of ATen's grid_sampler_3d
(align_corners=True == PyTorch 0.4.1), the same sequence the kernel executes (csrc/sph_render.hip: sample_pos, locate).

Formats (int32 unless noted; see include/genre_hip.h, "segment renderer"):
  segs      [nseg,4]   (ray q, first sample k0 | length L << 8, scratch line, bx | by << 10 | bz << 20 of its brick); grouped by row, inside a row sorted
                        by (L descending, q, k0): the 64 segments a wave marches together have (nearly) one length
  seg_rows  [rows,4]   (brick, seg begin, seg end, bx | by << 10 | bz << 20): one workgroup each, heaviest first; bricks with more than `split`
                        segments are cut into several rows
  ray_nseg  [RR]       segments of every ray; segment number s (in sample order) of ray q owns scratch line s*RR + q, so that
                        the per-ray pass (lane = ray) reads 256 contiguous bytes per wave and step
  ray_pre   float64 [RR,2]  (transmittance, partial sum) of the samples before the ray enters the volume (p = clamp(0) = 1e-5)
  kin       [RR]       first in-volume sample of every ray
  line_w    float32 [smax*RR,2]  per scratch line: depth weight of its segment's first and of its last sample
  bwd_rows  [rows,4]   like seg_rows with pieces of <= bwd_split segments; bit 30 of the last column: the brick is split over several rows,
                        bit 31: the first of them; the rows of split bricks come first -- the workgroups of the backward (seg_scatter_kernel)
  smax      [1]        scratch lines per ray (= max(ray_nseg)): the scratch holds smax*RR (P, S) pairs per image
"""
import os

import numpy as np

BRICK = 16                      # must match kBrick of csrc/sph_render.hip / sph_render_seg.hip
MAX_SEG = 16                    # samples per segment at most (a run of n is cut into ceil(n / MAX_SEG) equal parts)
MAX_SEG_SMALL = int(os.environ.get("GENRE_SEG_MAXSEG_SMALL", "16"))     # ... for batches of fewer than SMALL_BATCH images (A/B switch)
SPLIT = int(os.environ.get("GENRE_SEG_SPLIT", "1024"))                  # segments per row at most ...
SPLIT_SMALL = int(os.environ.get("GENRE_SEG_SPLIT_SMALL", "256"))       # ... when fewer than SMALL_BATCH images have to fill 256 CUs
BWD_SPLIT = int(os.environ.get("GENRE_SEG_BWD_SPLIT", "4096"))          # segments per row of the backward at most ...
BWD_SPLIT_SMALL = int(os.environ.get("GENRE_SEG_BWD_SPLIT_SMALL", "512"))   # ... for small batches (a row = one workgroup of 8 waves)
FIXED_COST = 6                  # weight of a row beyond its march steps (tile staging), in 64-segment march steps
LO = np.float32(1e-5)           # spherical_proj.py:66


def sample_cells(X, Y, Z, dirs64, z_res):
    """base voxel index of every sample per axis (ATen's floor(((g + 1) / 2) * (size - 1)) in fp32 on the fp64 -> fp32 sample
    position) and the mask of samples with a trilinear corner inside the volume"""
    d2 = dirs64.reshape(-1, 3).astype(np.float64) * 2.0
    step = 1.0 / (z_res - 1) if z_res > 1 else 0.0
    alpha = np.arange(z_res, dtype=np.float64) * step
    alpha[-1] = 1.0                                                          # numpy.linspace(0, 1, z_res)
    a = 1.0 - alpha
    one, two = np.float32(1), np.float32(2)
    cells, inside = [], None
    for ax, size in enumerate((X, Y, Z)):
        g = (d2[:, None, ax] * a[None, :]).astype(np.float32)                # float(grid), spherical_proj.py:56
        i0 = np.floor(((g + one) / two) * np.float32(size - 1)).astype(np.int32)
        ins = (i0 >= -1) & (i0 < size)
        inside = ins if inside is None else (inside & ins)
        cells.append(i0)
    return cells, inside


def build_seg_tables(X, Y, Z, dirs64, z_res, depth_weight, max_seg=MAX_SEG, split=SPLIT, bwd_split=None):
    R = dirs64.shape[0]
    RR = R * R
    bwd_split = BWD_SPLIT if bwd_split is None else bwd_split
    assert z_res <= 256 and RR < (1 << 24) and 1 <= max_seg <= 255 and max(X, Y, Z) <= 1023 * BRICK
    dw = np.asarray(depth_weight, np.float32).reshape(-1)
    assert dw.shape[0] == z_res
    cells, inside = sample_cells(X, Y, Z, dirs64, z_res)
    kin = np.where(inside.any(1), inside.argmax(1), z_res).astype(np.int32)
    assert (inside == (np.arange(z_res)[None, :] >= kin[:, None])).all(), "inside samples are not a suffix of the ray"
    nbx, nby, nbz = -(-X // BRICK), -(-Y // BRICK), -(-Z // BRICK)

    # ---- samples before the volume: p = clamp(0) = 1e-5 (spherical_proj.py:66) ----
    q1 = 1.0 - float(LO)
    pw = q1 ** np.arange(z_res + 1, dtype=np.float64)                       # T before sample k
    s_pre = np.concatenate(([0.0], np.cumsum(pw[:-1] * float(LO) * dw.astype(np.float64))))
    ray_pre = np.stack([pw[kin], s_pre[kin]], 1)

    # ---- runs of consecutive samples of one ray in one brick, cut into equal parts of <= max_seg samples ----
    qq, kk = np.nonzero(inside)                                             # ray-major, k ascending
    ns = qq.shape[0]
    b = [np.clip(cells[ax][qq, kk], 0, (X, Y, Z)[ax] - 1) // BRICK for ax in range(3)]     # a base corner of -1 is zero padding
    brick = (b[0].astype(np.int64) * nby + b[1]) * nbz + b[2]
    new = np.ones(ns, bool)
    if ns > 1:
        new[1:] = (qq[1:] != qq[:-1]) | (brick[1:] != brick[:-1])
    start = np.nonzero(new)[0]
    run_len = np.diff(np.concatenate((start, [ns])))
    run_id = np.cumsum(new) - 1
    pos = np.arange(ns) - start[run_id]
    parts = -(-run_len // max_seg)
    piece = -(-run_len // parts)                                            # length of all but the last part
    new |= (pos % piece[run_id]) == 0
    start = np.nonzero(new)[0]
    nseg = start.shape[0]
    seg_len = np.diff(np.concatenate((start, [ns]))).astype(np.int64)
    seg_q, seg_k0, seg_brick = qq[start].astype(np.int64), kk[start].astype(np.int64), brick[start]
    # number of the segment inside its ray (segments are in (q, k0) order already)
    ray_first = np.searchsorted(seg_q, np.arange(RR), side="left")
    ray_nseg = (np.searchsorted(seg_q, np.arange(RR), side="right") - ray_first).astype(np.int32)
    s_in_ray = np.arange(nseg) - ray_first[seg_q]
    smax = int(ray_nseg.max()) if RR else 0
    line = s_in_ray * RR + seg_q
    assert nseg == 0 or line.max() < (1 << 31)

    # ---- rows: per brick, longest segments first; the 64 segments of a wave then share their march length ----
    order = np.lexsort((seg_k0, seg_q, -seg_len, seg_brick))
    sbr = seg_brick[order]
    packed = (sbr // (nby * nbz)) | ((sbr // nbz) % nby) << 10 | (sbr % nbz) << 20              # (the kernels divide nothing)
    segs = np.stack([seg_q[order], seg_k0[order] | (seg_len[order] << 8), line[order], packed], 1).astype(np.int32)
    sb = np.searchsorted(seg_brick[order], np.arange(nbx * nby * nbz), side="left")
    se = np.searchsorted(seg_brick[order], np.arange(nbx * nby * nbz), side="right")
    lens = seg_len[order]
    def rows_for(split, flag_split):
        rows = []
        for bid in range(nbx * nby * nbz):
            b0, b1 = int(sb[bid]), int(se[bid])
            cnt = b1 - b0
            nparts = max(1, -(-cnt // split))
            size = -(-cnt // nparts) if cnt else 0
            size = -(-size // 64) * 64                                      # whole waves
            for r0 in (range(b0, b1, size) if cnt else [b0]):
                r1 = min(r0 + size, b1) if cnt else b0
                steps = int(lens[r0:r1:64].sum())                           # a wave marches its longest (= first) segment
                bxyz = (bid // (nby * nbz)) | ((bid // nbz) % nby) << 10 | (bid % nbz) << 20  # (the kernel divides nothing)
                if flag_split and nparts > 1:
                    bxyz |= (1 << 30) | ((1 << 31) if r0 == b0 else 0)      # split brick; the first of its rows
                rows.append((bid, r0, r1, bxyz, steps + FIXED_COST))
        # heaviest first; the rows of split bricks in front of all others (they are the heaviest pieces anyway): the per-ray pass of
        # the backward, which zeroes those bricks on its way, then finds them at the head of the table
        rows.sort(key=lambda r: (-((r[3] >> 30) & 1), -r[4]))
        return (np.asarray([r[:4] for r in rows], np.int64).reshape(-1, 4) & 0xffffffff).astype(np.uint32).view(np.int32)
    seg_rows = rows_for(split, False)
    # the backward's rows: the same segments, longer pieces (a workgroup of the backward carries a 46 KB accumulation tile that it
    # flushes with atomics when its brick is split: fewer, longer rows), the split flag in bit 30
    bwd_rows = rows_for(bwd_split, True)
    # per scratch line: the depth weights of its segment's first and last sample (the kernels keep a segment's partial sum and the
    # chain's R relative to them: csrc/sph_render_seg.hip, seg_combine_kernel)
    line_w = np.zeros((max(smax, 1) * RR, 2), np.float32)
    line_w[line, 0] = dw[seg_k0]
    line_w[line, 1] = dw[seg_k0 + seg_len - 1]
    return dict(segs=segs, seg_rows=seg_rows, ray_nseg=ray_nseg, ray_pre=ray_pre, line_w=line_w, kin=kin,
                bwd_rows=bwd_rows, smax=np.asarray([smax], np.int32))
