"""Fused render_spherical: HIP kernels instead of the reference's
grid_sample -> clamp -> CalcStopProb -> matmul -> prod -> add chain
(toolbox/spherical_proj.py:62-72).

Forward : segment kernel (csrc/sph_render_seg.hip: 18^3 voxel tile staged in LDS with coalesced reads, one lane marches one
          segment of a ray through it, trilinear taps read from LDS) leaves one (P, S) pair per segment; a per-ray pass
          chains them into the spherical map.  Nothing per sample goes through memory in a forward-only pass; when a
          gradient is wanted the raw sample values of the tiles a gradient can come back through are saved too.
Backward: per-ray chains over the (P, S) pairs, dL/dp[ray, k] per segment from the saved sample values; brick kernel
          accumulates the trilinear adjoint in 64-bit fixed-point LDS tiles and writes every voxel of grad_vox once (no
          global atomics).
Which samples touch which brick depends only on the geometry; the lists are built once here
with exactly the kernel's fp64/fp32 arithmetic and cached per geometry/device."""
import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .calc_prob.calc_prob._ext import _loader

BRICK = 16              # must match kBrick in csrc/sph_render.hip
SPLIT_BWD = 8192        # backward rows above this many samples are split (atomic flush)
SPLIT_FWD = 16384       # forward rows may be split freely (every sample is written exactly once) ...
SPLIT_FWD_SMALL = 4096  # ... more finely when fewer than SMALL_BATCH images have to fill 256 CUs
SMALL_BATCH = 4
_TABLES = {}
_MAX_TABLES = 8          # device table sets kept per process (oldest evicted)
_CONTENT = {}            # (data_ptr, _version, numel) of a buffer -> (sha1 of its contents, host copy)


def _remember(key, t):
    while len(_TABLES) >= _MAX_TABLES:
        _TABLES.pop(next(iter(_TABLES)))
    _TABLES[key] = t


def _content_of(buf):
    """(sha1, numpy copy) of a small device buffer, read back once per (tensor object, version): a buffer that is rewritten
    with the same values (DDP's buffer broadcast, load_state_dict) costs one device -> host copy per rewrite but maps to
    the same table entry.  The memo holds a WEAK reference to the tensor it read: a different tensor that the allocator
    later places at the same address (a fresh model's depth_weight at version 0) is a different object -- or finds the
    reference dead -- and is read again (the address alone is not an identity)."""
    import hashlib
    import weakref
    ident = (buf.data_ptr(), buf._version, buf.numel(), str(buf.device))
    c = _CONTENT.get(ident)
    if c is not None and c[2]() is not buf:
        c = None
    if c is None:
        host = buf.detach().cpu().numpy().copy()
        c = (hashlib.sha1(host.tobytes()).hexdigest(), host, weakref.ref(buf))
        if len(_CONTENT) >= 64:
            _CONTENT.clear()
        _CONTENT[ident] = c
    return c[0], c[1]


def available():
    return _loader().has_symbol("genre_render_spherical_forward")


def _rows(bricks, q, k, nb, split, shared_mode):
    """sorted (brick, q, k) triples -> (table [rows,4], sample words (q << 8) | k); rows = (brick, begin, end,
    mode) heaviest first, bricks with more than `split` samples cut into several rows"""
    words = ((q << 8) | k).astype(np.uint32).view(np.int32)
    begin = np.searchsorted(bricks, np.arange(nb), side="left")
    end = np.searchsorted(bricks, np.arange(nb), side="right")
    rows = []
    for b in range(nb):
        cnt = int(end[b] - begin[b])
        if cnt <= split:
            rows.append((b, int(begin[b]), int(end[b]), 0))
        else:
            parts = -(-cnt // split)
            size = -(-cnt // parts)
            for s0 in range(int(begin[b]), int(end[b]), size):
                rows.append((b, s0, min(s0 + size, int(end[b])), shared_mode))
    rows.sort(key=lambda r: -(r[2] - r[1]))
    return np.asarray(rows, np.int32).reshape(-1, 4), words


def build_brick_tables(X, Y, Z, dirs64, z_res, split=SPLIT_BWD, split_fwd=SPLIT_FWD):
    """Geometry-only tables of the brick kernels (see include/genre_hip.h for the formats).

    dirs64: float64 [R,R,3] unit directions (spherical_proj.py:43-49).  Sample positions and voxel
    coordinates are computed with exactly the fp64/fp32 operation sequence of the kernels
    (csrc/sph_render.hip: sample_pos, locate), so membership is exact.  Returns a dict of numpy
    int32 arrays:
      bwd_table/bwd_chunks : every sample listed under each brick one of its 8 corners falls in
      fwd_table/fwd_chunks : every in-volume sample listed once, under the brick of its base corner
                             (list entries are (ray << 8) | k, sorted by brick, ray, sample)
      kin                  : per ray, the first sample with a corner inside the volume (the inside
                             samples of a ray are always a suffix: rays end at the centre)"""
    R = dirs64.shape[0]
    assert z_res <= 256 and R * R < (1 << 24)
    d2 = dirs64.reshape(-1, 3).astype(np.float64) * 2.0
    step = 1.0 / (z_res - 1) if z_res > 1 else 0.0
    alpha = np.arange(z_res, dtype=np.float64) * step
    alpha[-1] = 1.0
    a = 1.0 - alpha
    nbx, nby, nbz = -(-X // BRICK), -(-Y // BRICK), -(-Z // BRICK)
    nb = nbx * nby * nbz
    one, two = np.float32(1), np.float32(2)
    axes, base = [], []
    anyin = None
    for ax, size in enumerate((X, Y, Z)):
        g = (d2[:, None, ax] * a[None, :]).astype(np.float32)                # [R*R, ZR]
        i0 = np.floor(((g + one) / two) * np.float32(size - 1)).astype(np.int32)
        inside = (i0 >= -1) & (i0 < size)
        anyin = inside if anyin is None else (anyin & inside)
        b0, v0 = i0 >> 4, i0 >= 0
        b1 = (i0 + 1) >> 4
        v1 = (i0 + 1 <= size - 1) & ((b1 != b0) | ~v0)
        axes.append(((b0, v0), (b1, v1)))
        base.append(np.clip(i0, 0, size - 1) >> 4)
    # inside samples form a suffix of every ray
    kin = np.where(anyin.any(1), anyin.argmax(1), z_res).astype(np.int32)
    assert (anyin == (np.arange(z_res)[None, :] >= kin[:, None])).all(), "inside set is not a suffix"
    sample_id = np.arange(R * R * z_res, dtype=np.int64).reshape(R * R, z_res)

    def finish(keys, split_at, mode):
        keys = np.sort(np.concatenate(keys)) if keys else np.zeros((0,), np.int64)
        sid = keys & 0xFFFFFFFF
        return _rows((keys >> 32).astype(np.int64), sid // z_res, sid % z_res, nb, split_at, mode)

    bkeys = []
    for cx in axes[0]:
        for cy in axes[1]:
            for cz in axes[2]:
                m = anyin & cx[1] & cy[1] & cz[1]
                if m.any():
                    brick = (cx[0][m].astype(np.int64) * nby + cy[0][m]) * nbz + cz[0][m]
                    bkeys.append((brick << 32) | sample_id[m])
    bwd_table, bwd_chunks = finish(bkeys, split, 1)
    fbrick = (base[0][anyin].astype(np.int64) * nby + base[1][anyin]) * nbz + base[2][anyin]
    fwd_table, fwd_chunks = finish([(fbrick << 32) | sample_id[anyin]], split_fwd, 0)
    return dict(bwd_table=bwd_table, bwd_chunks=bwd_chunks, fwd_table=fwd_table, fwd_chunks=fwd_chunks, kin=kin)


def _disk_cached(kind, key, arrays_for_hash, build):
    """geometry tables cost seconds of numpy on first use; they depend on the geometry only, so they are kept under
    $GENRE_TABLE_CACHE (default ~/.cache/genre_shapehd_amd; "0" disables) as .npz, keyed by a hash of every input
    and of the builder's source.  Never fatal: any cache problem falls back to building."""
    import hashlib
    import inspect
    import os
    root = os.environ.get("GENRE_TABLE_CACHE", os.path.join(os.path.expanduser("~"), ".cache", "genre_shapehd_amd"))
    if root == "0":
        return build()
    h = hashlib.sha256(repr(key).encode())
    for a in arrays_for_hash:
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(inspect.getsource(inspect.getmodule(build)).encode())
    path = os.path.join(root, "%s_%s.npz" % (kind, h.hexdigest()[:20]))
    try:
        if os.path.exists(path):
            with np.load(path) as z:
                return {k: z[k] for k in z.files}
    except Exception:
        pass
    t = build()
    try:
        os.makedirs(root, exist_ok=True)
        tmp = "%s.%d.tmp.npz" % (path, os.getpid())
        np.savez(tmp, **t)
        os.replace(tmp, path)
    except Exception:
        pass
    return t


def tables_for(vox_shape, device, dirs64, z_res):
    small = vox_shape[0] * vox_shape[1] < SMALL_BATCH
    key = (tuple(vox_shape[2:]), dirs64.shape[0], z_res, str(device), small)
    t = _TABLES.get(key)
    if t is None:
        d64 = dirs64.cpu().numpy()
        sf = SPLIT_FWD_SMALL if small else SPLIT_FWD
        np_t = _disk_cached("brick", (tuple(vox_shape[2:]), z_res, sf, SPLIT_BWD, BRICK), [d64],
                            lambda: build_brick_tables(vox_shape[2], vox_shape[3], vox_shape[4], d64, z_res, split_fwd=sf))
        t = {k: torch.from_numpy(v).to(device) for k, v in np_t.items()}
        _remember(key, t)
    return t


def seg_tables_for(vox_shape, device, dirs64, depth_weight):
    """tables of the segment forward (toolbox/_seg_tables.py), built on first use and cached per geometry and device; small
    batches get rows of fewer segments (more workgroups for 256 CUs).  Keyed on the CONTENTS of depth_weight (the prefix table
    depends on it), read back once per address + version (_content_of)."""
    from . import _seg_tables
    small = vox_shape[0] * vox_shape[1] < SMALL_BATCH
    dw_hash, dw = _content_of(depth_weight)
    key = ("seg", tuple(vox_shape[2:]), dirs64.shape[0], depth_weight.shape[0], dw_hash, str(device), small)   # (split, max_seg follow `small`)
    t = _TABLES.get(key)
    if t is None:
        d64 = dirs64.cpu().numpy()
        split = _seg_tables.SPLIT_SMALL if small else _seg_tables.SPLIT
        max_seg = _seg_tables.MAX_SEG_SMALL if small else _seg_tables.MAX_SEG
        bwd_split = _seg_tables.BWD_SPLIT_SMALL if small else _seg_tables.BWD_SPLIT

        def build():
            return _seg_tables.build_seg_tables(vox_shape[2], vox_shape[3], vox_shape[4], d64, dw.shape[0], dw, max_seg=max_seg,
                                                split=split, bwd_split=bwd_split)
        build.__module__ = _seg_tables.__name__
        np_t = _disk_cached("seg", (tuple(vox_shape[2:]), split, max_seg, _seg_tables.BRICK, bwd_split, "r6g"), [d64, dw], build)
        t = {"smax": int(np_t["smax"][0])}
        for k, v in np_t.items():
            if k == "smax":
                continue
            tv = torch.from_numpy(np.ascontiguousarray(v))
            if k == "ray_pre":
                tv = tv.view(torch.float32).reshape(-1, 4)
            t[k] = tv.to(device)
        _remember(key, t)
    return t


def seg_tr_scratch(ps, vox, dirs64):
    """tr_scratch of render_seg_backward: a (g T, w - R) pair per scratch line like ps, and behind them one word per image and
    block of the per-ray kernel (at most ceil(R*R / 64): include/genre_hip.h)"""
    imgs, rr = vox.shape[0] * vox.shape[1], dirs64.shape[0] * dirs64.shape[0]
    return torch.empty((ps.numel() + imgs * (-(-rr // 64)),), dtype=torch.float32, device=ps.device)


def seg_v_scratch(ts, imgs, device):
    """v_scratch of render_seg_forward / render_seg_backward: 16 floats per image and segment, segments in groups of 64"""
    return torch.empty((imgs * (-(-ts["segs"].shape[0] // 64)) * 64 * 16,), dtype=torch.float32, device=device)


def seg_halo_scratch(ts, vox):
    """halo_scratch of render_seg_backward: 832 floats per image and row of bwd_rows"""
    return torch.empty((vox.shape[0] * vox.shape[1] * ts["bwd_rows"].shape[0] * 832,), dtype=torch.float32, device=vox.device)


def attach_zero_hint(grad, words, stride, offset, group):
    """hangs "the gradient of these images is identically zero" on a gradient tensor this module just wrote: int32 `words`,
    word [(image // group) * stride + offset] == 0 <=> all zeros -- the renderer's backward knows it from the forward's clamp
    words (on GenRe's own chain the x50 clamp blocks every voxel of every image).  The consumer (the camera layer's backward,
    cam_back_projection.py) then writes zeros for those images without reading anything.  Guarded like the occupancy hint: the
    tensor's version counter (ATen writes) and _loader._call (raw C-ABI writes drop it)."""
    grad._genre_zero_hint = (words, int(stride), int(offset), int(group), grad._version)
    return grad


def zero_hint_of(grad):
    """(words, stride, offset, group) a producer hung on this gradient tensor, or None"""
    h = getattr(grad, "_genre_zero_hint", None)
    if h is None or h[4] != grad._version or h[0].device != grad.device:
        return None
    return h[:4]


def _bm_tables_module():
    from . import _bm_tables
    return _bm_tables


def bm_tables_for(vox_shape, device, dirs64, depth_weight):
    """tables of the batch-minor tile renderer (toolbox/_bm_tables.py), built on first use and cached per geometry and
    device; the float64 per-ray prefix table travels as its fp32 words"""
    import os
    from . import _bm_tables
    pull = {"488": (4, 8, 8), "888": (8, 8, 8)}[os.environ.get("GENRE_BM_PULL", "488")]      # backward brick (A/B switch)
    # keyed on the CONTENTS of the depth_weight buffer (hashed once per address + version, _content_of): no device ->
    # host copy -- and no stream synchronisation, which HIP-graph capture forbids -- while the buffer is untouched
    lib_brick = _loader()._lib.genre_bm_brick()
    assert lib_brick == _bm_tables.BX * 100 + _bm_tables.BY * 10 + _bm_tables.BZ, \
        "libgenre_hip.so was built for %d bricks, toolbox/_bm_tables.py builds %dx%dx%d" % (lib_brick, _bm_tables.BX, _bm_tables.BY, _bm_tables.BZ)
    dw_hash, dw = _content_of(depth_weight)
    key = ("bm", tuple(vox_shape[2:]), dirs64.shape[0], depth_weight.shape[0], dw_hash, str(device), pull)
    t = _TABLES.get(key)
    if t is None:
        d64 = dirs64.cpu().numpy()

        def build():
            return _bm_tables.build_bm_tables(vox_shape[2], vox_shape[3], vox_shape[4], d64, dw.shape[0], dw, pull=pull)
        build.__module__ = _bm_tables.__name__
        np_t = _disk_cached("bm", (tuple(vox_shape[2:]), pull, _bm_tables.ROW_ORDER, _bm_tables.SPLIT_F,
                                   _bm_tables.SPLIT_B, _bm_tables.MAXSEG, (_bm_tables.BX, _bm_tables.BY, _bm_tables.BZ), "r6b"), [d64, dw], build)
        t = {"pull_code": int(np_t["pull"][0]) * 100 + int(np_t["pull"][1]) * 10 + int(np_t["pull"][2])}
        for k, v in np_t.items():
            if k == "pull":
                continue
            tv = torch.from_numpy(np.ascontiguousarray(v))
            if k == "ray_pre":
                tv = tv.view(torch.float32).reshape(-1, 4)
            t[k] = tv.to(device)
        _remember(key, t)
    return t


# ---- occupancy hint: producer (Camera_back_projection_layer, image-minor volumes) -> consumer (the batch-minor forward) --------
# The camera forward's leader pass knows which 4x8x8-voxel bricks of which group of 32 images received a point; everything else
# holds its fill value.  It writes one word per (group, brick) -- set for a brick whose TILE (the brick plus the voxels one step
# beyond its high faces: what the sampler stages) can reach a point; include/genre_hip.h: tile_live -- and the layer hangs them on
# the tensor it returns, with the tensor's version counter.  The renderer uses them only if the SAME tensor object arrives with
# the same version (any in-place write, any other tensor: the hint is ignored and every tile is read), so nothing can go stale.
_GROUP = 32                         # images per group = lanes of a half-wave (csrc/sph_render_bm.hip: kImgs)


def _bm_brick():
    code = _loader()._lib.genre_bm_brick()
    return code // 100, (code // 10) % 10, code % 10


def new_brick_words(n, res, device):
    bx, by, bz = _bm_brick()
    return torch.empty((-(-n // _GROUP), -(-res // bx), -(-res // by), -(-res // bz)), dtype=torch.int32, device=device)


def shifted_fill(res):
    """csrc/cam_bp.hip: fill_val of the shifted op, 1 - res * (1 / res) in fp32"""
    return float(np.float32(1) - np.float32(res) * np.float32(1.0 / res))


def attach_hint(vol, words, res, cell=None):
    """hang the producer's occupancy words on the volume it returns.  cell = None: the leader pass's words per group of 32 images
    and 4x8x8-voxel brick (image-minor volumes -> csrc/sph_render_bm.hip); cell = cx*10000 + cy*100 + cz: the brick kernel's
    words per image and cx x cy x cz-voxel cell (dense NCXYZ volumes -> csrc/sph_render_seg.hip).  Tensors of
    torch.inference_mode() carry no version counter, i.e. nothing a later in-place write would invalidate the hint with:
    they get none."""
    if vol.is_inference():
        return vol
    if cell is None:
        vol._genre_brick_hint = (words, shifted_fill(res), vol._version)
    else:
        vol._genre_cell_hint = (words, shifted_fill(res), vol._version, int(cell))
    # ... and the producer's VALUE RANGE: an empty voxel holds the fill value, an occupied one 1 - res * tdf with tdf the mean
    # distance of the points inside the voxel from its centre (camera_backprojection_module.py:25-28, back_projection_kernel.cu
    # :215-305) -- at most half the voxel's diagonal, sqrt(3)/2 / res: occupied voxels hold >= 0.1339 (0.13 here, generously)
    vol._genre_range_hint = (shifted_fill(res), _VMIN_SHIFTED, vol._version)
    return vol


_VMIN_SHIFTED = 0.13
_LAZY_ZEROS = {}


def provably_blocked(vox, pre_scale):
    """True when the HOST can tell that clamp(vox * pre_scale, lo, hi) blocks every voxel: the producer's value range hangs on the
    volume (attach_hint: fill value, lower bound of every other value), the volume still is what the producer wrote, and both ends
    of the range land outside the clamp -- GenRe's own chain, depth_pred_with_sph_inpaint.py:120-126: clamp(proj * 50) of values in
    {0} u [0.13, 1].  The gradient of render_spherical w.r.t. such a volume is identically zero whatever comes from above (the
    reference computes exactly these zeros): the forward then saves nothing for a backward and the backward launches nothing
    (lazy_zero_grad).  GENRE_LAZY_ZERO_GRAD=0 switches it off (the kernels then find the same zeros themselves)."""
    import os
    if not pre_scale or os.environ.get("GENRE_LAZY_ZERO_GRAD", "1") == "0":
        return False
    h = _live_hint(vox, "_genre_range_hint")
    if h is None:
        return False
    fill, vmin = float(np.float32(h[0]) * np.float32(pre_scale)), float(np.float32(h[1]) * np.float32(pre_scale))
    return pre_scale > 0 and (fill < _LO or fill > _HI) and vmin > _HI


def lazy_zero_grad(shape, device):
    """the gradient of a provably blocked volume: zeros of the volume's shape that occupy four bytes (a stride-0 view of one
    zero -- every reader sees zeros; nothing is written per step) with the zero-gradient words of attach_zero_hint saying so for
    every image (one word, group = everything), so that the camera layer's backward does not even read it"""
    z = _LAZY_ZEROS.get(str(device))
    if z is None:
        z = (torch.zeros((1,), dtype=torch.float32, device=device), torch.zeros((1,), dtype=torch.int32, device=device))
        _LAZY_ZEROS[str(device)] = z
    g = z[0].expand(tuple(shape))
    return attach_zero_hint(g, z[1], 0, 0, 1 << 30)


_LO, _HI = float(np.float32(1e-5)), float(np.float32(1 - 1e-5))            # spherical_proj.py:66
_WARNED = set()


def _fill_passes_clamp(fill, pre_scale):
    """does clamp(fill * pre_scale, lo, hi) pass the gradient (lo <= fill * pre_scale <= hi, fp32)?"""
    raw = float(np.float32(fill) * np.float32(pre_scale))
    return _LO <= raw <= _HI


def _hint_usable(fill, pre_scale, with_grad):
    """A dead tile's saved state is not written (csrc/sph_render_bm.hip) / its clamp pass words stay 0 (sph_render_seg.hip):
    only right when no gradient can come back through it -- pre_scale folded in AND the fill value blocked by its clamp
    (GenRe: fill 0 at pre_scale 50).  Without a gradient the constants are all that matters."""
    return not with_grad or (pre_scale != 0.0 and not _fill_passes_clamp(fill, pre_scale))


def _ps_empty_slot(t, key):
    """a handful of (fill, pre_scale) constant tables per geometry, never evicted (captured HIP graphs hold raw pointers into
    them); the fifth pair is refused -- loudly, once: every tile is read from then on"""
    if key in t:
        return True
    if torch.cuda.is_current_stream_capturing():
        return False                        # (built outside graph capture only: callers warm up before they capture)
    if sum(1 for k in t if isinstance(k, tuple) and k[:1] == key[:1]) >= 4:
        if key[:1] not in _WARNED:
            import warnings
            _WARNED.add(key[:1])
            warnings.warn("render_spherical: more than four (fill, pre_scale) pairs on one geometry -- the occupancy hint is "
                          "ignored for the new ones (every tile is read)")
        return False
    return True


def _live_hint(vox, name):
    """the hint `name` of vox if the tensor still is what the producer wrote: same object, same version counter (any ATen
    in-place write bumps it; a raw write through the C ABI drops the attribute: _loader._call)"""
    hint = getattr(vox, name, None)
    if hint is None or vox.is_inference() or hint[2] != vox._version:
        return None
    return hint


def occupancy_hint(vox, t, pre_scale, lib, with_grad=False):
    """(tile_live, ps_empty) for render_bm_forward, or (None, None): the words the producer hung on `vox` -- if it still is what the
    producer wrote -- and the geometry's (P, S) constants on the constant volume, built on first use by rendering one"""
    hint = _live_hint(vox, "_genre_brick_hint")
    if hint is None or not _hint_usable(hint[1], float(pre_scale), with_grad):
        return None, None
    words, fill = hint[0], hint[1]
    bx, by, bz = _bm_brick()
    n, _, X, Y, Z = vox.shape
    if tuple(words.shape) != (-(-n // _GROUP), -(-X // bx), -(-Y // by), -(-Z // bz)) or words.device != vox.device:
        return None, None
    key = ("ps_empty", fill, float(pre_scale))
    if not _ps_empty_slot(t, key):
        return None, None
    if key not in t:
        const = empty_batch_minor((_GROUP, 1, X, Y, Z), torch.float32, vox.device).fill_(fill)
        res_map = int(round((t["ray_ptr"].shape[0] - 1) ** 0.5))
        out = torch.empty((_GROUP, 1, res_map, res_map), dtype=torch.float32, device=vox.device)
        ps = torch.empty((t["segs"].shape[0] * 2 * _GROUP,), dtype=torch.float32, device=vox.device)
        lib.render_bm_forward(const, out, t["segs"], t["rec_f"], t["fwd_rows"], t["ray_ptr"], t["ray_seg"], t["ray_pre"], ps,
                              None, None, float(pre_scale))
        line = t["segs"][:, 0].long()                                     # scratch line (ray order) of every segment, table order
        pe = torch.zeros((line.shape[0], 4), dtype=torch.int32, device=vox.device)
        pe.view(torch.float32)[:, :2] = ps.view(-1, 2, _GROUP)[:, :, 0][line]
        pe[:, 2] = t["segs"][:, 0]
        t[key] = pe.view(torch.float32)                                    # [nseg, 4] = (P, S, line bits, 0), table order
    return words, t[key]


def occupancy_hint_std(vox, t, dirs64, depth_weight, pre_scale, lib, with_grad=False):
    """(occ, ps_empty, occ_cell) for render_seg_forward, or (None, None, 0): the per-image cell words the camera forward's brick
    kernel hung on the dense volume `vox` -- if it still is what that op wrote -- and the (P, S) pair of every segment on the
    constant volume, built on first use by rendering one (the sampler's own output: bit-identical to what the march computes)"""
    hint = _live_hint(vox, "_genre_cell_hint")
    if hint is None or not _hint_usable(hint[1], float(pre_scale), with_grad):
        return None, None, 0
    words, fill, _, cell = hint
    n, nc, X, Y, Z = vox.shape
    cx, cy, cz = cell // 10000, (cell // 100) % 100, cell % 100
    if tuple(words.shape) != (n * nc, -(-X // cx), -(-Y // cy), -(-Z // cz)) or words.device != vox.device:
        return None, None, 0
    key = ("ps_empty_std", fill, float(pre_scale))
    if not _ps_empty_slot(t, key):
        return None, None, 0
    if key not in t:
        res = dirs64.shape[0]
        const = torch.full((1, 1, X, Y, Z), fill, dtype=torch.float32, device=vox.device)
        out = torch.empty((1, 1, res, res), dtype=torch.float32, device=vox.device)
        ps = torch.empty((t["smax"] * res * res * 2,), dtype=torch.float32, device=vox.device)
        # (with a buffer for sample values: the variant whose products are rounded once per segment, as the backward wants them)
        lib.render_seg_forward(const, dirs64.view(torch.float32), depth_weight, out, t["seg_rows"], t["segs"], t["ray_nseg"],
                               t["ray_pre"], t["line_w"], ps, float(pre_scale),
                               live=(torch.empty((1 + (-(-X // 16)) * (-(-Y // 16)) * (-(-Z // 16)),), dtype=torch.int32,
                                                 device=vox.device) if pre_scale else None),
                               v_scratch=seg_v_scratch(t, 1, vox.device))
        t[key] = ps.view(-1, 2)[t["segs"][:, 2].long()].contiguous()            # [nseg, 2], table order
    return words, t[key], cell


def is_batch_minor(vox):
    """image index fastest in memory (the layout the batch-minor kernels of csrc/sph_render.hip want)"""
    return vox.dim() == 5 and vox.shape[1] == 1 and vox.shape[0] >= 16 and vox.stride(0) == 1


def empty_batch_minor(shape, dtype, device):
    """uninitialised [N,1,X,Y,Z] tensor whose memory order is (X, Y, Z, N)"""
    n, c, x, y, z = shape
    assert c == 1
    return torch.empty_strided((n, c, x, y, z), (1, n * x * y * z, y * z * n, z * n, n), dtype=dtype, device=device)


class RenderSphericalFused(Function):
    """apply(vox [N,NC,X,Y,Z], dirs64 [R,R,3] float64, depth_weight [ZR], pre_scale=0.0, pad=0) -> [N,NC,R+2*pad,R+2*pad];
    pre_scale != 0 renders clamp(vox * pre_scale, 1e-5, 1 - 1e-5) without materialising it; pad > 0 lays the map
    out as sph_pad(map, pad) (spherical_proj.py:21-28) straight from the scan kernel"""

    @staticmethod
    def forward(ctx, vox, dirs64, depth_weight, pre_scale=0.0, pad=0):
        assert vox.dim() == 5 and vox.is_cuda and vox.dtype == torch.float32
        assert dirs64.dtype == torch.float64 and dirs64.dim() == 3 and dirs64.is_contiguous()
        lib = _loader().render_lib
        res, z_res = dirs64.shape[0], depth_weight.shape[0]
        pad = int(pad)
        assert 0 <= 2 * pad <= res
        out = torch.empty((vox.shape[0], vox.shape[1], res + 2 * pad, res + 2 * pad), dtype=vox.dtype, device=vox.device)
        ctx.pre_scale = float(pre_scale)
        ctx.batch_minor = is_batch_minor(vox)
        # a volume whose every voxel the folded clamp provably blocks (GenRe's own chain): zero gradient, nothing to save
        ctx.blocked = bool(ctx.needs_input_grad[0]) and provably_blocked(vox, ctx.pre_scale)
        want_grad = bool(ctx.needs_input_grad[0]) and not ctx.blocked
        ctx.vox_shape = vox.shape
        if ctx.batch_minor:
            # image index fastest in memory: tile renderer, lanes = images (csrc/sph_render_bm.hip)
            t = bm_tables_for(vox.shape, vox.device, dirs64, depth_weight)
            groups = -(-vox.shape[0] // 32)
            f32 = dict(dtype=torch.float32, device=vox.device)
            ps = torch.empty((groups * t["segs"].shape[0] * 64,), **f32)
            stash = mask = None
            if want_grad:
                stash = torch.empty((groups * t["rec_f"].shape[0] * 32,), **f32)
                if pre_scale:
                    mask = torch.empty((groups * vox.shape[2] * vox.shape[3] * vox.shape[4] + groups,), dtype=torch.int32,
                                       device=vox.device)
            words, ps_empty = occupancy_hint(vox, t, ctx.pre_scale, lib, with_grad=stash is not None)
            lib.render_bm_forward(vox, out, t["segs"], t["rec_f"], t["fwd_rows"], t["ray_ptr"], t["ray_seg"],
                                  t["ray_pre"], ps, stash, mask, ctx.pre_scale, words, ps_empty)
            ctx.mask = mask
            if want_grad:
                ctx.save_for_backward(dirs64, depth_weight, ps, stash)
            return out
        # standard layout: the segment forward (csrc/sph_render_seg.hip) -- one (P, S) pair per segment, nothing saved
        t = seg_tables_for(vox.shape, vox.device, dirs64, depth_weight)
        imgs = vox.shape[0] * vox.shape[1]
        ps = torch.empty((imgs * t["smax"] * res * res * 2,), dtype=torch.float32, device=vox.device)
        # with pre_scale and a backward to come: the clamp's pass words per image and per 16^3 brick -- what the clamp blocks
        # is then written as zeros by the backward, not computed (GenRe's own volumes: everything)
        ctx.live = None
        if pre_scale and want_grad:
            nb = -(-vox.shape[2] // BRICK) * -(-vox.shape[3] // BRICK) * -(-vox.shape[4] // BRICK)
            ctx.live = torch.empty((imgs * (1 + nb),), dtype=torch.int32, device=vox.device)
        occ, ps_empty, cell = occupancy_hint_std(vox, t, dirs64, depth_weight, ctx.pre_scale, lib, with_grad=want_grad)
        # a backward will follow: the raw sample values of the tiles a gradient can come back through -- with the (P, S) pairs
        # all the state the backward's segment form needs
        v = seg_v_scratch(t, imgs, vox.device) if want_grad else None
        lib.render_seg_forward(vox, dirs64.view(torch.float32), depth_weight, out, t["seg_rows"], t["segs"], t["ray_nseg"],
                               t["ray_pre"], t["line_w"], ps, ctx.pre_scale, ctx.live, occ, ps_empty, cell, v)
        if v is not None:
            ctx.save_for_backward(vox, dirs64, depth_weight, ps, v)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        lib = _loader().render_lib
        if ctx.blocked:                  # the clamp provably blocks every voxel (provably_blocked): nothing to launch
            return lazy_zero_grad(ctx.vox_shape, grad_out.device), None, None, None, None
        if ctx.batch_minor:
            dirs64, depth_weight, ps, stash = ctx.saved_tensors
            t = bm_tables_for(ctx.vox_shape, grad_out.device, dirs64, depth_weight)
            grad_vox = empty_batch_minor(ctx.vox_shape, grad_out.dtype, grad_out.device)
            groups = -(-ctx.vox_shape[0] // 32)
            lib.render_bm_backward(grad_out, grad_vox, t["segs"], t["ray_ptr"], t["ray_seg"], t["ray_pre"], t["ent"],
                                   t["rec_b"], t["bwd_rows"], depth_weight, ps, torch.empty_like(ps), stash, ctx.mask,
                                   ctx.pre_scale, t["pull_code"])
            if ctx.mask is not None and ctx.pre_scale:
                # the trailing word of a group of 32 images: does any of its voxels pass the clamp?
                nvox = ctx.vox_shape[2] * ctx.vox_shape[3] * ctx.vox_shape[4]
                attach_zero_hint(grad_vox, ctx.mask, 1, groups * nvox, 32)
            return grad_vox, None, None, None, None
        vox, dirs64, depth_weight, ps, v = ctx.saved_tensors
        ts = seg_tables_for(vox.shape, vox.device, dirs64, depth_weight)
        grad_vox = torch.empty(vox.shape, dtype=vox.dtype, device=vox.device)
        # per-ray chains over the (P, S) pairs, then one pass over the segments: dL/dp from the saved values, scattered into the
        # bricks' tiles (csrc/sph_render_seg.hip)
        lib.render_seg_backward(vox, dirs64.view(torch.float32), depth_weight, grad_out, grad_vox, ts["bwd_rows"], ts["segs"],
                                ts["ray_nseg"], ts["ray_pre"], ts["line_w"], ps, seg_tr_scratch(ps, vox, dirs64), v,
                                seg_halo_scratch(ts, vox), ctx.pre_scale, ctx.live)
        if ctx.live is not None and ctx.pre_scale and vox.shape[1] == 1:
            # word 0 of an image's live words: does any of its voxels pass the clamp?
            nb = (-(-vox.shape[2] // 16)) * (-(-vox.shape[3] // 16)) * (-(-vox.shape[4] // 16))
            attach_zero_hint(grad_vox, ctx.live, 1 + nb, 0, 1)
        return grad_vox, None, None, None, None
