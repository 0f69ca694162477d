"""Plain-numpy walk through the tables of genre-shapehd_amd/toolbox/_bm_tables.py -- TEST INFRASTRUCTURE.

Executes, one image at a time and in float64, exactly the data flow of csrc/sph_render_bm.hip (brick tiles, serial
segment marches, per-ray combination, brick-owned pull scatter), so that the tables and the algebra can be checked
against the oracle on the CPU, without a GPU."""
import numpy as np


def _tables(mod, X, Y, Z, dirs64, z_res, dw, **kw):
    return mod.build_bm_tables(X, Y, Z, dirs64, z_res, dw, **kw)


def forward(mod, t, vox, pre_scale=0.0, lo=np.float32(1e-5), hi=np.float32(1 - 1e-5)):
    """vox [X,Y,Z] float32 -> (map [RR], per-segment (P,S), per-slot stash of sign-coded p, mask [X,Y,Z] bool)"""
    X, Y, Z = vox.shape
    BX, BY, BZ, TX, TY, TZ = mod.BX, mod.BY, mod.BZ, mod.TX, mod.TY, mod.TZ
    nby, nbz = -(-Y // BY), -(-Z // BZ)
    segs, rec = t["segs"], t["rec_f"]
    w = rec[:, 4:12].view(np.float32).astype(np.float64)
    dwk = rec[:, 1].view(np.float32).astype(np.float64)
    PS = np.zeros((segs.shape[0], 2))
    stash = np.zeros(rec.shape[0])
    mask = np.zeros(vox.shape, bool)
    corner = np.array([(c & 1) * TY * TZ + ((c >> 1) & 1) * TZ + (c >> 2) for c in range(8)])
    for brick, s0, s1, flag in t["fwd_rows"]:
        if flag == mod.SKIP:
            continue
        ox, oy, oz = (brick // (nby * nbz)) * BX, ((brick // nbz) % nby) * BY, (brick % nbz) * BZ
        tile = np.zeros((TX, TY, TZ), np.float32)
        sub = vox[ox:ox + TX, oy:oy + TY, oz:oz + TZ]
        if pre_scale != 0.0:
            raw = sub * np.float32(pre_scale)
            own = raw[:BX, :BY, :BZ]
            mask[ox:ox + BX, oy:oy + BY, oz:oz + BZ][:own.shape[0], :own.shape[1], :own.shape[2]] = (own >= lo) & (own <= hi)
            sub = np.clip(raw, lo, hi)
        tile[:sub.shape[0], :sub.shape[1], :sub.shape[2]] = sub
        flat = tile.reshape(-1).astype(np.float64)
        for s in range(s0, s1):
            q, k0, L, slot0 = segs[s]
            T, S = 1.0, 0.0
            for i in range(L):
                r = slot0 + i
                line = rec[r, 0] // mod.LINE_F
                v = np.float32((flat[line + corner] * w[r]).sum())
                p = float(min(max(v, lo), hi))
                stash[r] = p if (v >= lo and v <= hi) else -p
                S += T * p * dwk[r]
                T *= 1.0 - p
            assert t["ray_seg"][q] == s                                 # q = the segment's line = its position in ray order
            PS[q] = (T, S)
    rr = t["ray_ptr"].shape[0] - 1
    out = np.zeros(rr)
    for q in range(rr):
        T, S = t["ray_pre"][q]
        for j in range(t["ray_ptr"][q], t["ray_ptr"][q + 1]):           # a ray's segments are neighbours in the scratch buffer
            S += T * PS[j, 1]
            T *= PS[j, 0]
        out[q] = S + T
    return out, PS, stash, mask


def backward(mod, t, shape, PS, stash, mask, g, dw, pre_scale=0.0):
    """g [RR] upstream gradient -> grad_vox [X,Y,Z] (float64)"""
    X, Y, Z = shape
    BX, BY, BZ = (int(v) for v in t["pull"])                           # the backward's (pull) bricks
    nby, nbz = -(-Y // BY), -(-Z // BZ)
    segs = t["segs"]
    TR = np.zeros((segs.shape[0], 2))
    rr = t["ray_ptr"].shape[0] - 1
    for q in range(rr):
        ids = range(t["ray_ptr"][q], t["ray_ptr"][q + 1])
        T = t["ray_pre"][q][0]
        for s in ids:
            TR[s, 0] = g[q] * T
            T *= PS[s, 0]
        Rr = 1.0
        for s in ids[::-1]:
            TR[s, 1] = Rr
            Rr = PS[s, 1] + PS[s, 0] * Rr
    grad = np.full(shape, np.nan)
    written = np.zeros(shape, np.int32)
    recb = t["rec_b"]
    wb = recb[:, 4:12].view(np.float32).astype(np.float64)
    corner = np.array([(c & 1) * BY * BZ + ((c >> 1) & 1) * BZ + (c >> 2) for c in range(8)])
    acc_shared = {}
    for brick, e0, e1, shared in t["bwd_rows"]:
        if shared == mod.SKIP:
            continue
        tile = np.zeros(BX * BY * BZ)
        for e in range(e0, e1):
            s, slot0, pk, rs = t["ent"][e]                               # s: the segment's scratch line (ray order)
            i0, i1, L, k0 = pk & 63, (pk >> 6) & 63, (pk >> 12) & 63, (pk >> 18) & 255
            sf = t["ray_seg"][s]
            assert (segs[sf][0], segs[sf][1], segs[sf][2], segs[sf][3]) == (s, k0, L, slot0)
            p = stash[slot0:slot0 + L]
            Tg, Rr = TR[s]
            c = np.zeros(L)
            for i in range(L):
                c[i] = Tg if p[i] > 0 else 0.0
                Tg *= 1.0 - abs(p[i])
            for i in range(L - 1, i0 - 1, -1):
                wk = float(dw[k0 + i])
                dp = c[i] * (wk - Rr)
                Rr = Rr + abs(p[i]) * (wk - Rr)
                if i < i1:
                    r = rs + (i - i0)
                    line = recb[r, 0] // mod.LINE_B
                    own = recb[r, 1]
                    for cc in range(8):
                        if (own >> ((cc & 3) + 4 * (cc >> 2))) & 1:
                            idx = line + corner[cc]
                            assert 0 <= idx < tile.shape[0]
                            tile[idx] += wb[r, cc] * dp
        if shared:
            acc_shared[brick] = acc_shared.get(brick, 0) + tile
            continue
        _flush(grad, written, tile, brick, (BX, BY, BZ), nby, nbz, mask, pre_scale)
    for brick, tile in acc_shared.items():
        _flush(grad, written, tile, brick, (BX, BY, BZ), nby, nbz, mask, pre_scale)
    assert (written == 1).all(), "every voxel must be written exactly once"
    return grad


def _flush(grad, written, tile, brick, dims, nby, nbz, mask, pre_scale):
    BX, BY, BZ = dims
    ox, oy, oz = (brick // (nby * nbz)) * BX, ((brick // nbz) % nby) * BY, (brick % nbz) * BZ
    tl = tile.reshape(BX, BY, BZ)
    view = grad[ox:ox + BX, oy:oy + BY, oz:oz + BZ]
    sx, sy, sz = view.shape
    val = tl[:sx, :sy, :sz]
    if pre_scale != 0.0:
        val = np.where(mask[ox:ox + BX, oy:oy + BY, oz:oz + BZ], val * pre_scale, 0.0)
    view[...] = val
    written[ox:ox + BX, oy:oy + BY, oz:oz + BZ] += 1


def backward_gather(mod, t, shape, PS, stash, mask, g, dw, pre_scale=0.0):
    """the gather form of the backward (bm_gather_kernel): per chunk the listed samples' dL/dp are parked in a sample
    buffer, then every voxel of the brick sums its list of (sample line, weight) contributions.  g [RR] -> grad_vox"""
    X, Y, Z = shape
    BX, BY, BZ = mod.GATHER_BRICK
    nby, nbz = -(-Y // BY), -(-Z // BZ)
    segs = t["segs"]
    TR = np.zeros((segs.shape[0], 2))
    rr = t["ray_ptr"].shape[0] - 1
    for q in range(rr):
        ids = range(t["ray_ptr"][q], t["ray_ptr"][q + 1])
        T = t["ray_pre"][q][0]
        for s in ids:
            TR[s, 0] = g[q] * T
            T *= PS[s, 0]
        Rr = 1.0
        for s in ids[::-1]:
            TR[s, 1] = Rr
            Rr = PS[s, 1] + PS[s, 0] * Rr
    grad = np.full(shape, np.nan)
    written = np.zeros(shape, np.int32)
    blob = t["g_blob"]
    vi = np.arange(BX * BY * BZ)
    lx, ly, lz = vi // (BY * BZ), (vi // BZ) % BY, vi % BZ
    hidx = (((ly + 2 * (lz >> 1) + 4 * lx) & 7) * 2 + (lz & 1)) * 16 + lx * 4 + (lz >> 1)      # header slot of voxel vi
    acc_shared = {}
    for brick, c0, c1, shared in t["g_rows"]:
        if shared == mod.SKIP:
            continue
        tile = np.zeros(BX * BY * BZ)
        assert c1 > c0, "every row has at least one chunk (a brick that nothing touches: one without entries)"
        for c in range(c0, c1):
            e0, e1, b0, nw = t["g_chunks"][c]
            sbuf = np.full(mod.GATHER_CH, np.nan)                       # a line that no entry fills must never be read
            Ls = [(t["g_ent"][e][2] >> 12) & 63 for e in range(e0, e1)]
            assert Ls == sorted(Ls, reverse=True), "entries of a chunk: longest first"
            for e in range(e0, e1):
                s, slot0, pk, ls0 = t["g_ent"][e]
                i0, i1, L, k0 = pk & 63, (pk >> 6) & 63, (pk >> 12) & 63, (pk >> 18) & 255
                sf = t["ray_seg"][s]
                assert (segs[sf][0], segs[sf][1], segs[sf][2], segs[sf][3]) == (s, k0, L, slot0)
                p = stash[slot0:slot0 + L]
                Tg, Rr = TR[s]
                cT = np.zeros(L)
                for i in range(L):
                    cT[i] = Tg
                    Tg *= 1.0 - abs(p[i])
                for i in range(L - 1, -1, -1):
                    d = float(dw[k0 + i]) - Rr
                    Rr = Rr + abs(p[i]) * d
                    if i0 <= i < i1:
                        assert np.isnan(sbuf[ls0 + i - i0]), "two samples share a line of the sample buffer"
                        sbuf[ls0 + i - i0] = cT[i] * d if p[i] > 0 else 0.0
            hdr = blob[b0:b0 + 256]
            lists = blob[b0 + 256:b0 + nw].reshape(-1, 2)
            assert nw == 256 + 2 * int((hdr >> 16).sum()) and (hdr >> 16).sum() <= mod.GATHER_LCAP
            for v in range(BX * BY * BZ):
                h, hp = hdr[hidx[v]], hdr[hidx[v ^ 1]]
                st, n = h & 0xFFFF, h >> 16
                assert n == hp >> 16 and n % 4 == 0 and st % 4 == 0       # the wave's two lists: one length, a multiple of 4
                sub = lists[st:st + n]
                off, w = sub[:, 0], sub[:, 1].view(np.float32).astype(np.float64)
                assert (off % 128 == 0).all()
                vals = sbuf[off // 128]
                vals = np.where(w == 0.0, 0.0, vals)                    # padding reads line 0, weight 0
                assert not np.isnan(vals).any()
                tile[v] += (w * vals).sum()
        if shared:
            acc_shared[brick] = acc_shared.get(brick, 0) + tile
            continue
        _flush(grad, written, tile, brick, (BX, BY, BZ), nby, nbz, mask, pre_scale)
    for brick, tile in acc_shared.items():
        _flush(grad, written, tile, brick, (BX, BY, BZ), nby, nbz, mask, pre_scale)
    assert (written == 1).all(), "every voxel must be written exactly once"
    return grad


def halo_index(mod, tx, ty, tz):
    """csrc/sph_render_bm.hip: halo_index -- line (tx, ty, tz) of a tile that lies outside its brick -> 0 .. 148"""
    if tx == mod.BX:
        return ty * mod.TZ + tz
    if ty == mod.BY:
        return mod.TY * mod.TZ + tx * mod.TZ + tz
    return mod.TY * mod.TZ + mod.BX * mod.TZ + tx * mod.BY + ty


def backward_halo(mod, t, shape, PS, stash, mask, g, dw, pre_scale=0.0):
    """the halo ("owner computes") form of the backward: every brick scatters its OWN segments (h_ent over the forward's
    rec_f) into a tile with halo, writes its brick, leaves the 149 halo lines in a scratch buffer, and a second pass adds the
    <= 7 neighbours' halo lines onto each brick's low faces -- the data flow of bm_scatter_kernel<HALO> and
    bm_halo_combine_kernel, indices included.  g [RR] -> grad_vox [X,Y,Z] (float64)"""
    X, Y, Z = shape
    BX, BY, BZ, TX, TY, TZ = mod.BX, mod.BY, mod.BZ, mod.TX, mod.TY, mod.TZ
    nbx, nby, nbz = -(-X // BX), -(-Y // BY), -(-Z // BZ)
    segs = t["segs"]
    TR = np.zeros((segs.shape[0], 2))
    rr = t["ray_ptr"].shape[0] - 1
    for q in range(rr):
        ids = range(t["ray_ptr"][q], t["ray_ptr"][q + 1])
        T = t["ray_pre"][q][0]
        for s in ids:
            TR[s, 0] = g[q] * T
            T *= PS[s, 0]
        Rr = 1.0
        for s in ids[::-1]:
            TR[s, 1] = Rr
            Rr = PS[s, 1] + PS[s, 0] * Rr
    rec = t["rec_f"]
    w = rec[:, 4:12].view(np.float32).astype(np.float64)
    corner = np.array([(c & 1) * TY * TZ + ((c >> 1) & 1) * TZ + (c >> 2) for c in range(8)])
    nb = nbx * nby * nbz
    tiles = np.zeros((nb, TX * TY * TZ))
    seen = np.zeros(nb, np.int32)
    for brick, e0, e1, shared in t["h_rows"]:
        if shared == mod.SKIP:
            continue
        seen[brick] += 1
        live = t["h_rows"][t["h_rows"][:, 3] != mod.SKIP]
        assert shared == (1 if (live[:, 0] == brick).sum() > 1 else 0)
        for e in range(e0, e1):
            s, slot0, pk, rs = t["h_ent"][e]
            i0, i1, L, k0 = pk & 63, (pk >> 6) & 63, (pk >> 12) & 63, (pk >> 18) & 255
            assert i0 == 0 and i1 == L and rs == slot0 and tuple(segs[e]) == (s, k0, L, slot0)
            p = stash[slot0:slot0 + L]
            Tg, Rr = TR[s]
            c = np.zeros(L)
            for i in range(L):
                c[i] = Tg if p[i] > 0 else 0.0
                Tg *= 1.0 - abs(p[i])
            for i in range(L - 1, -1, -1):
                wk = float(dw[k0 + i])
                dp = c[i] * (wk - Rr)
                Rr = Rr + abs(p[i]) * (wk - Rr)
                line = 2 * rec[rs + i, 0] // mod.LINE_B                      # fp32 tile offset doubled = fp64 tile offset
                for cc in range(8):
                    tiles[brick, line + corner[cc]] += w[rs + i, cc] * dp
    assert (seen >= 1).all(), "every brick needs a row"
    grad = np.zeros(shape)
    for brick in range(nb):
        bx, by, bz = brick // (nby * nbz), (brick // nbz) % nby, brick % nbz
        tl = tiles[brick].reshape(TX, TY, TZ)
        for lx in range(BX):
            for ly in range(BY):
                for lz in range(BZ):
                    x, y, z = bx * BX + lx, by * BY + ly, bz * BZ + lz
                    if x >= X or y >= Y or z >= Z:
                        continue
                    val = tl[lx, ly, lz]
                    for d in range(1, 8):
                        dx, dy, dz = d & 1, (d >> 1) & 1, d >> 2
                        if (dx and lx) or (dy and ly) or (dz and lz) or (dx and bx == 0) or (dy and by == 0) or (dz and bz == 0):
                            continue
                        nbrick = ((bx - dx) * nby + (by - dy)) * nbz + (bz - dz)
                        tx, ty, tz = (BX if dx else lx), (BY if dy else ly), (BZ if dz else lz)
                        h = halo_index(mod, tx, ty, tz)
                        # the scratch line h of the neighbour is its tile line (tx, ty, tz)
                        lines = [(a, b, c2) for a in range(TX) for b in range(TY) for c2 in range(TZ)
                                 if (a == BX or b == BY or c2 == BZ) and halo_index(mod, a, b, c2) == h]
                        assert lines == [(tx, ty, tz)]
                        val += tiles[nbrick].reshape(TX, TY, TZ)[tx, ty, tz]
                    if pre_scale != 0.0:
                        val = val * pre_scale if mask[x, y, z] else 0.0
                    grad[x, y, z] = val
    return grad
