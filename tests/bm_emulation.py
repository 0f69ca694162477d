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
        if mod.row_flag(flag) == mod.SKIP:
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
        shared = mod.row_flag(shared)
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
