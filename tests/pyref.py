"""Third, pure-Python restatement of the third-party sampler arithmetic (small cases only).
Independent of both the C oracle and the product's pgsgd_math.hpp; used to cross-check the oracle."""
import math
import struct

M64 = (1 << 64) - 1


def splitmix64(state):
    state = (state + 0x9e3779b97f4a7c15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
    return state, z ^ (z >> 31)


class Xoshiro256Plus:
    def __init__(self, seed):
        s = []
        st = seed & M64
        for _ in range(4):
            st, z = splitmix64(st)
            s.append(z)
        self.s = s

    def next(self):
        s = self.s
        result = (s[0] + s[3]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M64
        return result


def uniform_below(g, rng):
    x = g.next()
    prod = x * rng
    low = prod & M64
    if low < rng:
        thr = ((1 << 64) - rng) % rng
        while low < thr:
            x = g.next()
            prod = x * rng
            low = prod & M64
    return prod >> 64


def canonical(g):
    r = float(g.next()) * 2.0 ** -64   # int -> float is round-to-nearest-even, like the C cast
    return r if r < 1.0 else 1.0 - 2.0 ** -53


def fast_precise_pow(a, b):
    e = int(b)
    bits = struct.unpack("<q", struct.pack("<d", a))[0]
    hi = bits >> 32                      # arithmetic shift: signed high word
    nhi = int((b - e) * (hi - 1072632447) + 1072632447)
    frac = struct.unpack("<d", struct.pack("<q", (nhi & 0xFFFFFFFF) << 32 if nhi >= 0 else ((nhi & 0xFFFFFFFF) << 32) - (1 << 64)))[0]
    r = 1.0
    while e:
        if e & 1:
            r *= a
        a *= a
        e >>= 1
    return r * frac


def zipf(g, n, theta, zeta_n):
    alpha = 1.0 / (1.0 - theta)
    zeta2 = fast_precise_pow(1.0, theta) + fast_precise_pow(0.5, theta)
    num, den = 1.0 - fast_precise_pow(2.0 / n, 1.0 - theta), 1.0 - zeta2 / zeta_n
    eta = num / den if den != 0 else math.copysign(math.inf, num)
    u = canonical(g)
    uz = u * zeta_n
    if uz < 1.0:
        return 1
    if uz < 1.0 + fast_precise_pow(0.5, theta):
        return 2
    v = 1.0 + n * fast_precise_pow(eta * u - eta + 1.0, alpha)
    return max(1, min(n, int(v)))


def decode_og(path):
    """Independent pure-Python reading of an odgi .og file (graph_t::serialize, src/odgi.cpp:1632-1685;
    node_t::serialize, src/node.cpp:422-435; packed vector layout read off the fixture).  Returns
    dict(header, node_len, node_seq, edges [(from_handle, to_handle)], paths [(name, [handles])])."""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x80, 0xBD, 0xBA])
    off = 4

    def rq():
        nonlocal off
        v = struct.unpack_from("<Q", b, off)[0]
        off += 8
        return v

    def rvec():
        nonlocal off
        ws = rq()
        words = struct.unpack_from("<%dQ" % ws, b, off)
        off += 8 * ws
        mask, size = rq(), rq()
        bits, per_word = b[off], b[off + 1]
        off += 2
        return [(words[i // per_word] >> ((i % per_word) * bits)) & mask for i in range(size)]

    header = [rq() for _ in range(7)]
    n_nodes, n_paths = header[2], header[4]
    nodes = []
    for _ in range(n_nodes):
        sl = rq()
        seq = b[off:off + sl]
        off += sl
        nid = rq()
        nodes.append((nid, seq, rvec(), rvec(), rvec()))
    metas = []
    for _ in range(n_paths):
        length, first = rq(), (rq(), rq())
        rq(), rq()
        k = rq()
        metas.append((length, first, b[off:off + k].decode()))
        off += k
    assert off == len(b), "trailing bytes"
    edges = []
    for i, (nid, _, ev, _, _) in enumerate(nodes):
        for e in range(0, len(ev), 2):
            other, t = ev[e], ev[e + 1]
            if not t & 4:
                edges.append((2 * i + ((t >> 1) & 1), 2 * (other - 1) + (t & 1)))
    paths = []
    for j, (length, (h, rank), name) in enumerate(metas):
        node, handles = h >> 1, []
        for s in range(length):
            nid, _, _, dec, pv = nodes[node]
            rec = pv[6 * rank:6 * rank + 6]
            assert rec[0] == j + 1
            handles.append(2 * node + (rec[1] & 1))
            if (rec[1] >> 2) & 1:
                break
            d = dec[rec[4]]
            nxt = nid if d == 0 else (nid + (d >> 1) if d & 1 else nid - (d >> 1))
            node, rank = nxt - 1, rec[5]
        assert len(handles) == length
        paths.append((name, handles))
    return dict(header=header, node_len=[len(n[1]) for n in nodes], node_seq=[n[1] for n in nodes], edges=edges, paths=paths)


def build_tiles_py(path_first, step_handle, R, T, order="size", split=1):
    """Independent restatement of the tile table of the region-exclusive tile kernel: paths cut
    into tiles of T steps (single-step paths have none); a tile whose node ranks fit the window
    [r0*R, (r0+2)*R), r0 = rmin // R, joins work item r0; the launch of the even regions takes its items in
    order of decreasing step count, ties: smaller region first (order="region": in node order, the product's
    experiment PGSGD_TILE_ORDER=region), then every
    window-less tile as an item of its own; the launch of the odd regions follows.  split=k: the product's PGSGD_TILE_SPLIT
    (every window's tiles as k consecutive items).  Returns (tiles dict, items dict) shaped like
    LayoutSession.tile_table() / tile_items()."""
    import numpy as np
    pf = np.asarray(path_first, dtype=np.int64)
    ranks = np.asarray(step_handle, dtype=np.int64) >> 1
    raw = []
    for p in range(len(pf) - 1):
        b, cnt = int(pf[p]), int(pf[p + 1] - pf[p])
        if cnt <= 1:
            continue
        for o in range(0, cnt, T):
            n = min(T, cnt - o)
            r = ranks[b + o:b + o + n]
            raw.append((b + o, n, p, int(r.min()), int(r.max())))
    groups, nonlocal_ = ({}, {}), []
    for i, (t0, n, p, rmin, rmax) in enumerate(raw):
        r0 = rmin // R
        if rmax < (r0 + 2) * R:
            groups[r0 & 1].setdefault(r0, []).append(i)
        else:
            nonlocal_.append(i)
    tiles = dict(t0=[], cum=[], n=[], path=[])
    items = dict(tile_begin=[], tile_end=[], win0=[], local=[])
    total = 0

    def emit(i):
        nonlocal total
        t0, n, p, _, _ = raw[i]
        tiles["t0"].append(t0); tiles["cum"].append(total); tiles["n"].append(n); tiles["path"].append(p)
        total += n

    n_first = 0
    for colour in (0, 1):
        ordered = sorted(groups[colour].items(), key=(lambda kv: (-sum(raw[i][1] for i in kv[1]), kv[0])) if order == "size" else (lambda kv: kv[0]))
        for r0, members in ordered:
            items["tile_begin"].append(len(tiles["t0"]))
            for i in members:
                emit(i)
            items["tile_end"].append(len(tiles["t0"])); items["win0"].append(r0 * R); items["local"].append(1)
        if colour == 0:
            for i in nonlocal_:
                items["tile_begin"].append(len(tiles["t0"]))
                emit(i)
                items["tile_end"].append(len(tiles["t0"])); items["win0"].append(0); items["local"].append(0)
            n_first = len(items["local"])
    tiles = {k: np.array(v, dtype=np.uint64 if k in ("t0", "cum") else np.uint32) for k, v in tiles.items()}
    tiles["steps_total"] = total
    if split > 1:   # every window's tiles in `split` consecutive parts: part 0 of every window of a colour, then part 1, ... (split_items)
        cut = dict(tile_begin=[], tile_end=[], win0=[], local=[])
        bounds = ((0, n_first), (n_first, len(items["local"])))
        for lo, hi in bounds:
            idx = [i for i in range(lo, hi) if items["local"][i]]
            for j in range(split):
                for i in idx:
                    tb, te = items["tile_begin"][i], items["tile_end"][i]
                    parts = min(split, max(1, te - tb))
                    if j < parts:
                        cut["tile_begin"].append(tb + (te - tb) * j // parts); cut["tile_end"].append(tb + (te - tb) * (j + 1) // parts)
                        cut["win0"].append(items["win0"][i]); cut["local"].append(1)
            for i in range(lo, hi):
                if not items["local"][i]:
                    for k in cut:
                        cut[k].append(items[k][i])
            if lo == 0:
                n_first = len(cut["local"])
        items = cut
    items = {k: np.array(v, dtype=np.uint32) for k, v in items.items()}
    items["n_first"] = n_first
    return tiles, items
