"""Offline costing of STATIC ray orders (sort keys) from the diagnostic build's step log and ray log (RT_DIAG_LOG=<dir>, make diag):
key_eval.py <dir> [launch numbers...].

The step log gives every sampled ray's exact visited records (tools/lane_log.py), the ray log its origin and direction.  A wave's step count is the
size of the union of its rays' visited offsets, so ANY order of the sampled rays can be costed exactly: re-sort the rays of a sampled group (256
consecutive waves of the queue = 16 384 rays adjacent under the production key), cut the new order into waves of 64, count unions.  Re-sorting inside a
group is what a different minor key would do; it cannot bring in rays from elsewhere in the queue, so a key that does NOT win here has little to
offer globally either.  Printed per launch: steps of each candidate order / steps of the production order (1.0 = today), and two orders that use
knowledge no key has (the visited sets themselves) as bounds for what any static order could reach."""
import glob, os, sys
import numpy as np

GROUP = int(os.environ.get("RT_DIAG_LOG_GROUP", "256"))
# RT_SORT_GRID="minx miny minz scalex scaley scalez" (the diagnostic build prints it as [diag-sortgrid]): evaluate refinements of the production key
# as GLOBAL keys; without it: candidate keys quantised over each group's own extent
GRID = [float(x) for x in os.environ["RT_SORT_GRID"].split()] if os.environ.get("RT_SORT_GRID") else None


def load(dirname, launch):
    f = glob.glob(os.path.join(dirname, "launch_%03d_*.bin" % launch))[0]
    g = glob.glob(os.path.join(dirname, "rays_%03d_*.bin" % launch))[0]
    raw = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
    wave = raw[:, 0].astype(np.int64); off = (raw[:, 1] & ~np.uint32(15)).astype(np.int64)
    mask = (raw[:, 3].astype(np.uint64) << np.uint64(32)) | raw[:, 2].astype(np.uint64)
    keep = off != off.max()
    wave, off, mask = wave[keep], off[keep], mask[keep]
    bits = np.unpackbits(mask.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").astype(bool)
    st, lane = np.nonzero(bits)
    rid = wave[st] * 64 + lane
    roff = off[st] >> 4
    rays = np.fromfile(g, dtype=np.float32).reshape(-1, 8)
    ids = rays[:, 0].copy().view(np.uint32).astype(np.int64)
    return os.path.basename(f), rid, roff, ids, rays[:, 1:4].astype(np.float64), rays[:, 4:7].astype(np.float64), len(wave)


def spread(v, bits):
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (2 * b)
    return out


def spread3(v, bits):
    out = np.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def quant(x, bits):
    lo, hi = x.min(), x.max()
    return np.clip(((x - lo) / max(hi - lo, 1e-30) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)


def octa(d):   # octahedral map of the unit direction to [0, 1)^2
    n = d / np.abs(d).sum(axis=1, keepdims=True)
    u, v = n[:, 0].copy(), n[:, 1].copy()
    neg = n[:, 2] < 0
    uu = (1 - np.abs(v)) * np.sign(u + 1e-300); vv = (1 - np.abs(u)) * np.sign(v + 1e-300)
    u[neg], v[neg] = uu[neg], vv[neg]
    return (u + 1) * 0.5, (v + 1) * 0.5


def keys(o, d):
    """candidate keys for the rays of ONE group (quantisation ranges are the group's own)"""
    dn = d / np.linalg.norm(d, axis=1, keepdims=True)
    out = {}
    octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
    ad = np.abs(d); am = ad.max(axis=1, keepdims=True)
    for cb, db in ((4, 3), (5, 3), (6, 2), (3, 4), (6, 0), (0, 5)):
        cell = spread3(quant(o[:, 0], cb), cb) | (spread3(quant(o[:, 1], cb), cb) << 1) | (spread3(quant(o[:, 2], cb), cb) << 2) if cb else np.zeros(len(o), np.int64)
        dq = np.clip((ad / am * ((1 << db) - 1e-3)).astype(np.int64), 0, (1 << db) - 1) if db else np.zeros((len(o), 3), np.int64)
        dbin = dq[:, 0] | (dq[:, 1] << db) | (dq[:, 2] << (2 * db))
        out["octant|cell%d|dir%d" % (cb, db)] = (octant << (3 * cb + 3 * db)) | (cell << (3 * db)) | dbin
        out["octant|dir%d|cell%d" % (db, cb)] = (octant << (3 * cb + 3 * db)) | (dbin << (3 * cb)) | cell
    # line space: direction bin (octahedral, db bits per axis), then the 2-D position of the LINE in the plane perpendicular to the bin's centre direction
    c = o.mean(axis=0)
    ou, ov = octa(dn)
    for db, pb in ((3, 6), (4, 6), (5, 5), (2, 7)):
        bu, bv = np.clip((ou * (1 << db)).astype(np.int64), 0, (1 << db) - 1), np.clip((ov * (1 << db)).astype(np.int64), 0, (1 << db) - 1)
        dbin = spread(bu, db) | (spread(bv, db) << 1)
        # basis per bin from the bin's mean direction
        pos = np.zeros((len(o), 2))
        for b in np.unique(dbin):
            m = dbin == b
            w = dn[m].mean(axis=0); w /= np.linalg.norm(w)
            a = np.array([0.0, 1.0, 0.0]) if abs(w[0]) > 0.9 else np.array([1.0, 0.0, 0.0])
            e1 = np.cross(w, a); e1 /= np.linalg.norm(e1); e2 = np.cross(w, e1)
            rel = o[m] - c
            # the line's point closest to c, seen along each ray's OWN direction, expressed in the bin's basis
            foot = rel - (rel * dn[m]).sum(axis=1, keepdims=True) * dn[m]
            pos[m, 0] = foot @ e1; pos[m, 1] = foot @ e2
        pq = spread(quant(pos[:, 0], pb), pb) | (spread(quant(pos[:, 1], pb), pb) << 1)
        out["line: dir%d(octa)|foot%d" % (db, pb)] = (dbin << (2 * pb)) | pq
        out["line: foot%d|dir%d(octa)" % (pb, db)] = (pq << (2 * db)) | dbin
    return out


def refined_keys(o, d, grid):
    """the production key (rt_shade.hip.h sort_key: octant | cell 4 | direction 2 | foot point 5+5, 31 bits) computed as the device computes it from the
    scene's sort grid, extended by MINOR bits.  Sorting a group by such a key is exactly what the global sort by it would do to the group's rays
    (RT_KEY_BASE=legacy: refinements of the 24-bit key of r01-r03p instead)."""
    f32 = np.float32
    smin, sscale = np.array(grid[:3], f32), np.array(grid[3:], f32)
    of, df = o.astype(f32), d.astype(f32)
    f = (of - smin) * sscale                                        # 32 cells per axis over the grid
    octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
    ad = np.abs(df); am = np.maximum(ad.max(axis=1, keepdims=True), f32(1e-30))
    c5 = np.clip(f, 0, 31).astype(np.int64)
    def cellbits(bits):    # Morton code of the origin cell at `bits` per axis over the same grid (clamped)
        q = np.clip((f * f32((1 << bits) / 32.0)).astype(np.int64), 0, (1 << bits) - 1) if bits > 5 else c5 >> (5 - bits)
        return spread3(q[:, 0], bits) | (spread3(q[:, 1], bits) << 1) | (spread3(q[:, 2], bits) << 2), q
    def dirbits(bits):
        dq = (ad * (f32((1 << bits) - (0.001 if bits < 3 else 0.001)) / am)).astype(np.int64) & ((1 << bits) - 1)
        return dq[:, 0] | (dq[:, 1] << bits) | (dq[:, 2] << (2 * bits)), dq
    il = f32(1.0) / np.maximum(np.sqrt((df * df).sum(axis=1, keepdims=True)), f32(1e-30))
    n = df * il
    r = (of - smin) - f32(16.0) / sscale
    along = (r * n).sum(axis=1, keepdims=True)
    g = (r - along * n) * sscale * f32(1.0 / 32.0) + f32(0.5)
    dom = np.where((ad[:, 0] >= ad[:, 1]) & (ad[:, 0] >= ad[:, 2]), 0, np.where(ad[:, 1] >= ad[:, 2], 1, 2))
    fu = np.where(dom == 0, g[:, 1], g[:, 0]); fv = np.where(dom == 2, g[:, 1], g[:, 2])
    def footq(fb):
        return np.clip((fu * f32(1 << fb)).astype(np.int64), 0, (1 << fb) - 1), np.clip((fv * f32(1 << fb)).astype(np.int64), 0, (1 << fb) - 1)
    def footbits(fb):
        qu, qv = footq(fb)
        return spread(qu, fb) | (spread(qv, fb) << 1)
    def low3(q, extra):
        m = (1 << extra) - 1
        return spread3(q[:, 0] & m, extra) | (spread3(q[:, 1] & m, extra) << 1) | (spread3(q[:, 2] & m, extra) << 2)
    m4, _ = cellbits(4)
    out = {}
    if os.environ.get("RT_KEY_BASE") == "legacy":
        d3, _ = dirbits(3)
        prod = (octant << 21) | (m4 << 9) | d3
        out["production key r01-r03p (octant|cell4|dir3), ties in queue order"] = prod
        for fb in (3, 4, 5, 7):
            out["legacy | foot point %d+%d bits  [%d bits]" % (fb, fb, 24 + 2 * fb)] = (prod << (2 * fb)) | footbits(fb)
        for extra in (1, 2, 3, 4):
            out["legacy | cell +%d bits/axis" % extra] = (prod << (3 * extra)) | low3(cellbits(4 + extra)[1], extra)
        d2, _ = dirbits(2)
        out["octant|cell4|dir2|foot5+5  [31 bits]"] = (((((octant << 12) | m4) << 6) | d2) << 10) | footbits(5)
        return out
    d2, _ = dirbits(2)
    prod = (((((octant << 12) | m4) << 6) | d2) << 10) | footbits(5)
    out["production key (octant|cell4|dir2|foot5+5), ties in queue order"] = prod
    for fe in (1, 2, 3):
        qu, qv = footq(5 + fe); m = (1 << fe) - 1
        out["production | foot +%d bits/axis" % fe] = (prod << (2 * fe)) | spread(qu & m, fe) | (spread(qv & m, fe) << 1)
    for ce in (1, 2, 3):
        out["production | cell +%d bits/axis" % ce] = (prod << (3 * ce)) | low3(cellbits(4 + ce)[1], ce)
    for de in (1, 2):
        _, dqf = dirbits(2 + de); m = (1 << de) - 1
        out["production | dir +%d bits/axis" % de] = (prod << (3 * de)) | (dqf[:, 0] & m) | ((dqf[:, 1] & m) << de) | ((dqf[:, 2] & m) << (2 * de))
    for ab in (3, 5):      # position of the origin ALONG the line (grid units of the dominant axis), coarse
        a = along[:, 0] * sscale[dom] * f32(1.0 / 32.0) + f32(0.5)
        out["production | position along the line, %d bits" % ab] = (prod << ab) | np.clip((a * f32(1 << ab)).astype(np.int64), 0, (1 << ab) - 1)
    # other LAYOUTS (their major bits differ from production's, so re-sorting inside a production group only approximates the global sort: "~")
    m3, _ = cellbits(3); m2, _ = cellbits(2); m5, _ = cellbits(5); d1, _ = dirbits(1); d3, _ = dirbits(3)
    out["~ octant|cell3|dir2|foot5+5"] = (((((octant << 9) | m3) << 6) | d2) << 10) | footbits(5)
    out["~ octant|cell3|dir2|foot6+6"] = (((((octant << 9) | m3) << 6) | d2) << 12) | footbits(6)
    out["~ octant|cell2|dir2|foot6+6"] = (((((octant << 6) | m2) << 6) | d2) << 12) | footbits(6)
    out["~ octant|dir2|foot5+5|cell4"] = (((((octant << 6) | d2) << 10) | footbits(5)) << 12) | m4
    out["~ octant|dir2|foot4+4|cell4"] = (((((octant << 6) | d2) << 8) | footbits(4)) << 12) | m4
    out["~ octant|dom|foot6+6|cell3"] = (((((octant << 2) | dom) << 12) | footbits(6)) << 9) | m3
    out["~ octant|cell4|dir1|foot5+5"] = (((((octant << 12) | m4) << 3) | d1) << 10) | footbits(5)
    out["~ octant|dom|cell4|foot5+5"] = (((((octant << 2) | dom) << 12) | m4) << 10) | footbits(5)
    out["~ octant|cell5|dir2|foot4+4"] = (((((octant << 15) | m5) << 6) | d2) << 8) | footbits(4)
    out["~ octant|cell4|dir3|foot4+4"] = (((((octant << 12) | m4) << 9) | d3) << 8) | footbits(4)
    out["~ octant|dir2|cell2|foot5+5|cell+2"] = (((((((octant << 6) | d2) << 6) | m2) << 10) | footbits(5)) << 6) | low3(cellbits(4)[1], 2)
    qu, qv = footq(6); m = 1
    out["production | foot +1 | cell +1 bits/axis"] = (((prod << 2) | spread(qu & m, 1) | (spread(qv & m, 1) << 1)) << 3) | low3(cellbits(5)[1], 1)
    return out


def cost(order_rank, pair_ray, pair_off):
    w = order_rank[pair_ray] // 64
    return len(np.unique(w * (1 << 24) + pair_off))


def main():
    d = sys.argv[1]
    launches = [int(x) for x in sys.argv[2:]] or [1, 2]
    rng = np.random.default_rng(1)
    for L in launches:
        name, rid, roff, ids, o, dd, steps_logged = load(d, L)
        grp_of_ray = ids // (64 * GROUP)
        tot = {}
        base_total = 0
        for g in np.unique(grp_of_ray):
            m = np.flatnonzero(grp_of_ray == g)
            if len(m) < 64 * GROUP // 2:
                continue
            gid = ids[m]; order0 = np.argsort(gid)                      # production order inside the group
            gid, go, gd = gid[order0], o[m][order0], dd[m][order0]
            pm = np.flatnonzero((rid // (64 * GROUP)) == g)
            pair_ray = np.searchsorted(gid, rid[pm]); ok = (pair_ray < len(gid)) & (gid[np.minimum(pair_ray, len(gid) - 1)] == rid[pm])
            pair_ray, pair_off = pair_ray[ok], roff[pm][ok]
            n = len(gid)
            base = cost(np.arange(n), pair_ray, pair_off)
            base_total += base
            cand = keys(go, gd) if GRID is None else refined_keys(go, gd, GRID)
            for k, v in cand.items():
                rank = np.empty(n, np.int64); rank[np.argsort(v, kind="stable")] = np.arange(n)
                tot[k] = tot.get(k, 0) + cost(rank, pair_ray, pair_off)
            rank = rng.permutation(n)
            tot["(random order)"] = tot.get("(random order)", 0) + cost(rank, pair_ray, pair_off)
            # bounds from the visited sets themselves: min-hash order, and the order of each ray's LAST visited record
            nray = n
            sig = np.full((4, nray), np.iinfo(np.int64).max)
            for h in range(4):
                hv = (pair_off * np.int64(2654435761 + 97 * h) + np.int64(40503 * (h + 1))) % np.int64(2147483647)
                np.minimum.at(sig[h], pair_ray, hv)
            rank = np.empty(n, np.int64); rank[np.lexsort((sig[3], sig[2], sig[1], sig[0]))] = np.arange(n)
            tot["(bound: min-hash of the visited sets)"] = tot.get("(bound: min-hash of the visited sets)", 0) + cost(rank, pair_ray, pair_off)
            last = np.zeros(nray, np.int64); np.maximum.at(last, pair_ray, pair_off)
            cnt = np.bincount(pair_ray, minlength=nray)
            rank = np.empty(n, np.int64); rank[np.lexsort((cnt, last))] = np.arange(n)
            tot["(bound: by last visited record, then visit count)"] = tot.get("(bound: by last visited record, then visit count)", 0) + cost(rank, pair_ray, pair_off)
        print(name, "logged steps", steps_logged, " production order inside the complete groups:", base_total)
        for k, v in sorted(tot.items(), key=lambda kv: kv[1]):
            print("   %-52s %.3f" % (k, v / max(base_total, 1)))


if __name__ == "__main__":
    main()
