"""CPU tests (-m "not gpu") of the trace stream k_trace walks, built on the host by
rt_stream_build_host -- the same code path rt_scene_upload runs before its first HIP call.

Checked here, without a GPU: every record is well-formed and sits where the node table says, skip links land on
record starts of the same range or on that range's END record, the fp32 boxes / triangle blocks / sphere blocks
restate the blob's f64 tables, the error weights of the triangle filter are upper bounds, and walking the
stream (links only) visits exactly what the reference's recursion visits (bvh.rs:91-106), compared through the
oracle's box-test counts."""
import struct

import numpy as np
import pytest

import oracle_ffi as orc
import raytracing_amd as rt
from scene_text import SceneText, scale, translate

NO_BOX, DUP_RIGHT = 1, 2
INNER, TRI2, LEAF, SPH2 = 0, 4, 8, 12
REC_BYTES = {INNER: 64, TRI2: 144, LEAF: 48, SPH2: 112}
END_NODE = 0xFFFFFF
PAD = 384
SCENES = ["random", "cornell", "simple", "simple_triangle", "mesh", "simple_light", "sphere", "random_blur"]


def _records(stream):
    """Parses the stream front to back: list of dicts (off, box[6], skip, flags, kind, size, node)."""
    out, off, end = [], 0, len(stream) - PAD
    raw = stream.tobytes()
    while off < end:
        box = struct.unpack_from("<6f", raw, off)
        skip, word = struct.unpack_from("<2I", raw, off + 24)
        r = dict(off=off, box=box, skip=skip, flags=word & 3, kind=word & 12, size=((word >> 4) & 15) * 16, node=word >> 8)
        assert r["size"] == (32 if r["node"] == END_NODE else REC_BYTES[r["kind"]]), r
        out.append(r)
        off += r["size"]
    assert off == end and not stream[end:].any()          # records tile the stream exactly; the over-read pad is zero
    return out


def _ranges(hs):
    h, pr = hs.header(), hs.prims()
    rs = [(h.root_node, h.root_end)]
    rs += sorted({(int(p["index"]), int(p["sub_end"])) for p in pr if p["kind"] == 4})
    return rs


@pytest.mark.parametrize("name", SCENES)
def test_stream_structure(name):
    hs = rt.HostScene.builtin(name, bvh_seed=2, legacy_bvh=(name == "simple_triangle"))
    h, nodes, pr, geo = hs.header(), hs.nodes(), hs.prims(), hs.geoms()
    stream, root_off, node_off = hs.trace_stream()
    recs = _records(stream)
    by_off = {r["off"]: r for r in recs}
    real = [r for r in recs if r["node"] != END_NODE]
    ends = [r for r in recs if r["node"] == END_NODE]
    assert [r["node"] for r in real] == list(range(h.n_nodes))            # pre-order = node index order
    assert [r["off"] for r in real] == list(node_off[:h.n_nodes])
    ranges = _ranges(hs)
    assert len(ends) == len({e for _, e in ranges})                        # one END per distinct range end
    inf = float("inf")
    for e in ends:
        assert e["box"] == (-inf, inf, -inf, inf, -inf, inf) and e["flags"] == NO_BOX and e["kind"] == LEAF and e["skip"] == e["off"]
    assert root_off == (node_off[h.root_node] if h.root_end > h.root_node else ends[0]["off"])
    # END of a range sits directly behind the range's last record
    end_of = {}
    for b, e in ranges:
        last = by_off[int(node_off[e - 1])]
        nxt = by_off[last["off"] + last["size"]]
        assert nxt["node"] == END_NODE
        end_of[e] = nxt["off"]
    range_end = np.zeros(h.n_nodes, np.int64)
    for b, e in ranges:
        range_end[b:e] = e
    raw = stream.tobytes()
    for r in real:
        i = r["node"]; nd = nodes[i]
        assert r["flags"] == int(nd["flags"]) & 3
        # links: inside the range -> that node's record; to the range end -> the END record
        tgt = int(nd["skip"])
        assert r["skip"] == (end_of[tgt] if tgt == range_end[i] else int(node_off[tgt])), (i, tgt)
        assert r["skip"] > r["off"]
        if r["flags"] & NO_BOX:
            assert r["box"] == (-inf, inf, -inf, inf, -inf, inf)
        else:
            want = np.stack([nd["bmin"], nd["bmax"]], axis=1).reshape(6).astype(np.float32)
            assert np.array_equal(np.array(r["box"], np.float32), want)
        p0, p1 = int(nd["prim0"]), int(nd["prim1"])
        if p0 == rt.NONE:
            assert r["kind"] == INNER and p1 == rt.NONE
            continue
        kinds = [int(geo[p]["kind"]) for p in (p0, p1) if p != rt.NONE]
        same_xf = p1 == rt.NONE or geo[p0]["xform"] == geo[p1]["xform"]
        if all(k == 0 for k in kinds) and same_xf:
            assert r["kind"] == TRI2
            xf, = struct.unpack_from("<I", raw, r["off"] + 32)
            assert xf == int(geo[p0]["xform"])
            blk = np.frombuffer(raw, np.float32, 24, r["off"] + 40).reshape(12, 2)
            for t, p in enumerate((p0, p1 if p1 != rt.NONE else p0)):     # a single triangle fills both halves
                v0, e1, e2 = geo[p]["d"][:3], geo[p]["d"][3:6], geo[p]["d"][6:9]
                ml, nn = np.cross(v0, e2), np.cross(e1, e2)
                assert np.array_equal(blk[0:3, t], e2.astype(np.float32))
                assert np.allclose(blk[3:6, t], ml, rtol=1e-6, atol=1e-30) and np.allclose(blk[6:9, t], -nn, rtol=1e-6, atol=1e-30)
                # error weights are upper bounds of the analytic expressions, rounded up
                n1 = lambda v: np.abs(v).sum()
                ni = lambda v: np.abs(v).max()
                assert float(blk[9, t]) >= 2.0 ** -20 * ni(e2)
                assert float(blk[10, t]) >= 2.0 ** -20 * n1(ml) + 2.0 ** -46 * ni(v0) * n1(e2)
                assert float(blk[11, t]) >= (1 + 2.0 ** -19) * (2.0 ** -20 * n1(nn) + 2.0 ** -46 * n1(e1) * n1(e2))
                # ... and tight: not more than a few ulps above them
                assert float(blk[9, t]) <= 2.0 ** -20 * ni(e2) * (1 + 1e-6) + 1e-44
        elif all(k == 1 for k in kinds) and same_xf:
            assert r["kind"] == SPH2
            xf, cnt = struct.unpack_from("<2I", raw, r["off"] + 32)
            assert xf == int(geo[p0]["xform"]) and cnt == len(kinds)
            sp = np.frombuffer(raw, np.float64, 4 * cnt, r["off"] + 40).reshape(cnt, 4)
            for t, p in enumerate([p for p in (p0, p1) if p != rt.NONE]):
                assert np.array_equal(sp[t].view(np.uint64), geo[p]["d"][:4].view(np.uint64))
        else:
            assert r["kind"] == LEAF
            assert struct.unpack_from("<3I", raw, r["off"] + 32) == (p0, p1, int(geo[p0]["xform"]))


def _walk(stream, start, decide):
    """k_trace's control flow over the stream, links only: decide(record) -> does its box pass.
    Returns the node indices whose record was reached."""
    raw = stream.tobytes()
    off, seen = start, []
    for _ in range(1 << 22):
        skip, word = struct.unpack_from("<2I", raw, off + 24)
        node, kind, size = word >> 8, word & 12, ((word >> 4) & 15) * 16
        if node == END_NODE:
            return seen
        seen.append(node)
        off = off + size if decide(node, word & 3, kind) else skip
    raise AssertionError("walk did not reach an END record")


@pytest.mark.parametrize("name", SCENES)
def test_walks_terminate_and_cover_each_range(name):
    hs = rt.HostScene.builtin(name, bvh_seed=6, legacy_bvh=(name == "simple_triangle"))
    nodes = hs.nodes()
    stream, root_off, node_off = hs.trace_stream()
    for b, e in _ranges(hs):
        start = root_off if (b, e) == _ranges(hs)[0] else int(node_off[b])
        assert _walk(stream, start, lambda n, f, k: True) == list(range(b, e))       # every box passes: every node once, in order
        # every box fails: only the nodes no other node of the range contains (skip chain from the first)
        top, i = [], b
        while i < e:
            top.append(i); i = int(nodes[i]["skip"])
        assert _walk(stream, start, lambda n, f, k: False) == top


def _ground_scene():
    """Big far spheres around a small cloud of triangles in a space of their own: the spheres' ancestors hold every box
    that can be pruned."""
    rng = np.random.default_rng(11)
    s = SceneText(bvh_seed=4)
    m = s.lambertian_rgb(0.5, 0.5, 0.5)
    nt = 300
    v0 = rng.uniform(-1, 1, (nt, 3))
    tri = np.stack([v0, v0 + rng.normal(0, 0.1, (nt, 3)), v0 + rng.normal(0, 0.1, (nt, 3))], 1).reshape(-1, 3)
    me = s.mesh([tuple(map(float, p)) for p in tri], [(0, 0, 1)] * len(tri), [(0, 0)] * len(tri), [(3 * k, 3 * k + 1, 3 * k + 2) for k in range(nt)], m)
    xf = s.xform(translate(0.5, 2.0, -0.25))
    ids = [s.transformed(xf, s.triangle(me, k)) for k in range(nt)]
    ids += [s.sphere((0.0, -1003.0 - 40.0 * i, 5.0 * i), 1000.0, m) for i in range(4)]
    return rt.HostScene.from_description(s.finish(ids, build_bvh=True))


def _children(nodes, i):
    out, c = [], i + 1
    while c < int(nodes[i]["skip"]):
        out.append(c); c = int(nodes[c]["skip"])
    return out


def _expected_elision(hs, pn, r0, r1, local, area_q=0.7):
    """Rules (2) and (3) of prepare_scene (rt_device.hip), restated on the production node array: (2) with strict pruning on, inner nodes that are
    never pruned and whose box holds every node that can be; (3) inner nodes whose box has at least area_q of the surface area of their nearest
    ancestor that keeps its record.  `local(i)` = the fp32 local box of inner node i (infinite: never pruned)."""
    info = hs.prune_info()
    inner = [i for i in range(r0, r1) if pn[i]["prim0"] == rt.NONE]
    gone = set()
    if info["on"]:
        prunable = [i for i in inner if np.isfinite(local(i)).all()]
        if prunable:
            core_lo = np.min([pn[i]["bmin"] for i in prunable], axis=0); core_hi = np.max([pn[i]["bmax"] for i in prunable], axis=0)
            pr = set(prunable)
            for i in inner:
                if i not in pr and int(pn[i]["skip"]) > i + 1 and (pn[i]["bmin"] <= core_lo).all() and (core_hi <= pn[i]["bmax"]).all():
                    gone.add(i)

    def area(i):
        e = np.maximum(pn[i]["bmax"] - pn[i]["bmin"], 0.0)
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
    kept = {i: np.inf for i in range(r0, r1)}
    for i in inner:                                                      # pre-order: parents come first
        own = area(i)
        if i not in gone and int(pn[i]["skip"]) > i + 1 and np.isfinite(kept[i]) and own >= area_q * kept[i]:
            gone.add(i)
        down = kept[i] if i in gone else own
        for c in _children(pn, i):
            kept[c] = down
    return gone, inner


def _boxes_nest(hs):
    h, nodes = hs.header(), hs.nodes()
    rng_ = range(h.root_node, h.root_end)
    ok = all(not (int(nodes[i]["flags"]) & NO_BOX) and np.isfinite(nodes[i]["bmin"]).all() and np.isfinite(nodes[i]["bmax"]).all() for i in rng_)
    for i in rng_:
        if not ok or nodes[i]["prim0"] != rt.NONE:
            continue
        for c in _children(nodes, i):
            ok = ok and int(nodes[c]["skip"]) <= int(nodes[i]["skip"]) and (nodes[i]["bmin"] <= nodes[c]["bmin"]).all() and (nodes[c]["bmax"] <= nodes[i]["bmax"]).all()
    return ok


@pytest.mark.parametrize("name", ["mesh", "ground", "random", "cornell", "simple_triangle"])
def test_production_stream_is_the_reference_leaf_sequence_under_a_hierarchy_of_our_own(name):
    # The stream the kernel walks ("production stream").  When the reference's boxes nest, the reference reaches exactly the
    # leaves whose OWN box passes, in pre-order; what stands above the leaves is free as long as it encloses them.  So (r03): the scene-level
    # hierarchy is rebuilt over the same leaf order with surface-area splits, and inner nodes most waves enter anyway have no record.
    hs = _ground_scene() if name == "ground" else rt.HostScene.builtin(name, bvh_seed=3, legacy_bvh=(name == "simple_triangle"))
    h, nodes, info = hs.header(), hs.nodes(), hs.prune_info()
    full, root_full, off_full = hs.trace_stream()
    prod, root_prod, off_prod = hs.trace_stream(production=True)
    pn, r0, r1, gone_flags = hs.production_nodes()
    ref_leaves = [i for i in range(h.root_node, h.root_end) if nodes[i]["prim0"] != rt.NONE]
    rebuilt = _boxes_nest(hs) and len(ref_leaves) >= 4
    assert (len(pn) > h.n_nodes) == rebuilt
    assert pn[:h.n_nodes].tobytes() == nodes.tobytes()                      # the blob's nodes come first, untouched (the full stream indexes them)
    if not rebuilt:
        assert (r0, r1) == (h.root_node, h.root_end) and len(pn) == h.n_nodes
        if name == "cornell":
            assert not gone_flags.any() and np.array_equal(full, prod)      # a list of instances: no boxes at the scene level
    else:
        assert (r0, r1) == (h.n_nodes, h.n_nodes + 2 * len(ref_leaves) - 1) and len(pn) == r1
        # a pre-order binary tree over copies of the reference's leaves in the reference's order; inner boxes are the exact unions of their children's
        copies = [i for i in range(r0, r1) if pn[i]["prim0"] != rt.NONE]
        assert len(copies) == len(ref_leaves)
        for c, o in zip(copies, ref_leaves):
            for f in ("bmin", "bmax", "flags", "prim0", "prim1"):
                assert np.array_equal(pn[c][f], nodes[o][f])
            assert int(pn[c]["skip"]) == c + 1
            assert off_prod[o] == _record_at(prod, int(off_prod[o]))["off"] and _record_at(prod, int(off_prod[o]))["node"] == c     # the certificate's look-up by blob index lands on the copy
        for i in range(r0, r1):
            if pn[i]["prim0"] != rt.NONE:
                continue
            kids = _children(pn, i)
            assert len(kids) == 2 and int(pn[i]["skip"]) <= r1 and all(int(pn[k]["skip"]) <= int(pn[i]["skip"]) for k in kids)
            assert np.array_equal(pn[i]["bmin"], np.min([pn[k]["bmin"] for k in kids], axis=0)) and np.array_equal(pn[i]["bmax"], np.max([pn[k]["bmax"] for k in kids], axis=0))
        assert gone_flags[h.root_node:h.root_end].all()                     # the reference's own scene-level range has no record in this stream
    # records: one per node of the range that keeps one, in order; rules (2) and (3) restated
    rp = _records(prod)
    rawp = prod.tobytes()
    by_node = {r["node"]: r for r in rp if r["node"] != END_NODE}
    local = lambda i: struct.unpack_from("<6f", rawp, by_node[i]["off"] + 32) if i in by_node else (np.inf,) * 6
    # the local box of a node without a record is not in the stream: rule (2) only asks whether it is finite, i.e. whether every leaf below can be
    # covered -- equivalently whether every RECORDED inner node below it has a finite local box and every leaf below hangs under one or is coverable;
    # the library's own flags are checked for consistency instead: an elided node's children are where links to it land
    want_range = [i for i in range(r0, r1) if not gone_flags[i]]
    scene_recs = [r["node"] for r in rp if r["node"] != END_NODE and r0 <= r["node"] < r1]
    assert scene_recs == want_range
    n_gone = int(gone_flags[r0:r1].sum())
    assert n_gone == int(info["elided_nodes"])
    if rebuilt or _boxes_nest(hs):
        if name in ("mesh", "ground"):
            assert n_gone > 0
        for i in range(r0, r1):
            if gone_flags[i]:
                assert pn[i]["prim0"] == rt.NONE and int(pn[i]["skip"]) > i + 1          # only inner nodes go
        # rule (3) on its own is checkable from the node array: every node it names is gone, and every gone node is named by (2) or (3)
        want, _ = _expected_elision(hs, pn, r0, r1, lambda i: local(i) if not gone_flags[i] else _local_by_union(pn, i, gone_flags, local))
        assert want == {i for i in range(r0, r1) if gone_flags[i]}
    else:
        assert n_gone == 0
    # links: a record's skip is the record of the node-level skip target (or the first record below it, or the range's END)
    off_of = {r["node"]: r["off"] for r in rp if r["node"] != END_NODE}
    end_rec = [r for r in rp if r["node"] == END_NODE]

    def first_record_at_or_after(i):
        while i < r1 and gone_flags[i]:
            i += 1                                                           # an elided node's first child follows it in pre-order
        return off_of[i] if i < r1 else None
    for i in want_range:
        tgt = first_record_at_or_after(int(pn[i]["skip"]))
        if tgt is None:
            assert any(e["off"] == by_node[i]["skip"] for e in end_rec)
        else:
            assert by_node[i]["skip"] == tgt
    # walks: every box passes -> every recorded node once, in order, and the leaves in the reference's order
    seen = _walk(prod, root_prod, lambda n, f, k: True)
    assert seen == want_range
    assert [(int(pn[n]["prim0"]), int(pn[n]["prim1"])) for n in seen if pn[n]["prim0"] != rt.NONE] == [(int(nodes[o]["prim0"]), int(nodes[o]["prim1"])) for o in ref_leaves]
    if not _boxes_nest(hs):
        return
    # a decision that is monotone under enclosure (box contains the point q): the leaves whose OWN box holds q are reached by both walks, in the
    # same order -- the reference's walk reaches nothing else that passes, and neither does ours
    rng = np.random.default_rng(11)
    for _ in range(12):
        lf = ref_leaves[int(rng.integers(len(ref_leaves)))]
        q = nodes[lf]["bmin"] + rng.random(3) * (nodes[lf]["bmax"] - nodes[lf]["bmin"])
        inside_f = lambda n, f, k: bool((nodes[n]["bmin"] <= q).all() and (q <= nodes[n]["bmax"]).all())
        inside_p = lambda n, f, k: bool((pn[n]["bmin"] <= q).all() and (q <= pn[n]["bmax"]).all())
        hit_f = [(int(nodes[n]["prim0"]), int(nodes[n]["prim1"])) for n in _walk(full, root_full, inside_f) if nodes[n]["prim0"] != rt.NONE and inside_f(n, 0, 0)]
        hit_p = [(int(pn[n]["prim0"]), int(pn[n]["prim1"])) for n in _walk(prod, root_prod, inside_p) if pn[n]["prim0"] != rt.NONE and inside_p(n, 0, 0)]
        brute = [(int(nodes[o]["prim0"]), int(nodes[o]["prim1"])) for o in ref_leaves if (nodes[o]["bmin"] <= q).all() and (q <= nodes[o]["bmax"]).all()]
        assert hit_f == brute and hit_p == brute and brute


def _record_at(stream, off):
    raw = stream.tobytes()
    skip, word = struct.unpack_from("<2I", raw, off + 24)
    return dict(off=off, skip=skip, node=word >> 8, kind=word & 12)


def _local_by_union(pn, i, gone_flags, local):
    """fp32 local box of an inner node WITHOUT a record: the union of the local boxes of the recorded inner nodes below it, infinite as soon as a
    leaf hangs directly below an unrecorded chain (a leaf's own local box is not in the stream) -- enough to tell finite from infinite for the
    nodes rule (2) can name: their subtree holds a never-covered leaf, so some chain below them ends in an infinite box or a bare leaf."""
    lo = np.full(3, np.inf); hi = np.full(3, -np.inf)
    stack = _children(pn, i)
    while stack:
        c = stack.pop()
        if pn[c]["prim0"] != rt.NONE:
            return (-np.inf, np.inf) * 3
        if gone_flags[c]:
            stack += _children(pn, c)
            continue
        b = local(c)
        if not np.isfinite(b).all():
            return (-np.inf, np.inf) * 3
        lo = np.minimum(lo, b[0::2]); hi = np.maximum(hi, b[1::2])
    return (lo[0], hi[0], lo[1], hi[1], lo[2], hi[2])


@pytest.mark.parametrize("name,seed", [("random", 1), ("random", 8), ("mesh", 3), ("simple_triangle", 0)])
def test_walk_visits_what_the_reference_recursion_visits(name, seed):
    # Rays with a hit window [t0, t0 + 1e-12]: a box passes when the ray point at t0 lies in it, practically no surface
    # does, so t_max never shrinks and the oracle's box-test count (bvh.rs:91-106: one per node reached, list items of
    # hittable.rs:41-53 have none) must equal the number of box-carrying records the stream walk reaches.
    hs = rt.HostScene.builtin(name, bvh_seed=seed, legacy_bvh=(name == "simple_triangle"))
    osc = orc.OracleScene(hs.describe())
    h, nodes = hs.header(), hs.nodes()
    stream, root_off, _ = hs.trace_stream()
    rays = osc.primary_rays(24, 14, 0, 0x5EED)[::3]
    rng = np.random.default_rng(seed)
    t0 = rng.uniform(5.0, 60.0, len(rays))
    total, n_rays = 0, 0
    for ray, t in zip(rays, t0):
        prim, _, _, cnt = osc.trace_closest(ray[None, :], t, t + 1e-12)
        if prim[0] != rt.NONE:
            continue                                                               # (never seen; keeps the premise honest)
        n_rays += 1

        def decide(n, fl, kind):
            if fl & NO_BOX:
                return True
            return bool(orc.aabb_hit(np.concatenate([nodes[n]["bmin"], nodes[n]["bmax"]]), ray, t, t + 1e-12))
        seen = _walk(stream, root_off, decide)
        boxed = sum(1 for n in seen if not (int(nodes[n]["flags"]) & NO_BOX))
        assert boxed == int(cnt[0]), (boxed, cnt)
        total += boxed
    assert n_rays > len(rays) // 2 and total > n_rays                              # the walks did descend


def test_stream_of_lists_instances_and_odd_leaves():
    # list scene (no BVH: every record NO_BOX), instances (sub-ranges with their own END), mixed leaves
    s = SceneText(bvh_seed=1)
    m = s.lambertian_rgb(.5, .5, .5)
    a = s.sphere((0, 0, 0), 1.0, m)
    b = s.transformed(s.xform(translate(3, 0, 0)), s.sphere((0, 0, 0), 0.5, m))
    r = s.aarect(-1, 1, -1, 1, 2.0, m)
    hs = rt.HostScene.from_description(s.finish([a, b, r]))
    stream, root_off, node_off = hs.trace_stream()
    recs = _records(stream)
    assert [x["flags"] & NO_BOX for x in recs] == [1, 1, 1, 1]
    assert [x["kind"] for x in recs] == [SPH2, SPH2, LEAF, LEAF] and recs[-1]["node"] == END_NODE
    assert _walk(stream, root_off, lambda n, f, k: False) == [0, 1, 2]            # no boxes to fail: a list tests everything
    hs = rt.HostScene.builtin("cornell", bvh_seed=1)
    recs = _records(hs.trace_stream()[0])
    assert sum(1 for x in recs if x["node"] == END_NODE) == 1 + 4                  # scene array + 4 distinct cube BVHs


def test_stream_builder_rejects_what_upload_rejects():
    hs = rt.HostScene.builtin("sphere")
    blob = bytearray(hs.blob())
    n = rt._ffi.u64()
    lib = rt._ffi.hip()
    assert lib.rt_stream_build_host(bytes(blob[:100]), 100, None, 0, rt.C.byref(n), None, None) != 0
    blob[0] ^= 0xFF                                                              # magic
    assert lib.rt_stream_build_host(bytes(blob), len(blob), None, 0, rt.C.byref(n), None, None) != 0
    good = hs.blob()
    assert lib.rt_stream_build_host(good, len(good), None, 0, rt.C.byref(n), None, None) == 0
    small = np.zeros(16, np.uint8)
    assert lib.rt_stream_build_host(good, len(good), small.ctypes.data, small.nbytes, rt.C.byref(n), None, None) != 0
    assert b"too small" in lib.rt_last_error()


# ---- strict pruning: local boxes and the certificate, checked without a GPU ------------------------

def _soup_scene(seed, n_tri, extent, wrap=None):
    rng = np.random.default_rng(seed)
    s = SceneText(bvh_seed=4)
    m = s.lambertian_rgb(.5, .5, .5)
    centres = rng.uniform(-extent, extent, (n_tri, 3))
    verts, faces = [], []
    for k in range(n_tri):
        for _ in range(3):
            verts.append(tuple(centres[k] + rng.normal(size=3) * extent * 0.08))
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    me = s.mesh(verts, [(0, 0, 1)] * len(verts), [(0, 0)] * len(verts), faces, m)
    xf = s.xform(wrap) if wrap is not None else None
    ids = [s.transformed(xf, s.triangle(me, k)) if xf is not None else s.triangle(me, k) for k in range(n_tri)]
    ids.append(s.sphere((0.0, -extent * 3, 0.0), extent, m))          # a sphere leaf: its ancestors must not be pruned
    return rt.HostScene.from_description(s.finish(ids, build_bvh=True))


def _scene_for(name):
    if name == "soup":
        return _soup_scene(5, 300, 4.0)
    if name == "soup_scaled":
        return _soup_scene(6, 200, 3.0, wrap=scale(1.5, 0.75, 1.25))
    return rt.HostScene.builtin(name, bvh_seed=3)


@pytest.mark.parametrize("name", ["mesh", "soup", "soup_scaled", "cornell", "random", "simple_triangle"])
def test_local_boxes_contain_the_triangles_below_them(name):
    hs = _scene_for(name)
    h, nodes, geo = hs.header(), hs.nodes(), hs.geoms()
    info = hs.prune_info()
    stream, _, node_off = hs.trace_stream()
    raw = stream.tobytes()
    inf = float("inf")
    n_finite = 0
    for i in range(h.n_nodes):
        if nodes[i]["prim0"] != rt.NONE:
            continue
        lb = np.array(struct.unpack_from("<6f", raw, int(node_off[i]) + 32), np.float64).reshape(3, 2)
        sub = range(i + 1, int(nodes[i]["skip"]))
        leaves = [j for j in sub if nodes[j]["prim0"] != rt.NONE]
        prims = [int(p) for j in leaves for p in (nodes[j]["prim0"], nodes[j]["prim1"]) if p != rt.NONE]
        # coverable: triangles, spheres and moving spheres of the pruned space
        all_cov = bool(prims) and all(geo[p]["kind"] in (0, 1, 2) and geo[p]["xform"] == int(info["xform"]) for p in prims)
        if not np.isfinite(lb).all():
            assert (lb[:, 0] == -inf).all() and (lb[:, 1] == inf).all()      # never pruned
            # ... and there is a reason: pruning off, an instance range, or something that cannot be covered below
            assert (not info["on"]) or i >= h.root_end or not all_cov
            continue
        n_finite += 1
        assert info["on"] and i < h.root_end and all_cov
        pts = []
        for p in prims:
            d = geo["d"][p]
            if geo[p]["kind"] == 0:
                pts += [d[0:3], d[0:3] + d[3:6], d[0:3] + d[6:9]]
            elif geo[p]["kind"] == 1:
                pts += [d[0:3] - abs(d[3]), d[0:3] + abs(d[3])]
            else:   # moving sphere: centre at both ends of the camera's shutter
                for tm in (h.camera.time0, h.camera.time1):
                    c = d[0:3] + (d[3:6] - d[0:3]) * ((tm - d[6]) / (d[7] - d[6]))
                    pts += [c - abs(d[8]), c + abs(d[8])]
        verts = np.array(pts)
        assert (verts >= lb[:, 0]).all() and (verts <= lb[:, 1]).all()        # the fp32 box is rounded outward
        assert (verts.min(0) - lb[:, 0] <= 1e-6 * (1 + np.abs(lb[:, 0]))).all() and (lb[:, 1] - verts.max(0) <= 1e-6 * (1 + np.abs(lb[:, 1]))).all()  # and tight
    assert n_finite == int(info["prunable_nodes"])
    if name in ("mesh", "soup", "soup_scaled"):
        assert info["on"] and n_finite > 0.4 * (nodes["prim0"] == rt.NONE).sum()
    if name == "random":
        assert info["on"] and n_finite == (nodes["prim0"] == rt.NONE).sum()    # spheres and moving spheres: every inner node
    if name in ("cornell", "simple_triangle"):
        assert not info["on"]                                                  # instances only / 10 triangles


def test_certificate_is_conservative_on_the_pagoda():
    # A ray certified "clean" (code 1) must satisfy, for EVERY triangle of the pruned space, what the proof needs:
    # |d . n^| >= 2^-25 |d|_inf, or the origin at least delta_B away from the triangle's plane.
    hs = rt.HostScene.builtin("mesh", bvh_seed=3)
    info = hs.prune_info()
    geo = hs.geoms()
    tri = geo[(geo["kind"] == 0) & (geo["xform"] == int(info["xform"]))]["d"]
    v0, e1, e2 = tri[:, 0:3], tri[:, 3:6], tri[:, 6:9]
    n = np.cross(e1, e2); nh = n / np.linalg.norm(n, axis=1)[:, None]
    rng = np.random.default_rng(11)
    N = 1500
    o = rng.uniform(-40, 40, (N, 3)); d = rng.normal(size=(N, 3))
    # adversarial: direction inside a triangle's plane up to a perturbation eta, origin eps off that plane
    k = rng.integers(0, len(tri), N)
    a, b = rng.normal(size=N), rng.normal(size=N)
    eta = 10.0 ** rng.uniform(-12, -4, N) * rng.choice([-1, 1], N)
    off = 10.0 ** rng.uniform(-9, 0, N) * rng.choice([-1, 1], N)
    d_adv = a[:, None] * e1[k] + b[:, None] * e2[k]
    d_adv = d_adv / np.abs(d_adv).max(1)[:, None] + eta[:, None] * nh[k]
    o_adv = v0[k] + rng.normal(size=(N, 1)) * e1[k] * 3 + rng.normal(size=(N, 1)) * e2[k] * 3 + off[:, None] * nh[k]
    O, D = np.concatenate([o, o_adv]), np.concatenate([d, d_adv])
    _, code_raw = hs.prune_info(np.concatenate([O, D], axis=1))
    Dinf = np.abs(D).max(1); Oinf = np.abs(O).max(1)
    V = np.abs(v0).max(); E = max(np.abs(e1).sum(1).max(), np.abs(e2).sum(1).max())
    delta_b = 4.0 / info["shape_min"] * 2.0 ** -23 * (Oinf + V + E) * (1 + 2.0 ** -10)
    assert (info["delta_coef"] * (Oinf + info["radius"]) >= delta_b * (1 + 2.0 ** -10)).all()   # the device's delta covers the proof's
    clean = np.nonzero(code_raw == 1)[0]
    for i in clean:
        mu = np.abs(nh @ D[i]); lam = np.abs(((O[i] - v0) * nh).sum(1))
        bad = (mu < 2.0 ** -25 * Dinf[i]) & (lam < delta_b[i])
        assert not bad.any(), (i, int(bad.sum()))
    assert (code_raw[:N] == 1).mean() > 0.99                 # random rays are practically always certified
    in_plane = (np.abs(eta) < 2.0 ** -26) & (np.abs(off) < 1e-6)
    assert in_plane.sum() > 20 and (code_raw[N:][in_plane] != 1).all()   # rays lying in a triangle's plane never are
    assert (code_raw[N:] == 2).sum() > 0                     # and some of them walk with protected leaves


def test_pruned_walk_over_the_stream_finds_the_oracles_hits():
    # numpy emulation of the production walk on the CPU: pre-order records, the reference's per-axis test on the blob's boxes,
    # PLUS skipping every inner record whose LOCAL box (from the stream) the ray's line misses by more than tau.  The hits must
    # be the oracle's, with a small fraction of the records visited.
    hs = rt.HostScene.builtin("mesh", bvh_seed=3)
    osc = orc.OracleScene(hs.describe())
    h, nodes, geo = hs.header(), hs.nodes(), hs.geoms()
    info = hs.prune_info()
    stream, _, node_off = hs.trace_stream()
    raw = stream.tobytes()
    rays = osc.primary_rays(48, 27, 0, 0x5EED)[::3]
    n = len(rays)
    o, d = rays[:, :3], rays[:, 3:6]
    inv = 1.0 / d
    ol = o - 5.0                                           # the pagoda's transform: translate(5, 5, 5), main.rs:816
    tau = info["tau_coef"] * (np.abs(ol).max(1) + info["radius"])
    best = np.full(n, 1e5); prim = np.full(n, rt.NONE, np.int64)
    wait = np.zeros(n, np.int64)
    visited = 0; full = 0
    skip = nodes["skip"].astype(np.int64)
    for i in range(h.root_end):
        act = wait <= i
        if not act.any():
            continue
        nd = nodes[i]
        t0 = (nd["bmin"] - o) * inv; t1 = (nd["bmax"] - o) * inv
        ok = (np.minimum(t0, t1).max(1) <= best) & (np.maximum(t0, t1).min(1) >= 0.001)     # bounding_box.rs:27-46 (no NaN here)
        if nd["prim0"] == rt.NONE:
            lb = np.array(struct.unpack_from("<6f", raw, int(node_off[i]) + 32), np.float64).reshape(3, 2)
            if np.isfinite(lb).all():
                a0 = (lb[:, 0] - tau[:, None] - ol) * inv; a1 = (lb[:, 1] + tau[:, None] - ol) * inv
                b0 = (lb[:, 0] + tau[:, None] - ol) * inv; b1 = (lb[:, 1] - tau[:, None] - ol) * inv
                near = np.minimum(np.minimum(a0, a1), np.minimum(b0, b1)).max(1); far = np.maximum(np.maximum(a0, a1), np.maximum(b0, b1)).min(1)
                ok &= ~(near > far)                                                        # the line misses the inflated local box
        visited += int(act.sum())
        p = act & ok
        wait[act & ~ok] = skip[i]
        for pid in (int(nd["prim0"]), int(nd["prim1"])):
            if pid == rt.NONE or not p.any():
                continue
            g = geo[pid]
            if g["kind"] == 0:                                                             # mesh.rs:55-81
                v0, e1, e2 = g["d"][:3], g["d"][3:6], g["d"][6:9]
                hh = np.cross(d, e2); det = hh @ e1
                with np.errstate(all="ignore"):
                    f = 1.0 / det; s = ol - v0; u = f * (s * hh).sum(1); q = np.cross(s, e1); v = f * (d * q).sum(1); t = f * (q @ e2)
                acc = p & ~((det > -2.2e-16) & (det < 2.2e-16)) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 2.2e-16) & (t <= best) & (t > 0.001)
            elif g["kind"] == 1:                                                           # shape.rs:93-121
                c, r = g["d"][:3], g["d"][3]; oc = o - c; a = (d * d).sum(1); hb = (oc * d).sum(1); cc = (oc * oc).sum(1) - r * r
                disc = hb * hb - a * cc
                with np.errstate(all="ignore"):
                    sq = np.sqrt(disc); ta = (-hb - sq) / a; tb = (-hb + sq) / a
                first = ~((ta < 0.001) | (best < ta)); t = np.where(first, ta, tb)
                acc = p & (disc >= 0) & ~((t < 0.001) | (best < t))
            else:
                continue
            best = np.where(acc, t, best); prim = np.where(acc, pid, prim)
    want_prim, _, want_t, cnt = osc.trace_closest(rays, 0.001, 1e5)
    obj = np.where(prim == rt.NONE, rt.NONE, hs.prims()["object_id"][np.minimum(prim, h.n_prims - 1)])    # the oracle reports object ids
    assert np.array_equal(obj.astype(np.uint32), want_prim)
    hit = want_prim != rt.NONE
    assert hit.sum() > n // 2 and np.allclose(best[hit], want_t[hit], rtol=1e-9)
    assert visited < 0.05 * int(cnt[0])                    # a few percent of the reference's box tests
