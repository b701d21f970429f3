"""Offline costing of INNER-RECORD ELISION from the diagnostic build's step log: elide_eval.py <log dir> [scene] [bvh_seed].

The production trace stream may leave out any inner record when the boxes nest (rt_device.hip, prepare_scene: a ray then only visits MORE than
the reference, and every leaf re-decides with its own box).  Today only the never-prunable ancestors of a ground sphere are left out.  A wave pays
one step for an inner record and then, if ANY of its lanes passes, the steps of both children; without the record it pays the children always:
leaving record r out saves pass_r steps and costs fail_r * (F_r - 1), F_r = the records that stand in for it (2, more if children are left out
too), where pass_r / fail_r count the sampled waves that visited r and went on into it / skipped it.  The log holds exactly that: a wave's next
logged offset after an inner record is off + 64 iff it passed.

Prints, per launch and for the pass: the share of inner steps by wave-level pass rate of their record, and the step count (inner step = 1, leaf
step = LEAF_COST) under
  * the best set found bottom-up from the pooled counts of all launches (optimistic: tuned on the data it is scored on),
  * geometric rules that need no profile: records whose subtree holds at most K leaves; records whose box area is at least Q of the parent's.
"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as rt

LEAF_COST = float(os.environ.get("RT_ELIDE_LEAF_COST", "1.5"))
REC_SIZE = {0: 64, 1: 144, 2: 48, 3: 112}


def parse_stream(hs):
    stream, root, node_off = hs.trace_stream(production=True)
    w = stream.view(np.uint32)
    offs, kinds, skips, nodes = [], [], [], []
    off = root
    while True:
        skip, rec = int(w[off // 4 + 6]), int(w[off // 4 + 7])
        kind = (rec & 12) >> 2
        node = rec >> 8
        if node == 0xFFFFFF:
            end = off
            break
        offs.append(off); kinds.append(kind); skips.append(skip); nodes.append(node)
        off += REC_SIZE[kind]
    offs = np.array(offs, np.int64); kinds = np.array(kinds, np.int64); skips = np.array(skips, np.int64); nodes = np.array(nodes, np.int64)
    return offs, kinds, skips, nodes, end


def main():
    d = sys.argv[1]
    scene = sys.argv[2] if len(sys.argv) > 2 else "mesh"
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    hs = rt.HostScene.builtin(scene, bvh_seed=seed, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area"))
    offs, kinds, skips, nodes, end = parse_stream(hs)
    n = len(offs)
    idx_of = {int(o): i for i, o in enumerate(offs)}
    idx_of[int(end)] = n
    skip_idx = np.array([idx_of.get(int(s), n) for s in skips], np.int64)
    # children lists (records between i+1 and skip_idx[i], hopping by skip)
    children = [[] for _ in range(n)]
    parent = np.full(n, -1, np.int64)
    for i in range(n):
        if kinds[i] != 0:
            continue
        c = i + 1
        while c < skip_idx[i]:
            children[i].append(c); parent[c] = i
            c = int(skip_idx[c])
    leaves_under = np.zeros(n, np.int64)
    for i in range(n - 1, -1, -1):
        leaves_under[i] = 1 if kinds[i] != 0 else sum(leaves_under[c] for c in children[i])
    nd = hs.production_nodes()[0]     # the array the production stream's node words index (the blob's nodes + a re-bracketed scene-level range)
    ext = np.maximum(nd["bmax"][nodes] - nd["bmin"][nodes], 0.0)
    area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
    print(f"production stream: {n} records, {int((kinds == 0).sum())} inner, {int((kinds != 0).sum())} leaves, LEAF_COST {LEAF_COST}")

    files = sorted(glob.glob(os.path.join(d, "launch_*.bin")))
    V = np.zeros(n); PASS = np.zeros(n)
    per_launch = []
    if d.endswith(".npz"):     # counts saved by an earlier run (RT_ELIDE_SAVE): rules can be tried without the logs
        z = np.load(d)
        for name, v, p in zip(z["names"], z["v"], z["p"]):
            per_launch.append((str(name), v, p)); V += v; PASS += p
        files = []
    for f in files:
        raw = np.fromfile(f, dtype=np.uint32).reshape(-1, 4)
        if len(raw) == 0:
            continue
        wave = raw[:, 0].astype(np.int64); o = raw[:, 1].astype(np.int64)
        off = o & ~np.int64(15)
        order = np.argsort(wave, kind="stable")
        wave, off = wave[order], off[order]
        same = np.r_[wave[1:] == wave[:-1], False]
        nxt = np.r_[off[1:], -1]
        ridx = np.array([idx_of.get(int(x), -1) for x in np.unique(off)], np.int64)
        u, inv = np.unique(off, return_inverse=True)
        rec = ridx[inv]
        ok = (rec >= 0) & (rec < n)
        v = np.bincount(rec[ok], minlength=n).astype(np.float64)
        passed = ok & same & (nxt == off + 64) & (kinds[np.clip(rec, 0, n - 1)] == 0)
        p = np.bincount(rec[passed], minlength=n).astype(np.float64)
        # an inner record that is a wave's LAST logged step has no successor: counted as a fail (it is one: the wave went to END or beyond)
        per_launch.append((os.path.basename(f), v, p))
        V += v; PASS += p

    if os.environ.get("RT_ELIDE_SAVE"):
        np.savez_compressed(os.environ["RT_ELIDE_SAVE"], names=np.array([x[0] for x in per_launch]), v=np.array([x[1] for x in per_launch], np.float32),
                            p=np.array([x[2] for x in per_launch], np.float32))
    inner = kinds == 0
    cost = np.where(inner, 1.0, LEAF_COST)

    def steps_under(E, v, p):
        """cost-weighted steps of a launch with visit / pass counts (v, p) when the records in E have no record"""
        F = np.zeros(n)
        for i in range(n - 1, -1, -1):
            if E[i]:
                F[i] = sum((F[c] if E[c] else cost[c]) for c in children[i])
        base = float((v * cost)[~E].sum())
        fail = v - p
        return base + float((fail * F)[E].sum())

    def best_set(v, p):
        E = np.zeros(n, bool)
        F = np.zeros(n)
        for i in range(n - 1, -1, -1):
            if not inner[i] or not children[i]:
                continue
            Fi = sum((F[c] if E[c] else cost[c]) for c in children[i])
            gain = v[i] - (v[i] - p[i]) * Fi            # saves its own step on every visit, pays the stand-ins on every fail
            if gain > 0:
                E[i] = True; F[i] = Fi
        return E

    none = np.zeros(n, bool)
    Ebest = best_set(V, PASS)
    rules = {"best (pooled counts)": Ebest}
    for K in (2, 3, 4, 8):
        rules[f"subtree <= {K} leaves"] = inner & (leaves_under <= K)
    par_area = np.where(parent >= 0, area[np.clip(parent, 0, n - 1)], np.inf)
    for Q in (0.9, 0.8, 0.7):
        rules[f"area >= {Q} of parent"] = inner & (area >= Q * par_area) & (parent >= 0)
    for K, Q in ((4, 0.8), (8, 0.8)):
        rules[f"subtree <= {K} leaves or area >= {Q} of parent"] = inner & ((leaves_under <= K) | ((area >= Q * par_area) & (parent >= 0)))

    print("\nshare of INNER steps by the wave-level pass rate of their record (pooled over the pass's launches):")
    rate = np.where(V > 0, PASS / np.maximum(V, 1), 0.0)
    edges = [0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0001]
    tot_inner = V[inner].sum()
    for a, b in zip(edges[:-1], edges[1:]):
        sel = inner & (rate >= a) & (rate < b)
        print(f"  pass rate [{a:.1f}, {min(b, 1.0):.1f}{']' if b > 1 else ')'}: {V[sel].sum() / tot_inner:6.1%} of inner steps, {int(sel.sum())} records")
    print(f"  inner steps {tot_inner / V.sum():.1%} of all logged steps; overall wave-level pass rate of inner records {PASS[inner].sum() / tot_inner:.3f}")
    print("  by leaves under the record:")
    for a, b in ((2, 2), (3, 4), (5, 8), (9, 32), (33, 1024), (1025, 1 << 30)):
        sel = inner & (leaves_under >= a) & (leaves_under <= b)
        if V[sel].sum() > 0:
            print(f"    {a:>5}-{b if b < (1 << 30) else 'up':<6}: {V[sel].sum() / tot_inner:6.1%} of inner steps, pass rate {PASS[sel].sum() / V[sel].sum():.3f}")

    print("\ncost-weighted steps relative to today's stream:")
    names = list(rules)
    print("  %-28s" % "launch" + "".join("%-14s" % ("r%d" % i) for i in range(len(names))))
    for i, k in enumerate(names):
        print(f"    r{i} = {k}: {int(rules[k].sum())} records left out")
    tot0 = 0.0; tots = np.zeros(len(names))
    for name, v, p in per_launch:
        s0 = steps_under(none, v, p)
        row = [steps_under(rules[k], v, p) for k in names]
        tot0 += s0; tots += np.array(row)
        print("  %-28s" % name + "".join("%-14.3f" % (x / s0) for x in row))
    print("  %-28s" % "pass (sampled waves)" + "".join("%-14.3f" % (x / tot0) for x in tots))


if __name__ == "__main__":
    main()
