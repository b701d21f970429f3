"""Offline costing of ray regroupings from the diagnostic build's step log (RT_DIAG_LOG=<dir>, make diag): lane_log.py <dir>.

Every entry is one record step of a sampled wave: (wave, stream offset, mask of the lanes taking part).  A ray's visited records depend on
the ray alone, never on its wave-mates, so the log holds the exact visited set of every sampled ray and the step count of ANY wave made of
those rays is the size of the union of their visited offsets.  Per launch this prints
  * lane use and the share of steps by number of participating lanes,
  * the ray-stream bound: steps if, inside blocks of G consecutive waves, the rays standing at a record were always packed into full waves
    (sum over records of ceil(rays at the record / 64)) -- the floor of every regrouping scheme confined to such a block,
  * tail compaction: waves cut once at most T of their rays are still walking, the left-over rays of consecutive waves merged into full
    waves that resume where they stopped (cost: union of the merged rays' remaining offsets),
  * the per-lane floor (ray steps / 64),
  * staged walks: the stream cut into K offset ranges of equal work, stage s = the rays that visit range s, compacted in queue order into
    full waves that walk that range only (any offset range is a valid stage of the threaded pre-order walk).  Printed: steps / today's,
    waves started per ray-wave of today (each start pays a set-up), and the total in units of today's steps with a set-up priced at 15 or
    45 steps (today's set-up is 15 % of a 257-step wave: 45 steps; reloading saved filter state instead of recomputing it: about 15).
Sampled waves come in groups of RT_DIAG_LOG_GROUP consecutive waves (default 64: 4 096 rays adjacent in the sorted queue); the END record
(last offset of the launch, which every ray passes) is left out."""
import glob, os, sys
import numpy as np


GROUP = int(os.environ.get("RT_DIAG_LOG_GROUP", "64"))


def popcount64(m):
    return np.unpackbits(m.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1).astype(np.int64)


def analyse(path):
    raw = np.fromfile(path, dtype=np.uint32).reshape(-1, 4)
    if len(raw) == 0:
        return None
    wave = raw[:, 0].astype(np.int64); off = raw[:, 1].astype(np.int64)
    kind = off & 3; off = off & ~np.int64(15)          # 0 inner, 1 triangle leaf, 2 generic leaf, 3 sphere leaf
    mask = (raw[:, 3].astype(np.uint64) << np.uint64(32)) | raw[:, 2].astype(np.uint64)
    keep = off != off.max()                            # the END record closes every walk: not a step that regrouping could share or save
    wave, off, mask, kind = wave[keep], off[keep], mask[keep], kind[keep]
    order = np.argsort(wave, kind="stable")          # a wave's entries keep their step order
    wave, off, mask, kind = wave[order], off[order], mask[order], kind[order]
    act = popcount64(mask)
    steps = len(wave)
    out = {"file": os.path.basename(path), "sampled_waves": int(len(np.unique(wave))), "steps": steps,
           "lane_use": float(act.sum() / (64.0 * steps)), "per_lane_floor_steps": float(act.sum() / 64.0)}
    per_wave = np.bincount(np.unique(wave, return_inverse=True)[1])
    mean = per_wave.mean()
    out["steps_per_wave"] = {"mean": round(float(mean), 1), "p50": int(np.percentile(per_wave, 50)), "p90": int(np.percentile(per_wave, 90)), "p99": int(np.percentile(per_wave, 99)),
                             "p99.9": int(np.percentile(per_wave, 99.9)), "max": int(per_wave.max()),
                             "share_of_steps_in_waves_over_4x_mean": round(float(per_wave[per_wave > 4 * mean].sum() / per_wave.sum()), 4),
                             "waves_over_4x_mean": round(float((per_wave > 4 * mean).mean()), 5),
                             "share_of_steps_in_waves_over_10x_mean": round(float(per_wave[per_wave > 10 * mean].sum() / per_wave.sum()), 4),
                             "waves_over_10x_mean": round(float((per_wave > 10 * mean).mean()), 5)}
    edges = [0, 1, 2, 3, 5, 9, 17, 33, 65]
    names = ["0", "1", "2", "3-4", "5-8", "9-16", "17-32", "33-64"]
    h, _ = np.histogram(act, bins=edges)
    out["steps_by_lanes"] = {n: round(float(c) / steps, 4) for n, c in zip(names, h)}
    # ray-stream bound inside blocks of G consecutive waves
    bound = {}
    for g in (1, 4, 16, 64):
        key = (wave // g) * (1 << 32) + off
        _, inv = np.unique(key, return_inverse=True)
        rays_at = np.bincount(inv, weights=act.astype(np.float64))
        bound[str(g)] = round(float(np.ceil(np.maximum(rays_at, 1.0) / 64.0).sum() / steps), 4)
    out["ray_stream_bound_steps_ratio_by_block_waves"] = bound
    # live lanes per step (a lane is live while it still takes part in a later step) and tail compaction
    starts = np.flatnonzero(np.r_[True, wave[1:] != wave[:-1]]); ends = np.r_[starts[1:], steps]
    live = np.empty(steps, dtype=np.int64)
    live_mask = np.empty(steps, dtype=np.uint64)
    for a, b in zip(starts, ends):
        lm = np.bitwise_or.accumulate(mask[a:b][::-1])[::-1]
        live_mask[a:b] = lm
    live = popcount64(live_mask)
    h, _ = np.histogram(live, bins=edges)
    out["steps_by_live_lanes"] = {n: round(float(c) / steps, 4) for n, c in zip(names, h)}
    tails = {}
    for T in (4, 8, 16, 32):
        total = 0
        group = wave // GROUP
        for gid in np.unique(group):
            sel = np.flatnonzero(group == gid)
            w_ids = np.unique(wave[sel])
            pend_offs, pend_lanes = [], 0
            for w in w_ids:
                idx = sel[wave[sel] == w]
                lv = live[idx]
                cut = np.flatnonzero(lv <= T)
                c = cut[0] if len(cut) else len(idx)
                total += c                                   # head: walked as today
                if c < len(idx):
                    n_left = int(lv[c])
                    if pend_lanes + n_left > 64:
                        total += len(np.unique(np.concatenate(pend_offs))); pend_offs, pend_lanes = [], 0
                    pend_offs.append(off[idx[c:]]); pend_lanes += n_left
            if pend_offs:
                total += len(np.unique(np.concatenate(pend_offs)))
        tails[str(T)] = round(float(total) / steps, 4)
    out["tail_compaction_steps_ratio_by_threshold"] = tails
    # runs: maximal sequences of consecutive steps of a wave in which the same SINGLE lane takes part (a lone ray walking a subtree by itself)
    single = act == 1
    brk = np.r_[True, (wave[1:] != wave[:-1]) | (mask[1:] != mask[:-1])]
    rid = np.cumsum(brk) - 1
    rlen = np.bincount(rid); rsingle = np.zeros(len(rlen), bool); rsingle[rid[single]] = True
    first_off = np.zeros(len(rlen), np.int64); last_off = np.zeros(len(rlen), np.int64)
    first_off[rid[::-1]] = off[::-1]; last_off[rid] = off
    leafs = np.bincount(rid, weights=(kind != 0).astype(np.float64))
    runs = {}
    for L in (1, 2, 4, 8, 16, 32, 64):
        sel = rsingle & (rlen >= L)
        n = rlen[sel].sum()
        span = (last_off[sel] - first_off[sel]).astype(np.float64)
        runs[str(L)] = {"share_of_steps": round(float(n) / steps, 4), "runs": int(sel.sum()), "mean_len": round(float(rlen[sel].mean()) if sel.any() else 0.0, 1),
                        "mean_span_bytes": round(float(span.mean()) if sel.any() else 0.0, 0), "median_span_bytes": round(float(np.median(span)) if sel.any() else 0.0, 0),
                        "leaf_share": round(float(leafs[sel].sum()) / max(float(n), 1.0), 3)}
    out["single_lane_runs_by_min_length"] = runs
    out["kind_share_all_steps"] = {k: round(float((kind == i).mean()), 4) for i, k in enumerate(["inner", "tri_leaf", "generic_leaf", "sphere_leaf"])}
    out["kind_share_single_lane_steps"] = {k: round(float((kind[single] == i).mean()), 4) for i, k in enumerate(["inner", "tri_leaf", "generic_leaf", "sphere_leaf"])}
    # staged walks over K offset ranges of equal work
    bits = np.unpackbits(mask.view(np.uint8).reshape(-1, 8), axis=1, bitorder="little").astype(bool)   # [step, lane]
    st_idx, lane = np.nonzero(bits)
    ray = wave[st_idx] * 64 + lane; roff = off[st_idx]
    blk = wave[st_idx] // GROUP
    staged = {}
    waves_today = len(np.unique(wave))
    for K in (2, 4, 8, 16, 32, 64, 256):
        cuts = np.quantile(roff, np.linspace(0, 1, K + 1)[1:-1])
        stage = np.searchsorted(cuts, roff, side="right")
        # participation: distinct (block, stage, ray), ranked in ray order inside (block, stage)
        _, gid = np.unique(blk * K + stage, return_inverse=True)            # dense (block, stage) ids
        pk = gid * (1 << 32) + ray
        upk, inv = np.unique(pk, return_inverse=True)
        grp = upk >> 32
        first = np.r_[True, grp[1:] != grp[:-1]]
        rank = np.arange(len(upk)) - np.maximum.accumulate(np.where(first, np.arange(len(upk)), 0))
        newwave = rank // 64
        n_starts = int(len(np.unique(grp * (1 << 20) + newwave)))
        wk = (grp[inv] * (1 << 20) + newwave[inv]) * (1 << 24) + (roff >> 4)   # records are 16-byte aligned, the stream is < 256 MB
        n_steps = int(len(np.unique(wk)))
        staged[str(K)] = {"steps": round(n_steps / steps, 4), "starts_per_wave": round(n_starts / waves_today, 3),
                          "stages_per_ray": round(len(upk) / len(np.unique(ray)), 2),
                          "total_at_15": round((n_steps + 15.0 * n_starts) / (steps + 15.0 * waves_today), 4),
                          "total_at_45": round((n_steps + 45.0 * n_starts) / (steps + 45.0 * waves_today), 4)}
    out["staged"] = staged
    return out


def main():
    d = sys.argv[1]
    for f in sorted(glob.glob(os.path.join(d, "launch_*.bin"))):
        r = analyse(f)
        if r is None:
            continue
        print(r["file"], "waves", r["sampled_waves"], "steps", r["steps"], "lane use %.3f" % r["lane_use"], "per-lane floor %.3f" % (r["per_lane_floor_steps"] / r["steps"]))
        print("   steps per wave:", r["steps_per_wave"])
        print("   steps by participating lanes:", r["steps_by_lanes"])
        print("   steps by live lanes:         ", r["steps_by_live_lanes"])
        print("   ray-stream bound (steps / today's) by block of G waves:", r["ray_stream_bound_steps_ratio_by_block_waves"])
        print("   tail compaction (steps / today's) by live-lane threshold:", r["tail_compaction_steps_ratio_by_threshold"])
        print("   record kinds, all steps:", r["kind_share_all_steps"], " single-lane steps:", r["kind_share_single_lane_steps"])
        for L, v in r["single_lane_runs_by_min_length"].items():
            print("   single-lane runs of >= %-2s steps:" % L, v)
        for K, v in r["staged"].items():
            print("   staged K=%-3s" % K, v)


if __name__ == "__main__":
    main()
