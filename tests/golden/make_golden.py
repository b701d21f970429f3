"""Generates tests/golden/*.npz from the oracle (the reference is Rust and cannot run here; these
vectors freeze the oracle's outputs so that later edits to oracle/ or to the HIP path are caught
even if both drift together).  Run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi as orc
import raytracing_amd as rt

CASES = [("simple_triangle", False, 48, 27), ("random", False, 48, 27), ("cornell", False, 48, 27), ("simple", False, 48, 27),
         ("sphere", False, 32, 18), ("mesh", False, 32, 18)]
BVH_SEED, RAY_SEED, RAD_SEED = 5, 0x5EED, 77


def main():
    out = {}
    for name, legacy, w, h in CASES:
        hs = rt.HostScene.builtin(name, bvh_seed=BVH_SEED, legacy_bvh=legacy)
        osc = orc.OracleScene(hs.describe())
        rays = osc.primary_rays(w, h, 0, RAY_SEED)
        prim, inst, t, cnt = osc.trace_closest(rays, 0.001, 100000.0)
        out[f"{name}_wh"] = np.array([w, h], np.uint32)
        out[f"{name}_rays"] = rays; out[f"{name}_prim"] = prim; out[f"{name}_inst"] = inst; out[f"{name}_t"] = t; out[f"{name}_cnt"] = cnt
        rng = np.random.default_rng(1)
        n = 24 if name == "mesh" else 64
        picks = np.stack([rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 50, n)], 1).astype(np.uint32)
        out[f"{name}_picks"] = picks
        for integ in range(4):
            both = [osc.sample_radiance_sig(w, h, int(x), int(y), int(s), 5, integ, RAD_SEED) for x, y, s in picks]
            out[f"{name}_rad{integ}"] = np.array([b[0] for b in both])
            out[f"{name}_sig{integ}"] = np.array([b[1] for b in both], np.uint64)   # path signatures: the discrete decisions each path took
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"))


if __name__ == "__main__":
    main()
