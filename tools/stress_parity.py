"""One-off stress of k_trace against the oracle on the pagoda: random, surface-origin and in-plane rays (not a pytest: ~1 min of CPU)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi as orc
import raytracing_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
hs = rt.HostScene.builtin("mesh", bvh_seed=3, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area"))
osc = orc.OracleScene(hs.describe())
ctx = rt.Context(0); gs = ctx.upload(hs)
geo = hs.geoms(); tri = geo[geo["kind"] == 0]["d"]
v0, e1, e2 = tri[:, 0:3], tri[:, 3:6], tri[:, 6:9]
rng = np.random.default_rng(77)
def rays_of(o, d):
    r = np.zeros((len(o), 7)); r[:, :3] = o; r[:, 3:6] = d; return r
k = rng.integers(0, len(tri), n)
u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n); fl = u + v > 1; u[fl], v[fl] = 1 - u[fl], 1 - v[fl]
surf = v0[k] + u[:, None] * e1[k] + v[:, None] * e2[k] + 5.0
nh = np.cross(e1[k], e2[k]); nh /= np.linalg.norm(nh, axis=1)[:, None]
a, b = rng.normal(size=n), rng.normal(size=n)
eta = 10.0 ** rng.uniform(-16, -2, n) * rng.choice([-1, 1], n); eta[::5] = 0
off = 10.0 ** rng.uniform(-14, 0, n) * rng.choice([-1, 1], n); off[::7] = 0
dpl = a[:, None] * e1[k] + b[:, None] * e2[k]; dpl = dpl / np.abs(dpl).max(1)[:, None] + eta[:, None] * nh
opl = v0[k] + rng.normal(size=(n, 1)) * e1[k] * 5 + rng.normal(size=(n, 1)) * e2[k] * 5 + off[:, None] * nh + 5.0
sets = {"random": rays_of(rng.uniform(-60, 60, (n, 3)), rng.normal(size=(n, 3))),
        "surface": rays_of(surf, rng.normal(size=(n, 3))),
        "in-plane": rays_of(opl, dpl),
        "surface-tangent": rays_of(surf, dpl)}
bad = 0
for name, r in sets.items():
    t0 = time.time(); o = osc.trace_closest(r, 0.001, 1e5); t1 = time.time()
    g = gs.trace_closest(r, 0.001, 1e5); c = gs.trace_closest(r, 0.001, 1e5, counters=True)
    same = np.array_equal(g[0], o[0]) and np.array_equal(g[2].view(np.uint64), o[2].view(np.uint64)) and np.array_equal(c[0], o[0]) and np.array_equal(c[2].view(np.uint64), o[2].view(np.uint64))
    print(f"{name}: {len(r)} rays, hits {(o[0] != rt.NONE).sum()}, bit-exact {same}, counters equal {list(map(int, c[3][:3])) == list(map(int, o[3][:3]))}, violations {int(c[3][3])}, uncertified {int(c[3][4])}, oracle {t1 - t0:.1f} s", flush=True)
    bad += (not same) or int(c[3][3]) != 0
sys.exit(1 if bad else 0)
