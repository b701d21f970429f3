"""k_trace against the oracle on adversarial ray sets over the pagoda (promoted from tools/stress_parity.py at a bounded size):
random rays, rays leaving random surface points, rays lying in triangle planes (eta from 0 to 1e-2 off the plane, origins 0 to 1 off
it) and surface rays with tangent directions.  Bit-exact (primitive, t) from the production kernel and from the counting kernel,
equal reference-order test counters, zero contradicted filter / pruning decisions."""
import numpy as np
import pytest

import oracle_ffi as orc
import raytracing_amd as rt

pytestmark = pytest.mark.gpu
N = 3000


def _sets(hs, n):
    geo = hs.geoms(); tri = geo[geo["kind"] == 0]["d"]
    v0, e1, e2 = tri[:, 0:3], tri[:, 3:6], tri[:, 6:9]
    rng = np.random.default_rng(77)

    def rays_of(o, d):
        r = np.zeros((len(o), 7)); r[:, :3] = o; r[:, 3:6] = d; return r
    k = rng.integers(0, len(tri), n)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n); fl = u + v > 1; u[fl], v[fl] = 1 - u[fl], 1 - v[fl]
    surf = v0[k] + u[:, None] * e1[k] + v[:, None] * e2[k] + 5.0          # the pagoda's triangles sit under translate(5, 5, 5)
    nh = np.cross(e1[k], e2[k]); nh /= np.linalg.norm(nh, axis=1)[:, None]
    a, b = rng.normal(size=n), rng.normal(size=n)
    eta = 10.0 ** rng.uniform(-16, -2, n) * rng.choice([-1, 1], n); eta[::5] = 0
    off = 10.0 ** rng.uniform(-14, 0, n) * rng.choice([-1, 1], n); off[::7] = 0
    dpl = a[:, None] * e1[k] + b[:, None] * e2[k]; dpl = dpl / np.abs(dpl).max(1)[:, None] + eta[:, None] * nh
    opl = v0[k] + rng.normal(size=(n, 1)) * e1[k] * 5 + rng.normal(size=(n, 1)) * e2[k] * 5 + off[:, None] * nh + 5.0
    return {"random": rays_of(rng.uniform(-60, 60, (n, 3)), rng.normal(size=(n, 3))),
            "surface": rays_of(surf, rng.normal(size=(n, 3))),
            "in-plane": rays_of(opl, dpl),
            "surface-tangent": rays_of(surf, dpl)}


@pytest.mark.parametrize("which", ["random", "surface", "in-plane", "surface-tangent"])
def test_adversarial_rays_bit_exact(gpu_ctx, which):
    hs = rt.HostScene.builtin("mesh", bvh_seed=3)
    osc = orc.OracleScene(hs.describe())
    gs = gpu_ctx.upload(hs)
    r = _sets(hs, N)[which]
    o = osc.trace_closest(r, 0.001, 1e5)
    g = gs.trace_closest(r, 0.001, 1e5)
    c = gs.trace_closest(r, 0.001, 1e5, counters=True)
    for got in (g, c):
        assert np.array_equal(got[0], o[0]) and np.array_equal(got[1], o[1])
        assert np.array_equal(got[2].view(np.uint64), o[2].view(np.uint64))
    assert list(map(int, c[3][:3])) == list(map(int, o[3][:3]))
    assert int(c[3][3]) == 0, "filter / pruning decisions contradicted by the exact walk"
    assert (o[0] != rt.NONE).sum() > 0
    gs.close()
