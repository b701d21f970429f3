"""PDF set of renderer/src/pdf.rs, by itself: the oracle's MixturePdf (Uniform arm, pdf.rs:185-224) against its
definition on the CPU, and the device PDFs + mixture (rt_pdf_eval) against the oracle on the GPU.  Parity unpinned by the reference
(it holds no PDF test); value() involves only IEEE arithmetic and must agree in bits, generate() goes through sin/cos/acos of two
different libm's and is compared to 1e-12."""
import numpy as np
import pytest

import oracle_ffi as orc
import raytracing_amd as rt


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1)[:, None]


def _descs(rng, n, kind):
    d = np.zeros(n, rt.PDF_DESC)
    d["kind"] = kind
    d["a"] = _unit(rng, n)
    if kind == rt.PDF_UNIFORM_SPHERE:
        d["a"] = rng.uniform(-5, 5, (n, 3)); d["s"] = rng.uniform(0.1, 3, n)
    elif kind == rt.PDF_FUZZY:
        d["s"] = rng.uniform(0, 1, n)
    elif kind == rt.PDF_DIELECTRIC:
        d["b"] = _unit(rng, n); d["s"] = rng.uniform(0, 1, n); d["cannot_refract"] = rng.integers(0, 2, n)
    elif kind == rt.PDF_CONE:
        d["s"] = rng.uniform(-0.5, 0.999, n)
    elif kind in (rt.PDF_COSINE, rt.PDF_UNIFORM_HEMI):
        d["a"] *= rng.uniform(0.2, 3.0, (n, 1))   # the basis keeps w as passed: un-normalised normals occur (raw triangle normals, onb.rs:10-25)
    return d


def test_oracle_mixture_is_the_uniform_mixture_of_its_members():
    rng = np.random.default_rng(5)
    n = 400
    comps = [_descs(rng, n, k) for k in (rt.PDF_COSINE, rt.PDF_CONE, rt.PDF_UNIFORM_HEMI, rt.PDF_UNIFORM_SPHERE)]
    dirs = _unit(rng, n) * rng.uniform(0.5, 2, (n, 1))
    for m in (2, 4):    # weights 1/2, 1/4 are exact: the sum in list order is reproducible in numpy
        mix = np.stack(comps[:m], axis=1)
        v, g = orc.pdf_eval(mix, dirs, seed=9)
        singles = [orc.pdf_eval(c, dirs, seed=9, skip=1) for c in comps[:m]]   # the member's own generate after the draw `choose` consumed
        want = np.zeros(n)
        for sv, _ in singles:
            want = want + sv * (1.0 / m)
        assert np.array_equal(v, want)
        pick = np.array([min(int(orc.rng_stream(9, i, 0, 0, 1)[0] * m), m - 1) for i in range(n)])
        assert len(set(pick)) == m
        for i in range(n):
            gi = singles[pick[i]][1][i]
            assert np.allclose(g[i], gi / np.linalg.norm(gi), rtol=0, atol=1e-15)
        assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-14)
    # cosine pdf integrates to one over the sphere of directions (Monte Carlo over uniform directions): a sanity anchor for value()
    d = _descs(rng, 1, rt.PDF_COSINE); d["a"] = [[0.0, 0.0, 1.0]]
    u = _unit(rng, 200000)
    v, _ = orc.pdf_eval(np.repeat(d, len(u)), u)
    assert abs(v.mean() * 4 * np.pi - 1.0) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [rt.PDF_COSINE, rt.PDF_UNIFORM_HEMI, rt.PDF_UNIFORM_SPHERE, rt.PDF_FUZZY, rt.PDF_DIELECTRIC, rt.PDF_CONE])
def test_gpu_pdf_matches_oracle(gpu_ctx, kind):
    rng = np.random.default_rng(100 + kind)
    n = 5000
    d = _descs(rng, n, kind)
    dirs = _unit(rng, n) * rng.uniform(0.5, 2, (n, 1))
    gv, gg = gpu_ctx.pdf_eval(d, dirs, seed=77)
    ov, og = orc.pdf_eval(d, dirs, seed=77)
    assert np.array_equal(gv.view(np.uint64), ov.view(np.uint64))
    assert np.allclose(gg, og, rtol=1e-12, atol=1e-12), np.abs(gg - og).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kinds", [(rt.PDF_COSINE, rt.PDF_CONE), (rt.PDF_COSINE, rt.PDF_CONE, rt.PDF_UNIFORM_HEMI), (rt.PDF_COSINE, rt.PDF_UNIFORM_SPHERE, rt.PDF_FUZZY, rt.PDF_DIELECTRIC)])
def test_gpu_mixture_pdf_matches_oracle(gpu_ctx, kinds):
    # MixturePDF::Uniform (pdf.rs:199-224): e.g. the cosine + light-cone mixture "MIS mixture PDFs" would use
    rng = np.random.default_rng(7 + len(kinds))
    n = 5000
    mix = np.stack([_descs(rng, n, k) for k in kinds], axis=1)
    dirs = _unit(rng, n)
    gv, gg = gpu_ctx.pdf_eval(mix, dirs, seed=31)
    ov, og = orc.pdf_eval(mix, dirs, seed=31)
    assert np.array_equal(gv.view(np.uint64), ov.view(np.uint64))
    assert np.allclose(gg, og, rtol=1e-12, atol=1e-12), np.abs(gg - og).max()
    assert np.allclose(np.linalg.norm(gg, axis=1), 1.0, atol=1e-12)


@pytest.mark.gpu
def test_pdf_eval_rejects_bad_arguments(gpu_ctx):
    d = np.zeros((2, 9), rt.PDF_DESC)
    with pytest.raises(rt.RtError, match="n_comp"):
        gpu_ctx.pdf_eval(d, np.zeros((2, 3)))
    d = np.zeros(2, rt.PDF_DESC); d["kind"] = 6
    with pytest.raises(rt.RtError, match="kind"):
        gpu_ctx.pdf_eval(d, np.zeros((2, 3)))
