"""Device RNG vs NumPy (bit-exact): SeedSequence children, PCG64, ziggurat."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def test_seed_children_golden(ctx, golden_rng):
    got = ctx.seed_children(golden_rng["ss/entropy"], 0, 5)
    np.testing.assert_array_equal(got, golden_rng["ss/pcg_state"])
    # first_child offset
    got = ctx.seed_children(golden_rng["ss/entropy"], 3, 2)
    np.testing.assert_array_equal(got, golden_rng["ss/pcg_state"][3:5])


def test_seed_children_vs_numpy(ctx):
    from dynesty_amd import _lib
    rs = np.random.default_rng(5)
    # dynesty's get_seed_sequence: 4 ints below 2**63-1 (utils.py:1002-1009)
    ent = rs.integers(0, 2**63 - 1, size=4)
    kids = np.random.SeedSequence(ent).spawn(300)
    want = np.array([_lib.pcg_state_words(np.random.PCG64(c)) for c in kids])
    np.testing.assert_array_equal(ctx.seed_children(ent, 0, 300), want)
    # small / zero entropy words change the word count
    for ent in ([0, 1, 2, 3], [7], [2**40, 0, 5, 2**33 + 1]):
        kids = np.random.SeedSequence(ent).spawn(3)
        want = np.array([_lib.pcg_state_words(np.random.PCG64(c)) for c in kids])
        np.testing.assert_array_equal(ctx.seed_children(ent, 0, 3), want)


def test_stream_golden(ctx, golden_rng):
    st = golden_rng["ss/pcg_state"][2]
    nrm, unf, out = ctx.rng_stream(st, 5000, 100)
    np.testing.assert_array_equal(nrm, golden_rng["stream/normals"])
    np.testing.assert_array_equal(unf, golden_rng["stream/uniforms"])
    nrm2, _, out2 = ctx.rng_stream(out, 7, 0)
    np.testing.assert_array_equal(nrm2, golden_rng["stream/normals2"])
    np.testing.assert_array_equal(out2, golden_rng["stream/final_state"])


def test_stream_long_vs_numpy(ctx):
    """200k normals exercise the wedge and the tail branch of the ziggurat."""
    from dynesty_amd import _lib
    bg = np.random.PCG64(20240925)
    st = _lib.pcg_state_words(bg)
    want = np.random.Generator(bg).standard_normal(200000)
    nrm, _, out = ctx.rng_stream(st, 200000, 0)
    assert np.abs(want).max() > 3.66  # tail reached
    # the tail uses log1p: allow 1 ulp there, exact elsewhere
    assert np.max(np.abs(nrm - want)) < 1e-15
    assert np.mean(nrm == want) > 0.9999
    np.testing.assert_array_equal(out, _lib.pcg_state_words(bg))
