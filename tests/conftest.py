import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (builds libsvo_hip.so on first use)."""
    return entry.build()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker only)."""
    return entry.load_oracle()


def load_vox_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"{name}_vox.npz"))
    return int(z["size"][0]), z["xyzi"], z["palette"], int(z["n_words"][0]), int(z["words_crc"][0])


@pytest.fixture(scope="session")
def small_words(pkg):
    size, xyzi, pal, n, _ = load_vox_fixture("small")
    w = pkg.CpuOctree.from_voxels(size, xyzi, pal).to_octree_words()
    assert w.size == n
    return w


@pytest.fixture(scope="session")
def monu9_words(pkg):
    size, xyzi, pal, n, _ = load_vox_fixture("monu9")
    w = pkg.CpuOctree.from_voxels(size, xyzi, pal).to_octree_words()
    assert w.size == n
    return w


@pytest.fixture(scope="session")
def gpu(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = pkg.Gpu(0)
    yield g
    g.close()


def set_uniforms_from_oracle(render, u):
    """Copy an oracle Uniforms into the product's Render (identical inputs on both sides)."""
    render.uniforms.camera[:] = list(u.camera)
    render.uniforms.camera_inverse[:] = list(u.camera_inverse)
    render.uniforms.dimensions[:] = list(u.dimensions)
    render.uniforms.sun_dir[:] = list(u.sun_dir)
    render.uniforms.flags = u.flags
    render.uniforms.misc_value = u.misc_value
    render.size = (int(u.dimensions[0]), int(u.dimensions[1]))
    render.upload_uniforms()


def assert_hits_equal(got, want, what=""):
    """Bit-exact on every integer field; t compared bit-exact too (the stated bar is 1e-5)."""
    got = got.reshape(-1)
    want = want.reshape(-1)
    assert got.shape == want.shape, what
    for f in ("value", "info", "normal_bits"):
        bad = np.flatnonzero(got[f] != want[f])
        assert bad.size == 0, f"{what}: {bad.size} rays differ in {f}; first {bad[:5]}: got {got[f][bad[:5]]} want {want[f][bad[:5]]}"
    gt, wt = got["t"], want["t"]
    both_nan = np.isnan(gt) & np.isnan(wt)
    assert np.all(both_nan | (np.abs(gt - wt) <= 1e-5) | (gt == wt)), f"{what}: hit-t beyond 1e-5"
    assert np.array_equal(gt.view(np.uint32)[~both_nan], wt.view(np.uint32)[~both_nan]), f"{what}: hit-t not bit-identical"
