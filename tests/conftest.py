import os
import sys

# numpy.convolve (the oracle's float64 direct sum) is hundreds of thousands of short OpenBLAS dot products; with OpenBLAS's spinning
# thread pool each of them needs every pool thread scheduled, and on a box whose cores are busy (a build running beside the tests) the
# suite then crawls for tens of minutes.  One BLAS thread is as fast on a quiet box and immune to that.
for _var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, "1")

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:  # numpy may have been imported (by a plugin) before the variables above were set
        import threadpoolctl
        config._adsp_blas_limit = threadpoolctl.threadpool_limits(limits=1, user_api="blas")
    except Exception:
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def seeded_stream(seed, n_total):
    """Same generator as tests/golden/make_golden.py."""
    return np.random.default_rng(seed).uniform(-1, 1, n_total).astype(np.float32)


def assert_parity(got, ref, rel=1e-5, what=""):
    """The tolerance BASELINE.md section 3 states: max|d| <= 1e-5*max|ref| and allclose(1e-5, 1e-6)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max()
    assert err <= rel * max(scale, 1e-1), f"{what}: max|d|={err:.3e} scale={scale:.3e}"
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6 * max(scale, 1.0)), f"{what}: allclose failed"


@pytest.fixture(scope="session")
def golden():
    return {n: load_golden(n) for n in ("design", "kat_streams", "kat_chain", "kat_example1", "kat_edges", "kat_effects")}
