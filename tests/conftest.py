import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The Philox round count is part of the sampler's identity (a 7-round library and a 10-round checker draw different samples from the
    same seeds): library and oracle must have been built with the same one before any parity test means anything."""
    try:
        import oracle as O
        from em_model_manned_bayes_amd import _lib as L
        lib_rounds, oracle_rounds = int(L.lib().emgpu_philox_rounds()), O.philox_rounds()
    except Exception:      # not built yet: the tests that need them say so themselves
        return
    if lib_rounds != oracle_rounds:
        raise pytest.UsageError("libemgpu.so draws with Philox4x32-%d, oracle/libem_oracle.so with Philox4x32-%d: rebuild one of them "
                                "(EMGPU_PHILOX_ROUNDS / EM_PHILOX_ROUNDS)" % (lib_rounds, oracle_rounds))


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    """Directory with the packed models materialised as reference-format .txt files."""
    return str(tmp_path_factory.mktemp("models_txt"))


@pytest.fixture(scope="session")
def gpu_ctx():
    from em_model_manned_bayes_amd import native
    return native.Context(0)
