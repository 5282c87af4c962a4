"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle as O
from em_model_manned_bayes_amd import native, _lib as L
from util import load_pair, uncor_indices, assert_uncor_parity

pytestmark = pytest.mark.gpu

FAST_MODELS = ["uncor_1200code_v2p1", "uncor_1200only_fwse_v1p2", "uncor_1200exclude_rotorcraft_v1p2",
               "uncor_allcode_fwmulti_v1", "dueregard_v1", "haa_v1", "blimp_v1"]
DEP_MODELS = ["uncor_1200code_v1", "littoral_uncor_v1", "glider_v1", "paraglider_v1", "fai1_v1", "paramotor_v1", "skydiving_v1"]


@pytest.mark.parametrize("name", FAST_MODELS + DEP_MODELS)
@pytest.mark.parametrize("T", [240, 61])
def test_uncor_sample_matches_oracle(name, T, gpu_ctx, model_dir):
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    n, seed, first = 3000, 0x5EED0002, 12345678901
    idx = uncor_indices(pp)
    ref = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, first_index=first)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=True, **idx)
    assert_uncor_parity(got, ref, T)


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "uncor_1200code_v1", "glider_v1"])
def test_per_step_mode_matches_oracle(name, gpu_ctx, model_dir):
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    n, T, seed = 2000, 120, 77
    idx = uncor_indices(pp)
    ref = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, per_step=True)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=True,
                                 transition_mode=L.TRANSITION_PER_STEP, **idx)
    assert_uncor_parity(got, ref, T)


@pytest.mark.parametrize("name", FAST_MODELS)
@pytest.mark.parametrize("T,n", [(240, 5000), (61, 1000), (3, 300), (1, 100), (4, 257)])
def test_dense_only_fast_kernel_matches_oracle(name, T, n, gpu_ctx, model_dir):
    """Dense-only output takes the specialised kernel (k_uncor_fast) for fast-branch models."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    seed, first = 0xABCDEF12345, 2**33 + 17
    idx = uncor_indices(pp)
    ref = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, first_index=first, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=False, **idx)
    if name != "blimp_v1" or True:
        assert got["kernel"].startswith("k_uncor_fast"), got["kernel"]
    assert_uncor_parity(got, ref, T)
