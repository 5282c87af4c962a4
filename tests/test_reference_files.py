"""Container-only checks against the reference mount (/root/reference is absent on the GPU box: every
test here skips there).  (1) The product's C++ loader (emgpu_model_load_txt, the em_read.m replacement)
parses EVERY model file the reference ships, directly from where it lies, and agrees entry for entry with
the oracle's independent parser and with the packed copy under models/.  (2) The MATLAB-readable
goldens under tests/golden/matlab/ are what the oracle produces today."""
import glob
import os

import numpy as np
import pytest

import oracle as O
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import em_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MODELS = "/root/reference/model"
FILES = sorted(glob.glob(os.path.join(REF_MODELS, "**", "*.txt"), recursive=True))


@pytest.mark.skipif(not FILES, reason="the reference mount is only present in the build container")
@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, REF_MODELS) for f in FILES])
def test_native_loader_reads_the_reference_file_as_shipped(path):
    p = E.em_read(path)                  # product: C++ loader behind the em_read name
    q = O.parse_model_txt(path)          # oracle: independent numpy parser
    assert p["labels_initial"] == q["labels_initial"] and p["n_initial"] == q["n_initial"]
    assert np.array_equal(p["G_initial"], q["G_initial"]) and np.array_equal(p["r_initial"], q["r_initial"])
    assert np.array_equal(p["order_initial"], q["order_initial"])
    for v in range(p["n_initial"]):
        assert np.array_equal(p["N_initial"][v], q["N_initial"][v])
        assert np.array_equal(p["boundaries"][v], q["boundaries"][v])
    assert [0 if z == [] else z for z in p["zero_bins"]] == q["zero_bins"].tolist()
    assert np.array_equal(p["resample_rates"], q["resample_rates"])
    if q["n_transition"]:
        assert p["labels_transition"] == q["labels_transition"]
        assert np.array_equal(p["G_transition"], q["G_transition"]) and np.array_equal(p["r_transition"], q["r_transition"])
        assert np.array_equal(p["order_transition"], q["order_transition"]) and np.array_equal(p["temporal_map"], q["temporal_map"])
        for v, N in q["N_transition"].items():
            assert np.array_equal(p["N_transition"][v], N)
    # the packed copy that travels to the GPU box holds the same numbers
    stem = os.path.splitext(os.path.basename(path))[0]
    packed = os.path.join(ROOT, "models", stem + ".npz")
    assert os.path.exists(packed), "models/%s.npz missing: run tools/pack_models.py" % stem
    z = em_io.load_npz(packed)
    for v in range(p["n_initial"]):
        assert np.array_equal(np.asarray(z["N_initial"][v], dtype=np.float64), p["N_initial"][v])
        assert np.array_equal(np.asarray(z["boundaries"][v], dtype=np.float64).reshape(-1), np.asarray(p["boundaries"][v]).reshape(-1))
    assert z["labels_initial"] == p["labels_initial"] and np.array_equal(z["resample_rates"], p["resample_rates"])
    if q["n_transition"]:
        for v in q["N_transition"]:
            assert np.array_equal(np.asarray(z["N_transition"][v], dtype=np.float64), p["N_transition"][v])


def test_matlab_goldens_are_current(tmp_path):
    """tests/golden/matlab/*.csv == what the oracle (MT19937 mode) answers today, and config 1 == the npz golden."""
    gold = os.path.join(ROOT, "tests", "golden")
    pp = O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", str(tmp_path)))
    r = O.uncor_sample(O.OracleModel(pp), 100, 120, 1, mode=O.RNG_MT19937)
    inits = np.loadtxt(os.path.join(gold, "matlab", "config1_inits.csv"), delimiter=",", comments="%")
    ev = np.loadtxt(os.path.join(gold, "matlab", "config1_events.csv"), delimiter=",", comments="%")
    assert np.array_equal(inits, r["init_val"])
    z = np.load(os.path.join(gold, "config1_uncor_v2p1_mt19937_seed1_100x120.npz"))
    assert np.array_equal(inits, z["init_val"]) and np.array_equal(ev[:, 1:4], z["ev_flat"][:, :3])
    for i in (0, 57, 99):
        assert np.array_equal(ev[ev[:, 0] == i + 1][:, 1:4], r["events"][i][:, :3])
    for name, n, T, seed in (("uncor_1200code_v1", 40, 60, 7), ("cor_v1", 25, 50, 11)):
        pp = O.parse_model_txt(em_io.materialize_model(name, str(tmp_path)))
        r = O.uncor_sample(O.OracleModel(pp), n, T, seed, mode=O.RNG_MT19937, reject=False)
        assert np.array_equal(np.loadtxt(os.path.join(gold, "matlab", "hier_%s_inits.csv" % name), delimiter=",", comments="%"), r["init_val"])
        ev = np.loadtxt(os.path.join(gold, "matlab", "hier_%s_events.csv" % name), delimiter=",", comments="%")
        assert np.array_equal(ev[:, 1:4], np.concatenate(r["events"])[:, :3])
    for src, stem in (("terminalradar", "terminal_v3_radar_encounter_model"), ("opensky", "terminal_v3_opensky_encounter_model")):
        pp = O.parse_model_txt(em_io.materialize_model(stem, str(tmp_path)))
        labs = [x.strip('"') for x in pp["labels_initial"]]
        for tag, start in (("", [0] * 15), ("_start", [2, 1, 3] + [0] * 12)):
            _, ov, _ = O.geom_sample(O.OracleModel(pp, start=start), 500, 1, mode=O.RNG_MT19937, idx_own_speed=labs.index("own_speed") + 1,
                                     idx_int_speed=labs.index("int_speed") + 1, lim1=(50, 506), lim2=(50, 506))
            assert np.array_equal(np.loadtxt(os.path.join(gold, "matlab", "geom_%s%s_inits.csv" % (src, tag)), delimiter=",", comments="%"), ov)
