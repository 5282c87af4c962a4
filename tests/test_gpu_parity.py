"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle as O
from em_model_manned_bayes_amd import native, _lib as L
from util import load_pair, uncor_indices, assert_uncor_parity, assert_parting_only_on_a_threshold, assert_f32_of_f64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
    if name in FAST_MODELS:   # event lists of fast-branch models come from the fast kernel itself (haa_v1, 7 variables with a resample rate: the wide list, round 4)
        assert got["kernel"].startswith("k_uncor_fast_evw" if name == "haa_v1" else "k_uncor_fast_ev<"), got["kernel"]
    else:                     # ... and those of dependent-branch models from the per-timestep kernel
        assert got["kernel"].startswith("k_dbn_step2") and got["kernel"].endswith("+events"), got["kernel"]


@pytest.mark.parametrize("name", FAST_MODELS + DEP_MODELS + ["cor_v1", "littoral_cor_v1"])
@pytest.mark.parametrize("T,n,cap", [(240, 4000, 256), (8, 700, 64), (16, 500, 64), (1, 100, 8), (7, 300, 3), (33, 2500, 4096)])
def test_event_lists_from_the_fast_kernel_match_oracle(name, T, n, cap, gpu_ctx, model_dir):
    """k_uncor_fast_ev and k_dbn_step2<...>+events: events only (no dense trace asked for), lengths that are multiples of the 8-second block and not (the
    resample rows of second T come from the tail, dbn_hierarchical_sample.m:16-37 / resample_events.m:16-37), a list capacity that
    some trajectories overrun (EMGPU_ERR_EVENT_CAP, counts still exact) -- against the oracle's lists row for row: [dt, variable,
    bin] identical, values equal to the oracle's f64 rounded to f32, static-variable re-draws and rows hidden by a transition included."""
    nm, pp, _ = load_pair(name, model_dir)
    idx = uncor_indices(pp)
    seed, first = 0xE7E7, 2**36 + 11
    ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, mode=O.RNG_PHILOX, first_index=first)
    ref_cnt = np.array([len(e) for e in ref["events"]])
    if ref_cnt.max() > cap:
        with pytest.raises(L.EmgpuError) as ei:
            native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=False, want_events=True, event_cap=cap, **idx)
        assert ei.value.code == L.ERR_EVENT_CAP
        return
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=False, want_events=True, event_cap=cap, **idx)
    # the list alone: the rows of a block are built 64 at a time by the wave (haa_v1's seven rated variables included)
    assert got["kernel"].startswith("k_uncor_fast_evu") if name in FAST_MODELS else got["kernel"].endswith("+rows-by-wave+events"), got["kernel"]
    assert np.array_equal(got["ev_count"], ref_cnt)
    for i in range(n):
        g, r = got["events"][i], ref["events"][i]
        assert np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 3]), i
        assert np.array_equal(g["value"], r[:, 2].astype(np.float32)), i
    assert np.array_equal(got["init_bin"], ref["init_bin"])


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


# ---------------------------------------------------------------------------------------------
import glob
import re
import os

import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import em_io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unpack_events(cnt, flat):
    out, pos = [], 0
    for c in cnt:
        out.append(flat[pos: pos + c]); pos += c
    return out


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*_philox_*.npz"))))
def test_hip_path_reproduces_committed_golden_vectors(path, gpu_ctx, model_dir):
    g = np.load(path)
    n, T, seed, first, per_step = [int(x) for x in g["meta"]]
    name = os.path.basename(path).split("_philox_")[0]
    nm, pp, _ = load_pair(name, model_dir)
    ref = {k: g[k] for k in ("init_bin", "init_val", "dense_bin", "dense_val", "attempts")}
    ref["events"] = _unpack_events(g["ev_count"], g["ev_flat"])
    mode = L.TRANSITION_PER_STEP if per_step else L.TRANSITION_REFERENCE_AUTO
    for want_events in (False, True):
        got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=want_events,
                                     transition_mode=mode, **uncor_indices(pp))
        assert_uncor_parity(got, ref, T)


@pytest.mark.parametrize("name", ["cor_v1", "littoral_cor_v1", "cor_v2p1_like"])
def test_correlated_model_matches_oracle(name, gpu_ctx, model_dir):
    """16 initial / 4 dynamic variables; cor_v1 takes the dependent branch, littoral_cor_v1 the fast
    branch (which errors in the reference, dbn_sample.m:61,156; here it just works)."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    n, T, seed = 1500, 120, 31337
    ref = O.uncor_sample(om, n, T, seed)           # no "v"/"\dot h" labels -> rejection test disabled on both sides
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=True)
    assert_uncor_parity(got, ref, T)
    # dense only: the per-timestep kernel -- for the fast-branch model with its columns frozen at the initial state (dbn_sample.m:110-135)
    for T2, n2 in ((120, 1500), (7, 300), (8, 257), (1, 64)):
        ref2 = O.uncor_sample(om, n2, T2, seed + 1, first_index=2**33, want_events=False)
        got2 = native.sample_dbn_host(gpu_ctx, nm, n2, T2, seed + 1, first_index=2**33, want_dense=True, want_events=False)
        assert got2["kernel"].startswith("k_dbn_step2<16,4,") and (("[frozen]" in got2["kernel"]) == (name == "littoral_cor_v1")), got2["kernel"]
        assert_uncor_parity(got2, ref2, T2)


def test_start_presets_prior_layers_quantize(gpu_ctx, model_dir):
    nm, pp, path = load_pair("uncor_1200only_fwse_v1p2", model_dir, is_overwrite_zero_boundaries=True)
    idx = uncor_indices(pp)
    start = [1, 4, 2, 0, 0, 0, 0]                   # RUN_uncor.m:43-45
    layers = np.array([[50, 500], [500, 1200], [1200, 3000], [3000, 5000]], dtype=np.float64)
    n, T, seed = 2000, 90, 5
    nm.set_start(start)
    nm.set_prior("dbe")
    try:
        om = O.OracleModel(pp, prior="dbe", start=start)
        for kw in (dict(), dict(layers=layers), dict(is_quantize500=True), dict(layers=layers, is_quantize500=True)):
            ref = O.uncor_sample(om, n, T, seed, layers=kw.get("layers"), is_quantize500=kw.get("is_quantize500", False))
            flags = L.FLAG_QUANTIZE500 if kw.get("is_quantize500") else 0
            got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=True, flags=flags,
                                         layers=kw.get("layers"), **idx)
            assert_uncor_parity(got, ref, T)
            assert np.all(got["init_bin"][:, :3] == [1, 4, 2])
            if kw.get("is_quantize500"):
                level = got["init_val"][:, 5] == 0
                assert np.all(got["init_val"][level, 2] % 500 == 0)
    finally:
        nm.set_start([0] * 7)
        nm.set_prior(0)
    with pytest.raises(E.EmgpuError) as ei:          # bn_sample.m:45-47
        nm.set_start([0, 0, 0, 0, 3, 0, 0])
        native.sample_dbn_host(gpu_ctx, nm, 8, 10, 1)
    assert ei.value.code == L.ERR_PRESET
    nm.set_start([0] * 7)


@pytest.mark.parametrize("name", FAST_MODELS + DEP_MODELS + ["cor_v1", "littoral_cor_v1", "balloon_v1"])
def test_plain_dbn_sample_from_the_fast_kernel(name, gpu_ctx, model_dir):
    """dbn_sample.m:1 itself -- no resample rows, no dediscretize, no terminator: [initial bins, rows (dt, variable, new bin)] -- comes
    from k_uncor_fast_evu on a fast-branch model and from k_dbn_step2<...>+events (resample streams off, values = bins) on the others
    (rounds 1-3: k_dbn_generic), row for row the oracle's; lengths on and off the block."""
    path = em_io.materialize_model(name, model_dir)
    p = E.em_read(path)
    om = O.OracleModel(O.parse_model_txt(path))
    di = E.bn_dirichlet_prior(p["N_initial"], 0)
    dt = E.bn_dirichlet_prior(p["N_transition"], 0)
    for T, n, seed in [(240, 700, 5), (8, 300, 6), (29, 500, 7), (1, 100, 8)]:
        inits, evs = E.dbn_sample(p, di, dt, T, seed=seed, num_samples=n, ctx=gpu_ctx)
        k = gpu_ctx.last_kernel()
        assert k.startswith("k_uncor_fast_evu") if name in FAST_MODELS else (k.startswith("k_dbn_step2") and k.endswith("+events")), k
        rb, rev = O.dbn_sample(om, n, T, seed)
        assert np.array_equal(inits, rb)
        assert sum(len(b) for b in rev) > 0 or T == 1
        for a, b in zip(evs, rev):
            assert np.array_equal(a, b), (name, T)


def test_dbn_sample_and_hierarchical_functions(gpu_ctx, model_dir):
    path = em_io.materialize_model("uncor_1200code_v1", model_dir)
    p = E.em_read(path)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    di = E.bn_dirichlet_prior(p["N_initial"], 0)
    dt = E.bn_dirichlet_prior(p["N_transition"], 0)
    inits, evs = E.dbn_sample(p, di, dt, 40, seed=77, num_samples=300, ctx=gpu_ctx)
    rb, rev = O.dbn_sample(om, 300, 40, 77)
    assert np.array_equal(inits, rb)
    for a, b in zip(evs, rev):
        assert np.array_equal(a, b)                 # rows (dt, variable, new bin), dbn_sample.m:84-91
    iv, ev = E.dbn_hierarchical_sample(p, di, dt, 40, p["boundaries"], p["zero_bins"], p["resample_rates"], [None] * 6,
                                       seed=77, num_samples=200, ctx=gpu_ctx)
    om2 = O.OracleModel(pp)
    L_ = O.lib()
    import ctypes as C
    o = O._UncorOpts(); o.max_attempts = 1          # no rejection test in dbn_hierarchical_sample itself
    ref = O.uncor_sample(om2, 200, 40, 77)
    # uncor_sample adds only the (rarely failing) rejection test on top; compare the never-rejected ones
    keep = ref["attempts"] == 1
    assert keep.sum() > 150
    for i in np.nonzero(keep)[0]:
        assert np.array_equal(iv[i].astype(np.float32), ref["init_val"][i].astype(np.float32))
        assert np.array_equal(ev[i][:, :2], ref["events"][i][:, :2])
        assert np.array_equal(ev[i][:, 2].astype(np.float32), ref["events"][i][:, 2].astype(np.float32))


def test_bn_sample_function_and_terminal_geometry(gpu_ctx, model_dir):
    path = em_io.materialize_model("terminal_v3_radar_encounter_model", model_dir)
    p = E.em_read(path)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    n, seed = 4000, 2024
    S = E.bn_sample(p["G_initial"], p["r_initial"], p["N_initial"], E.bn_dirichlet_prior(p["N_initial"], 0), n,
                    [None] * 15, p["order_initial"], seed=seed, ctx=gpu_ctx)
    ob, _, _ = O.geom_sample(om, n, seed, max_attempts=1)
    assert np.array_equal(S, ob)                     # bn_sample.m:39-57, r up to 36 bins
    # @CorTerminalModel/sample.m:29-77 with the GENERIC speed limits and a bounds box
    t = E.CorTerminalModel(srcData="terminalradar")
    labs = t.labels_initial
    io, ii = labs.index('"own_speed"') + 1, labs.index('"int_speed"') + 1
    outInits, outSamples = t.sample(n, seed=seed, ctx=gpu_ctx)
    rb, rv, ra = O.geom_sample(om, n, seed, idx_own_speed=io, idx_int_speed=ii, lim1=(50, 506), lim2=(50, 506))
    assert np.array_equal(outInits.astype(np.float32), rv.astype(np.float32))
    t.acType1, t.acType2 = "RTCA228_A1", "RTCA228_A3"  # 169-491 ft/s and 68-186 ft/s: the speed test rejects
    outInits, _ = t.sample(n, seed=seed, ctx=gpu_ctx)
    rb, rv, ra = O.geom_sample(om, n, seed, idx_own_speed=io, idx_int_speed=ii, lim1=(169, 491), lim2=(68, 186))
    assert np.array_equal(outInits.astype(np.float32), rv.astype(np.float32))
    assert ra.max() > 1                              # the rejection loop was exercised
    assert outInits[:, io - 1].min() >= 169 and outInits[:, ii - 1].max() <= 186
    t.acType1 = t.acType2 = "GENERIC"
    assert set(outSamples[0].keys()) == {lab.replace('"', "") for lab in labs}
    bs = np.column_stack([-np.inf * np.ones(15), np.inf * np.ones(15)])
    bs[labs.index('"own_distance"')] = [0, 3]
    t.bounds_sample = bs
    outInits, _ = t.sample(1000, seed=9, ctx=gpu_ctx)
    _, rv, _ = O.geom_sample(om, 1000, 9, bounds_sample=bs, idx_own_speed=io, idx_int_speed=ii, lim1=(50, 506), lim2=(50, 506))
    assert np.array_equal(outInits.astype(np.float32), rv.astype(np.float32))
    assert outInits[:, labs.index('"own_distance"')].max() <= 3


def test_uncor_class_sample_matches_reference_outputs(gpu_ctx, model_dir):
    """UncorEncounterModel.sample end to end: out_inits, out_events, out_samples, out_EME
    (UncorEncounterModel.m:283-300) against the oracle's events2samples / events2controls."""
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    mdl = E.UncorEncounterModel(parameters_filename=path)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    n, T, seed = 100, 120, 1                         # the shape of BASELINE.json configs[0]
    out_inits, out_events, out_samples, out_EME = mdl.sample(n, T, seed=seed, ctx=gpu_ctx)
    ref = O.uncor_sample(om, n, T, seed)
    assert out_inits.shape == (n, 7) and len(out_events) == n
    for i in range(n):
        r32 = ref["events"][i].copy()
        assert np.array_equal(out_events[i][:, :2], r32[:, :2])
        assert np.array_equal(out_events[i][:, 2].astype(np.float32), r32[:, 2].astype(np.float32))
        assert np.array_equal(out_inits[i].astype(np.float32), ref["init_val"][i].astype(np.float32))      # values cross the boundary as f32
        s = O.events2samples(ref["init_val"][i], r32[:, :3])
        assert out_samples[i].shape == (7, T)
        assert np.array_equal(out_samples[i].astype(np.float32), s.astype(np.float32))
        ctl = O.events2controls(om, ref["init_val"][i], r32[:, :3])[:, [0, 2, 3, 1]]
        ctl[:, 1] /= 60.0; ctl[:, 2] = np.deg2rad(ctl[:, 2]); ctl[:, 3] *= 1.68780972222222
        np.testing.assert_allclose(out_EME[i].event, ctl, rtol=1.2e-7, atol=0)      # unit conversions of f32-stored values: 2^-23
        assert out_EME[i].event[0, 0] == 0
    # a call that spans several internal chunks gives the same samples as one-at-a-time calls at the same global index
    big = mdl.sample(20000, 120, seed=5, ctx=gpu_ctx)
    for i in (0, 14978, 14979, 14980, 19999):
        one = mdl.sample(1, 120, seed=5, first_index=i, ctx=gpu_ctx)
        assert np.array_equal(big[0][i], one[0][0]) and np.array_equal(big[1][i], one[1][0])
        assert np.array_equal(big[2][i], one[2][0]) and np.array_equal(big[3][i].event, one[3][0].event)
        r1 = O.uncor_sample(om, 1, 120, 5, first_index=i)
        assert np.array_equal(big[2][i].astype(np.float32), O.events2samples(r1["init_val"][0], r1["events"][0][:, :3]).astype(np.float32))
    with pytest.raises(E.EmgpuError) as ei:          # UncorEncounterModel.m:231-234
        E.UncorEncounterModel(parameters_filename=em_io.materialize_model("balloon_v1", model_dir)).sample(1, 10, seed=1, ctx=gpu_ctx)
    assert ei.value.identifier == "dynvar:empty"


@pytest.mark.parametrize("name", ["uncor_1200only_fwse_v1p2", "uncor_1200exclude_rotorcraft_v1p2"])
def test_uncor_class_sample_of_a_v1p2_model_redraws_static_rows(name, gpu_ctx, model_dir):
    """The *_v1p2 files carry non-zero resample rates on the STATIC variables L and v (model/uncor_1200only_fwse_v1p2.txt:41): the rows of
    out_samples for L and v are re-drawn inside their bin over time (resample_events.m:24-28, events2samples.m:25).  Class level -- the numpy
    reconstruction of encounter_model.py, not only the native event list: out_samples, out_events and the controls against the oracle's
    events2samples / events2controls, and the static rows really do change."""
    path = em_io.materialize_model(name, model_dir)
    mdl = E.UncorEncounterModel(parameters_filename=path)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    rates = np.asarray(pp["resample_rates"], dtype=np.float64)
    iL, iV = pp["labels_initial"].index('"L"'), pp["labels_initial"].index('"v"')
    assert rates[iL] > 0 and rates[iV] > 0                       # the premise: static variables with a resample rate
    n, T, seed = 400, 240, 0x5EED0004
    out_inits, out_events, out_samples, out_EME = mdl.sample(n, T, seed=seed, ctx=gpu_ctx)
    ref = O.uncor_sample(om, n, T, seed)
    moved_L = moved_v = 0
    for i in range(n):
        r = ref["events"][i]
        assert np.array_equal(out_events[i][:, :2], r[:, :2]) and np.array_equal(out_events[i][:, 2].astype(np.float32), r[:, 2].astype(np.float32))
        s = O.events2samples(ref["init_val"][i], r[:, :3])
        assert np.array_equal(out_samples[i].astype(np.float32), s.astype(np.float32)), i
        moved_L += int(np.ptp(s[iL]) > 0); moved_v += int(np.ptp(s[iV]) > 0)
        # a re-drawn static value stays inside the bin of the initial draw (resample_events.m:26 emits the CURRENT bin)
        for var, row in ((iL, out_samples[i][iL]), (iV, out_samples[i][iV])):
            b = pp["boundaries"][var]
            if len(b):
                k = ref["init_bin"][i][var]
                assert (row >= np.float32(b[k - 1])).all() and (row <= np.float32(b[k])).all(), (i, var)
        ctl = O.events2controls(om, ref["init_val"][i], r[:, :3])[:, [0, 2, 3, 1]]
        ctl[:, 1] /= 60.0; ctl[:, 2] = np.deg2rad(ctl[:, 2]); ctl[:, 3] *= 1.68780972222222
        np.testing.assert_allclose(out_EME[i].event, ctl, rtol=1.2e-7, atol=0)
    assert moved_L > n // 4 and moved_v > n // 4, (moved_L, moved_v)      # rates 0.0013-0.04 per second over 240 s


def test_sharded_calls_equal_one_call(gpu_ctx, model_dir):
    """Multi-GPU rule on one GPU: two calls over [0, n/2) and [n/2, n) equal one call over [0, n)."""
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 6000, 240, 0x5EED0004
    full = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=0, want_dense=True, want_events=False, **idx)
    a = native.sample_dbn_host(gpu_ctx, nm, 2500, T, seed, first_index=0, want_dense=True, want_events=False, **idx)
    b = native.sample_dbn_host(gpu_ctx, nm, 3500, T, seed, first_index=2500, want_dense=True, want_events=False, **idx)
    for k in ("init_bin", "init_val", "dyn_bin", "dyn_val"):
        assert np.array_equal(np.concatenate([a[k], b[k]]), full[k]), k


def test_full_size_properties(gpu_ctx, model_dir):
    """BASELINE.json configs[1] at full size (10 M x 240 s, device resident): size-independent
    properties -- determinism, bins in range, column 0 == initial state, values inside their bin's
    boundaries (dediscretize.m:33-39) or exactly 0 in the zero bin, and a spot check of 4096
    trajectories out of the middle of the batch against the oracle."""
    import torch
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 10_000_000, 240, 0x5EED0002
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    G4 = T // 4
    ib = torch.empty((7, n), dtype=torch.uint8, device=dev); iv = torch.empty((7, n), dtype=torch.float32, device=dev)
    db = torch.empty((G4, 3, n), dtype=torch.int32, device=dev); dv = torch.empty((G4, 3, n, 4), dtype=torch.float32, device=dev)
    p, _ = native.make_params(n, T, seed, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
    ctx.sync()
    assert ctx.last_kernel().startswith("k_uncor_fast")
    chk = int(db.to(torch.int64).sum().item()), float(dv.double().sum().item())
    bytes_ = db.view(torch.uint8).view(G4, 3, n, 4)
    r_dyn = [5, 7, 7]
    bnd = [pp["boundaries"][v] for v in (4, 5, 6)]
    zb = [3, 4, 4]
    for k in range(3):
        bk = bytes_[:, k]
        assert int(bk.min()) >= 1 and int(bk.max()) <= r_dyn[k]
        assert torch.equal(bk[0, :, 0], ib[4 + k])                       # column 0 is the initial state
        assert torch.equal(dv[0, k, :, 0], iv[4 + k])
        lo = torch.tensor(bnd[k][:-1], dtype=torch.float32, device=dev)[bk.long() - 1]
        hi = torch.tensor(bnd[k][1:], dtype=torch.float32, device=dev)[bk.long() - 1]
        v = dv[:, k]
        zero = bk == zb[k]
        assert bool(((v == 0) | ~zero).all())                 # zero bin => exactly 0 (dediscretize.m:24-25)
        assert bool((((v >= lo) & (v <= hi)) | zero).all())   # otherwise inside [b(d), b(d+1)]
        del lo, hi, zero
    # determinism: a second launch writes identical bytes
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
    ctx.sync()
    assert chk == (int(db.to(torch.int64).sum().item()), float(dv.double().sum().item()))
    # spot check against the oracle in the middle of the batch
    lo_i, m = 5_000_000, 4096
    ref = O.uncor_sample(O.OracleModel(pp), m, T, seed, first_index=lo_i, want_events=False)
    gb = native.unpack_dyn_bin(db[:, :, lo_i: lo_i + m].contiguous().cpu().numpy().view(np.uint32), T)
    gv = native.unpack_dyn_val(dv[:, :, lo_i: lo_i + m].contiguous().cpu().numpy(), T)
    assert np.array_equal(gb, ref["dense_bin"]) and np.array_equal(gv, ref["dense_val"].astype(np.float32))


@pytest.mark.parametrize("name", DEP_MODELS + ["cor_v1"])
@pytest.mark.parametrize("T,n", [(240, 3000), (61, 700), (2, 130), (9, 257)])
def test_dense_only_step_kernel_matches_oracle(name, T, n, gpu_ctx, model_dir):
    """Dense-only output of dependent-branch models takes k_dbn_step."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    seed, first = 0xFEEDFACE, 2**35 + 5
    idx = uncor_indices(pp)
    ref = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, first_index=first, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=False, **idx)
    assert got["kernel"].startswith("k_dbn_step"), got["kernel"]
    assert_uncor_parity(got, ref, T)


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "uncor_1200only_fwme_v1p2", "dueregard_v1", "haa_v1", "littoral_cor_v1"])
def test_dense_only_per_step_mode_matches_oracle(name, gpu_ctx, model_dir):
    """EMGPU_TRANSITION_PER_STEP on fast-branch models: the true DBN instead of the frozen parents."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    n, T, seed = 2500, 120, 4242
    idx = uncor_indices(pp)
    ref = O.uncor_sample(om, n, T, seed, per_step=True, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=False,
                                 transition_mode=L.TRANSITION_PER_STEP, **idx)
    assert got["kernel"].startswith("k_dbn_step"), got["kernel"]
    assert_uncor_parity(got, ref, T)
    frozen = O.uncor_sample(om, n, T, seed, per_step=False, want_events=False)
    assert not np.array_equal(frozen["dense_bin"], ref["dense_bin"])   # the two semantics really differ


def test_mixed_model_batch_matches_oracle(gpu_ctx, model_dir):
    """BASELINE.json configs[3] in miniature: the six uncor_*_v1p2 files (same shapes, different CPTs)
    plus two models of other shapes, contiguous index blocks per model; two 'ranks' own half the index
    range each and fill THEIR HALF of one shared trace with one emgpu_sample_dbn_blocks_device call
    (one launch per block, written in place through ld / col_offset: no temporaries, no copies)."""
    import torch
    from em_model_manned_bayes_amd import sharding
    names = ["uncor_1200only_fwse_v1p2", "uncor_1200only_fwme_v1p2", "uncor_1200only_rotorcraft_v1p2",
             "uncor_1200exclude_fwse_v1p2", "uncor_1200exclude_fwme_v1p2", "uncor_1200exclude_rotorcraft_v1p2",
             "uncor_1200code_v2p1", "uncor_allcode_rotorcraft_v1"]
    pairs = [load_pair(nm_, model_dir) for nm_ in names]
    n_total, T, seed = 4001, 64, 0x5EED0004
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    G4 = T // 4
    ib = torch.zeros((7, n_total), dtype=torch.uint8, device=dev)
    iv = torch.zeros((7, n_total), dtype=torch.float32, device=dev)
    db = torch.zeros((G4, 3, n_total), dtype=torch.int32, device=dev)
    dv = torch.zeros((G4, 3, n_total, 4), dtype=torch.float32, device=dev)
    for rank in range(2):
        lo, hi = sharding.shard_range(n_total, rank, 2)
        blocks = native.mixed_blocks(n_total, len(names), lo, hi)
        assert blocks == sharding.mixed_batch_blocks(n_total, len(names), lo, hi)
        p, _ = native.make_params(hi - lo, T, seed, first_index=lo, **uncor_indices(pairs[0][1]))
        native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, init_bin=ib.data_ptr(), init_val=iv.data_ptr(),
                                        dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(), ld=n_total, col_offset=lo)
    ctx.sync()
    gb = native.unpack_dyn_bin(db.cpu().numpy().view(np.uint32), T)
    gv = native.unpack_dyn_val(dv.cpu().numpy(), T)
    gib, giv = ib.cpu().numpy().T, iv.cpu().numpy().T
    for m, first, cnt in sharding.mixed_batch_blocks(n_total, len(names)):
        ref = O.uncor_sample(O.OracleModel(pairs[m][1]), cnt, T, seed, first_index=first, want_events=False)
        assert np.array_equal(gb[first: first + cnt], ref["dense_bin"]), names[m]
        assert np.array_equal(gv[first: first + cnt], ref["dense_val"].astype(np.float32)), names[m]
        assert np.array_equal(gib[first: first + cnt], ref["init_bin"]), names[m]
        assert np.array_equal(giv[first: first + cnt], ref["init_val"].astype(np.float32)), names[m]
    # a block outside the range the trace covers is refused
    p, _ = native.make_params(100, T, seed, first_index=1000)
    with pytest.raises(L.EmgpuError):
        native.sample_dbn_blocks_device(ctx, [pairs[0][0]], p, [(0, 1050, 51)], dyn_bin=db.data_ptr(), ld=n_total)
    with pytest.raises(L.EmgpuError):   # models of different trace shapes cannot share a trace
        native.sample_dbn_blocks_device(ctx, [pairs[0][0], load_pair("cor_v1", model_dir)[0]], p, [(0, 1000, 10)], dyn_bin=db.data_ptr(), ld=n_total)


@pytest.mark.gpu
def test_mixed_batch_is_one_launch_at_full_size_and_exact_across_every_block_boundary(model_dir):
    """BASELINE.json configs[3] at one rank's full size: 6.25 M trajectories x 240 s over the six uncor_*_v1p2 files in ONE launch
    (k_uncor_fast_mixed: model id per workgroup; RUN_1_emsample.m:13,24-47 is the reference's per-file loop).  Oracle slices
    straddle every one of the five block boundaries, the two ends of the trace and the middle of every block; the
    per-block-launch path (EMGPU_DEBUG_NO_MIXED_LAUNCH, in a child process: the switch is read once) writes the same bytes."""
    import subprocess
    import torch
    from em_model_manned_bayes_amd import sharding
    from bench import V1P2
    pairs = [load_pair(nm_, model_dir) for nm_ in V1P2]
    n, T, seed, first = 6_250_000, 240, 0x5EED0004, 3 * 6_250_000
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    G4 = T // 4
    ib = torch.zeros((7, n), dtype=torch.uint8, device=dev); iv = torch.zeros((7, n), dtype=torch.float32, device=dev)
    db = torch.zeros((G4, 3, n), dtype=torch.int32, device=dev); dv = torch.zeros((G4, 3, n, 4), dtype=torch.float32, device=dev)
    at = torch.zeros(n, dtype=torch.int32, device=dev)
    # the rank's shard of a 50 M batch is itself cut at the model boundaries of the GLOBAL range; here the shard is its own batch
    blocks = [(m, first + f, c) for (m, f, c) in native.mixed_blocks(n, 6, 0, n)]
    p, _ = native.make_params(n, T, seed, first_index=first, **uncor_indices(pairs[0][1]))
    native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, init_bin=ib.data_ptr(), init_val=iv.data_ptr(),
                                    dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(), attempts=at.data_ptr())
    ctx.sync()
    assert ctx.last_launches() == 1 and ctx.last_kernel() == "k_uncor_fast_mixed<7,4,6,6>", (ctx.last_launches(), ctx.last_kernel())
    oms = [O.OracleModel(pr[1]) for pr in pairs]

    def check(lo, hi):   # columns [lo, hi) of the trace against the oracle, model by model
        gb = native.unpack_dyn_bin(db[:, :, lo:hi].contiguous().cpu().numpy().view(np.uint32), T)
        gv = native.unpack_dyn_val(dv[:, :, lo:hi].contiguous().cpu().numpy(), T)
        gib, giv, gat = ib[:, lo:hi].cpu().numpy().T, iv[:, lo:hi].cpu().numpy().T, at[lo:hi].cpu().numpy()
        for (m, f, c) in blocks:
            a, b = max(lo, f - first), min(hi, f - first + c)
            if b <= a:
                continue
            ref = O.uncor_sample(oms[m], b - a, T, seed, first_index=first + a, want_events=False)
            sl = slice(a - lo, b - lo)
            assert np.array_equal(gb[sl], ref["dense_bin"]) and np.array_equal(gv[sl], ref["dense_val"].astype(np.float32)), (m, a, b)
            assert np.array_equal(gib[sl], ref["init_bin"]) and np.array_equal(giv[sl], ref["init_val"].astype(np.float32)), (m, a, b)
            assert np.array_equal(gat[sl], ref["attempts"]), (m, a, b)
    edges = [f - first for (_, f, _) in blocks[1:]]
    for e in edges:
        check(e - 300, e + 300)          # both sides of a model boundary: the last (partial) workgroup of one block, the first of the next
    check(0, 300); check(n - 300, n)
    for (_, f, c) in blocks:
        check(f - first + c // 2, f - first + c // 2 + 200)
    chk = (int(db.to(torch.int64).sum().item()), float(dv.double().sum().item()), int(ib.to(torch.int64).sum().item()))
    code = r'''
import os, sys, torch
for q in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(os.environ["EMGPU_ROOT"], q))
from em_model_manned_bayes_amd import native
from util import load_pair, uncor_indices
from bench import V1P2
pairs = [load_pair(nm_, os.environ["EMGPU_MODEL_DIR"]) for nm_ in V1P2]
n, T, seed, first = 6_250_000, 240, 0x5EED0004, 3 * 6_250_000
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ib = torch.zeros((7, n), dtype=torch.uint8, device=dev); iv = torch.zeros((7, n), dtype=torch.float32, device=dev)
db = torch.zeros((T // 4, 3, n), dtype=torch.int32, device=dev); dv = torch.zeros((T // 4, 3, n, 4), dtype=torch.float32, device=dev)
blocks = [(m, first + f, c) for (m, f, c) in native.mixed_blocks(n, 6, 0, n)]
p, _ = native.make_params(n, T, seed, first_index=first, **uncor_indices(pairs[0][1]))
native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
ctx.sync()
print(ctx.last_launches(), int(db.to(torch.int64).sum().item()), repr(float(dv.double().sum().item())), int(ib.to(torch.int64).sum().item()))
'''
    del ib, iv, db, dv, at
    torch.cuda.empty_cache()
    env = dict(os.environ, EMGPU_DEBUG_NO_MIXED_LAUNCH="1", EMGPU_ROOT=ROOT, EMGPU_MODEL_DIR=str(model_dir))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    launches, s_db, s_dv, s_ib = out.stdout.decode().split()
    assert int(launches) == 6 and (int(s_db), float(s_dv), int(s_ib)) == chk


@pytest.mark.gpu
def test_mixed_launch_groups_blocks_by_kernel_instance(gpu_ctx, model_dir):
    """emgpu_sample_dbn_blocks_device: blocks whose models share a k_uncor_fast instance are one launch of at most 16 blocks; models
    of other instances or kernel families get launches of their own -- 20 blocks of one v1.2 model (16 + 4), then v2p1 (another
    instance), a dependent-branch model (k_dbn_step2) and empty blocks in between; every column against the oracle."""
    import torch
    names = ["uncor_1200only_fwse_v1p2", "uncor_1200code_v2p1", "uncor_1200code_v1"]
    pairs = [load_pair(nm_, model_dir) for nm_ in names]
    T, seed, first = 40, 0xB10C, 2**34 + 3
    sizes = [257, 1, 300, 255, 256, 511, 64, 63, 700, 2, 100, 333, 256, 257, 90, 10, 500, 129, 128, 1]   # 20 blocks of model 0 (odd offsets throughout)
    blocks, col = [], 0
    for c in sizes:
        blocks.append((0, first + col, c)); col += c
    blocks.append((1, first + col, 0))                         # an empty block
    blocks.append((1, first + col, 1000)); col += 1000         # another instance: its own launch
    blocks.append((2, first + col, 777)); col += 777           # dependent branch: k_dbn_step2
    n = col
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    G4 = T // 4
    ld = n + 57   # a padded leading dimension
    ib = torch.zeros((7, ld), dtype=torch.uint8, device=dev); iv = torch.zeros((7, ld), dtype=torch.float32, device=dev)
    db = torch.zeros((G4, 3, ld), dtype=torch.int32, device=dev); dv = torch.zeros((G4, 3, ld, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    # (uncor_1200code_v1 has 6 initial variables: the three models do not share a trace shape)
    p, _ = native.make_params(n, T, seed, first_index=first, **uncor_indices(pairs[0][1]))
    with pytest.raises(L.EmgpuError):
        native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, dyn_bin=db.data_ptr(), ld=ld)
    pairs, blocks, n = pairs[:2], blocks[:-1], n - 777
    p, _ = native.make_params(n, T, seed, first_index=first, **uncor_indices(pairs[0][1]))
    native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, init_bin=ib.data_ptr(), init_val=iv.data_ptr(),
                                    dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(), ld=ld)
    ctx.sync()
    assert ctx.last_launches() == 3, ctx.last_launches()      # 16 + 4 blocks of the v1.2 instance, 1 of the v2p1 instance
    gb = native.unpack_dyn_bin(db.cpu().numpy().view(np.uint32), T)
    gv = native.unpack_dyn_val(dv.cpu().numpy(), T)
    for (m, f, c) in blocks:
        if c == 0:
            continue
        ref = O.uncor_sample(O.OracleModel(pairs[m][1]), c, T, seed, first_index=f, want_events=False)
        sl = slice(f - first, f - first + c)
        assert np.array_equal(gb[sl], ref["dense_bin"]) and np.array_equal(gv[sl], ref["dense_val"].astype(np.float32)), (m, f, c)
        assert np.array_equal(ib[:, sl].cpu().numpy().T, ref["init_bin"])
    assert not gb[n:].any() and not gv[n:].any()              # nothing written past the batch
    # event lists through the block entry point: one k_uncor_fast_ev launch per block, lists in the call's columns
    cap = 128
    ec = torch.zeros(n, dtype=torch.int32, device=dev); ev = torch.zeros((n, cap, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    p2, _ = native.make_params(n, T, seed, first_index=first, event_cap=cap, **uncor_indices(pairs[0][1]))
    sub = [b for b in blocks if b[2] > 0][:3] + [blocks[-1]]
    native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p2, sub, ev_count=ec.data_ptr(), events=ev.data_ptr())
    ctx.sync()
    assert ctx.last_launches() == len(sub) and ctx.last_kernel().startswith("k_uncor_fast_ev")
    cnt = ec.cpu().numpy(); evh = ev.cpu().numpy().reshape(n, cap * 2).view(native.EVENT_DTYPE)
    for (m, f, c) in sub:
        ref = O.uncor_sample(O.OracleModel(pairs[m][1]), c, T, seed, first_index=f)
        for q in range(0, c, 37):
            r = ref["events"][q]; g = evh[f - first + q, : cnt[f - first + q]]
            assert len(g) == len(r) and np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["value"], r[:, 2].astype(np.float32))


@pytest.mark.gpu
def test_event_lists_at_two_million_trajectories_properties(model_dir):
    """k_uncor_fast_ev at 2 M x 240 s, device resident: every list's dt column sums to T (the terminator closes it,
    dbn_hierarchical_sample.m:15-19), variable ids and bins are in range, no list overruns its capacity, the mean list length is the
    CPU oracle's for the same index range (a slice), and a second launch writes the same bytes."""
    import torch
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed, cap = 2_000_000, 240, 0x5EED0002, 256
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ec = torch.zeros(n, dtype=torch.int32, device=dev); ev = torch.zeros((n, cap, 2), dtype=torch.int32, device=dev)
    p, _ = native.make_params(n, T, seed, event_cap=cap, **idx)
    native.sample_dbn_device(ctx, nm, p, ev_count=ec.data_ptr(), events=ev.data_ptr())
    ctx.sync()
    assert ctx.last_kernel().startswith("k_uncor_fast_ev")
    w0 = ev[:, :, 0]                                             # dt u16 | var u8 << 16 | bin u8 << 24
    live = torch.arange(cap, device=dev)[None, :] < ec[:, None]
    dt = torch.where(live, w0 & 0xFFFF, torch.zeros_like(w0))
    assert int(ec.max()) <= cap and int(ec.min()) >= 1
    assert bool((dt.sum(dim=1) == T).all())
    var = torch.where(live, (w0 >> 16) & 0xFF, torch.zeros_like(w0))
    assert int(var.max()) <= 7
    last = ev[torch.arange(n, device=dev), (ec - 1).long(), 0]
    assert bool((((last >> 16) & 0xFF) == 0).all())             # the terminator row: variable 0
    chk = (int(ec.sum().item()), int(ev.to(torch.int64).sum().item()))
    ec.zero_(); ev.zero_()
    native.sample_dbn_device(ctx, nm, p, ev_count=ec.data_ptr(), events=ev.data_ptr())
    ctx.sync()
    assert chk == (int(ec.sum().item()), int(ev.to(torch.int64).sum().item()))
    lo, m = 1_234_567, 3000
    ref = O.uncor_sample(O.OracleModel(pp), m, T, seed, first_index=lo)
    assert np.array_equal(ec[lo: lo + m].cpu().numpy(), np.array([len(e) for e in ref["events"]]))


@pytest.mark.gpu
def test_index_lists_are_cut_with_the_shards_and_blocks(gpu_ctx, model_dir):
    """emgpu_sample_params.indices through the multi-context and the block entry points: shard d / block b draws ITS
    entries of the list, not the list's head again."""
    import torch
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 1003, 29, 0xABCD
    rng = np.random.default_rng(5)
    pick = rng.permutation(50_000)[:n].astype(np.uint64) + 7
    one = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=True, indices=pick, **idx)
    ctxs = [native.Context(0) for _ in range(3)]
    multi = native.sample_dbn_host(ctxs, nm, n, T, seed, want_dense=True, want_events=True, indices=pick, **idx)
    for k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "attempts", "ev_count"):
        assert np.array_equal(one[k], multi[k]), k
    assert all(np.array_equal(a, b) for a, b in zip(one["events"], multi["events"]))
    for q in (0, 334, 335, 669, n - 1):   # rows of the list against the oracle at their own global index
        r1 = O.uncor_sample(O.OracleModel(pp), 1, T, seed, first_index=int(pick[q]), want_events=False)
        assert np.array_equal(multi["dyn_bin"][q], r1["dense_bin"][0]) and np.array_equal(multi["init_bin"][q], r1["init_bin"][0])
    # device variant and the block entry point with a device-resident list
    dev = torch.device("cuda", 0)
    G4 = (T + 3) // 4
    dpick = torch.from_numpy(pick.view(np.int64)).to(dev)
    bufs, outs = [], []
    for d in range(3):
        lo, hi = native.shard_range(n, d, 3)
        b = torch.zeros((G4, 3, hi - lo), dtype=torch.int32, device=dev)
        bufs.append(b); outs.append(dict(dyn_bin=b.data_ptr()))
    torch.cuda.synchronize()
    p, keep = native.make_params(n, T, seed, **idx)
    p.indices = dpick.data_ptr()
    native.sample_dbn_multi_device(ctxs, nm, p, outs)
    for c in ctxs:
        c.sync()
    got = np.concatenate([native.unpack_dyn_bin(b.cpu().numpy().view(np.uint32), T) for b in bufs])
    assert np.array_equal(got, one["dyn_bin"])
    db = torch.zeros((G4, 3, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    native.sample_dbn_blocks_device(gpu_ctx, [nm, nm], p, [(0, 0, 400), (1, 400, n - 400)], dyn_bin=db.data_ptr())
    gpu_ctx.sync()
    assert np.array_equal(native.unpack_dyn_bin(db.cpu().numpy().view(np.uint32), T), one["dyn_bin"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,per_step", [("uncor_1200code_v2p1", False), ("cor_v1", False), ("uncor_1200only_fwme_v1p2", True), ("haa_v1", False)])
def test_shards_written_into_one_trace_through_ld_and_col_offset(name, per_step, gpu_ctx, model_dir):
    """emgpu_sample_out.ld / col_offset on every dense kernel family and on the event-list kernel: three uneven shards of one
    index range written in place (device pointers) equal one call; so does the host entry point with ld / col_offset."""
    import torch
    nm, pp, _ = load_pair(name, model_dir)
    n, T, seed = 1000, 37, 99
    idx = uncor_indices(pp)
    mode = L.TRANSITION_PER_STEP if per_step else L.TRANSITION_REFERENCE_AUTO
    one = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=True, transition_mode=mode, **idx)
    dev = torch.device("cuda", 0)
    ni, nd, G4, cap = nm.n_initial, nm.n_dyn, (T + 3) // 4, 512
    ib = torch.zeros((ni, n), dtype=torch.uint8, device=dev); iv = torch.zeros((ni, n), dtype=torch.float32, device=dev)
    db = torch.zeros((G4, nd, n), dtype=torch.int32, device=dev); dv = torch.zeros((G4, nd, n, 4), dtype=torch.float32, device=dev)
    ec = torch.zeros(n, dtype=torch.int32, device=dev); ev = torch.zeros((n, cap, 2), dtype=torch.int32, device=dev)
    at = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for lo, hi in ((0, 70), (70, 701), (701, n)):
        p, _ = native.make_params(hi - lo, T, seed, first_index=lo, transition_mode=mode, event_cap=cap, **idx)
        native.sample_dbn_device(gpu_ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(),
                                 attempts=at.data_ptr(), ld=n, col_offset=lo)            # dense kernels
        native.sample_dbn_device(gpu_ctx, nm, p, ev_count=ec.data_ptr(), events=ev.data_ptr(), ld=n, col_offset=lo)   # event-list kernel
    gpu_ctx.sync()
    assert np.array_equal(ib.cpu().numpy().T, one["init_bin"]) and np.array_equal(iv.cpu().numpy().T, one["init_val"])
    assert np.array_equal(native.unpack_dyn_bin(db.cpu().numpy().view(np.uint32), T), one["dyn_bin"])
    assert np.array_equal(native.unpack_dyn_val(dv.cpu().numpy(), T), one["dyn_val"])
    assert np.array_equal(at.cpu().numpy(), one["attempts"])
    cnt = ec.cpu().numpy()
    assert np.array_equal(cnt, one["ev_count"].astype(np.int32))
    evh = ev.cpu().numpy().reshape(n, cap * 2).view(native.EVENT_DTYPE)
    for i in (0, 69, 70, 700, 701, n - 1):
        assert np.array_equal(evh[i, : cnt[i]], one["events"][i])
    with pytest.raises(L.EmgpuError):   # a shard that does not fit the leading dimension
        p, _ = native.make_params(10, T, seed)
        native.sample_dbn_device(gpu_ctx, nm, p, dyn_bin=db.data_ptr(), ld=n, col_offset=n - 5)


@pytest.mark.gpu
def test_multi_device_driver_equals_one_call(gpu_ctx, model_dir):
    """emgpu_sample_dbn_multi_host / _multi_device: one library call, one host thread + one stream per context.  A 1-GPU box
    has one device, so the three contexts here share it (three streams, three threads): the split, the threads, the per-shard
    host copies and the error folding are the same code that drives 8 devices."""
    import torch
    from em_model_manned_bayes_amd import sharding
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 10_001, 61, 0x5EED0002
    ctxs = [native.Context(0) for _ in range(3)]
    one = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=12345, want_dense=True, want_events=True, **idx)
    multi = native.sample_dbn_host(ctxs, nm, n, T, seed, first_index=12345, want_dense=True, want_events=True, **idx)
    for k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "attempts", "ev_count"):
        assert np.array_equal(one[k], multi[k]), k
    assert all(np.array_equal(a, b) for a, b in zip(one["events"], multi["events"]))
    assert_uncor_parity(multi, O.uncor_sample(O.OracleModel(pp), n, T, seed, first_index=12345), T)
    shard = sharding.run_sharded(nm, 3000, T, seed, devices=[0, 0], want_events=False, **idx)   # the package-level driver
    assert np.array_equal(shard["dyn_bin"], native.sample_dbn_host(gpu_ctx, nm, 3000, T, seed, want_events=False, **idx)["dyn_bin"])
    # device variant: per-context shard buffers
    dev = torch.device("cuda", 0)
    G4 = (T + 3) // 4
    bufs, outs = [], []
    for d in range(3):
        lo, hi = native.shard_range(n, d, 3)
        b = torch.zeros((G4, 3, hi - lo), dtype=torch.int32, device=dev)
        bufs.append(b); outs.append(dict(dyn_bin=b.data_ptr()))
    torch.cuda.synchronize()
    p, _ = native.make_params(n, T, seed, first_index=12345, **idx)
    native.sample_dbn_multi_device(ctxs, nm, p, outs)
    for c in ctxs:
        c.sync()
    got = np.concatenate([native.unpack_dyn_bin(b.cpu().numpy().view(np.uint32), T) for b in bufs])
    assert np.array_equal(got, one["dyn_bin"])
    # a deferred per-trajectory error on one shard surfaces from the call (rejection cap 1 cannot be met by every trajectory)
    with pytest.raises(L.EmgpuError) as e:
        native.sample_dbn_host(ctxs, nm, n, T, seed, want_dense=False, max_attempts=1, **idx)
    assert e.value.code == L.ERR_REJECT_CAP and "device" in str(e.value)


_RANK_WORKER = r"""
import os, sys, pickle
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["EMGPU_ROOT"])
from em_model_manned_bayes_amd import native, sharding, _lib as L
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)    # two ranks share the box's one GPU: RCCL would refuse
dev = torch.device("cuda", 0)
nm = native.NativeModel.load_txt(os.environ["EMGPU_MODEL"])
labs = nm.get_labels(L.F_LABELS_INITIAL)
idx = {k: (labs.index('"%s"' % v) + 1) for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\dot h"))}
n_total, T, seed = 20000, 48, 77
lo, hi = sharding.shard_range(n_total, rank, world)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
G4 = T // 4
db = torch.zeros((G4, 3, hi - lo), dtype=torch.int32, device=dev)
dv = torch.zeros((G4, 3, hi - lo, 4), dtype=torch.float32, device=dev)
p, _ = native.make_params(hi - lo, T, seed, first_index=lo, **idx)
native.sample_dbn_device(ctx, nm, p, dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
ctx.sync()
dist.barrier()
pickle.dump((lo, hi, db.cpu().numpy(), dv.cpu().numpy(), ctx.last_kernel()), open(os.environ["EMGPU_OUT"] + str(rank), "wb"))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_rank_processes_on_one_gpu(gpu_ctx, model_dir, tmp_path):
    """Two rank PROCESSES (started before they touch the GPU; gloo barrier because both sit on the box's one device) each run
    the product's sample_dbn_device on their shard of the global range: the union equals one single-process call."""
    import pickle
    import socket
    import subprocess
    nm, pp, path = load_pair("uncor_1200code_v2p1", model_dir)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EMGPU_ROOT=ROOT,
                   EMGPU_MODEL=path, EMGPU_OUT=str(tmp_path / "out"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    parts = [pickle.load(open(str(tmp_path / "out") + str(r), "rb")) for r in range(2)]
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == 20000
    one = native.sample_dbn_host(gpu_ctx, nm, 20000, 48, 77, want_dense=True, want_events=False, **uncor_indices(pp))
    assert np.array_equal(np.concatenate([native.unpack_dyn_bin(q[2].view(np.uint32), 48) for q in parts]), one["dyn_bin"])
    assert np.array_equal(np.concatenate([native.unpack_dyn_val(q[3], 48) for q in parts]), one["dyn_val"])
    assert parts[0][4] == one["kernel"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts two rank processes itself and says n_gpus 2 (--oversubscribe: this box
    has one GPU); without the flag it refuses instead of benchmarking one GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd + ["--oversubscribe"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["oversubscribed"] is True and line["value"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
        assert r.returncode == 3 and r.stdout.decode().strip() == ""
    r = subprocess.run(cmd[:2] + ["--config", "mixed", "--steps", "2", "--warmup", "1", "--n", "300000", "--no-cpu-baseline"], env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["kernel"] == "k_uncor_fast_mixed<7,4,6,6>" and len(line["config"]["models"]) == 6
    # the line holds numbers only (round 6); the verbose record of the run is the DETAIL line on stderr
    detail = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("DETAIL ")][-1][7:])
    assert line["config"]["launches_per_step"] == 1 and detail["config"]["model_blocks_per_step"] == 6
    assert len(r.stdout.decode().strip().splitlines()[-1]) < 4000 and line["config"]["notes"].startswith("profiles/")
    rf = line["roofline"]   # every timed launch listed; the write ceiling of this box measured on the step's own buffers
    assert len(rf["step_ms"]) == 2 and rf["streaming_write_GBps"] > 1000 and 0 < rf["frac_of_streaming_write"] < 1.5
    # round 6: the trace is the library's (emgpu_trace_alloc): six candidates timed before anything else, candidate 0 a plain hipMalloc block,
    # the fastest kept; the device pre-warmed; the line and the DETAIL record say so
    pc = rf["placement"]
    assert pc["candidates"] == 6 and len(pc["ms"]) == 6 and pc["ms"][pc["kept"]] == min(pc["ms"]) and pc["reused"] == 0
    assert rf["first_allocation_ms"] == pc["ms"][0] and detail["roofline"]["placement"]["bytes"] >= 300000 * 3635
    assert detail["roofline"]["prewarm"]["seconds"] == 1.0 and detail["roofline"]["prewarm"]["steps"] >= 8
    assert line["config"]["philox_rounds"] == O.philox_rounds() and line["config"]["box_state"]
    if "sclk_mhz" in rf:   # (a box that exposes its hwmon sensors) the shader clock of THIS rank's GPU under the load: an idle neighbour card reads ~100 MHz
        assert rf["sclk_mhz"] > 1000 and detail["roofline"]["gpu_telemetry"]["samples"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("config,n", [("terminal", 200000), ("mixed", 300000)])
def test_bench_ranks_of_the_other_configs_cover_disjoint_ranges(config, n, terminal_dir, model_dir, tmp_path):
    """8-GPU readiness without the node: `bench.py --gpus 2 --oversubscribe --config terminal / mixed` -- the configs
    the driver's SCALE run would shard, through the launcher on hardware.  Every rank writes the global range of every step it launched
    (--ranges-out): the ranks' ranges of a step are disjoint and tile [k W n, (k + 1) W n); the line's value is the whole job's units over
    the slowest rank's time; and each rank's last step digests to what ONE process computes for that global range (results do not depend
    on which rank ran an index)."""
    import json
    import subprocess
    import torch
    out_dir = str(tmp_path / "ranges")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    W, steps, warm = 2, 2, 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--oversubscribe", "--config", config, "--n", str(n), "--steps", str(steps),
           "--warmup", str(warm), "--no-cpu-baseline", "--telemetry-s", "0", "--ranges-out", out_dir]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == W and line["oversubscribed"] is True and line["scaling"] == "weak"
    assert abs(line["value"] - W * n * steps / (line["ms_per_step"] * steps * 1e-3)) < 1e-6 * line["value"]
    assert line["config"]["philox_rounds"] == O.philox_rounds() and "box_state" in line["config"]
    ranks = [json.load(open(os.path.join(out_dir, "rank%d.json" % q))) for q in range(W)]
    for k in range(warm + steps):
        spans = sorted((rk["ranges"][k]["first"], rk["ranges"][k]["first"] + rk["ranges"][k]["n"]) for rk in ranks)
        assert all(rk["ranges"][k]["step"] == k for rk in ranks)
        assert spans[0][0] == k * W * n and spans[-1][1] == (k + 1) * W * n
        assert all(spans[q][1] == spans[q + 1][0] for q in range(W - 1))                 # disjoint and gap-free
    dev = torch.device("cuda", 0)
    if config == "terminal":
        t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
        f = _FusedTerminal(t, n)
        for rk in ranks:
            f.run(0x5EED0005, rk["ranges"][-1]["first"])
            got = {"rows": int(f.rows.long().sum()), "model_of": int(f.mof.long().sum()), "geom_val_bits": int(f.gval.view(torch.int32).long().sum()),
                   "attempts": int(f.att.long().sum())}
            assert got == rk["digest"], (rk["rank"], got, rk["digest"])
    else:
        names = line["config"]["models"]
        assert len(names) == 6 and all(nm.endswith("_v1p2") for nm in names)
        pairs = [load_pair(nm, model_dir) for nm in names]
        ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        T, G4 = 240, 60
        for rk in ranks:
            last = rk["ranges"][-1]
            blocks = [tuple(b) for b in last["blocks"]]
            assert sum(c for _, _, c in blocks) == n and blocks[0][1] == last["first"]
            assert all(blocks[q][1] + blocks[q][2] == blocks[q + 1][1] for q in range(len(blocks) - 1))
            ib = torch.zeros((7, n), dtype=torch.uint8, device=dev); iv = torch.zeros((7, n), dtype=torch.float32, device=dev)
            db = torch.zeros((G4, 3, n), dtype=torch.int32, device=dev); dv = torch.zeros((G4, 3, n, 4), dtype=torch.float32, device=dev)
            p, _k = native.make_params(n, T, 0x5EED0004, first_index=last["first"], **uncor_indices(pairs[0][1]))
            native.sample_dbn_blocks_device(ctx, [pr[0] for pr in pairs], p, blocks, init_bin=ib.data_ptr(), init_val=iv.data_ptr(),
                                            dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(), ld=n)
            ctx.sync()
            got = {"init_bin": int(ib.long().sum()), "dyn_bin": int(db.long().sum()), "init_val_bits": int(iv.view(torch.int32).long().sum())}
            assert got == rk["digest"], (rk["rank"], got, rk["digest"])


def test_bench_launcher_with_eight_ranks(tmp_path):
    """The node layout of SCALE runs, executed on hardware before a driver ever needs it: `python bench.py --gpus 8` as its own launcher --
    eight rank processes (spawned, watched, rank 0's line relayed), here sharing the box's one GPU behind a gloo barrier (--oversubscribe,
    a test-only flag the line reports).  Tiny n: what is tested is the launcher and the rank logic, not a rate."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--n", "20000", "--no-cpu-baseline", "--oversubscribe"]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["oversubscribed"] is True and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 8 * 20000 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]   # the whole job's units / the slowest rank's time
    assert "configs" not in line and "cpu_baseline" not in line


@pytest.fixture(scope="module")
def terminal_dir(tmp_path_factory):
    from em_model_manned_bayes_amd import synthetic
    return synthetic.write_terminal_directory(str(tmp_path_factory.mktemp("terminal")))


@pytest.mark.parametrize("actypes", [("GENERIC", "GENERIC"), ("RTCA228_A1", "RTCA228_A2"), ("RTCA228_A3", "TEST")])
def test_terminal_propagation_matches_oracle(actypes, terminal_dir, gpu_ctx):
    """PropagateTrajectory (createEncounter.m:93-265) for both aircraft and both directions, on synthetic
    trajectory models (the trained files are absent from the reference mount) and real geometry samples."""
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    t.acType1, t.acType2 = actypes
    n, seed = 600, 0x5EED0005
    _, samples = t.sample(n, seed=seed, ctx=gpu_ctx)
    geo, mo = t._geo_rows(samples)
    dl = t._dyn_rows()
    files = [m.parameters_filename for m in t._traj]
    oms = []
    for f in files:
        pp = O.parse_model_txt(f)
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    ref, ref_rows = O.propagate(oms, mo, geo, seed, dl, tmax_s=120.0)
    got, rows = native.propagate_terminal_host(gpu_ctx, [m.native for m in t._traj], geo, mo, seed, tmax_s=120.0, dyn_limits=dl)
    assert np.array_equal(rows, ref_rows)
    assert rows.min() >= 1 and rows.max() <= 122
    for L_ in range(4 * n):
        r = rows[L_]
        assert_f32_of_f64(got[L_, :r], ref[L_, :r], "track %d" % L_)                 # the oracle's f64 rounded to f32, or one f32 step
    # the class method: forward + backward combined and ordered in time (createEncounter.m:74-84)
    traj = t.createEncounter(samples[:5], 120, seed=seed, ctx=gpu_ctx, local_smooth=False)
    smooth = t.createEncounter(samples[:5], 120, seed=seed, ctx=gpu_ctx)        # the default: createEncounter.m:88-89 through the stand-in
    for e_, pair in enumerate(traj):
        for a in range(2):
            tt = pair[a]["t_s"]
            assert np.all(np.diff(tt) == 1) and tt[0] <= 0 <= tt[-1]
            k0 = int(np.nonzero(tt == 0)[0][0])
            assert abs(pair[a]["v_ft_s"][k0] - samples[e_][("own", "int")[a] + "_speed"]) < 1e-3
            sm = smooth[e_][a]
            for f in ("t_s", "x_nm", "y_nm", "heading_deg"):
                assert np.array_equal(sm[f], pair[a][f])
            np.testing.assert_allclose(sm["v_ft_s"], O.local_smooth(pair[a]["v_ft_s"], 5), rtol=1e-6)
            np.testing.assert_allclose(sm["z_ft"], O.local_smooth(pair[a]["z_ft"], 15), rtol=1e-6)



def test_terminal_ten_million_encounters_properties(terminal_dir):
    """BASELINE.json configs[4] at one GPU's scale: 10 M terminal encounters (40 M tracks) propagated in five chunks of 2 M,
    checked through properties that hold for every track (createEncounter.m:160-264, :296-329) and against the oracle on slices:
    row counts within [1, tmax+2], time = +-(row index), speeds inside the dynamic limits, altitude steps inside maxVertRate,
    headings in (0, 360], every track ending for one of the reasons CheckTrajectoryConditions knows."""
    import ctypes as C
    import torch
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    seed, m, n, cap = 0x5EED0005, 8192, 2_000_000, 123
    _, samples = t.sample(m, seed=seed, ctx=ctx)
    g, mo = t._geo_rows(samples)
    reps = (n + m - 1) // m
    geo = torch.tensor(np.tile(g, (reps, 1))[:n], device=dev)
    mof = torch.tensor(np.tile(mo, (reps, 1))[:n].reshape(-1), dtype=torch.int32, device=dev)
    c0 = native.terminal_t0_row(cap)
    W = 2 * c0
    out = torch.empty((2 * n, W, 5), dtype=torch.float32, device=dev)     # the joined track of aircraft 2e + a, row c0 + t
    rows = torch.empty(4 * n, dtype=torch.int32, device=dev)
    dl = t._dyn_rows()
    handles = (C.c_void_p * 10)(*[x.native._h for x in t._traj])
    oms = []
    for f in [mm.parameters_filename for mm in t._traj]:
        pp = O.parse_model_txt(f)
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    total_seconds = 0
    lim = torch.tensor(dl, dtype=torch.float32, device=dev)[torch.arange(2 * n, device=dev) & 1]   # [2n, 5] by aircraft
    for chunk in range(5):
        out.fill_(float("nan"))                                             # a row outside a track's span must stay untouched
        p = L.TermParams()
        p.seed, p.first_index, p.n, p.tmax_s, p.max_resample, p.cap = seed, chunk * n, n, 120.0, 100000, cap
        for i, v in enumerate(dl.reshape(-1)):
            p.dyn_limits[i] = float(v)
        L.check(L.lib().emgpu_propagate_terminal_device(ctx._h, handles, 10, C.byref(p), C.c_void_p(geo.data_ptr()), C.c_void_p(mof.data_ptr()),
                                                        C.c_void_p(out.data_ptr()), C.c_void_p(rows.data_ptr())))
        ctx.sync()
        assert int(rows.min()) >= 1 and int(rows.max()) <= 122
        total_seconds += int(rows.sum())
        rf, rb = rows[0::2], rows[1::2]                                     # [2n] forward / backward rows of every aircraft
        r = torch.arange(W, device=dev)[None, :]
        valid = (r >= (c0 - (rb - 1))[:, None]) & (r <= (c0 + (rf - 1))[:, None])        # [2n, W]: the joined track's span
        written = ~torch.isnan(out[:, :, 0])
        assert bool((written == valid).all())                               # every row of the span, and nothing else (t_s = row - c0)
        v = out[:, :, 4]
        assert bool((((v >= lim[:, 0][:, None] - 1e-2) & (v <= lim[:, 1][:, None] + 1e-2)) | ~valid).all())
        hdg = out[:, :, 3]
        assert bool((((hdg > 0) & (hdg <= 360)) | ~valid).all())
        dz = (out[:, 1:, 2] - out[:, :-1, 2]).abs()
        assert bool(((dz <= lim[:, 4][:, None] * 1.0001 + 1e-2) | ~(valid[:, 1:] & valid[:, :-1])).all())
        del valid, written, dz
        if chunk in (0, 3):                                                                   # slices against the oracle
            for lo_e in (0, n - 300):
                sl = slice(lo_e, lo_e + 300)
                ref, ref_rows = O.propagate(oms, mof[4 * lo_e: 4 * lo_e + 1200].cpu().numpy(), geo[sl].cpu().numpy(), seed, dl,
                                            first_index=chunk * n + lo_e, tmax_s=120.0)
                got_rows = rows[4 * lo_e: 4 * lo_e + 1200].cpu().numpy()
                assert np.array_equal(got_rows, ref_rows)
                got = native.split_joined_tracks(np.nan_to_num(out[2 * lo_e: 2 * lo_e + 600].cpu().numpy()), got_rows, cap)
                for q in range(0, 1200, 7):
                    assert_f32_of_f64(got[q, : got_rows[q]], ref[q, : got_rows[q]], "chunk %d track %d" % (chunk, 4 * lo_e + q))
    assert 300 < total_seconds / (5 * n) < 488        # mean track-seconds per encounter (4 tracks x <= 122)


def test_terminal_propagation_reproduces_the_committed_golden(terminal_dir, gpu_ctx):
    """The HIP path against tests/golden/terminal_propagate_phx_*.npz (oracle-made; the slot map of DESIGN.md section 3, TERM_TRANS word 3
    included): track lengths bit-exact, values the golden's f64 rounded to f32 or one f32 step."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "terminal_propagate_phx_seed5eed0005_first7_32.npz"))
    n, seed, first, cap = [int(x) for x in g["meta"]]
    # the ten files in synthetic.TERMINAL_FILE_STEMS order, each intent with its OWN reverse model (the golden's list; CorTerminalModel's
    # default reproduces CorTerminalModel.m:97,100 and loads the landing reverse file three times)
    import glob as _g
    from em_model_manned_bayes_amd import synthetic
    models = []
    for stem in synthetic.TERMINAL_FILE_STEMS:
        nm = native.NativeModel.load_txt(_g.glob(os.path.join(terminal_dir, "*_" + stem + ".txt"))[0])
        nm.set_transition_stay_prior(1.0)                      # createEncounter.m:128-129
        models.append(nm)
    got, rows = native.propagate_terminal_host(gpu_ctx, models, g["geo"], g["model_of"], seed, first_index=first, tmax_s=120.0,
                                               dyn_limits=g["dyn_limits"], cap=cap)
    assert np.array_equal(rows, g["rows"])
    for q in range(4 * n):
        assert_f32_of_f64(got[q, :rows[q]], g["tracks"][q, :rows[q]], "track %d" % q)


def _terminal_oracle_models(t):
    oms = []
    for f in [m.parameters_filename for m in t._traj]:
        pp = O.parse_model_txt(f)
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))     # createEncounter.m:128-129
    return oms


class _FusedTerminal:
    """One emgpu_sample_terminal_device call (the entry point bench.py --config terminal times: k_bn -> k_terminal_geo -> k_terminal_propagate,
    @CorTerminalModel/sample.m:29-77 -> createEncounter.m:13-72) into torch buffers on cuda:0."""

    def __init__(self, t, n, cap=123):
        import torch
        self.torch, self.t, self.n, self.cap = torch, t, n, cap
        self.dev = torch.device("cuda", 0)
        self.ctx = native.Context(0, stream=torch.cuda.current_stream(self.dev).cuda_stream)
        ni = t.native.n_initial
        self.c0 = native.terminal_t0_row(cap)
        self.gbin = torch.zeros((ni, n), dtype=torch.uint8, device=self.dev)
        self.gval = torch.zeros((ni, n), dtype=torch.float32, device=self.dev)
        self.geo = torch.zeros((n, 12), dtype=torch.float64, device=self.dev)
        self.mof = torch.zeros((4 * n,), dtype=torch.int32, device=self.dev)
        self.traj = torch.empty((2 * n, 2 * self.c0, 5), dtype=torch.float32, device=self.dev)
        self.rows = torch.zeros((4 * n,), dtype=torch.int32, device=self.dev)
        self.att = torch.zeros((n,), dtype=torch.int32, device=self.dev)

    def run(self, seed, first_index=0):
        t = self.t
        self.traj.fill_(float("nan"))                     # a row outside a track's span must stay untouched
        bs = None if np.all(np.isinf(t.bounds_sample)) else t.bounds_sample
        p, self._keep = native.terminal_sample_params(t.native, self.n, seed, t._dyn_rows(), first_index=first_index, tmax_s=120.0, cap=self.cap,
                                                      bounds_sample=bs)
        native.sample_terminal_device(self.ctx, t.native, [x.native for x in t._traj], p, self.gval.data_ptr(), self.geo.data_ptr(),
                                      self.mof.data_ptr(), self.traj.data_ptr(), self.rows.data_ptr(), geom_bin=self.gbin.data_ptr(),
                                      attempts=self.att.data_ptr())
        self.ctx.sync()
        k = self.ctx.last_kernel()
        assert k.startswith("k_bn<16>") and " + k_terminal_geo + k_terminal_propagate<35,6,4>" in k, k
        assert self.ctx.last_launches() == 3

    def check_slice_against_oracle(self, oms, om_geom, seed, first_index, lo, m, stride=1):
        """Encounters [lo, lo + m) of the call against the oracle, stage by stage: (1) the geometry sample == em_geom_sample_batch
        (sample.m:29-77: bins, attempts, f32 values bit for bit); (2) geo / model_of == createEncounter.m:13-49 restated on the f32 sample
        (pyref.create_encounter_inputs; the device's sind / cosd may differ from the host's in the last bit: 4 ulp of f64); (3) rows and the
        joined tracks == em_propagate_batch on the call's own geo (createEncounter.m:52-84: lengths bit-exact, values one f32 step)."""
        import pyref
        t, n = self.t, self.n
        labs = t.labels_initial
        names = [x.replace('"', "") for x in labs]
        io, ii = labs.index('"own_speed"') + 1, labs.index('"int_speed"') + 1
        d1, d2 = t.dynLimits1, t.dynLimits2
        bs = None if np.all(np.isinf(t.bounds_sample)) else t.bounds_sample
        rb, rv, ra = O.geom_sample(om_geom, m, seed, first_index=first_index + lo, bounds_sample=bs, idx_own_speed=io, idx_int_speed=ii,
                                   lim1=(d1["minVel_ft_s"], d1["maxVel_ft_s"]), lim2=(d2["minVel_ft_s"], d2["maxVel_ft_s"]))
        gval = self.gval[:, lo: lo + m].cpu().numpy().T
        assert np.array_equal(self.gbin[:, lo: lo + m].cpu().numpy().T.astype(np.int32), rb)
        assert np.array_equal(self.att[lo: lo + m].cpu().numpy(), ra)
        assert np.array_equal(gval, rv.astype(np.float32))
        geo = self.geo[lo: lo + m].cpu().numpy()
        mof = self.mof[4 * lo: 4 * (lo + m)].cpu().numpy()
        for e_ in range(0, m, stride):
            g, mo = pyref.create_encounter_inputs(dict(zip(names, gval[e_].astype(np.float64))))
            assert list(mof[4 * e_: 4 * e_ + 4]) == mo
            np.testing.assert_allclose(geo[e_], g, rtol=1e-15, atol=1e-15)
        dl = t._dyn_rows()
        ref, ref_rows = O.propagate(oms, mof, geo, seed, dl, first_index=first_index + lo, tmax_s=120.0, cap=self.cap)
        got_rows = self.rows[4 * lo: 4 * (lo + m)].cpu().numpy()
        assert np.array_equal(got_rows, ref_rows)
        assert got_rows.min() >= 1 and got_rows.max() <= 122
        raw = self.traj[2 * lo: 2 * (lo + m)].cpu().numpy()
        span = np.zeros(raw.shape[:2], dtype=bool)                                   # only rows C-(rb-1) .. C+(rf-1) are written
        for a in range(2 * m):
            span[a, self.c0 - (got_rows[2 * a + 1] - 1): self.c0 + got_rows[2 * a]] = True
        assert np.array_equal(~np.isnan(raw[:, :, 0]), span)
        got = native.split_joined_tracks(np.nan_to_num(raw), got_rows, self.cap)
        worst = 0
        for q in range(0, 4 * m, stride):
            r = got_rows[q]
            worst = max(worst, assert_f32_of_f64(got[q, :r], ref[q, :r], "track %d" % (4 * lo + q)))
        return worst


@pytest.mark.parametrize("actypes", [("GENERIC", "GENERIC"), ("RTCA228_A1", "RTCA228_A2"), ("RTCA228_A3", "TEST")])
def test_fused_terminal_call_matches_oracle(actypes, terminal_dir):
    """emgpu_sample_terminal_device -- the call `bench.py --config terminal` times -- against the oracle, stage by stage, on three
    aircraft-type pairs (the speed rejection of sample.m:64-70 and the limits of PropagateTrajectory differ between them)."""
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    t.acType1, t.acType2 = actypes
    n, seed, first = 2000, 0x5EED0005, 12345
    f = _FusedTerminal(t, n)
    f.run(seed, first)
    om_geom = O.OracleModel(O.parse_model_txt(t.parameters_filename))
    f.check_slice_against_oracle(_terminal_oracle_models(t), om_geom, seed, first, 0, n)
    if actypes[0] != "GENERIC":
        assert int(f.att.max()) > 1              # the rejection loop ran inside the fused call
    # a bounds box (sample.m:45-53) travels through the fused call's host-pointer argument
    bs = np.column_stack([-np.inf * np.ones(15), np.inf * np.ones(15)])
    bs[t.labels_initial.index('"own_distance"')] = [0, 3]
    t.bounds_sample = bs
    f.run(seed + 1, 0)
    f.check_slice_against_oracle(_terminal_oracle_models(t), om_geom, seed + 1, 0, 0, 400)
    assert float(f.gval[t.labels_initial.index('"own_distance"')].max()) <= 3


def test_fused_terminal_call_equals_the_chain_of_its_parts(terminal_dir):
    """2 M encounters: the fused call's bytes == emgpu_sample_bn_device -> (the host's _geo_rows is NOT used: the call's own geo) ->
    emgpu_propagate_terminal_device, the entry points the other terminal tests pin one by one."""
    import torch
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    n, seed, first, cap = 2_000_000, 0x5EED0005, 7 * 2_000_000, 123
    f = _FusedTerminal(t, n, cap)
    f.run(seed, first)
    dev, ctx = f.dev, f.ctx
    ni = t.native.n_initial
    # part 1: the geometry draw alone
    labs = t.labels_initial
    bp = L.BnParams()
    bp.seed, bp.first_index, bp.n, bp.max_attempts = seed, first, n, 100000
    bp.idx_own_speed, bp.idx_int_speed = labs.index('"own_speed"') + 1, labs.index('"int_speed"') + 1
    d1, d2 = t.dynLimits1, t.dynLimits2
    bp.min_vel1, bp.max_vel1, bp.min_vel2, bp.max_vel2 = d1["minVel_ft_s"], d1["maxVel_ft_s"], d2["minVel_ft_s"], d2["maxVel_ft_s"]
    gbin = torch.zeros((ni, n), dtype=torch.uint8, device=dev)
    gval = torch.zeros((ni, n), dtype=torch.float32, device=dev)
    att = torch.zeros((n,), dtype=torch.int32, device=dev)
    L.check(L.lib().emgpu_sample_bn_device(ctx._h, t.native._h, C.byref(bp), C.c_void_p(gbin.data_ptr()), C.c_void_p(gval.data_ptr()),
                                           C.c_void_p(att.data_ptr())))
    ctx.sync()
    assert torch.equal(gbin, f.gbin) and torch.equal(gval, f.gval) and torch.equal(att, f.att)
    del gbin, gval, att
    # part 2: the propagation alone, from the fused call's geo / model_of
    out = torch.full((2 * n, 2 * f.c0, 5), float("nan"), dtype=torch.float32, device=dev)
    rows = torch.zeros(4 * n, dtype=torch.int32, device=dev)
    p = L.TermParams()
    p.seed, p.first_index, p.n, p.tmax_s, p.max_resample, p.cap = seed, first, n, 120.0, 100000, cap
    for i, v in enumerate(t._dyn_rows().reshape(-1)):
        p.dyn_limits[i] = float(v)
    handles = (C.c_void_p * 10)(*[x.native._h for x in t._traj])
    L.check(L.lib().emgpu_propagate_terminal_device(ctx._h, handles, 10, C.byref(p), C.c_void_p(f.geo.data_ptr()), C.c_void_p(f.mof.data_ptr()),
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(rows.data_ptr())))
    ctx.sync()
    assert torch.equal(rows, f.rows)
    assert torch.equal(out.view(torch.int32), f.traj.view(torch.int32))          # bit for bit, the untouched rows' NaN pattern included
    # and the geo rows are what the class layer's host code builds from the same sample (createEncounter.m:41-49), to the last bits of sind / cosd
    m = 5000
    names = [x.replace('"', "") for x in labs]
    samples = [dict(zip(names, row)) for row in f.gval[:, :m].cpu().numpy().T.astype(np.float64)]
    g, mo = t._geo_rows(samples)
    assert np.array_equal(mo.reshape(-1), f.mof[: 4 * m].cpu().numpy())
    np.testing.assert_allclose(f.geo[:m].cpu().numpy(), g, rtol=1e-15, atol=1e-15)


def test_fused_terminal_call_at_the_benchmark_shape(terminal_dir):
    """ONE emgpu_sample_terminal_device call of 12.5 M encounters -- bench.py's default step for config 5 (one GPU's share of 100 M): fresh
    geometry per encounter, 50 M tracks, 131 GB of joined tracks -- through the properties every track must have (createEncounter.m:160-264,
    :296-329) and against the oracle on slices at the start, in the middle and at the end of the batch."""
    import torch
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    n, seed, first, cap = 12_500_000, 0x5EED0005, 3 * 12_500_000, 123        # rank 3's shard of the 100 M job
    f = _FusedTerminal(t, n, cap)
    f.run(seed, first)
    dev, c0, W = f.dev, f.c0, 2 * f.c0
    rows = f.rows
    assert int(rows.min()) >= 1 and int(rows.max()) <= 122
    total_seconds = int(rows.sum())
    assert 300 < total_seconds / n < 488
    dl = torch.tensor(t._dyn_rows(), dtype=torch.float32, device=dev)
    labs = t.labels_initial
    # the geometry sample: inside the speed limits of its aircraft (sample.m:64-70), intents valid, model_of by intent (createEncounter.m:13-38)
    for a, pre in enumerate(("own", "int")):
        v = f.gval[labs.index('"%s_speed"' % pre)]
        assert bool(((v >= dl[a, 0]) & (v <= dl[a, 1])).all())
    oi, ii = f.gval[labs.index('"own_intent"')].to(torch.int32), f.gval[labs.index('"int_intent"')].to(torch.int32)
    assert int(oi.min()) >= 1 and int(oi.max()) <= 2 and int(ii.min()) >= 1 and int(ii.max()) <= 3
    mo = f.mof.view(n, 4)
    assert bool((mo[:, 0] == 2 * (oi - 1)).all() and (mo[:, 1] == mo[:, 0] + 1).all() and (mo[:, 2] == 4 + 2 * (ii - 1)).all() and (mo[:, 3] == mo[:, 2] + 1).all())
    assert bool((f.geo[:, 2] == f.gval[labs.index('"own_alt"')].double()).all() and (f.geo[:, 9] == f.gval[labs.index('"int_speed"')].double()).all())
    # the tracks, in chunks of 1 M aircraft (a [2n, W] mask of 25 M x 256 would not fit beside 131 GB of tracks)
    r = torch.arange(W, device=dev)[None, :]
    step = 1_000_000
    for lo in range(0, 2 * n, step):
        hi = min(2 * n, lo + step)
        blk = f.traj[lo:hi]
        rf, rb = rows[2 * lo: 2 * hi: 2], rows[2 * lo + 1: 2 * hi: 2]
        valid = (r >= (c0 - (rb - 1))[:, None]) & (r <= (c0 + (rf - 1))[:, None])
        assert bool(((~torch.isnan(blk[:, :, 0])) == valid).all())            # every row of the span and nothing else
        lim = dl[torch.arange(lo, hi, device=dev) & 1]
        v = blk[:, :, 4]
        assert bool((((v >= lim[:, 0][:, None] - 1e-2) & (v <= lim[:, 1][:, None] + 1e-2)) | ~valid).all())
        hdg = blk[:, :, 3]
        # wrapTo360(atan2d(v)) (:176) is in [0, 360) -- 0 itself when the velocity points along +x exactly: a geometry heading within half an f32
        # step of 360 is stored as 360.0f, cosd / sind(360) are (1, 0) exactly, atan2d(0, v) = 0 (met at this batch size, never on 8 192 tiled
        # geometries); a value just below 360 rounds to 360.0f on store
        assert bool((((hdg >= 0) & (hdg <= 360)) | ~valid).all())
        dz = (blk[:, 1:, 2] - blk[:, :-1, 2]).abs()
        assert bool(((dz <= lim[:, 4][:, None] * 1.0001 + 1e-2) | ~(valid[:, 1:] & valid[:, :-1])).all())
        # t = 0 row = the geometry sample (createEncounter.m:41-45): z, heading, speed of aircraft 2e + a
        e_idx = torch.arange(lo, hi, device=dev) // 2
        a_idx = torch.arange(lo, hi, device=dev) & 1
        g6 = f.geo.view(n, 2, 6)[e_idx, a_idx]
        # (z is handed through; the speed is norm(rotationmatrix(heading0) * [v0; 0]), :155,:172 -- v0 to an f32 step)
        assert bool((blk[:, c0, 2] == g6[:, 2].float()).all())
        assert bool(((blk[:, c0, 4] - g6[:, 3].float()).abs() <= 2.4e-7 * g6[:, 3].float()).all())
        del valid, dz, v, hdg
    oms = _terminal_oracle_models(t)
    om_geom = O.OracleModel(O.parse_model_txt(t.parameters_filename))
    for lo in (0, n // 2 - 150, n - 300):
        f.check_slice_against_oracle(oms, om_geom, seed, first, lo, 300, stride=3)


@pytest.mark.parametrize("actypes,n,cap,cum_override,smooth", [(("GENERIC", "GENERIC"), 2000, 150, None, False), (("GENERIC", "RTCA228_A1"), 120, 600, None, False),
                                                               (("RTCA228_A3", "RTCA228_A2"), 120, 600, None, False), (("GENERIC", "GENERIC"), 600, 150, (40.0, 40.0), False),
                                                               (("GENERIC", "GENERIC"), 1000, 150, None, True), (("GENERIC", "RTCA228_A1"), 120, 600, None, True)])
def test_terminal_track_matches_oracle(actypes, n, cap, cum_override, smooth, terminal_dir, gpu_ctx):
    """CorTerminalModel.track (track.m:45-150) on the GPU -- rounds of geometry draw -> createEncounter inputs -> propagation ->
    the filters of CorTerminalModel.m:117-316 -- against the oracle's per-encounter loop on the same Philox keys (attempt j:
    seed + j).  EVERY encounter must accept the same attempt (or stay rejected through the cap on both sides), except where the
    oracle's decision margin of the parting attempt is at rounding level: the filters read the tracks as the f32 values the
    propagation stores, and those agree between device and host to 1e-6 only (libm's last bits before the rounding), so a value
    may sit on a threshold to within two f32 ulps (tolerance 2^-22 relative to the value's own scale; oracle em_note covers
    every discretize of the propagation, every limit test, the CPA argmin, the fractions and the rounding of CheckCumTurn).
    The geometry sample, lengths and CPA metadata of agreeing encounters are identical, the tracks agree to 1e-6.
    cum_override: GENERIC limits with a 40 degree cumulative-turn limit -- CheckCumTurn (CorTerminalModel.m:135-185) becomes the
    deciding filter for a good part of the encounters (checked against a run without it)."""
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    t.acType1, t.acType2 = actypes
    gom = O.OracleModel(O.parse_model_txt(t.parameters_filename))
    oms = []
    for m in t._traj:
        pp = O.parse_model_txt(m.parameters_filename)
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    d = (t.dynLimits1, t.dynLimits2)
    cum, pitch = [x["maxCumTurn_deg"] for x in d], [x["pitch_deg"] for x in d]
    if cum_override:
        cum = list(cum_override)
    seed = 0xF2
    # smooth: EMGPU_FLAG_LOCAL_SMOOTH on both sides -- createEncounter.m:88-89 through the documented stand-in (em-core's local_smooth is not
    # vendored: UNPINNED); the filters then read the smoothed speed and altitude, like the reference's
    ref = O.terminal_track(gom, oms, n, seed, t._dyn_rows(), cum, pitch, first_index=5, max_track_attempts=cap, local_smooth=smooth)
    got = native.track_terminal_host(gpu_ctx, t.native, [m.native for m in t._traj], n, seed, t._dyn_rows(), cum, pitch, first_index=5,
                                     max_track_attempts=cap, allow_cap=True, local_smooth=smooth)   # encounters still rejected after `cap` attempts: -1 on both sides
    assert ("k_terminal_smooth" in got["kernel"]) == smooth
    if (ref["attempts"] < 0).any() and n <= 200:
        with pytest.raises(L.EmgpuError) as ei:
            native.track_terminal_host(gpu_ctx, t.native, [m.native for m in t._traj], n, seed, t._dyn_rows(), cum, pitch, first_index=5, max_track_attempts=cap,
                                       local_smooth=smooth)
        assert ei.value.code == L.ERR_REJECT_CAP
    assert "k_terminal_filter" in got["kernel"] and "k_terminal_propagate" in got["kernel"]
    same = assert_parting_only_on_a_threshold(got["attempts"], ref["attempts"], ref["margins"], 2.0 ** -22, "encounter")
    assert same.sum() >= n - max(2, n // 500), "more threshold coincidences than %d encounters can explain: %d" % (n, (~same).sum())
    ok = same & (ref["attempts"] > 0)
    if actypes == ("GENERIC", "GENERIC"):   # (the RTCA limits -- pitch 15 deg, cumulative turn 180 deg, narrow speed bands -- reject most synthetic tracks:
        assert ok.sum() >= n // 5 and (ref["attempts"][ok] > 1).any()   #  there the test is that both sides reject the same attempts)
    if cum_override:   # CheckCumTurn decides: without the limit the same encounters accept an earlier attempt
        free = O.terminal_track(gom, oms, n, seed, t._dyn_rows(), [np.inf, np.inf], pitch, first_index=5, max_track_attempts=cap, local_smooth=smooth)
        decided = (free["attempts"] > 0) & (free["attempts"] != ref["attempts"])
        assert decided.sum() >= n // 20, decided.sum()
    assert np.array_equal(got["sample"][ok], ref["sample"][ok]) and np.array_equal(got["len"][ok], ref["len"][ok])
    np.testing.assert_allclose(got["meta"][ok], ref["meta"][ok], rtol=1e-5, atol=1e-3)
    for i in np.flatnonzero(ok)[:400]:
        for a in range(2):
            k = ref["len"][i, a]
            assert_f32_of_f64(got["traj"][i, a, :k, 1:], ref["traj"][i, a, :k, 1:], "encounter %d aircraft %d" % (i, a), abs_floor=1e-6 if smooth else 1e-9)
            assert np.array_equal(got["traj"][i, a, :k, 0], ref["traj"][i, a, :k, 0])
            assert np.all(np.diff(got["traj"][i, a, :k, 0]) == 1)           # time-ordered 1 s samples through t = 0
    assert np.all(np.abs(got["meta"][ok, 0]) <= 10) and np.all(got["meta"][ok, 3] >= 30)     # track.m:86, :91


def test_terminal_class_track_and_cum_turn(terminal_dir, gpu_ctx):
    """The class method (out_results(ii).sample / .traj, reformatTrajFiles) and CheckCumTurn known answers."""
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    out, gen_time, info = t.track(12, initialSeed=3, max_track_attempts=3000, ctx=gpu_ctx, return_info=True)
    assert len(out) == 12 and gen_time.shape == (12,) and np.all(info["attempts"] >= 1)
    for r in out:
        s, tr = r["sample"], r["traj"]
        assert {"own_intent", "int_intent", "id", "tcpa", "hmd_ft", "vmd_ft", "nmac", "own_initSpeed_ftps"} <= set(s)
        assert len(tr) == 2 and tr[0]["t"][0] == 0 and np.array_equal(tr[0]["t"], tr[1]["t"]) and len(tr[0]["t"]) >= 30
        assert s["own_intent"] in (1, 2) and s["nmac"] == (abs(s["hmd_ft"]) < 500 and abs(s["vmd_ft"]) < 100)
    # CheckCumTurn (CorTerminalModel.m:135-185) on hand-made headings: a steady 3 deg/s turn through 210 deg is rejected at 180,
    # the same amount split into a left and a right turn is not
    steady = np.concatenate([np.zeros(5), np.arange(0, 211, 3.0), np.full(5, 210.0)])
    zigzag = np.concatenate([np.zeros(5), np.arange(0, 106, 3.0), np.arange(105, -1, -3.0), np.zeros(5)])
    assert O.check_cum_turn(steady, 180.0) and not O.check_cum_turn(zigzag, 180.0) and not O.check_cum_turn(steady, np.inf)


def test_edge_cases_and_error_paths(gpu_ctx, model_dir):
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    om = O.OracleModel(pp)
    # empty batch
    got = native.sample_dbn_host(gpu_ctx, nm, 0, 16, 1, want_dense=True, want_events=True, **idx)
    assert got["init_bin"].shape == (0, 7) and got["dyn_bin"].shape == (0, 16, 3) and got["events"] == []
    # a single trajectory, a single second (no transition at all: dbn_sample.m:66,138 loops run 2:t_max)
    for want_events in (False, True):
        got = native.sample_dbn_host(gpu_ctx, nm, 1, 1, 9, want_dense=True, want_events=want_events, **idx)
        ref = O.uncor_sample(om, 1, 1, 9)
        assert_uncor_parity(got, ref, 1)
    if True:
        assert ref["events"][0].shape[0] >= 1 and ref["events"][0][-1, 1] == 0     # only resample rows + the terminator
    # long trajectories, ragged batch sizes around the 64 / 256 lane boundaries
    for n, T in ((63, 1000), (65, 999), (255, 17), (257, 16), (1, 4001)):
        ref = O.uncor_sample(om, n, T, 123, first_index=2**40, want_events=False)
        got = native.sample_dbn_host(gpu_ctx, nm, n, T, 123, first_index=2**40, want_dense=True, want_events=False, **idx)
        assert_uncor_parity(got, ref, T)
    # event lists that do not fit event_cap: deferred EMGPU_ERR_EVENT_CAP (the counts are still reported)
    with pytest.raises(E.EmgpuError) as ei:
        native.sample_dbn_host(gpu_ctx, nm, 500, 240, 5, want_dense=False, want_events=True, event_cap=8, **idx)
    assert ei.value.code == L.ERR_EVENT_CAP
    # rejection cap: preset a start that can never satisfy v*1.68781 > |dh|/60 often enough within 1 attempt
    nm.set_start([1, 4, 2, 1, 2, 1, 0])          # slowest speed bin with the steepest descent bin (parents preset too)
    try:
        with pytest.raises(E.EmgpuError) as ei:
            native.sample_dbn_host(gpu_ctx, nm, 4000, 8, 5, want_dense=True, max_attempts=1, **idx)
        assert ei.value.code == L.ERR_REJECT_CAP
        ref = O.uncor_sample(O.OracleModel(pp, start=[1, 4, 2, 1, 2, 1, 0]), 300, 8, 5, max_attempts=100000)
        got = native.sample_dbn_host(gpu_ctx, nm, 300, 8, 5, want_dense=True, want_events=True, max_attempts=100000, **idx)
        assert ref["attempts"].max() > 3
        assert_uncor_parity(got, ref, 8)
    finally:
        nm.set_start([0] * 7)
    # arguments
    for bad in (dict(sample_time=0), dict(sample_time=70000)):
        with pytest.raises(E.EmgpuError):
            native.sample_dbn_host(gpu_ctx, nm, 4, bad["sample_time"], 1, **idx)
    with pytest.raises(E.EmgpuError):
        native.sample_dbn_host(gpu_ctx, nm, 4, 8, 1, layers=np.array([[0, 1]] * 4), **idx)   # L is not a bin index here


def test_large_sample_frequencies_match_the_cpts(gpu_ctx, model_dir):
    """4 M trajectories from the HIP path: empirical frequencies of the root node, of A | G and of the
    first transition of \\dot\\psi given the most common initial configuration against the normalised CPT
    columns (chi-square), i.e. the distribution is the reference's, not just the oracle's."""
    import torch
    from scipy import stats
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T = 4_000_000, 8
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ib = torch.empty((7, n), dtype=torch.uint8, device=dev)
    db = torch.empty((2, 3, n), dtype=torch.int32, device=dev)
    p, _ = native.make_params(n, T, 20261003, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), dyn_bin=db.data_ptr())
    ctx.sync()
    ibc = ib.cpu().numpy()
    # root node G
    w = pp["N_initial"][0][:, 0]
    obs = np.bincount(ibc[0], minlength=5)[1:]
    assert stats.chisquare(obs, w / w.sum() * n).pvalue > 1e-4
    # A | G = 1.  (The rejection test v*1.68781 > |dh|/60 removes <1e-4 of the samples: no visible bias here.)
    sel = ibc[0] == 1
    w = pp["N_initial"][1][:, 0]
    obs = np.bincount(ibc[1][sel], minlength=5)[1:]
    assert obs[w == 0].sum() == 0
    assert stats.chisquare(obs[w > 0], w[w > 0] / w.sum() * sel.sum()).pvalue > 1e-4
    # first transition of \dot\psi for the most common initial configuration (frozen-parent column, dbn_sample.m:110-135)
    keys = (ibc.astype(np.int64) * (10 ** np.arange(7))[:, None]).sum(axis=0)
    vals, counts = np.unique(keys, return_counts=True)
    common = vals[np.argmax(counts)]
    sel = keys == common
    cfg = [(common // 10 ** k) % 10 for k in range(7)]
    G = pp["G_transition"]; r = pp["r_transition"]
    par = np.nonzero(G[:, 9])[0]
    j, kk = 0, 1
    for q in par:
        j += kk * (cfg[q] - 1); kk *= r[q]
    w = pp["N_transition"][9][:, j]
    col1 = (db[0, 2].cpu().numpy().view(np.uint32) >> 8) & 0xFF          # second 1 of \dot\psi
    obs = np.bincount(col1[sel], minlength=8)[1:]
    assert obs[w == 0].sum() == 0
    exp = w[w > 0] / w.sum() * sel.sum()
    keep = exp >= 5
    o, e_ = obs[w > 0][keep], exp[keep]
    assert stats.chisquare(o, e_ * o.sum() / e_.sum()).pvalue > 1e-4


def test_sample2track_kernel_matches_oracle(gpu_ctx):
    """emgpu_sample2track_host (f64 planar input) against sample2track.m:183-243 restated."""
    rng = np.random.RandomState(3)
    n, T = 3000, 61
    alt0 = rng.uniform(-50, 12000, n)
    v0 = rng.uniform(20, 320, n)
    upd = np.stack([rng.normal(0, 800, (n, T)), rng.normal(0, 1.0, (n, T)), rng.normal(0, 3.0, (n, T))], axis=2)
    upd[::7, :, 2] = 22.5                      # headings that pass through exact multiples of 90 degrees
    upd[::11] = 0.0
    ur = ((1852.0 / 0.3048) / 3600.0, 1.0 / 60.0, 1.0)
    ref_xyz, ref_fl, ref_vmm = O.sample2track(alt0, v0, upd, *ur, 30.0, 300.0)
    xyz, fl, vmm = native.sample2track_host(gpu_ctx, alt0, v0, upd, *ur, 30.0, 300.0)
    assert gpu_ctx.last_kernel() == "k_sample2track<planar>"
    assert np.array_equal(fl, ref_fl) and 0 < (fl == 0).sum() < n and (fl & 1).any() and (fl & 2).any()
    np.testing.assert_allclose(xyz, ref_xyz, rtol=1e-12, atol=1e-7)      # f64 on both sides; sin/cos differ by ulps
    np.testing.assert_allclose(vmm, ref_vmm, rtol=1e-14)
    assert np.array_equal(xyz[::11][:, :, 1], np.zeros_like(xyz[::11][:, :, 1]))
    # T not a multiple of 4, n not a multiple of 256, single row
    for n2, T2 in [(1, 1), (257, 2), (130, 7)]:
        a, b = native.sample2track_host(gpu_ctx, alt0[:n2], v0[:n2], upd[:n2, :T2], *ur, 30.0, 300.0)[:2]
        ra, rb = O.sample2track(alt0[:n2], v0[:n2], upd[:n2, :T2], *ur, 30.0, 300.0)[:2]
        np.testing.assert_allclose(a, ra, rtol=1e-12, atol=1e-7)
        assert np.array_equal(b, rb)


def test_sample2track_consumes_the_dense_trace_on_the_device(gpu_ctx, model_dir):
    """The device consumer: sampler output stays in HBM, k_sample2track<dense> reads it in place."""
    import torch
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 4000, 61, 5
    ni, nd, G4 = 7, 3, (T + 3) // 4
    dev = torch.device("cuda:0")
    iv = torch.zeros((ni, n), dtype=torch.float32, device=dev)
    dv = torch.zeros((G4, nd, n, 4), dtype=torch.float32, device=dev)
    xyz = torch.zeros((T + 1, 3, n), dtype=torch.float64, device=dev)
    fl = torch.zeros(n, dtype=torch.uint8, device=dev)
    vmm = torch.zeros((2, n), dtype=torch.float64, device=dev)
    gpu_ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p, _keep = native.make_params(n, T, seed, **idx)
    native.sample_dbn_device(gpu_ctx, nm, p, init_val=iv.data_ptr(), dyn_val=dv.data_ptr())
    tm = np.asarray(pp["temporal_map"]).reshape(-1, 2)
    lab = [pp["labels_initial"][int(r[0]) - 1] for r in tm]
    s_acc, s_vr, s_tr = lab.index('"\\dot v"'), lab.index('"\\dot h"'), lab.index('"\\dot \\psi"')
    b_v = np.asarray(pp["boundaries"][idx["idx_v"] - 1])
    ur = ((1852.0 / 0.3048) / 3600.0, 1.0 / 60.0, 1.0)
    tp = native.track_params(n, T, *ur, float(b_v[0]), float(b_v[-1]), nd=nd, slot_vertrate=s_vr, slot_acc=s_acc, slot_turnrate=s_tr)
    native.sample2track_device(gpu_ctx, tp, iv[idx["idx_L"] - 1].data_ptr(), iv[idx["idx_v"] - 1].data_ptr(), dv.data_ptr(),
                               xyz.data_ptr(), fl.data_ptr(), vmm.data_ptr())
    gpu_ctx.sync()
    assert gpu_ctx.last_kernel() == "k_sample2track<dense>"
    gpu_ctx.set_stream(None)
    ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, want_events=False)
    upd = ref["dense_val"].astype(np.float32).astype(np.float64)[:, :, [s_vr, s_acc, s_tr]]
    a0 = ref["init_val"][:, idx["idx_L"] - 1].astype(np.float32).astype(np.float64)
    v0 = ref["init_val"][:, idx["idx_v"] - 1].astype(np.float32).astype(np.float64)
    rx, rf, rv = O.sample2track(a0, v0, upd, *ur, float(b_v[0]), float(b_v[-1]))
    np.testing.assert_allclose(xyz.cpu().numpy().transpose(2, 0, 1), rx, rtol=1e-12, atol=1e-7)
    assert np.array_equal(fl.cpu().numpy(), rf)
    np.testing.assert_allclose(vmm.cpu().numpy().T, rv, rtol=1e-14)


def test_em_sample_and_sample2track_files(gpu_ctx, model_dir, tmp_path):
    """RUN_1_emsample / RUN_2_sample2track end to end: file formats, values, directories, rejection."""
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    fi, ft = str(tmp_path / "initial.txt"), str(tmp_path / "transition.txt")
    n, T, seed = 120, 60, 42
    initial, trace = E.em_sample(path, initial_output_filename=fi, transition_output_filename=ft, num_initial_samples=n,
                                 num_transition_samples=T, rng_seed=seed, ctx=gpu_ctx)
    li, lt = open(fi).read().split("\n"), open(ft).read().split("\n")
    assert li[0] == 'id "G" "A" "L" "v" "\\dot v" "\\dot h" "\\dot \\psi" ' and li[-1] == "" and len(li) == n + 2    # em_sample.m:64-68
    assert lt[0] == 'initial_id t "\\dot v(t+1)" "\\dot h(t+1)" "\\dot \\psi(t+1)" ' and len(lt) == n * T + 2     # :71-75
    assert lt[1].split(" ")[:2] == ["1", "0"] and lt[T].split(" ")[:2] == ["1", str(T - 1)] and lt[T + 1].split(" ")[0] == "2"
    # values: dbn_hierarchical_sample without the rejection test == the oracle's never-rejected samples
    pp = O.parse_model_txt(path)
    ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, want_events=False)
    keep = np.nonzero(ref["attempts"] == 1)[0]
    assert keep.size > n // 2
    Ti = np.loadtxt(fi, skiprows=1, ndmin=2)
    Tt = np.loadtxt(ft, skiprows=1, ndmin=2).reshape(n, T, 5)
    assert np.array_equal(Ti[:, 0], np.arange(1, n + 1)) and np.array_equal(Tt[0, :, 1], np.arange(T))
    np.testing.assert_allclose(Ti[keep, 1:], ref["init_val"][keep], rtol=1.1e-5, atol=1e-12)        # %g keeps 6 significant digits
    np.testing.assert_allclose(Tt[keep][:, :, 2:], ref["dense_val"][keep], rtol=1.1e-5, atol=1e-12)
    assert np.array_equal(initial[keep].astype(np.float32), ref["init_val"][keep].astype(np.float32))
    # tracks
    out = str(tmp_path / "tracks")
    is_good, T_initial = E.sample2track(path, fi, ft, out_dir_parent=out, verbose=False, ctx=gpu_ctx)
    ur = ((1852.0 / 0.3048) / 3600.0, 1.0 / 60.0, 1.0)
    b_v = np.asarray(pp["boundaries"][3])
    rx, rf, _ = O.sample2track(Ti[:, 3], Ti[:, 4], Tt[:, :, [3, 2, 4]], *ur, float(b_v[0]), float(b_v[-1]))
    assert np.array_equal(is_good, rf == 0) and is_good.any()
    np.testing.assert_allclose(T_initial["v"], Ti[:, 4] * ur[0], rtol=1e-15)
    files = sorted(glob.glob(os.path.join(out, "G*", "A*", "*ft", "BAYES_t%d_id*_alt*_speed*.csv" % T)))
    assert len(files) == int(is_good.sum())
    i = int(np.nonzero(is_good)[0][0])
    mine = [f for f in files if "_id%d_" % (i + 1) in f]
    assert len(mine) == 1
    parts = mine[0].split(os.sep)
    assert parts[-4] == "G%d" % Ti[i, 1] and parts[-3] == "A%d" % Ti[i, 2]
    alt_dir = int(parts[-2][:-2])
    assert alt_dir <= Ti[i, 3] < alt_dir + 100 + 1e-9
    rows = open(mine[0]).read().split("\n")
    assert rows[0] == "time_s,x_ft,y_ft,z_ft" and len(rows) == T + 3
    csv = np.array([[float(v) for v in r.split(",")] for r in rows[1:-1]])
    assert np.array_equal(csv[:, 0], np.arange(T + 1))
    assert np.max(np.abs(csv[:, 1:] - rx[i])) <= 0.5 + 1e-6           # %0.0f
    # a track that starts below the speed range is rejected and writes nothing
    assert not glob.glob(os.path.join(out, "**", "*_id%d_*" % (int(np.nonzero(~is_good)[0][0]) + 1 if (~is_good).any() else 0)), recursive=True)


def test_fallback_per_step_kernel_still_matches_oracle(model_dir):
    """k_dbn_step (columns of any width up to 9 bins) is the fallback of k_dbn_step2; the dispatch is decided
    once per process, so the fallback is exercised in a child process with EMGPU_DEBUG_NO_STEP2 set."""
    import subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle as O
from em_model_manned_bayes_amd import native
from util import load_pair, uncor_indices, assert_uncor_parity, assert_parting_only_on_a_threshold, assert_f32_of_f64
ctx = native.Context(0)
for name, T, n in [("glider_v1", 61, 600), ("cor_v1", 24, 300)]:
    nm, pp, _ = load_pair(name, %r)
    idx = uncor_indices(pp) if name != "cor_v1" else {}
    got = native.sample_dbn_host(ctx, nm, n, T, 11, want_dense=True, want_events=False, **idx)
    assert got["kernel"].startswith("k_dbn_step<"), got["kernel"]
    om = O.OracleModel(pp)
    ref = O.uncor_sample(om, n, T, 11, want_events=False)
    assert_uncor_parity(got, ref, T, check_events=False)
print("fallback ok")
''' % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"), os.path.join(ROOT_DIR, "oracle"), str(model_dir))
    env = dict(os.environ, EMGPU_DEBUG_NO_STEP2="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("rows", ["lane", "wide", "long"])
def test_every_fast_model_through_the_per_lane_event_kernels(rows, model_dir):
    """A list asked for alone comes from k_uncor_fast_evu (rows built by the wave: the tests above); k_uncor_fast_ev / k_uncor_fast_evw
    (result slots + a row loop per lane) serve the calls that want the dense trace as well.  A child process with
    EMGPU_DEBUG_EVENT_ROWS sends every fast-branch model's lists through them too ("lane": ev, evw for haa_v1; "wide": evw, whose
    instance takes any fast-branch shape; "long": k_uncor_fast_evu_long, the rows-by-the-wave form with the 1 024-request queue that only
    haa_v1 takes by default): row for row against the oracle, lengths on and off the block boundary, an overrun capacity."""
    import subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle as O
from em_model_manned_bayes_amd import native, _lib as L
from util import load_pair, uncor_indices
ctx = native.Context(0)
rows = %r
for name in %r:
    fast = name in %r
    want = ("k_uncor_fast_evu_long" if rows == "long" else "k_uncor_fast_evw" if rows == "wide" or name == "haa_v1" else "k_uncor_fast_ev<") if fast else "k_dbn_step2"
    nm, pp, _ = load_pair(name, %r)
    idx = uncor_indices(pp)
    for T, n, cap in [(240, 1500, 1024), (8, 300, 64), (13, 500, 64), (1, 100, 8), (33, 700, 4096)]:
        seed, first = 0xE7E8, 2**35 + 7
        ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, mode=O.RNG_PHILOX, first_index=first)
        got = native.sample_dbn_host(ctx, nm, n, T, seed, first_index=first, want_dense=False, want_events=True, event_cap=cap, **idx)
        assert got["kernel"].startswith(want) and "rows-by-wave" not in got["kernel"], got["kernel"]
        assert np.array_equal(got["ev_count"], np.array([len(e) for e in ref["events"]]))
        for i in range(n):
            g, r = got["events"][i], ref["events"][i]
            assert np.array_equal(g["dt"], r[:, 0]) and np.array_equal(g["var"], r[:, 1]) and np.array_equal(g["bin"], r[:, 3]), (name, T, i)
            assert np.array_equal(g["value"], r[:, 2].astype(np.float32)), (name, T, i)
    # a capacity that some lists overrun: the error, and the counts still exact
    try:
        native.sample_dbn_host(ctx, nm, 400, 240, 5, want_dense=False, want_events=True, event_cap=6, **idx)
        raise SystemExit("no overrun reported")
    except L.EmgpuError as e:
        assert e.code == L.ERR_EVENT_CAP
print("wide ok")
''' % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"), os.path.join(ROOT_DIR, "oracle"), rows, FAST_MODELS + (DEP_MODELS[:3] + ["cor_v1", "littoral_cor_v1"] if rows == "lane" else []),
       FAST_MODELS, str(model_dir))
    env = dict(os.environ, EMGPU_DEBUG_EVENT_ROWS=rows)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "wide ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_terminal_propagation_on_the_run_time_shape_instance(terminal_dir):
    """k_terminal_propagate<0,0,0> (row lengths read from the plan: any trajectory-model shape) is never picked for the shipped
    36/7/5-bin shape, which has its own instance; a child process with EMGPU_DEBUG_TERM_GENERIC runs it on that shape against
    the oracle."""
    import subprocess, sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle as O
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import native
from util import assert_f32_of_f64
ctx = native.Context(0)
t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=%r)
n, seed = 500, 0x5EED0005
_, samples = t.sample(n, seed=seed, ctx=ctx)
geo, mo = t._geo_rows(samples)
dl = t._dyn_rows()
oms = []
for m in t._traj:
    pp = O.parse_model_txt(m.parameters_filename)
    oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
ref, ref_rows = O.propagate(oms, mo, geo, seed, dl, tmax_s=120.0)
got, rows = native.propagate_terminal_host(ctx, [m.native for m in t._traj], geo, mo, seed, tmax_s=120.0, dyn_limits=dl)
assert ctx.last_kernel() == "k_terminal_propagate", ctx.last_kernel()
assert np.array_equal(rows, ref_rows)
for L_ in range(4 * n):
    assert_f32_of_f64(got[L_, :rows[L_]], ref[L_, :rows[L_]], "track " + str(L_))
print("generic ok")
''' % (ROOT_DIR, os.path.join(ROOT_DIR, "tests"), os.path.join(ROOT_DIR, "oracle"), str(terminal_dir))
    env = dict(os.environ, EMGPU_DEBUG_TERM_GENERIC="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "generic ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "uncor_1200only_fwme_v1p2", "glider_v1", "cor_v1"])
def test_every_short_length_matches_oracle(name, gpu_ctx, model_dir):
    """T = 1 .. 18 (every position of the 8-second block boundary, the 4-second output blocks and the
    partial last block), n not a multiple of the wave size."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    idx = uncor_indices(pp)
    for T in range(1, 19):
        n = 130 + 7 * T
        ref = O.uncor_sample(om, n, T, 1000 + T, want_events=False)
        got = native.sample_dbn_host(gpu_ctx, nm, n, T, 1000 + T, want_dense=True, want_events=False, **idx)
        assert_uncor_parity(got, ref, T, check_events=False)


def test_plain_c_host_runs_the_abi_end_to_end(model_dir, tmp_path):
    """examples/c_api_demo.c: model file -> trajectories -> tracks through the C ABI only."""
    import subprocess
    from test_host import _build_c_demo
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([exe, em_io.materialize_model("uncor_1200code_v2p1", model_dir), "5000", "120", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "7 initial / 3 dynamic variables; 5000 trajectories x 120 s (kernel k_sample2track<planar>)" in r.stdout
    m = re.search(r"bin changes per trajectory: ([0-9.]+); rejection retries: (\d+); tracks accepted by sample2track: (\d+)", r.stdout)
    assert m and 2.0 < float(m.group(1)) < 8.0 and int(m.group(2)) < 20 and 3000 < int(m.group(3)) <= 5000


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_models_match_oracle(seed, gpu_ctx, tmp_path):
    """Fuzz: generated models (random DAGs, sparse tables with all-zero columns and huge counts, 1-4 dynamic
    variables, fast and dependent branches, categorical variables, zero bins, zero resample rates) through every
    kernel family: event lists (k_dbn_generic), dense REFERENCE_AUTO (k_uncor_fast / k_dbn_step2 / generic) and
    dense PER_STEP."""
    from util import random_model
    rs = np.random.RandomState(7000 + seed)
    # seeds 2, 6, 10, ...: three independent dynamic variables = the shape k_uncor_fast takes
    parms = random_model(rs, nd=(seed % 4) + 1 if seed < 16 else None, dependent=False if seed % 4 == 2 else None)
    path = str(tmp_path / "m.txt")
    em_io.em_write(parms, path)
    nm = native.NativeModel.load_txt(path)
    pp = O.parse_model_txt(path)
    om = O.OracleModel(pp)
    n, T = 700 + 13 * seed, int(rs.choice([5, 33, 64, 97]))
    ref = O.uncor_sample(om, n, T, 50 + seed)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 50 + seed, want_dense=True, want_events=True)
    assert_uncor_parity(got, ref, T)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 50 + seed, want_dense=True, want_events=False)
    assert_uncor_parity(got, ref, T, check_events=False)
    kernels = {got["kernel"].split("<")[0]}
    refp = O.uncor_sample(om, n, T, 50 + seed, per_step=True, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 50 + seed, want_dense=True, want_events=False, transition_mode=L.TRANSITION_PER_STEP)
    assert_uncor_parity(got, refp, T, check_events=False)
    kernels.add(got["kernel"].split("<")[0])
    assert kernels <= {"k_uncor_fast", "k_dbn_step2", "k_dbn_step", "k_dbn_generic"}


@pytest.mark.parametrize("name,kernel", [("uncor_1200code_v2p1", "k_uncor_fast"), ("cor_v1", "k_dbn_step2"), ("glider_v1", "k_dbn_step2")])
def test_hundreds_of_dediscretize_requests_per_wave_block(name, kernel, gpu_ctx, model_dir):
    """Resample rates of 0.75 on every variable: ~6 dediscretize requests per lane and variable-block, i.e. 1 100-1 500 per
    wave-block -- several rounds of the 256-entry request queue and a dozen worker passes per block (the shipped rates give
    20-200 requests, one round)."""
    nm0, pp, path = load_pair(name, model_dir)
    nm = native.NativeModel.load_txt(path)
    rates = np.full(len(pp["resample_rates"]), 0.75)
    nm.set_f64(L.F_RESAMPLE_RATES, 0, rates)
    pp2 = dict(pp); pp2["resample_rates"] = rates
    om = O.OracleModel(pp2)
    idx = uncor_indices(pp)
    n, T, seed = 1500, 50, 77
    ref = O.uncor_sample(om, n, T, seed, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, want_dense=True, want_events=False, **idx)
    assert got["kernel"].startswith(kernel), got["kernel"]
    assert_uncor_parity(got, ref, T, check_events=False)


@pytest.mark.parametrize("name,n,kernel", [("cor_v1", 10_000_000, "k_dbn_step2<16,4,w4,reg>"), ("cor_v2p1_like", 10_000_000, "k_dbn_step2<16,4,w8,reg>"),
                                           ("glider_v1", 4_000_000, "k_dbn_step2<7,3"),
                                           ("uncor_1200only_rotorcraft_v1p2", 6_250_000, "k_uncor_fast<7,4,6,6>")])
def test_large_batches_of_the_other_kernels_spot_checked(name, n, kernel, model_dir):
    """BASELINE configs 3 and 4 at sizes the oracle cannot finish -- config 3 at its full 10 M encounters x 240 s on cor_v1 AND on the
    generator-made "v2p1-like" correlated model (seed 0x5EED0003), config 4 at one rank's 6.25 M -- slices from the
    start, the middle and the ragged end of the batch against the oracle (slots are keyed by the global index, so the oracle can
    produce any slice on its own)."""
    import torch
    nm, pp, _ = load_pair(name, model_dir)
    idx = uncor_indices(pp)
    n += 77                                     # not a multiple of the workgroup
    T, seed, first = 240, 99, 2**33 + 11
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ni, nd, G4 = nm.n_initial, nm.n_dyn, T // 4
    ib = torch.empty((ni, n), dtype=torch.uint8, device=dev); iv = torch.empty((ni, n), dtype=torch.float32, device=dev)
    db = torch.empty((G4, nd, n), dtype=torch.int32, device=dev); dv = torch.empty((G4, nd, n, 4), dtype=torch.float32, device=dev)
    p, _ = native.make_params(n, T, seed, first_index=first, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
    ctx.sync()
    assert ctx.last_kernel().startswith(kernel), ctx.last_kernel()
    om = O.OracleModel(pp)
    m = 1500
    for lo_i in (0, n // 2 + 13, n - m):
        ref = O.uncor_sample(om, m, T, seed, first_index=first + lo_i, want_events=False)
        gb = native.unpack_dyn_bin(db[:, :, lo_i: lo_i + m].contiguous().cpu().numpy().view(np.uint32), T)
        gv = native.unpack_dyn_val(dv[:, :, lo_i: lo_i + m].contiguous().cpu().numpy(), T)
        assert np.array_equal(gb, ref["dense_bin"]) and np.array_equal(gv, ref["dense_val"].astype(np.float32))
        assert np.array_equal(ib[:, lo_i: lo_i + m].cpu().numpy().T.astype(np.int32), ref["init_bin"])
        assert np.array_equal(iv[:, lo_i: lo_i + m].cpu().numpy().T, ref["init_val"].astype(np.float32))


def test_sample2track_kernel_reproduces_the_committed_golden(gpu_ctx):
    g = np.load(os.path.join(GOLD, "sample2track_48x40.npz"))
    xyz, flags, vmm = native.sample2track_host(gpu_ctx, g["alt0"], g["speed0"], g["updates"], *g["ur"], float(g["min_speed"][0]), float(g["max_speed"][0]))
    assert np.array_equal(flags, g["flags"])
    np.testing.assert_allclose(xyz, g["xyz"], rtol=1e-12, atol=1e-7)
    np.testing.assert_allclose(vmm, g["speed_minmax"], rtol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("name,rot,T", [("uncor_1200code_v2p1", False, 60), ("uncor_1200only_rotorcraft_v1p2", True, 45), ("glider_v1", False, 37)])
def test_uncor_track_matches_oracle(name, rot, T, gpu_ctx, model_dir):
    """UncorEncounterModel.track on the GPU (UncorEncounterModel.m:419-471): rounds of sample -> point-mass dynamics ->
    getDynamicLimits rejection, against the oracle's per-trajectory loop on the same Philox keys (attempt j: seed + j).
    Accepted attempt and limits must be identical -- a trajectory may part only at an attempt whose decision margin (oracle em_note:
    altitude / speed / vertical-rate against their limits, the min / max altitude and speed against the cut points that pick the limits)
    is below 1e-9 --; the f64 track agrees to 1e-9 relative (device vs host libm differ in the last bits of asin / atan / tan / sin /
    cos, and 600 steps accumulate them)."""
    nm, pp, _ = load_pair(name, model_dir)
    om = O.OracleModel(pp)
    n, seed = 3000, 0xF1
    got = native.track_uncor_host(gpu_ctx, nm, n, T, seed, first_index=77, is_rotorcraft=rot)
    ref = O.uncor_track(om, n, T, seed, first_index=77, is_rotorcraft=rot)
    assert "k_uncor_track" in got["kernel"]
    assert (ref["attempts"] > 1).sum() > 10, "the case must exercise the retry rounds"
    # every trajectory accepts the same attempt, or parted where a value of that attempt sat within 1e-9 (relative) of the limit or
    # cut point it was tested against (the f64 tracks of the two sides agree to 1e-9: asserted below)
    same = assert_parting_only_on_a_threshold(got["attempts"], ref["attempts"], ref["margins"], 1e-9, "trajectory")
    assert same.sum() >= n - 3, "more threshold coincidences than 3000 trajectories can explain: %d" % (~same).sum()
    assert np.array_equal(got["limits"][same], ref["limits"][same])
    np.testing.assert_allclose(got["tracks"][same], ref["tracks"][same], rtol=1e-9, atol=1e-6)
    # the sampled part is bit-exact: time 0 row = the initial state of the accepted attempt
    assert np.array_equal(got["tracks"][same][:, 0, :5], ref["tracks"][same][:, 0, :5])
    # 1 Hz recording = every 10th row of the 10 Hz result; results do not depend on how the batch is cut
    one_hz = native.track_uncor_host(gpu_ctx, nm, 500, T, seed, first_index=77, is_rotorcraft=rot, record_stride=10)
    assert np.array_equal(one_hz["tracks"], got["tracks"][:500, ::10])
    part = native.track_uncor_host(gpu_ctx, nm, 300, T, seed, first_index=77 + 200, is_rotorcraft=rot)
    assert np.array_equal(part["tracks"], got["tracks"][200:500]) and np.array_equal(part["attempts"], got["attempts"][200:500])
    assert "k_uncor_track<fastbank>" in got["kernel"]   # r_max = 1e6 (:414) can never bind: the algebraically reduced step


@pytest.mark.gpu
def test_uncor_track_reduced_step_equals_the_literal_step(model_dir, tmp_path):
    """k_uncor_track<fastbank> (bank angle = its command, heading and pitch by rotation) against the literal step of the same kernel
    (EMGPU_DEBUG_UTRACK_LITERAL, read once per process: a child runs it): the same accepted attempts and limits, tracks equal to 1e-10
    relative (both are compared with the oracle's literal form in test_uncor_track_matches_oracle)."""
    import pickle
    import subprocess
    code = r'''
import os, sys, pickle
for q in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(os.environ["EMGPU_ROOT"], q))
from em_model_manned_bayes_amd import native
from util import load_pair
nm, pp, _ = load_pair("uncor_1200code_v2p1", os.environ["EMGPU_MODEL_DIR"])
got = native.track_uncor_host(native.Context(0), nm, 4000, 80, 0xF7, first_index=5)
pickle.dump((got["tracks"], got["attempts"], got["limits"], got["kernel"]), open(os.environ["EMGPU_OUT"], "wb"))
'''
    res = {}
    for tag, extra in (("fast", {}), ("literal", {"EMGPU_DEBUG_UTRACK_LITERAL": "1"})):
        out = str(tmp_path / (tag + ".pkl"))
        env = dict(os.environ, EMGPU_ROOT=ROOT, EMGPU_MODEL_DIR=str(model_dir), EMGPU_OUT=out, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[tag] = pickle.load(open(out, "rb"))
    assert "fastbank" in res["fast"][3] and "fastbank" not in res["literal"][3]
    assert np.array_equal(res["fast"][1], res["literal"][1]) and np.array_equal(res["fast"][2], res["literal"][2])
    np.testing.assert_allclose(res["fast"][0], res["literal"][0], rtol=1e-10, atol=1e-7)


@pytest.mark.gpu
def test_uncor_track_pitch_command_clamped_to_vertical_then_reversed(gpu_ctx):
    """A slow rotorcraft whose commanded vertical rate exceeds its speed: asin(clamp(hdot / v)) = +-90 degrees; when the command then changes
    sign it is ~180 degrees away from the pitch, where sin(command - pitch) is as small as next to it.  The reduced step
    (k_uncor_track<fastbank>) must slew at q_max like the literal one (round 3 jumped: ADVICE r3) -- both against the oracle's literal
    em_point_mass_dynamics on the same f32 inputs, 1e-9."""
    T, n = 70, 6
    dyn = np.array([1.7, 250.0, -50.0, 50.0, np.deg2rad(3.0), 1e6])          # v_low v_high dh_min dh_max q_max r_max (:414)
    init = np.zeros((n, 5), dtype=np.float32)
    ctrl = np.zeros((n, T, 3), dtype=np.float32)
    rng = np.random.RandomState(4)
    for i in range(n):
        v_kt = [2.0, 3.0, 5.0, 8.0, 4.0, 60.0][i]
        up = 1500.0 if i % 2 == 0 else -1500.0                                   # ft/min: 25 ft/s against 3.4 .. 13.5 ft/s of speed (and 101 ft/s: never clamped)
        init[i] = [3000.0, v_kt, 0.0, up * 0.1, 1.0]
        ctrl[i, :, 0] = np.where(np.arange(T) < 35, up, -up)                     # climb (descend) for 35 s, then the opposite
        ctrl[i, :, 1] = rng.uniform(-3, 3, T).round(1)
        ctrl[i, :, 2] = 0.0
    S = 10 * T + 1
    got = {}
    for literal in (0, 1):
        tr = np.zeros((n, S, 8))
        L.check(L.lib().emgpu_debug_uncor_dynamics_host(gpu_ctx._h, n, T, 1, literal, dyn.ctypes.data_as(C.c_void_p), init.ctypes.data_as(C.c_void_p),
                                                         ctrl.ctypes.data_as(C.c_void_p), tr.ctypes.data_as(C.c_void_p)))
        got[literal] = tr
        assert ("fastbank" in gpu_ctx.last_kernel()) == (literal == 0)
    for i in range(n):
        v0 = float(init[i, 1]) * 1.68780972222222
        dh0 = float(init[i, 3]) / 60.0
        dpsi0 = float(init[i, 4]) * (np.pi / 180.0)
        ic = [v0, 0, 0, float(init[i, 0]), 0, np.arcsin(dh0 / v0), np.arctan(v0 * dpsi0 / 32.2), 0.0]
        c = np.stack([ctrl[i, :, 0].astype(np.float64) / 60.0, ctrl[i, :, 1].astype(np.float64) * (np.pi / 180.0),
                      ctrl[i, :, 2].astype(np.float64) * 1.68780972222222], axis=1)
        ref, _ = O.point_mass_dynamics(ic, c, dyn)
        if i < 5:
            assert np.abs(ref[:, 6]).max() > 1.5 and ref[:, 6].min() * ref[:, 6].max() < 0       # the case is the case: vertical, then through level
            steps = np.abs(np.diff(ref[:, 6]))
            assert steps.max() <= dyn[4] * 0.1 + 1e-12                                            # ... at q_max, never a jump
        for literal in (0, 1):
            np.testing.assert_allclose(got[literal][i], ref, rtol=1e-9, atol=1e-6, err_msg="trajectory %d, literal=%d" % (i, literal))


def test_start_grid_in_one_launch_terminal(terminal_dir, gpu_ctx):
    """SURVEY.md 8 f4 / InitStartTerminal.m:57-90 + RUN_terminal.m:33-50: the 18-row start grid (airspace class x ownship intent x intruder
    intent preset in turn) drawn in ONE launch, sample i with the presets of row i -- against the oracle run ROW BY ROW with the model's
    `start` set (sample.m:34 -> bn_sample.m:44-50), and against the library's own per-row calls; the per-sample log-weights equal
    emgpu_model_start_log_weight of the row."""
    t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=terminal_dir)
    per, seed, first = 40, 0x5EED0005, 1000
    grid = t.InitStartTerminal(nSamples=18 * per)
    n = len(grid)
    assert n == 18 * per and len({tuple(r[:3]) for r in grid}) == 18
    inits, samples, lw = t.sample(n, seed=seed, first_index=first, ctx=gpu_ctx, start_grid=grid, return_log_weight=True)
    assert gpu_ctx.last_kernel() == "k_bn<16>+start"
    pp = O.parse_model_txt(t.parameters_filename)
    labs = t.labels_initial
    io, ii = labs.index('"own_speed"') + 1, labs.index('"int_speed"') + 1
    d1, d2 = t.dynLimits1, t.dynLimits2
    bs = None if np.all(np.isinf(t.bounds_sample)) else t.bounds_sample
    for k in range(18):
        row = grid[k * per]
        sl = slice(k * per, (k + 1) * per)
        assert np.all(inits[sl, :3] == np.array(row[:3], dtype=float))
        # the oracle, this row only: the same global indices
        om = O.OracleModel(pp, start=[v or 0 for v in row])
        _, ov, _ = O.geom_sample(om, per, seed, first_index=first + k * per, bounds_sample=bs, idx_own_speed=io, idx_int_speed=ii,
                                 lim1=(d1["minVel_ft_s"], d1["maxVel_ft_s"]), lim2=(d2["minVel_ft_s"], d2["maxVel_ft_s"]))
        assert np.array_equal(inits[sl].astype(np.float32), ov.astype(np.float32)), "row %d of the grid" % k
        # the library, this row only (the model's own start, like RUN_terminal.m:36-39)
        t.start = row
        one, _ = t.sample(per, seed=seed, first_index=first + k * per, ctx=gpu_ctx)
        assert gpu_ctx.last_kernel() == "k_bn<16>"
        assert np.array_equal(one, inits[sl])
        assert np.all(lw[sl] == lw[k * per]) and abs(lw[k * per] - t.start_log_weight) < 1e-12 and np.isfinite(lw[k * per]) and lw[k * per] < 0
    t.start = [None] * t.n_initial
    # a row that presets a node without its parent: 'Attempt to preset a dependent variable' (bn_sample.m:47)
    G = np.array(t.G_initial)
    child = next(c for c in range(t.n_initial) if G[:, c].any())
    bad = [[None] * t.n_initial for _ in range(8)]
    bad[5][child] = 1
    with pytest.raises(L.EmgpuError) as ei:
        t.sample(8, seed=1, ctx=gpu_ctx, start_grid=bad)
    assert ei.value.code == L.ERR_PRESET
    bad[5][child] = None
    bad[2][0] = 99                                           # a bin outside 1..r
    with pytest.raises(L.EmgpuError) as ei:
        t.sample(8, seed=1, ctx=gpu_ctx, start_grid=bad)
    assert ei.value.code == L.ERR_PRESET


def test_start_grid_in_one_launch_dbn(gpu_ctx, model_dir):
    """The same for the DBN sampler (UncorEncounterModel.m:204 `start`, RUN_uncor.m:43-48): a grid of presets, one row per trajectory, in
    one call -- dense trace, event lists and log-weights equal to per-row calls with the model's start set, and to the oracle's."""
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    rows = [[1, 4, 2, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0], [3, 0, 0, 0, 0, 0, 0], [2, 2, 0, 0, 0, 0, 0]]   # RUN_uncor.m:43-45 is the first
    per, T, seed, first = 60, 50, 0xA5, 300
    grid = np.repeat(np.array(rows, dtype=np.int32), per, axis=0)
    n = grid.shape[0]
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=True, want_log_weight=True, start=grid, **idx)
    assert got["kernel"].startswith("k_dbn_generic")
    for k, row in enumerate(rows):
        sl = slice(k * per, (k + 1) * per)
        nm.set_start([v or None for v in row])
        one = native.sample_dbn_host(gpu_ctx, nm, per, T, seed, first_index=first + k * per, want_dense=True, want_events=True, **idx)
        lw_row = nm.start_log_weight()
        nm.set_start([None] * 7)
        for f in ("init_bin", "init_val", "dyn_bin", "dyn_val", "attempts", "ev_count"):
            assert np.array_equal(one[f], got[f][sl]), (k, f)
        assert all(np.array_equal(a, b) for a, b in zip(one["events"], got["events"][sl]))
        assert np.all(got["log_weight"][sl] == got["log_weight"][k * per]) and abs(got["log_weight"][k * per] - lw_row) < 1e-12
        ref = O.uncor_sample(O.OracleModel(pp, start=row), per, T, seed, mode=O.RNG_PHILOX, first_index=first + k * per)
        sub = {f: (v[sl] if isinstance(v, (np.ndarray, list)) else v) for f, v in got.items()}
        assert_uncor_parity(sub, ref, T)
    assert got["log_weight"][per] == 0.0 and got["log_weight"][0] < 0 and got["log_weight"][2 * per] < 0 and got["log_weight"][3 * per] < 0


@pytest.mark.gpu
def test_uncor_class_track_and_index_lists(gpu_ctx, model_dir):
    """The class method (timetable columns, initialSeed semantics, the cap) and emgpu_sample_params.indices on its own:
    an arbitrary subset of a batch re-drawn through the index list equals those rows of the batch."""
    nm, pp, path = load_pair("uncor_1200code_v2p1", model_dir)
    mdl = E.UncorEncounterModel(path)
    res, info = mdl.track(50, 30, initialSeed=5, ctx=gpu_ctx, return_info=True)
    assert len(res) == 50 and set(res[0]) == set(E.UncorEncounterModel.TRACK_FIELDS) and res[0]["time_s"].shape == (301,)
    assert res[0]["time_s"][10] == 1.0 and res[0]["north_ft"][0] == 0.0 and np.all(info["attempts"] >= 1)
    ref = O.uncor_track(O.OracleModel(pp), 50, 30, 5)
    assert np.array_equal(info["attempts"], ref["attempts"])
    with pytest.raises(L.EmgpuError) as e:                          # an impossible cap: some trajectory needs a second attempt
        native.track_uncor_host(gpu_ctx, nm, 3000, 60, 0xF1, first_index=77, max_track_attempts=1)
    assert e.value.code == L.ERR_REJECT_CAP
    idx = uncor_indices(pp)
    full = native.sample_dbn_host(gpu_ctx, nm, 400, 33, 9, first_index=1000, want_dense=True, want_events=True, **idx)
    pick = np.array([1399, 1000, 1007, 1250, 1251, 1003], dtype=np.uint64)
    sub = native.sample_dbn_host(gpu_ctx, nm, len(pick), 33, 9, want_dense=True, want_events=True, indices=pick, **idx)
    assert sub["kernel"].startswith("k_uncor_fast_ev")   # an index list runs on the fast kernels too (round 3: the workers read the owner's index from LDS)
    rows = (pick - 1000).astype(int)
    for k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "attempts"):
        assert np.array_equal(sub[k], full[k][rows]), k
    assert all(np.array_equal(sub["events"][q], full["events"][r]) for q, r in enumerate(rows))


@pytest.mark.gpu
def test_start_log_weight_is_the_frequency_of_the_preset_values(gpu_ctx, model_dir):
    """f4: exp(start_log_weight) == how often unconstrained sampling lands on the preset values (G=1, A=4, L=2)."""
    nm, pp, path = load_pair("uncor_1200code_v2p1", model_dir)
    mdl = E.UncorEncounterModel(path)
    st = [None] * 7
    st[0], st[1], st[2] = 1, 4, 2
    mdl.start = st
    p = np.exp(mdl.start_log_weight)
    n = 400_000
    ob, _, _ = native.sample_bn_host(gpu_ctx, nm, n, 123)             # plain bn_sample on the unconstrained model
    hit = np.mean((ob[:, 0] == 1) & (ob[:, 1] == 4) & (ob[:, 2] == 2))
    assert abs(hit - p) < 5 * np.sqrt(p * (1 - p) / n), (hit, p)


@pytest.mark.gpu
@pytest.mark.parametrize("name,both,one", [("uncor_1200code_v2p1", "k_uncor_fast<7,2,4,2>", "k_uncor_fast_idx<7,2,4,2>"),
                                           ("cor_v1", "k_dbn_step2<16,4,w4,reg>[cor]", "k_dbn_step2<16,4,w4,reg>"),
                                           ("glider_v1", "k_dbn_step2<7,3,reg>[chain,w884]", "k_dbn_step2<7,3,reg>"),
                                           ("littoral_cor_v1", "k_dbn_step2<16,4,w4,reg>[frozen]", "k_dbn_step2<16,4>[frozen]")])
def test_one_dense_output_alone_equals_both(name, both, one, model_dir):
    """The benchmark instances store dyn_bin AND dyn_val without null tests (coop_fill_store_msb<BOTH>); a call that asks for only
    one of the two runs on the instance that tests -- same numbers, and the mixed batch falls back to one launch per block."""
    import torch
    nm, pp, _ = load_pair(name, model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 70_000, 101, 0x5EED0B07
    dev = torch.device("cuda", 0)
    ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    G4, nd = (T + 3) // 4, nm.n_dyn
    p, _ = native.make_params(n, T, seed, first_index=123, **idx)

    def run(want_bin, want_val):
        db = torch.zeros((G4, nd, n), dtype=torch.int32, device=dev)
        dv = torch.zeros((G4, nd, n, 4), dtype=torch.float32, device=dev)
        native.sample_dbn_device(ctx, nm, p, dyn_bin=db.data_ptr() if want_bin else 0, dyn_val=dv.data_ptr() if want_val else 0)
        ctx.sync()
        return db, dv, ctx.last_kernel()
    b0, v0, k0 = run(True, True)
    assert k0 == both, k0
    b1, v1, k1 = run(True, False)
    b2, v2, k2 = run(False, True)
    assert k1 == one and k2 == one, (k1, k2)
    assert torch.equal(b1, b0) and torch.equal(v2, v0)
    assert int(v1.abs().max()) == 0 and int(b2.abs().max()) == 0        # the output that was not asked for stays untouched


@pytest.mark.gpu
@pytest.mark.parametrize("dependent", [True, False])
def test_event_lists_of_a_wide_model_with_five_rates(dependent, gpu_ctx, tmp_path):
    """11 initial variables, 3 dynamic, FIVE variables with a resample rate: the model fits the event kernels' own limit (8 - n_dyn = 5
    resample streams) but runs on the 16-variable shape's instances, which are built for four dynamic variables and have four: it must be
    sent elsewhere (step2_eligible), not lose a stream.  Dependent and fast branch."""
    from util import random_model
    for seed in range(3):
        rs = np.random.RandomState(9100 + seed)
        parms = random_model(rs, nd=3, dependent=dependent, ni=11)
        dyn = [i for i, lab in enumerate(parms["labels_transition"][:11]) if lab.endswith('(t)"')]
        rates = np.zeros(11)
        rates[dyn] = [0.11, 0.07, 0.05]
        static = [v for v in range(11) if v not in dyn]
        rates[static[0]], rates[static[3]] = 0.09, 0.13
        parms["resample_rates"] = rates
        path = str(tmp_path / ("wide%d.txt" % seed))
        em_io.em_write(parms, path)
        nm, pp = native.NativeModel.load_txt(path), O.parse_model_txt(path)
        n, T = 900, 61
        ref = O.uncor_sample(O.OracleModel(pp), n, T, 77 + seed)
        got = native.sample_dbn_host(gpu_ctx, nm, n, T, 77 + seed, want_dense=True, want_events=True)
        assert_uncor_parity(got, ref, T)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(12)))
def test_random_wide_models_match_oracle(seed, gpu_ctx, tmp_path):
    """The fuzzer of test_random_models_match_oracle on models of 8 to 14 initial variables (the 9- and 16-variable kernel shapes;
    1 to 4 dynamic variables on instances built for 3 or 4): event lists, dense REFERENCE_AUTO, dense PER_STEP."""
    from util import random_model
    rs = np.random.RandomState(9300 + seed)
    parms = random_model(rs, nd=(seed % 4) + 1, dependent=None if seed < 8 else bool(seed & 1), ni=int(rs.randint(8, 15)))
    path = str(tmp_path / "w.txt")
    em_io.em_write(parms, path)
    nm, pp = native.NativeModel.load_txt(path), O.parse_model_txt(path)
    om = O.OracleModel(pp)
    n, T = 600 + 11 * seed, int(rs.choice([9, 40, 73]))
    ref = O.uncor_sample(om, n, T, 90 + seed)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 90 + seed, want_dense=True, want_events=True)
    assert_uncor_parity(got, ref, T)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 90 + seed, want_dense=True, want_events=False)
    assert_uncor_parity(got, ref, T, check_events=False)
    refp = O.uncor_sample(om, n, T, 90 + seed, per_step=True, want_events=False)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, 90 + seed, want_dense=True, want_events=False, transition_mode=L.TRANSITION_PER_STEP)
    assert_uncor_parity(got, refp, T, check_events=False)
