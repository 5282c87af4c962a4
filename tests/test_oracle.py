"""CPU tests that pin the oracle: RNG-free known answers (derived from the
cited reference lines), the independent numpy restatement (oracle/pyref.py), RNG known-answer
vectors, the committed golden fixtures and chi-square checks against the CPTs."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import oracle as O
import pyref as P
from em_model_manned_bayes_amd import em_io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _arr(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def test_asub2ind_known_answers():
    L = O.lib()
    f = lambda siz, x: L.em_asub2ind(_arr(siz, np.int32).ctypes.data_as(C.c_void_p), _arr(x, np.int32).ctypes.data_as(C.c_void_p), len(siz))
    assert f([4, 4, 4], [2, 3, 1]) == 10          # asub2ind.m:13-14 : 1 + 1*1 + 2*4 + 0*16
    assert f([4, 4, 4], [1, 1, 1]) == 1
    assert f([4, 4, 4], [4, 4, 4]) == 64
    assert f([7], [5]) == 5
    assert P.asub2ind([4, 4, 4], [2, 3, 1]) == 10


def test_select_random_known_answers():
    L = O.lib()
    w = _arr([0, 5, 0, 5], np.float64)
    sel = lambda r: L.em_select_random_r(w.ctypes.data_as(C.c_void_p), 4, C.c_double(r))
    assert sel(0.5) == 2                           # s=[0 5 5 10], sthres=5 -> first s>=5 is index 2
    assert sel(0.25) == 2 and sel(0.500001) == 4 and sel(0.99) == 4
    assert all(sel(r) in (2, 4) for r in np.linspace(1e-9, 1 - 1e-9, 1001))  # bins 1 and 3 unreachable
    z = _arr([0, 0, 0], np.float64)
    assert L.em_select_random_r(z.ctypes.data_as(C.c_void_p), 3, C.c_double(0.7)) == 1  # all-zero column -> bin 1


def test_discretize_and_priors_known_answers():
    L = O.lib()
    th = _arr([30, 60, 90], np.float64)
    d = lambda x: L.em_discretize_bayes(C.c_double(x), th.ctypes.data_as(C.c_void_p), 3)
    assert [d(29.9), d(30), d(89.9), d(90), d(1e9)] == [1, 2, 3, 4, 4]   # discretize_bayes.m:17-21
    a = np.zeros(18)
    L.em_transition_prior_node(3, C.c_int64(6), C.c_double(1.0), a.ctypes.data_as(C.c_void_p))
    A = a.reshape(6, 3).T                                                   # column-major 3 x 6
    expect = np.zeros((3, 6)); expect[0, 0:2] = 1; expect[1, 2:4] = 1; expect[2, 4:6] = 1
    assert np.array_equal(A, expect)                                        # setTransitionPriors.m:20-27
    a = np.zeros(5 * 7840)
    L.em_dirichlet_prior_node(5, C.c_int64(7840), 1, C.c_double(0), a.ctypes.data_as(C.c_void_p))
    assert np.all(a == 1.0 / 39200)                                         # bn_dirichlet_prior.m:22-25


def test_events_formatting_known_answers():
    ev = [[2, 2, 3], [0, 1, 2], [3, 0, 0]]
    assert np.array_equal(O.events2samples([1, 2], ev), [[1, 1, 2, 2, 2], [2, 2, 3, 3, 3]])  # events2samples.m:15-26
    assert np.array_equal(P.events2samples(np.array([1., 2.]), np.array(ev, dtype=float)), [[1, 1, 2, 2, 2], [2, 2, 3, 3, 3]])
    assert np.array_equal(P.events2controls(np.array([1., 2.]), np.array(ev, dtype=float), np.array([[2, 3]])), [[0, 2], [2, 3]])


def test_dediscretize_known_answers():
    L = O.lib()
    p = _arr([-2, -1, -0.25, 0.25, 1, 2], np.float64)
    used = C.c_int(0)
    f = lambda d, zb, u: L.em_dediscretize_u(d, p.ctypes.data_as(C.c_void_p), 6, zb, C.c_double(u), C.byref(used))
    assert f(3, 3, 0.9) == 0.0 and used.value == 0          # zero bin: 0, no draw (dediscretize.m:24-25)
    assert f(1, 3, 0.5) == -1.5 and used.value == 1         # a + (b-a)*rand
    assert f(5, 3, 0.0) == 1.0
    assert L.em_dediscretize_u(4, None, 0, 0, C.c_double(0.3), C.byref(used)) == 4.0 and used.value == 0  # empty params: the bin


def test_extract_zero_bins_and_temporal_map(model_dir):
    assert list(O._extract_zero_bins([np.array([-2, -1, -.25, .25, 1, 2])])) == [3]      # em_read.m:143-156
    assert list(O._extract_zero_bins([np.array([0, 30, 60])])) == [0]
    pp = O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir))
    assert pp["temporal_map"].tolist() == [[5, 8], [6, 9], [7, 10]]                       # UncorEncounterModel.m:55
    assert pp["zero_bins"].tolist() == [0, 0, 0, 0, 3, 4, 4]                              # UncorEncounterModel.m:84
    assert pp["r_initial"].tolist() == [4, 4, 4, 8, 5, 7, 7]
    assert pp["N_initial"][0][:, 0].tolist() == [459966244, 5365919, 15357480, 6158859]   # model file line 14


def test_rng_known_answer_vectors():
    # Random123 kat_vectors, philox4x32-10
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # Random123 kat_vectors, philox4x32-7: the round count the generator runs (EM_PHILOX_ROUNDS == EMGPU_PHILOX_ROUNDS)
    assert O.philox_rounds() == 7
    assert O.philox4x32_r([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert O.philox4x32_r([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert O.philox4x32_r([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 7) == \
        [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    # MATLAB rng(1,'twister'); rand(1,4) == numpy RandomState(1): 0.417022 0.720324 0.000114 0.302333
    for seed in (1, 5489, 2**31 + 7):
        assert np.array_equal(O.mt_doubles(seed, 50), np.random.RandomState(seed).random_sample(50))
    assert abs(O.mt_doubles(1, 1)[0] - 0.417022004702574) < 1e-15
    u = O.lib().em_uniform32
    assert 0 < u(0) < u(1) < u(0xFFFFFFFE) == u(0xFFFFFFFF) < 1


@pytest.mark.parametrize("name,n,T,seed", [("uncor_1200code_v2p1", 40, 120, 1), ("uncor_1200only_fwse_v1p2", 10, 210, 1),
                                          ("uncor_1200code_v1", 12, 60, 7), ("glider_v1", 8, 50, 3),
                                          ("dueregard_v1", 8, 60, 5), ("littoral_uncor_v1", 8, 40, 9), ("haa_v1", 4, 30, 2)])
def test_c_oracle_matches_numpy_restatement_draw_for_draw(name, n, T, seed, model_dir):
    pp = O.parse_model_txt(em_io.materialize_model(name, model_dir))
    om = O.OracleModel(pp)
    ref, cnt = P.uncor_sample(pp, n, T, seed)
    r = O.uncor_sample(om, n, T, seed, mode=O.RNG_MT19937)
    assert cnt == r["n_draws"]
    for i in range(n):
        ini, ev, smp, ctl = ref[i]
        assert np.array_equal(ini, r["init_val"][i])
        assert np.array_equal(ev, r["events"][i][:, :3])
        assert np.array_equal(O.events2samples(r["init_val"][i], r["events"][i][:, :3]), smp)
        assert np.array_equal(O.events2controls(om, r["init_val"][i], r["events"][i][:, :3]), ctl)


def _unpack_events(cnt, flat):
    out, pos = [], 0
    for c in cnt:
        out.append(flat[pos: pos + c]); pos += c
    return out


def test_golden_config1_mt19937(model_dir):
    """BASELINE.json configs[0]: uncor_1200code_v2p1, sample(100, 120, 'seed', 1)."""
    g = np.load(os.path.join(GOLD, "config1_uncor_v2p1_mt19937_seed1_100x120.npz"))
    om = O.OracleModel(O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir)))
    r = O.uncor_sample(om, 100, 120, 1, mode=O.RNG_MT19937)
    assert np.array_equal(r["init_bin"], g["init_bin"]) and np.array_equal(r["init_val"], g["init_val"])
    assert int(g["n_draws"][0]) == r["n_draws"] == 121592
    for a, b in zip(r["events"], _unpack_events(g["ev_count"], g["ev_flat"])):
        assert np.array_equal(a, b)
    assert np.array_equal(r["dense_bin"], g["dense_bin"])
    assert np.allclose(r["init_val"][0], [1, 4, 1194.36151, 71.4070949, 0, -253.649563, -2.94358725], rtol=1e-8)


def _philox_stats(rounds, n=1 << 20):
    """The generator exactly as the slot map drives it -- sequential global indices in counter word 0, small section / block numbers in
    word 3, one key -- reduced to the numbers the tests below judge: worst avalanche deviation (in sigmas) over all 192 input bits,
    chi-square z-score of the 16-bit halfwords the compares read, and the lag-1 correlation between neighbouring trajectories."""
    rs = np.random.RandomState(12345)
    # avalanche: flip one input bit, every output bit should flip with probability 1/2
    m = 4096
    base = rs.randint(0, 2**32, size=(6, m), dtype=np.uint64).astype(np.uint32)
    ref = np.stack(O.philox4x32_np(base[0], base[1], base[2], base[3], 0x5EED0002, 0x0, rounds))
    worst = 0.0
    for word in range(4):
        for bit in range(32):
            c = [base[q].copy() for q in range(4)]
            c[word] ^= np.uint32(1 << bit)
            out = np.stack(O.philox4x32_np(c[0], c[1], c[2], c[3], 0x5EED0002, 0x0, rounds))
            d = ref ^ out
            flips = np.array([[((d[w] >> np.uint32(b)) & 1).sum() for b in range(32)] for w in range(4)], dtype=np.float64)
            worst = max(worst, float(np.abs(flips - m / 2).max() / np.sqrt(m / 4)))
    # the slot map's own counters: gidx = 0..n-1, attempt 0, word 3 = section 3 (TRANS), variable 5, block 0..3
    gidx = np.arange(n, dtype=np.uint32)
    zs, r1 = [], []
    for blk in range(4):
        w3 = np.uint32((3 << 28) | (5 << 20) | blk)
        out = O.philox4x32_np(gidx, 0, 0, w3, 0x5EED0002, 0x0, rounds)
        for w in out[:2]:
            for h in (w >> np.uint32(16), w & np.uint32(0xFFFF)):
                cnt = np.bincount(h.astype(np.int64), minlength=65536).astype(np.float64)
                e = n / 65536.0
                chi = ((cnt - e) ** 2 / e).sum()
                zs.append((chi - 65535.0) / np.sqrt(2 * 65535.0))
            u = w.astype(np.float64) / 2**32 - 0.5
            r1.append(float((u[1:] * u[:-1]).mean() / u.var() * np.sqrt(n)))   # in sigmas
    return worst, float(np.abs(zs).max()), float(np.abs(r1).max())


def test_philox_round_count_statistics():
    """HISTORY.md section 5 (the RNG-cost question): the generator runs Philox4x32-7.  On the counters the slot map really uses, 7 rounds
    are statistically indistinguishable from Random123's default of 10 -- full avalanche over every counter bit, uniform
    halfwords, no correlation between neighbouring trajectories -- and the same three tests DO see a generator that is too weak
    (3 rounds), so they have the power to notice.  (Crush-resistance of 7 rounds with sequential counters: Salmon et al., SC'11.)"""
    a7, z7, r7 = _philox_stats(7)
    a10, z10, r10 = _philox_stats(10, n=1 << 18)
    assert a7 < 6.0 and z7 < 5.0 and r7 < 5.0, (a7, z7, r7)
    assert a10 < 6.0 and z10 < 5.0 and r10 < 5.0, (a10, z10, r10)
    a3, z3, r3 = _philox_stats(3, n=1 << 18)
    assert a3 > 20.0 or z3 > 20.0, (a3, z3, r3)
    # the numpy restatement used above is the C generator: same words for random inputs at the round count in use
    rs = np.random.RandomState(7)
    for _ in range(50):
        c = [int(x) for x in rs.randint(0, 2**32, size=4, dtype=np.uint64)]
        k = [int(x) for x in rs.randint(0, 2**32, size=2, dtype=np.uint64)]
        got = [int(x[0]) for x in O.philox4x32_np([c[0]], [c[1]], [c[2]], [c[3]], k[0], k[1], O.philox_rounds())]
        assert got == O.philox4x32_r(c, k, O.philox_rounds())


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*_philox_*.npz"))))
def test_golden_philox_slot_map(path, model_dir):
    g = np.load(path)
    n, T, seed, first, per_step = [int(x) for x in g["meta"]]
    name = os.path.basename(path).split("_philox_")[0]
    om = O.OracleModel(O.parse_model_txt(em_io.materialize_model(name, model_dir)))
    r = O.uncor_sample(om, n, T, seed, mode=O.RNG_PHILOX, first_index=first, per_step=bool(per_step))
    assert np.array_equal(r["init_bin"], g["init_bin"]) and np.array_equal(r["init_val"], g["init_val"])
    assert np.array_equal(r["dense_bin"], g["dense_bin"]) and np.array_equal(r["dense_val"], g["dense_val"])
    for a, b in zip(r["events"], _unpack_events(g["ev_count"], g["ev_flat"])):
        assert np.array_equal(a, b)


def test_philox_results_do_not_depend_on_batching(model_dir):
    """Keyed by global index: any split of [0, n) gives the same trajectories (multi-GPU sharding)."""
    om = O.OracleModel(O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir)))
    full = O.uncor_sample(om, 60, 80, 42, first_index=1000)
    a = O.uncor_sample(om, 25, 80, 42, first_index=1000)
    b = O.uncor_sample(om, 35, 80, 42, first_index=1025)
    assert np.array_equal(np.concatenate([a["dense_bin"], b["dense_bin"]]), full["dense_bin"])
    assert np.array_equal(np.concatenate([a["dense_val"], b["dense_val"]]), full["dense_val"])


def test_dense_trace_is_events2samples_of_the_event_list(model_dir):
    pp = O.parse_model_txt(em_io.materialize_model("uncor_1200only_fwse_v1p2", model_dir))
    om = O.OracleModel(pp)
    r = O.uncor_sample(om, 20, 100, 3)
    dyn = pp["temporal_map"][:, 0] - 1
    for i in range(20):
        s = O.events2samples(r["init_val"][i], r["events"][i][:, :3])
        assert s.shape == (7, 100)
        assert np.array_equal(s[dyn].T, r["dense_val"][i])


@pytest.mark.parametrize("mode", [O.RNG_MT19937, O.RNG_PHILOX])
def test_chi_square_initial_and_transition_frequencies(mode, model_dir):
    """Empirical bin frequencies against the normalised CPT columns."""
    from scipy import stats
    pp = O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir))
    om = O.OracleModel(pp)
    n = 20000
    init, ev = O.dbn_sample(om, n, 2, 12345, mode=mode)
    # root node G: marginal
    w = pp["N_initial"][0][:, 0]
    obs = np.bincount(init[:, 0], minlength=5)[1:]
    assert stats.chisquare(obs, w / w.sum() * n).pvalue > 1e-4
    # node A | G=1
    sel = init[:, 0] == 1
    w = pp["N_initial"][1][:, 0]
    obs = np.bincount(init[sel, 1], minlength=5)[1:]
    assert stats.chisquare(obs[w > 0], w[w > 0] / w.sum() * sel.sum()).pvalue > 1e-4
    assert obs[w == 0].sum() == 0
    # one transition step of \dot psi given the most common initial configuration
    key = [tuple(r) for r in init]
    common = max(set(key), key=key.count)
    sel = np.array([k == common for k in key])
    G = pp["G_transition"]; r = pp["r_transition"]
    par = np.nonzero(G[:, 9])[0]
    j = P.asub2ind(r[par], np.array(common)[par])
    w = pp["N_transition"][9][:, j - 1]
    new = np.array([common[6]] * n)
    for i, e in enumerate(ev):
        for row in e:
            if row[1] == 7:
                new[i] = row[2]
    obs = np.bincount(new[sel], minlength=8)[1:]
    assert obs[w == 0].sum() == 0
    if (w > 0).sum() > 1 and sel.sum() > 200:
        exp = w[w > 0] / w.sum() * sel.sum()
        keep = exp >= 5
        if keep.sum() > 1:
            o, e_ = obs[w > 0][keep], exp[keep]
            assert stats.chisquare(o, e_ * o.sum() / e_.sum()).pvalue > 1e-4


def test_propagate_trajectory_restatement_properties(tmp_path):
    """a15 (createEncounter.m:93-329) on synthetic trajectory models: termination rules, rate limits,
    and the MT19937 / Philox modes agree on everything that does not depend on the draws."""
    from em_model_manned_bayes_amd import synthetic
    d = synthetic.write_terminal_directory(str(tmp_path))
    import glob
    f_fwd = glob.glob(os.path.join(d, "*_ownship_landing_model.txt"))[0]
    f_bck = glob.glob(os.path.join(d, "*_ownship_landing_model_reverse.txt"))[0]
    oms = []
    for f in (f_fwd, f_bck):
        pp = O.parse_model_txt(f)
        assert pp["temporal_map"].tolist() == [[4, 7], [5, 8], [6, 9]]
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    n = 40
    rs = np.random.RandomState(3)
    geo = np.zeros((n, 12))
    for a in range(2):
        dist, bear = rs.uniform(1, 6, n), rs.uniform(0, 360, n)
        geo[:, 6 * a + 0] = dist * np.cos(np.deg2rad(bear)); geo[:, 6 * a + 1] = dist * np.sin(np.deg2rad(bear))
        geo[:, 6 * a + 2] = rs.uniform(300, 3000, n); geo[:, 6 * a + 3] = rs.uniform(100, 400, n)
        geo[:, 6 * a + 4] = rs.uniform(0, 360, n); geo[:, 6 * a + 5] = 1
    mo = np.tile([0, 1, 0, 1], (n, 1))
    dl = np.array([[50, 506, 12, 5000, 100.0], [169, 491, 1.5, 5000, 2500 / 60]])
    for mode in (O.RNG_PHILOX, O.RNG_MT19937):
        out, rows = O.propagate(oms, mo, geo, 11, dl, mode=mode, tmax_s=120.0)
        assert rows.min() >= 1 and rows.max() <= 122
        for L_ in range(4 * n):
            tr = out[L_, : rows[L_]]
            sign = -1.0 if (L_ & 1) else 1.0
            assert np.array_equal(tr[:, 0], sign * np.arange(rows[L_]))             # t_s advances by dt_s (:259)
            ac = (L_ >> 1) & 1
            assert np.all(np.abs(np.diff(tr[:, 3])) <= dl[ac, 4] + 1e-9)             # altitude is rate limited (:180-184)
            assert np.all((tr[:, 4] >= 0) & (tr[:, 4] <= 360))                       # wrapTo360 (:176)
            e = L_ >> 2
            assert abs(tr[0, 1] - geo[e, 6 * ac]) < 1e-12 and abs(tr[0, 5] - geo[e, 6 * ac + 3]) < 1e-9
            # the loop stops at |t| > tmax, beyond the distance bound, within 0.25 nm of the runway (landing/take-off)
            # or, for the ownship, right of the runway (:296-329): every recorded row but the last satisfies "continue"
            if rows[L_] < 122:
                assert rows[L_] >= 1
    # Philox results are keyed by the global index: a shifted first_index shifts the trajectories
    a, ra = O.propagate(oms, mo[:10], geo[:10], 5, dl, first_index=100)
    b, rb = O.propagate(oms, mo[5:10], geo[5:10], 5, dl, first_index=105)
    assert np.array_equal(ra[20:], rb) and np.array_equal(a[20:], b)
    # bench.py's all-cores CPU leg does the same work (thread-private buffers): the same number of track rows, whatever the thread count
    _, rows = O.propagate(oms, mo, geo, 11, dl, first_index=7)
    for threads in (1, 3, 8):
        assert O.propagate_throughput_mt(oms, mo, geo, 11, dl, threads, first_index=7) == int(rows.sum())


def test_sample2track_known_answers():
    """sample2track.m:183-243 restated: hand-checkable tracks."""
    ur_s, ur_v = (1852.0 / 0.3048) / 3600.0, 1.0 / 60.0
    T = 8
    upd = np.zeros((4, T, 3))
    upd[1, :, 2] = 90.0 / 4            # quarter turn in 4 s: heading 0, 22.5, 45, 67.5, 90 (exact cosd(90) = 0), ...
    upd[2, :, 0] = -600.0              # 10 ft/s descent from 35 ft: below ground after 4 s
    upd[3, 3, 1] = 1000.0              # acceleration spike: leaves the speed range
    alt0 = np.array([1000.0, 1000.0, 35.0, 1000.0])
    v0 = np.array([100.0, 100.0, 100.0, 100.0])
    xyz, flags, vmm = O.sample2track(alt0, v0, upd, ur_s, ur_v, 1.0, 30.0, 300.0)
    s = 100.0 * ur_s
    assert np.allclose(xyz[0, :, 0], s * np.arange(T + 1), rtol=1e-15) and np.all(xyz[0, :, 1] == 0) and np.all(xyz[0, :, 2] == 1000.0)
    hd = 22.5 * np.arange(T)           # heading used for the step t -> t+1 is the one BEFORE the update (:211-212)
    assert np.allclose(xyz[1, 1:, 0], np.cumsum(s * np.cos(np.deg2rad(hd))), rtol=1e-12, atol=1e-9)
    assert np.allclose(xyz[1, 1:, 1], np.cumsum(s * np.sin(np.deg2rad(hd))), rtol=1e-12, atol=1e-9)
    assert xyz[1, 5, 0] == xyz[1, 4, 0]                      # heading exactly 90 degrees: cosd(90) is exactly 0
    assert xyz[1, 5, 1] == xyz[1, 4, 1] + s                  # and sind(90) exactly 1
    assert np.allclose(xyz[2, :, 2], 35.0 - 10.0 * np.arange(T + 1))
    assert flags.tolist() == [0, 0, 1, 2]
    assert np.allclose(vmm[3], [s, s + 1000.0 * ur_s]) and np.allclose(vmm[0], [s, s])
    # the bounds are exclusive (<=, >=: sample2track.m:240)
    _, f2, _ = O.sample2track([10.0], [30.0], np.zeros((1, 2, 3)), ur_s, ur_v, 1.0, 30.0, 300.0)
    assert f2[0] == 2


def test_golden_sample2track():
    g = np.load(os.path.join(GOLD, "sample2track_48x40.npz"))
    xyz, flags, vmm = O.sample2track(g["alt0"], g["speed0"], g["updates"], *g["ur"], float(g["min_speed"][0]), float(g["max_speed"][0]))
    assert np.array_equal(xyz, g["xyz"]) and np.array_equal(flags, g["flags"]) and np.array_equal(vmm, g["speed_minmax"])
    assert flags[0] == 0 and np.all(np.diff(g["xyz"][0, :, 0]) > 0) and np.all(g["xyz"][0, :, 1] == 0)   # straight and level
    assert flags[2] & 1                                                                                # dives into the ground


def test_local_smooth_stand_in_known_answers():
    """The stand-in for em-core's local_smooth (UNPINNED): centred moving average, the window shrinking symmetrically at the ends."""
    x = np.array([1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0])
    assert O.local_smooth(x, 5).tolist() == [1.0, 7.0 / 3.0, 31.0 / 5.0, 62.0 / 5.0, 124.0 / 5.0, 112.0 / 3.0, 64.0]
    assert O.local_smooth(x, 15).tolist() == [1.0, 7.0 / 3.0, 31.0 / 5.0, 127.0 / 7.0, 124.0 / 5.0, 112.0 / 3.0, 64.0]
    assert O.local_smooth(x[:1], 5).tolist() == [1.0] and O.local_smooth(x, 1).tolist() == x.tolist()


_TERM_ORDER = ["ownship_landing_model", "ownship_landing_model_reverse", "ownship_takeoff_model", "ownship_takeoff_model_reverse",
               "intruder_landing_model", "intruder_landing_model_reverse", "intruder_takeoff_model", "intruder_landing_model_reverse",
               "intruder_transit_model", "intruder_landing_model_reverse"]        # CorTerminalModel.m:84-100 (its copy-paste of the reverse files kept)
_DL = {"GENERIC": dict(minVel_ft_s=50, maxVel_ft_s=506, maxTurnRate_deg_s=12, maxAltitude_ft=5000, maxVertRate_ft_s=6000 / 60, maxCumTurn_deg=np.inf, pitch_deg=np.inf),
       "RTCA228_A2": dict(minVel_ft_s=68, maxVel_ft_s=338, maxTurnRate_deg_s=3, maxAltitude_ft=5000, maxVertRate_ft_s=1500 / 60, maxCumTurn_deg=180, pitch_deg=15),
       "TEST": dict(minVel_ft_s=68, maxVel_ft_s=186, maxTurnRate_deg_s=7, maxAltitude_ft=1200, maxVertRate_ft_s=500 / 60, maxCumTurn_deg=180, pitch_deg=15)}


def _dl_rows(a, b):
    k = ("minVel_ft_s", "maxVel_ft_s", "maxTurnRate_deg_s", "maxAltitude_ft", "maxVertRate_ft_s")
    return np.array([[_DL[a][q] for q in k], [_DL[b][q] for q in k]], dtype=float)


@pytest.mark.parametrize("src", ["terminal_v3_radar_encounter_model", "terminal_v3_opensky_encounter_model"])
def test_geometry_restatements_agree_draw_for_draw(src, model_dir):
    """@CorTerminalModel/sample.m:29-77 twice: oracle/em_oracle.c (em_geom_sample_batch, MT19937 mode) and oracle/pyref.py, with the box
    and speed rejection exercised and with presets (InitStartTerminal's first three variables)."""
    pp = O.parse_model_txt(em_io.materialize_model(src, model_dir))
    labs = [x.strip('"') for x in pp["labels_initial"]]
    io, ii = labs.index("own_speed") + 1, labs.index("int_speed") + 1
    bs = np.tile([-np.inf, np.inf], (pp["n_initial"], 1))
    bs[labs.index("own_distance")] = [0.0, 4.0]                      # a box that rejects a good part of the draws
    for start, lim1, lim2, seed in (([0] * 15, (50, 506), (50, 506), 1), ([2, 1, 3] + [0] * 12, (169, 491), (68, 338), 7)):
        om = O.OracleModel(pp, start=start)
        n = 30
        ob, ov, att = O.geom_sample(om, n, seed, mode=O.RNG_MT19937, bounds_sample=bs, idx_own_speed=io, idx_int_speed=ii, lim1=lim1, lim2=lim2)
        ref, R = P.geom_sample(pp, n, seed, bounds_sample=bs, lim1=lim1, lim2=lim2, start=[v or None for v in start])
        assert np.array_equal(ov, ref)
        assert att.sum() > n                                         # the rejection loop ran
        if start[0]:
            assert np.all(ov[:, :3] == np.array(start[:3], dtype=float))


def test_terminal_restatements_agree_draw_for_draw(tmp_path, model_dir):
    """createEncounter.m:52-329 (PropagateTrajectory, CreateStartDistribution, CheckTrajectoryConditions), the static checks of
    CorTerminalModel.m:117-316 and the filters of track.m:79-145, twice: the C oracle (MT19937 mode: one stream through the encounters,
    aircraft 1 forward, backward, aircraft 2 forward, backward) and pyref.py -- every recorded row equal to the last bit, the same filter
    decisions and CPA metadata."""
    import glob
    from em_model_manned_bayes_amd import synthetic
    d = synthetic.write_terminal_directory(str(tmp_path))
    pps = [O.parse_model_txt(glob.glob(os.path.join(d, "*_" + s + ".txt"))[0]) for s in _TERM_ORDER]
    oms = [O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)) for pp in pps]
    gp = O.parse_model_txt(em_io.materialize_model("terminal_v3_radar_encounter_model", model_dir))
    labs = [x.strip('"') for x in gp["labels_initial"]]
    decided = {True: 0, False: 0}
    for (a1, a2), seed, n in ((("GENERIC", "GENERIC"), 3, 14), (("RTCA228_A2", "TEST"), 11, 6)):
        dl = _dl_rows(a1, a2)
        geom, _ = P.geom_sample(gp, n, seed + 100, lim1=tuple(dl[0, :2]), lim2=tuple(dl[1, :2]))
        samples = [dict(zip(labs, row)) for row in geom]
        geo = np.zeros((n, 12)); mo = np.zeros((n, 4), dtype=np.int32)
        for e, sg in enumerate(samples):
            for a, pre in enumerate(("own", "int")):
                sn, cs = P.sincosd(sg[pre + "_bearing"])
                geo[e, 6 * a: 6 * a + 6] = [sg[pre + "_distance"] * cs, sg[pre + "_distance"] * sn, sg[pre + "_alt"], sg[pre + "_speed"], sg[pre + "_heading"], sg[pre + "_intent"]]
            oi, ii_ = int(sg["own_intent"]), int(sg["int_intent"])
            mo[e] = [2 * (oi - 1), 2 * (oi - 1) + 1, 4 + 2 * (ii_ - 1), 4 + 2 * (ii_ - 1) + 1]
        out, rows = O.propagate(oms, mo, geo, seed, dl, mode=O.RNG_MT19937, tmax_s=120.0)
        R = P.Rand(seed)
        dlp = [_DL[a1], _DL[a2]]
        for e, sg in enumerate(samples):
            traj = P.create_encounter(pps, mo[e], sg, 120.0, dlp, R)
            rows_c = []
            for a in range(2):
                fwd = out[4 * e + 2 * a, : rows[4 * e + 2 * a]]
                bck = out[4 * e + 2 * a + 1, 1: rows[4 * e + 2 * a + 1]]
                both = np.concatenate([fwd, bck], axis=0)
                both = both[np.argsort(both[:, 0], kind="stable")]
                mine = np.stack([traj[a][k] for k in ("t_s", "x_nm", "y_nm", "z_ft", "heading_deg", "v_ft_s")], axis=1)
                assert np.array_equal(both, mine), "encounter %d aircraft %d" % (e, a)
                rows_c.append(both)
            good_c, meta_c = O.terminal_filters(rows_c[0], rows_c[1], sg["own_intent"], sg["int_intent"], dl, [dlp[0]["maxCumTurn_deg"], dlp[1]["maxCumTurn_deg"]],
                                                [dlp[0]["pitch_deg"], dlp[1]["pitch_deg"]])
            good_p, meta_p = P.terminal_filters(traj, int(sg["own_intent"]), int(sg["int_intent"]), dlp, [dlp[0]["maxCumTurn_deg"], dlp[1]["maxCumTurn_deg"]],
                                                [dlp[0]["pitch_deg"], dlp[1]["pitch_deg"]])
            assert good_c == good_p and np.array_equal(meta_c, np.array(meta_p)), (e, good_c, good_p, meta_c, meta_p)
            decided[good_c] += 1
            for a in range(2):                                      # CheckCumTurn on its own, with a limit that decides
                for lim in (20.0, 60.0, 180.0):
                    assert O.check_cum_turn(traj[a]["heading_deg"], lim) == P.check_cum_turn(traj[a]["heading_deg"], lim)
            sm = P.create_encounter(pps, mo[e], sg, 120.0, dlp, P.Rand(seed + 999), smooth=True) if e == 0 else None
            if sm is not None:
                raw = P.create_encounter(pps, mo[e], sg, 120.0, dlp, P.Rand(seed + 999))
                assert np.array_equal(sm[0]["v_ft_s"], O.local_smooth(raw[0]["v_ft_s"], 5)) and np.array_equal(sm[1]["z_ft"], O.local_smooth(raw[1]["z_ft"], 15))
    assert decided[False] > 0                                        # (the synthetic tables pass the filters about once in 150 attempts)


def test_sample2track_restatements_agree():
    """sample2track.m:183-243 twice (em_sample2track_batch and pyref.sample2track): positions to the last bit, the same CFIT / speed flags."""
    rs = np.random.RandomState(8)
    n, T = 25, 40
    alt0 = rs.uniform(50, 3000, n); speed0 = rs.uniform(30, 200, n)
    alt0[:8] = rs.uniform(5, 120, 8)                                 # low starts: some tracks reach the ground (CFIT, :234-237)
    upd = np.stack([rs.uniform(-1500, 1500, (n, T)), rs.uniform(-2, 2, (n, T)), rs.uniform(-6, 6, (n, T))], axis=2)
    ur_speed, ur_vr, ur_h = 6076.1154855643 / 3600.0, 1.0 / 60.0, 1.0
    xyz, flags, vmm = O.sample2track(alt0, speed0, upd, ur_speed, ur_vr, ur_h, 40.0, 180.0)
    seen = set()
    for i in range(n):
        x, y, z, cfit, rej = P.sample2track(alt0[i], speed0[i], upd[i], ur_speed, ur_vr, ur_h, 40.0, 180.0)
        assert np.array_equal(xyz[i, :, 0], x) and np.array_equal(xyz[i, :, 1], y) and np.array_equal(xyz[i, :, 2], z)
        assert bool(flags[i] & 1) == cfit and bool(flags[i] & 2) == rej
        seen.add((cfit, rej))
    assert len(seen) >= 3



def test_golden_terminal_propagation_slot_map(tmp_path):
    """The committed PropagateTrajectory golden (Philox mode, createEncounter.m:93-265 on the synthetic trajectory tables): pins the TERM_TRANS /
    TERM_DEDISC slots across rounds -- since round 5 an attempt's first dediscretize draw is word 3 of its TERM_TRANS block."""
    import glob as _g
    from em_model_manned_bayes_amd import synthetic
    g = np.load(os.path.join(GOLD, "terminal_propagate_phx_seed5eed0005_first7_32.npz"))
    n, seed, first, cap = [int(x) for x in g["meta"]]
    d = synthetic.write_terminal_directory(str(tmp_path))
    oms = []
    for stem in synthetic.TERMINAL_FILE_STEMS:
        pp = O.parse_model_txt(_g.glob(os.path.join(d, "*_" + stem + ".txt"))[0])
        oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
    out, rows = O.propagate(oms, g["model_of"], g["geo"], seed, g["dyn_limits"], first_index=first, tmax_s=120.0, cap=cap)
    assert np.array_equal(rows, g["rows"])
    keep = np.arange(cap)[None, :] < rows[:, None]
    assert np.array_equal(np.where(keep[:, :, None], out, 0.0), g["tracks"])
    assert rows.min() >= 1 and rows.max() <= 122 and len(set(rows.tolist())) > 10
