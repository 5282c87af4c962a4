"""CPU tests of the product's host side: the C ABI exports, the .txt loader (against the oracle's
independent parser and the data invariants of SURVEY.md section 8c), em_write round trips, priors,
presets, error behaviour, the plan compiler's integer thresholds, and the sharding rule."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import oracle as O
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import _lib as L, em_io, native, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL_MODELS = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, "models")) if f.endswith(".npz"))


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "emgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(emgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.lib()
    for s in declared:
        assert hasattr(lib, s)
    assert lib.emgpu_version().decode().startswith("emgpu")


def test_version_names_the_generator_and_the_sources():
    v = L.lib().emgpu_version().decode()
    r = int(L.lib().emgpu_philox_rounds())
    assert r in (7, 10) and ("philox4x32-%d" % r) in v and "src:" in v
    assert r == O.philox_rounds()        # (tests/conftest.py refuses to start a session otherwise)


def _source_hash():
    """The hash csrc/Makefile bakes into emgpu_version(): sha256 over the sorted sources + the public header, first 12 hex digits."""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "em_model_manned_bayes_amd", "csrc")
    names = sorted({os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h", "*.hpp") for f in glob.glob(os.path.join(csrc, pat))})
    h = hashlib.sha256()
    h.update(open(os.path.join(ROOT, "include", "emgpu.h"), "rb").read())       # "../../include/emgpu.h" sorts first
    for nme in names:
        h.update(open(os.path.join(csrc, nme), "rb").read())
    return h.hexdigest()[:12]


def test_the_library_and_the_rounds_profiles_are_of_the_sources_in_the_tree():
    """bench.py reports roofline.traffic only from a profile summary recorded with the SAME sources as the loaded library (the hash in
    emgpu_version()).  A comment edited in a header after the profiles were taken silently turns every `traffic` into null at the next build:
    the built library must be current, and the newest round's summaries must name the sources that are in the tree."""
    import glob
    import json
    want = _source_hash()
    assert ("src:" + want) in L.lib().emgpu_version().decode(), "libemgpu.so is stale: rebuild (make -C em_model_manned_bayes_amd/csrc)"
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[-1]
    tag = os.path.basename(newest).split("_")[0]
    for f in glob.glob(os.path.join(ROOT, "profiles", tag + "*_summary.json")):
        lib = json.load(open(f))["bench_line"]["config"]["lib"]
        assert lib.endswith("src:" + want), "%s was recorded from %s, the tree is src:%s: re-run tools/profile_round.sh" % (os.path.basename(f), lib, want)


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(E.EmgpuError) as ei:
        native.Context(0)
    assert ei.value.code == L.ERR_NO_DEVICE


def test_memory_entry_points_reject_missing_handles_without_touching_a_device():
    """emgpu_trace_* / emgpu_device_* / emgpu_host_* (round 6) with null handles: an argument error and a message, never a crash, and nothing
    here needs a GPU (there is no context to own the memory, and no CPU stand-in for one)."""
    import ctypes as C
    lib = L.lib()
    p, _ = native.make_params(10, 10, 1)
    h = C.c_void_p()
    assert lib.emgpu_trace_alloc(None, None, C.byref(p), L.TRACE_DENSE, 0, C.byref(h)) == L.ERR_ARG and b"null" in lib.emgpu_last_error()
    assert lib.emgpu_trace_out(None, None) == L.ERR_ARG and lib.emgpu_trace_report(None, None) == L.ERR_ARG
    assert lib.emgpu_trace_free(None, None) == L.OK                      # freeing nothing is fine, like free(NULL)
    assert lib.emgpu_device_alloc(None, 1024, C.byref(h)) == L.ERR_ARG and lib.emgpu_device_free(None, None) == L.OK
    assert lib.emgpu_host_alloc(None, 1024, C.byref(h)) == L.ERR_ARG and lib.emgpu_host_free(None, None) == L.OK
    assert lib.emgpu_host_free(None, C.c_void_p(64)) == L.ERR_ARG and lib.emgpu_host_stats(None, None) == L.ERR_ARG
    assert int(lib.emgpu_slot_map_revision()) == 2 and "emgpu 0.4" in lib.emgpu_version().decode()
    # the structs the binding mirrors: sizes as the header lays them out (a drifted field would shift everything behind it)
    assert C.sizeof(L.TraceReport) == 8 + 8 + 4 * 4 + 8 * 4 + 4 + 4 and C.sizeof(L.HostStats) == 4 * 8 + 2 * 8 + 4 * 4


@pytest.mark.parametrize("name", ALL_MODELS)
def test_loader_matches_independent_parser(name, model_dir):
    path = em_io.materialize_model(name, model_dir)
    p = E.em_read(path)
    q = O.parse_model_txt(path)
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
        assert np.array_equal(p["order_transition"], q["order_transition"])
        assert np.array_equal(p["temporal_map"], q["temporal_map"])
        for v, N in q["N_transition"].items():
            assert np.array_equal(p["N_transition"][v], N)
        assert p["native"].is_dynvar_depend == O.OracleModel(q).is_dynvar_depend()
    # em_write -> em_read round trip keeps every number
    out = os.path.join(model_dir, name + "_roundtrip.txt")
    E.em_write(p, out)
    p2 = E.em_read(out)
    for v in range(p["n_initial"]):
        assert np.array_equal(p["N_initial"][v], p2["N_initial"][v]) and np.array_equal(p["boundaries"][v], p2["boundaries"][v])
    assert np.array_equal(p["resample_rates"], p2["resample_rates"])


def test_model_file_invariants(model_dir):
    """Counts and shapes stated in SURVEY.md section 8c / Appendix B, computed from the shipped files."""
    expect = {"uncor_1200code_v2p1": (146516, 74480, False), "uncor_1200only_fwse_v1p2": (183145, 74480, False),
              "cor_v1": (21193, 8100, True), "haa_v1": (1633348, 1472, False), "glider_v1": (9156, 51744, True),
              "uncor_1200code_v1": (36628, 22344, True), "littoral_cor_v1": (21193, 324, False),
              "dueregard_v1": (143898, 11620, False), "terminal_v3_radar_encounter_model": (311650, 0, False)}
    for name, (ni, nt, dep) in expect.items():
        m = native.NativeModel.load_txt(em_io.materialize_model(name, model_dir))
        assert (m.info.n_N_initial, m.info.n_N_transition, m.is_dynvar_depend) == (ni, nt, dep), name
    m = native.NativeModel.load_txt(em_io.materialize_model("terminal_v3_radar_encounter_model", model_dir))
    assert m.get_i32(L.F_R_INITIAL).tolist() == [4, 2, 3, 2, 4, 7, 36, 7, 5, 36, 7, 36, 7, 5, 36]
    m = native.NativeModel.load_txt(em_io.materialize_model("cor_v1", model_dir))
    order = m.get_i32(L.F_ORDER_INITIAL).tolist()     # not upper-triangular: L is a parent of A
    assert order.index(2) < order.index(1) and sorted(order) == list(range(1, 17))
    assert m.get_i32(L.F_TEMPORAL_MAP).reshape(-1, 2).tolist() == [[11, 17], [12, 18], [13, 19], [14, 20]]
    assert m.get_i32(L.F_ZERO_BINS).tolist() == [0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 5, 5, 5, 5, 0, 0]


def test_overwrite_zero_boundaries(model_dir):
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    p = E.em_read(path, isOverwriteZeroBoundaries=True)              # em_read.m:119-121
    assert [len(b) for b in p["boundaries"][:4]] == [0, 0, 0, 9]
    p = E.em_read(path)
    assert [len(b) for b in p["boundaries"][:4]] == [0, 0, 5, 9]
    assert p["cutpoints_initial"][2].tolist() == [1200, 3000, 5000] and p["bounds_initial"][2].tolist() == [500, 12500]
    assert p["cutpoints_initial"][0].tolist() == [2, 3, 4]          # '*' -> 2:n  (em_read.m:130-132)


def test_loader_errors(tmp_path, model_dir):
    with pytest.raises(E.EmgpuError) as ei:
        E.em_read(str(tmp_path / "missing.txt"))
    assert ei.value.code == L.ERR_IO
    bad = tmp_path / "bad.txt"
    bad.write_text("# labels_initial\n\"a\" \n# G_initial\n0 \n# r_initial\n2 \n# N_initial\n1 2 \n# bogus\n1 \n")
    with pytest.raises(E.EmgpuError) as ei:
        E.em_read(str(bad))
    assert ei.value.code == L.ERR_PARSE and "Unknown field" in str(ei.value)   # em_read.m:104
    short = tmp_path / "short.txt"
    short.write_text("# labels_initial\n\"a\", \"b\" \n# G_initial\n0 1 \n0 0 \n# r_initial\n2 3 \n# N_initial\n1 2 3 \n")
    with pytest.raises(E.EmgpuError) as ei:
        E.em_read(str(short))
    assert ei.value.code == L.ERR_PARSE
    cyc = tmp_path / "cyc.txt"
    cyc.write_text("# labels_initial\n\"a\", \"b\" \n# G_initial\n0 1 \n1 0 \n# r_initial\n2 2 \n# N_initial\n1 2 3 4 1 2 3 4 \n")
    with pytest.raises(E.EmgpuError) as ei:
        E.em_read(str(cyc))
    assert ei.value.code == L.ERR_SORT                                           # bn_sort.m:23


def test_priors_and_start(model_dir):
    m = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir))
    assert np.all(m.get_f64(L.F_ALPHA_INITIAL, 7) == 0)
    m.set_prior("dbe")
    assert np.all(m.get_f64(L.F_ALPHA_TRANSITION, 8) == 1.0 / 39200)            # 5 x 7840 node
    m.set_prior(0.5)
    assert np.all(m.get_f64(L.F_ALPHA_INITIAL, 1) == 0.5)
    with pytest.raises(E.EmgpuError) as ei:
        m.set_prior("constant")                                                  # em_sample.m:50 is broken in the reference too
    assert ei.value.identifier == "prior:notdbe"
    m.set_transition_stay_prior(1.0)
    a = m.get_f64(L.F_ALPHA_TRANSITION, 10).reshape(-1, 7).T                     # 7 x 1120
    expect = np.zeros((7, 1120))
    for kk in range(7):
        expect[kk, 160 * kk: 160 * (kk + 1)] = 1
    assert np.array_equal(a, expect)
    assert np.array_equal(a, E.setTransitionPriors(E.em_read(em_io.materialize_model("uncor_1200code_v2p1", model_dir))["G_transition"],
                                                   m.get_i32(L.F_R_TRANSITION), m.get_i32(L.F_TEMPORAL_MAP).reshape(-1, 2), 1)[9])
    m.set_start([1, 4, 2, None, [], float("nan"), None])                          # RUN_uncor.m:43-45
    assert m.get_i32(L.F_START).tolist() == [1, 4, 2, 0, 0, 0, 0]
    with pytest.raises(E.EmgpuError):
        m.set_start([9, 0, 0, 0, 0, 0, 0])


def test_from_arrays_equals_load_txt(model_dir):
    p = E.em_read(em_io.materialize_model("glider_v1", model_dir))
    m = native.NativeModel.from_arrays(p["G_initial"], p["r_initial"], p["N_initial"], p["G_transition"], p["r_transition"],
                                       p["N_transition"], p["temporal_map"], p["boundaries"], p["zero_bins"], p["resample_rates"],
                                       p["labels_initial"], p["labels_transition"])
    for f in (L.F_ORDER_INITIAL, L.F_ORDER_TRANSITION, L.F_ZERO_BINS, L.F_TEMPORAL_MAP):
        assert np.array_equal(m.get_i32(f), p["native"].get_i32(f))
    assert m.get_i32(L.F_ORDER_TRANSITION).tolist() == [1, 2, 3, 4, 5, 7, 8, 6]   # edges among the (t+1) nodes
    assert np.array_equal(m.get_f64(L.F_N_TRANSITION, 6), p["native"].get_f64(L.F_N_TRANSITION, 6))


def _model_fields(m):
    """Everything the C ABI hands out about a model, as comparable Python values."""
    info = (m.n_initial, m.n_transition, m.n_dyn, m.is_dynvar_depend)
    ints = {f: m.get_i32(f).tolist() for f in (L.F_R_INITIAL, L.F_R_TRANSITION, L.F_ORDER_INITIAL, L.F_ORDER_TRANSITION, L.F_TEMPORAL_MAP, L.F_ZERO_BINS,
                                               L.F_START, L.F_G_INITIAL, L.F_G_TRANSITION)}
    tabs = {}
    for v in range(1, m.n_initial + 1):
        tabs[("Ni", v)] = m.get_f64(L.F_N_INITIAL, v).tobytes()
        tabs[("Ai", v)] = m.get_f64(L.F_ALPHA_INITIAL, v).tobytes()
        tabs[("b", v)] = m.get_f64(L.F_BOUNDARIES, v).tobytes()
    for v in range(m.n_initial + 1, m.n_transition + 1):
        tabs[("Nt", v)] = m.get_f64(L.F_N_TRANSITION, v).tobytes()
        tabs[("At", v)] = m.get_f64(L.F_ALPHA_TRANSITION, v).tobytes()
    rates = m.get_f64(L.F_RESAMPLE_RATES).tobytes()
    return info, ints, tabs, rates, m.get_labels(L.F_LABELS_INITIAL), m.get_labels(L.F_LABELS_TRANSITION)


def _plan_columns(m, cols=(0, 1, 7)):
    """The compiled plan as the kernels read it, through the debug hooks (a model without dynamic variables: nothing)."""
    out = []
    for k in range(m.n_dyn):
        for col in cols:
            tvar, r, q, meff, mp = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_uint32()
            thr = np.zeros(64, dtype=np.uint32); cthr = np.zeros(7, dtype=np.uint32)
            rc = L.lib().emgpu_debug_dynamic_column(m._h, k, col, C.byref(tvar), C.byref(r), C.byref(q), thr.ctypes.data, C.byref(meff), cthr.ctypes.data, C.byref(mp))
            out.append((rc, tvar.value, r.value, q.value, meff.value, mp.value, thr.tobytes(), cthr.tobytes()))
    return out


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "cor_v1", "glider_v1", "haa_v1", "terminal_v3_radar_encounter_model", "balloon_v1"])
def test_binary_cache_equals_the_txt_load(name, model_dir, tmp_path):
    """emgpu_model_save_bin / emgpu_model_load_bin -- the parsed model (em_read.m:47-107) and its compiled plan from
    one file.  A model read from the cache equals the model parsed from the .txt in every field the ABI exposes and in the plan's columns;
    priors and `start` set before saving survive; a file from other sources, a truncated file and a foreign file are refused."""
    import time
    path = em_io.materialize_model(name, model_dir)
    t0 = time.perf_counter()
    a = native.NativeModel.load_txt(path)
    cols_a = _plan_columns(a)                       # (forces the plan: what a first upload pays)
    t1 = time.perf_counter()
    binp = str(tmp_path / (name + ".emgpubin"))
    a.save_bin(binp)
    t2 = time.perf_counter()
    b = native.NativeModel.load_bin(binp)
    cols_b = _plan_columns(b)
    t3 = time.perf_counter()
    assert _model_fields(a) == _model_fields(b)
    assert cols_a == cols_b
    print("\n%s: .txt parse + plan %.1f ms, cache read + plan %.1f ms (file %.1f MB)" % (name, (t1 - t0) * 1e3, (t3 - t2) * 1e3, os.path.getsize(binp) / 1e6))
    # a model with a prior and a preset keeps both
    a.set_prior(0.5)
    st = np.zeros(a.n_initial, dtype=np.int32); st[0] = 1
    a.set_start(st)
    a.save_bin(binp)
    c = native.NativeModel.load_bin(binp)
    assert _model_fields(a) == _model_fields(c) and c.get_i32(L.F_START)[0] == 1
    c.set_prior(0.0)                                # a loaded model is an ordinary model: setters work and invalidate its plan
    c.set_start(np.zeros(a.n_initial, dtype=np.int32))
    assert _model_fields(c) == _model_fields(b) and _plan_columns(c) == cols_b
    # refused files
    raw = open(binp, "rb").read()
    bad = str(tmp_path / "bad.emgpubin")
    def flip(k):                                    # one bit of byte k: caught by the payload checksum wherever it is
        return raw[:k] + bytes([raw[k] ^ 0x10]) + raw[k + 1:]
    damaged = [flip(60), flip(len(raw) // 3), flip(len(raw) - 40), flip(len(raw) - 1), flip(47)]   # model fields, tables, the plan's tail, the checksum itself
    for blob in [raw[:12] + b"0123456789ab" + raw[24:], raw[: len(raw) // 2], b"not a model", raw + b"x"] + damaged:
        open(bad, "wb").write(blob)
        with pytest.raises(L.EmgpuError) as e:
            native.NativeModel.load_bin(bad)
        assert e.value.code == L.ERR_PARSE
    with pytest.raises(L.EmgpuError) as e:
        native.NativeModel.load_bin(str(tmp_path / "missing.emgpubin"))
    assert e.value.code == L.ERR_IO


def test_plan_of_a_cold_model_from_many_threads(model_dir):
    """the multi-device entry points run one host thread per device on the SAME model and each resolves the model's plan;
    on a cold model (or after a setter) they all arrive at an empty cache at once.  emgpu_debug_padded_column goes through the same
    plan_of: eight threads on a freshly loaded model, then again after a setter, must all see one and the same plan."""
    import threading
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    for round_ in range(3):
        m = native.NativeModel.load_txt(path)
        for phase in range(2):
            if phase:
                m.set_prior(1.0)                     # bumps the version: the cache is stale for every thread at once
            got, errs = [None] * 8, []
            gate = threading.Barrier(8)

            def work(i):
                try:
                    gate.wait()
                    got[i] = _plan_columns(m)
                except Exception as e:              # noqa
                    errs.append(e)
            ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not errs, errs
            assert all(g == got[0] for g in got)


def test_plan_of_is_clean_under_thread_sanitizer(model_dir, tmp_path):
    """The same scenario in a ThreadSanitizer build of the host-side model code (csrc/emgpu_model.cpp has no HIP in it): the unguarded
    plan_of of round 4 gives 22 race reports here, the locked one none."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "plan_of_race")
    csrc = os.path.join(ROOT, "em_model_manned_bayes_amd", "csrc")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", csrc, os.path.join(ROOT, "tools", "tsan", "plan_of_race.cpp"),
                         os.path.join(csrc, "emgpu_model.cpp"), "-o", exe, "-lpthread"], capture_output=True, text=True)
    if cc.returncode != 0 and "tsan" in cc.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime: " + cc.stderr[-200:])
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, em_io.materialize_model("uncor_1200code_v2p1", model_dir)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr and run.stdout.strip().endswith("ok"), run.stderr[-3000:]


def test_load_cached_writes_reads_and_refreshes(model_dir, tmp_path):
    import shutil
    src = em_io.materialize_model("uncor_1200code_v1", model_dir)
    path = str(tmp_path / "m.txt")
    shutil.copy(src, path)
    a = native.NativeModel.load_cached(path)
    binp = path + ".z123.o0.emgpubin"
    assert os.path.exists(binp)
    b = native.NativeModel.load_cached(path)                      # from the cache
    assert _model_fields(a) == _model_fields(b)
    c = native.NativeModel.load_cached(path, is_overwrite_zero_boundaries=True)   # other options, another cache file
    v = next(k for k in (1, 2, 3) if a.get_f64(L.F_BOUNDARIES, k).size > 0)
    assert os.path.exists(path + ".z123.o1.emgpubin") and c.get_f64(L.F_BOUNDARIES, v).size == 0
    open(binp, "wb").write(b"garbage")                             # a broken cache is ignored and rewritten
    os.utime(binp, None)
    d = native.NativeModel.load_cached(path)
    assert _model_fields(a) == _model_fields(d) and os.path.getsize(binp) > 1000


def test_number_scanner_equals_strtod(tmp_path):
    """load_txt's own number scanner (digits -> one exact multiply or divide, strtod for the rest) returns the doubles Python's float()
    does -- on the forms the model files use and on the ones that must take the fallback."""
    toks = ["0", "-0", "7", "459966244", "1558178395", "0.0127706", ".5", "-.25", "5.", "1e-3", "1E+3", "-2.5e-7", "3.141592653589793", "0.1", "0.3",
            "123456789012345678", "1234567890123456789012", "9007199254740993", "1e22", "1e23", "1.7976931348623157e308", "4.9e-324", "2.2250738585072014e-308",
            "0.000001", "1e-22", "1e-23", "100000000000000000000", "6076.1154855643", "1.68780972222222", "0.592484", "-1500", "+12", "00012", "1.0e0",
            "8.5e-3", "99999999999999999999e-20", "0.30000000000000004", "5e-1", "72057594037927936", "72057594037927937"]
    rng = np.random.RandomState(3)
    toks += ["%.17g" % v for v in rng.standard_normal(200) * 10.0 ** rng.randint(-12, 12, 200)]
    toks += ["%d" % v for v in rng.randint(0, 2**31 - 1, 200)]
    toks += ["%.7f" % v for v in rng.uniform(-1000, 1000, 200)]
    want = sorted(set(float(t) for t in toks))
    # strictly increasing boundaries of ONE variable: the values themselves, sorted (duplicates removed)
    ordered = sorted(set(toks), key=float)
    vals = []
    for t in ordered:
        if not vals or float(t) > float(vals[-1]):
            vals.append(t)
    r = len(vals) - 1
    txt = "# labels_initial\n\"x\"\n# G_initial\n0\n# r_initial\n%d\n# N_initial\n%s\n# boundaries\n%s\n# resample_rates\n0\n" % (
        r, " ".join(["1"] * r), "  ".join(vals) + " ")
    path = str(tmp_path / "scan.txt")
    open(path, "w").write(txt)
    m = native.NativeModel.load_txt(path, is_overwrite_zero_boundaries=False)
    got = m.get_f64(L.F_BOUNDARIES, 1)
    assert got.tolist() == [float(t) for t in vals] and len(vals) > 500 and set(got.tolist()) <= set(want)


def _oracle_bin(w, u):
    w = np.ascontiguousarray(np.asarray(w, dtype=np.float64))
    return O.lib().em_select_random_r(w.ctypes.data_as(C.c_void_p), len(w), C.c_double(u))


def test_integer_thresholds_reproduce_the_f64_compare(model_dir):
    """select_random.m:17-20 in f64 (oracle) == count of u32 thresholds (what the kernels do)."""
    lib = L.lib()
    rng = np.random.RandomState(5)
    pp = O.parse_model_txt(em_io.materialize_model("dueregard_v1", model_dir))      # largest counts (1.56e9)
    cols = [pp["N_transition"][8][:, j] for j in rng.randint(0, 378, 40)]
    cols += [pp["N_initial"][6][:, j] for j in rng.randint(0, pp["N_initial"][6].shape[1], 40)]
    cols += [np.array([0., 0, 0, 0]), np.array([0., 5, 0, 5]), np.array([1., 0, 0, 0]), np.array([0., 0, 0, 7]),
             np.array([1e-5, 2.5e-5, 1e9]), np.full(5, 1 / 39200.0) + np.array([3, 0, 1, 0, 0.])]
    u32 = O.lib().em_uniform32
    for w in cols:
        r = len(w)
        thr = np.zeros(max(r - 1, 1), dtype=np.uint32)
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert lib.emgpu_debug_column_thresholds(w.ctypes.data, r, thr.ctypes.data) == 0
        assert np.all(np.diff(thr[: r - 1].astype(np.int64)) >= 0)
        xs = set(int(x) for x in rng.randint(0, 2**32, 300, dtype=np.uint64))
        xs |= {0, 1, 2**32 - 1, 2**32 - 2, 2**31}
        for t in thr[: r - 1]:
            xs |= {int(t), max(int(t) - 1, 0), min(int(t) + 1, 2**32 - 1)}
        for x in xs:
            xp = min(x, 2**32 - 2)
            got = 1 + int(np.sum(xp >= thr[: r - 1].astype(np.uint64)))
            assert got == _oracle_bin(w, u32(x)), (w, x)
    for rate in [0.0, 0.0127706, 0.0771499, 0.5, 1e-9, 2.0 ** -32, 0.999999]:
        R = lib.emgpu_debug_bernoulli_threshold(rate)
        for x in {0, 1, max(R - 1, 0), R, min(R + 1, 2**32 - 1), 2**32 - 1, 12345678}:
            assert (min(x, 2**32 - 2) < R) == (u32(x) < rate), (rate, x)


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "uncor_1200only_fwse_v1p2", "cor_v1", "haa_v1", "glider_v1"])
def test_compacted_columns_select_the_same_bin(name, model_dir):
    """EmgpuPlan::cthr (distinct thresholds + bin map) against the plain r-1 thresholds, on sampled and
    on every adjacent x, and both against select_random on the Dirichlet-smoothed column (prior 0)."""
    p = E.em_read(em_io.materialize_model(name, model_dir))
    pp = O.parse_model_txt(em_io.materialize_model(name, model_dir))
    lib = L.lib()
    h = p["native"]._h
    rng = np.random.RandomState(11)
    u32 = O.lib().em_uniform32
    nd = len(pp["temporal_map"])
    seen_meff = []
    for k in range(nd):
        tvar, r, q, meff, mp = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_uint32()
        thr = np.zeros(15, dtype=np.uint32); cthr = np.zeros(7, dtype=np.uint32)
        args = lambda col: (h, k, col, C.byref(tvar), C.byref(r), C.byref(q), thr.ctypes.data, C.byref(meff), cthr.ctypes.data, C.byref(mp))
        assert lib.emgpu_debug_dynamic_column(*args(0)) == 0
        N = pp["N_transition"][tvar.value - 1]
        assert N.shape == (r.value, q.value)
        cols = sorted(set(int(c) for c in rng.randint(0, q.value, 60)) | {0, q.value - 1})
        worst = 0
        for col in cols:
            assert lib.emgpu_debug_dynamic_column(*args(col)) == 0
            if meff.value == 0:
                continue
            t = thr[: r.value - 1].astype(np.uint64)
            ct = cthr[: meff.value].astype(np.uint64)
            real = sorted(set(int(x) for x in t if 0 < x < 2**32 - 1))
            assert len(real) <= meff.value and [int(x) for x in ct[: len(real)]] == real
            assert all(int(x) == (real[-1] if real else 2**32 - 1) for x in ct[len(real):])   # padding repeats the last threshold
            worst = max(worst, len(real))
            xs = set(int(x) for x in rng.randint(0, 2**32, 40, dtype=np.uint64)) | {0, 1, 2**32 - 1, 2**32 - 2}
            for x in t:
                xs |= {int(x), max(int(x) - 1, 0), min(int(x) + 1, 2**32 - 1)}
            w = np.ascontiguousarray(N[:, col], dtype=np.float64)   # all-zero columns: select_random.m:17-20 gives bin 1
            width = C.c_int32()
            pw = np.zeros(8, dtype=np.uint32)
            assert lib.emgpu_debug_padded_column(h, k, col, C.byref(width), pw.ctypes.data) == 0
            assert width.value == (4 if meff.value <= 3 else 8 if meff.value <= 6 else 0)
            mfix = {4: 3, 8: 6, 0: 0}[width.value]
            pt = pw[:mfix].astype(np.uint64)
            table = [int(b) for b in pw[mfix: mfix + (1 if mfix == 3 else 2)].view(np.uint8)] if mfix else []
            for x in xs:
                xp = min(x, 2**32 - 2)
                full = 1 + int(np.sum(xp >= t))
                n = int(np.sum(xp >= ct))
                assert (mp.value >> (4 * n)) & 15 == full, (name, k, col, x)
                assert full == _oracle_bin(w, u32(x)), (name, k, col, x)
                if mfix:                                  # EmgpuPlan::pthr: byte table indexed by the thresholds that did not fire
                    assert table[int(np.sum(xp < pt))] == full, (name, k, col, x)
        seen_meff.append(meff.value)
    if name == "uncor_1200code_v2p1":
        assert seen_meff == [2, 4, 2]                    # what k_uncor_fast<7,2,4,2> is built for
    assert lib.emgpu_debug_dynamic_column(h, nd, 0, *([C.byref(C.c_int64())] * 7)) == L.ERR_ARG   # no such variable


@pytest.mark.parametrize("name", ["cor_v1", "glider_v1", "uncor_1200code_v1", "uncor_1200code_v2p1", "haa_v1", "cor_v2p1_like"])
def test_packed_compare_columns_decide_like_select_random(name, model_dir):
    """EmgpuPlan::d_poffpk, the form k_dbn_step2 decides a draw from: for EVERY high halfword x_h of a column, either the packed
    rule flags the draw for the full 32-bit compare (an odd sum of min(sat16(x_h - T'), 2), or x_h == 0), or the bin it reads off
    (nibble sum / 2) is the bin of select_random.m:17-20 for every low halfword -- checked at the two extreme low halfwords and
    against the plain thresholds; whenever the low halfword can change the bin the draw IS flagged."""
    from em_model_manned_bayes_amd import synthetic
    path = synthetic.write_correlated_v2p1_like(model_dir) if name == "cor_v2p1_like" else em_io.materialize_model(name, model_dir)
    p = E.em_read(path)
    pp = O.parse_model_txt(path)
    lib = L.lib()
    h = p["native"]._h
    rng = np.random.RandomState(5)
    xh = np.arange(65536, dtype=np.int64)
    checked = 0
    for k in range(len(pp["temporal_map"])):
        tvar, r, q, meff, mp = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_uint32()
        thr = np.zeros(15, dtype=np.uint32); cthr = np.zeros(7, dtype=np.uint32)
        args = lambda col: (h, k, col, C.byref(tvar), C.byref(r), C.byref(q), thr.ctypes.data, C.byref(meff), cthr.ctypes.data, C.byref(mp))
        assert lib.emgpu_debug_dynamic_column(*args(0)) == 0
        if meff.value == 0 or meff.value > 6:
            continue
        for col in sorted(set(int(c) for c in rng.randint(0, q.value, 25)) | {0, q.value - 1}):
            assert lib.emgpu_debug_dynamic_column(*args(col)) == 0
            pk = np.zeros(4, dtype=np.uint32)
            assert lib.emgpu_debug_pk_column(h, k, col, pk.ctypes.data) == 0
            if meff.value <= 3:   # the PLAIN form {H0, H1, H2, map}: a_t = H_t - x_h, fired <=> a_t < 0, tie <=> a_t == 0, bins 7 bits apart
                H = pk[:3].astype(np.int64)
                a = H[None, :] - xh[:, None]
                flagged = (a == 0).any(axis=1)
                off = 7 * (a < 0).sum(axis=1)
                t = thr[: r.value - 1].astype(np.int64)
                lo_bin = 1 + (np.minimum((xh << 16), 2**32 - 2)[:, None] >= t[None, :]).sum(axis=1)
                hi_bin = 1 + (np.minimum((xh << 16) | 0xFFFF, 2**32 - 2)[:, None] >= t[None, :]).sum(axis=1)
                pk_bin = (int(pk[3]) >> off) & 15
                assert np.all(flagged | ((pk_bin == lo_bin) & (pk_bin == hi_bin))), (name, k, col)
                assert np.all(flagged[lo_bin != hi_bin]), (name, k, col)
                assert flagged.mean() < 0.02, (name, k, col, flagged.mean())
                checked += 1
                continue
            tq = np.array([(int(pk[w]) >> (16 * hh)) & 0xFFFF for w in range(3) for hh in range(2)], dtype=np.int64)
            assert np.all(np.diff(tq[tq < 0xFFFF]) > 0)                      # strictly increasing where real
            d = np.minimum(np.maximum(xh[:, None] - tq[None, :], 0), 2)      # min(sat16(x_h - T'), 2)
            ssum = d.sum(axis=1)
            flagged = (ssum % 2 == 1) | (xh == 0)
            t = thr[: r.value - 1].astype(np.int64)
            lo_bin = 1 + (np.minimum((xh << 16), 2**32 - 2)[:, None] >= t[None, :]).sum(axis=1)              # low halfword 0x0000
            hi_bin = 1 + (np.minimum((xh << 16) | 0xFFFF, 2**32 - 2)[:, None] >= t[None, :]).sum(axis=1)     # low halfword 0xFFFF
            pk_bin = (int(pk[3]) >> (4 * (ssum // 2))) & 15
            assert np.all(flagged | ((pk_bin == lo_bin) & (pk_bin == hi_bin))), (name, k, col)
            assert np.all(flagged[lo_bin != hi_bin]), (name, k, col)       # the low halfword matters => referred to the exact compare
            assert flagged.mean() < 0.02, (name, k, col, flagged.mean())     # and that stays rare
            checked += 1
    assert checked > 0


def test_shard_ranges_cover_exactly():
    for n, w in [(10, 3), (50_000_000, 8), (7, 8), (0, 2), (1, 1)]:
        ranges = [sharding.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 1
    firsts = sorted(sharding.step_first_index(k, r, 4, 100) for k in range(3) for r in range(4))
    assert firsts == [100 * i for i in range(12)]


def test_mixed_batch_blocks_partition():
    n, M, W = 50_000_000, 6, 8
    seen = 0
    for r in range(W):
        lo, hi = sharding.shard_range(n, r, W)
        calls = sharding.mixed_batch_blocks(n, M, lo, hi)
        assert sum(c for _, _, c in calls) == hi - lo
        assert all(lo <= f and f + c <= hi for _, f, c in calls)
        seen += sum(c for _, _, c in calls)
    assert seen == n
    assert [m for m, _, _ in sharding.mixed_batch_blocks(60, 6)] == list(range(6))


def test_class_surface(model_dir):
    mdl = E.UncorEncounterModel(parameters_filename=em_io.materialize_model("uncor_1200only_fwse_v1p2", model_dir))
    assert mdl.n_initial == 7 and mdl.n_transition == 10 and not mdl.isRotorcraft
    assert mdl.r_initial.tolist() == [5, 4, 4, 8, 5, 7, 7] and mdl.r_transition.tolist() == [5, 4, 4, 8, 5, 7, 7, 5, 7, 7]
    assert mdl.order_transition.tolist() == list(range(1, 11))
    assert mdl.dediscretize_parameters[0].size == 0 and mdl.dediscretize_parameters[6].tolist() == [-8, -6, -4.5, -1.5, 1.5, 4.5, 6, 8]
    assert [len(c) for c in mdl.cutpoints_fine[4]] == [2] * 5
    s = mdl.struct()
    for k in ("G_initial", "G_transition", "temporal_map", "r_transition", "n_initial", "N_initial", "N_transition",
              "order_initial", "order_transition"):                               # dbn_sample.m:25-33
        assert k in s
    start = [None] * 7
    start[0], start[1], start[2] = 1, 4, 2
    mdl.start = start
    assert mdl.native.get_i32(L.F_START).tolist() == [1, 4, 2, 0, 0, 0, 0]
    ev = E.EncounterModelEvents()
    assert ev.event.tolist() == [[0, 0, 0, 0]]
    ev = E.EncounterModelEvents(event=[[0, 1, 2, 3], [5, 4, 3, 2]])
    assert ev.time_s.tolist() == [0, 5] and ev.longitudeAccel_ftpss.tolist() == [3, 2]
    t = E.CorTerminalModel(parameters_directory=None)
    assert t.n_initial == 15 and t.getDynamicLimits(2)["maxVel_ft_s"] == 506 and t.bounds_sample.shape == (15, 2)
    with pytest.raises(NotImplementedError):
        mdl.track(1, 10, coordSys="geodetic")      # DEM / obstacle file / placeTrack: out of scope
    with pytest.raises(NotImplementedError):
        t.track(1)                                 # the trajectory-model files are not in this (default) directory


def test_init_start_terminal_grid():
    t = E.CorTerminalModel()
    grid = t.InitStartTerminal(nSamples=36)                               # InitStartTerminal.m defaults: class B excluded
    assert len(grid) == 36 and len(grid[0]) == 15
    combos = sorted({tuple(g[:3]) for g in grid})
    assert len(combos) == 3 * 2 * 3 and all(c[0] in (2, 3, 4) for c in combos)
    assert all(v is None for v in grid[0][3:])
    assert len(t.InitStartTerminal(nSamples=5)) == 18                     # fewer samples than combinations -> one each (:50-53)
    t.start = grid[0]
    assert t.native.get_i32(L.F_START).tolist()[:3] == list(grid[0][:3])


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under the package (Python or C++/HIP) may import,
    link or open it, and nothing may read /root/reference at run time."""
    pkg = os.path.join(ROOT, "em_model_manned_bayes_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".c", ".m")) and f != "Makefile":
                continue
            text = open(os.path.join(base, f), errors="replace").read()
            if re.search(r"(import\s+oracle|from\s+oracle|em_oracle|libem_oracle|oracle/|/root/reference)", text):
                offenders.append(os.path.relpath(os.path.join(base, f), ROOT))
    assert offenders == [], offenders
    for f in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(ROOT, f)).read()
        assert "/root/reference" not in text


def test_legacy_label_names_and_number_format():
    from em_model_manned_bayes_amd import legacy
    labs = ['"G"', '"A"', '"L"', '"v"', '"\\dot v"', '"\\dot h"', '"\\dot \\psi"']
    assert [legacy.make_valid_name(legacy._erase(s)) for s in labs] == ["G", "A", "L", "v", "dotV", "dotH", "dotPsi"]   # sample2track.m:27-31
    tl = ['"\\dot v(t+1)"', '"\\dot h(t+1)"', '"\\dot \\psi(t+1)"']
    assert [legacy.make_valid_name(legacy._erase(s)) for s in tl] == ["dotV_t_1_", "dotH_t_1_", "dotPsi_t_1_"]          # :38-40
    assert [legacy._g(x) for x in (3, 0.5, 1234567.0, 1e-7, 145.123456789, -0.0)] == ["3", "0.5", "1.23457e+06", "1e-07", "145.123", "-0"]
    assert abs(legacy.FT_PER_NM / 3600.0 - 1.68780985710119) < 1e-13


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_api_demo")
    pkg = os.path.join(ROOT, "em_model_manned_bayes_amd")
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-std=c99", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_api_demo.c"), "-L" + pkg, "-lemgpu", "-Wl,-rpath," + pkg, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_the_demo_fails_cleanly_without_a_device(tmp_path, model_dir):
    """include/emgpu.h compiles as C99 with -Wall -Werror; the plain-C host reports errors through the ABI."""
    import subprocess
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "missing.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "-> -2: cannot open" in r.stderr                      # EMGPU_ERR_IO
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, em_io.materialize_model("uncor_1200code_v2p1", model_dir), "10", "8"], capture_output=True, text=True)
        assert r.returncode == 1 and "emgpu_ctx_create" in r.stderr and "-> -8" in r.stderr   # EMGPU_ERR_NO_DEVICE: no CPU fallback


def test_random_models_parse_identically(tmp_path):
    """Generated models (tests/util.random_model) through em_write -> the C++ loader and the oracle's parser."""
    from util import random_model
    for seed in range(12):
        parms = random_model(np.random.RandomState(100 + seed))
        path = str(tmp_path / ("rand%d.txt" % seed))
        em_io.em_write(parms, path)
        p = E.em_read(path)
        pp = O.parse_model_txt(path)
        assert np.array_equal(np.asarray(p["temporal_map"]).reshape(-1, 2), np.asarray(pp["temporal_map"]).reshape(-1, 2))
        assert [int(z) if np.size(z) else 0 for z in p["zero_bins"]] == [int(z) for z in pp["zero_bins"]]   # [] (MATLAB) == 0 (none)
        assert np.array_equal(p["order_transition"], pp["order_transition"])
        for v in range(parms["n_initial"]):
            assert np.array_equal(p["N_initial"][v], parms["N_initial"][v]) and np.array_equal(pp["N_initial"][v], parms["N_initial"][v])
            assert np.array_equal(np.asarray(p["boundaries"][v]), np.asarray(pp["boundaries"][v]))
        for v in range(parms["n_initial"], parms["n_transition"]):
            assert np.array_equal(p["N_transition"][v], parms["N_transition"][v]) and np.array_equal(pp["N_transition"][v], parms["N_transition"][v])


def test_mex_gateway_compiles():
    """The MATLAB gateway cannot be run here (no MATLAB); it is at least type-checked against the C ABI header with a
    declaration-only stand-in for mex.h (tests/stubs/mex.h)."""
    import subprocess
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "stubs"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "em_model_manned_bayes_amd", "matlab", "emgpu_mex.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every command a .m file sends exists in the gateway, and the gateway covers every caller SURVEY.md 8(b) lists
    src = open(os.path.join(ROOT, "em_model_manned_bayes_amd", "matlab", "emgpu_mex.c")).read()
    have = set(re.findall(r'!strcmp\(cmd, "([a-z_0-9]+)"\)', src))
    assert {"load_txt", "em_read", "from_struct", "set_prior", "set_alpha", "set_start", "bn_sample", "sample_uncor", "geom_sample",
            "propagate_terminal", "track_uncor", "sample2track", "device_count", "use_devices", "free", "save_bin", "load_bin"} <= have
    used = set()
    mdir = os.path.join(ROOT, "em_model_manned_bayes_amd", "matlab")
    for base, _, files in os.walk(mdir):
        for f in files:
            if f.endswith(".m"):
                used |= set(re.findall(r"emgpu_mex\('([a-z_0-9]+)'", open(os.path.join(base, f)).read()))
    assert used and used <= have, used - have
    for f in ("shadow/bn_sample.m", "shadow/dbn_sample.m", "shadow/dbn_hierarchical_sample.m", "shadow/em_read.m",
              "@UncorEncounterModelGPU/UncorEncounterModelGPU.m", "@CorTerminalModelGPU/CorTerminalModelGPU.m"):
        assert os.path.exists(os.path.join(mdir, f)), f


def _matlab_code_lines(text):
    """Code of a .m file with comments, strings and line continuations removed (enough for a structural lint)."""
    out, cont = [], ""
    for raw in text.split("\n"):
        line, i, code = raw, 0, []
        in_s = in_d = False
        prev = ""
        while i < len(line):
            c = line[i]
            if in_s:
                if c == "'":
                    if i + 1 < len(line) and line[i + 1] == "'": i += 1
                    else: in_s = False
            elif in_d:
                if c == '"':
                    if i + 1 < len(line) and line[i + 1] == '"': i += 1
                    else: in_d = False
            elif c == "%":
                break
            elif c == "'" and not (prev.isalnum() or prev in "_)]}.'"):   # a quote after an operand is a transpose
                in_s = True; code.append(" S ")
            elif c == '"':
                in_d = True; code.append(" S ")
            else:
                code.append(c)
            prev = c      # (a quote right after a blank starts a string, as inside [a 'b'])
            i += 1
        assert not in_s and not in_d, "unterminated string: " + raw
        t = "".join(code).rstrip()
        if t.endswith("..."):
            cont += t[:-3] + " "
            continue
        out.append(cont + t); cont = ""
    return out


def test_matlab_files_are_structurally_sound():
    """No MATLAB here: a lint of every shipped .m file -- strings terminated, brackets balanced per statement, every block
    opener (function / if / for / while / switch / try / classdef / methods / properties / parfor) closed by an `end`."""
    import glob
    files = glob.glob(os.path.join(ROOT, "em_model_manned_bayes_amd", "matlab", "**", "*.m"), recursive=True) + glob.glob(os.path.join(ROOT, "tools", "matlab", "*.m"))
    assert len(files) >= 9
    openers = {"function", "if", "for", "while", "switch", "try", "classdef", "methods", "properties", "parfor"}
    for f in files:
        depth, n_fun, has_classdef = 0, 0, False
        for line in _matlab_code_lines(open(f).read()):
            assert line.count("(") == line.count(")") and line.count("[") == line.count("]") and line.count("{") == line.count("}"), (f, line)
            # `end` inside an index expression is arithmetic, not a block end: drop bracketed text first
            flat = line
            for _ in range(8):
                flat = re.sub(r"\([^()]*\)|\[[^\[\]]*\]|\{[^{}]*\}", " ", flat)
            for stmt in re.split(r"[;,]", flat):
                words = re.findall(r"[A-Za-z_][A-Za-z_0-9]*", stmt)
                if not words: continue
                if words[0] in openers:
                    depth += 1
                    n_fun += words[0] == "function"; has_classdef |= words[0] == "classdef"
                depth -= sum(1 for w in words if w == "end")
                assert depth >= 0, (f, line)
        if has_classdef or n_fun == 0:
            assert depth == 0, (f, depth)
        else:
            assert depth in (0, n_fun), (f, depth, n_fun)      # script-style function files may leave their functions open


_MATLAB_BUILTINS = set("""abs addParameter addRequired all any arrayfun assert cell cell2struct cellfun char cosd double eps erase error
fclose fgetl fileparts find fopen fprintf fullfile getenv ischar isempty isequal isfolder isinf isnan logical mat2str max mfilename min nan
num2cell numel onCleanup parse randi readmatrix repmat rethrow rng seconds sind size sortrows sprintf squeeze str2double strcmp strcmpi
strsplit timetable toc tic zeros inputParser struct isnumeric isfield""".split())


@pytest.mark.skipif(not os.path.isdir("/root/reference/code/matlab"), reason="needs the reference mount (build container only)")
def test_matlab_files_call_only_functions_that_exist():
    """Every name the shipped .m files call like a function is a variable of that file, one of our own functions, a function or
    method of the reference (read from the mount), the mex gateway, or on a hand-kept list of MATLAB builtins: a misspelt helper
    would otherwise only surface on a machine with MATLAB."""
    import glob
    files = glob.glob(os.path.join(ROOT, "em_model_manned_bayes_amd", "matlab", "**", "*.m"), recursive=True) + glob.glob(os.path.join(ROOT, "tools", "matlab", "*.m"))
    fun_re = r"function\s+(?:\[?([^=\n]*?)\]?\s*=\s*)?([A-Za-z_]\w*)\s*(?:\(([^)]*)\))?"
    known = {"emgpu_mex"} | {os.path.basename(f)[:-2] for f in files}
    for f in glob.glob("/root/reference/code/matlab/**/*.m", recursive=True):
        known.add(os.path.basename(f)[:-2])
        known |= {m.group(2) for m in re.finditer(fun_re, open(f, errors="ignore").read())}
    for f in files:
        text = "\n".join(_matlab_code_lines(open(f).read()))
        names = set(re.findall(r"([A-Za-z_]\w*)\s*(?:\([^=\n]*\)|\{[^=\n]*\})?\s*=(?!=)", text))          # assigned
        for grp in re.findall(r"\[([^\]=\n]*)\]\s*=", text) + re.findall(r"@\(([^)]*)\)", text):          # [a, b] = ..., @(x) ...
            names |= set(re.findall(r"[A-Za-z_]\w*", grp))
        names |= set(re.findall(r"for\s+([A-Za-z_]\w*)\s*=", text)) | set(re.findall(r"catch\s+([A-Za-z_]\w*)", text))
        for m in re.finditer(fun_re, text):
            known.add(m.group(2))
            for g in (m.group(1), m.group(3)):
                if g: names |= set(re.findall(r"[A-Za-z_]\w*", g))
        calls = {m.group(1) for m in re.finditer(r"(?<![\w.])([A-Za-z_]\w*)\s*\(", text)}
        unknown = sorted(c for c in calls if c not in names and c not in known and c not in _MATLAB_BUILTINS)
        assert not unknown, (os.path.basename(f), unknown)


def test_uncor_dynamic_limits_match_the_oracle_and_known_answers(model_dir):
    """@UncorEncounterModel/getDynamicLimits.m: the product's host restatement (emgpu_limits.cpp, the table the track
    kernel indexes) against the oracle's separately written one over random arguments, on an 'ordered' model
    (G A L v . \\dot h at 1 2 3 4 6: the sliced branch, :17-83), a rotorcraft file (:107-109) and a model whose
    variables are elsewhere (glider_v1: the global branch, :85-88)."""
    rs = np.random.RandomState(7)
    for name, rot in (("uncor_1200code_v2p1", False), ("uncor_1200only_rotorcraft_v1p2", True), ("glider_v1", False)):
        path = em_io.materialize_model(name, model_dir)
        om, nm = O.OracleModel(O.parse_model_txt(path)), native.NativeModel.load_txt(path)
        for _ in range(120):
            init = np.array([rs.randint(1, r + 1) for r in om.parms["r_initial"]], dtype=float)
            up, sp = np.sort(rs.uniform(300, 13000, 2)), np.sort(rs.uniform(10, 520, 2))
            a = O.uncor_dynamic_limits(om, init, up[0], up[1], sp[0], sp[1], rot)
            b = native.uncor_dynamic_limits(nm, init, up[0], up[1], sp[0], sp[1], rot)
            assert np.array_equal(a, b), (name, init, up, sp)
            assert b[2] >= 0      # (b[0] <= b[1] only holds for (G, A, L) combinations the model has counts for)
            if rot:
                assert b[1] <= 304                       # :107-109
            else:
                assert b[0] >= 30                        # :110-112
    # hand-checkable: a 2-variable toy has no G/A/L => global branch; v counts [1 97 1 1] -> 1st pct in bin 1, 99th in bin 3
    mdl = E.UncorEncounterModel(em_io.materialize_model("uncor_1200code_v2p1", model_dir))
    lim = mdl.getDynamicLimits(np.array([1, 4, 700.0, 100.0, 0.0, 0.0, 0.0]), {"up_ft": np.array([650.0, 700.0]), "speed_ftps": np.array([160.0, 170.0])})
    assert set(lim) == {"minVel_ft_s", "maxVel_ft_s", "maxVertRate_ft_s"} and lim["minVel_ft_s"] >= 30


def test_point_mass_dynamics_known_answers():
    """The documented stand-in for em-core's run_dynamics_fast (oracle/em_oracle.c, section f1): straight and level flight
    stays straight and level; a commanded climb converges to hdot; a coordinated turn of psidot closes a circle."""
    dyn = [1.7, 500.0, -50.0, 50.0, np.deg2rad(3.0), 1e6]
    rows, mm = O.point_mass_dynamics([200.0, 0, 0, 1000.0, 0, 0, 0, 0], np.zeros((30, 3)), dyn)
    assert rows.shape == (301, 8) and np.allclose(rows[:, 0], np.arange(301) / 10.0)
    assert np.allclose(rows[-1, 1:5], [6000.0, 0.0, 1000.0, 200.0]) and mm[4] == 0
    ctrl = np.zeros((60, 3)); ctrl[:, 0] = 10.0                                   # 600 ft/min climb
    rows, mm = O.point_mass_dynamics([200.0, 0, 0, 1000.0, 0, 0, 0, 0], ctrl, dyn)
    assert abs((rows[-1, 3] - rows[-11, 3]) - 10.0) < 1e-9 and abs(rows[-1, 6] - np.arcsin(10.0 / 200.0)) < 1e-12
    assert np.all(np.abs(np.diff(rows[:, 6])) <= np.deg2rad(3.0) * 0.1 + 1e-15)  # pitch rate limited by dyn(5)
    ctrl = np.zeros((120, 3)); ctrl[:, 1] = np.deg2rad(3.0)                       # standard-rate turn: 360 deg in 120 s
    rows, mm = O.point_mass_dynamics([200.0, 0, 0, 1000.0, 0, 0, np.arctan(200.0 * np.deg2rad(3.0) / 32.2), 0], ctrl, dyn)
    assert abs(rows[-1, 7] - 2 * np.pi) < 1e-9 and np.hypot(rows[-1, 1], rows[-1, 2]) < 25.0
    ctrl = np.zeros((40, 3)); ctrl[:, 2] = 20.0                                   # acceleration stops at v_high
    rows, mm = O.point_mass_dynamics([400.0, 0, 0, 1000.0, 0, 0, 0, 20.0], ctrl, dyn)
    assert rows[:, 4].max() == 500.0 and mm[3] == 500.0


def test_start_log_weight(model_dir):
    """the importance weight of a start distribution = the model probability of the preset values
    (UncorEncounterModel.m:204, RUN_uncor.m:43-48), from N + alpha like select_random; dependent presets are refused."""
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    mdl = E.UncorEncounterModel(path)
    assert mdl.start_log_weight == 0.0
    st = [None] * 7
    st[0], st[1], st[2] = 1, 4, 2                                     # RUN_uncor.m:43-45
    mdl.start = st
    N = mdl.N_initial
    col_L = (1 - 1) + 4 * (4 - 1)                                      # asub2ind([4 4], [1 4]) - 1
    want = np.log(N[0][0, 0] / N[0][:, 0].sum()) + np.log(N[1][3, 0] / N[1][:, 0].sum()) + np.log(N[2][1, col_L] / N[2][:, col_L].sum())
    assert abs(mdl.start_log_weight - want) < 1e-12
    mdl.prior = 0.5                                                    # alpha enters like in select_random
    want = sum(np.log((N[v][b, c] + 0.5) / (N[v][:, c] + 0.5).sum()) for v, b, c in ((0, 0, 0), (1, 3, 0), (2, 1, col_L)))
    assert abs(mdl.start_log_weight - want) < 1e-12
    st = [None] * 7
    st[2] = 2                                                          # L preset without its parents G, A
    mdl.start = st
    with pytest.raises(E.EmgpuError) as ei:
        mdl.start_log_weight
    assert ei.value.code == L.ERR_PRESET


def test_one_hip_runtime_whatever_the_import_order():
    """_lib._share_hip_runtime: a process that loads libemgpu.so and imports torch (in either order) holds ONE libamdhip64 /
    libhsa-runtime64, not the system's next to the wheel's (two HSA instances on one GPU: foreign stream handles, and on some
    boxes `No HIP GPUs are available` from the second one)."""
    import subprocess
    code = r'''
import sys
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch-first":
    import torch
from em_model_manned_bayes_amd import _lib
_lib.lib()
import torch
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime64" in l})
print(len([x for x in libs if "libamdhip64" in x]), len([x for x in libs if "libhsa-runtime64" in x]))
''' % ROOT
    for order in ("emgpu-first", "torch-first"):
        out = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split() == ["1", "1"], (order, out.stdout)


def test_bench_telemetry_reads_the_card_with_this_ranks_pci_address(tmp_path):
    """bench.py's GpuTelemetry: a box lists more cards in sysfs than the process can see (a 1-GPU slice of an 8-GPU node), so the sensor
    is chosen by the PCI address of the rank's device, not by its index -- round 4 first read an idle neighbour's 94 MHz."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    root = tmp_path / "sys"
    for card, addr, mhz, watts in ((0, "0000:05:00.0", 94, 246), (8, "0000:0d:00.0", 2240, 1350)):
        dev = root / "devices" / "pci0000:00" / addr
        hw = dev / "hwmon" / ("hwmon%d" % card)
        hw.mkdir(parents=True)
        (hw / "freq1_label").write_text("sclk\n"); (hw / "freq1_input").write_text("%d\n" % (mhz * 1000000))
        (hw / "power1_input").write_text("%d\n" % (watts * 1000000)); (hw / "power1_cap").write_text("1400000000\n")
        (hw / "temp1_label").write_text("mem\n"); (hw / "temp1_input").write_text("66000\n")
        drm = root / "class" / "drm" / ("card%d" % card)
        drm.mkdir(parents=True)
        os.symlink(str(dev), str(drm / "device"))
    t = bench.GpuTelemetry(0, "0000:0D:00.0", sysfs_root=str(root))
    assert t.card.startswith("card8") and t._read() == (2240.0, 1350.0)
    t.samples = [t._read()] * 5
    out = t.summary()
    assert out["sclk_mhz"]["median"] == 2240.0 and out["socket_power_w"]["median"] == 1350.0 and out["power_cap_w"] == 1400.0 and out["temperature_c"] == {"mem": 66.0}
    assert bench.GpuTelemetry(0, "0000:ff:00.0", sysfs_root=str(root)).dir is None       # an address no card has: no sensor rather than a neighbour's
    assert bench.GpuTelemetry(0, None, sysfs_root=str(root)).card.startswith("card0")     # no address known: by index, and the line says so


def test_host_discretisers_known_answers():
    """The host mirrors of SURVEY.md section 8 row a16 against answers derived by hand from the reference's lines (no RNG, no oracle):
    discretize_bayes.m:17-21 through the C ABI (emgpu_discretize_bayes), the worked example of hierarchical_discretize.m:4-8,
    hierarchical_cutpoints.m:5-15 and aind2sub.m:8-18 (the inverse of asub2ind.m:13-14)."""
    from em_model_manned_bayes_amd import functions as F
    lib = L.lib()
    lib.emgpu_discretize_bayes.restype = C.c_int32
    th = np.array([1.0, 2.0, 3.0])

    def d(x, t=th):
        return int(lib.emgpu_discretize_bayes(C.c_double(x), t.ctypes.data_as(C.c_void_p), C.c_int32(t.size)))
    # :17-18 `x >= thresholds(end)` -> numel + 1; :20 `find(x < thresholds, 1)`: a value ON a cut point belongs to the bin above it
    assert [d(-np.inf), d(0.5), d(1.0), d(1.5), d(2.0), d(2.999999), d(3.0), d(10.0), d(np.inf)] == [1, 1, 2, 2, 3, 3, 4, 4, 4]
    one = np.array([5.0])
    assert [d(4.0, one), d(5.0, one), d(6.0, one)] == [1, 2, 2]
    assert np.array_equal(F.discretize_bayes([0.5, 1.0, 3.0], th), [1, 2, 4]) and F.discretize_bayes(2.5, th) == 3.0
    bearing = np.arange(10.0, 360.0, 10.0)                       # the 36-bin grids of the terminal model: 35 cut points
    assert [d(0.0, bearing), d(9.99, bearing), d(10.0, bearing), d(349.99, bearing), d(350.0, bearing), d(360.0, bearing)] == [1, 1, 2, 35, 36, 36]

    # hierarchical_cutpoints.m:9-14: thresholds = [60 80 100 120 140 160 180], fine{i} = a + (1:n-1) * (b - a) / n
    coarse = np.arange(80.0, 161.0, 20.0)
    fine = F.hierarchical_cutpoints(coarse, [60.0, 180.0], 4)
    assert len(fine) == 6
    for i, a in enumerate([60.0, 80.0, 100.0, 120.0, 140.0, 160.0]):
        assert np.array_equal(fine[i], [a + 5.0, a + 10.0, a + 15.0])
    assert np.array_equal(F.hierarchical_cutpoints([0.0], [-1.0, 1.0], 2)[0], [-0.5]) and np.array_equal(F.hierarchical_cutpoints([0.0], [-1.0, 1.0], 2)[1], [0.5])
    assert all(c.size == 0 for c in F.hierarchical_cutpoints([1.0, 2.0], [0.0, 3.0], 1))      # n = 1: (1:0) is empty

    # hierarchical_discretize.m:4-8 -- by hand from :25-48: d = coarse bins; fine bins f = [2 1 1 1 1 3 3 4];
    # neighbours in the same coarse bin: (3,3) x 3 with equal f -> repeat 3; (1,1) f 3,3 -> repeat 4; (1,1) f 3,4 -> change 1
    x = [65, 100, 100, 100, 100, 72, 71, 78]
    dd, repeat, change = F.hierarchical_discretize(x, coarse, fine)
    assert np.array_equal(dd, [1, 3, 3, 3, 3, 1, 1, 1]) and (repeat, change) == (4, 1)
    dd, repeat, change = F.hierarchical_discretize(x, coarse, fine, zero_bins=[3])            # :41 `~any(zero_bins == d(ii))`: bin 3's pairs do not count
    assert np.array_equal(dd, [1, 3, 3, 3, 3, 1, 1, 1]) and (repeat, change) == (1, 1)
    dd, repeat, change = F.hierarchical_discretize(x, coarse, [])                             # :16-21 no fine cut points
    assert np.array_equal(dd, [1, 3, 3, 3, 3, 1, 1, 1]) and (repeat, change) == (0, 0)
    dd, _, _ = F.hierarchical_discretize([170.0, 65.0], coarse[:-1], F.hierarchical_cutpoints(coarse[:-1], [60.0, 160.0], 4), wrap=1)
    assert np.array_equal(dd, [1, 1])                                                         # :27-30 wrap: bin 5 of 4 cut points -> 1

    # aind2sub.m:12-18 undoes asub2ind.m:13-14 (k = [1 4 12] for siz = [4 3 2])
    siz = [4, 3, 2]
    assert np.array_equal(F.aind2sub(siz, 1), [1, 1, 1]) and np.array_equal(F.aind2sub(siz, 10), [2, 3, 1])
    assert np.array_equal(F.aind2sub(siz, 13), [1, 1, 2]) and np.array_equal(F.aind2sub(siz, 24), [4, 3, 2])
    assert F.asub2ind(siz, [2, 3, 1]) == 10 and F.asub2ind(siz, [4, 3, 2]) == 24
    for ndx in range(1, 25):
        assert F.asub2ind(siz, F.aind2sub(siz, ndx)) == ndx
    assert np.array_equal(F.aind2sub([7], 5), [5])


def test_local_smooth_defaults_follow_the_reference_function_each_layer_mirrors():
    """one rule instead of per-layer habits -- a Python entry point smooths by default exactly when the reference function it
    mirrors does (createEncounter.m:88-89 is inside createEncounter, hence inside track.m; PropagateTrajectory, :93-265, has no smoothing)."""
    import inspect
    d = lambda f: inspect.signature(f).parameters["local_smooth"].default
    assert d(E.CorTerminalModel.createEncounter) is True and d(E.CorTerminalModel.track) is True and d(native.track_terminal_host) is True
    assert d(native.propagate_terminal_host) is False and d(native.propagate_terminal_joined_host) is False and d(native.terminal_sample_params) is False


def test_bench_box_state_rule():
    """bench.box_state: the telemetry rule that names a line's state (HISTORY.md section 7; round 5: the state belongs to the trace's
    placement) -- fast at the socket's power limit, slow at 1 27x W with the clock UP at 2 3xx MHz, neither for a kernel at the full clock."""
    sys.path.insert(0, ROOT)
    import bench
    tel = lambda w, mhz: {"socket_power_w": {"median": w}, "sclk_mhz": {"median": mhz}}
    assert bench.box_state(tel(1346.0, 2195.0)) == "fast" and bench.box_state(tel(1361.0, 2251.0)) == "fast"
    assert bench.box_state(tel(1270.0, 2330.0)) == "slow" and bench.box_state(tel(1256.0, 2349.0)) == "slow"
    assert bench.box_state(tel(1287.0, 2392.0)).startswith("below-the-power-limit")        # full clock: the kernel never reaches the limit
    assert bench.box_state(tel(1293.0, 2273.0)).startswith("below-the-power-limit")        # (between the two: 6.56 ms in profiles/r05_placement_probe.txt)
    assert bench.box_state(None).startswith("unknown") and bench.box_state({}).startswith("unknown")
