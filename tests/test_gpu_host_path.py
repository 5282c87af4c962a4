"""-m gpu: the entry points that own memory for the caller (round 6) -- the pipelined host path of emgpu_sample_dbn_host, the pinned pool,
and the trace pool with its placement probe -- through the C ABI, against the CPU oracle."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from em_model_manned_bayes_amd import native, _lib as L
from util import load_pair, uncor_indices, assert_uncor_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def small_chunks(monkeypatch):
    """Chunks of 1 024 trajectories (the minimum), so that a few thousand trajectories are a pipeline of several chunks."""
    monkeypatch.setenv("EMGPU_HOST_CHUNK_MB", "1")


@pytest.mark.parametrize("name", ["uncor_1200code_v2p1", "uncor_1200only_fwse_v1p2", "glider_v1", "cor_v1"])
@pytest.mark.parametrize("pinned", [True, False])
def test_host_pipeline_of_several_chunks_matches_oracle(name, pinned, gpu_ctx, model_dir, small_chunks):
    """n = 5 300 in chunks of 1 024: six chunks, the last one short; dense + event lists (packed on the device, unpacked on the host),
    pinned outputs (the copy engine writes into the caller's pitch) and pageable ones (staging + host threads)."""
    nm, pp, _ = load_pair(name, model_dir)
    n, T, seed, first = 5300, 240, 0x5EED0002, 2**34 + 5
    idx = uncor_indices(pp)
    ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, mode=O.RNG_PHILOX, first_index=first)
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=True, event_cap=512, pinned=pinned, **idx)
    st = got["host_stats"]
    assert st["chunks"] == 6 and st["chunk_n"] == 1024 and st["direct"] == int(pinned), st
    assert st["event_rows"] == sum(len(e) for e in ref["events"])          # only the rows there are crossed PCIe
    assert_uncor_parity(got, ref, T)
    # the lists alone, and the dense trace alone
    ev = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=False, want_events=True, event_cap=512, pinned=pinned, **idx)
    assert all(np.array_equal(a, b) for a, b in zip(ev["events"], got["events"]))
    de = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=False, pinned=pinned, **idx)
    assert np.array_equal(de["dyn_bin"], got["dyn_bin"]) and np.array_equal(de["dyn_val"], got["dyn_val"])
    assert de["host_stats"]["bytes_d2h"] >= n * (5 * nm.n_initial + 4 + 20 * ((T + 3) // 4) * nm.n_dyn)


def test_host_pipeline_fills_columns_of_a_larger_pageable_array(gpu_ctx, model_dir, small_chunks):
    """ld / col_offset through the chunked path: three calls fill one caller-owned trace (pageable numpy arrays, as a C or MATLAB host has
    them), columns outside the calls untouched; equal to one call for the whole range."""
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    T, seed, first, ld = 61, 99, 10**12, 9000
    ni, nd, G4 = nm.n_initial, nm.n_dyn, (T + 3) // 4
    ib = np.full((ni, ld), 0xEE, np.uint8); iv = np.full((ni, ld), -7.0, np.float32)
    db = np.full((G4, nd, ld), 0xDDDDDDDD, np.uint32); dv = np.full((G4, nd, ld, 4), -9.0, np.float32)
    ec = np.full(ld, 0xCCCCCCCC, np.uint32); cap = 96
    ev = np.zeros((ld, cap), native.EVENT_DTYPE); att = np.full(ld, -5, np.int32)
    parts = [(100, 2500), (2600, 3100), (5700, 1200)]            # (column, count): 3 + 4 + 2 chunks
    for col, cnt in parts:
        p, keep = native.make_params(cnt, T, seed, first_index=first + col, event_cap=cap, **idx)
        o = L.SampleOut()
        o.init_bin, o.init_val, o.dyn_bin, o.dyn_val = ib.ctypes.data, iv.ctypes.data, db.ctypes.data, dv.ctypes.data
        o.ev_count, o.events, o.attempts, o.ld, o.col_offset = ec.ctypes.data, ev.ctypes.data, att.ctypes.data, ld, col
        L.check(L.lib().emgpu_sample_dbn_host(gpu_ctx._h, nm._h, C.byref(p), C.byref(o)))
        assert gpu_ctx.host_stats()["direct"] == 0 and gpu_ctx.host_stats()["chunks"] == -(-cnt // 1024)
    lo, hi = 100, 6900
    one = native.sample_dbn_host(gpu_ctx, nm, hi - lo, T, seed, first_index=first + lo, want_dense=True, want_events=True, event_cap=cap, **idx)
    assert np.array_equal(ib[:, lo:hi].T, one["init_bin"]) and np.array_equal(iv[:, lo:hi].T, one["init_val"])
    assert np.array_equal(native.unpack_dyn_bin(db[:, :, lo:hi], T), one["dyn_bin"])
    assert np.array_equal(native.unpack_dyn_val(dv[:, :, lo:hi], T), one["dyn_val"])
    assert np.array_equal(ec[lo:hi], one["ev_count"]) and np.array_equal(att[lo:hi], one["attempts"])
    for i in range(0, hi - lo, 7):
        assert np.array_equal(ev[lo + i, : ec[lo + i]], one["events"][i])
    for a, fill in ((ib, 0xEE), (iv, -7.0), (db, 0xDDDDDDDD), (dv, -9.0)):
        assert (a[..., :lo] == fill).all() and (a[..., hi:ld] == fill).all() if a.ndim < 4 else ((a[:, :, :lo] == fill).all() and (a[:, :, hi:] == fill).all())
    assert (ec[:lo] == 0xCCCCCCCC).all() and (ec[hi:] == 0xCCCCCCCC).all() and (att[:lo] == -5).all() and (att[hi:] == -5).all()


def test_host_pipeline_with_an_index_list_and_a_start_grid(gpu_ctx, model_dir, small_chunks):
    """The caller's index list and start grid are uploaded once; every chunk reads its own rows."""
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 3000, 40, 4242
    rs = np.random.RandomState(5)
    ind = rs.permutation(50_000)[:n].astype(np.uint64) + 7_000_000
    full = native.sample_dbn_host(gpu_ctx, nm, 50_000, T, seed, first_index=7_000_000, want_dense=True, want_events=True, **idx)
    sub = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, indices=ind, want_dense=True, want_events=True, **idx)
    assert sub["host_stats"]["chunks"] == 3
    rows = (ind - 7_000_000).astype(np.int64)
    assert np.array_equal(sub["init_bin"], full["init_bin"][rows]) and np.array_equal(sub["dyn_val"], full["dyn_val"][rows])
    assert all(np.array_equal(sub["events"][q], full["events"][r]) for q, r in enumerate(rows))
    # a start grid: blocks of rows preset G = 1..4 or nothing, log-weights per sample; the presets against the oracle block by block
    per = n // 5
    start = np.zeros((n, nm.n_initial), np.int32)
    for k in range(1, 5):
        start[k * per: (k + 1) * per, 0] = k
    got = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, start=start, want_dense=True, want_events=False, want_log_weight=True, **idx)
    assert got["host_stats"]["chunks"] == 3
    for k in range(5):
        ref = O.uncor_sample(O.OracleModel(pp, start=[k, 0, 0, 0, 0, 0, 0]), per, T, seed, mode=O.RNG_PHILOX, first_index=k * per, want_events=False)
        sl = slice(k * per, (k + 1) * per)
        assert np.array_equal(got["init_bin"][sl], ref["init_bin"]) and np.array_equal(got["dyn_bin"][sl], ref["dense_bin"]), k
        assert np.array_equal(got["dyn_val"][sl], ref["dense_val"].astype(np.float32)), k
    one = native.sample_dbn_host(gpu_ctx, nm, n, T, seed, start=start, want_dense=True, want_events=False, want_log_weight=True, pinned=False, **idx)
    assert np.array_equal(one["init_bin"], got["init_bin"]) and np.array_equal(one["dyn_bin"], got["dyn_bin"]) and np.array_equal(one["log_weight"], got["log_weight"])
    assert (got["log_weight"][:per] == 0).all() and (got["log_weight"][per:] < 0).all()


def test_event_cap_overrun_is_reported_by_the_chunked_path(gpu_ctx, model_dir, small_chunks):
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    with pytest.raises(L.EmgpuError) as ei:
        native.sample_dbn_host(gpu_ctx, nm, 4000, 240, 3, want_dense=False, want_events=True, event_cap=20, **uncor_indices(pp))
    assert ei.value.code == L.ERR_EVENT_CAP


def test_pinned_pool_hands_blocks_back_and_forth(gpu_ctx):
    a = gpu_ctx.pinned_empty((1000, 3), np.float32)
    addr = a.ctypes.data
    a[:] = 1.5
    assert a.sum() == 4500.0
    del a
    b = gpu_ctx.pinned_empty((3000,), np.float32)     # the same size: the same block
    assert b.ctypes.data == addr
    c = gpu_ctx.pinned_empty((3000,), np.float32)     # b is still alive: another block
    assert c.ctypes.data != addr
    bogus = C.c_void_p(12345)
    assert L.lib().emgpu_host_free(gpu_ctx._h, bogus) == L.ERR_ARG


def test_trace_alloc_report_pool_and_use(model_dir):
    """emgpu_trace_alloc: a small trace is one allocation, nothing timed; explicit candidates are timed with the caller's own launch; a freed
    trace comes back from the pool without a probe; the trace is what emgpu_sample_dbn_device writes (parity with the host path)."""
    ctx = native.Context(0)
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    n, T, seed = 70_000, 240, 0x5EED0002
    p, _ = native.make_params(n, T, seed, **idx)
    t = native.Trace(ctx, nm, p)                        # automatic: 254 MB < 1 GiB -> one block, no probe
    assert t.report["candidates"] == 1 and t.report["reused"] == 0 and t.report["ms"] == [] and t.ld == 70_656
    addr = t.ptrs()["dyn_val"]
    t.free()
    t3 = native.Trace(ctx, nm, p, candidates=3)         # the pool's block is candidate 0 (never probed), two more beside it
    r = t3.report
    assert r["candidates"] == 3 and len(r["ms"]) == 3 and min(r["ms"]) > 0 and r["kept_ms"] == min(r["ms"]) and r["first_allocation_ms"] == r["ms"][0], r
    kept_addr = t3.ptrs()["dyn_val"]
    assert (kept_addr == addr) == (r["kept"] == 0)
    native.sample_dbn_device(ctx, nm, p, **{k: v for k, v in t3.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")})
    ctx.sync()
    # read the trace back through hipMemcpy (the runtime libemgpu.so is linked against) and compare with the host path
    hip = C.CDLL(None)
    ni, nd, G4, ld = nm.n_initial, nm.n_dyn, T // 4, t3.ld
    dv = np.empty((G4, nd, ld, 4), np.float32); db = np.empty((G4, nd, ld), np.uint32)
    for arr, key in ((dv, "dyn_val"), (db, "dyn_bin")):
        assert hip.hipMemcpy(C.c_void_p(arr.ctypes.data), C.c_void_p(t3.ptrs()[key]), C.c_size_t(arr.nbytes), 2) == 0
    one = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=True, want_events=False, **idx)
    assert np.array_equal(native.unpack_dyn_val(dv[:, :, :n], T), one["dyn_val"]) and np.array_equal(native.unpack_dyn_bin(db[:, :, :n], T), one["dyn_bin"])
    t3.free()
    again = native.Trace(ctx, nm, p, candidates=3)      # placed already: reused, nothing timed
    assert again.report["reused"] == 1 and again.report["candidates"] == 1 and again.ptrs()["dyn_val"] == kept_addr and again.report["kept_ms"] == r["kept_ms"]
    again.free()
    ctx.trim()                                          # the pool is given back
    fresh = native.Trace(ctx, nm, p, candidates=1)
    assert fresh.report["reused"] == 0
    # events + attempts in a trace
    pe, _ = native.make_params(5000, 60, seed, event_cap=64, **idx)
    te = native.Trace(ctx, nm, pe, want=L.TRACE_EVENTS | L.TRACE_ATTEMPTS)
    ptr = te.ptrs()
    assert ptr["events"] and ptr["ev_count"] and ptr["attempts"] and not ptr["dyn_val"] and te.ld == 5120
    # ... filled by the device entry point and read back: the lists and attempts of the host path
    native.sample_dbn_device(ctx, nm, pe, ev_count=ptr["ev_count"], events=ptr["events"], attempts=ptr["attempts"], ld=te.ld)
    ctx.sync()
    cnt = np.empty(te.ld, np.uint32); evs = np.empty((te.ld, 64), native.EVENT_DTYPE); att = np.empty(te.ld, np.int32)
    for arr, key in ((cnt, "ev_count"), (evs, "events"), (att, "attempts")):
        assert hip.hipMemcpy(C.c_void_p(arr.ctypes.data), C.c_void_p(ptr[key]), C.c_size_t(arr.nbytes), 2) == 0
    ref = native.sample_dbn_host(ctx, nm, 5000, 60, seed, want_dense=False, want_events=True, event_cap=64, **idx)
    assert np.array_equal(cnt[:5000], ref["ev_count"]) and np.array_equal(att[:5000], ref["attempts"])
    assert all(np.array_equal(evs[i, : cnt[i]], ref["events"][i]) for i in range(0, 5000, 11))
    with pytest.raises(L.EmgpuError):
        native.Trace(ctx, nm, p, want=L.TRACE_EVENTS)   # event_cap 0


PLACED_PROCESS = r"""
import ctypes as C, json, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from em_model_manned_bayes_amd import native, em_io, _lib as L
import tempfile
path = em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp(prefix="emgpu_placed_"))
nm = native.NativeModel.load_txt(path)
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = {k: labels.index('"%%s"' %% v) + 1 for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
ctx = native.Context(0)
n, T = %(n)d, 240
p, _ = native.make_params(n, T, 0x5EED0002, **idx)
hip = C.CDLL(None)
ev = [C.c_void_p(), C.c_void_p()]
for e in ev: assert hip.hipEventCreate(C.byref(e)) == 0

def ms_per_launch(tr, warm=2, timed=5):      # HIP events on the NULL stream; the ctx launches there too (set_stream(0))
    best = 1e9
    for rnd in range(2):
        for k in range(warm): native.sample_dbn_device(ctx, nm, p, **ptrs(tr))
        hip.hipEventRecord(ev[0], None)
        for k in range(timed): native.sample_dbn_device(ctx, nm, p, **ptrs(tr))
        hip.hipEventRecord(ev[1], None); hip.hipEventSynchronize(ev[1])
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]); best = min(best, ms.value / timed)
    return best
def ptrs(tr): return {k: v for k, v in tr.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")}
ctx.set_stream(0)
placed = native.Trace(ctx, nm, p)            # automatic candidates: what a consumer of the C ABI calls instead of hipMalloc
others = [native.Trace(ctx, nm, p, candidates=1) for _ in range(5)]
t_end = time.time() + 0.5
while time.time() < t_end: ms_per_launch(others[-1], 0, 4)
res = {"placed_report": placed.report, "placed_ms": ms_per_launch(placed), "five_ms": [ms_per_launch(t) for t in others]}
res["placed_ms"] = min(res["placed_ms"], ms_per_launch(placed))
print("RESULT " + json.dumps(res))
"""


def test_two_consecutive_processes_both_get_a_fast_trace_through_the_c_abi():
    """two processes in a row on one box; each asks emgpu_trace_alloc for the headline's trace (10 M x 240 s, 36 GB;
    automatic candidates) and then allocates five more traces as they come (candidates = 1): the placed trace is written within 2 % of
    the best of the five.  Only the C ABI and the HIP runtime it is linked against are used (no torch)."""
    out = []
    for proc in range(2):
        r = subprocess.run([sys.executable, "-c", PLACED_PROCESS % dict(root=ROOT, n=10_000_000)], capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("RESULT ")][-1]
        out.append(json.loads(line[7:]))
    print(json.dumps(out))
    for res in out:
        assert res["placed_report"]["candidates"] == 6 and not res["placed_report"]["reused"], res
        assert res["placed_ms"] <= 1.02 * min(res["five_ms"]), res


def test_library_blocks_survive_copies_release_and_reuse(model_dir):
    """The sequence that made a HIP runtime segfault in hipMemMap (the 7.0 build PyTorch wheels bundle; HISTORY.md section 12.1): chunked blocks are the
    SOURCE of device -> host copies that span two physical chunks, are released, and a larger block is built afterwards.  The allocator keeps its address
    ranges, so the new block never overlaps a released one: the sequence runs, twice, and the trace built last holds what the sampler writes."""
    ctx = native.Context(0)
    nm, pp, _ = load_pair("uncor_1200code_v2p1", model_dir)
    idx = uncor_indices(pp)
    hip = C.CDLL(None)
    host = ctx.pinned_empty((1200 << 20,), np.uint8)
    n, T, seed = 320_000, 240, 77                     # 1.16 GB of trace: a chunked block (>= 1 GiB)
    p, _ = native.make_params(n, T, seed, **idx)
    for cycle in range(2):
        blocks = [ctx.device_alloc(1200 << 20) for _ in range(3)]
        for addr in blocks:                           # 1 200 MiB out of a block of two 1 GiB chunks: the copy spans both
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(addr), C.c_size_t(1200 << 20), 2) == 0
        for addr in blocks:
            ctx.device_free(addr)
        t = native.Trace(ctx, nm, p, candidates=1)
        assert t.bytes >= 1 << 30 and t.report["candidates"] == 1
        native.sample_dbn_device(ctx, nm, p, **{k: v for k, v in t.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")})
        ctx.sync()
        dv = np.empty((T // 4, nm.n_dyn, t.ld, 4), np.float32)
        assert hip.hipMemcpy(C.c_void_p(dv.ctypes.data), C.c_void_p(t.ptrs()["dyn_val"]), C.c_size_t(dv.nbytes), 2) == 0   # (a copy out of the trace too)
        t.free()
        ctx.trim()
    one = native.sample_dbn_host(ctx, nm, 5000, T, seed, want_dense=True, want_events=False, **idx)
    assert np.array_equal(native.unpack_dyn_val(dv[:, :, :5000], T), one["dyn_val"])
    with pytest.raises(L.EmgpuError):
        ctx.device_free(12345)                        # not a block of this context


@pytest.mark.parametrize("pinned", [True, False])
def test_multi_context_host_call_through_the_chunked_pipeline(pinned, model_dir, small_chunks):
    """emgpu_sample_dbn_multi_host (one host thread + one stream per context inside ONE call) with every context running its own pipeline of
    several chunks into its columns of the caller's arrays: three contexts (on this box's one device), 7 000 trajectories -> shards of 2 334 /
    2 333 / 2 333 in chunks of 1 024; equal to one single-context call, event lists included."""
    nm, pp, _ = load_pair("uncor_1200only_fwse_v1p2", model_dir)
    idx = uncor_indices(pp)
    n, T, seed, first = 7000, 120, 0xBEEF, 2**35
    ctxs = [native.Context(0) for _ in range(3)]
    got = native.sample_dbn_host(ctxs, nm, n, T, seed, first_index=first, want_dense=True, want_events=True, event_cap=300, pinned=pinned, **idx)
    one = native.sample_dbn_host(ctxs[0], nm, n, T, seed, first_index=first, want_dense=True, want_events=True, event_cap=300, pinned=pinned, **idx)
    for k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ev_count", "attempts"):
        assert np.array_equal(got[k], one[k]), k
    assert all(np.array_equal(a, b) for a, b in zip(got["events"], one["events"]))
    assert all(c.host_stats()["chunks"] == 3 for c in ctxs[1:]) and ctxs[1].host_stats()["direct"] == int(pinned)
    ref = O.uncor_sample(O.OracleModel(pp), 500, T, seed, mode=O.RNG_PHILOX, first_index=first + 4000)
    assert np.array_equal(got["dyn_bin"][4000:4500], ref["dense_bin"]) and np.array_equal(got["dyn_val"][4000:4500], ref["dense_val"].astype(np.float32))
