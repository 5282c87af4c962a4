"""Thin object layer over the C ABI: NativeModel (emgpu_model*), Context (emgpu_ctx*) and the
sampling calls with numpy (host) or raw device pointers (torch tensors' data_ptr()).
"""
import ctypes as C

import numpy as np

from . import _lib as L

EVENT_DTYPE = np.dtype([("dt", "<u2"), ("var", "u1"), ("bin", "u1"), ("value", "<f4")])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class NativeModel:
    """Owns an emgpu_model handle (what em_read.m returns plus priors and start)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._refresh()

    def _refresh(self):
        info = L.ModelInfo()
        L.check(L.lib().emgpu_model_info(self._h, C.byref(info)))
        self.info = info
        self.n_initial, self.n_transition, self.n_dyn = info.n_initial, info.n_transition, info.n_dyn
        self.is_dynvar_depend = bool(info.is_dynvar_depend)

    @classmethod
    def load_txt(cls, path, idx_zero_boundaries=(1, 2, 3), is_overwrite_zero_boundaries=False):
        idx = np.asarray(list(idx_zero_boundaries), dtype=np.int32)
        h = C.c_void_p()
        L.check(L.lib().emgpu_model_load_txt(str(path).encode(), _p(idx), len(idx), int(bool(is_overwrite_zero_boundaries)), C.byref(h)))
        return cls(h.value)

    def save_bin(self, path):
        """emgpu_model_save_bin: the parsed model + its compiled plan as one binary file."""
        L.check(L.lib().emgpu_model_save_bin(self._h, str(path).encode()))

    @classmethod
    def load_bin(cls, path):
        """emgpu_model_load_bin; raises EmgpuError(ERR_PARSE) for a file written by another build of the library."""
        h = C.c_void_p()
        L.check(L.lib().emgpu_model_load_bin(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load_cached(cls, path, idx_zero_boundaries=(1, 2, 3), is_overwrite_zero_boundaries=False, cache_dir=None):
        """load_txt through the binary cache: `<name>.z<idx>.o<flag>.emgpubin` next to the .txt (or in cache_dir) is read when it is at
        least as new as the .txt and was written by this build of the library; otherwise the .txt is parsed and the cache (re)written."""
        import os
        tag = ".z%s.o%d.emgpubin" % ("".join(str(int(i)) for i in idx_zero_boundaries), int(bool(is_overwrite_zero_boundaries)))
        base = os.path.join(cache_dir, os.path.basename(str(path))) if cache_dir else str(path)
        binp = base + tag
        try:
            if os.path.getmtime(binp) >= os.path.getmtime(str(path)):
                return cls.load_bin(binp)
        except (OSError, L.EmgpuError):
            pass
        m = cls.load_txt(path, idx_zero_boundaries, is_overwrite_zero_boundaries)
        try:
            m.save_bin(binp)
        except L.EmgpuError:
            pass   # a read-only directory: the cache is an optimisation
        return m

    @classmethod
    def from_arrays(cls, G_initial, r_initial, N_initial, G_transition=None, r_transition=None, N_transition=None,
                    temporal_map=None, boundaries=None, zero_bins=None, resample_rates=None,
                    labels_initial=None, labels_transition=None):
        """N_initial: list of r_i x q_i arrays (all nodes); N_transition: list/dict for nodes n_i+1..n_t."""
        ni = len(r_initial)
        d = L.ModelDesc()
        keep = []

        def k(a, dt):
            a = np.ascontiguousarray(np.asarray(a, dtype=dt))
            keep.append(a)
            return a
        Gi = k(np.asarray(G_initial) != 0, np.uint8)
        ri = k(r_initial, np.int32)
        Ni = k(np.concatenate([np.asarray(N, dtype=np.float64).T.reshape(-1) for N in N_initial]), np.float64)
        d.n_initial = ni
        d.G_initial, d.r_initial, d.N_initial, d.n_N_initial = _p(Gi), _p(ri), _p(Ni), Ni.size
        if G_transition is not None and len(r_transition) > 0:
            nt = len(r_transition)
            Gt = k(np.asarray(G_transition) != 0, np.uint8)
            rt = k(r_transition, np.int32)
            if isinstance(N_transition, dict):
                seq = [N_transition[v] for v in range(ni, nt)]
            else:
                seq = list(N_transition)
                if len(seq) == nt:
                    seq = seq[ni:]
            Nt = k(np.concatenate([np.asarray(N, dtype=np.float64).T.reshape(-1) for N in seq]), np.float64)
            d.n_transition = nt
            d.G_transition, d.r_transition, d.N_transition, d.n_N_transition = _p(Gt), _p(rt), _p(Nt), Nt.size
            if temporal_map is not None:
                tm = k(np.asarray(temporal_map).reshape(-1, 2), np.int32)
                d.temporal_map, d.n_dyn = _p(tm), tm.shape[0]
        if boundaries is not None:
            bl = k([len(b) for b in boundaries], np.int32)
            bf = k(np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1) for b in boundaries] + [np.zeros(1)]), np.float64)
            d.boundaries, d.bnd_len = _p(bf), _p(bl)
        if zero_bins is not None:
            zb = k([0 if (z is None or (hasattr(z, "__len__") and len(z) == 0)) else int(np.asarray(z).reshape(-1)[0]) for z in zero_bins], np.int32)
            d.zero_bins = _p(zb)
        if resample_rates is not None:
            rr = k(resample_rates, np.float64)
            d.resample_rates = _p(rr)
        if labels_initial:
            d.labels_initial = "\n".join(labels_initial).encode()
        if labels_transition:
            d.labels_transition = "\n".join(labels_transition).encode()
        h = C.c_void_p()
        L.check(L.lib().emgpu_model_from_arrays(C.byref(d), C.byref(h)))
        return cls(h.value)

    def __del__(self):
        try:
            if self._h:
                L.lib().emgpu_model_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- field access
    def get_i32(self, field):
        n = L.check(L.lib().emgpu_model_get_i32(self._h, field, None, 0))
        out = np.zeros(n, dtype=np.int32)
        L.check(L.lib().emgpu_model_get_i32(self._h, field, _p(out), n))
        return out

    def get_f64(self, field, node=0):
        n = L.check(L.lib().emgpu_model_get_f64(self._h, field, node, None, 0))
        out = np.zeros(n, dtype=np.float64)
        L.check(L.lib().emgpu_model_get_f64(self._h, field, node, _p(out), n))
        return out

    def get_labels(self, field):
        n = L.check(L.lib().emgpu_model_get_text(self._h, field, None, 0))
        buf = C.create_string_buffer(n)
        L.check(L.lib().emgpu_model_get_text(self._h, field, buf, n))
        s = buf.value.decode()
        return s.split("\n") if s else []

    def set_f64(self, field, node, values):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        L.check(L.lib().emgpu_model_set_f64(self._h, field, node, _p(v), v.size))

    def set_prior(self, prior):
        """EncounterModel.prior semantics: number or 'dbe' (bn_dirichlet_prior.m:18-37)."""
        if isinstance(prior, str):
            if prior.lower() != "dbe":
                raise L.EmgpuError(L.ERR_PRIOR, "Unknown prior of %s, if char expecting prior = 'dbe'" % prior)
            L.check(L.lib().emgpu_model_set_prior(self._h, 1, 0.0))
        elif isinstance(prior, (int, float, np.floating, np.integer)):
            L.check(L.lib().emgpu_model_set_prior(self._h, 0, float(prior)))
        else:
            raise L.EmgpuError(L.ERR_PRIOR, "Second argument must be a char or double")

    def set_transition_stay_prior(self, prior):
        L.check(L.lib().emgpu_model_set_transition_stay_prior(self._h, float(prior)))

    def start_log_weight(self):
        """log P(preset values of `start`) under the model: the importance weight of every sample drawn with it."""
        out = C.c_double(0.0)
        L.check(L.lib().emgpu_model_start_log_weight(self._h, C.byref(out)))
        return float(out.value)

    def set_zero_bins(self, zero_bins):
        zb = np.array([0 if (z is None or (hasattr(z, "__len__") and len(z) == 0)) else int(np.asarray(z).reshape(-1)[0]) for z in zero_bins],
                      dtype=np.int32)
        L.check(L.lib().emgpu_model_set_zero_bins(self._h, _p(zb), zb.size))

    def set_start(self, start):
        st = np.zeros(self.n_initial, dtype=np.int32)
        for i, s in enumerate(start):
            if s is None:
                continue
            a = np.asarray(s, dtype=np.float64).reshape(-1)
            if a.size == 0 or np.isnan(a[0]):
                continue
            st[i] = int(a[0])
        L.check(L.lib().emgpu_model_set_start(self._h, _p(st), st.size))


class Context:
    """One device + one stream (emgpu_ctx).  Raises EmgpuError(ERR_NO_DEVICE) without a GPU."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        L.check(L.lib().emgpu_ctx_create(int(device), C.byref(h)))
        self._h = h
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream_ptr):
        L.check(L.lib().emgpu_ctx_set_stream(self._h, C.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def sync(self):
        L.check(L.lib().emgpu_ctx_sync(self._h))

    def trim(self):
        """emgpu_ctx_trim: release the device scratch the host-pointer / .track entry points keep between calls (the tables stay)."""
        L.check(L.lib().emgpu_ctx_trim(self._h))

    def last_kernel(self):
        return L.lib().emgpu_last_kernel_name(self._h).decode()

    def host_stats(self):
        """emgpu_host_stats: the phases of the last sample_dbn_host call on this context, as a dict."""
        st = L.HostStats()
        L.check(L.lib().emgpu_host_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in L.HostStats._fields_}

    def device_alloc(self, nbytes):
        """emgpu_device_alloc: device memory from the allocator the traces come from; returns the address (free with device_free, or with the context)."""
        ptr = C.c_void_p()
        L.check(L.lib().emgpu_device_alloc(self._h, int(nbytes), C.byref(ptr)))
        return int(ptr.value)

    def device_free(self, addr):
        L.check(L.lib().emgpu_device_free(self._h, C.c_void_p(int(addr))))

    def pinned_empty(self, shape, dtype):
        """A numpy array over pinned host memory of this context's pool (emgpu_host_alloc): the copy engine writes the *_host entry
        points' outputs straight into it.  The block goes back to the pool when the array (and every view of it) is gone."""
        import weakref
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        ptr = C.c_void_p()
        L.check(L.lib().emgpu_host_alloc(self._h, max(nbytes, 1), C.byref(ptr)))
        buf = (C.c_char * max(nbytes, 1)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        weakref.finalize(buf, _host_free, self, ptr.value)   # (numpy keeps `buf` alive as the base of every view; the finalizer keeps the context alive)
        return a

    def last_launches(self):
        """Kernel launches of the last sample_dbn_*_device call on this context."""
        return int(L.lib().emgpu_last_launch_count(self._h))

    def __del__(self):
        try:
            if self._h:
                L.lib().emgpu_ctx_free(self._h)
                self._h = None
        except Exception:
            pass


def _host_free(ctx, addr):
    try:
        if ctx._h:
            L.lib().emgpu_host_free(ctx._h, C.c_void_p(addr))
    except Exception:
        pass


class Trace:
    """emgpu_trace: device memory for the outputs of sample_dbn_device, allocated and PLACED by the library (emgpu_trace_alloc times the
    caller's own launch on a few candidate allocations and keeps the fastest: profiles/r05_placement_probe.txt).  `ptrs()` are the keyword
    arguments of sample_dbn_device / sample_dbn_blocks_device; `report` says what was measured."""

    def __init__(self, ctx, model, params, want=L.TRACE_INIT | L.TRACE_DENSE, candidates=0):
        self._ctx = ctx
        h = C.c_void_p()
        L.check(L.lib().emgpu_trace_alloc(ctx._h, model._h, C.byref(params), int(want), int(candidates), C.byref(h)))
        self._h = h
        o = L.SampleOut()
        L.check(L.lib().emgpu_trace_out(self._h, C.byref(o)))
        self.out = o
        r = L.TraceReport()
        L.check(L.lib().emgpu_trace_report(self._h, C.byref(r)))
        self.ld, self.bytes = int(r.ld), int(r.bytes)
        self.report = {"candidates": int(r.candidates), "kept": int(r.kept), "reused": int(r.reused),
                       "ms": [round(float(r.ms[i]), 3) for i in range(min(int(r.candidates), 8))] if r.candidates > 1 else [],
                       "first_allocation_ms": round(float(r.first_allocation_ms), 3), "kept_ms": round(float(r.kept_ms), 3)}

    def ptrs(self):
        o = self.out
        return dict(init_bin=o.init_bin or 0, init_val=o.init_val or 0, dyn_bin=o.dyn_bin or 0, dyn_val=o.dyn_val or 0,
                    ev_count=o.ev_count or 0, events=o.events or 0, attempts=o.attempts or 0, ld=int(o.ld))

    def free(self):
        """Back to the context's pool (the next Trace it fits takes it without a new probe)."""
        if self._h:
            L.check(L.lib().emgpu_trace_free(self._ctx._h, self._h))
            self._h = None

    def __del__(self):
        try:
            if self._h and self._ctx._h:
                L.lib().emgpu_trace_free(self._ctx._h, self._h)
                self._h = None
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def all_device_contexts():
    """One default Context per visible device: pass the list as `ctx` to spread one call over every GPU."""
    return [default_context(d) for d in range(device_count())]


def make_params(n, sample_time, seed, first_index=0, transition_mode=L.TRANSITION_REFERENCE_AUTO, flags=0,
                max_attempts=1000, idx_L=0, idx_v=0, idx_dh=0, layers=None, event_cap=0, indices=None, start=None):
    """emgpu_sample_params.  indices: a numpy uint64 array (host calls) or a raw device pointer (device calls) of n
    global indices replacing first_index + i.  start: a start grid [n, n_initial] of preset bins (0 = unset), numpy (host calls) or a raw
    device pointer."""
    p = L.SampleParams()
    p.seed, p.first_index, p.n, p.sample_time = int(seed) & (2**64 - 1), int(first_index), int(n), int(sample_time)
    p.transition_mode, p.flags, p.max_attempts = int(transition_mode), int(flags), int(max_attempts)
    p.idx_L, p.idx_v, p.idx_dh = int(idx_L), int(idx_v), int(idx_dh)
    keep = None
    if layers is not None:
        keep = np.ascontiguousarray(np.asarray(layers, dtype=np.float64).reshape(-1, 2))
        p.layers, p.n_layers = _p(keep), keep.shape[0]
    p.event_cap = int(event_cap)
    if indices is not None:
        if isinstance(indices, int):
            p.indices = indices
        else:
            idx = np.ascontiguousarray(indices, dtype=np.uint64)
            assert idx.size == int(n)
            p.indices = idx.ctypes.data
            keep = (keep, idx)
    if start is not None:
        if isinstance(start, int):
            p.start = start
        else:
            st = np.ascontiguousarray(start, dtype=np.int32)
            assert st.ndim == 2 and st.shape[0] == int(n)
            p.start = st.ctypes.data
            keep = (keep, st)
    return p, keep


def device_count():
    c = C.c_int32(0)
    L.check(L.lib().emgpu_device_count(C.byref(c)))
    return int(c.value)


def shard_range(n_total, rank, world):
    """emgpu_shard_range: the split every sharded entry point uses (== sharding.shard_range)."""
    lo, hi = C.c_int64(0), C.c_int64(0)
    L.check(L.lib().emgpu_shard_range(int(n_total), int(rank), int(world), C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)


def _sample_out(init_bin=0, init_val=0, dyn_bin=0, dyn_val=0, ev_count=0, events=0, attempts=0, ld=0, col_offset=0):
    o = L.SampleOut()
    o.init_bin, o.init_val, o.dyn_bin, o.dyn_val = init_bin or None, init_val or None, dyn_bin or None, dyn_val or None
    o.ev_count, o.events, o.attempts = ev_count or None, events or None, attempts or None
    o.ld, o.col_offset = int(ld), int(col_offset)
    return o


def sample_dbn_device(ctx, model, params, init_bin=0, init_val=0, dyn_bin=0, dyn_val=0, ev_count=0, events=0, attempts=0,
                      ld=0, col_offset=0):
    """Asynchronous launch with raw device pointers (ints, 0 = skip).  ld / col_offset: write this call's
    trajectories into columns [col_offset, col_offset + n) of buffers dimensioned for ld trajectories."""
    o = _sample_out(init_bin, init_val, dyn_bin, dyn_val, ev_count, events, attempts, ld, col_offset)
    L.check(L.lib().emgpu_sample_dbn_device(ctx._h, model._h, C.byref(params), C.byref(o)))


def mixed_blocks(n_total, n_models, lo=0, hi=None):
    """emgpu_mixed_blocks: [(model, first_index, count)] of the equal-contiguous-block assignment inside [lo, hi)."""
    hi = n_total if hi is None else hi
    buf = (L.Block * max(1, int(n_models)))()
    k = L.check(L.lib().emgpu_mixed_blocks(int(n_total), int(n_models), int(lo), int(hi), buf))
    return [(int(buf[i].model), int(buf[i].first_index), int(buf[i].n)) for i in range(k)]


def sample_dbn_blocks_device(ctx, models, params, blocks, **ptrs):
    """emgpu_sample_dbn_blocks_device: one shared trace (device pointers in ptrs, as for sample_dbn_device) filled by
    blocks = [(model index, first_index, count)]; params.n / params.first_index describe the range the trace covers."""
    o = _sample_out(**ptrs)
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    arr = (L.Block * max(1, len(blocks)))()
    for i, (m, f, c) in enumerate(blocks):
        arr[i].model, arr[i].first_index, arr[i].n = int(m), int(f), int(c)
    L.check(L.lib().emgpu_sample_dbn_blocks_device(ctx._h, handles, len(models), C.byref(params), arr, len(blocks), C.byref(o)))


def sample_dbn_multi_device(ctxs, model, params, outs):
    """emgpu_sample_dbn_multi_device: outs = one dict of device pointers (sample_dbn_device keywords) per ctx, each
    holding that ctx's shard (native.shard_range(params.n, d, len(ctxs))).  Asynchronous: sync every ctx afterwards."""
    hs = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    arr = (L.SampleOut * len(ctxs))()
    for d, kw in enumerate(outs):
        o = _sample_out(**kw)
        for f, _ in L.SampleOut._fields_:
            setattr(arr[d], f, getattr(o, f))
    L.check(L.lib().emgpu_sample_dbn_multi_device(hs, len(ctxs), model._h, C.byref(params), arr))


def sample_dbn_host(ctx, model, n, sample_time, seed, want_dense=True, want_events=False, event_cap=None, want_log_weight=False, pinned=True,
                    raw=False, **kw):
    """Synchronous host-buffer call.  Returns a dict of numpy arrays in user-facing shapes:
    init_bin [n, n_i] u8, init_val [n, n_i] f32, dyn_bin [n, T, n_d] u8, dyn_val [n, T, n_d] f32,
    events: list of structured arrays (EVENT_DTYPE), attempts [n].
    ctx may be a list of Contexts (one per device): the batch is then split over them inside ONE library call
    (emgpu_sample_dbn_multi_host: one host thread + one stream per device) with identical results.
    pinned: the library's arrays come from the context's pinned pool (the copy engine writes straight into them); False: pageable numpy
    arrays, which the library fills through its own staging buffers -- what a caller's own arrays (MATLAB's, a C host's) get.
    raw: return the arrays in the library's layout (init_* [n_i, n], dyn_bin [G4, n_d, n] u32, dyn_val [G4, n_d, n, 4]) without the
    transposing copies; `host_stats` = emgpu_host_stats of the call either way.
    """
    ni, nd, T = model.n_initial, model.n_dyn, int(sample_time)
    if want_events and event_cap is None:
        event_cap = min((ni + nd + 1) * T + 2, 4096)
    p, keep = make_params(n, T, seed, event_cap=event_cap or 0, **kw)
    G4 = (T + 3) // 4
    ctx0 = ctx[0] if isinstance(ctx, (list, tuple)) else ctx
    empty = ctx0.pinned_empty if pinned else (lambda shape, dt: np.zeros(shape, dtype=dt))
    o = L.SampleOut()
    ib = empty((ni, n), np.uint8)
    iv = empty((ni, n), np.float32)
    att = empty((n,), np.int32)
    o.init_bin, o.init_val, o.attempts = _p(ib), _p(iv), _p(att)
    if want_dense and nd > 0:
        db = empty((G4, nd, n), np.uint32)
        dv = empty((G4, nd, n, 4), np.float32)
        o.dyn_bin, o.dyn_val = _p(db), _p(dv)
    if want_events:
        ec = empty((n,), np.uint32)
        ev = empty((n, event_cap), EVENT_DTYPE)
        o.ev_count, o.events = _p(ec), _p(ev)
    if want_log_weight:
        lw = np.zeros(n, dtype=np.float64)
        o.log_weight = _p(lw)
    if isinstance(ctx, (list, tuple)):
        hs = (C.c_void_p * len(ctx))(*[c._h for c in ctx])
        L.check(L.lib().emgpu_sample_dbn_multi_host(hs, len(ctx), model._h, C.byref(p), C.byref(o)))
    else:
        L.check(L.lib().emgpu_sample_dbn_host(ctx._h, model._h, C.byref(p), C.byref(o)))
    ctx = ctx0
    out = {"attempts": np.array(att), "kernel": ctx.last_kernel(), "host_stats": ctx.host_stats()}
    if raw:
        out["init_bin"], out["init_val"] = ib, iv
    else:
        out["init_bin"], out["init_val"] = ib.T.copy(), iv.T.copy()
    if want_log_weight:
        out["log_weight"] = lw
    if want_dense and nd > 0:
        out["dyn_bin"] = db if raw else unpack_dyn_bin(db, T)
        out["dyn_val"] = dv if raw else unpack_dyn_val(dv, T)
    if want_events:
        ec = np.array(ec)
        flat = ev[np.arange(ev.shape[1], dtype=np.uint32)[None, :] < ec[:, None]]   # all rows in order, one array (no per-sample concatenation; a copy)
        out["ev_count"] = ec
        out["events_flat"] = flat
        ends = np.cumsum(ec.astype(np.int64))
        out["events"] = split_rows(flat, ends)
    return out


def split_rows(a, ends):
    """[a[0:ends[0]], a[ends[0]:ends[1]], ...] as views -- what np.split(a, ends[:-1]) returns, without its per-piece swapaxes round trip
    (a million pieces: 1 us each instead of 3)."""
    e = ends.tolist()
    return [a[s:t] for s, t in zip([0] + e[:-1], e)]


def unpack_dyn_bin(db, T):
    """[G4][nd][n] uint32 (4 seconds per word) -> [n, T, nd] uint8."""
    G4, nd, n = db.shape
    b = db.view(np.uint8).reshape(G4, nd, n, 4)          # little endian: byte w = column 4g+w
    return np.ascontiguousarray(b.transpose(2, 0, 3, 1).reshape(n, G4 * 4, nd)[:, :T, :])


def unpack_dyn_val(dv, T):
    """[G4][nd][n][4] f32 -> [n, T, nd] f32."""
    G4, nd, n, _ = dv.shape
    return np.ascontiguousarray(dv.transpose(2, 0, 3, 1).reshape(n, G4 * 4, nd)[:, :T, :])


def sample_bn_host(ctx, model, n, seed, first_index=0, dediscretize=False, max_attempts=100000, bounds_sample=None,
                   idx_own_speed=0, idx_int_speed=0, lim1=(0.0, np.inf), lim2=(0.0, np.inf), start=None, want_log_weight=False):
    """bn_sample.m (dediscretize=False) or the CorTerminalModel geometry draw (sample.m:29-77).  start: a start grid [n, n_initial]
    (0 = unset) -- one row of presets per sample, ONE launch; want_log_weight: also return the per-sample log-weights (4th value)."""
    p = L.BnParams()
    p.seed, p.first_index, p.n = int(seed) & (2**64 - 1), int(first_index), int(n)
    p.flags = 0 if dediscretize else L.FLAG_NO_DEDISC
    p.max_attempts = int(max_attempts)
    bs = None
    if bounds_sample is not None:
        bs = np.ascontiguousarray(np.asarray(bounds_sample, dtype=np.float64).reshape(model.n_initial, 2))
        p.bounds_sample = _p(bs)
    p.idx_own_speed, p.idx_int_speed = int(idx_own_speed), int(idx_int_speed)
    p.min_vel1, p.max_vel1, p.min_vel2, p.max_vel2 = float(lim1[0]), float(lim1[1]), float(lim2[0]), float(lim2[1])
    ob = np.zeros((model.n_initial, n), dtype=np.uint8)
    ov = np.zeros((model.n_initial, n), dtype=np.float32)
    att = np.zeros(n, dtype=np.int32)
    st = lw = None
    if start is not None:
        st = np.ascontiguousarray(start, dtype=np.int32)
        assert st.shape == (n, model.n_initial)
        p.start = _p(st)
    if want_log_weight:
        lw = np.zeros(n, dtype=np.float64)
        p.log_weight = _p(lw)
    L.check(L.lib().emgpu_sample_bn_host(ctx._h, model._h, C.byref(p), _p(ob), _p(ov), _p(att)))
    if want_log_weight:
        return ob.T.copy(), ov.T.copy(), att, lw
    return ob.T.copy(), ov.T.copy(), att


def terminal_t0_row(cap):
    """EMGPU_TERMINAL_T0_ROW: the row of t = 0 in an aircraft's block of 2 * terminal_t0_row(cap) rows."""
    return (int(cap) + 7) & ~7


def propagate_terminal_joined_host(ctx, models, geo, model_of, seed, first_index=0, tmax_s=120.0, dyn_limits=None,
                                   max_resample=100000, cap=None, local_smooth=False):
    """emgpu_propagate_terminal_host: PropagateTrajectory for 4 tracks per encounter (createEncounter.m:52-72) in the library's
    own layout.  models: list of NativeModel (stay prior already applied); geo [n, 12]; model_of [n, 4].
    Returns (traj [2n, 2 C, 5] f32, C = terminal_t0_row(cap): the joined track of aircraft 2e + a, row C + t = second t, fields x_nm y_nm z_ft
    heading_deg v_ft_s; rows [4n]: rows of track 4e + 2a + backward, < 0 = failed).  Rows outside a track's span are 0."""
    geo = np.ascontiguousarray(np.asarray(geo, dtype=np.float64).reshape(-1, 12))
    n = geo.shape[0]
    model_of = np.ascontiguousarray(np.asarray(model_of, dtype=np.int32).reshape(-1))
    assert model_of.size == 4 * n
    cap = int(cap or (int(tmax_s) + 3))
    p = L.TermParams()
    p.seed, p.first_index, p.n, p.tmax_s = int(seed) & (2**64 - 1), int(first_index), n, float(tmax_s)
    p.max_resample, p.cap = int(max_resample), cap
    p.flags = L.FLAG_LOCAL_SMOOTH if local_smooth else 0
    dl = np.asarray(dyn_limits, dtype=np.float64).reshape(10)
    for i in range(10):
        p.dyn_limits[i] = float(dl[i])
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    traj = np.zeros((2 * n, 2 * terminal_t0_row(cap), 5), dtype=np.float32)
    rows = np.zeros(4 * n, dtype=np.int32)
    L.check(L.lib().emgpu_propagate_terminal_host(ctx._h, handles, len(models), C.byref(p), _p(geo), _p(model_of), _p(traj), _p(rows)))
    return traj, rows


def split_joined_tracks(traj, rows, cap):
    """The four PropagateTrajectory results of every encounter, as the reference's function returns them one by one
    (createEncounter.m:96, 162-167), cut out of the joined tracks: out [4n, cap, 6] f32 = t_s x_nm y_nm z_ft heading_deg v_ft_s of
    track 4e + 2a + backward, row r = second +-r (rows beyond rows[l] are 0)."""
    n4 = rows.size
    out = np.zeros((n4, cap, 6), dtype=np.float32)
    c0 = terminal_t0_row(cap)
    r = np.arange(cap)
    for d, sign in ((0, 1), (1, -1)):
        src = traj[:, c0 + sign * r, :]                      # [2n, cap, 5]
        lanes = np.arange(d, n4, 2)                            # lane 4e + 2a + d  <->  aircraft 2e + a
        keep = r[None, :] < np.abs(np.where(rows[lanes] < 0, -rows[lanes] - 1, rows[lanes]))[:, None]
        out[lanes, :, 1:] = np.where(keep[:, :, None], src, 0)
        out[lanes, :, 0] = np.where(keep, sign * r[None, :], 0)
    return out


def propagate_terminal_host(ctx, models, geo, model_of, seed, first_index=0, tmax_s=120.0, dyn_limits=None,
                            max_resample=100000, cap=None, local_smooth=False):
    """propagate_terminal_joined_host, returned per PropagateTrajectory call like the reference does:
    (out [4n, cap, 6] f32 as t_s x_nm y_nm z_ft heading_deg v_ft_s, rows [4n])."""
    cap = int(cap or (int(tmax_s) + 3))
    traj, rows = propagate_terminal_joined_host(ctx, models, geo, model_of, seed, first_index, tmax_s, dyn_limits, max_resample, cap, local_smooth)
    return split_joined_tracks(traj, rows, cap), rows


def utrack_params(model, n, sample_time, seed, first_index=0, is_quantize500=False, is_rotorcraft=False,
                  max_track_attempts=200, max_attempts=1000, record_stride=1):
    """emgpu_utrack_params with the variable ids looked up by label like UncorEncounterModel.m:385-391."""
    labels = model.get_labels(L.F_LABELS_INITIAL)

    def lab(name):
        q = '"%s"' % name
        return labels.index(q) + 1 if q in labels else 0
    p = L.UTrackParams()
    p.seed, p.first_index, p.n, p.sample_time = int(seed) & (2**64 - 1), int(first_index), int(n), int(sample_time)
    p.flags = L.FLAG_QUANTIZE500 if is_quantize500 else 0
    p.max_track_attempts, p.max_attempts, p.record_stride = int(max_track_attempts), int(max_attempts), int(record_stride)
    p.idx_G, p.idx_A, p.idx_L, p.idx_v = lab("G"), lab("A"), lab("L"), lab("v")
    p.idx_dv, p.idx_dh, p.idx_dpsi = lab("\\dot v"), lab("\\dot h"), lab("\\dot \\psi")
    p.is_rotorcraft = int(bool(is_rotorcraft))
    return p


def track_uncor_host(ctx, model, n, sample_time, seed, want_tracks=True, **kw):
    """emgpu_track_uncor_host: UncorEncounterModel.track on the GPU (sample -> point-mass dynamics -> rejection rounds).
    Returns dict: tracks [n, S, 8] (time north east up speed phi theta psi), limits [n, 3], attempts [n], kernel."""
    p = utrack_params(model, n, sample_time, seed, **kw)
    S = 10 * int(sample_time) // p.record_stride + 1
    tracks = np.zeros((n, S, 8)) if want_tracks else None
    limits = np.zeros((n, 3))
    attempts = np.zeros(n, dtype=np.int32)
    L.check(L.lib().emgpu_track_uncor_host(ctx._h, model._h, C.byref(p), _p(tracks), _p(limits), _p(attempts)))
    return {"tracks": tracks, "limits": limits, "attempts": attempts, "kernel": ctx.last_kernel()}


def uncor_dynamic_limits(model, initial, up_min, up_max, speed_min, speed_max, is_rotorcraft=False):
    """emgpu_uncor_dynamic_limits: getDynamicLimits.m for one trajectory (host only, no GPU needed)."""
    p = utrack_params(model, 1, 1, 0, is_rotorcraft=is_rotorcraft)
    iv = np.ascontiguousarray(initial, dtype=np.float64)
    out = np.zeros(3)
    L.check(L.lib().emgpu_uncor_dynamic_limits(model._h, C.byref(p), _p(iv), float(up_min), float(up_max), float(speed_min), float(speed_max), _p(out)))
    return out


TERMINAL_GEO_FIELDS = ("distance", "bearing", "alt", "speed", "heading", "intent")


def terminal_sample_params(geom_model, n, seed, dyn_limits, first_index=0, tmax_s=120.0, cap=None, bounds_sample=None,
                           max_attempts=100000, max_resample=100000, local_smooth=False):
    """emgpu_tsample_params for emgpu_sample_terminal_device; returns (params, keep-alive) -- the variable ids are looked up by label like
    @CorTerminalModel/sample.m:56-62 / createEncounter.m:45-49."""
    labels = [s.strip('"') for s in geom_model.get_labels(L.F_LABELS_INITIAL)]
    p = L.TSampleParams()
    p.seed, p.first_index, p.n, p.tmax_s = int(seed) & (2**64 - 1), int(first_index), int(n), float(tmax_s)
    p.max_resample, p.cap, p.max_attempts = int(max_resample), int(cap or (int(tmax_s) + 3)), int(max_attempts)
    p.flags = L.FLAG_LOCAL_SMOOTH if local_smooth else 0
    for i, v in enumerate(np.asarray(dyn_limits, dtype=np.float64).reshape(10)):
        p.dyn_limits[i] = float(v)
    bs = None
    if bounds_sample is not None:
        bs = np.ascontiguousarray(np.asarray(bounds_sample, dtype=np.float64).reshape(geom_model.n_initial, 2))
        p.bounds_sample = _p(bs)
    for a, pre in enumerate(("own", "int")):
        for k, f in enumerate(TERMINAL_GEO_FIELDS):
            p.idx[6 * a + k] = labels.index(pre + "_" + f) + 1
    return p, bs


def sample_terminal_device(ctx, geom_model, traj_models, p, geom_val, geo, model_of, traj, rows, geom_bin=0, attempts=0):
    """emgpu_sample_terminal_device: geometry draw + createEncounter inputs + PropagateTrajectory x 4, device pointers (integers)."""
    if len(traj_models) != 10:
        raise ValueError("traj_models: the 10 trajectory models in CorTerminalModel.m:84-100 order")
    handles = (C.c_void_p * 10)(*[m._h for m in traj_models])
    L.check(L.lib().emgpu_sample_terminal_device(ctx._h, geom_model._h, handles, 10, C.byref(p), C.c_void_p(geom_bin), C.c_void_p(geom_val),
                                                 C.c_void_p(geo), C.c_void_p(model_of), C.c_void_p(traj), C.c_void_p(rows), C.c_void_p(attempts)))


def track_terminal_host(ctx, geom_model, traj_models, n, seed, dyn_limits, max_cum_turn_deg, pitch_deg, first_index=0, tmax_s=120.0,
                        min_enc_time_s=30.0, thres_dist_ft=2.5 * 6076, thres_alt_low_ft=750.0, thres_vertrate_ft_s=300.0 / 60.0,
                        bounds_sample=None, max_track_attempts=500, max_attempts=100000, max_resample=100000, allow_cap=False, local_smooth=True):
    """emgpu_track_terminal_host: CorTerminalModel.track (track.m:45-150) on the GPU.  Returns dict: sample [n, n_i], traj [n, 2, cap2, 6]
    (t_s x_nm y_nm z_ft heading_deg v_ft_s, time-ordered), len [n, 2], meta [n, 4] (tcpa_s hmd_ft vmd_ft enc_time_s), attempts [n].
    local_smooth -- ONE rule for every Python layer: a function smooths by default exactly when the reference function it mirrors does.
    track.m calls createEncounter, whose lines 88-89 smooth speed and altitude, so this helper, CorTerminalModel.track and
    CorTerminalModel.createEncounter default to True; PropagateTrajectory (createEncounter.m:93-265) does not smooth, so
    propagate_terminal_host / _joined_host default to False.  The C ABI has no defaults (a zeroed `flags` field is off: set
    EMGPU_FLAG_LOCAL_SMOOTH).  The smoother itself is the library's documented stand-in for em-core's un-vendored local_smooth: UNPINNED."""
    labels = [s.strip('"') for s in geom_model.get_labels(L.F_LABELS_INITIAL)]
    p = L.TTrackParams()
    p.seed, p.first_index, p.n, p.tmax_s = int(seed) & (2**64 - 1), int(first_index), int(n), float(tmax_s)
    p.max_resample, p.max_track_attempts, p.max_attempts = int(max_resample), int(max_track_attempts), int(max_attempts)
    p.flags = L.FLAG_LOCAL_SMOOTH if local_smooth else 0
    for i, v in enumerate(np.asarray(dyn_limits, dtype=np.float64).reshape(10)):
        p.dyn_limits[i] = float(v)
    for a in range(2):
        p.max_cum_turn_deg[a], p.pitch_deg[a] = float(max_cum_turn_deg[a]), float(pitch_deg[a])
    p.min_enc_time_s, p.thres_dist_ft, p.thres_alt_low_ft, p.thres_vertrate_ft_s = float(min_enc_time_s), float(thres_dist_ft), float(thres_alt_low_ft), float(thres_vertrate_ft_s)
    bs = None
    if bounds_sample is not None:
        bs = np.ascontiguousarray(np.asarray(bounds_sample, dtype=np.float64).reshape(geom_model.n_initial, 2))
        p.bounds_sample = _p(bs)
    for a, pre in enumerate(("own", "int")):
        for k, f in enumerate(TERMINAL_GEO_FIELDS):
            p.idx[6 * a + k] = labels.index(pre + "_" + f) + 1
    ni, cap2 = geom_model.n_initial, 2 * (int(tmax_s) + 3)
    sample = np.zeros((n, ni)); traj = np.zeros((n, 2, cap2, 6)); ln = np.zeros((n, 2), dtype=np.int32)
    meta = np.zeros((n, 4)); att = np.zeros(n, dtype=np.int32)
    if len(traj_models) != 10:
        raise ValueError("traj_models: the 10 trajectory models in CorTerminalModel.m:84-100 order")
    handles = (C.c_void_p * len(traj_models))(*[m._h for m in traj_models])
    rc = L.lib().emgpu_track_terminal_host(ctx._h, geom_model._h, handles, len(traj_models), C.byref(p), _p(sample), _p(traj), cap2, _p(ln), _p(meta), _p(att))
    if not (allow_cap and rc == L.ERR_REJECT_CAP):   # the outputs of the encounters that were accepted are delivered either way (attempts -1 marks the rest)
        L.check(rc)
    return {"sample": sample, "traj": traj, "len": ln, "meta": meta, "attempts": att, "kernel": ctx.last_kernel()}


def track_params(n, T, ur_speed, ur_vertrate, ur_heading, min_speed, max_speed, nd=0, slot_vertrate=0, slot_acc=0, slot_turnrate=0):
    p = L.TrackParams()
    p.n, p.T, p.nd = int(n), int(T), int(nd)
    p.slot_vertrate, p.slot_acc, p.slot_turnrate = int(slot_vertrate), int(slot_acc), int(slot_turnrate)
    p.ur_speed, p.ur_vertrate, p.ur_heading = float(ur_speed), float(ur_vertrate), float(ur_heading)
    p.min_speed, p.max_speed = float(min_speed), float(max_speed)
    return p


def sample2track_host(ctx, alt0, speed0, updates, ur_speed, ur_vertrate, ur_heading, min_speed, max_speed):
    """emgpu_sample2track_host: sample2track.m:183-243 on the GPU for values parsed from the em_sample files.
    alt0, speed0 [n]; updates [n, T, 3] = vertical rate, acceleration, turn rate (model units).
    Returns (xyz [n, T+1, 3] f64 feet, flags [n] u8 (bit 0 CFIT, bit 1 speed), speed_minmax [n, 2])."""
    alt0 = np.ascontiguousarray(alt0, dtype=np.float64).reshape(-1)
    speed0 = np.ascontiguousarray(speed0, dtype=np.float64).reshape(-1)
    updates = np.ascontiguousarray(updates, dtype=np.float64)
    n, T = updates.shape[0], updates.shape[1]
    assert updates.shape == (n, T, 3) and alt0.size == n and speed0.size == n
    p = track_params(n, T, ur_speed, ur_vertrate, ur_heading, min_speed, max_speed)
    xyz = np.zeros((n, T + 1, 3))
    flags = np.zeros(n, dtype=np.uint8)
    vmm = np.zeros((n, 2))
    L.check(L.lib().emgpu_sample2track_host(ctx._h, C.byref(p), _p(alt0), _p(speed0), _p(updates), _p(xyz), _p(flags), _p(vmm)))
    return xyz, flags, vmm


def sample2track_device(ctx, params, alt0, speed0, dyn_val, xyz=0, flags=0, speed_minmax=0):
    """emgpu_sample2track_device with raw device pointers (ints, 0 = skip an output): consumes the sampler's
    device output in place (alt0 / speed0 = rows of init_val, dyn_val = the dense trace).  Asynchronous."""
    L.check(L.lib().emgpu_sample2track_device(ctx._h, C.byref(params), alt0, speed0, dyn_val, xyz or None, flags or None,
                                              speed_minmax or None))
