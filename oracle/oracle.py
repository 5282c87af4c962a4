"""oracle/oracle.py -- ctypes front-end of the CPU oracle (oracle/em_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.

Also holds an independent numpy parser of the reference's model .txt format
(em_read.m:47-141) so that the product's C++ loader can be checked against it.
"parity unpinned" against MATLAB: see the header of em_oracle.c.
"""
import ctypes as C
import heapq
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RNG_MT19937 = 0
RNG_PHILOX = 1


def build(force=False):
    if os.environ.get("EM_ORACLE_LIB"):   # another build of the same oracle (tools/check_philox10.sh: the 10-round generator)
        return os.environ["EM_ORACLE_LIB"]
    so = os.path.join(_HERE, "libem_oracle.so")
    src = os.path.join(_HERE, "em_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libem_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        _LIB = C.CDLL(so)
        _LIB.em_asub2ind.restype = C.c_int64
        _LIB.em_select_random_r.restype = C.c_int
        _LIB.em_uniform32.restype = C.c_double
        _LIB.em_uniform32.argtypes = [C.c_uint32]
        _LIB.em_dediscretize_u.restype = C.c_double
        _LIB.em_uncor_sample_batch.restype = C.c_int64
        _LIB.em_geom_sample_batch.restype = C.c_int64
        _LIB.em_dbn_sample_batch.restype = C.c_int64
    return _LIB


# ---------------------------------------------------------------------------
# bn_sort.m:17-20 -- toposort(digraph(G),'Order','stable').  ASSUMED semantics
# of 'stable' (unverifiable without MATLAB): lexicographically smallest
# topological order == Kahn's algorithm always taking the lowest-index ready node.
# ---------------------------------------------------------------------------
def bn_sort(G):
    G = np.asarray(G, dtype=bool)
    n = G.shape[0]
    indeg = G.sum(axis=0).astype(int)
    heap = [i for i in range(n) if indeg[i] == 0]
    heapq.heapify(heap)
    order = []
    while heap:
        i = heapq.heappop(heap)
        order.append(i + 1)
        for c in np.nonzero(G[i])[0]:
            indeg[c] -= 1
            if indeg[c] == 0:
                heapq.heappush(heap, int(c))
    if len(order) != n:
        raise ValueError("Network could not be hierarchically sorted")
    return np.array(order, dtype=np.int32)


def _extract_temporal_map(labels):
    # em_read.m:158-177
    tm = []
    for ii, lab in enumerate(labels):
        t = lab.find("(t)")
        if t >= 0:
            base = lab[: t + 1]
            fut = [k for k, l in enumerate(labels) if (base + "t+1)") in l]
            past = [k for k, l in enumerate(labels) if (base + "t-1)") in l]
            for k in fut:
                tm.append((ii + 1, k + 1))
            for k in past:
                tm.append((ii + 1, k + 1))
    return np.array(tm, dtype=np.int32).reshape(-1, 2)


def _extract_zero_bins(boundaries):
    # em_read.m:143-156: last j with b(j-1) < 0 < b(j)  ->  bin j-1
    out = []
    for b in boundaries:
        z = 0
        if len(b) > 2:
            for j in range(1, len(b)):
                if b[j - 1] < 0 and b[j] > 0:
                    z = j
        out.append(z)
    return np.array(out, dtype=np.int32)


def parse_model_txt(path, idx_zero_boundaries=(1, 2, 3), is_overwrite_zero_boundaries=False):
    """Independent numpy restatement of em_read.m:47-141.  Returns a dict."""
    with open(path, "r") as f:
        lines = [ln.strip("\r\n") for ln in f.read().split("\n")]
    lines = [ln for ln in lines if ln.strip() != ""]
    p = {}
    i = 0

    def nums(s):
        return np.array(s.split(), dtype=np.float64)

    while i < len(lines):
        ln = lines[i].strip()
        if not ln.startswith("#"):
            i += 1
            continue
        field = ln
        row = i + 1
        if field == "# labels_initial":
            p["labels_initial"] = [s.strip() for s in lines[row].split(",")]
            p["n_initial"] = len(p["labels_initial"])
        elif field == "# G_initial":
            n = p["n_initial"]
            p["G_initial"] = np.array([nums(lines[row + k]) for k in range(n)]) != 0
        elif field == "# r_initial":
            p["r_initial"] = nums(lines[row]).astype(np.int32)
        elif field == "# N_initial":
            p["_N_initial_flat"] = nums(lines[row])
        elif field == "# labels_transition":
            p["labels_transition"] = [s.strip() for s in lines[row].split(",")]
            p["n_transition"] = len(p["labels_transition"])
        elif field == "# G_transition":
            n = p["n_transition"]
            p["G_transition"] = np.array([nums(lines[row + k]) for k in range(n)]) != 0
        elif field == "# r_transition":
            p["r_transition"] = nums(lines[row]).astype(np.int32)
        elif field == "# N_transition":
            p["_N_transition_flat"] = nums(lines[row])
        elif field == "# boundaries":
            b = []
            for k in range(p["n_initial"]):
                s = lines[row + k].strip()
                b.append(np.zeros(0) if s == "*" else nums(s))
            p["boundaries"] = b
        elif field == "# resample_rates":
            p["resample_rates"] = nums(lines[row])
        else:
            raise ValueError("Unknown field: %s" % field)
        i += 1

    # array2cells / getdims (em_read.m:191-206)
    def cells(flat, G, r, vars_):
        out = {}
        idx = 0
        for v in vars_:
            q = int(np.prod(r[G[:, v]])) if G[:, v].any() else 1
            cnt = int(r[v]) * q
            out[v] = flat[idx: idx + cnt].reshape(q, int(r[v])).T.copy()  # column-major r x q
            idx += cnt
        assert idx == len(flat), (idx, len(flat))
        return out

    ni = p["n_initial"]
    p["N_initial"] = cells(p.pop("_N_initial_flat"), p["G_initial"], p["r_initial"], range(ni))
    p["order_initial"] = bn_sort(p["G_initial"])
    if "labels_transition" in p:
        nt = p["n_transition"]
        p["N_transition"] = cells(p.pop("_N_transition_flat"), p["G_transition"], p["r_transition"], range(ni, nt))
        p["order_transition"] = bn_sort(p["G_transition"])
        p["temporal_map"] = _extract_temporal_map(p["labels_transition"])
    else:
        p["n_transition"] = 0
        p["temporal_map"] = np.zeros((0, 2), dtype=np.int32)
    if "boundaries" in p:
        p["zero_bins"] = _extract_zero_bins(p["boundaries"])
        if is_overwrite_zero_boundaries:
            for k in idx_zero_boundaries:
                p["boundaries"][k - 1] = np.zeros(0)
    if "resample_rates" not in p:
        p["resample_rates"] = np.zeros(ni)
    return p


# ---------------------------------------------------------------------------
class _EmModel(C.Structure):
    _fields_ = [
        ("n_initial", C.c_int32), ("n_transition", C.c_int32), ("n_dyn", C.c_int32), ("_pad", C.c_int32),
        ("G_initial", C.c_void_p), ("G_transition", C.c_void_p),
        ("r_initial", C.c_void_p), ("r_transition", C.c_void_p),
        ("order_initial", C.c_void_p), ("order_transition", C.c_void_p),
        ("temporal_map", C.c_void_p),
        ("N_initial", C.c_void_p), ("A_initial", C.c_void_p), ("off_initial", C.c_void_p),
        ("N_transition", C.c_void_p), ("A_transition", C.c_void_p), ("off_transition", C.c_void_p),
        ("boundaries", C.c_void_p), ("bnd_off", C.c_void_p), ("bnd_len", C.c_void_p),
        ("zero_bins", C.c_void_p), ("resample_rates", C.c_void_p), ("start", C.c_void_p),
    ]


class _UncorOpts(C.Structure):
    _fields_ = [("idxL", C.c_int32), ("idxV", C.c_int32), ("idxDH", C.c_int32), ("is_quantize500", C.c_int32),
                ("layers", C.c_void_p), ("max_attempts", C.c_int32), ("per_step", C.c_int32)]


class _GeomOpts(C.Structure):
    _fields_ = [("bounds_sample", C.c_void_p), ("idx_own_speed", C.c_int32), ("idx_int_speed", C.c_int32),
                ("min1", C.c_double), ("max1", C.c_double), ("min2", C.c_double), ("max2", C.c_double),
                ("max_attempts", C.c_int32), ("_pad", C.c_int32)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """Flat-array view of a parsed model for em_oracle.c.

    parms: dict as returned by parse_model_txt (or the product's em_read).
    prior: 0 / float constant / 'dbe' (bn_dirichlet_prior.m:18-37), or
    alpha_initial / alpha_transition dicts {var0: r x q array} given explicitly.
    start: sequence of n_initial ints, 0/None = unset.
    """

    def __init__(self, parms, prior=0.0, start=None, alpha_initial=None, alpha_transition=None):
        p = parms
        ni = int(p["n_initial"])
        nt = int(p.get("n_transition", 0)) or 0
        self.parms = p
        self.n_initial, self.n_transition = ni, nt
        self._keep = []

        def keep(a):
            self._keep.append(a)
            return a

        def prior_of(N):
            if isinstance(prior, str):
                if prior.lower() != "dbe":
                    raise ValueError("prior:notdbe")
                return np.full(N.shape, 1.0 / (N.shape[0] * N.shape[1]))
            return np.full(N.shape, float(prior))

        Gi = keep(np.ascontiguousarray(np.asarray(p["G_initial"], dtype=np.uint8)))
        ri = keep(np.ascontiguousarray(np.asarray(p["r_initial"], dtype=np.int32)))
        oi = keep(np.ascontiguousarray(np.asarray(p["order_initial"], dtype=np.int32)))
        Nf, Af, off = [], [], []
        pos = 0
        for v in range(ni):
            N = np.asarray(p["N_initial"][v], dtype=np.float64)
            A = np.asarray(alpha_initial[v], dtype=np.float64) if alpha_initial is not None else prior_of(N)
            off.append(pos)
            Nf.append(N.T.reshape(-1))
            Af.append(A.T.reshape(-1))
            pos += N.size
        Ni = keep(np.ascontiguousarray(np.concatenate(Nf)))
        Ai = keep(np.ascontiguousarray(np.concatenate(Af)))
        offi = keep(np.array(off, dtype=np.int64))

        m = _EmModel()
        m.n_initial, m.n_transition = ni, nt
        m.G_initial, m.r_initial, m.order_initial = _ptr(Gi), _ptr(ri), _ptr(oi)
        m.N_initial, m.A_initial, m.off_initial = _ptr(Ni), _ptr(Ai), _ptr(offi)
        tm = np.asarray(p.get("temporal_map", np.zeros((0, 2))), dtype=np.int32).reshape(-1, 2)
        tm = keep(np.ascontiguousarray(tm))
        m.n_dyn = tm.shape[0]
        m.temporal_map = _ptr(tm)
        self.temporal_map = tm
        if nt > 0:
            Gt = keep(np.ascontiguousarray(np.asarray(p["G_transition"], dtype=np.uint8)))
            rt = keep(np.ascontiguousarray(np.asarray(p["r_transition"], dtype=np.int32)))
            ot = keep(np.ascontiguousarray(np.asarray(p["order_transition"], dtype=np.int32)))
            Nf, Af, off = [], [], []
            pos = 0
            for v in range(nt):
                if v in p["N_transition"]:
                    N = np.asarray(p["N_transition"][v], dtype=np.float64)
                    if alpha_transition is not None and v in alpha_transition and alpha_transition[v] is not None:
                        A = np.asarray(alpha_transition[v], dtype=np.float64)
                    else:
                        A = prior_of(N)
                    off.append(pos)
                    Nf.append(N.T.reshape(-1))
                    Af.append(A.T.reshape(-1))
                    pos += N.size
                else:
                    off.append(-1)
            Nt = keep(np.ascontiguousarray(np.concatenate(Nf)))
            At = keep(np.ascontiguousarray(np.concatenate(Af)))
            offt = keep(np.array(off, dtype=np.int64))
            m.G_transition, m.r_transition, m.order_transition = _ptr(Gt), _ptr(rt), _ptr(ot)
            m.N_transition, m.A_transition, m.off_transition = _ptr(Nt), _ptr(At), _ptr(offt)
        b = p.get("boundaries", [np.zeros(0)] * ni)
        blen = keep(np.array([len(x) for x in b], dtype=np.int32))
        boff = keep(np.concatenate([[0], np.cumsum(blen)[:-1]]).astype(np.int32))
        bflat = keep(np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in b] + [np.zeros(1)])))
        zb = keep(np.ascontiguousarray(np.asarray(p.get("zero_bins", np.zeros(ni)), dtype=np.int32)))
        rr = keep(np.ascontiguousarray(np.asarray(p.get("resample_rates", np.zeros(ni)), dtype=np.float64)))
        st = np.zeros(ni, dtype=np.int32)
        if start is not None:
            for k, s in enumerate(start):
                st[k] = 0 if (s is None or (isinstance(s, float) and np.isnan(s))) else int(s)
        st = keep(st)
        m.boundaries, m.bnd_off, m.bnd_len = _ptr(bflat), _ptr(boff), _ptr(blen)
        m.zero_bins, m.resample_rates, m.start = _ptr(zb), _ptr(rr), _ptr(st)
        self.c = m
        assert C.sizeof(_EmModel) == lib().em_sizeof_model()

    def is_dynvar_depend(self):
        return bool(lib().em_is_dynvar_depend(C.byref(self.c)))

    def label_index(self, name):
        """1-based index of a label such as 'v' (labels keep their quotes, UncorEncounterModel.m:225)."""
        q = '"%s"' % name
        labs = self.parms["labels_initial"]
        return labs.index(q) + 1 if q in labs else 0


def uncor_sample(om, n, T, seed, mode=RNG_PHILOX, first_index=0, per_step=False,
                 is_quantize500=False, layers=None, max_attempts=1000,
                 want_events=True, want_dense=True, ev_cap=None, reject=True):
    """UncorEncounterModel.sample restated (UncorEncounterModel.m:192-313), n samples.

    Returns dict: init_bin [n,ni] int32, init_val [n,ni] f64, events (list of
    [K,4] arrays: dt, var, value, bin), dense_bin [n,T,nd] u8, dense_val [n,T,nd] f64,
    attempts [n], n_draws.
    """
    L = lib()
    ni, nd = om.n_initial, om.temporal_map.shape[0]
    o = _UncorOpts()
    o.idxL, o.idxV, o.idxDH = om.label_index("L"), om.label_index("v"), om.label_index("\\dot h")
    if not reject:   # a plain loop of dbn_hierarchical_sample calls on one stream (no UncorEncounterModel.m:275 test)
        o.idxL = o.idxV = o.idxDH = 0
    o.is_quantize500 = int(bool(is_quantize500))
    lay = None
    if layers is not None:
        lay = np.ascontiguousarray(np.asarray(layers, dtype=np.float64).reshape(-1, 2))
        o.layers = _ptr(lay)
    o.max_attempts, o.per_step = int(max_attempts), int(bool(per_step))
    init_bin = np.zeros((n, ni), dtype=np.int32)
    init_val = np.zeros((n, ni), dtype=np.float64)
    attempts = np.zeros(n, dtype=np.int32)
    ndraw = C.c_uint64(0)
    if ev_cap is None:
        ev_cap = (ni + nd + 1) * T + 8
    if want_events:
        ev_dt = np.zeros((n, ev_cap)); ev_val = np.zeros((n, ev_cap))
        ev_var = np.zeros((n, ev_cap), dtype=np.int32); ev_bin = np.zeros((n, ev_cap), dtype=np.int32)
        ev_cnt = np.zeros(n, dtype=np.int32)
        evp = (_ptr(ev_dt), _ptr(ev_var), _ptr(ev_bin), _ptr(ev_val), _ptr(ev_cnt), C.c_int(ev_cap))
    else:
        evp = (None, None, None, None, None, C.c_int(0))
    if want_dense:
        dense_bin = np.zeros((n, T, nd), dtype=np.uint8)
        dense_val = np.zeros((n, T, nd), dtype=np.float64)
        dp = (_ptr(dense_bin), _ptr(dense_val))
    else:
        dp = (None, None)
    rc = L.em_uncor_sample_batch(C.byref(om.c), C.c_int(mode), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n),
                                 C.c_int(T), C.byref(o), _ptr(init_bin), _ptr(init_val), *evp, *dp,
                                 _ptr(attempts), C.byref(ndraw))
    if rc != 0:
        raise RuntimeError("em_uncor_sample_batch failed rc=%d" % rc)
    out = {"init_bin": init_bin, "init_val": init_val, "attempts": attempts, "n_draws": ndraw.value}
    if want_events:
        out["events"] = [np.stack([ev_dt[i, :k], ev_var[i, :k].astype(float), ev_val[i, :k], ev_bin[i, :k].astype(float)], axis=1)
                         for i, k in enumerate(ev_cnt)]
    if want_dense:
        out["dense_bin"], out["dense_val"] = dense_bin, dense_val
    return out


def uncor_sample_mt(om, n, T, seed, threads, first_index=0, per_step=False, max_attempts=1000):
    """Dense-only batch over `threads` OpenMP threads (Philox mode); returns (dense_bin, dense_val)."""
    L = lib()
    nd = om.temporal_map.shape[0]
    o = _UncorOpts()
    o.idxL, o.idxV, o.idxDH = om.label_index("L"), om.label_index("v"), om.label_index("\\dot h")
    o.max_attempts, o.per_step = int(max_attempts), int(bool(per_step))
    db = np.zeros((n, T, nd), dtype=np.uint8)
    dv = np.zeros((n, T, nd), dtype=np.float64)
    L.em_uncor_sample_batch_mt.restype = C.c_int64
    rc = L.em_uncor_sample_batch_mt(C.byref(om.c), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n), C.c_int(T),
                                    C.byref(o), C.c_int(int(threads)), _ptr(db), _ptr(dv), None, None)
    if rc != 0:
        raise RuntimeError("em_uncor_sample_batch_mt failed rc=%d" % rc)
    return db, dv


def uncor_sample_throughput_mt(om, n, T, seed, threads, first_index=0, per_step=False, max_attempts=1000, chunk=256):
    """em_uncor_sample_throughput_mt: the dense batch on `threads` threads with thread-private output blocks; returns the
    sum of all dense bins (equal to uncor_sample(...)['dense_bin'].sum() for the same range)."""
    L = lib()
    o = _UncorOpts()
    o.idxL, o.idxV, o.idxDH = om.label_index("L"), om.label_index("v"), om.label_index("\\dot h")
    o.max_attempts, o.per_step = int(max_attempts), int(bool(per_step))
    chk = C.c_uint64(0)
    L.em_uncor_sample_throughput_mt.restype = C.c_int64
    rc = L.em_uncor_sample_throughput_mt(C.byref(om.c), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n), C.c_int(T),
                                         C.byref(o), C.c_int(int(threads)), C.c_int64(int(chunk)), C.byref(chk))
    if rc != 0:
        raise RuntimeError("em_uncor_sample_throughput_mt failed rc=%d" % rc)
    return int(chk.value)


def dbn_sample(om, n, t_max, seed, mode=RNG_PHILOX, first_index=0, per_step=False):
    """dbn_sample.m restated: returns init_bin [n,ni], list of raw events [K,3] (dt, var, bin)."""
    L = lib()
    ni = om.n_initial
    cap = (ni + 1) * t_max + 8
    init_bin = np.zeros((n, ni), dtype=np.int32)
    ev_dt = np.zeros((n, cap)); ev_var = np.zeros((n, cap), dtype=np.int32); ev_bin = np.zeros((n, cap), dtype=np.int32)
    cnt = np.zeros(n, dtype=np.int32)
    rc = L.em_dbn_sample_batch(C.byref(om.c), C.c_int(mode), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n),
                               C.c_int(t_max), C.c_int(int(per_step)), _ptr(init_bin), _ptr(ev_dt), _ptr(ev_var), _ptr(ev_bin),
                               _ptr(cnt), C.c_int(cap))
    if rc != 0:
        raise RuntimeError("em_dbn_sample_batch failed rc=%d" % rc)
    ev = [np.stack([ev_dt[i, :k], ev_var[i, :k].astype(float), ev_bin[i, :k].astype(float)], axis=1) for i, k in enumerate(cnt)]
    return init_bin, ev


def geom_sample(om, n, seed, mode=RNG_PHILOX, first_index=0, bounds_sample=None,
                idx_own_speed=0, idx_int_speed=0, lim1=(0.0, np.inf), lim2=(0.0, np.inf), max_attempts=100000):
    """@CorTerminalModel/sample.m:29-77 restated."""
    L = lib()
    ni = om.n_initial
    o = _GeomOpts()
    bs = None
    if bounds_sample is not None:
        bs = np.ascontiguousarray(np.asarray(bounds_sample, dtype=np.float64).reshape(ni, 2))
        o.bounds_sample = _ptr(bs)
    o.idx_own_speed, o.idx_int_speed = int(idx_own_speed), int(idx_int_speed)
    o.min1, o.max1, o.min2, o.max2 = float(lim1[0]), float(lim1[1]), float(lim2[0]), float(lim2[1])
    o.max_attempts = int(max_attempts)
    ob = np.zeros((n, ni), dtype=np.int32); ov = np.zeros((n, ni)); att = np.zeros(n, dtype=np.int32)
    rc = L.em_geom_sample_batch(C.byref(om.c), C.c_int(mode), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n),
                                C.byref(o), _ptr(ob), _ptr(ov), _ptr(att))
    if rc != 0:
        raise RuntimeError("em_geom_sample_batch failed rc=%d" % rc)
    return ob, ov, att


def philox4x32_10(ctr, key):
    """Random123's default (10 rounds): the function the published known-answer vectors pin."""
    out = (C.c_uint32 * 4)()
    lib().em_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def philox4x32_r(ctr, key, rounds):
    out = (C.c_uint32 * 4)()
    lib().em_philox4x32_r((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), C.c_int(int(rounds)), out)
    return list(out)


def philox_rounds():
    """Rounds the oracle's generator runs (EM_PHILOX_ROUNDS; the kernels' EMGPU_PHILOX_ROUNDS must agree)."""
    return int(lib().em_philox_rounds())


def philox4x32_np(c0, c1, c2, c3, k0, k1, rounds):
    """Vectorised numpy Philox4x32-`rounds` (an independent restatement used by the statistical tests): arrays of uint32 counters."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(int(rounds)):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & M
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def mt_doubles(seed, n):
    out = np.zeros(n)
    lib().em_mt_doubles(C.c_uint32(seed), C.c_int(n), _ptr(out))
    return out


def events2samples(initial, events):
    """events2samples.m restated by em_oracle.c; events [K,3] (dt,var,value). Returns n x T."""
    initial = np.asarray(initial, dtype=np.float64)
    n = len(initial)

    class Ev(C.Structure):
        _fields_ = [("dt", C.c_double), ("var", C.c_int32), ("bin", C.c_int32), ("val", C.c_double),
                    ("kind", C.c_int32), ("atime", C.c_int32)]
    K = len(events)
    arr = (Ev * max(K, 1))()
    for i, e in enumerate(events):
        arr[i].dt, arr[i].var, arr[i].val = float(e[0]), int(e[1]), float(e[2])
    T = int(sum(e[0] for e in events))
    d = np.zeros((T, n))
    r = lib().em_events2samples(C.c_int(n), _ptr(initial), None, arr, C.c_int(K), _ptr(d), None, C.c_int(T))
    assert r == T, (r, T)
    return d.T.copy()


def events2controls(om, initial, events):
    """events2controls.m restated; returns [rows, 1+n_dyn]."""
    initial = np.asarray(initial, dtype=np.float64)

    class Ev(C.Structure):
        _fields_ = [("dt", C.c_double), ("var", C.c_int32), ("bin", C.c_int32), ("val", C.c_double),
                    ("kind", C.c_int32), ("atime", C.c_int32)]
    K = len(events)
    arr = (Ev * max(K, 1))()
    for i, e in enumerate(events):
        arr[i].dt, arr[i].var, arr[i].val = float(e[0]), int(e[1]), float(e[2])
    nd = om.temporal_map.shape[0]
    ctl = np.zeros((max(K, 1), 1 + nd))
    r = lib().em_events2controls(C.byref(om.c), _ptr(initial), arr, C.c_int(K), _ptr(ctl))
    return ctl[:r].copy()


class _DynLims(C.Structure):
    _fields_ = [("minVel_ft_s", C.c_double), ("maxVel_ft_s", C.c_double), ("maxTurnRate_deg_s", C.c_double),
                ("maxAltitude_ft", C.c_double), ("maxVertRate_ft_s", C.c_double)]


def propagate(oms, model_of, geo, seed, dyn_limits, mode=RNG_PHILOX, first_index=0, tmax_s=120.0, max_resample=100000, cap=None):
    """PropagateTrajectory restated (createEncounter.m:93-265) for 4 tracks per encounter.
    oms: list of OracleModel whose alpha_transition holds setTransitionPriors(...,1).
    Returns (out [4n, cap, 6] f64, rows [4n])."""
    L = lib()
    L.em_propagate_batch.restype = C.c_int64
    geo = np.ascontiguousarray(np.asarray(geo, dtype=np.float64).reshape(-1, 12))
    n = geo.shape[0]
    model_of = np.ascontiguousarray(np.asarray(model_of, dtype=np.int32).reshape(-1))
    cap = int(cap or (int(tmax_s) + 3))
    ptrs = (C.c_void_p * len(oms))(*[C.addressof(om.c) for om in oms])
    dl = (_DynLims * 2)()
    d = np.asarray(dyn_limits, dtype=np.float64).reshape(2, 5)
    for a in range(2):
        dl[a].minVel_ft_s, dl[a].maxVel_ft_s, dl[a].maxTurnRate_deg_s, dl[a].maxAltitude_ft, dl[a].maxVertRate_ft_s = [float(x) for x in d[a]]
    out = np.zeros((4 * n, cap, 6))
    rows = np.zeros(4 * n, dtype=np.int32)
    rc = L.em_propagate_batch(ptrs, _ptr(model_of), C.c_int(mode), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n),
                              _ptr(geo), dl, C.c_double(tmax_s), C.c_int(max_resample), _ptr(out), _ptr(rows), C.c_int(cap))
    if rc != 0:
        raise RuntimeError("em_propagate_batch failed rc=%d" % rc)
    return out, rows


def propagate_throughput_mt(oms, model_of, geo, seed, dyn_limits, threads, first_index=0, tmax_s=120.0, max_resample=100000, cap=None):
    """em_propagate_throughput_mt: the work of `propagate` on `threads` threads with thread-private track buffers (bench.py's CPU leg
    for config 5; Philox mode).  Returns the track rows produced."""
    L = lib()
    L.em_propagate_throughput_mt.restype = C.c_int64
    geo = np.ascontiguousarray(np.asarray(geo, dtype=np.float64).reshape(-1, 12))
    n = geo.shape[0]
    model_of = np.ascontiguousarray(np.asarray(model_of, dtype=np.int32).reshape(-1))
    cap = int(cap or (int(tmax_s) + 3))
    ptrs = (C.c_void_p * len(oms))(*[C.addressof(om.c) for om in oms])
    dl = (_DynLims * 2)()
    d = np.asarray(dyn_limits, dtype=np.float64).reshape(2, 5)
    for a in range(2):
        dl[a].minVel_ft_s, dl[a].maxVel_ft_s, dl[a].maxTurnRate_deg_s, dl[a].maxAltitude_ft, dl[a].maxVertRate_ft_s = [float(x) for x in d[a]]
    rc = L.em_propagate_throughput_mt(ptrs, _ptr(model_of), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n), _ptr(geo), dl,
                                      C.c_double(tmax_s), C.c_int(max_resample), C.c_int(int(threads)), C.c_int(cap))
    if rc < 0:
        raise RuntimeError("em_propagate_throughput_mt failed rc=%d" % rc)
    return int(rc)


def sample2track(alt0, speed0, updates, ur_speed, ur_vertrate, ur_heading, min_speed, max_speed):
    """sample2track.m:183-243 restated: (xyz [n, T+1, 3], flags [n], speed_minmax [n, 2])."""
    L = lib()
    L.em_sample2track_batch.restype = None
    alt0 = np.ascontiguousarray(alt0, dtype=np.float64)
    speed0 = np.ascontiguousarray(speed0, dtype=np.float64)
    updates = np.ascontiguousarray(updates, dtype=np.float64)
    n, T = updates.shape[0], updates.shape[1]
    xyz = np.zeros((n, T + 1, 3))
    flags = np.zeros(n, dtype=np.uint8)
    vmm = np.zeros((n, 2))
    L.em_sample2track_batch(C.c_int64(n), C.c_int(T), C.c_double(ur_speed), C.c_double(ur_vertrate), C.c_double(ur_heading),
                            C.c_double(min_speed), C.c_double(max_speed), _ptr(alt0), _ptr(speed0), _ptr(updates),
                            _ptr(xyz), _ptr(flags), _ptr(vmm))
    return xyz, flags, vmm


def stay_prior_alpha(parms, prior=1.0):
    """setTransitionPriors.m:12-33 through em_oracle.c, as {var0: r x q} for OracleModel(alpha_transition=...)."""
    L = lib()
    out = {}
    G = np.asarray(parms["G_transition"], dtype=bool)
    r = np.asarray(parms["r_transition"])
    for tm in np.asarray(parms["temporal_map"]).reshape(-1, 2):
        ii, jj = int(tm[1]) - 1, int(tm[0]) - 1
        if not G[:, ii].any():
            continue
        q = int(np.prod(r[G[:, ii]]))
        a = np.zeros(int(r[jj]) * q)
        L.em_transition_prior_node(C.c_int(int(r[jj])), C.c_int64(q), C.c_double(prior), _ptr(a))
        out[ii] = a.reshape(q, int(r[jj])).T.copy()
    return out


class _TrackVars(C.Structure):
    _fields_ = [("idxG", C.c_int32), ("idxA", C.c_int32), ("idxL", C.c_int32), ("idxV", C.c_int32), ("idxDV", C.c_int32),
                ("idxDH", C.c_int32), ("idxDPsi", C.c_int32), ("is_rotorcraft", C.c_int32)]


def _track_vars(om, is_rotorcraft=False):
    tv = _TrackVars()
    tv.idxG, tv.idxA, tv.idxL, tv.idxV = (om.label_index(s) for s in ("G", "A", "L", "v"))
    tv.idxDV, tv.idxDH, tv.idxDPsi = (om.label_index(s) for s in ("\\dot v", "\\dot h", "\\dot \\psi"))
    tv.is_rotorcraft = int(bool(is_rotorcraft))
    return tv


def uncor_dynamic_limits(om, initial, up_min, up_max, speed_min, speed_max, is_rotorcraft=False):
    """@UncorEncounterModel/getDynamicLimits.m restated: (minVel_ft_s, maxVel_ft_s, maxVertRate_ft_s)."""
    L = lib()
    L.em_uncor_dynamic_limits.restype = None
    tv = _track_vars(om, is_rotorcraft)
    iv = np.ascontiguousarray(initial, dtype=np.float64)
    out = np.zeros(3)
    L.em_uncor_dynamic_limits(C.byref(om.c), C.byref(tv), _ptr(iv), C.c_double(up_min), C.c_double(up_max),
                              C.c_double(speed_min), C.c_double(speed_max), _ptr(out))
    return out


def point_mass_dynamics(ic, ctrl, dyn):
    """The point-mass model that stands in for em-core's run_dynamics_fast (header of the f1 section of em_oracle.c).
    ctrl [T, 3] per whole second; returns (rows [10T+1, 8] = time north east up speed phi theta psi, minmax[5])."""
    L = lib()
    L.em_point_mass_dynamics.restype = None
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(-1, 3)
    T = ctrl.shape[0]
    out = np.zeros((10 * T + 1, 8)); mm = np.zeros(5)
    icv = np.ascontiguousarray(ic, dtype=np.float64); dy = np.ascontiguousarray(dyn, dtype=np.float64)
    L.em_point_mass_dynamics(_ptr(icv), _ptr(ctrl), C.c_int(T), _ptr(dy), _ptr(out), _ptr(mm))
    return out, mm


def uncor_track(om, n, T, seed, mode=RNG_PHILOX, first_index=0, is_quantize500=False, is_rotorcraft=False,
                max_track_attempts=200, max_attempts=1000, f32_inputs=True, want_tracks=True, margin_cap=64):
    """UncorEncounterModel.track restated (UncorEncounterModel.m:318-471, coordSys 'NEU') on the documented point-mass
    dynamics.  Returns dict: tracks [n, 10T+1, 8], limits [n, 3], attempts [n] (-1: cap hit)."""
    L = lib()
    L.em_uncor_track_batch.restype = C.c_int64
    o = _UncorOpts()
    o.idxL, o.idxV, o.idxDH = om.label_index("L"), om.label_index("v"), om.label_index("\\dot h")
    o.is_quantize500, o.max_attempts, o.per_step = int(bool(is_quantize500)), int(max_attempts), 0
    tv = _track_vars(om, is_rotorcraft)
    tracks = np.zeros((n, 10 * T + 1, 8)) if want_tracks else None
    limits = np.zeros((n, 3)); attempts = np.zeros(n, dtype=np.int32)
    margins = np.full((n, margin_cap), np.inf)   # [i, j]: how close attempt j + 1 of trajectory i came to deciding differently (em_note)
    rc = L.em_uncor_track_batch(C.byref(om.c), C.c_int(mode), C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n), C.c_int(T),
                                C.byref(o), C.byref(tv), C.c_int(max_track_attempts), C.c_int(int(f32_inputs)),
                                _ptr(tracks) if want_tracks else None, _ptr(limits), _ptr(attempts), _ptr(margins), C.c_int(margin_cap))
    if rc != 0:
        raise RuntimeError("em_uncor_track_batch failed rc=%d" % rc)
    return {"tracks": tracks, "limits": limits, "attempts": attempts, "margins": margins}


class _TTrackOpts(C.Structure):
    _fields_ = [("idx", C.c_int32 * 12), ("min_enc_time_s", C.c_double), ("thres_dist_ft", C.c_double), ("thres_alt_low_ft", C.c_double),
                ("thres_vertrate_ft_s", C.c_double), ("max_cum_turn_deg", C.c_double * 2), ("pitch_deg", C.c_double * 2),
                ("local_smooth", C.c_int32), ("pad", C.c_int32)]


def local_smooth(x, w):
    """The stand-in for em-core's local_smooth (createEncounter.m:88-89; UNPINNED): em_local_smooth of em_oracle.c."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    lib().em_local_smooth(_ptr(x), C.c_int(x.size), C.c_int(int(w)), _ptr(out))
    return out


def check_cum_turn(heading_deg, limit):
    """CorTerminalModel.CheckCumTurn restated (CorTerminalModel.m:135-185): True = reject."""
    h = np.ascontiguousarray(heading_deg, dtype=np.float64)
    return bool(lib().em_check_cum_turn(_ptr(h), C.c_int(h.size), C.c_double(limit)))


def terminal_filters(own, intr, own_intent, int_intent, dyn_limits, max_cum_turn_deg, pitch_deg, min_enc_time_s=30.0, thres_dist_ft=2.5 * 6076,
                     thres_alt_low_ft=750.0, thres_vertrate_ft_s=5.0):
    """The filters of track.m:79-145 on two time-ordered tracks given as rows [t_s x_nm y_nm z_ft heading_deg v_ft_s]:
    (is_good, meta = tcpa_s hmd_ft vmd_ft enc_time_s)."""
    L = lib()
    o = _TTrackOpts()
    o.min_enc_time_s, o.thres_dist_ft, o.thres_alt_low_ft, o.thres_vertrate_ft_s = min_enc_time_s, thres_dist_ft, thres_alt_low_ft, thres_vertrate_ft_s
    for a in range(2):
        o.max_cum_turn_deg[a], o.pitch_deg[a] = float(max_cum_turn_deg[a]), float(pitch_deg[a])
    d = np.asarray(dyn_limits, dtype=np.float64).reshape(2, 5)
    dl = (_DynLims * 2)()
    for a in range(2):
        dl[a].minVel_ft_s, dl[a].maxVel_ft_s, dl[a].maxTurnRate_deg_s, dl[a].maxAltitude_ft, dl[a].maxVertRate_ft_s = [float(x) for x in d[a]]
    own = np.ascontiguousarray(own, dtype=np.float64).reshape(-1, 6)
    intr = np.ascontiguousarray(intr, dtype=np.float64).reshape(-1, 6)
    meta = np.zeros(4)
    rc = L.em_terminal_filters_rows(_ptr(own), C.c_int(own.shape[0]), _ptr(intr), C.c_int(intr.shape[0]), C.c_int(int(own_intent)), C.c_int(int(int_intent)),
                                    dl, C.byref(o), _ptr(meta))
    if rc < 0:
        raise RuntimeError("em_terminal_filters_rows failed rc=%d" % rc)
    return bool(rc), meta


def terminal_track(gom, oms, n, seed, dyn_limits, max_cum_turn_deg, pitch_deg, first_index=0, tmax_s=120.0, min_enc_time_s=30.0,
                   thres_dist_ft=2.5 * 6076, thres_alt_low_ft=750.0, thres_vertrate_ft_s=5.0, bounds_sample=None, max_track_attempts=500,
                   max_attempts=100000, max_resample=100000, f32=True, margin_cap=None, local_smooth=False):
    """CorTerminalModel.track restated (track.m:45-150), Philox mode.  gom: geometry OracleModel; oms: the 10 trajectory OracleModels with the
    stay prior.  Returns dict like native.track_terminal_host."""
    L = lib()
    L.em_terminal_track_batch.restype = C.c_int64
    labels = [s.strip('"') for s in gom.parms["labels_initial"]]
    o = _TTrackOpts()
    for a, pre in enumerate(("own", "int")):
        for k, f in enumerate(("distance", "bearing", "alt", "speed", "heading", "intent")):
            o.idx[6 * a + k] = labels.index(pre + "_" + f) + 1
    o.min_enc_time_s, o.thres_dist_ft, o.thres_alt_low_ft, o.thres_vertrate_ft_s = min_enc_time_s, thres_dist_ft, thres_alt_low_ft, thres_vertrate_ft_s
    for a in range(2):
        o.max_cum_turn_deg[a], o.pitch_deg[a] = float(max_cum_turn_deg[a]), float(pitch_deg[a])
    o.local_smooth = int(bool(local_smooth))
    go = _GeomOpts()
    bs = None
    if bounds_sample is not None:
        bs = np.ascontiguousarray(np.asarray(bounds_sample, dtype=np.float64).reshape(gom.n_initial, 2))
        go.bounds_sample = _ptr(bs)
    d = np.asarray(dyn_limits, dtype=np.float64).reshape(2, 5)
    go.idx_own_speed, go.idx_int_speed = o.idx[3], o.idx[9]
    go.min1, go.max1, go.min2, go.max2 = d[0, 0], d[0, 1], d[1, 0], d[1, 1]
    go.max_attempts = int(max_attempts)
    dl = (_DynLims * 2)()
    for a in range(2):
        dl[a].minVel_ft_s, dl[a].maxVel_ft_s, dl[a].maxTurnRate_deg_s, dl[a].maxAltitude_ft, dl[a].maxVertRate_ft_s = [float(x) for x in d[a]]
    ptrs = (C.c_void_p * 11)(C.addressof(gom.c), *[C.addressof(om.c) for om in oms])
    ni, cap2 = gom.n_initial, 2 * (int(tmax_s) + 3)
    sample = np.zeros((n, ni)); traj = np.zeros((n, 2, cap2, 6)); ln = np.zeros((n, 2), dtype=np.int32)
    meta = np.zeros((n, 4)); att = np.zeros(n, dtype=np.int32)
    mcap = int(max_track_attempts if margin_cap is None else margin_cap)
    margins = np.full((n, mcap), np.inf)   # [i, j]: how close attempt j + 1 of encounter i came to deciding differently (em_note)
    rc = L.em_terminal_track_batch(ptrs, C.c_uint64(seed), C.c_uint64(first_index), C.c_int64(n), C.byref(go), dl, C.byref(o), C.c_double(tmax_s),
                                   C.c_int(max_resample), C.c_int(max_track_attempts), C.c_int(int(f32)), _ptr(sample), _ptr(traj), _ptr(ln), _ptr(meta),
                                   _ptr(att), C.c_int(cap2), _ptr(margins), C.c_int(mcap))
    if rc != 0:
        raise RuntimeError("em_terminal_track_batch failed rc=%d" % rc)
    return {"sample": sample, "traj": traj, "len": ln, "meta": meta, "attempts": att, "margins": margins}
