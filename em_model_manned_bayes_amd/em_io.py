"""Model file I/O on the Python side.

em_read   -- mirror of em_read.m:1-206: parses a model .txt through the native loader
             (emgpu_model_load_txt) and returns a Parms object with MATLAB's field names.
em_write  -- writes the same ASCII format (sections and layout of em_read.m:47-107); used by
             the synthetic-model generator and to materialise the packed models under models/.
load_npz / save_npz -- compact binary container of the same fields (the .txt files of the
             reference are up to 3.3 MB of decimal text; counts are stored as uint32).
"""
import json
import os

import numpy as np

from . import _lib as L
from .native import NativeModel


class Parms(dict):
    """MATLAB-struct-like dict (attribute access) as returned by em_read; `.native` is the handle."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def parms_from_native(m):
    """Read every em_read.m field back out of a native model."""
    p = Parms()
    ni, nt = m.n_initial, m.n_transition
    p["labels_initial"] = m.get_labels(L.F_LABELS_INITIAL)
    p["n_initial"] = ni
    p["G_initial"] = m.get_i32(L.F_G_INITIAL).reshape(ni, ni).astype(bool)
    p["order_initial"] = m.get_i32(L.F_ORDER_INITIAL)
    p["r_initial"] = m.get_i32(L.F_R_INITIAL)
    p["N_initial"] = []
    for v in range(ni):
        r = int(p["r_initial"][v])
        p["N_initial"].append(m.get_f64(L.F_N_INITIAL, v + 1).reshape(-1, r).T.copy())
    if nt > 0:
        p["labels_transition"] = m.get_labels(L.F_LABELS_TRANSITION)
        p["n_transition"] = nt
        p["G_transition"] = m.get_i32(L.F_G_TRANSITION).reshape(nt, nt).astype(bool)
        p["order_transition"] = m.get_i32(L.F_ORDER_TRANSITION)
        p["r_transition"] = m.get_i32(L.F_R_TRANSITION)
        p["N_transition"] = []
        for v in range(nt):
            r = int(p["r_transition"][v])
            a = m.get_f64(L.F_N_TRANSITION, v + 1)
            p["N_transition"].append(a.reshape(-1, r).T.copy() if a.size else np.zeros((0, 0)))
        p["temporal_map"] = m.get_i32(L.F_TEMPORAL_MAP).reshape(-1, 2)
    p["boundaries"] = [m.get_f64(L.F_BOUNDARIES, v + 1) for v in range(ni)]
    p["resample_rates"] = m.get_f64(L.F_RESAMPLE_RATES)
    zb = m.get_i32(L.F_ZERO_BINS)
    p["zero_bins"] = [([] if z == 0 else int(z)) for z in zb]
    # em_read.m:123-140
    bounds = np.zeros((ni, 2))
    cut = []
    for v in range(ni):
        b = p["boundaries"][v]
        if b.size == 0:
            cut.append(np.arange(2, int(p["r_initial"][v]) + 1, dtype=np.float64))
        else:
            bounds[v] = [b.min(), b.max()]
            cut.append(b[1:-1].copy())
    p["bounds_initial"] = bounds
    p["cutpoints_initial"] = cut
    p["native"] = m
    return p


def em_read(parameters_filename, idxZeroBoundaries=(1, 2, 3), isOverwriteZeroBoundaries=False):
    """parms = em_read(parameters_filename, 'idxZeroBoundaries', ..., 'isOverwriteZeroBoundaries', ...)
    (em_read.m:1,41-42)."""
    m = NativeModel.load_txt(parameters_filename, idxZeroBoundaries, isOverwriteZeroBoundaries)
    return parms_from_native(m)


def _fmt(x):
    x = float(x)
    if x == int(x) and abs(x) < 1e15:
        return "%d" % int(x)
    return repr(x)


def _line(values):
    return " ".join(_fmt(v) for v in values) + " \n"


def em_write(parms, path):
    """Write a model in the ASCII format em_read.m parses (fixed section order, one long line per
    N_*, trailing space before each newline like the shipped files)."""
    p = parms
    ni = int(p["n_initial"])
    with open(path, "w") as f:
        f.write("# labels_initial\n" + ", ".join(p["labels_initial"]) + " \n")
        f.write("# G_initial\n")
        for row in np.asarray(p["G_initial"]).astype(int):
            f.write(_line(row))
        f.write("# r_initial\n" + _line(p["r_initial"]))
        f.write("# N_initial\n")
        f.write(_line(np.concatenate([np.asarray(p["N_initial"][v], dtype=np.float64).T.reshape(-1) for v in range(ni)])))
        if p.get("n_transition", 0):
            nt = int(p["n_transition"])
            f.write("# labels_transition\n" + ", ".join(p["labels_transition"]) + " \n")
            f.write("# G_transition\n")
            for row in np.asarray(p["G_transition"]).astype(int):
                f.write(_line(row))
            f.write("# r_transition\n" + _line(p["r_transition"]))
            f.write("# N_transition\n")
            Nt = p["N_transition"]
            seq = [Nt[v] for v in range(ni, nt)]
            f.write(_line(np.concatenate([np.asarray(N, dtype=np.float64).T.reshape(-1) for N in seq])))
        if "boundaries" in p:
            f.write("# boundaries\n")
            for b in p["boundaries"]:
                f.write("* \n" if len(b) == 0 else _line(b))
        if "resample_rates" in p:
            f.write("# resample_rates\n" + _line(p["resample_rates"]))


def save_npz(parms, path):
    p = parms
    ni = int(p["n_initial"])
    nt = int(p.get("n_transition", 0) or 0)

    def pack(flat):
        flat = np.asarray(flat, dtype=np.float64)
        if flat.size and np.all(flat == np.floor(flat)) and flat.min() >= 0 and flat.max() < 2**32:
            return flat.astype(np.uint32)
        return flat
    d = {
        "meta": np.frombuffer(json.dumps({"labels_initial": list(p["labels_initial"]),
                                          "labels_transition": list(p.get("labels_transition", []))}).encode(), dtype=np.uint8),
        "G_initial": np.asarray(p["G_initial"]).astype(np.uint8),
        "r_initial": np.asarray(p["r_initial"], dtype=np.int32),
        "N_initial": pack(np.concatenate([np.asarray(p["N_initial"][v]).T.reshape(-1) for v in range(ni)])),
        "bnd_len": np.array([len(b) for b in p["boundaries"]], dtype=np.int32),
        "boundaries": np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1) for b in p["boundaries"]] + [np.zeros(0)]),
        "resample_rates": np.asarray(p["resample_rates"], dtype=np.float64),
    }
    if nt:
        d["G_transition"] = np.asarray(p["G_transition"]).astype(np.uint8)
        d["r_transition"] = np.asarray(p["r_transition"], dtype=np.int32)
        d["N_transition"] = pack(np.concatenate([np.asarray(p["N_transition"][v]).T.reshape(-1) for v in range(ni, nt)]))
    np.savez_compressed(path, **d)


def load_npz(path):
    """Packed model -> plain dict with em_write's fields (no native handle)."""
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    p = Parms()
    p["labels_initial"] = meta["labels_initial"]
    ni = len(p["labels_initial"])
    p["n_initial"] = ni
    p["G_initial"] = z["G_initial"].astype(bool)
    p["r_initial"] = z["r_initial"]

    def cells(flat, G, r, vars_):
        out, idx = {}, 0
        flat = flat.astype(np.float64)
        for v in vars_:
            q = int(np.prod(r[G[:, v]])) if G[:, v].any() else 1
            cnt = int(r[v]) * q
            out[v] = flat[idx: idx + cnt].reshape(q, int(r[v])).T.copy()
            idx += cnt
        assert idx == flat.size
        return out
    ci = cells(z["N_initial"], p["G_initial"], p["r_initial"], range(ni))
    p["N_initial"] = [ci[v] for v in range(ni)]
    if "G_transition" in z:
        p["labels_transition"] = meta["labels_transition"]
        nt = len(p["labels_transition"])
        p["n_transition"] = nt
        p["G_transition"] = z["G_transition"].astype(bool)
        p["r_transition"] = z["r_transition"]
        ct = cells(z["N_transition"], p["G_transition"], p["r_transition"], range(ni, nt))
        p["N_transition"] = [ct.get(v, np.zeros((0, 0))) for v in range(nt)]
    else:
        p["n_transition"] = 0
    bl = z["bnd_len"]
    off = np.concatenate([[0], np.cumsum(bl)])
    p["boundaries"] = [z["boundaries"][off[v]: off[v + 1]].copy() for v in range(ni)]
    p["resample_rates"] = z["resample_rates"]
    return p


MODELS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")


def materialize_model(name, out_dir):
    """Write models/<name>.npz as <out_dir>/<name>.txt (the reference's ASCII format); returns the path."""
    src = os.path.join(MODELS_DIR, name + ".npz")
    dst = os.path.join(out_dir, name + ".txt")
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        em_write(load_npz(src), dst)
    return dst
