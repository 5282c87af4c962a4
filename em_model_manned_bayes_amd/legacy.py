"""The reference's file pipeline: em_sample (model -> initial.txt / transition.txt) and sample2track
(those files -> 1 Hz dead-reckoning tracks as CSV), RUN_1_emsample.m / RUN_2_sample2track.m.

Same names, arguments and file formats as code/matlab/em_sample.m and code/matlab/sample2track.m.
Sampling and the track integration run on the GPU (libemgpu: emgpu_sample_dbn_host,
emgpu_sample2track_host); parsing and formatting are host-side Python, as they are host-side MATLAB
in the reference.  A device-resident consumer that skips the text files exists as
native.sample2track_device (it reads the sampler's dense trace in place).
"""
import os
import re

import numpy as np

from . import native
from .em_io import em_read
from .functions import _model_of, _take, bn_dirichlet_prior

FT_PER_NM = 1852.0 / 0.3048          # unitsratio('ft', 'nm')


def _g(x):
    """fprintf('%g', x): C and Python agree, except for the spelling of non-finite values."""
    x = float(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    return "%g" % x


def em_sample(parameters_filename, initial_output_filename=None, transition_output_filename=None, num_initial_samples=100,
              num_transition_samples=60, start=None, isOverwriteZeroBoundaries=False, idxZeroBoundaries=(1, 2, 3),
              rng_seed=42, prior=0, ctx=None):
    """em_sample(parameters_filename, 'initial_output_filename', ..., 'num_initial_samples', 100,
    'num_transition_samples', 60, 'start', {}, 'rng_seed', 42)  (em_sample.m:1-104).

    Writes `id <labels_initial>` rows (dediscretised initial sample, %g) and `initial_id t <dynamic labels>`
    rows (the dense trace of the dynamic variables at t = 0 .. num_transition_samples-1).
    `prior`: em_sample.m:52 assigns the string 'constant', which bn_dirichlet_prior.m:28 rejects
    (prior:notdbe), so the reference as shipped stops there; the constant prior 0 is what its
    documentation describes and is the default here.  Returns (initial [n, n_initial], trace [n, T, n_dyn])."""
    out_dir = os.path.join(os.environ.get("AEM_DIR_BAYES", "."), "output")
    initial_output_filename = initial_output_filename or os.path.join(out_dir, "initial.txt")
    transition_output_filename = transition_output_filename or os.path.join(out_dir, "transition.txt")
    parms = em_read(parameters_filename, isOverwriteZeroBoundaries=isOverwriteZeroBoundaries, idxZeroBoundaries=list(idxZeroBoundaries))
    di = bn_dirichlet_prior(parms["N_initial"], prior)
    dt = bn_dirichlet_prior(parms["N_transition"], prior)
    m = _model_of(parms, di, dt, start)
    n, T = int(num_initial_samples), int(num_transition_samples)
    seed, first = _take(rng_seed, n)
    # dbn_hierarchical_sample + events2samples (em_sample.m:78-82): no rejection test, dense trace
    res = native.sample_dbn_host(ctx or native.default_context(), m, n, T, seed, first_index=first, want_dense=True,
                                 max_attempts=1)
    initial = res["init_val"].astype(np.float64)
    trace = res["dyn_val"].astype(np.float64)
    tm = np.asarray(parms["temporal_map"]).reshape(-1, 2)
    for f in (initial_output_filename, transition_output_filename):
        if os.path.dirname(f):
            os.makedirs(os.path.dirname(f), exist_ok=True)
    with open(initial_output_filename, "w", encoding="utf-8", newline="\n") as f:
        f.write("id " + "".join("%s " % s for s in parms["labels_initial"]) + "\n")             # :64-68
        for i in range(n):
            f.write("%d " % (i + 1) + " ".join(_g(v) for v in initial[i]) + "\n")                # :85-88
    with open(transition_output_filename, "w", encoding="utf-8", newline="\n") as f:
        f.write("initial_id t " + "".join("%s " % parms["labels_transition"][int(r[1]) - 1] for r in tm) + "\n")   # :71-75
        rows = []
        for i in range(n):
            for j in range(T):
                rows.append("%s %s " % (_g(i + 1), _g(j)) + " ".join(_g(v) for v in trace[i, j]) + "\n")        # :91-96
        f.write("".join(rows))
    return initial, trace


def make_valid_name(s):
    """matlab.lang.makeValidName for the label strings of the model files: white space is removed and the
    letter after it capitalised, other invalid characters become '_', a leading non-letter gets an 'x'."""
    s = s.strip()
    s = re.sub(r"\s+([a-z])", lambda mo: mo.group(1).upper(), s)
    s = re.sub(r"\s+", "", s)
    s = re.sub(r"[^A-Za-z0-9_]", "_", s)
    if not s or not s[0].isalpha():
        s = "x" + s
    return s


def _erase(s, chars=('"', "\\")):
    for c in chars:
        s = s.replace(c, "")
    return s


def _read_table(filename, ncol):
    """readtable(..., 'Delimiter', ' ', 'HeaderLines', 1): numeric rows, one header line skipped."""
    with open(filename, "r", encoding="utf-8") as f:
        f.readline()
        data = np.loadtxt(f, dtype=np.float64, ndmin=2)
    if data.size == 0:
        return np.zeros((0, ncol))
    if data.shape[1] != ncol:
        raise ValueError("%s: expected %d columns, found %d" % (filename, ncol, data.shape[1]))
    return data


def _matlab_round(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def sample2track(parameters_filename, initial_filename, transition_filename, num_max_tracks=10000, out_dir_parent=None,
                 label_initial_geographic="G", label_initial_airspace="A", label_initial_altitude="L", label_initial_speed="v",
                 label_initial_acceleration="dotV", label_initial_vertrate="dotH", label_initial_turnrate="dotPsi",
                 label_transition_speed="dotV_t_1_", label_transition_altitude="dotH_t_1_", label_transition_heading="dotPsi_t_1_",
                 isOverwriteZeroBoundaries=False, idxZeroBoundaries=(1, 2, 3), min_altitude_ft=0, rng_seed=42, isPlot=False,
                 write_files=True, verbose=True, ctx=None):
    """[is_good, T_initial] = sample2track(parameters_filename, initial_filename, transition_filename, ...)
    (sample2track.m:1-287): 1 Hz dead reckoning of every sampled trajectory (on the GPU), CFIT and speed
    rejection, one `BAYES_t<T>_id<i>_alt<z0>_speed<v0>.csv` per accepted track under
    out_dir_parent/[G<g>/A<a>/]<alt>ft/.  T_initial is returned as a dict of columns (units converted like
    sample2track.m:126-128).  When the initial file holds more than num_max_tracks rows the reference keeps
    randperm(rows, num_max_tracks) of MATLAB's stream; here numpy's RandomState(rng_seed) chooses them."""
    out_dir_parent = out_dir_parent or os.path.join(os.environ.get("AEM_DIR_BAYES", "."), "output", "tracks")
    parameters = em_read(parameters_filename, isOverwriteZeroBoundaries=isOverwriteZeroBoundaries, idxZeroBoundaries=list(idxZeroBoundaries))
    tm = np.asarray(parameters["temporal_map"]).reshape(-1, 2)
    labels_init = [make_valid_name(_erase(s)) for s in parameters["labels_initial"]]                      # :63
    labels_trans = [make_valid_name(_erase(parameters["labels_transition"][int(r[1]) - 1])) for r in tm]  # :64
    names_i = ["id"] + labels_init
    names_t = ["id", "t"] + labels_trans
    Ti = _read_table(initial_filename, len(names_i))
    Tt = _read_table(transition_filename, len(names_t))
    if Ti.shape[0] > num_max_tracks:                                                                       # :75-77
        keep = np.random.RandomState(int(rng_seed)).permutation(Ti.shape[0])[: int(num_max_tracks)]
        Ti = Ti[keep]
    num_tracks = Ti.shape[0]

    def col(names, label):
        return names.index(label) if label in names else None

    ci_geo, ci_air = col(names_i, label_initial_geographic), col(names_i, label_initial_airspace)
    ci_alt, ci_spd = col(names_i, label_initial_altitude), col(names_i, label_initial_speed)
    ci_acc, ci_vr = col(names_i, label_initial_acceleration), col(names_i, label_initial_vertrate)
    cu_acc, cu_vr, cu_tr = col(names_t, label_transition_speed), col(names_t, label_transition_altitude), col(names_t, label_transition_heading)
    for what, c in (("altitude", ci_alt), ("speed", ci_spd), ("transition speed", cu_acc), ("transition altitude", cu_vr),
                    ("transition heading", cu_tr)):
        if c is None:
            raise ValueError("sample2track: no %s column with the given label" % what)
    b_alt = np.asarray(parameters["boundaries"][labels_init.index(label_initial_altitude)], dtype=np.float64)
    b_spd = np.asarray(parameters["boundaries"][labels_init.index(label_initial_speed)], dtype=np.float64)
    min_alt, max_alt = float(b_alt[0]), float(b_alt[-1])                                                   # :100-101
    min_speed, max_speed = float(b_spd[0]), float(b_spd[-1])                                               # :104-105
    base = os.path.basename(parameters_filename)
    ur_speed, ur_vertrate, ur_heading = FT_PER_NM / 3600.0, 1.0 / 60.0, 1.0                                # :113-123 (both branches)
    is_uncor = "uncor_" in base

    # group the transition rows by id, in file order (:192-193)
    T_of = {}
    order = np.argsort(Tt[:, 0], kind="stable") if Tt.shape[0] else np.zeros(0, dtype=np.int64)
    ids_sorted = Tt[order, 0] if Tt.shape[0] else np.zeros(0)
    starts = np.flatnonzero(np.r_[True, ids_sorted[1:] != ids_sorted[:-1]]) if ids_sorted.size else np.zeros(0, dtype=np.int64)
    ends = np.r_[starts[1:], ids_sorted.size] if ids_sorted.size else np.zeros(0, dtype=np.int64)
    for s, e in zip(starts, ends):
        T_of[ids_sorted[s]] = order[s:e]
    lens = np.array([len(T_of.get(Ti[i, 0], ())) for i in range(num_tracks)], dtype=np.int64)

    xyz_all = [None] * num_tracks
    flags = np.zeros(num_tracks, dtype=np.uint8)
    vmm = np.zeros((num_tracks, 2))
    context = ctx or native.default_context()
    for T in np.unique(lens):                       # one launch per distinct track length
        sel = np.flatnonzero(lens == T)
        if T == 0:
            for i in sel:                           # no transition rows: the track is its initial point (:196 never runs)
                z0, v0 = Ti[i, ci_alt], Ti[i, ci_spd] * ur_speed
                xyz_all[i] = np.array([[0.0, 0.0, z0]])
                flags[i] = (1 if z0 < 0 else 0) | (2 if (v0 <= min_speed * ur_speed or v0 >= max_speed * ur_speed) else 0)
                vmm[i] = (v0, v0)
            continue
        upd = np.stack([Tt[T_of[Ti[i, 0]]][:, [cu_vr, cu_acc, cu_tr]] for i in sel])
        x, f, v = native.sample2track_host(context, Ti[sel, ci_alt], Ti[sel, ci_spd], upd, ur_speed, ur_vertrate, ur_heading,
                                           min_speed, max_speed)
        for q, i in enumerate(sel):
            xyz_all[i] = x[q]
        flags[sel], vmm[sel] = f, v
    is_good = flags == 0                                                                                   # :243

    # altitude directories, step 100 ft (:150-158)
    if min_alt % 100.0 != 0:
        Lgrid = np.arange(np.floor(min_alt - 50.0), max_alt + 200.0 + 1e-9, 100.0)
    else:
        Lgrid = np.arange(min_alt, max_alt + 200.0 + 1e-9, 100.0)
    if Lgrid[0] < 0:
        Lgrid[0] = 0.0
    is_geo, is_air = ci_geo is not None, ci_air is not None
    if write_files:
        os.makedirs(out_dir_parent, exist_ok=True)
        if is_uncor and is_geo and is_air:                                                                # :161-173
            for g in np.unique(Ti[:, ci_geo]):
                for a in np.unique(Ti[:, ci_air]):
                    for l in Lgrid:
                        os.makedirs(os.path.join(out_dir_parent, "G%i" % g, "A%i" % a, "%ift" % l), exist_ok=True)
        else:                                                                                              # :175-178
            for l in Lgrid:
                os.makedirs(os.path.join(out_dir_parent, "%ift" % l), exist_ok=True)

    for i in range(num_tracks):
        xyz = xyz_all[i]
        if not is_good[i]:
            if verbose:
                print("Reject i=%i, CFIT = %i, v = [%0.3f, %0.3f]" % (i + 1, int(flags[i] & 1), vmm[i, 0], vmm[i, 1]))   # :284
            continue
        if not write_files:
            continue
        z0, v0 = xyz[0, 2], Ti[i, ci_spd] * ur_speed
        out_name = "BAYES_t%i_id%i_alt%i_speed%i.csv" % (xyz.shape[0] - 1, i + 1, _matlab_round(z0), _matlab_round(v0))     # :249
        k = int(np.searchsorted(Lgrid, z0, side="right")) - 1                                              # discretize(z0, L), :263
        if z0 == Lgrid[-1]:
            k = len(Lgrid) - 2
        if k < 0 or k >= len(Lgrid) - 1:
            raise ValueError("sample2track: initial altitude %g ft is outside the altitude directories" % z0)
        parts = []
        if is_geo:
            parts.append("G%i" % Ti[i, ci_geo])                                                            # :253-255
        if is_air:
            parts.append("A%i" % Ti[i, ci_air])                                                            # :258-260
        parts.append("%ift" % Lgrid[k])
        out_dir = os.path.join(out_dir_parent, *parts)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, out_name), "w", encoding="utf-8", newline="\n") as f:              # :274-279
            f.write("time_s,x_ft,y_ft,z_ft\n")
            f.write("".join("%i,%0.0f,%0.0f,%0.0f\n" % (t, xyz[t, 0], xyz[t, 1], xyz[t, 2]) for t in range(xyz.shape[0])))

    T_initial = {name: Ti[:, c].copy() for c, name in enumerate(names_i)}
    T_initial[names_i[ci_spd]] *= ur_speed                                                                 # :126-128
    if ci_acc is not None:
        T_initial[names_i[ci_acc]] *= ur_speed
    if ci_vr is not None:
        T_initial[names_i[ci_vr]] *= ur_vertrate
    return is_good, T_initial
