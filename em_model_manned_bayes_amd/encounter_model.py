"""Class-level mirror of the reference's model classes:

  EncounterModel         @EncounterModel/EncounterModel.m:1-353
  EncounterModelEvents   @EncounterModelEvents/EncounterModelEvents.m:1-91
  UncorEncounterModel    @UncorEncounterModel/UncorEncounterModel.m (constructor, .sample)
  CorTerminalModel       @CorTerminalModel/CorTerminalModel.m, sample.m, getDynamicLimits.m

Same property and method names and argument meaning.  `.sample` runs on the GPU through
libemgpu (there is no CPU path).  UncorEncounterModel.track runs on the GPU too, on a documented point-mass
model in place of the un-vendored em-core dynamics ("dynamics unpinned"); its 'geodetic' branch needs em-core / DEM data
and raises NotImplementedError.  CorTerminalModel.track (geometry draw -> createEncounter -> the filters of track.m) runs on
the GPU as well, given the trajectory-model files.
"""
import os

import numpy as np

from . import _lib as L
from . import em_io, native
from .functions import (_take, bn_dirichlet_prior, events2controls, events2samples, hierarchical_cutpoints)


class EncounterModelEvents:
    """[time_s, verticalRate_fps, turnRate_radps, longitudeAccel_ftpss] container
    (EncounterModelEvents.m:19-74)."""

    def __init__(self, event=None, time_s=0.0, verticalRate_fps=0.0, turnRate_radps=0.0, longitudeAccel_ftpss=0.0):
        if event is not None:
            self.event = event
        else:
            self.time_s = np.atleast_1d(np.asarray(time_s, dtype=np.float64))
            self.verticalRate_fps = np.atleast_1d(np.asarray(verticalRate_fps, dtype=np.float64))
            self.turnRate_radps = np.atleast_1d(np.asarray(turnRate_radps, dtype=np.float64))
            self.longitudeAccel_ftpss = np.atleast_1d(np.asarray(longitudeAccel_ftpss, dtype=np.float64))
            sizes = {a.shape for a in (self.time_s, self.verticalRate_fps, self.turnRate_radps, self.longitudeAccel_ftpss)}
            assert len(sizes) == 1, "Sizes of time_s, verticalRate_fps, turnRate_radps, longitudeAccel_ftpss are not equal"

    @classmethod
    def _of_rows(cls, m):
        """An object over the columns of an [k, 4] float64 control matrix, without the constructor's conversions (.sample builds one per trajectory)."""
        e = object.__new__(cls)
        e.time_s, e.verticalRate_fps, e.turnRate_radps, e.longitudeAccel_ftpss = m[:, 0], m[:, 1], m[:, 2], m[:, 3]
        return e

    @property
    def event(self):
        m = np.stack([self.time_s.reshape(-1), self.verticalRate_fps.reshape(-1), self.turnRate_radps.reshape(-1),
                      self.longitudeAccel_ftpss.reshape(-1)], axis=1)
        return m if m.shape[0] else np.zeros((1, 4))  # always at least one row (EncounterModelEvents.m:41-47)

    @event.setter
    def event(self, m):
        m = np.asarray(m, dtype=np.float64).reshape(-1, 4)
        self.time_s, self.verticalRate_fps, self.turnRate_radps, self.longitudeAccel_ftpss = m[:, 0], m[:, 1], m[:, 2], m[:, 3]


class EncounterModel:
    """Holds the model parameters (EncounterModel.m:5-70) and a native emgpu_model that every
    mutable property is pushed into (N_*, prior, start, boundaries, resample_rates)."""

    def __init__(self, parameters_filename="", idxZeroBoundaries=(), isOverwriteZeroBoundaries=False, **fields):
        self.isAutoUpdate = False
        self._native = None
        self._prior = 0
        if parameters_filename:
            p = em_io.em_read(parameters_filename, idxZeroBoundaries=idxZeroBoundaries or (1, 2, 3),
                              isOverwriteZeroBoundaries=isOverwriteZeroBoundaries)
            self._native = p["native"]
        else:
            p = dict(fields)
        self.parameters_filename = parameters_filename
        self.labels_initial = list(p.get("labels_initial", []))
        self.labels_transition = list(p.get("labels_transition", []))
        self.temporal_map = np.asarray(p.get("temporal_map", np.zeros((0, 2))), dtype=np.int64).reshape(-1, 2)
        self.G_initial = np.asarray(p.get("G_initial", np.zeros((0, 0))), dtype=bool)
        self.G_transition = np.asarray(p.get("G_transition", np.zeros((0, 0))), dtype=bool)
        self.bounds_initial = np.asarray(p.get("bounds_initial", np.zeros((0, 2))), dtype=np.float64).reshape(-1, 2)
        self.cutpoints_initial = [np.asarray(c, dtype=np.float64).reshape(-1) for c in p.get("cutpoints_initial", [])]
        self._boundaries = [np.asarray(b, dtype=np.float64).reshape(-1) for b in p.get("boundaries", [])]
        self.zero_bins = list(p.get("zero_bins", []))
        self._N_initial = [np.asarray(N, dtype=np.float64) for N in p.get("N_initial", [])]
        self._N_transition = [np.asarray(N, dtype=np.float64) for N in p.get("N_transition", [])]
        self._resample_rates = np.asarray(p.get("resample_rates", np.zeros(len(self.labels_initial))), dtype=np.float64).reshape(-1)
        self.all_repeat = None
        self.all_change = None
        if not self._N_initial and self.labels_initial:
            self.preallocNInitial()
        self.preallocStart()
        self.updateDirichletInitial()
        self.updateDirichletTransition()
        self.isAutoUpdate = True

    # ---- dependent properties (EncounterModel.m:290-350)
    @property
    def n_initial(self):
        return len(self.labels_initial)

    @property
    def n_transition(self):
        return len(self.labels_transition)

    @property
    def order_initial(self):
        return self.native.get_i32(L.F_ORDER_INITIAL)

    @property
    def order_transition(self):
        return self.native.get_i32(L.F_ORDER_TRANSITION)

    @property
    def r_initial(self):
        return np.array([len(c) + 1 for c in self.cutpoints_initial], dtype=np.int64)

    @property
    def cutpoints_transition(self):
        is_dot = ["\\dot" in lab for lab in self.labels_initial]
        return list(self.cutpoints_initial) + [c for c, d in zip(self.cutpoints_initial, is_dot) if d]

    @property
    def r_transition(self):
        # EncounterModel.m:313-318 (appends every "\dot" variable; wrong for the correlated models,
        # SURVEY.md Appendix C).  The sampling path uses r from the file / the N shapes instead.
        return np.array([len(c) + 1 for c in self.cutpoints_transition], dtype=np.int64)

    @property
    def bounds_transition(self):
        is_dot = np.array(["\\dot" in lab for lab in self.labels_initial], dtype=bool)
        return np.vstack([self.bounds_initial, self.bounds_initial[is_dot]])

    @property
    def dediscretize_parameters(self):
        out = []
        for i in range(self.n_initial):
            if self.bounds_initial[i, 0] == self.bounds_initial[i, 1]:
                out.append(np.zeros(0))
            else:
                out.append(np.concatenate([[self.bounds_initial[i, 0]], self.cutpoints_initial[i], [self.bounds_initial[i, 1]]]))
        return out

    @property
    def cutpoints_fine(self):
        out = []
        for i in range(self.n_initial):
            if self.bounds_initial[i, 0] == self.bounds_initial[i, 1]:
                out.append([])
            else:
                out.append(hierarchical_cutpoints(self.cutpoints_initial[i], self.bounds_initial[i], 3))
        return out

    # ---- native handle, rebuilt lazily when the structure was given by arrays
    @property
    def native(self):
        if self._native is None:
            nt = self.n_transition
            r_i = [N.shape[0] for N in self._N_initial]
            r_t = r_i + [self._N_transition[v].shape[0] for v in range(self.n_initial, nt)] if nt else None
            self._native = native.NativeModel.from_arrays(
                self.G_initial, r_i, self._N_initial, self.G_transition if nt else None, r_t,
                self._N_transition if nt else None, self.temporal_map if nt else None,
                self._boundaries or None, self.zero_bins or None, self._resample_rates, self.labels_initial, self.labels_transition or None)
            self._push_prior()
            self._native.set_start(self._start)
        return self._native

    # ---- mutable properties with write-through (EncounterModel.m:156-210)
    @property
    def N_initial(self):
        return self._N_initial

    @N_initial.setter
    def N_initial(self, v):
        self._N_initial = [np.asarray(N, dtype=np.float64) for N in v]
        if self._native is not None:
            for i, N in enumerate(self._N_initial):
                self._native.set_f64(L.F_N_INITIAL, i + 1, N.T.reshape(-1))

    @property
    def N_transition(self):
        return self._N_transition

    @N_transition.setter
    def N_transition(self, v):
        self._N_transition = [np.asarray(N, dtype=np.float64) for N in v]
        if self._native is not None:
            for i, N in enumerate(self._N_transition):
                if N.size:
                    self._native.set_f64(L.F_N_TRANSITION, i + 1, N.T.reshape(-1))

    @property
    def boundaries(self):
        return self._boundaries

    @boundaries.setter
    def boundaries(self, v):
        self._boundaries = [np.asarray(b, dtype=np.float64).reshape(-1) for b in v]
        if self._native is not None:
            for i, b in enumerate(self._boundaries):
                self._native.set_f64(L.F_BOUNDARIES, i + 1, b)

    @property
    def resample_rates(self):
        return self._resample_rates

    @resample_rates.setter
    def resample_rates(self, v):
        self._resample_rates = np.asarray(v, dtype=np.float64).reshape(-1)
        if self._native is not None:
            self._native.set_f64(L.F_RESAMPLE_RATES, 0, self._resample_rates)

    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, v):
        old = self._prior
        if not isinstance(v, (str, int, float, np.floating, np.integer)):
            raise L.EmgpuError(L.ERR_PRIOR, "Second argument must be a char or double")
        self._prior = v
        if self.isAutoUpdate and str(old).lower() != str(v).lower():  # EncounterModel.m:194-203
            self.updateDirichletInitial()
            self.updateDirichletTransition()

    @property
    def start(self):
        return self._start

    @start.setter
    def start(self, v):
        v = list(v)
        assert len(v) == len(self._N_initial)
        self._start = v
        if self._native is not None:
            self._native.set_start(v)

    @property
    def start_log_weight(self):
        """Importance-sampling hook: log of the model probability of the values `start` forces -- the
        log-weight of every sample drawn with this start distribution (0.0 without presets)."""
        return self.native.start_log_weight()

    def _push_prior(self):
        if self._native is not None:
            self._native.set_prior(self._prior)

    def updateDirichletInitial(self):
        self.dirichlet_initial = bn_dirichlet_prior(self._N_initial, self._prior)
        self._push_prior()

    def updateDirichletTransition(self):
        self.dirichlet_transition = bn_dirichlet_prior(self._N_transition, self._prior)
        self._push_prior()

    def updateBoundaries(self):
        _ = self.dediscretize_parameters  # EncounterModel.m:235-237 assigns to a local: a no-op in the reference too

    def updateResampleRates(self):
        if self.all_change is None or self.all_repeat is None:
            return
        nv = np.asarray(self.all_change, dtype=np.float64) / (np.asarray(self.all_repeat, dtype=np.float64) + np.asarray(self.all_change, dtype=np.float64))
        nv = nv.reshape(-1)
        nv[:2] = 0
        self.resample_rates = nv

    def preallocStart(self):
        self._start = [None] * len(self._N_initial)
        if self._native is not None:
            self._native.set_start(self._start)

    def preallocNInitial(self):
        r = self.r_initial
        self._N_initial = [np.zeros((int(r[i]), int(np.prod(r[self.G_initial[:, i]])))) for i in range(self.n_initial)]

    def setParameters(self, N_initial, N_transition, all_repeat, all_change):
        self.N_initial, self.N_transition, self.all_repeat, self.all_change = N_initial, N_transition, all_repeat, all_change
        self.updateResampleRates()

    def struct(self):
        """Cast to a plain dict (EncounterModel.m:217-233): what dbn_sample's `parms` argument reads."""
        keys = ["labels_initial", "labels_transition", "temporal_map", "G_initial", "G_transition", "bounds_initial",
                "cutpoints_initial", "boundaries", "zero_bins", "N_initial", "N_transition", "resample_rates", "prior",
                "dirichlet_initial", "dirichlet_transition", "start", "n_initial", "n_transition", "order_initial",
                "r_initial", "dediscretize_parameters"]
        s = em_io.Parms({k: getattr(self, k) for k in keys})
        if self.n_transition:
            s["order_transition"] = self.order_transition
            s["r_transition"] = self.native.get_i32(L.F_R_TRANSITION)
        s["native"] = self.native
        return s


class UncorEncounterModel(EncounterModel):
    """@UncorEncounterModel/UncorEncounterModel.m.  Default model: uncor_1200only_fwse_v1p2 (:26)."""

    def __init__(self, parameters_filename=None, idxZeroBoundaries=(1, 2, 3), isOverwriteZeroBoundaries=False, input_type="file"):
        if input_type != "file":
            raise NotImplementedError("input_type '%s': the training-side constructors are outside the sampling path" % input_type)
        if parameters_filename is None:
            base = os.environ.get("AEM_DIR_BAYES")
            if base:
                parameters_filename = os.path.join(base, "model", "uncor_1200only_fwse_v1p2.txt")
            else:
                import tempfile
                parameters_filename = em_io.materialize_model("uncor_1200only_fwse_v1p2", tempfile.mkdtemp(prefix="emgpu_model_"))
        super().__init__(parameters_filename=parameters_filename, idxZeroBoundaries=idxZeroBoundaries,
                         isOverwriteZeroBoundaries=isOverwriteZeroBoundaries)
        self.isRotorcraft = "rotorcraft" in os.path.basename(str(parameters_filename))  # :181-185

    def sample(self, n_samples, sample_time, seed=None, isQuantize500=False, layers=None,
               transition_mode=L.TRANSITION_REFERENCE_AUTO, max_attempts=1000, first_index=None, ctx=None):
        """[out_inits, out_events, out_samples, out_EME] = sample(self, n_samples, sample_time, 'seed', s,
        'isQuantize500', b, 'layers', L)   (UncorEncounterModel.m:192-313)."""
        labs = self.labels_initial

        def find(name):
            q = '"%s"' % name
            return labs.index(q) + 1 if q in labs else 0
        idxL, idxV, idxDV, idxDH, idxDPsi = find("L"), find("v"), find("\\dot v"), find("\\dot h"), find("\\dot \\psi")
        if not (idxDV and idxDH and idxDPsi):  # :231-234
            e = L.EmgpuError(L.ERR_ARG, "Model does not have a dynamic variable for either acceleration, vertical rate, or turn rate")
            e.identifier = "dynvar:empty"
            raise e
        s, first = _take(seed, n_samples)
        if first_index is not None:
            first = int(first_index)
        ctx = ctx or native.default_context()
        m = self.native
        flags = L.FLAG_QUANTIZE500 if isQuantize500 else 0
        n_samples = int(n_samples)
        ni, T = self.n_initial, int(sample_time)
        out_inits = np.zeros((n_samples, ni))
        out_events, out_samples, out_EME = [None] * n_samples, [None] * n_samples, [None] * n_samples
        tm = self.temporal_map
        idxEME = [int(np.nonzero(tm[:, 0] == v)[0][0]) + 1 for v in (idxDH, idxDPsi, idxDV)]  # :291
        vars_dyn = tm[:, 0].astype(np.int64) - 1
        chunk, cap, pos = max(1, min(32768, (96 << 20) // (8 * ni * T))), 256, 0   # the dense [chunk, ni, T] f64 block stays < 100 MB
        import time as _time
        t_call, tm = _time.perf_counter(), {"native_s": 0.0, "kernel_ms": 0.0, "d2h_ms": 0.0, "bytes_d2h": 0, "calls": 0}
        while pos < n_samples:
            nn = min(chunk, n_samples - pos)
            t_nat = _time.perf_counter()
            try:
                res = native.sample_dbn_host(ctx, m, nn, T, s, first_index=first + pos, want_dense=False, want_events=True,
                                             event_cap=cap, flags=flags, layers=layers, transition_mode=transition_mode,
                                             max_attempts=max_attempts, idx_L=idxL, idx_v=idxV, idx_dh=idxDH)
            except L.EmgpuError as e:
                if e.code == L.ERR_EVENT_CAP:   # longest event list did not fit: retry this chunk with more room
                    cap *= 2
                    continue
                raise
            tm["native_s"] += _time.perf_counter() - t_nat
            st = res["host_stats"]
            tm["kernel_ms"] += st["kernel_ms"]; tm["d2h_ms"] += st["d2h_ms"]; tm["bytes_d2h"] += st["bytes_d2h"]; tm["calls"] += 1
            iv = res["init_val"].astype(np.float64)
            out_inits[pos: pos + nn] = iv
            # The whole chunk at once (events2samples.m:9-26, events2controls.m:11-31 restated on flat arrays):
            # row r of sample i happens at absolute second at = cumsum(dt) and, if it names a variable, sets it
            # from then on; the state during [t, t+dt) before the row is what the row's control line reports.
            cnt = res["ev_count"].astype(np.int64)
            flat = res["events_flat"]
            sid = np.repeat(np.arange(nn), cnt)
            dt = flat["dt"].astype(np.float64); var = flat["var"].astype(np.int64); val = flat["value"].astype(np.float64)
            ends = np.cumsum(cnt)
            csum = np.cumsum(dt)
            base = np.concatenate([[0.0], csum[ends[:-1] - 1]]) if nn > 1 else np.zeros(nn)
            at = (csum - np.repeat(base, cnt)).astype(np.int64)            # time after the row
            t0 = at - dt.astype(np.int64)                                  # time before the row
            ev_all = np.stack([dt, var.astype(np.float64), val], axis=1)
            ch = (var > 0) & (at < T)
            if ch.any():
                # index (+1) of the latest change at or before t, per (sample, variable): scattered through ONE flat index, carried forward by a
                # running maximum, then gathered (a take by index, and the initial value where nothing has changed yet: six times faster than
                # boolean-mask assignment on a [nn, ni, T] block)
                order = np.flatnonzero(ch)
                last = np.zeros(nn * ni * T, dtype=np.int32)
                last[(sid[order] * ni + (var[order] - 1)) * T + at[order]] = order + 1
                last = last.reshape(nn, ni, T)
                np.maximum.accumulate(last, axis=2, out=last)
                D = np.concatenate(([0.0], val))[last]
                np.copyto(D, iv[:, :, None], where=last == 0)
            else:
                D = np.broadcast_to(iv[:, :, None], (nn, ni, T)).copy()
            crow = dt > 0                                                   # rows that open a control line (:19-24)
            csid, ct0 = sid[crow], t0[crow]
            ctrl = np.empty((csid.size, 4))
            ctrl[:, 0] = ct0
            ctrl[:, 1] = D[csid, vars_dyn[idxEME[0] - 1], ct0] / 60.0                    # dh: fpm -> fps          :295
            ctrl[:, 2] = np.deg2rad(D[csid, vars_dyn[idxEME[1] - 1], ct0])               # dpsi: deg/s -> rad/s    :296
            ctrl[:, 3] = D[csid, vars_dyn[idxEME[2] - 1], ct0] * 1.68780972222222        # dv: kt/s -> ft/s^2      :297
            ev_split = native.split_rows(ev_all, ends)
            ctrl_split = native.split_rows(ctrl, np.cumsum(np.bincount(csid, minlength=nn)))
            out_events[pos: pos + nn] = ev_split
            out_samples[pos: pos + nn] = list(D)                              # nn views of the chunk's block
            out_EME[pos: pos + nn] = [EncounterModelEvents._of_rows(c) for c in ctrl_split]
            pos += nn
        # the call's three phases (bench.py `host_path.class_sample`): the library calls (kernel + PCIe + the binding's own copies), of which
        # kernel_ms / d2h_ms are the device's share, and the numpy / Python reconstruction of out_samples, controls and EncounterModelEvents
        tm["total_s"] = _time.perf_counter() - t_call
        tm["format_s"] = tm["total_s"] - tm["native_s"]
        self.last_sample_timing = tm
        return out_inits, out_events, out_samples, out_EME

    TRACK_FIELDS = ("time_s", "north_ft", "east_ft", "up_ft", "speed_ft_s", "phi_rad", "theta_rad", "psi_rad")

    def track(self, nSamples, sample_time, initialSeed=None, isQuantize500=False, coordSys="NEU", max_track_attempts=200,
              record_stride=1, first_index=None, ctx=None, return_info=False):
        """out_results = track(self, nSamples, sample_time, 'initialSeed', s, 'isQuantize500', b, 'coordSys', 'NEU')
        (UncorEncounterModel.m:318-471).  Each result is a dict of 1-D arrays with the columns of the reference's timetable
        (time_s instead of the row times), sampled at 10 Hz (record_stride=1) like results.time.

        Runs on the GPU end to end: per round the still-rejected trajectories are sampled (attempt j with the key
        initialSeed + j, :424-428), integrated and tested against getDynamicLimits (:459-470) on the device.
        DYNAMICS UNPINNED: the reference integrates with em-core's run_dynamics_fast, which it does not vendor; this build
        uses the point-mass model documented in DESIGN.md / emgpu.h.  coordSys 'geodetic' (:480-540) needs a DEM, the FAA
        obstacle file and em-core's placeTrack and is out of scope."""
        if str(coordSys).lower() != "neu":
            raise NotImplementedError("coordSys 'geodetic' needs a DEM, the digital obstacle file and em-core's placeTrack "
                                      "(UncorEncounterModel.m:480-540): out of scope, use 'NEU'")
        for lab in ('"\\dot v"', '"\\dot h"', '"\\dot \\psi"'):
            if lab not in self.labels_initial:
                e = L.EmgpuError(L.ERR_ARG, "Model does not have a dynamic variable for either acceleration, vertical rate, or turn rate")
                e.identifier = "dynvar:empty"
                raise e
        s, first = _take(initialSeed, nSamples)
        if first_index is not None:
            first = int(first_index)
        res = native.track_uncor_host(ctx or native.default_context(), self.native, int(nSamples), int(sample_time), s, first_index=first,
                                      is_quantize500=isQuantize500, is_rotorcraft=self.isRotorcraft, max_track_attempts=max_track_attempts,
                                      record_stride=record_stride)
        out_results = [{f: res["tracks"][i, :, k].copy() for k, f in enumerate(self.TRACK_FIELDS)} for i in range(int(nSamples))]
        return (out_results, res) if return_info else out_results

    def getDynamicLimits(self, initial, results, idx_G=None, idx_A=None, idx_L=None, idx_V=None, idx_DH=None, is_discretized=None):
        """dynamiclimits = getDynamicLimits(self, initial, results, ...)  (@UncorEncounterModel/getDynamicLimits.m).  The index
        arguments are accepted for signature compatibility; they are looked up from the labels like .track does."""
        up, sp = np.asarray(results["up_ft"], dtype=np.float64), np.asarray(results.get("speed_ftps", results.get("speed_ft_s")), dtype=np.float64)
        lim = native.uncor_dynamic_limits(self.native, initial, up.min(), up.max(), sp.min(), sp.max(), self.isRotorcraft)
        return {"minVel_ft_s": lim[0], "maxVel_ft_s": lim[1], "maxVertRate_ft_s": lim[2]}


# @CorTerminalModel/getDynamicLimits.m:15-62
_DYN_LIMITS = {
    "GENERIC": dict(minVel_ft_s=50, maxVel_ft_s=506, maxTurnRate_deg_s=12, maxAltitude_ft=5000, maxVertRate_ft_s=6000 / 60, maxCumTurn_deg=np.inf, pitch_deg=np.inf),
    "RTCA228_A1": dict(minVel_ft_s=169, maxVel_ft_s=491, maxTurnRate_deg_s=1.5, maxAltitude_ft=5000, maxVertRate_ft_s=2500 / 60, maxCumTurn_deg=180, pitch_deg=15),
    "RTCA228_A2": dict(minVel_ft_s=68, maxVel_ft_s=338, maxTurnRate_deg_s=3, maxAltitude_ft=5000, maxVertRate_ft_s=1500 / 60, maxCumTurn_deg=180, pitch_deg=15),
    "RTCA228_A3": dict(minVel_ft_s=68, maxVel_ft_s=186, maxTurnRate_deg_s=7, maxAltitude_ft=5000, maxVertRate_ft_s=500 / 60, maxCumTurn_deg=180, pitch_deg=15),
    "TEST": dict(minVel_ft_s=68, maxVel_ft_s=186, maxTurnRate_deg_s=7, maxAltitude_ft=1200, maxVertRate_ft_s=500 / 60, maxCumTurn_deg=180, pitch_deg=15),
}


class CorTerminalModel(EncounterModel):
    """@CorTerminalModel: the encounter-geometry Bayesian network and its rejection sampler
    (CorTerminalModel.m:45-111, sample.m:1-82).  The 20 trajectory-model files are absent from the
    reference mount (.MISSING_LARGE_BLOBS): with only the geometry model in parameters_directory .sample works and
    .createEncounter / .track raise NotImplementedError; with the files (or synthetic.write_terminal_directory's stand-ins)
    they run on the GPU."""

    def __init__(self, srcData="terminalradar", parameters_directory=None, compatBackwardModels=True):
        """compatBackwardModels: CorTerminalModel.m:97,100 load the intruder LANDING reverse file as the
        backward model of the intruder take-off and transit intents as well (a copy-paste slip in the
        reference).  True reproduces that; False uses the file each intent names."""
        self.srcData = srcData
        self.compatBackwardModels = bool(compatBackwardModels)
        self._traj = None
        name = {"terminalradar": "terminal_v3_radar_encounter_model", "opensky": "terminal_v3_opensky_encounter_model"}.get(srcData)
        if parameters_directory is None:
            base = os.environ.get("AEM_DIR_BAYES")
            if base and os.path.isdir(os.path.join(base, "model", "correlated_terminal", srcData)):
                parameters_directory = os.path.join(base, "model", "correlated_terminal", srcData)
        if parameters_directory is not None:
            import glob
            hits = glob.glob(os.path.join(parameters_directory, "*_encounter_model.txt"))
            if not hits:
                raise FileNotFoundError("no *_encounter_model.txt in %s" % parameters_directory)
            fname = hits[0]
        else:
            if name is None:
                raise ValueError("unknown srcData %r" % srcData)
            import tempfile
            fname = em_io.materialize_model(name, tempfile.mkdtemp(prefix="emgpu_model_"))
        self.parameters_directory = parameters_directory
        super().__init__(parameters_filename=fname, idxZeroBoundaries=(1, 2, 3), isOverwriteZeroBoundaries=False)
        self.bounds_sample = np.column_stack([-np.inf * np.ones(self.n_initial), np.inf * np.ones(self.n_initial)])
        self.acType1 = "GENERIC"
        self.acType2 = "GENERIC"
        # trajectory models (CorTerminalModel.m:84-100), present only when the directory holds them
        if parameters_directory is not None:
            import glob

            def find(stem):
                hits = glob.glob(os.path.join(parameters_directory, "*_" + stem + ".txt"))
                return hits[0] if hits else None
            bck_to = "intruder_landing_model_reverse" if self.compatBackwardModels else None
            stems = ["ownship_landing_model", "ownship_landing_model_reverse", "ownship_takeoff_model", "ownship_takeoff_model_reverse",
                     "intruder_landing_model", "intruder_landing_model_reverse", "intruder_takeoff_model",
                     bck_to or "intruder_takeoff_model_reverse", "intruder_transit_model", bck_to or "intruder_transit_model_reverse"]
            files = [find(s) for s in stems]
            if all(files):
                names = ["mdlFwd1_1", "mdlBck1_1", "mdlFwd1_2", "mdlBck1_2", "mdlFwd2_1", "mdlBck2_1", "mdlFwd2_2", "mdlBck2_2", "mdlFwd2_3", "mdlBck2_3"]
                self._traj = []
                for nm_, f in zip(names, files):
                    mdl = EncounterModel(parameters_filename=f, idxZeroBoundaries=(1, 2, 3), isOverwriteZeroBoundaries=False)
                    mdl.native.set_transition_stay_prior(1.0)      # createEncounter.m:128-129
                    setattr(self, nm_, mdl)
                    self._traj.append(mdl)

    def getDynamicLimits(self, acId):
        if acId not in (1, 2):
            raise ValueError("acId must be either 1 or 2")
        actype = (self.acType1 if acId == 1 else self.acType2).upper()
        if actype not in _DYN_LIMITS:
            raise ValueError("Unknown aircraft type of %s" % actype)
        d = dict(_DYN_LIMITS[actype])
        # maxAccel_ft_s_s = max(diff(cutpoints_initial{speed})) of the ownship trajectory model (:78-79);
        # that file is absent, the geometry model's own_speed cut points have the same grid.
        q = '"own_speed"'
        if q in self.labels_initial:
            d["maxAccel_ft_s_s"] = float(np.max(np.diff(self.cutpoints_initial[self.labels_initial.index(q)])))
        return d

    @property
    def dynLimits1(self):
        return self.getDynamicLimits(1)

    @property
    def dynLimits2(self):
        return self.getDynamicLimits(2)

    def sample(self, nSamples, seed=None, max_attempts=100000, first_index=None, ctx=None, start_grid=None, return_log_weight=False):
        """[outInits, outSamples] = sample(self, nSamples, 'seed', s)  (@CorTerminalModel/sample.m:1-82).
        start_grid: the rows InitStartTerminal returns (or any list of `start` values, one per sample: None / 0 = unset) -- the whole grid
        is drawn in ONE launch, sample i with the presets of row i (RUN_terminal.m:33-50 sets self.start and calls sample once per row);
        nSamples must then equal its length.  return_log_weight: also return log P(presets of the row) per sample (importance weights)."""
        grid = None
        if start_grid is not None:
            grid = np.array([[0 if (v is None or (isinstance(v, float) and np.isnan(v))) else int(v) for v in row] for row in start_grid], dtype=np.int32)
            if grid.shape != (int(nSamples), self.n_initial):
                raise ValueError("start_grid must have nSamples rows of n_initial entries")
        s, first = _take(seed, nSamples)
        if first_index is not None:
            first = int(first_index)
        labs = self.labels_initial
        io, ii = labs.index('"own_speed"') + 1, labs.index('"int_speed"') + 1
        d1, d2 = self.dynLimits1, self.dynLimits2
        bs = None if np.all(np.isinf(self.bounds_sample)) else self.bounds_sample
        res = native.sample_bn_host(ctx or native.default_context(), self.native, int(nSamples), s, first_index=first,
                                    dediscretize=True, max_attempts=max_attempts, bounds_sample=bs,
                                    idx_own_speed=io, idx_int_speed=ii,
                                    lim1=(d1["minVel_ft_s"], d1["maxVel_ft_s"]), lim2=(d2["minVel_ft_s"], d2["maxVel_ft_s"]),
                                    start=grid, want_log_weight=return_log_weight)
        outInits = res[1].astype(np.float64)
        names = [lab.replace('"', "") for lab in labs]
        outSamples = [dict(zip(names, row)) for row in outInits]
        if return_log_weight:
            return outInits, outSamples, res[3]
        return outInits, outSamples

    def InitStartTerminal(self, nSamples=1000000, airspace_class=(False, True, True, True), own_intent=(True, True),
                          int_intent=(True, True, True), isVerbose=False):
        """out_start = InitStartTerminal(self, nSamples, airspace_class, own_intent, int_intent)
        (@CorTerminalModel/InitStartTerminal.m:1-92): a grid of start distributions, one row per encounter,
        the first three variables preset to every kept (airspace class, ownship intent, intruder intent)
        combination in turn.  Rows are lists usable as `self.start`."""
        assert self.labels_initial[0] == '"airspace_class"' and self.labels_initial[1] == '"own_intent"' and self.labels_initial[2] == '"int_intent"'
        r = self.r_initial
        assert len(airspace_class) == r[0] and len(own_intent) == r[1] and len(int_intent) == r[2]
        idx_class = [i + 1 for i, k in enumerate(airspace_class) if k]
        idx_own = [i + 1 for i, k in enumerate(own_intent) if k]
        idx_int = [i + 1 for i, k in enumerate(int_intent) if k]
        n_combs = len(idx_class) * len(idx_own) * len(idx_int)
        n_samples = int(nSamples)
        if n_combs > n_samples:
            n_samples = n_combs                                          # :50-53
        per = -(-n_samples // n_combs)                                   # ceil, :56
        out = []
        for ii in idx_class:
            for jj in idx_own:
                for kk in idx_int:
                    for _ in range(per):
                        out.append([ii, jj, kk] + [None] * (self.n_initial - 3))
        return out

    _warned_smooth = False

    @classmethod
    def _note_smooth(cls, local_smooth):
        """createEncounter.m:88-89 smooth through em-core's local_smooth, which the reference does not vendor: the default (True, like the
        reference) runs the library's documented stand-in.  Said once per process, so nobody takes the smoothed columns for pinned ones."""
        if local_smooth and not cls._warned_smooth:
            import warnings
            cls._warned_smooth = True
            warnings.warn("CorTerminalModel: speed and altitude are smoothed like createEncounter.m:88-89, but em-core's local_smooth is not part of the "
                          "reference checkout: this is the library's stand-in (EMGPU_FLAG_LOCAL_SMOOTH, a centred moving average; UNPINNED). "
                          "Pass local_smooth=False for the propagated values as they are.", stacklevel=3)

    # ---- createEncounter.m:1-91 without em-core's local_smooth (:88-89)
    @staticmethod
    def _sincosd(deg):
        # sind / cosd with MATLAB's reduction in degrees: n = round(x/90), x - 90 n in [-45, 45], quadrant m = mod(n, 4)
        deg = np.asarray(deg, dtype=np.float64)
        n = np.sign(deg) * np.floor(np.abs(deg) / 90.0 + 0.5)          # round half away from zero
        x = (np.pi / 180.0) * (deg - n * 90.0)
        m = np.mod(n, 4.0)
        sx, cx = np.sin(x), np.cos(x)
        s = np.select([m == 0, m == 1, m == 2], [sx, cx, -sx], -cx)
        c = np.select([m == 0, m == 1, m == 2], [cx, -sx, -cx], sx)
        return s, c

    def _geo_rows(self, samples):
        """x0 y0 z0 v0 heading0 intent for aircraft 1 and 2 (createEncounter.m:41-49) and the model of each of the 4 tracks."""
        g = np.zeros((len(samples), 12))
        mo = np.zeros((len(samples), 4), dtype=np.int32)
        for e_, sg in enumerate(samples):
            for a, pre in enumerate(("own", "int")):
                s, c = self._sincosd(sg[pre + "_bearing"])
                g[e_, 6 * a: 6 * a + 6] = [sg[pre + "_distance"] * c, sg[pre + "_distance"] * s, sg[pre + "_alt"], sg[pre + "_speed"],
                                           sg[pre + "_heading"], sg[pre + "_intent"]]
            oi, ii_ = int(sg["own_intent"]), int(sg["int_intent"])
            if oi not in (1, 2) or ii_ not in (1, 2, 3):
                raise ValueError("Unknown intent")      # createEncounter.m:21,38
            mo[e_] = [2 * (oi - 1), 2 * (oi - 1) + 1, 4 + 2 * (ii_ - 1), 4 + 2 * (ii_ - 1) + 1]
        return g, mo

    def _dyn_rows(self):
        keys = ("minVel_ft_s", "maxVel_ft_s", "maxTurnRate_deg_s", "maxAltitude_ft", "maxVertRate_ft_s")
        return np.array([[self.dynLimits1[k] for k in keys], [self.dynLimits2[k] for k in keys]], dtype=np.float64)

    def createEncounter(self, sample_geo, tmax_s=120, seed=None, first_index=None, ctx=None, local_smooth=True):
        """traj = createEncounter(self, sample_geo, tmax_s)  (createEncounter.m:1): a list of two dicts with
        t_s, x_nm, y_nm, z_ft, heading_deg, v_ft_s sorted in time.  sample_geo may be one dict or a list of
        dicts (then a list of pairs is returned).  local_smooth: speed and altitude smoothed over 5 / 15 s like :88-89 -- em-core's
        local_smooth is not vendored by the reference, so this is the library's stand-in (EMGPU_FLAG_LOCAL_SMOOTH in include/emgpu.h: a centred
        moving average whose window shrinks at a track's ends), UNPINNED; False returns the propagated values as they are."""
        if self._traj is None:
            raise NotImplementedError("the terminal trajectory-model files are not in parameters_directory (they are absent from the "
                                      "reference mount); synthetic.write_terminal_directory builds stand-ins")
        self._note_smooth(local_smooth)
        single = isinstance(sample_geo, dict)
        samples = [sample_geo] if single else list(sample_geo)
        s, first = _take(seed, len(samples))
        if first_index is not None:
            first = int(first_index)
        g, mo = self._geo_rows(samples)
        cap = int(tmax_s) + 3
        traj, rows = native.propagate_terminal_joined_host(ctx or native.default_context(), [m.native for m in self._traj], g, mo, s,
                                                           first_index=first, tmax_s=float(tmax_s), dyn_limits=self._dyn_rows(), cap=cap,
                                                           local_smooth=local_smooth)
        res = []
        fields = ("x_nm", "y_nm", "z_ft", "heading_deg", "v_ft_s")
        for e_ in range(len(samples)):
            pair = []
            for a in range(2):
                # the library hands back what :74-84 build: [fwd, bck(1, 2:end)] ordered in time, row c0 + t for second t
                rf, rb = int(rows[4 * e_ + 2 * a]), int(rows[4 * e_ + 2 * a + 1])
                if rf < 1 or rb < 1:
                    raise RuntimeError("createEncounter: a track exceeded the re-draw cap (the reference would loop forever, createEncounter.m:192)")
                c0 = native.terminal_t0_row(cap)
                lo, hi = c0 - (rb - 1), c0 + (rf - 1)
                both = traj[2 * e_ + a, lo: hi + 1].astype(np.float64)
                d = {"t_s": np.arange(lo - c0, hi - c0 + 1, dtype=np.float64)}
                d.update({f: both[:, k].copy() for k, f in enumerate(fields)})
                pair.append(d)
            res.append(pair)
        return res[0] if single else res

    def track(self, nSamples, initialSeed=None, firstID=1, minEncTime_s=30, thresDist_ft=2.5 * 6076, thresAltLow_ft=750,
              thresVertRate_ft_s=300 / 60, max_track_attempts=500, first_index=None, ctx=None, return_info=False, local_smooth=True):
        """[out_results, gen_time_s] = track(self, nSamples, 'initialSeed', s, 'firstID', 1, 'minEncTime_s', 30, ...)
        (@CorTerminalModel/track.m).  out_results[i] = {"sample": dict of the geometry sample + id, tcpa, hmd_ft, vmd_ft, nmac, ...,
        "traj": [ownship, intruder]} with each trajectory cut to the common time span and re-based like reformatTrajFiles
        (CorTerminalModel.m:318-345: t from 0, x / y in feet, h, v).

        Runs on the GPU in rounds (geometry draw -> createEncounter inputs -> propagation -> filters, attempt j under the key
        initialSeed + j).  Unpinned pieces: em-core's computeVerticalRate / computeHeadingRate are forward differences here,
        `isClimb` (track.m:122,134, undefined in the reference) is read as is_climb, local_smooth (createEncounter.m:88-89) is the library's
        stand-in (see createEncounter; local_smooth=False: the filters read the unsmoothed tracks); the
        trajectory-model files must be in parameters_directory (synthetic.write_terminal_directory builds stand-ins)."""
        if self._traj is None:
            raise NotImplementedError("the terminal trajectory-model files are not in parameters_directory (they are absent from the "
                                      "reference mount); synthetic.write_terminal_directory builds stand-ins")
        import time
        self._note_smooth(local_smooth)
        t0 = time.perf_counter()
        s, first = _take(initialSeed, nSamples)
        if first_index is not None:
            first = int(first_index)
        d = (self.dynLimits1, self.dynLimits2)
        bs = None if np.all(np.isinf(self.bounds_sample)) else self.bounds_sample
        res = native.track_terminal_host(ctx or native.default_context(), self.native, [m.native for m in self._traj], int(nSamples), s,
                                         self._dyn_rows(), [x["maxCumTurn_deg"] for x in d], [x["pitch_deg"] for x in d], first_index=first,
                                         min_enc_time_s=minEncTime_s, thres_dist_ft=thresDist_ft, thres_alt_low_ft=thresAltLow_ft,
                                         thres_vertrate_ft_s=thresVertRate_ft_s, bounds_sample=bs, max_track_attempts=max_track_attempts,
                                         local_smooth=local_smooth)
        names = [lab.replace('"', "") for lab in self.labels_initial]
        out = []
        for i in range(int(nSamples)):
            sample = dict(zip(names, res["sample"][i]))
            tr = [res["traj"][i, a, : res["len"][i, a]] for a in range(2)]
            tcpa, hmd, vmd, _ = res["meta"][i]
            lo, hi = max(tr[0][0, 0], tr[1][0, 0]), min(tr[0][-1, 0], tr[1][-1, 0])      # reformatTrajFiles: intersect(t_s)
            fm = []
            for a in range(2):
                q = tr[a][(tr[a][:, 0] >= lo) & (tr[a][:, 0] <= hi)]
                fm.append({"t": q[:, 0] - q[0, 0], "y": q[:, 2] * 6076.1154855643, "x": q[:, 1] * 6076.1154855643, "h": q[:, 3], "v": q[:, 5]})
            tcpa_adj = int(tcpa - lo) or 1                                                 # track.m:160-163
            sample.update(id=i + int(firstID), tcpa=tcpa_adj, hmd_ft=hmd, vmd_ft=vmd, nmac=bool(abs(hmd) < 500 and abs(vmd) < 100))
            for a, pre in enumerate(("own", "int")):                                       # :171-180
                vr = np.abs(np.diff(fm[a]["h"])) * 60
                if vr.size:
                    sample[pre + "_vertRate_ftpm"] = vr[min(tcpa_adj, vr.size) - 1]
                    sample[pre + "_initVertRate_ftpm"] = vr[0]
                sample[pre + "_initSpeed_ftps"], sample[pre + "_initAlt_ft"] = fm[a]["v"][0], fm[a]["h"][0]
            out.append({"sample": sample, "traj": fm})
        gen_time_s = np.full(int(nSamples), (time.perf_counter() - t0) / max(1, int(nSamples)))
        return (out, gen_time_s, res) if return_info else (out, gen_time_s)
