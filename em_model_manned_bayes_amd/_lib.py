"""ctypes binding of libemgpu.so (include/emgpu.h).

The product path has no CPU fallback: if the HIP library is missing this module
raises at import of the symbol table, and creating a Context without a GPU raises
EmgpuError(EMGPU_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# EMGPU_LIB: alternative build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("EMGPU_LIB") or os.path.join(_HERE, "libemgpu.so")

OK = 0
ERR_ARG, ERR_IO, ERR_PARSE, ERR_PRESET, ERR_HIP = -1, -2, -3, -4, -5
ERR_REJECT_CAP, ERR_EVENT_CAP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_PRIOR, ERR_SORT = -6, -7, -8, -9, -10, -11

# field ids (emgpu.h)
F_R_INITIAL, F_R_TRANSITION, F_ORDER_INITIAL, F_ORDER_TRANSITION, F_TEMPORAL_MAP = 1, 2, 3, 4, 5
F_ZERO_BINS, F_START, F_G_INITIAL, F_G_TRANSITION = 6, 7, 8, 9
F_N_INITIAL, F_N_TRANSITION, F_ALPHA_INITIAL, F_ALPHA_TRANSITION, F_BOUNDARIES, F_RESAMPLE_RATES = 32, 33, 34, 35, 36, 37
F_LABELS_INITIAL, F_LABELS_TRANSITION = 64, 65

TRANSITION_REFERENCE_AUTO, TRANSITION_PER_STEP = 0, 1
FLAG_QUANTIZE500, FLAG_NO_RESAMPLE, FLAG_NO_DEDISC, FLAG_NO_TERMINATOR, FLAG_LOCAL_SMOOTH = 1, 2, 4, 8, 16

# MATLAB error identifiers the reference raises for the same condition
_MATLAB_IDS = {
    ERR_PRESET: "Attempt to preset a dependent variable",     # bn_sample.m:47
    ERR_PRIOR: "prior:notdbe",                                # bn_dirichlet_prior.m:28
    ERR_SORT: "Network could not be hierarchically sorted",   # bn_sort.m:23
}


class EmgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("emgpu error %d: %s" % (code, msg))
        self.code = code
        self.identifier = _MATLAB_IDS.get(code, "emgpu:error%d" % -code)


class ModelDesc(C.Structure):
    _fields_ = [("n_initial", C.c_int32), ("n_transition", C.c_int32), ("n_dyn", C.c_int32), ("_pad", C.c_int32),
                ("G_initial", C.c_void_p), ("G_transition", C.c_void_p),
                ("r_initial", C.c_void_p), ("r_transition", C.c_void_p),
                ("temporal_map", C.c_void_p),
                ("N_initial", C.c_void_p), ("n_N_initial", C.c_int64),
                ("N_transition", C.c_void_p), ("n_N_transition", C.c_int64),
                ("boundaries", C.c_void_p), ("bnd_len", C.c_void_p), ("zero_bins", C.c_void_p),
                ("resample_rates", C.c_void_p),
                ("labels_initial", C.c_char_p), ("labels_transition", C.c_char_p)]


class ModelInfo(C.Structure):
    _fields_ = [("n_initial", C.c_int32), ("n_transition", C.c_int32), ("n_dyn", C.c_int32),
                ("is_dynvar_depend", C.c_int32), ("n_N_initial", C.c_int64), ("n_N_transition", C.c_int64),
                ("max_r", C.c_int32), ("n_resample_active", C.c_int32)]


class SampleParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64),
                ("sample_time", C.c_int32), ("transition_mode", C.c_int32), ("flags", C.c_uint32),
                ("max_attempts", C.c_int32), ("idx_L", C.c_int32), ("idx_v", C.c_int32), ("idx_dh", C.c_int32),
                ("n_layers", C.c_int32), ("layers", C.c_void_p), ("event_cap", C.c_int32), ("_pad", C.c_int32),
                ("indices", C.c_void_p), ("start", C.c_void_p)]


class SampleOut(C.Structure):
    _fields_ = [("init_bin", C.c_void_p), ("init_val", C.c_void_p), ("dyn_bin", C.c_void_p), ("dyn_val", C.c_void_p),
                ("ev_count", C.c_void_p), ("events", C.c_void_p), ("attempts", C.c_void_p),
                ("ld", C.c_int64), ("col_offset", C.c_int64), ("log_weight", C.c_void_p)]


class UTrackParams(C.Structure):  # emgpu_utrack_params
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64), ("sample_time", C.c_int32), ("flags", C.c_uint32),
                ("max_track_attempts", C.c_int32), ("max_attempts", C.c_int32),
                ("idx_G", C.c_int32), ("idx_A", C.c_int32), ("idx_L", C.c_int32), ("idx_v", C.c_int32), ("idx_dv", C.c_int32),
                ("idx_dh", C.c_int32), ("idx_dpsi", C.c_int32), ("is_rotorcraft", C.c_int32), ("record_stride", C.c_int32), ("_pad", C.c_int32)]


class TTrackParams(C.Structure):  # emgpu_ttrack_params
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64), ("tmax_s", C.c_double),
                ("max_resample", C.c_int32), ("max_track_attempts", C.c_int32), ("max_attempts", C.c_int32), ("flags", C.c_uint32),
                ("dyn_limits", C.c_double * 10), ("max_cum_turn_deg", C.c_double * 2), ("pitch_deg", C.c_double * 2),
                ("min_enc_time_s", C.c_double), ("thres_dist_ft", C.c_double), ("thres_alt_low_ft", C.c_double), ("thres_vertrate_ft_s", C.c_double),
                ("bounds_sample", C.c_void_p), ("idx", C.c_int32 * 12)]


class Block(C.Structure):         # emgpu_block
    _fields_ = [("model", C.c_int32), ("_pad", C.c_int32), ("first_index", C.c_uint64), ("n", C.c_int64)]


class TrackParams(C.Structure):   # emgpu_track_params
    _fields_ = [("n", C.c_int64), ("T", C.c_int32), ("nd", C.c_int32), ("slot_vertrate", C.c_int32), ("slot_acc", C.c_int32),
                ("slot_turnrate", C.c_int32), ("reserved", C.c_int32), ("ur_speed", C.c_double), ("ur_vertrate", C.c_double),
                ("ur_heading", C.c_double), ("min_speed", C.c_double), ("max_speed", C.c_double)]


class TermParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64), ("tmax_s", C.c_double),
                ("max_resample", C.c_int32), ("cap", C.c_int32), ("dyn_limits", C.c_double * 10), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class TSampleParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64), ("tmax_s", C.c_double),
                ("max_resample", C.c_int32), ("cap", C.c_int32), ("dyn_limits", C.c_double * 10),
                ("max_attempts", C.c_int32), ("flags", C.c_uint32), ("bounds_sample", C.c_void_p), ("idx", C.c_int32 * 12)]


class TraceReport(C.Structure):   # emgpu_trace_report_t
    _fields_ = [("bytes", C.c_int64), ("ld", C.c_int64), ("candidates", C.c_int32), ("kept", C.c_int32), ("reused", C.c_int32), ("_pad", C.c_int32),
                ("ms", C.c_float * 8), ("first_allocation_ms", C.c_float), ("kept_ms", C.c_float)]


class HostStats(C.Structure):     # emgpu_host_stats_t
    _fields_ = [("total_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("scatter_ms", C.c_double),
                ("bytes_d2h", C.c_int64), ("event_rows", C.c_int64), ("chunks", C.c_int32), ("chunk_n", C.c_int32),
                ("threads", C.c_int32), ("direct", C.c_int32)]


TRACE_INIT, TRACE_DENSE, TRACE_EVENTS, TRACE_ATTEMPTS = 1, 2, 4, 8


class BnParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_index", C.c_uint64), ("n", C.c_int64), ("flags", C.c_uint32),
                ("max_attempts", C.c_int32), ("bounds_sample", C.c_void_p),
                ("idx_own_speed", C.c_int32), ("idx_int_speed", C.c_int32),
                ("min_vel1", C.c_double), ("max_vel1", C.c_double), ("min_vel2", C.c_double), ("max_vel2", C.c_double),
                ("start", C.c_void_p), ("log_weight", C.c_void_p)]


# every symbol include/emgpu.h declares (tests check the list against the header)
SYMBOLS = [
    "emgpu_last_error", "emgpu_version", "emgpu_model_load_txt", "emgpu_model_from_arrays", "emgpu_model_free",
    "emgpu_model_info", "emgpu_model_get_i32", "emgpu_model_get_f64", "emgpu_model_get_text", "emgpu_model_set_f64",
    "emgpu_model_set_prior", "emgpu_model_set_transition_stay_prior", "emgpu_model_set_start",
    "emgpu_ctx_create", "emgpu_ctx_set_stream", "emgpu_ctx_sync", "emgpu_ctx_free",
    "emgpu_sample_dbn_device", "emgpu_sample_dbn_host", "emgpu_sample_bn_device", "emgpu_sample_bn_host",
    "emgpu_last_kernel_name", "emgpu_discretize_bayes", "emgpu_asub2ind",
    "emgpu_debug_column_thresholds", "emgpu_debug_bernoulli_threshold", "emgpu_debug_dynamic_column", "emgpu_debug_padded_column",
    "emgpu_propagate_terminal_device", "emgpu_propagate_terminal_host", "emgpu_sample_terminal_device",
    "emgpu_sample2track_device", "emgpu_sample2track_host",
    "emgpu_model_set_zero_bins", "emgpu_shard_range", "emgpu_device_count", "emgpu_mixed_blocks",
    "emgpu_sample_dbn_blocks_device", "emgpu_sample_dbn_multi_host", "emgpu_sample_dbn_multi_device",
    "emgpu_track_uncor_host", "emgpu_track_uncor_device", "emgpu_uncor_dynamic_limits", "emgpu_model_start_log_weight",
    "emgpu_track_terminal_host", "emgpu_debug_parent_masks", "emgpu_last_launch_count", "emgpu_debug_pk_column", "emgpu_debug_terminal_counters", "emgpu_debug_uncor_dynamics_host", "emgpu_model_save_bin", "emgpu_model_load_bin", "emgpu_philox_rounds", "emgpu_ctx_trim",
    "emgpu_slot_map_revision", "emgpu_trace_alloc", "emgpu_trace_out", "emgpu_trace_report", "emgpu_trace_free", "emgpu_host_alloc", "emgpu_host_free", "emgpu_host_stats",
    "emgpu_device_alloc", "emgpu_device_free",
]

_lib = None
_hip_runtime = None


def _share_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (soname
    libamdhip64.so.7, like the system's).  Loaded first, libemgpu.so binds the system copy, and a later `import torch` brings a
    SECOND runtime into the process (torch's loader asks for the file name, which no loaded soname matches): two HSA instances on
    one GPU, a torch stream handed to emgpu_ctx_create is a foreign handle, and on some boxes the second initialisation fails
    ("No HIP GPUs are available").  So when a torch wheel with a bundled runtime is installed, that copy is loaded first (the
    wheel is located, not imported): libemgpu.so's NEEDED soname resolves to it, and so does torch's own load, whenever it comes.
    EMGPU_HIP_RUNTIME=system keeps the system runtime (a process that never imports torch)."""
    global _hip_runtime
    if _hip_runtime is not None or os.environ.get("EMGPU_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        # only when the wheel's copy answers to the soname libemgpu.so asks for: a wheel of another ROCm major (libamdhip64.so.6 beside a
        # system .7) would be loaded IN ADDITION to the system copy the NEEDED entry still resolves to -- the two runtimes this function
        # exists to prevent
        want, have = _needed_hip_soname(LIB_PATH), _soname_of(path)
        if want and have and want != have:
            import warnings
            warnings.warn("emgpu: the installed torch wheel bundles %s but libemgpu.so was linked against %s: not preloading it "
                          "(importing torch in this process as well would map two HIP runtimes)" % (have, want))
            return
        try:
            _hip_runtime = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            _hip_runtime = None


def _soname_of(path):
    """The soname a libamdhip64 build carries (its DT_SONAME string, "libamdhip64.so.<major>", read from the file's string table)."""
    return _needed_hip_soname(path)


def _needed_hip_soname(lib_path):
    import mmap
    import re
    try:
        with open(lib_path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as data:
            m = re.search(rb"libamdhip64\.so\.\d+", data)
            return m.group(0).decode() if m else None
    except (OSError, ValueError):
        return None


def mapped_hip_runtimes():
    """The distinct libamdhip64 files mapped into this process (/proc/self/maps): more than one is the failure _share_hip_runtime prevents."""
    seen = set()
    try:
        for ln in open("/proc/self/maps"):
            if "libamdhip64" in ln:
                seen.add(os.path.realpath(ln.split()[-1]))
    except OSError:
        pass
    return sorted(seen)


def lib():
    """Load libemgpu.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libemgpu.so not found at %s: build it with `make -C em_model_manned_bayes_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime()
    L = C.CDLL(LIB_PATH)
    two = mapped_hip_runtimes()
    if len(two) > 1:
        raise ImportError("two HIP runtimes are mapped into this process (%s): libemgpu.so and torch would each talk to the GPU through its own; "
                          "import em_model_manned_bayes_amd before torch, or set EMGPU_HIP_RUNTIME=system in a process that never imports torch" % ", ".join(two))
    for s in SYMBOLS:
        getattr(L, s)  # AttributeError if the ABI is incomplete
    L.emgpu_last_error.restype = C.c_char_p
    L.emgpu_version.restype = C.c_char_p
    L.emgpu_last_kernel_name.restype = C.c_char_p
    L.emgpu_last_kernel_name.argtypes = [C.c_void_p]
    L.emgpu_last_launch_count.argtypes = [C.c_void_p]
    L.emgpu_last_launch_count.restype = C.c_int32
    L.emgpu_model_load_txt.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.emgpu_model_from_arrays.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
    L.emgpu_model_free.argtypes = [C.c_void_p]
    L.emgpu_model_free.restype = None
    L.emgpu_model_info.argtypes = [C.c_void_p, C.POINTER(ModelInfo)]
    L.emgpu_model_get_i32.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.emgpu_model_get_i32.restype = C.c_int64
    L.emgpu_model_get_f64.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
    L.emgpu_model_get_f64.restype = C.c_int64
    L.emgpu_model_get_text.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.emgpu_model_get_text.restype = C.c_int64
    L.emgpu_model_set_f64.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
    L.emgpu_model_set_prior.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    L.emgpu_model_set_transition_stay_prior.argtypes = [C.c_void_p, C.c_double]
    L.emgpu_model_set_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.emgpu_model_start_log_weight.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.emgpu_model_set_zero_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.emgpu_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.emgpu_device_count.argtypes = [C.POINTER(C.c_int32)]
    L.emgpu_mixed_blocks.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.POINTER(Block)]
    L.emgpu_mixed_blocks.restype = C.c_int32
    L.emgpu_sample_dbn_blocks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(SampleParams), C.POINTER(Block), C.c_int32,
                                                 C.POINTER(SampleOut)]
    L.emgpu_sample_dbn_multi_host.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(SampleParams), C.POINTER(SampleOut)]
    L.emgpu_sample_dbn_multi_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(SampleParams), C.POINTER(SampleOut)]
    for f in (L.emgpu_track_uncor_host, L.emgpu_track_uncor_device):
        f.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(UTrackParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_track_terminal_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(TTrackParams), C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_uncor_dynamic_limits.argtypes = [C.c_void_p, C.POINTER(UTrackParams), C.c_void_p] + [C.c_double] * 4 + [C.c_void_p]
    L.emgpu_ctx_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    L.emgpu_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.emgpu_ctx_sync.argtypes = [C.c_void_p]
    L.emgpu_ctx_free.argtypes = [C.c_void_p]
    L.emgpu_ctx_free.restype = None
    L.emgpu_sample_dbn_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SampleParams), C.POINTER(SampleOut)]
    L.emgpu_sample_dbn_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SampleParams), C.POINTER(SampleOut)]
    L.emgpu_sample_bn_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BnParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_sample_bn_host.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BnParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_discretize_bayes.argtypes = [C.c_double, C.c_void_p, C.c_int32]
    L.emgpu_discretize_bayes.restype = C.c_int32
    L.emgpu_asub2ind.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.emgpu_asub2ind.restype = C.c_int64
    L.emgpu_debug_column_thresholds.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.emgpu_debug_bernoulli_threshold.argtypes = [C.c_double]
    L.emgpu_debug_dynamic_column.argtypes = [C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 7
    L.emgpu_debug_padded_column.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
    L.emgpu_debug_pk_column.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    L.emgpu_debug_bernoulli_threshold.restype = C.c_uint32
    L.emgpu_debug_parent_masks.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for f in (L.emgpu_propagate_terminal_device, L.emgpu_propagate_terminal_host):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(TermParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_philox_rounds.restype = C.c_int32
    L.emgpu_slot_map_revision.restype = C.c_int32
    L.emgpu_ctx_trim.argtypes = [C.c_void_p]
    L.emgpu_model_save_bin.argtypes = [C.c_void_p, C.c_char_p]
    L.emgpu_model_load_bin.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.emgpu_debug_uncor_dynamics_host.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emgpu_debug_terminal_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.emgpu_sample_terminal_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(TSampleParams)] + [C.c_void_p] * 7
    for f in (L.emgpu_sample2track_device, L.emgpu_sample2track_host):
        f.argtypes = [C.c_void_p, C.POINTER(TrackParams)] + [C.c_void_p] * 6
    L.emgpu_trace_alloc.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SampleParams), C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]
    L.emgpu_trace_out.argtypes = [C.c_void_p, C.POINTER(SampleOut)]
    L.emgpu_trace_report.argtypes = [C.c_void_p, C.POINTER(TraceReport)]
    L.emgpu_trace_free.argtypes = [C.c_void_p, C.c_void_p]
    L.emgpu_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.emgpu_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.emgpu_host_stats.argtypes = [C.c_void_p, C.POINTER(HostStats)]
    L.emgpu_device_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.emgpu_device_free.argtypes = [C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise EmgpuError(int(rc), lib().emgpu_last_error().decode("utf-8", "replace"))
    return rc
