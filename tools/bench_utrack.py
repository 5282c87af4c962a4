"""tools/bench_utrack.py [n] [T] -- UncorEncounterModel.track end to end on one GPU (emgpu_track_uncor_device): rounds of
sample -> k_uncor_track (point-mass dynamics + getDynamicLimits rejection) -> compaction, 1 Hz tracks resident in HBM."""
import ctypes as C, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import em_io, native, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 240
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp()))
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
p = native.utrack_params(nm, n, T, 0x5EED0002, record_stride=10)
S = 10 * T // 10 + 1
tracks = torch.empty((n, S, 8), dtype=torch.float64, device=dev)
att = torch.empty(n, dtype=torch.int32, device=dev)
def run():
    L.check(L.lib().emgpu_track_uncor_device(ctx._h, nm._h, C.byref(p), C.c_void_p(tracks.data_ptr()), None, C.c_void_p(att.data_ptr())))
run()
t0 = time.perf_counter(); run(); run(); dt = (time.perf_counter() - t0) / 2
print("uncor track: %d trajectories x %d s (10 Hz dynamics, 1 Hz tracks kept): %.1f ms -> %.3e tracks/s; attempts mean %.3f max %d; %s"
      % (n, T, dt * 1e3, n / dt, float(att.float().mean()), int(att.max()), ctx.last_kernel()))
