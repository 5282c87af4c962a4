"""tools/bench_terminal.py -- throughput of terminal trajectory propagation (BASELINE.json configs[4] in
miniature): n encounters x 4 tracks x <=121 s on one GPU, synthetic trajectory models, device-resident output."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import synthetic, native, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.write_terminal_directory(tempfile.mkdtemp())
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=d)
m = 4096
_, samples = t.sample(m, seed=1, ctx=ctx)
g, mo = t._geo_rows(samples)
reps = (n + m - 1) // m
geo = torch.tensor(np.tile(g, (reps, 1))[:n], device=dev)
mof = torch.tensor(np.tile(mo, (reps, 1))[:n].reshape(-1), dtype=torch.int32, device=dev)
cap = 123
out = torch.empty((6, cap, 4 * n), dtype=torch.float32, device=dev)
rows = torch.empty(4 * n, dtype=torch.int32, device=dev)
p = L.TermParams()
p.seed, p.first_index, p.n, p.tmax_s, p.max_resample, p.cap = 7, 0, n, 120.0, 100000, cap
for i, v in enumerate(t._dyn_rows().reshape(-1)):
    p.dyn_limits[i] = float(v)
handles = (C.c_void_p * 10)(*[x.native._h for x in t._traj])
def run():
    L.check(L.lib().emgpu_propagate_terminal_device(ctx._h, handles, 10, C.byref(p), C.c_void_p(geo.data_ptr()), C.c_void_p(mof.data_ptr()),
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(rows.data_ptr())))
run(); ctx.sync(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); run(); run(); b.record(); torch.cuda.synchronize(); ctx.sync()
ms = a.elapsed_time(b) / 3
secs = int(rows.clamp(min=0).sum().item())
print("terminal propagation: %d encounters in %.2f ms -> %.3e encounters/s, %.3e track-seconds/s, %.1f GB/s written (24 B per second)"
      % (n, ms, n / ms * 1e3, secs / ms * 1e3, secs * 24 / ms / 1e6))
