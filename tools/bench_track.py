"""tools/bench_track.py [n] [T] -- the sampler and its device consumer back to back on one GPU:
k_uncor_fast writes the dense trace of n trajectories x T seconds, k_sample2track<dense> reads it in place
(sample2track.m:183-243) and writes the f64 track + rejection flags.  Everything stays in HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tempfile
import numpy as np
import torch
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import em_io, native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 240
parms = E.em_read(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp()))
nm = parms["native"]
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ni, nd, G4 = 7, 3, (T + 3) // 4
iv = torch.empty((ni, n), dtype=torch.float32, device=dev)
ib = torch.empty((ni, n), dtype=torch.uint8, device=dev)
dv = torch.empty((G4, nd, n, 4), dtype=torch.float32, device=dev)
db = torch.empty((G4, nd, n), dtype=torch.int32, device=dev)
xyz = torch.empty((T + 1, 3, n), dtype=torch.float64, device=dev)
fl = torch.empty(n, dtype=torch.uint8, device=dev)
labs = parms["labels_initial"]
iL, iV, iDH = labs.index('"L"') + 1, labs.index('"v"') + 1, labs.index('"\\dot h"') + 1
p, _k = native.make_params(n, T, 1, idx_L=iL, idx_v=iV, idx_dh=iDH)
tm = np.asarray(parms["temporal_map"]).reshape(-1, 2)
lab = [labs[int(r[0]) - 1] for r in tm]
bv = np.asarray(parms["boundaries"][iV - 1])
tp = native.track_params(n, T, (1852.0 / 0.3048) / 3600.0, 1.0 / 60.0, 1.0, float(bv[0]), float(bv[-1]), nd=nd,
                         slot_vertrate=lab.index('"\\dot h"'), slot_acc=lab.index('"\\dot v"'), slot_turnrate=lab.index('"\\dot \\psi"'))

def sample():
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())

def track():
    native.sample2track_device(ctx, tp, iv[iL - 1].data_ptr(), iv[iV - 1].data_ptr(), dv.data_ptr(), xyz.data_ptr(), fl.data_ptr())

def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize(); ctx.sync()
    return a.elapsed_time(b) / reps

ms_s = timed(sample); k_s = ctx.last_kernel(); ms_t = timed(track)
good = int((fl == 0).sum().item())
rd, wr = n * (T * 12 + 8), n * ((T + 1) * 24 + 1)
print("sampler %s: %.2f ms; k_sample2track<dense>: %.2f ms (%.3e tracks/s, %.0f GB/s read+written of %d B/track); accepted %.1f %%; "
      "pipeline %.3e tracks/s" % (k_s, ms_s, ms_t, n / ms_t * 1e3, (rd + wr) / ms_t / 1e6, (rd + wr) // n,
                                   100.0 * good / n, n / (ms_s + ms_t) * 1e3))
