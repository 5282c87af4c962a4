"""tools/bench_events.py [model ...] -- the event-list output (what UncorEncounterModel.sample returns: [dt, variable, value, bin] rows per
trajectory, dbn_hierarchical_sample.m:33-60) on one GPU, events only: 1 M trajectories x 240 s, event_cap 512."""
import sys, time, tempfile
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
from em_model_manned_bayes_amd import em_io, native, _lib as L
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
def bench(name):
    nm = native.NativeModel.load_txt(em_io.materialize_model(name, tempfile.mkdtemp()))
    labs = nm.get_labels(L.F_LABELS_INITIAL)
    idx = {k: (labs.index('"%s"' % v) + 1 if '"%s"' % v in labs else 0) for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
    n, T = 1_000_000, 240
    for cap in ((int(os.environ["EVCAP"]),) if "EVCAP" in os.environ else (512, 2048)):                  # (haa_v1: seven rated variables, lists of up to ~700 rows)
        try:
            return bench_cap(name, nm, idx, n, T, cap)
        except L.EmgpuError as e:
            if e.code != L.ERR_EVENT_CAP or cap == 2048:
                raise
            ctx.trim()


def bench_cap(name, nm, idx, n, T, cap):
    ni = nm.n_initial
    ib = torch.empty((ni, n), dtype=torch.uint8, device=dev); iv = torch.empty((ni, n), dtype=torch.float32, device=dev)
    evc = torch.empty(n, dtype=torch.int32, device=dev); ev = torch.empty((n, cap, 2), dtype=torch.float32, device=dev)
    p, _ = native.make_params(n, T, 5, event_cap=cap, **idx)
    def run():
        native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), ev_count=evc.data_ptr(), events=ev.data_ptr())
    run(); ctx.sync()
    t0 = time.perf_counter(); run(); run(); ctx.sync(); dt = (time.perf_counter() - t0) / 2
    print("event-list path, %s: %d x %d s in %.2f ms -> %.3e traj/s, kernel %s, mean events %.1f, max %d" % (name, n, T, dt * 1e3, n / dt, ctx.last_kernel(), float(evc.float().mean()), int(evc.max())))


for name in (sys.argv[1:] or ["uncor_1200code_v2p1"]):
    bench(name)
