"""tools/stream_probe.py [n_streams] -- the headline kernel (uncor_1200code_v2p1, 10 M x 240 s) timed on the default stream and on several
streams of ONE process: does the launch time depend on the hardware queue a stream is bound to?  (Round 5: the "slow box state" of
HISTORY.md section 7 alternates from PROCESS to process on one box -- profiles/r05_queue_probe.txt.)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import native, em_io, _lib as L
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tmp))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = dict(idx_L=labels.index('"L"') + 1, idx_v=labels.index('"v"') + 1, idx_dh=labels.index('"\\dot h"') + 1)
n, T = 10_000_000, 240
ld = -(-n // 1024) * 1024
ib = torch.empty((7, ld), dtype=torch.uint8, device=dev); iv = torch.empty((7, ld), dtype=torch.float32, device=dev)
db = torch.empty((60, 3, ld), dtype=torch.int32, device=dev); dv = torch.empty((60, 3, ld, 4), dtype=torch.float32, device=dev)
streams = [("default", torch.cuda.current_stream(dev))] + [("s%d" % i, torch.cuda.Stream(dev)) for i in range(ns)]
ctx = native.Context(0)
def run(stream, k):
    ctx.set_stream(stream.cuda_stream)
    p, _ = native.make_params(n, T, 0x5EED0002, first_index=k * n, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr(), ld=ld)
for cycle in range(int(os.environ.get("CYCLES", "2"))):
    for name, st in streams:
        for k in range(6):
            run(st, k)
        ctx.sync()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for k in range(12):
            run(st, 100 + k)
        b.record(st)
        ctx.sync()
        print("cycle %d stream %-8s %.3f ms per launch  (%s)" % (cycle, name, a.elapsed_time(b) / 12, ctx.last_kernel()), flush=True)
