"""tools/placement_probe3.py [sets] -- what distinguishes a slow trace placement (profiles/r05_placement_probe.txt)?  For each of several
separately allocated traces: the headline kernel's launch time, a plain fill of the value plane (sequential writes), and two strided
writes that touch one float every 4 KiB / every 2 MiB (one access per page / per 2 MiB fragment: sensitive to the page-table walk, not to
bandwidth).  A placement that is slow for the kernel AND for the strided writes but not for the fill points at address translation."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import native, em_io, _lib as L
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp()))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = dict(idx_L=labels.index('"L"') + 1, idx_v=labels.index('"v"') + 1, idx_dh=labels.index('"\\dot h"') + 1)
n, T = 10_000_000, 240
ld = -(-n // 1024) * 1024
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bufs = [(torch.empty((7, ld), dtype=torch.uint8, device=dev), torch.empty((7, ld), dtype=torch.float32, device=dev),
         torch.empty((60, 3, ld), dtype=torch.int32, device=dev), torch.empty((60, 3, ld, 4), dtype=torch.float32, device=dev)) for _ in range(sets)]
def run(b, k):
    p, _ = native.make_params(n, T, 0x5EED0002, first_index=k * n, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=b[0].data_ptr(), init_val=b[1].data_ptr(), dyn_bin=b[2].data_ptr(), dyn_val=b[3].data_ptr(), ld=ld)
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for k in range(40):          # load the device first
    run(bufs[-1], k)
ctx.sync()
for cycle in range(2):
    for s_, b in enumerate(bufs):
        flat = b[3].view(-1)
        for k in range(4):
            run(b, k)
        ctx.sync()
        kms = timed(lambda: [run(b, 100 + q) for q in range(4)], 3) / 4
        fill = timed(lambda: b[3].zero_(), 5)
        p4k = timed(lambda: flat[::1024].fill_(1.0), 5)
        p2m = timed(lambda: flat[::524288].fill_(1.0), 20)
        print("cycle %d trace %d (dyn_val 0x%x): kernel %.3f ms | fill %.2f TB/s | one float per 4 KiB: %.3f ms (%.1f M pages) | per 2 MiB: %.4f ms"
              % (cycle, s_, b[3].data_ptr(), kms, flat.numel() * 4 / fill / 1e9, p4k, flat.numel() / 1024 / 1e6, p2m), flush=True)
