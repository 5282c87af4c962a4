"""tools/placement_probe4.py -- does ONE large allocation avoid the slow ranges?  Five traces allocated one tensor at a time (as torch does
it: four hipMallocs each) against five traces carved out of a single 185 GB allocation, the headline kernel timed on each."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import native, em_io, _lib as L
dev = torch.device("cuda", 0)
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp()))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = dict(idx_L=labels.index('"L"') + 1, idx_v=labels.index('"v"') + 1, idx_dh=labels.index('"\\dot h"') + 1)
n, T = 10_000_000, 240
ld = -(-n // 1024) * 1024
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
sizes = [7 * ld, 7 * ld * 4, 60 * 3 * ld * 4, 60 * 3 * ld * 16]           # init_bin, init_val, dyn_bin, dyn_val in bytes
def run(ptrs, k):
    p, _ = native.make_params(n, T, 0x5EED0002, first_index=k * n, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ptrs[0], init_val=ptrs[1], dyn_bin=ptrs[2], dyn_val=ptrs[3], ld=ld)
def timed(ptrs):
    for k in range(4):
        run(ptrs, k)
    ctx.sync()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(12):
        run(ptrs, 100 + k)
    e.record(); ctx.sync()
    return a.elapsed_time(e) / 12
def report(tag, sets):
    for k in range(40):
        run(sets[-1], k)
    ctx.sync()
    for cyc in range(2):
        print(tag, "cycle", cyc, " ".join("%.3f" % timed(s) for s in sets), flush=True)
order = sys.argv[1] if len(sys.argv) > 1 else "separate-first"
def separate():
    keep = [[torch.empty(sz, dtype=torch.uint8, device=dev) for sz in sizes] for _ in range(5)]
    report("five traces, one tensor at a time:", [[t.data_ptr() for t in s] for s in keep])
    del keep; torch.cuda.empty_cache()
def pooled():
    per = sum(-(-sz // 4096) * 4096 for sz in sizes)
    pool = torch.empty(5 * per, dtype=torch.uint8, device=dev)
    sets = []
    for s_ in range(5):
        off, ptrs = s_ * per, []
        for sz in sizes:
            ptrs.append(pool.data_ptr() + off); off += -(-sz // 4096) * 4096
        sets.append(ptrs)
    report("five traces inside ONE allocation:  ", sets)
    del pool; torch.cuda.empty_cache()
for what in ((separate, pooled) if order == "separate-first" else (pooled, separate)):
    what()
