"""tools/placement_probe2.py [sets] -- the headline kernel timed on several separately allocated copies of its trace buffers inside ONE
process: does the launch time depend on WHERE the 36 GB trace lies?  (Round 5: the "slow box state" of HISTORY.md section 7 alternates
from process to process on some boxes: profiles/r05_queue_probe.txt.)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import native, em_io, _lib as L
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tmp))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = dict(idx_L=labels.index('"L"') + 1, idx_v=labels.index('"v"') + 1, idx_dh=labels.index('"\\dot h"') + 1)
n, T = 10_000_000, 240
ld = -(-n // 1024) * 1024
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bufs = []
for s_ in range(sets):
    bufs.append((torch.empty((7, ld), dtype=torch.uint8, device=dev), torch.empty((7, ld), dtype=torch.float32, device=dev),
                 torch.empty((60, 3, ld), dtype=torch.int32, device=dev), torch.empty((60, 3, ld, 4), dtype=torch.float32, device=dev)))
def run(b, k):
    p, _ = native.make_params(n, T, 0x5EED0002, first_index=k * n, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=b[0].data_ptr(), init_val=b[1].data_ptr(), dyn_bin=b[2].data_ptr(), dyn_val=b[3].data_ptr(), ld=ld)
for cycle in range(int(os.environ.get("CYCLES", "2"))):
    for s_, b in enumerate(bufs):
        for k in range(6):
            run(b, k)
        ctx.sync()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(12):
            run(b, 100 + k)
        e.record()
        ctx.sync()
        print("cycle %d buffers %d (dyn_val at 0x%x) %.3f ms per launch" % (cycle, s_, b[3].data_ptr(), a.elapsed_time(e) / 12), flush=True)
