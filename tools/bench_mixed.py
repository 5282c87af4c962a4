"""tools/bench_mixed.py [n_total] [gpus] -- BASELINE.json configs[3] on ONE GPU's share: a mixed batch over all
uncor_*_v1p2 model files (heterogeneous CPT shapes), n_total trajectories x 240 s sharded by sample index over
`gpus` ranks; this process plays rank 0 and runs its (model, first_index, count) launches back to back."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import em_io, native, sharding, _lib as L

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
gpus = int(sys.argv[2]) if len(sys.argv) > 2 else 8
names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, "models")) if f.startswith("uncor_") and f.endswith("_v1p2.npz"))
tmp = tempfile.mkdtemp()
models = [native.NativeModel.load_txt(em_io.materialize_model(nm, tmp)) for nm in names]
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
T = 240
# the two ranks in the middle of the index range own pieces of several models
best = None
for rank in range(gpus):
    lo, hi = sharding.shard_range(n_total, rank, gpus)
    calls = sharding.mixed_batch_blocks(n_total, len(models), lo, hi)
    if best is None or len(calls) > len(best[1]):
        best = (rank, calls, lo, hi)
rank, calls, lo, hi = best
n = hi - lo
ni, nd, G4 = 7, 3, T // 4

def idx_of(m):
    labs = m.get_labels(L.F_LABELS_INITIAL)
    f = lambda s: labs.index('"%s"' % s) + 1 if '"%s"' % s in labs else 0
    return dict(idx_L=f("L"), idx_v=f("v"), idx_dh=f("\\dot h"))

def run():
    # a launch of `count` trajectories addresses [.., count] arrays: every (model, block) has its own output buffers
    for (m, first, count) in calls:
        p, _ = native.make_params(count, T, 7, first_index=first, **idx_of(models[m]))
        native.sample_dbn_device(ctx, models[m], p, init_bin=bufs[m][0].data_ptr(), init_val=bufs[m][1].data_ptr(),
                                 dyn_bin=bufs[m][2].data_ptr(), dyn_val=bufs[m][3].data_ptr())

bufs = {}
for (m, first, count) in calls:
    bufs[m] = (torch.empty((ni, count), dtype=torch.uint8, device=dev), torch.empty((ni, count), dtype=torch.float32, device=dev),
               torch.empty((G4, nd, count), dtype=torch.int32, device=dev), torch.empty((G4, nd, count, 4), dtype=torch.float32, device=dev))
for _ in range(3):
    run()
torch.cuda.synchronize(); ctx.sync()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    run()
b.record(); torch.cuda.synchronize(); ctx.sync()
ms = a.elapsed_time(b) / 5
print("mixed batch: %d models %s; rank %d of %d owns [%d, %d) = %d trajectories in %d launches (%s): %.2f ms -> %.3e trajectories/s per GPU, "
      "%.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (len(models), [n_[6:] for n_ in names], rank, gpus, lo, hi, n, len(calls), ctx.last_kernel(), ms, n / ms * 1e3,
                                                     3635 * n / ms / 1e6, 3635 * n / ms / 1e6 / 80))
