import ctypes as C, json, sys, tempfile, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
mode = sys.argv[1]
from em_model_manned_bayes_amd import native, em_io, _lib as L
if mode in ("torch", "torch_stream", "torch_many", "torch_copy", "torch_copy_after_ctx"):
    import torch
    torch.cuda.init(); x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp(prefix="emgpu_hp_")))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = {k: labels.index('"%s"' % v) + 1 for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
extra = []
if mode == "torch_many":
    for q in range(6):
        c = native.Context(0); p, _ = native.make_params(100000, 240, 1, **idx); t = native.Trace(c, nm, p, candidates=1)
        native.sample_dbn_device(c, nm, p, **{k: v for k, v in t.ptrs().items() if k in ("init_bin","init_val","dyn_bin","dyn_val","ld")}); c.sync(); extra.append((c, t))
    r0 = native.sample_dbn_host(extra[0][0], nm, 100000, 240, 1, want_dense=True, **idx); del r0
    extra = None
def torch_copy():
    d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); h = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); h.copy_(d, non_blocking=True); b.record(); torch.cuda.synchronize()
    print("torch pinned copy %.1f GB/s" % ((1 << 30) / a.elapsed_time(b) / 1e6))
if mode == "torch_copy":
    torch_copy()
if mode == "torch_stream":
    ctx = native.Context(0, stream=torch.cuda.current_stream().cuda_stream)
else:
    ctx = native.Context(0)
if mode == "torch_copy_after_ctx":
    r = native.sample_dbn_host(ctx, nm, 100000, 240, 1, want_dense=True, want_events=False, pinned=True, raw=True, **idx); del r
    torch_copy()
n, T = 1000000, 240
for rep in range(3):
    r = native.sample_dbn_host(ctx, nm, n, T, 1, want_dense=True, want_events=False, pinned=True, raw=True, **idx); st = r["host_stats"]; del r
print(mode, "pinned  %.1f GB/s d2h %.1f ms" % (st["bytes_d2h"] / st["total_ms"] / 1e6, st["d2h_ms"]))
