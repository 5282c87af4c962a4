"""tools/dbg_counters.py [model ...] -- wave-level counts of the data-dependent paths of k_uncor_fast (diagnostic build:
tools/build_variant.sh dbg -DEMGPU_DEBUG_COUNTERS; run with EMGPU_LIB=tools/ab/dbg.so)."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from em_model_manned_bayes_amd import em_io, native, _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n, T = 1_000_000, 240
tmp = tempfile.mkdtemp()
for name in sys.argv[1:] or ["uncor_1200code_v2p1", "uncor_1200only_fwse_v1p2"]:
    if name == "cor_v2p1_like":
        from em_model_manned_bayes_amd import synthetic
        m = native.NativeModel.load_txt(synthetic.write_correlated_v2p1_like(tmp))
    else:
        m = native.NativeModel.load_txt(em_io.materialize_model(name, tmp))
    labs = m.get_labels(L.F_LABELS_INITIAL)
    idx = {k: (labs.index('"%s"' % v) + 1 if '"%s"' % v in labs else 0) for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
    nd = m.n_dyn
    db = torch.empty((T // 4, nd, n), dtype=torch.int32, device=dev); dv = torch.empty((T // 4, nd, n, 4), dtype=torch.float32, device=dev)
    p, _ = native.make_params(n, T, 5, **idx)
    out = (C.c_ulonglong * 8)()
    lib.emgpu_debug_counters(out, 1); lib.emgpu_debug_counters_step2(out, 1); lib.emgpu_debug_counters_step2b(out, 1)
    native.sample_dbn_device(ctx, m, p, dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
    ctx.sync()
    if "step2" in ctx.last_kernel():   # the kernel's instances live in two translation units, each with its own counters
        lib.emgpu_debug_counters_step2(out, 1)
        more = (C.c_ulonglong * 8)()
        lib.emgpu_debug_counters_step2b(more, 1)
        for q in range(8):
            out[q] += more[q]
    else:
        lib.emgpu_debug_counters(out, 1)
    blocks = out[5] or 1
    print("%s %s: per wave-block: exact redos %.3f, compaction rounds %.3f, compaction steps %.2f, worker passes %.3f, requests %.1f"
          % (name, ctx.last_kernel(), out[0] / blocks, out[1] / blocks, out[2] / blocks, out[3] / blocks, out[4] / blocks))
