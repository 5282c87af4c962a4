"""tools/copy_placement_probe.py -- does WHERE a device buffer lies decide how fast the copy engine reads it?  (Round 6: the host path's copies run at
47 / 40 GB/s instead of 56.5 / 54.7 in a process that allocated and freed a 36 GB trace before.)  Times device -> pinned copies of buffers allocated
(a) in a fresh process, (b) right after a 36 GB trace of the library was freed: plain hipMalloc blocks of 290 MB and 1.2 GB, library blocks
(emgpu_device_alloc: one address range over 1 GiB chunks) of 1.2 GB.
Also the reproducer of the HIP runtime crash the allocator works round (SKIP=hipMalloc EMGPU_VMM_FREE_VA=1: segfault in hipMemMap under the PyTorch
wheel's bundled runtime)."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from em_model_manned_bayes_amd import native, em_io, _lib as L
hip = None
def setup():
    global hip
    nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp(prefix="emgpu_cp_")))
    labels = nm.get_labels(L.F_LABELS_INITIAL)
    idx = {k: labels.index('"%s"' % v) + 1 for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
    ctx = native.Context(0)
    hip = C.CDLL(None)
    return nm, idx, ctx
def copy_gbps(dptr, nbytes, hptr, stream, ev):
    best = 0.0
    for rep in range(3):
        hip.hipEventRecord(ev[0], stream)
        assert hip.hipMemcpyAsync(C.c_void_p(hptr), C.c_void_p(dptr), C.c_size_t(nbytes), 2, stream) == 0
        hip.hipEventRecord(ev[1], stream); hip.hipEventSynchronize(ev[1])
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]); best = max(best, nbytes / ms.value / 1e6)
    return best
def round_of(tag, ctx, hptr, stream, ev):
    out = []
    kinds = (("hipMalloc 290 MB", 290 << 20), ("hipMalloc 1.2 GB", 1200 << 20), ("library 1.2 GB", 1200 << 20))
    skip = os.environ.get("SKIP", "")
    kinds = tuple(k for k in kinds if not (skip and k[0].startswith(skip)))
    for kind, nbytes in kinds:
        vals = []
        held = []
        for k in range(4):
            if kind.startswith("hipMalloc"):
                p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0; addr = p.value
            else:
                addr = ctx.device_alloc(nbytes)
            held.append((kind, addr))
            vals.append(copy_gbps(addr, nbytes, hptr, stream, ev) if not os.environ.get("NOCOPY") else 0.0)
        for kind_, addr in held:
            if kind_.startswith("hipMalloc"): hip.hipFree(C.c_void_p(addr))
            else: ctx.device_free(addr)
        out.append("%s: %s" % (kind, " ".join("%.1f" % v for v in vals)))
    print(tag, "|", " | ".join(out), flush=True)
nm, idx, ctx = setup()
stream = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0
ev = [C.c_void_p(), C.c_void_p()]
for e in ev: hip.hipEventCreate(C.byref(e))
hp = C.c_void_p(); assert hip.hipHostMalloc(C.byref(hp), C.c_size_t(1300 << 20), 0) == 0
round_of("fresh process              ", ctx, hp.value, stream, ev)
p, _ = native.make_params(10_000_000, 240, 1, **idx)
t = native.Trace(ctx, nm, p, candidates=1)
native.sample_dbn_device(ctx, nm, p, **{k: v for k, v in t.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")}); ctx.sync()
round_of("while a 36 GB trace is held", ctx, hp.value, stream, ev)
if os.environ.get("PTRATTR"):
    buf = (C.c_char * 256)()
    for k in ("dyn_val", "dyn_bin", "init_val", "init_bin"):
        rc = hip.hipPointerGetAttributes(buf, C.c_void_p(t.ptrs()[k]))
    round_of("after hipPointerGetAttributes on the trace's arrays (rc %d)" % rc, ctx, hp.value, stream, ev)
if os.environ.get("TORCHWRAP"):
    import torch
    class _M:
        __cuda_array_interface__ = {"shape": (60, 3, t.ld, 4), "typestr": "<f4", "data": (t.ptrs()["dyn_val"], False), "version": 2}
    x = torch.as_tensor(_M(), device="cuda:0")
    round_of("after torch.as_tensor over dyn_val", ctx, hp.value, stream, ev)
    del x
hp2 = C.c_void_p(); assert hip.hipHostMalloc(C.byref(hp2), C.c_size_t(1300 << 20), 0) == 0
round_of("... into pinned memory allocated NOW", ctx, hp2.value, stream, ev)
t.free(); ctx.trim()
round_of("right after it was freed   ", ctx, hp.value, stream, ev)
hp3 = C.c_void_p(); assert hip.hipHostMalloc(C.byref(hp3), C.c_size_t(1300 << 20), 0) == 0
round_of("... into pinned memory allocated NOW", ctx, hp3.value, stream, ev)
st2 = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(st2), 1) == 0
round_of("... on a stream created NOW", ctx, hp.value, st2, ev)
time.sleep(8)
round_of("8 s later                  ", ctx, hp.value, stream, ev)
