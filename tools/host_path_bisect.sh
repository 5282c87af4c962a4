#!/bin/bash
# tools/host_path_bisect.sh -- which phase of bench.py makes its host_path copies slower (47 GB/s) than the same call in a fresh process (56.5)?
cd "$GRAFT_REPO_ROOT"
one() { python - "$@" <<'PY'
import sys, json, copy
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
args = bench.parse_args(sys.argv[1:])
pl = bench.TorchRocm(0, 0, 1)
detail = {"host_path": {}}
if "--skip-headline" not in " ".join(sys.argv):
    pass
import os
mode = os.environ.get("MODE", "full")
if mode in ("rawtrace", "rawtrace_plain", "rawtrace_keep"):
    from em_model_manned_bayes_amd import native, _lib as L
    import tempfile
    nm = native.NativeModel.load_txt(bench._materialize("uncor_1200code_v2p1", tempfile.mkdtemp()))
    c = pl.context() if mode != "rawtrace_plain" else native.Context(0)
    p, _k = native.make_params(10_000_000, 240, 1, **bench._label_indices(nm))
    t = native.Trace(c, nm, p, candidates=1)
    wrapped = []
    if os.environ.get("WRAP"):
        ld, ptrs = t.ld, t.ptrs()
        shapes = {"init_bin": ((7, ld), "uint8"), "init_val": ((7, ld), "float32"), "dyn_bin": ((60, 3, ld), "int32"), "dyn_val": ((60, 3, ld, 4), "float32")}
        for k in os.environ["WRAP"].split(","):
            wrapped.append(pl.wrap(ptrs[k], *shapes[k]))
        if os.environ.get("TOUCH"):
            print("touch", int(wrapped[0].reshape(-1)[:1000].sum()))
        if os.environ.get("DROP"):
            wrapped = []
            import gc; gc.collect()
    if mode != "rawtrace_keep":
        t.free(); c.trim()
elif os.environ.get("HEADLINE", "1") == "1":
    w = bench.make_workload(args, pl, 0, 1)
    if mode == "full":
        bench.measure(w, pl, args, args.warmup, args.steps)
        if os.environ.get("SW", "1") == "1":
            w.streaming_write()
    elif mode.startswith("steps"):
        for k in range(int(mode[5:])):
            w.step(k)
        w.sync()
        if "check" in os.environ.get("EXTRA", ""):
            w.check()
    w.close(); w = None; pl.release()
h = bench.host_path(args, pl, detail)
print(os.environ.get("TAG", ""), "pinned", h["dense_pinned"]["GBps"], "pageable", h["dense_pageable"]["GBps"], "plain copy", h["pinned_d2h_GBps"])
PY
}
TAG="raw trace + wrap 4 arrays (kept)     " MODE=rawtrace_keep WRAP=init_bin,init_val,dyn_bin,dyn_val one --no-cpu-baseline
TAG="raw trace + wrap dyn_val only (kept)  " MODE=rawtrace_keep WRAP=dyn_val one --no-cpu-baseline
TAG="raw trace + wrap 4, dropped, freed    " MODE=rawtrace WRAP=init_bin,init_val,dyn_bin,dyn_val DROP=1 one --no-cpu-baseline
TAG="raw trace + wrap init_bin only (kept) " MODE=rawtrace_keep WRAP=init_bin one --no-cpu-baseline
