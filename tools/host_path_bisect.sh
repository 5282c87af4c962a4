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
TAG="raw Trace on torch stream ctx, freed " MODE=rawtrace one --no-cpu-baseline
TAG="raw Trace on own-stream ctx, freed   " MODE=rawtrace_plain one --no-cpu-baseline
TAG="raw Trace kept alive                 " MODE=rawtrace_keep one --no-cpu-baseline
TAG="EMGPU_TRACE_ALLOC=plain, workload    " EMGPU_TRACE_ALLOC=plain MODE=alloc one --no-cpu-baseline --placement-candidates 1
