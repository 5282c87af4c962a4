"""tools/host_path_probe.py [n] -- emgpu_sample_dbn_host at n x 240 s of uncor_1200code_v2p1 under the pipeline's knobs (one child process per
setting: the knobs are read once): EMGPU_HOST_DIRECT (pinned outputs: rows | 2d), EMGPU_HOST_CHUNK_MB, EMGPU_HOST_THREADS."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, json, sys, tempfile, time
sys.path.insert(0, %(root)r)
import numpy as np
from em_model_manned_bayes_amd import native, em_io, _lib as L
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp(prefix="emgpu_hp_")))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = {k: labels.index('"%%s"' %% v) + 1 for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
ctx = native.Context(0)
n, T = %(n)d, 240
res = {}
for rep in range(3):
    r = native.sample_dbn_host(ctx, nm, n, T, 1, want_dense=True, want_events=False, pinned=True, raw=True, **idx); st = r["host_stats"]; del r
res["pinned"] = st
ni, nd, G4 = nm.n_initial, nm.n_dyn, 60
ib, iv = np.zeros((ni, n), np.uint8), np.zeros((ni, n), np.float32)
db, dv = np.zeros((G4, nd, n), np.uint32), np.zeros((G4, nd, n, 4), np.float32)
p, _ = native.make_params(n, T, 1, **idx)
o = L.SampleOut(); o.init_bin, o.init_val, o.dyn_bin, o.dyn_val = ib.ctypes.data, iv.ctypes.data, db.ctypes.data, dv.ctypes.data
for rep in range(3):
    L.check(L.lib().emgpu_sample_dbn_host(ctx._h, nm._h, C.byref(p), C.byref(o)))
res["pageable"] = ctx.host_stats()
print("RESULT " + json.dumps(res))
"""
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
settings = [{}, {"EMGPU_HOST_DIRECT": "2d"}, {"EMGPU_HOST_CHUNK_MB": "4096"}, {"EMGPU_HOST_CHUNK_MB": "512"}, {"EMGPU_HOST_CHUNK_MB": "128"},
            {"EMGPU_HOST_CHUNK_MB": "64"}, {"EMGPU_HOST_THREADS": "4"}, {"EMGPU_HOST_THREADS": "16"}, {"EMGPU_HOST_THREADS": "16", "EMGPU_HOST_CHUNK_MB": "128"},
            {"EMGPU_HOST_THREADS": "2"}]
for sset in settings:
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, n=n)], capture_output=True, env=dict(os.environ, **sset), timeout=600)
    got = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("RESULT ")]
    if not got:
        print(sset, "FAILED", r.stderr.decode()[-800:]); continue
    d = json.loads(got[-1][7:])
    for kind in ("pinned", "pageable"):
        st = d[kind]
        print("%-60s %-8s total %7.1f ms  %5.1f GB/s  kernel %5.1f  d2h %6.1f  scatter %6.1f  chunks %3d x %7d  threads %d" % (
            json.dumps(sset), kind, st["total_ms"], st["bytes_d2h"] / st["total_ms"] / 1e6, st["kernel_ms"], st["d2h_ms"], st["scatter_ms"], st["chunks"], st["chunk_n"], st["threads"]))
    sys.stdout.flush()
