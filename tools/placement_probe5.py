"""tools/placement_probe5.py [K] -- which ALLOCATION gives a fast trace?  Round 6: the same launch writes a 36 GB trace in 5.9, 6.6 or 7.0 ms
depending on the allocation (three classes on one box, the fast one rare).  One child process per allocation mode of the library's trace
pool (EMGPU_TRACE_ALLOC = plain | contiguous | vmm:<MiB>), each allocating K traces through the C ABI (emgpu_trace_alloc, candidates = 1),
timing the headline launch on each (two rounds of 2 + 5 launches, HIP events) and printing the times.  -> profiles/r06_placement_probe.txt
(The recorded runs also show "vmm:<MiB>:<shift>" and "vmmu:" modes: diagnostic builds of round 6 that mapped the chunks off the start of the range /
placed the block by hand on a 1 GiB boundary.  Neither makes a difference and neither is in the library.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes as C, json, sys, time, tempfile
sys.path.insert(0, %(root)r)
from em_model_manned_bayes_amd import native, em_io, _lib as L
nm = native.NativeModel.load_txt(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp(prefix="emgpu_pp5_")))
labels = nm.get_labels(L.F_LABELS_INITIAL)
idx = {k: labels.index('"%%s"' %% v) + 1 for k, v in (("idx_L", "L"), ("idx_v", "v"), ("idx_dh", "\\dot h"))}
ctx = native.Context(0); ctx.set_stream(0)
p, _ = native.make_params(%(n)d, 240, 0x5EED0002, **idx)
hip = C.CDLL(None)
ev = [C.c_void_p(), C.c_void_p()]
for e in ev: assert hip.hipEventCreate(C.byref(e)) == 0
def ptrs(tr): return {k: v for k, v in tr.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")}
def ms_per_launch(tr, warm=2, timed=5, rounds=2):
    best = 1e9
    for rnd in range(rounds):
        for k in range(warm): native.sample_dbn_device(ctx, nm, p, **ptrs(tr))
        hip.hipEventRecord(ev[0], None)
        for k in range(timed): native.sample_dbn_device(ctx, nm, p, **ptrs(tr))
        hip.hipEventRecord(ev[1], None); hip.hipEventSynchronize(ev[1])
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]); best = min(best, ms.value / timed)
    return best
traces, t_alloc = [], []
for k in range(%(K)d):
    t0 = time.time()
    try:
        traces.append(native.Trace(ctx, nm, p, candidates=1))
    except L.EmgpuError as e:
        print("allocation %%d failed: %%s" %% (k, e)); break
    t_alloc.append(round((time.time() - t0) * 1e3, 1))
t_end = time.time() + 0.5
while time.time() < t_end: ms_per_launch(traces[-1], 0, 4, 1)
res = []
for cyc in range(2):
    res.append([round(ms_per_launch(t), 3) for t in traces])
print("RESULT " + json.dumps({"mode": %(mode)r, "alloc_ms": t_alloc, "dyn_val": ["0x%%x" %% t.ptrs()["dyn_val"] for t in traces], "ms": res}))
"""

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
modes = sys.argv[2:] or ["plain", "plain", "contiguous", "vmm:1024", "vmm:64", "vmm:4096", "plain"]
for mode in modes:
    env = dict(os.environ, EMGPU_TRACE_ALLOC=mode)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, n=10_000_000, K=K, mode=mode)], capture_output=True, env=env, timeout=600)
    out = r.stdout.decode()
    got = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
    if got:
        d = json.loads(got[-1][7:])
        print("%-12s alloc ms %s" % (mode, d["alloc_ms"]))
        for cyc, row in enumerate(d["ms"]):
            print("%-12s cycle %d  ms per launch  %s" % (mode, cyc, "  ".join("%.3f" % x for x in row)))
        print("%-12s dyn_val at %s" % (mode, " ".join(d["dyn_val"])))
    else:
        print("%-12s FAILED rc=%d\n%s\n%s" % (mode, r.returncode, out[-500:], r.stderr.decode()[-1500:]))
    sys.stdout.flush()
