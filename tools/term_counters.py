"""tools/term_counters.py [n] -- path counters of k_terminal_propagate (a -DEMGPU_TERM_COUNTERS build, EMGPU_LIB=tools/ab/<name>.so):
how many lanes of how many wave-iterations take each path of the loop."""
import os, sys, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import native, synthetic, _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=synthetic.write_terminal_directory(tempfile.mkdtemp()))
cap = 123
ni = t.native.n_initial
val = torch.empty((ni, n), dtype=torch.float32, device=dev); geo = torch.empty((n, 12), dtype=torch.float64, device=dev)
mof = torch.empty(4 * n, dtype=torch.int32, device=dev); traj = torch.empty((2 * n, 2 * native.terminal_t0_row(cap), 5), dtype=torch.float32, device=dev)
rows = torch.empty(4 * n, dtype=torch.int32, device=dev)
p, keep = native.terminal_sample_params(t.native, n, 0x5EED0005, t._dyn_rows(), cap=cap)
out = np.zeros(24, dtype=np.uint64)
L.lib().emgpu_debug_terminal_counters(ctx._h, out.ctypes.data_as(C.c_void_p), 24)
native.sample_terminal_device(ctx, t.native, [x.native for x in t._traj], p, val.data_ptr(), geo.data_ptr(), mof.data_ptr(), traj.data_ptr(), rows.data_ptr())
ctx.sync()
rc = L.lib().emgpu_debug_terminal_counters(ctx._h, out.ctypes.data_as(C.c_void_p), 24)
if rc != 1:
    sys.exit("not a -DEMGPU_TERM_COUNTERS build")
c = [int(x) for x in out]
it = c[0]
names = {1: "active lanes", 2: "lanes beginning a step", 7: "lanes re-drawing (att > 0)", 3: "lanes with an event", 4: "  heading event", 5: "  altitude event", 6: "  speed event",
         19: "lanes whose draw is rejected", 8: "lanes turning", 11: "lanes refilled", 12: "lanes on the pivot path"}
print("encounters %d, track rows %d, wave-iterations %d (%.1f per 64 track rows)" % (n, int(rows.sum()), it, it / (int(rows.sum()) / 64.0)))
for k in (1, 2, 7, 3, 4, 5, 6, 19, 8, 11, 12):
    print("%-32s %6.2f of 64 per wave-iteration" % (names[k], c[k] / it))
print("wave-iterations with: an event %.3f, a speed event %.3f, a turn %.3f, a rejected draw %.3f, a pivot-path lane %.3f" % (c[15] / it, c[18] / it, c[16] / it, c[17] / it, c[20] / it))
print("flush trips per wave-iteration %.2f; refills per wave-iteration %.3f; bearing walk trips down %.2f up %.2f per wave-iteration" % (c[9] / it, c[10] / it, c[13] / it, c[14] / it))
if c[22] or c[23]:
    print("event queue: parked lanes %.2f of 64 per wave-iteration; the event code ran in %.3f of the wave-iterations "
          "(the per-path lines above that are counted inside the event block see only the wave-iterations in which lane 0 is parked)" % (c[22] / it, c[23] / it))
