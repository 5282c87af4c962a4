"""tools/write_bw_probe.py -- what a plain streaming WRITE of the benchmark's trace size achieves on this box, next to the benchmark
kernel itself (same process, same buffers' size): torch fill_ (a float4 store per lane, nothing else), zero_ (hipMemset) and a
device-to-device copy, 20 timed repetitions each after 5 untimed ones, HIP events.  The HBM roofline of bench.py is priced against
8 TB/s; this says how much of that a kernel that does nothing but store reaches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

dev = torch.device("cuda", 0)
n_bytes = 10_000_000 * 3635                      # the benchmark's algorithmic bytes per launch
buf = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev)
src = torch.empty(n_bytes // 8, dtype=torch.float32, device=dev)


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[0]


out = {}
med, best = timed(lambda: buf.fill_(1.5))
out["fill_36GB"] = {"median_ms": med, "best_ms": best, "TB_s_median": n_bytes / med / 1e9}
med, best = timed(lambda: buf.zero_())
out["memset_36GB"] = {"median_ms": med, "best_ms": best, "TB_s_median": n_bytes / med / 1e9}
half = buf[: n_bytes // 8]
med, best = timed(lambda: half.copy_(src))
out["copy_18GB_read_18GB_write"] = {"median_ms": med, "best_ms": best, "TB_s_median_read_plus_write": n_bytes / med / 1e9}
med, best = timed(lambda: float(0) if buf.sum() is None else None, reps=10, warm=2)
out["sum_36GB_read"] = {"median_ms": med, "best_ms": best, "TB_s_median": n_bytes / med / 1e9}
del buf, src, half
torch.cuda.empty_cache()
line = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs --no-host-path"],
                      capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(line)
out["benchmark_kernel"] = {"kernel": d["config"]["kernel"], "avg_launch_ms": d["roofline"]["avg_launch_ms"], "TB_s_algorithmic": d["roofline"]["achieved"] / 1e3}
out["benchmark_vs_fill"] = out["benchmark_kernel"]["TB_s_algorithmic"] / out["fill_36GB"]["TB_s_median"]
print(json.dumps(out, indent=1))
