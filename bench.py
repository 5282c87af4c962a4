#!/usr/bin/env python3
"""bench.py -- trajectory samples/sec of the 240-s uncorrelated DBN sampler on N MI355X.

Workload (BASELINE.json configs[1]): uncor_1200code_v2p1 initial + transition DBN,
10 M trajectories x 240 s per GPU, REFERENCE_AUTO transition semantics, compact dense trace
output (5*n_i + 5*T*n_d = 3635 B / trajectory) resident in HBM.  One "step" = one pass of the hot
path over one batch of 10 M fresh trajectories (new global indices every step).  One process per
GPU; trajectories are independent, so ranks shard the global index range with no collective
(torch.distributed is used only for the barrier and the max-over-ranks clock).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- algorithmic bytes / average kernel launch duration (HIP events on the launch
                  stream) against the 8 TB/s HBM3E peak,
  cpu_baseline -- the CPU oracle (oracle/em_oracle.c, "port") timed on this host, rank 0, N=1 only.
MATLAB cannot be timed: it is not installed here or on the GPU box.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "uncor_1200code_v2p1"
N_PER_GPU = 10_000_000
DEFAULT_T = 240
SEED = 0x5EED0002
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first launches run while the GPU clocks still ramp up)")
    ap.add_argument("--n", type=int, default=N_PER_GPU, help="trajectories per GPU per step")
    ap.add_argument("--model", default=MODEL)
    ap.add_argument("--seconds", type=int, default=DEFAULT_T, help="trajectory length (the headline metric is quoted at 240)")
    ap.add_argument("--per-step", action="store_true", help="PER_STEP transition semantics instead of REFERENCE_AUTO")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="minimum trajectories timed on the CPU oracle (scaled up to ~15 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from em_model_manned_bayes_amd import em_io, native, sharding, _lib as L

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    tmp = tempfile.mkdtemp(prefix="emgpu_bench_")
    path = em_io.materialize_model(args.model, tmp)      # packed model -> reference-format .txt
    model = native.NativeModel.load_txt(path)            # the drop-in loader (em_read.m)
    labels = model.get_labels(L.F_LABELS_INITIAL)

    def lab(name):
        q = '"%s"' % name
        return labels.index(q) + 1 if q in labels else 0

    T = args.seconds
    n, ni, nd = args.n, model.n_initial, model.n_dyn
    G4 = (T + 3) // 4
    init_bin = torch.empty((ni, n), dtype=torch.uint8, device=dev)
    init_val = torch.empty((ni, n), dtype=torch.float32, device=dev)
    dyn_bin = torch.empty((G4, nd, n), dtype=torch.int32, device=dev)
    dyn_val = torch.empty((G4, nd, n, 4), dtype=torch.float32, device=dev)
    bytes_per_traj = 5 * ni + 5 * T * nd

    stream = torch.cuda.current_stream(dev)
    ctx = native.Context(local_rank, stream=stream.cuda_stream)
    mode = L.TRANSITION_PER_STEP if args.per_step else L.TRANSITION_REFERENCE_AUTO

    def step(k):
        first = sharding.step_first_index(k, rank, world, n)  # fresh global indices every step, disjoint across ranks
        p, _ = native.make_params(n, T, SEED, first_index=first, transition_mode=mode,
                                  idx_L=lab("L"), idx_v=lab("v"), idx_dh=lab("\\dot h"))
        native.sample_dbn_device(ctx, model, p, init_bin=init_bin.data_ptr(), init_val=init_val.data_ptr(),
                                 dyn_bin=dyn_bin.data_ptr(), dyn_val=dyn_val.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    ctx.sync()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step(args.warmup + k)
        ev[k][1].record(stream)
    barrier()
    t1 = time.perf_counter()
    ctx.sync()  # surfaces deferred rejection-cap errors
    elapsed = t1 - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_name = ctx.last_kernel()

    # size-independent sanity on the full-size output: every bin within 1..r, no NaN
    assert int(init_bin.min()) >= 1 and bool(torch.isfinite(dyn_val[:, :, : min(n, 100000)]).all())

    if rank == 0:
        total = n * world * args.steps
        value = total / elapsed
        avg_kernel_s = (sum(kern_ms) / len(kern_ms)) * 1e-3
        achieved = bytes_per_traj * n / avg_kernel_s / 1e9
        out = {
            "metric": "trajectory samples/sec (240 s uncor DBN)",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s initial+transition DBN, %d trajectories x %d s per GPU" % (args.model, n, T),
                       "transition_mode": "PER_STEP" if args.per_step else "REFERENCE_AUTO",
                       "output": "dense trace u8 bin + f32 value, %d B/trajectory" % bytes_per_traj,
                       "kernel": kernel_name, "sharding": "global sample index, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel_name, "avg_launch_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": bytes_per_traj * n},
        }
        out["roofline"].update(recorded_traffic(kernel_name, bytes_per_traj * n))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(path, args.cpu_sample, args.per_step, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def recorded_traffic(kernel_name, algorithmic_bytes):
    """roofline.traffic: HBM bytes per launch from the PMC passes (WRITE_SIZE + 2 x FETCH_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md) of the newest committed profile of the SAME kernel and
    launch size (profiles/*_summary.json, produced by tools/profile_bench.sh: counters need their
    own rocprofv3 passes and cannot be read from inside this process).  null if none matches."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            s = json.load(open(f))
        except Exception:
            continue
        line = s.get("bench_line", {})
        same = kernel_name.replace(" ", "") in s.get("kernel", "").replace(" ", "") and line.get("roofline", {}).get("algorithmic_bytes_per_launch") == algorithmic_bytes
        if same and "hbm_traffic_bytes_per_launch" in s:
            best = (s["hbm_traffic_bytes_per_launch"], os.path.basename(f))
    if best is None:
        return {"traffic": None}
    return {"traffic": best[0], "traffic_source": "profiles/" + best[1]}


def cpu_baseline(model_txt, n_cpu, per_step, T):
    """The CPU oracle (a faithful scalar port of the reference algorithm) on the same workload,
    bounded sample, one thread.  Reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    om = O.OracleModel(O.parse_model_txt(model_txt))
    t0 = time.perf_counter()
    O.uncor_sample(om, 2000, T, SEED, mode=O.RNG_PHILOX, want_events=False, want_dense=True)  # warm + calibrate
    rate = 2000 / (time.perf_counter() - t0)
    n_cpu = int(min(max(n_cpu, rate * 10.0), 2_000_000))  # about 10 s of CPU work per leg
    t0 = time.perf_counter()
    O.uncor_sample(om, n_cpu, T, SEED, mode=O.RNG_PHILOX, per_step=per_step, want_events=False, want_dense=True)
    dt1 = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_mt = int(min(n_cpu * cores, 1_000_000))  # dense f64 outputs: 6.5 GB of host memory at 1 M
    t0 = time.perf_counter()
    O.uncor_sample_mt(om, n_mt, T, SEED, cores, per_step=per_step)
    dtm = time.perf_counter() - t0
    return {"value": n_mt / dtm, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "single_thread_value": n_cpu / dt1,
            "sample": "oracle/em_oracle.c (scalar port of the reference algorithm), Philox mode, same workload: %d trajectories x %d s "
                      "on %d threads in %.1f s; 1 thread: %d trajectories in %.1f s; MATLAB itself is not installed and cannot be timed"
                      % (n_mt, T, cores, dtm, n_cpu, dt1)}


if __name__ == "__main__":
    main()
