#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native encounter sampler on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config uncor|cor|mixed|terminal]

--config uncor (default; BASELINE.json configs[1], the configuration the headline metric is quoted
on): uncor_1200code_v2p1 initial + transition DBN, 10 M trajectories x 240 s per GPU,
REFERENCE_AUTO transition semantics, compact dense trace (5*n_i + 5*T*n_d = 3635 B / trajectory)
resident in HBM.  cor / mixed / terminal are BASELINE.json configs[2..4] (see CONFIGS below).
One "step" = one pass of the hot path over one batch of fresh units (new global indices every step).

N > 1: one process per GPU.  Under a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) this process is one rank; WITHOUT one, `--gpus N` makes this process the launcher: it
starts N rank processes itself -- before anything here touches a GPU -- and relays rank 0's line.
It never runs fewer ranks than asked: a mismatch between --gpus and the ranks that ran is an error
(exit code 3), not a quietly smaller benchmark.  Units are independent, so ranks shard the global
index range with no collective (torch.distributed is only the barrier and the max-over-ranks clock).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- algorithmic bytes / average step duration on the launch stream (HIP events)
                  against the 8 TB/s HBM3E peak,
  cpu_baseline -- the CPU oracle (oracle/em_oracle.c, "port") timed on this host, rank 0, N=1 only,
  configs      -- BASELINE.json configs[2..4] + config 2 under PER_STEP, one short entry each,
  host_path    -- emgpu_sample_dbn_host end to end (PCIe-inclusive; never the headline).
The line holds NUMBERS ONLY (it must fit the few KB of tail its reader keeps): what every field means and how it is measured is
NOTES below = profiles/r06_bench_notes.json (`--write-notes`); the verbose record of a run (settle groups, telemetry, pre-warm) goes to
stderr as one `DETAIL {...}` line and to `--detail-out FILE`.
MATLAB cannot be timed: it is not installed here or on the GPU box.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_T = 240
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
V1P2 = ["uncor_1200exclude_fwme_v1p2", "uncor_1200exclude_fwse_v1p2", "uncor_1200exclude_rotorcraft_v1p2",
        "uncor_1200only_fwme_v1p2", "uncor_1200only_fwse_v1p2", "uncor_1200only_rotorcraft_v1p2"]

# BASELINE.json configs[1..4] as concrete synthetic inputs
CONFIGS = {
    "uncor": dict(models=["uncor_1200code_v2p1"], n=10_000_000, seed=0x5EED0002, unit="trajectories/s",
                  metric="trajectory samples/sec (240 s uncor DBN)",
                  workload="%(model)s initial+transition DBN, %(n)d trajectories x %(T)d s per GPU"),
    "uncor_per_step": dict(models=["uncor_1200code_v2p1"], n=10_000_000, seed=0x5EED0002, unit="trajectories/s", per_step=True,
                           metric="trajectory samples/sec (240 s uncor DBN, PER_STEP transition semantics)",
                           workload="%(model)s under EMGPU_TRANSITION_PER_STEP (the true per-timestep DBN of dbn_sample.m:65-93 instead of the frozen-parent "
                                    "branch the reference takes for this file), %(n)d trajectories x %(T)d s per GPU"),
    "cor": dict(models=["cor_v1"], n=10_000_000, seed=0x5EED0003, unit="encounters/s",
                metric="encounter samples/sec (240 s correlated two-aircraft DBN)",
                workload="%(model)s correlated two-aircraft joint network (stand-in for cor_v2p1, absent from the reference mount), "
                         "%(n)d encounters x %(T)d s per GPU"),
    "cor_v2p1_like": dict(models=["cor_v2p1_like"], n=10_000_000, seed=0x5EED0003, unit="encounters/s",
                          metric="encounter samples/sec (240 s correlated two-aircraft DBN)",
                          workload="%(model)s: generator-made correlated network with cor_v2p1's table sizes (4x the columns of cor_v1; "
                                   "em_model_manned_bayes_amd/synthetic.py, seed 0x5EED0003), %(n)d encounters x %(T)d s per GPU"),
    "mixed": dict(models=V1P2, n=6_250_000, seed=0x5EED0004, unit="trajectories/s",
                  metric="trajectory samples/sec (240 s uncor DBN, mixed batch over the six uncor_*_v1p2 files)",
                  workload="mixed batch over all uncor_*_v1p2 model files, model = contiguous block of the global index range, "
                           "%(n)d trajectories x %(T)d s per GPU (50 M on 8 GPUs)"),
    "terminal": dict(models=[], n=12_500_000, seed=0x5EED0005, unit="encounters/s",   # config 5's 100 M on 8 GPUs: one GPU's share (131 GB of tracks)
                     metric="terminal encounters/sec (geometry draw + forward/backward propagation of both aircraft)",
                     workload="CorTerminalModel: terminal_v3_radar geometry network + 10 synthetic trajectory models (the trained "
                              "files are absent from the reference mount), %(n)d encounters x 4 tracks x <=121 s per GPU (100 M on 8 GPUs)"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first launches run while the GPU clocks still ramp up)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="uncor")
    ap.add_argument("--n", type=int, default=0, help="units per GPU per step (default: the config's)")
    ap.add_argument("--model", default=None, help="uncor / cor: another packed model (or a reference-format .txt path)")
    ap.add_argument("--seconds", type=int, default=DEFAULT_T, help="trajectory length (the headline metric is quoted at 240)")
    ap.add_argument("--per-step", action="store_true", help="PER_STEP transition semantics instead of REFERENCE_AUTO")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ld-pad", type=int, default=1024, help="pad the trace's leading dimension to a multiple of this many columns")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `configs` object (BASELINE.json configs[2..4] measured in the same process after the headline)")
    ap.add_argument("--step-gap-ms", type=float, default=0.0,
                    help="DIAGNOSTIC (tools/power_probe.sh): idle this long after every launch; the line then says so and is no benchmark line")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each entry of `configs` (after --other-warmup untimed steps and the settle phase)")
    ap.add_argument("--other-warmup", type=int, default=5, help="untimed steps ahead of each entry of `configs`")
    ap.add_argument("--settle-max", type=int, default=48,
                    help="entries of `configs`: after the warm-up steps, untimed steps are launched back to back in groups of four until the last two of a "
                         "group agree within 1 %% and two consecutive groups' means within 0.5 %% (the GPU's clocks ramp after the idle seconds of a CPU "
                         "leg), at most this many")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="minimum units timed on the CPU oracle (scaled up to ~10 s)")
    ap.add_argument("--local-smooth", action="store_true", help="terminal: add the smoothing pass of createEncounter.m:88-89 (k_terminal_smooth: the flagged stand-in for em-core's local_smooth; a second pass over the tracks)")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="seconds of the same step launched back to back BEFORE the warm-up steps, untimed and reported in the line (`roofline.prewarm`): "
                         "from idle the GPU's clocks take 100-200 ms of load to settle (the first timed steps of a cold process read 5-10 %% long: "
                         "round 5, profiles/r05_bench_lines.jsonl), which W = 5 short warm-up steps do not cover.  0: the W warm-up steps only. "
                         "(Not what separates a box's fast and slow states: tools/thermal_probe.sh, HISTORY.md section 7.)")
    ap.add_argument("--placement-candidates", type=int, default=0,
                    help="uncor / cor / mixed: the `candidates` argument of emgpu_trace_alloc, the library's trace allocator (include/emgpu.h): it "
                         "allocates that many candidate traces, times the step on each BEFORE anything else here runs and keeps the fastest -- where a "
                         "trace lies in memory decides how fast it is written (profiles/r05_placement_probe.txt).  0 (default): the library's own "
                         "policy (3, more while the two fastest disagree by over 1 %%); 1: the first allocation as it comes.  The line reports the "
                         "candidates' times (roofline.placement) and the first one's as roofline.first_allocation_ms")
    ap.add_argument("--no-host-path", action="store_true", help="skip the `host_path` object (emgpu_sample_dbn_host end to end at 1 M trajectories)")
    ap.add_argument("--host-n", type=int, default=1_000_000, help="trajectories of the `host_path` measurements")
    ap.add_argument("--detail-out", default=None, help="write the verbose record of the run (what the line leaves out: settle groups, pre-warm, telemetry, samples) to this file")
    ap.add_argument("--verbose-line", action="store_true",
                    help="print the verbose roofline / config / cpu_baseline objects in the line itself (the round-5 format: the probes under tools/ that read "
                         "roofline.gpu_telemetry, roofline.avg_step_ms ... pass this); the default line holds numbers only")
    ap.add_argument("--write-notes", default=None, help="write NOTES (what every field of the line means and how it is measured) to this file and exit")
    ap.add_argument("--telemetry-s", type=float, default=2.5,
                    help="seconds of untimed back-to-back steps AFTER the timed region during which the shader clock and socket power are read (0: skip); "
                         "nothing is sampled inside the timed region")
    ap.add_argument("--ranges-out", default=None,
                    help="TEST HOOK: a directory into which every rank writes rank<r>.json = the global index range (and model blocks) of every step it "
                         "launched + a digest of its last step's output (tests check the ranks' ranges are disjoint and the digests are what one "
                         "process computes for those ranges)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: allow more ranks than GPUs (ranks share devices, gloo barrier); the line says so")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher's environment
# ------------------------------------------------------------------------------------------------
def visible_gpu_count():
    """GPUs this process tree may use, counted WITHOUT loading a GPU runtime into THIS process: a child process asks the library
    (emgpu_device_count: what the ranks themselves will see -- the runtime applies HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES and only
    counts devices whose files the container may open) and exits.  Only when that child cannot run (no library yet) the KFD topology in
    sysfs is counted instead (a node with simd_count > 0 is a GPU; sysfs ignores cgroup / device-file restrictions, so this can
    over-count inside a container that was granted some of the host's GPUs), narrowed by the *_VISIBLE_DEVICES lists."""
    import glob
    import re
    code = ("import ctypes,sys; sys.path.insert(0, %r); from em_model_manned_bayes_amd import _lib as L; c = ctypes.c_int32(0); "
            "L.lib().emgpu_device_count(ctypes.byref(c)); print('emgpu_device_count', c.value)" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300).stdout.decode()
        m = re.search(r"emgpu_device_count (\d+)", out)
        if m:
            return int(m.group(1))
    except Exception:
        pass
    total = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            m = re.search(r"^simd_count\s+(\d+)", open(f).read(), re.M)
        except OSError:
            continue
        total += 1 if (m and int(m.group(1)) > 0) else 0
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            total = min(total, len([x for x in v.split(",") if x.strip() != ""]))
    return total


def launch_ranks(args, argv):
    """Start args.gpus rank processes of this script and relay rank 0's JSON line.  Nothing in this
    process touches a GPU or imports torch: the devices are counted by a child process that asks the library (visible_gpu_count;
    the KFD topology in sysfs only when that child cannot run)."""
    have = visible_gpu_count()
    if have < args.gpus and not args.oversubscribe:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, have))
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting at a barrier: watch them all, stop the rest (exact PIDs)
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    out0 = procs[0].communicate()[0].decode()   # one short line: the pipe cannot fill up
    rcs = [p.wait() for p in procs]
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return max(abs(r) for r in rcs) or 1
    if line is None or json.loads(line).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs but the line says %s\n" % (args.gpus, line))
        return 3
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
# plumbing: device memory, streams, events and the barrier (PyTorch-ROCm; not the product)
# ------------------------------------------------------------------------------------------------
class TorchRocm:
    def __init__(self, rank, local_rank, world, oversubscribe=False):
        import torch
        import torch.distributed as dist
        from em_model_manned_bayes_amd import native
        self.torch, self.dist, self.native = torch, dist, native
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        have = torch.cuda.device_count()
        self.shared = world > have
        if self.shared and not oversubscribe:
            raise SystemExit("bench.py: %d ranks but %d GPU(s)" % (world, have))
        self.rank, self.world = rank, world
        self.dev = torch.device("cuda", local_rank % have)
        torch.cuda.set_device(self.dev)
        if world > 1:
            if self.shared:   # RCCL refuses two ranks on one device: the test-only path uses gloo for the barrier
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.dev)
        self.stream = torch.cuda.current_stream(self.dev)

    def context(self):
        return self.native.Context(self.dev.index, stream=self.stream.cuda_stream)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=getattr(self.torch, dtype), device=self.dev)

    def from_numpy(self, a):
        return self.torch.from_numpy(a).to(self.dev)

    def wrap(self, ptr, shape, dtype):
        """A tensor over device memory somebody else owns (a trace of the library's pool): __cuda_array_interface__, no copy."""
        ts = {"uint8": "|u1", "float32": "<f4", "int32": "<i4", "float64": "<f8"}[dtype]

        class _Mem:
            __cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": ts, "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(_Mem(), device=self.dev)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def record(self, ev):
        ev.record(self.stream)

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.shared else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def release(self):
        """Give the memory of dropped workloads back before the next one allocates."""
        import gc
        gc.collect()
        self.torch.cuda.empty_cache()

    def finish(self):
        if self.world > 1:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def _materialize(name, tmp):
    from em_model_manned_bayes_amd import em_io, synthetic
    if name == "cor_v2p1_like":   # generator-made stand-in for the absent cor_v2p1.txt (seed 0x5EED0003)
        return synthetic.write_correlated_v2p1_like(tmp)
    return name if os.path.isfile(name) else em_io.materialize_model(name, tmp)


def _label_indices(model):
    from em_model_manned_bayes_amd import _lib as L
    labels = model.get_labels(L.F_LABELS_INITIAL)

    def lab(name):
        q = '"%s"' % name
        return labels.index(q) + 1 if q in labels else 0
    return dict(idx_L=lab("L"), idx_v=lab("v"), idx_dh=lab("\\dot h"))


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (the GPU
    boxes show 256 hardware threads behind a 16-CPU quota: 256 busy threads there are throttled to 16 cores' worth of time)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "%d hardware threads in the affinity mask" % aff
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < aff:
                return q, note + ", cgroup cpu.max = %s/%s -> %d cores' worth of CPU time" % (quota, period, q)
    except (OSError, ValueError):
        pass
    return aff, note


class DbnWorkload:
    """uncor / cor / mixed: emgpu_sample_dbn_device (one model) or emgpu_sample_dbn_blocks_device (several
    models filling one shared trace), dense output."""

    def __init__(self, args, cfg, pl, rank, world):
        from em_model_manned_bayes_amd import _lib as L
        native = pl.native
        self.pl, self.native, self.rank, self.world, self.cfg = pl, native, rank, world, cfg
        self.tmp = tempfile.mkdtemp(prefix="emgpu_bench_")
        names = [args.model] if args.model else cfg["models"]
        self.paths = [_materialize(nm, self.tmp) for nm in names]
        self.names = [os.path.splitext(os.path.basename(nm))[0] for nm in names]
        self.models = [native.NativeModel.load_txt(p) for p in self.paths]   # the drop-in loader (em_read.m)
        self.idx = [_label_indices(m) for m in self.models]
        self.T, self.n, self.seed = args.seconds, args.n or cfg["n"], cfg["seed"]
        m0 = self.models[0]
        self.ni, self.nd = m0.n_initial, m0.n_dyn
        G4 = (self.T + 3) // 4
        # the trace's leading dimension (emgpu_sample_out.ld) is padded to a multiple of 1024 columns: every row of every array then
        # starts on a 1 KiB boundary and no wave store straddles a 128-byte line (6.25 M columns unpadded cost 18 %)
        self.ld = ld = -(-self.n // args.ld_pad) * args.ld_pad
        shapes = (((self.ni, ld), "uint8"), ((self.ni, ld), "float32"), ((G4, self.nd, ld), "int32"), ((G4, self.nd, ld, 4), "float32"))
        self.bytes_per_unit = 5 * self.ni + 5 * self.T * self.nd
        self.mode = L.TRANSITION_PER_STEP if args.per_step else L.TRANSITION_REFERENCE_AUTO
        self.per_step = args.per_step
        self.ctx = pl.context()
        self.launches_per_step = 1
        self.kernels = []
        self.ranges = []       # (step, first global index, n[, model blocks]) of every step launched (--ranges-out)
        self.trace = None
        if hasattr(pl, "wrap") and hasattr(native, "Trace"):
            # THE TRACE IS THE LIBRARY'S: emgpu_trace_alloc allocates the candidates, times this workload's own launch on each and keeps the
            # fastest -- what any consumer of the C ABI gets by calling it instead of hipMalloc (a mixed batch is placed with its first model's
            # launch over the whole trace: same kernel family, same store pattern)
            p, _keep = native.make_params(self.n, self.T, self.seed, first_index=0, transition_mode=self.mode, **self.idx[0])
            self.trace = native.Trace(self.ctx, self.models[0], p, want=L.TRACE_INIT | L.TRACE_DENSE, candidates=int(getattr(args, "placement_candidates", 0)))
            assert self.trace.ld == ld, (self.trace.ld, ld)
            self._ptrs = {k: v for k, v in self.trace.ptrs().items() if k in ("init_bin", "init_val", "dyn_bin", "dyn_val", "ld")}
            self.init_bin, self.init_val, self.dyn_bin, self.dyn_val = (pl.wrap(self._ptrs[k], shape, dt) for k, (shape, dt) in
                                                                        zip(("init_bin", "init_val", "dyn_bin", "dyn_val"), shapes))
            self.placement = dict(self.trace.report, bytes=self.trace.bytes)
        else:   # (tests/mp_plumbing.py: host memory behind the same interface)
            self.init_bin, self.init_val, self.dyn_bin, self.dyn_val = (pl.empty(shape, dt) for shape, dt in shapes)
            self._ptrs = dict(init_bin=self.init_bin.data_ptr(), init_val=self.init_val.data_ptr(), dyn_bin=self.dyn_bin.data_ptr(),
                              dyn_val=self.dyn_val.data_ptr(), ld=self.ld)

    def close(self):
        """Give the trace back (the ctx's pool goes with the ctx) before the next workload allocates."""
        self.init_bin = self.init_val = self.dyn_bin = self.dyn_val = None
        if self.trace is not None:
            self.trace.free()
            self.trace = None
        if hasattr(self.ctx, "trim"):
            self.ctx.trim()

    def ptrs(self):
        return self._ptrs

    def step(self, k):
        from em_model_manned_bayes_amd import sharding
        native, n = self.native, self.n
        # fresh global indices every step, disjoint across ranks: step k covers [k*world*n, (k+1)*world*n)
        first = sharding.step_first_index(k, self.rank, self.world, n)
        if len(self.models) == 1:
            p, _ = native.make_params(n, self.T, self.seed, first_index=first, transition_mode=self.mode, **self.idx[0])
            native.sample_dbn_device(self.ctx, self.models[0], p, **self.ptrs())
            self.ranges.append({"step": k, "first": first, "n": n})
            return
        # mixed batch: model m owns the m-th contiguous block of the step's global range; this rank's shard of that
        # range intersects one or two blocks -> one launch each into the rank's ONE trace
        total = n * self.world
        base = k * total
        blocks = [(m, base + f, c) for (m, f, c) in native.mixed_blocks(total, len(self.models), first - base, first - base + n)]
        p, _ = native.make_params(n, self.T, self.seed, first_index=first, transition_mode=self.mode, **self.idx[0])
        native.sample_dbn_blocks_device(self.ctx, self.models, p, blocks, **self.ptrs())
        self.ranges.append({"step": k, "first": first, "n": n, "blocks": [[int(a), int(b), int(c)] for a, b, c in blocks]})
        self.blocks_per_step = max(getattr(self, "blocks_per_step", 0), len(blocks))
        self.launches_per_step = max(self.launches_per_step, self.ctx.last_launches())   # models that share a kernel instance share ONE launch

    def sync(self):
        self.ctx.sync()  # surfaces deferred rejection-cap errors

    def digest(self):
        """Sums over the last step's output (int64, exact): what a test compares with one process's result for the same global range."""
        n = self.n
        return {"init_bin": int(self.init_bin[:, :n].long().sum()), "dyn_bin": int(self.dyn_bin[:, :, :n].long().sum()),
                "init_val_bits": int(self.init_val[:, :n].contiguous().view(self.pl.torch.int32).long().sum())}

    def kernel_name(self):
        return self.ctx.last_kernel()

    def check(self):
        # size-independent sanity on the full-size output: every bin within 1..r, no NaN
        t = self.pl.torch if hasattr(self.pl, "torch") else None
        if t is not None:
            assert int(self.init_bin[:, : self.n].min()) >= 1 and bool(t.isfinite(self.dyn_val[:, :, : min(self.n, 100000)]).all())
            assert int((self.dyn_bin[0, :, : self.n] & 0xFF).min()) >= 1

    def streaming_write(self, reps=20):
        """What a kernel that does nothing but store reaches on THIS box: torch zero_() (its vectorised fill kernel) over the trace buffers the step just filled,
        median of `reps` after 3 untimed ones, HIP events.  The roofline's `peak` stays the guide's 8 TB/s; this is the ceiling a write
        stream meets in practice (tools/ubench/store_ubench.hip: a store-only twin of the sampler's own store pattern reaches the same)."""
        t = getattr(self.pl, "torch", None)
        if t is None:
            return None
        bufs = [self.dyn_bin, self.dyn_val]
        nbytes = sum(b.numel() * b.element_size() for b in bufs)
        ms = []
        for r in range(reps + 3):
            a, b = self.pl.event(), self.pl.event()
            self.pl.record(a)
            for x in bufs:
                x.zero_()
            self.pl.record(b)
            t.cuda.synchronize()
            if r >= 3:
                ms.append(self.pl.elapsed_ms(a, b))
        ms.sort()
        med = ms[len(ms) // 2]
        return {"GB/s": nbytes / med / 1e6, "bytes": nbytes, "median_ms": med,
                "how": "torch zero_() (a plain fill kernel) over the step's own dyn_bin + dyn_val buffers on this box, median of %d" % reps}

    def config(self):
        return {"workload": self.cfg["workload"] % dict(model=self.names[0], n=self.n, T=self.T),
                "transition_mode": "PER_STEP" if self.per_step else "REFERENCE_AUTO",
                "output": "dense trace of the DYNAMIC variables only (u8 bin + f32 value per variable-second) + initial "
                          "state (u8 + f32 per variable): %d B/unit; re-draws of static variables appear only in the "
                          "event-list output" % self.bytes_per_unit,
                "values": "f32 at the boundary (f64 arithmetic inside, rounded on store); uniforms are 32-bit",
                "trace_ld": self.ld, "models": self.names, "launches_per_step": self.launches_per_step, "model_blocks_per_step": getattr(self, "blocks_per_step", 1),
                "sharding": "global sample index, no collective"}

    def cpu_baseline(self, n_cpu, seconds=10.0):
        """The CPU oracle (a faithful scalar port of the reference algorithm) on the same workload,
        bounded sample (about `seconds` of work per leg), one thread and all threads.  Reported baseline, not the target."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        om = O.OracleModel(O.parse_model_txt(self.paths[0]))
        T, seed = self.T, self.seed
        t0 = time.perf_counter()
        O.uncor_sample(om, 2000, T, seed, mode=O.RNG_PHILOX, want_events=False, want_dense=True)  # warm + calibrate
        rate = 2000 / (time.perf_counter() - t0)
        n_cpu = int(min(max(n_cpu if seconds >= 10.0 else 1000, rate * seconds), 2_000_000))  # about `seconds` of CPU work per leg
        t0 = time.perf_counter()
        O.uncor_sample(om, n_cpu, T, seed, mode=O.RNG_PHILOX, per_step=self.per_step, want_events=False, want_dense=True)
        dt1 = time.perf_counter() - t0
        cores, cores_note = usable_cores()
        # all threads: thread-private output blocks (oracle/em_oracle.c em_uncor_sample_throughput_mt), about 10 s of work
        n_cal = max(cores * 512, 4096)   # calibrate first: the sample is sized from the rate the threads really reach
        t0 = time.perf_counter()
        O.uncor_sample_throughput_mt(om, n_cal, T, seed, cores, per_step=self.per_step)
        n_mt = int(min(max(n_cal, n_cal / (time.perf_counter() - t0) * seconds), 20_000_000))
        t0 = time.perf_counter()
        O.uncor_sample_throughput_mt(om, n_mt, T, seed, cores, per_step=self.per_step)
        dtm = time.perf_counter() - t0
        return {"value": n_mt / dtm, "unit": self.cfg["unit"], "cores": cores, "kind": "port",
                "single_thread_value": n_cpu / dt1, "thread_scaling": (n_mt / dtm) / (n_cpu / dt1), "cores_note": cores_note,
                "sample_short": "%s: %d units x %d s on %d threads in %.1f s; 1 thread: %d units in %.1f s" % (self.names[0], n_mt, T, cores, dtm, n_cpu, dt1),
                "sample": "oracle/em_oracle.c (scalar port of the reference algorithm), Philox mode, model %s: %d units x %d s "
                          "on %d threads in %.1f s (thread-private dense outputs); 1 thread: %d units in %.1f s; MATLAB itself is not "
                          "installed and cannot be timed" % (self.names[0], n_mt, T, cores, dtm, n_cpu, dt1)}


class TerminalWorkload:
    """configs[4]: per step, fresh geometry draws (k_bn: the 15-variable network + box / speed rejection, @CorTerminalModel/sample.m:29-77),
    the inputs of createEncounter (k_terminal_geo) and PropagateTrajectory of both aircraft in both directions (k_terminal_propagate) on
    synthetic trajectory tables -- emgpu_sample_terminal_device, three launches, everything device-resident."""

    def __init__(self, args, cfg, pl, rank, world):
        import numpy as np
        import em_model_manned_bayes_amd as E
        from em_model_manned_bayes_amd import synthetic, _lib as L
        self.np, self.L = np, L
        self.pl, self.native, self.rank, self.world, self.cfg = pl, pl.native, rank, world, cfg
        self.n, self.seed = args.n or cfg["n"], cfg["seed"]
        self.dir = synthetic.write_terminal_directory(tempfile.mkdtemp(prefix="emgpu_bench_term_"))
        self.ctx = pl.context()
        self.t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=self.dir)
        self.cap = 123
        n, ni = self.n, self.t.native.n_initial
        self.ni = ni
        self.geom_val = pl.empty((ni, n), "float32")
        self.geo = pl.empty((n, 12), "float64")
        self.mof = pl.empty((4 * n,), "int32")
        tshape = (2 * n, 2 * self.native.terminal_t0_row(self.cap), 5)
        self._traj_addr = None
        if hasattr(pl, "wrap") and hasattr(self.ctx, "device_alloc"):   # 131 GB of joined tracks: from the library's allocator (emgpu_device_alloc), like the traces
            self._traj_addr = self.ctx.device_alloc(4 * tshape[0] * tshape[1] * tshape[2])
            self.traj = pl.wrap(self._traj_addr, tshape, "float32")
        else:
            self.traj = pl.empty(tshape, "float32")
        self.rows = pl.empty((4 * n,), "int32")
        self.att = pl.empty((n,), "int32")
        self.bs = None if np.all(np.isinf(self.t.bounds_sample)) else self.t.bounds_sample
        self.local_smooth = bool(getattr(args, "local_smooth", False))
        # geometry 5 B x 15 variables + 5 B (u8 bin + f32 value) x 3 dynamic variables per track-second (<= 4 x 122 of them)
        self.bytes_bound = 75 + 4 * 122 * 15
        self.bytes_data_dependent = True
        self.bytes_per_unit = self.bytes_bound   # replaced in check() by 75 + 15 B x the track-seconds the run really produced
        self.bytes_stored_per_unit = None        # 75 + 20 B x the rows the run really wrote (what the kernel stores: five f32 per row)
        self.launches_per_step = 1
        self.ranges = []

    def close(self):
        self.traj = None
        if self._traj_addr:
            self.ctx.device_free(self._traj_addr)
            self._traj_addr = None

    def digest(self):
        t = self.pl.torch
        return {"rows": int(self.rows.long().sum()), "model_of": int(self.mof.long().sum()), "geom_val_bits": int(self.geom_val.view(t.int32).long().sum()),
                "attempts": int(self.att.long().sum())}

    def step(self, k):
        from em_model_manned_bayes_amd import sharding
        first = sharding.step_first_index(k, self.rank, self.world, self.n)
        self.ranges.append({"step": k, "first": first, "n": self.n})
        p, self._keep = self.native.terminal_sample_params(self.t.native, self.n, self.seed, self.t._dyn_rows(), first_index=first, tmax_s=120.0,
                                                           cap=self.cap, bounds_sample=self.bs, local_smooth=self.local_smooth)
        self.native.sample_terminal_device(self.ctx, self.t.native, [x.native for x in self.t._traj], p, self.geom_val.data_ptr(), self.geo.data_ptr(),
                                           self.mof.data_ptr(), self.traj.data_ptr(), self.rows.data_ptr(), attempts=self.att.data_ptr())

    def sync(self):
        self.ctx.sync()

    def kernel_name(self):
        return self.ctx.last_kernel().split(" + ")[-1]   # the dominant kernel of the step's launches

    def check(self):
        # rows < 0: a track whose inner re-draw loop hit max_resample (the reference would spin on it, createEncounter.m:218-262)
        self.failed = int((self.rows < 0).sum())
        self.track_seconds = float(self.rows.clamp(min=0).sum().item()) / self.n
        # `frac` follows SURVEY.md 8(d): 15 B per track-second (a u8 bin + f32 value for each of heading, altitude, speed) -- the unit rounds
        # 1-3 used.  What the kernel actually STORES is the reference's output, createEncounter.m:162-167: x y z heading speed as five f32 per
        # row of the joined track (the t = 0 row of an aircraft once: rows written = track-seconds - 2 per encounter) = 20 B per row: reported
        # beside it as `output_bytes_per_unit` / `frac_of_output_bytes` (round 4 had silently made that the numerator).
        self.bytes_per_unit = 75.0 + 15.0 * self.track_seconds
        self.bytes_stored_per_unit = 75.0 + 20.0 * (self.track_seconds - 2.0)
        self.geom_attempts = float(self.att.float().mean().item())
        assert int(self.rows.max()) <= self.cap and int(self.rows.max()) >= 2 and self.failed <= 0.01 * 4 * self.n, (int(self.rows.min()), self.failed)

    def config(self):
        return {"workload": self.cfg["workload"] % dict(n=self.n),
                "output": "geometry sample f32 [15][n] + joined tracks f32 [2n][%d][5] (x y z heading speed per track-second; t_s = row number) + rows; "
                          "algorithmic bytes (SURVEY.md 8d) = 75 + 15 B x track-seconds = %.0f B/encounter as measured (bound: %d); bytes the kernel stores = "
                          "75 + 20 B x rows written = %.0f B/encounter"
                          % (2 * self.native.terminal_t0_row(self.cap), self.bytes_per_unit, self.bytes_bound, self.bytes_stored_per_unit or 0.0),
                "launches_per_step": 4 if self.local_smooth else 3, "kernels": self.ctx.last_kernel(),
                "timed_region": "k_bn (fresh geometry draw with rejection) + k_terminal_geo + k_terminal_propagate%s, every step"
                                % (" + k_terminal_smooth (createEncounter.m:88-89 through the documented stand-in for em-core's local_smooth)" if self.local_smooth else ""),
                "sharding": "global encounter index, no collective",
                "track_seconds_per_encounter": getattr(self, "track_seconds", None),
                "geometry_attempts_per_encounter": getattr(self, "geom_attempts", None),
                "tracks_over_the_redraw_cap": getattr(self, "failed", None)}

    def cpu_baseline(self, n_cpu, seconds=10.0):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import glob
        import oracle as O
        np = self.np
        files = [glob.glob(os.path.join(self.dir, "*_" + s + ".txt"))[0] for s in
                 ["ownship_landing_model", "ownship_landing_model_reverse", "ownship_takeoff_model", "ownship_takeoff_model_reverse",
                  "intruder_landing_model", "intruder_landing_model_reverse", "intruder_takeoff_model", "intruder_landing_model_reverse",
                  "intruder_transit_model", "intruder_landing_model_reverse"]]
        oms = []
        for f in files:
            pp = O.parse_model_txt(f)
            oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
        geo_all, mo_all = self.geo.cpu().numpy(), self.mof.cpu().numpy()
        n_have = geo_all.shape[0]
        dl = self.t._dyn_rows()

        def sample(m):          # the first m encounters of the step the GPU just ran (repeated when the GPU batch is smaller than the CPU sample)
            reps = -(-m // n_have)
            return (np.tile(geo_all, (reps, 1))[:m], np.tile(mo_all.reshape(-1, 4), (reps, 1))[:m].reshape(-1))
        # one thread: calibrate on 300 encounters, then about `seconds` of work
        g, mo = sample(300)
        t0 = time.perf_counter()
        O.propagate(oms, mo, g, self.seed, dl)
        n1 = int(min(max(300, 300 / (time.perf_counter() - t0) * seconds), 200_000))
        g, mo = sample(n1)
        t0 = time.perf_counter()
        O.propagate(oms, mo, g, self.seed, dl)
        dt1 = time.perf_counter() - t0
        # all allowed cores (the cgroup quota, like the DBN configs): thread-private track buffers, about `seconds` of work
        cores, cores_note = usable_cores()
        n_cal = max(cores * 128, 1024)
        g, mo = sample(n_cal)
        t0 = time.perf_counter()
        O.propagate_throughput_mt(oms, mo, g, self.seed, dl, cores)
        n_mt = int(min(max(n_cal, n_cal / (time.perf_counter() - t0) * seconds), 2_000_000))
        g, mo = sample(n_mt)
        t0 = time.perf_counter()
        O.propagate_throughput_mt(oms, mo, g, self.seed, dl, cores)
        dtm = time.perf_counter() - t0
        return {"value": n_mt / dtm, "unit": self.cfg["unit"], "cores": cores, "kind": "port",
                "single_thread_value": n1 / dt1, "thread_scaling": (n_mt / dtm) / (n1 / dt1), "cores_note": cores_note,
                "sample_short": "propagation of %d encounters on %d threads in %.1f s; 1 thread: %d in %.1f s" % (n_mt, cores, dtm, n1, dt1),
                "sample": "oracle/em_oracle.c em_propagate_trajectory (scalar port of createEncounter.m:93-329), Philox mode: propagation of %d encounters "
                          "on %d threads in %.1f s (thread-private track buffers); 1 thread: %d encounters in %.1f s (the geometry draw, <1 %% of the "
                          "work, is not in these figures); MATLAB itself is not installed and cannot be timed" % (n_mt, cores, dtm, n1, dt1)}


def make_workload(args, pl, rank, world):
    shift_mb = int(os.environ.get("EMGPU_BENCH_SHIFT_MB", "0") or 0)   # diagnostic: move the step's buffers to other addresses
    if shift_mb > 0 and hasattr(pl, "torch"):
        pl._shift = pl.empty((shift_mb, 1 << 20), "uint8")
        pl._shift.zero_()
    cfg = CONFIGS[args.config]
    if cfg.get("per_step"):   # (--config uncor_per_step == --per-step on config 2)
        args.per_step = True
    return (TerminalWorkload if args.config == "terminal" else DbnWorkload)(args, cfg, pl, rank, world)


class GpuTelemetry:
    """Shader clock and socket power of this rank's GPU under the benchmark's own back-to-back launches (an untimed run AFTER the timed region), read from amdgpu's hwmon files in sysfs (freq1_input =
    sclk in Hz, power1_input = package power in microwatts: 0.03 ms per read, no profiler, no privileges) by a thread that samples every
    2 ms.  The benchmark kernel runs at the board's power limit; how far a box lets the clock drop there differs from box to box by up to
    20 % (HISTORY.md section 7): with the clock in the line a reader can tell a slow box from a regression."""

    def __init__(self, local_index=0, pci=None, sysfs_root="/sys"):
        """`pci` = "dddd:bb:dd.f" of the device this rank computes on: a box may list more cards in sysfs than the process can see
        (a 1-GPU slice of an 8-GPU node), so the visible device's index says nothing about the card number; without it the
        `local_index`-th card with an sclk sensor is read."""
        import glob
        cards = []
        for f in glob.glob(os.path.join(sysfs_root, "class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            try:
                if open(os.path.join(os.path.dirname(f), "freq1_label")).read().strip() == "sclk":
                    dev = os.path.realpath(f.split("/hwmon/")[0])
                    cards.append((int(f.split("/card")[1].split("/")[0]), os.path.dirname(f), os.path.basename(dev).lower()))
            except (OSError, ValueError):
                pass
        cards.sort()
        self.dir, self.card = None, None
        if pci:
            for n, d, addr in cards:
                if addr == pci.lower():
                    self.dir, self.card = d, "card%d %s" % (n, addr)
        if self.dir is None and pci is None and local_index < len(cards):
            self.dir, self.card = cards[local_index][1], "card%d %s (by index)" % (cards[local_index][0], cards[local_index][2])
        self.samples, self._stop, self._th = [], None, None

    def _read(self):
        try:
            mhz = int(open(os.path.join(self.dir, "freq1_input")).read()) / 1e6
        except (OSError, ValueError):
            return None
        try:
            watts = int(open(os.path.join(self.dir, "power1_input")).read()) / 1e6
        except (OSError, ValueError):
            watts = None
        return mhz, watts

    def start(self):
        if not self.dir:
            return
        import threading
        self.samples, self._stop = [], threading.Event()

        def run():
            while not self._stop.is_set():
                v = self._read()
                if v:
                    self.samples.append(v)
                self._stop.wait(0.002)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        if self._th:
            self._stop.set()
            self._th.join()
            self._th = None

    def summary(self, tail=1.0):
        """Statistics of the last `tail` fraction of the samples."""
        if not self.samples:
            return None

        def stat(v):
            v = sorted(v)
            return {"median": v[len(v) // 2], "min": v[0], "max": v[-1]}
        keep = self.samples[int(len(self.samples) * (1.0 - tail)):]
        out = {"sclk_mhz": stat([a for a, _ in keep]), "samples": len(keep)}
        pw = [b for _, b in keep if b is not None]
        if pw:
            out["socket_power_w"] = stat(pw)
        try:
            out["power_cap_w"] = int(open(os.path.join(self.dir, "power1_cap")).read()) / 1e6
        except (OSError, ValueError):
            pass
        # the board's temperature sensors at the end of the probe (junction / memory: a hot HBM stack refreshes more often), and the memory clock
        import glob
        temps = {}
        for f in glob.glob(os.path.join(self.dir, "temp*_input")):
            try:
                lab = open(f.replace("_input", "_label")).read().strip()
                temps[lab] = int(open(f).read()) / 1000.0
            except (OSError, ValueError):
                pass
        if temps:
            out["temperature_c"] = temps
        for f in glob.glob(os.path.join(self.dir, "freq*_label")):
            try:
                if open(f).read().strip() == "mclk":
                    out["mclk_mhz"] = int(open(f.replace("_label", "_input")).read()) / 1e6
            except (OSError, ValueError):
                pass
        return out


# ------------------------------------------------------------------------------------------------
def measure(w, pl, args, warmup, steps, settle_max=0):
    """warmup untimed steps, then `steps` timed ones bracketed by barrier + device synchronisation on both sides; the launch
    durations come from HIP events on the stream the kernels are launched on.  Returns (elapsed_s max over ranks, [ms per step]).
    settle_max > 0 (the entries of `configs`, which start after seconds of GPU idleness): between the warm-up and the timed region, untimed
    groups of four back-to-back steps until the last two of a group agree within 1 % (w.settle says how many it took)."""
    # one untimed barrier + max-reduce first: a process group's first collectives set its communicator up (RCCL: lazily, hundreds of ms
    # with 8 ranks) -- that must never land between t0 and t1 of a 0.1 s timed region
    pl.barrier()
    pl.max_over_ranks(0.0)
    # (events and the telemetry reader are made BEFORE the pre-warm and the warm-up: milliseconds of host work between the warm-up and the timed region are
    # milliseconds of GPU idleness, and the governor drops the clock within 3 ms: the first timed steps then read 5-10 % long)
    ev = [(pl.event(), pl.event()) for _ in range(steps)]
    tel = None
    if hasattr(pl, "torch") and getattr(args, "telemetry_s", 0.0) > 0.0:
        idx = getattr(getattr(pl, "dev", None), "index", 0) or 0
        try:
            pr = pl.torch.cuda.get_device_properties(idx)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except (AttributeError, RuntimeError):
            pci = None
        tel = GpuTelemetry(idx, pci)   # (started after the timed region: a 2 ms sampling thread would share the interpreter with the launch loop)
    prewarm_s = float(getattr(args, "prewarm_s", 0.0) or 0.0)
    if prewarm_s > 0.0 and hasattr(pl, "torch"):   # the device's sustained state first (untimed; the line says so)
        t_end, k, first = time.perf_counter() + prewarm_s, 0, []
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            for _ in range(8):
                w.step(1_000_000 + k)   # (global indices far from the timed steps')
                k += 1
            w.sync()
            if len(first) < 4:
                first.append(round((time.perf_counter() - t0) * 1e3 / 8, 3))
        w.prewarm = {"seconds": prewarm_s, "steps": k, "ms_per_step_of_its_first_rounds": first}
    for k in range(warmup):
        w.step(k)
    w.sync()
    if settle_max > 0 and hasattr(pl, "torch"):
        used, last, prev_mean = 0, None, None
        while used < settle_max:
            evs = [(pl.event(), pl.event()) for _ in range(4)]
            for a, b in evs:
                pl.record(a)
                w.step(500_000 + used)      # (global indices away from the warm-up's and the timed steps')
                pl.record(b)
                used += 1
            w.sync()
            last = [pl.elapsed_ms(a, b) for a, b in evs]
            mean = sum(last) / 4.0
            if abs(last[3] - last[2]) <= 0.01 * last[2] and prev_mean is not None and abs(mean - prev_mean) <= 0.005 * prev_mean:
                break
            prev_mean = mean
        w.settle = {"untimed_steps": used, "last_group_ms": [round(x, 3) for x in last],
                    "rule": "groups of 4 until the last two steps agree within 1 %% and the group's mean is within 0.5 %% of the group before, at most %d steps" % settle_max}
    pl.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        pl.record(ev[k][0])
        w.step(warmup + k)
        pl.record(ev[k][1])
        if getattr(args, "step_gap_ms", 0.0) > 0.0:   # diagnostic only: what the kernel does when it is NOT launched back to back
            w.sync()
            time.sleep(args.step_gap_ms * 1e-3)
    pl.barrier()
    t1 = time.perf_counter()
    w.sync()
    elapsed = pl.max_over_ranks(t1 - t0)
    step_ms = [pl.elapsed_ms(a, b) for a, b in ev]
    w.check()
    if tel and tel.dir and getattr(args, "telemetry_s", 0.0) > 0.0:
        # The SMU's clock / power readings are moving averages that trail the load by more than a second (tools/power_probe.sh: 2 s from
        # idle to the sustained values), the timed region lasts tens of milliseconds: what the box does under THIS step is read in an
        # untimed run of the same back-to-back launches, from its last 40 %.
        tel.start()
        t_end, k = time.perf_counter() + args.telemetry_s, warmup + steps
        while time.perf_counter() < t_end:
            for _ in range(8):
                w.step(k)
                k += 1
            w.sync()
        tel.stop()
        w.telemetry = tel.summary(tail=0.4)
        if w.telemetry:
            w.telemetry["how"] = ("amdgpu hwmon in sysfs (freq1_input, power1_input), every 2 ms during %.1f s of the same step launched back to back after the timed "
                                  "region; statistics of the last 40 %% (the readings trail the load by over a second)" % args.telemetry_s)
            w.telemetry["sensor"] = tel.card
    return elapsed, step_ms


def box_state(tel, w=None):
    """"fast" / "slow" by the telemetry rule of HISTORY.md section 7 -- round 5 (profiles/r05_placement_probe.txt): the state belongs to the
    memory the trace lies in, not to the box; bench.py picks the fastest of a few candidate traces first (roofline.placement).  Under the
    headline kernel, which runs the socket at its power limit, the FAST state reads >= 1 310 W with the shader clock sagging below 2 280 MHz;
    the SLOW state reads 1 27x W at 2 3xx MHz (something other than socket power holds the chip back while the reported clock stays up;
    the kernel loses 17 %).  A kernel that does not reach the limit in either state cannot tell them apart: "below-the-power-limit"."""
    if isinstance(w, TerminalWorkload):
        return "not applicable (k_terminal_propagate does not reach the socket's power limit)"
    try:
        w_, mhz = tel["socket_power_w"]["median"], tel["sclk_mhz"]["median"]
    except (KeyError, TypeError):
        return "unknown (no telemetry)"
    if w_ >= 1310.0:
        return "fast"
    if w_ >= 1200.0 and 2280.0 <= mhz < 2385.0:
        return "slow"
    return "below-the-power-limit"   # (e.g. 1 27x W at the full 2 39x MHz: a kernel that never reaches the limit)


def roofline_of(w, step_ms, lib_version):
    avg_step_s = (sum(step_ms) / len(step_ms)) * 1e-3
    alg = w.bytes_per_unit * w.n
    achieved = alg / avg_step_s / 1e9
    kernel = w.kernel_name()
    per_launch = alg / w.launches_per_step
    per_launch = int(per_launch) if float(per_launch).is_integer() else per_launch
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kernel,
         "avg_launch_ms": avg_step_s * 1e3 / w.launches_per_step, "launches_per_step": w.launches_per_step,
         "avg_step_ms": avg_step_s * 1e3, "step_ms": [round(x, 3) for x in step_ms], "algorithmic_bytes_per_launch": per_launch,
         "algorithmic_bytes_per_unit": w.bytes_per_unit}
    if getattr(w, "bytes_stored_per_unit", None):   # terminal: the bytes the kernel stores (five f32 per row of the joined tracks), beside SURVEY 8(d)'s unit
        r["algorithmic_bytes_unit"] = "SURVEY.md 8(d): 75 B geometry + 15 B per track-second (u8 bin + f32 value of heading, altitude, speed)"
        r["output_bytes_per_unit"] = w.bytes_stored_per_unit
        r["output_bytes_unit"] = "75 B geometry + 20 B per row written (x y z heading speed as f32: createEncounter.m:162-167, the t = 0 row of an aircraft once)"
        r["frac_of_output_bytes"] = w.bytes_stored_per_unit * w.n / avg_step_s / 1e9 / HBM_PEAK_GBS
    if getattr(w, "placement", None):
        r["placement"] = w.placement
    if getattr(w, "settle", None):
        r["settle"] = w.settle
    if getattr(w, "prewarm", None):     # untimed launches ahead of the warm-up steps (--prewarm-s)
        r["prewarm"] = w.prewarm
    if getattr(w, "telemetry", None):   # the box, while it ran the timed region
        r["sclk_mhz"] = w.telemetry["sclk_mhz"]["median"]
        r["gpu_telemetry"] = w.telemetry
    if w.launches_per_step > 1:   # blocks of different kernel instances: they run on the ctx stream and three side streams
        r["launches_overlap"] = "avg_launch_ms is avg_step_ms / launches_per_step; single launches in a kernel trace overlap"
    rec = recorded_traffic(kernel, per_launch, lib_version, data_dependent=getattr(w, "bytes_data_dependent", False))
    insts = rec.pop("insts_valu", None)
    r.update(rec)
    if isinstance(w, TerminalWorkload):
        # this kernel's ceiling is vector ISSUE at two-thirds-full waves, not HBM (HISTORY.md section 11.1): the HBM fraction stays (SURVEY.md 8d's
        # unit), the bound is named for what it is, and the issue-slot fraction stands beside it when a committed PMC pass of this build has it
        r["bound"] = "valu-issue"
        if insts and r.get("sclk_mhz"):
            r["issue_frac"] = insts * 4.0 / (1024.0 * (r["avg_launch_ms"] * 1e-3) * (r["sclk_mhz"] * 1e6))
    sw = w.streaming_write() if hasattr(w, "streaming_write") else None
    if sw:   # measured AFTER the timed region (the buffers' contents were checked by then)
        r["streaming_write"] = sw
        r["frac_of_streaming_write"] = achieved / sw["GB/s"]
    return r


OTHER_CONFIGS = ["uncor_per_step", "cor", "cor_v2p1_like", "mixed", "terminal"]

# What the line's fields mean and how they are measured: static text, kept OUT of the line (`--write-notes` -> profiles/r06_bench_notes.json).
NOTES = {
    "step": "one pass of the hot path over one batch of fresh units (new global indices every step); 5 warm-up steps (headline: after --prewarm-s of "
            "untimed launches), then K timed steps bracketed by barrier + device synchronisation; value = units of all ranks / max-over-ranks wall time",
    "output": "dense trace of the DYNAMIC variables (u8 bin + f32 value per variable-second) + initial state (u8 + f32 per variable): "
              "5 n_i + 5 T n_d bytes per unit (3 635 uncor, 4 880 cor at T = 240); re-draws of static variables appear only in the event-list output; "
              "values are f32 at the boundary (f64 arithmetic inside, rounded on store), uniforms 32-bit",
    "roofline": "achieved = algorithmic bytes (SURVEY.md 8d) / average launch duration from HIP events on the launch stream; peak = 8 TB/s; traffic = "
                "WRITE_SIZE + 2 x FETCH_SIZE of a committed PMC pass of the same kernel, launch size and source hash (profiles/*_summary.json), else null; "
                "frac_of_streaming_write = achieved / what torch's fill kernel reaches on the same buffers on this box",
    "placement": "the trace comes from emgpu_trace_alloc (include/emgpu.h): the library allocates `candidates` traces, loads the device for 0.5 s, times "
                 "2 untimed + 5 timed launches of this workload's own call on each, twice, keeps the fastest and frees the others; ms = per candidate in "
                 "allocation order; first_allocation_ms = ms[0] = what a caller who keeps its first allocation gets on this box (profiles/r05_placement_probe.txt)",
    "configs": "BASELINE.json configs[2..4] at their single-GPU sizes + config 2 under PER_STEP, measured in this process after the headline: 5 warm-up steps, "
               "settle (groups of 4 untimed steps until the last two agree within 1 % and two groups' means within 0.5 %, at most 48), 10 timed steps; "
               "cpu = the oracle on all allowed cores, about 3 s per leg",
    "uncor_per_step": "config 2 under EMGPU_TRANSITION_PER_STEP: the true per-timestep DBN of dbn_sample.m:65-93 instead of the frozen-parent branch "
                      "(dbn_sample.m:97-135) the reference takes for this file; SURVEY.md section 7 hard part 1 asks for both",
    "cor": "cor_v1: stand-in for cor_v2p1.txt, which is absent from the reference mount; cor_v2p1_like: generator-made network with cor_v2p1's table sizes "
           "(em_model_manned_bayes_amd/synthetic.py, seed 0x5EED0003)",
    "mixed": "six uncor_*_v1p2 files, model = contiguous block of the global index range, one launch for the whole batch (6.25 M per GPU = 50 M on 8)",
    "terminal": "CorTerminalModel: terminal_v3_radar geometry network + 10 synthetic trajectory models (the trained files are absent); per step k_bn (geometry "
                "draw with rejection) + k_terminal_geo + k_terminal_propagate; frac follows SURVEY.md 8(d): 75 B + 15 B per track-second; "
                "frac_of_output_bytes charges what is stored: 75 B + 20 B per row written; bound valu-issue: issue_frac = SQ_INSTS_VALU x 4 cycles / "
                "(1 024 SIMDs x kernel time x sclk) from the committed PMC pass",
    "cpu_baseline": "oracle/em_oracle.c (scalar port of the reference algorithm), Philox mode, thread-private dense outputs, on `cores` threads = the cgroup's "
                    "CPU quota; single_thread_value beside it; MATLAB itself is not installed and cannot be timed",
    "host_path": "emgpu_sample_dbn_host end to end at --host-n trajectories x 240 s of uncor_1200code_v2p1, third call of each kind (the first pins memory), "
                 "measured BEFORE the headline (a process that has laid a torch tensor over a 36 GB trace -- this script's own plumbing -- and freed the trace copies at "
                 "47 GB/s afterwards: tools/host_path_bisect.sh): "
                 "dense_pinned = outputs in emgpu_host_alloc memory (the copy engine writes into the caller's arrays); dense_pageable = the caller's own "
                 "(pre-faulted) numpy arrays through the library's pinned staging + host threads; events_pinned = event lists only, packed on the device "
                 "(sum(ev_count) rows cross PCIe); GBps = bytes_d2h / total_ms; kernel_ms / d2h_ms / scatter_ms = the pipeline's phases (they overlap); "
                 "pinned_d2h_GBps = one 1 GiB hipMemcpy device -> pinned on this box; class_sample = UncorEncounterModel.sample(n, 240) (class level, "
                 "event lists + numpy reconstruction of out_samples / EncounterModelEvents): native_s = the library calls, format_s = the numpy / Python part",
    "box_state": "fast / slow / below-the-power-limit by socket power and shader clock under the step (HISTORY.md section 7); since round 5 known to follow "
                 "the trace's placement, which the library now picks",
}


def _r(x, nd=4):
    """Round floats for the line (4 significant decimals of a fraction, 3 of a millisecond value are what anybody reads)."""
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, list):
        return [_r(v, nd) for v in x]
    return x


ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_launch_ms", "launches_per_step", "step_ms",
             "algorithmic_bytes_per_launch", "algorithmic_bytes_per_unit", "output_bytes_per_unit", "frac_of_output_bytes", "frac_of_streaming_write",
             "issue_frac", "sclk_mhz")


def compact_roofline(r):
    out = {k: _r(r[k]) for k in ROOF_KEYS if k in r}
    pm = r.get("placement")
    if pm:
        out["placement"] = {k: pm[k] for k in ("candidates", "ms", "kept", "reused") if k in pm}
        out["first_allocation_ms"] = pm.get("first_allocation_ms") or None
    if "streaming_write" in r:
        out["streaming_write_GBps"] = round(r["streaming_write"]["GB/s"], 1)
    tel = r.get("gpu_telemetry") or {}
    if "socket_power_w" in tel:
        out["socket_power_w"] = tel["socket_power_w"]["median"]
    return out


def compact_cpu(c):
    return {"value": _r(c["value"]), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "single_thread_value": _r(c["single_thread_value"]),
            "sample": c.get("sample_short", c["sample"])}


def other_configs(args, pl, lib_version, detail):
    """BASELINE.json configs[2..4] (+ the cor_v2p1-sized stand-in, + config 2 under PER_STEP) at their full single-GPU sizes, measured in this
    process after the headline: each entry carries its own ms_per_step, kernel and roofline figures on that config's algorithmic bytes."""
    import copy
    res = {}
    for name in OTHER_CONFIGS:
        a = copy.copy(args)
        cfg = CONFIGS[name]
        a.config, a.n, a.model, a.per_step = name, 0, None, bool(cfg.get("per_step", False))
        a.prewarm_s = float(getattr(args, "prewarm_s", 0.0) or 0.0)   # (the CPU leg of the entry before left the GPU idle for seconds)
        w = None
        try:
            w = (TerminalWorkload if name == "terminal" else DbnWorkload)(a, cfg, pl, 0, 1)
            elapsed, step_ms = measure(w, pl, a, args.other_warmup, args.other_steps, settle_max=args.settle_max)
            roof = roofline_of(w, step_ms, lib_version)
            cr = compact_roofline(roof)
            e = {"value": _r(w.n * args.other_steps / elapsed), "unit": cfg["unit"], "ms_per_step": _r(elapsed / args.other_steps * 1e3), "n": w.n,
                 "kernel": w.kernel_name(), "bound": cr.get("bound"), "frac": cr.get("frac"), "traffic": cr.get("traffic"),
                 "box_state": box_state(getattr(w, "telemetry", None), w)}
            for k in ("issue_frac", "frac_of_output_bytes", "frac_of_streaming_write", "first_allocation_ms"):
                if cr.get(k) is not None:
                    e[k] = cr[k]
            if "placement" in cr:
                e["placement_ms"] = cr["placement"]["ms"]
            if isinstance(w, TerminalWorkload):
                e["track_seconds"] = _r(w.track_seconds)
            detail["configs"][name] = {"metric": cfg["metric"], "steps": args.other_steps, "warmup": args.other_warmup, "config": w.config(), "roofline": roof}
            if not args.no_cpu_baseline:   # the same oracle beside every config, on a smaller sample (about 3 s per leg)
                c = w.cpu_baseline(args.cpu_sample, seconds=3.0)
                e["cpu"] = {"value": _r(c["value"]), "cores": c["cores"], "single_thread_value": _r(c["single_thread_value"])}
                detail["configs"][name]["cpu_baseline"] = c
            res[name] = e
        except Exception as ex:   # an entry that cannot run says so; the headline line is still printed
            res[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if w is not None and hasattr(w, "close"):
            w.close()
        w = None
        pl.release()
    return res


def host_path(args, pl, detail):
    """emgpu_sample_dbn_host end to end (VERDICT r5 next #2b / #3): the path every drop-in caller of UncorEncounterModel.sample is on.
    PCIe-inclusive figures: reported beside the headline, never as it."""
    import numpy as np
    import em_model_manned_bayes_amd as E
    from em_model_manned_bayes_amd import _lib as L
    native, t = pl.native, pl.torch
    tmp = tempfile.mkdtemp(prefix="emgpu_bench_host_")
    path = _materialize("uncor_1200code_v2p1", tmp)
    nm = native.NativeModel.load_txt(path)
    idx = _label_indices(nm)
    n, T, seed = int(args.host_n), DEFAULT_T, 0x5EED0002
    ctx = native.Context(pl.dev.index)
    out = {"n": n, "T": T}

    def summarise(st, units):
        return {"total_ms": _r(st["total_ms"]), "kernel_ms": _r(st["kernel_ms"]), "d2h_ms": _r(st["d2h_ms"]), "scatter_ms": _r(st["scatter_ms"]),
                "GBps": _r(st["bytes_d2h"] / st["total_ms"] / 1e6), "bytes_d2h": st["bytes_d2h"], "units_per_s": _r(units / st["total_ms"] * 1e3),
                "chunks": st["chunks"], "threads": st["threads"], "direct": st["direct"]}
    # what the box's copy engine reaches into pinned memory: one 1 GiB device -> pinned copy, best of 3
    dsrc = t.empty(1 << 30, dtype=t.uint8, device=pl.dev)
    hdst = t.empty(1 << 30, dtype=t.uint8, pin_memory=True)
    best = 1e9
    for _ in range(4):
        a, b = pl.event(), pl.event()
        pl.record(a)
        hdst.copy_(dsrc, non_blocking=True)
        pl.record(b)
        t.cuda.synchronize()
        best = min(best, pl.elapsed_ms(a, b))
    out["pinned_d2h_GBps"] = _r((1 << 30) / best / 1e6)
    del dsrc, hdst
    # dense, pinned outputs (third call: the first pins the pool's blocks, the second still warms the mappings up)
    for rep in range(3):
        r = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=True, want_events=False, pinned=True, raw=True, **idx)
        st = r["host_stats"]
        del r
    out["dense_pinned"] = summarise(st, n)
    detail["host_path"]["dense_pinned"] = st
    # dense, the caller's own pageable arrays (touched once before: a fresh numpy array's first touch is the kernel's page-fault rate, not ours)
    ni, nd, G4 = nm.n_initial, nm.n_dyn, (T + 3) // 4
    ib, iv = np.zeros((ni, n), np.uint8), np.zeros((ni, n), np.float32)
    db, dv = np.zeros((G4, nd, n), np.uint32), np.zeros((G4, nd, n, 4), np.float32)
    att = np.zeros(n, np.int32)
    p, _keep = native.make_params(n, T, seed, **idx)
    o = L.SampleOut()
    o.init_bin, o.init_val, o.dyn_bin, o.dyn_val, o.attempts = ib.ctypes.data, iv.ctypes.data, db.ctypes.data, dv.ctypes.data, att.ctypes.data
    import ctypes as C
    for rep in range(3):
        t0 = time.perf_counter()
        L.check(L.lib().emgpu_sample_dbn_host(ctx._h, nm._h, C.byref(p), C.byref(o)))
        cold_or_warm = time.perf_counter() - t0
        if rep == 0:
            out["dense_pageable_first_call_ms"] = _r(cold_or_warm * 1e3)
    st = ctx.host_stats()
    out["dense_pageable"] = summarise(st, n)
    detail["host_path"]["dense_pageable"] = st
    del ib, iv, db, dv, att
    # event lists only (what the class layer asks for): packed on the device
    for rep in range(3):
        r = native.sample_dbn_host(ctx, nm, n, T, seed, want_dense=False, want_events=True, event_cap=256, pinned=True, raw=True, **idx)
        st = r["host_stats"]
        del r
    e = summarise(st, n)
    e["event_rows"] = st["event_rows"]
    e["bytes_if_unpacked"] = n * 256 * 8
    out["events_pinned"] = e
    detail["host_path"]["events_pinned"] = st
    ctx.trim()
    # the class level: UncorEncounterModel.sample(n_class, 240) -- host arrays, cells and EncounterModelEvents like the reference returns
    n_class = min(n, 100_000)
    mdl = E.UncorEncounterModel(path)
    mdl.sample(2048, T, seed=1, ctx=ctx)       # (warm: tables uploaded, pool blocks pinned)
    t0 = time.perf_counter()
    mdl.sample(n_class, T, seed=seed, ctx=ctx)
    dt = time.perf_counter() - t0
    tm = dict(mdl.last_sample_timing)
    out["class_sample"] = {"n": n_class, "total_s": _r(dt), "native_s": _r(tm["native_s"]), "kernel_ms": _r(tm["kernel_ms"]), "d2h_ms": _r(tm["d2h_ms"]),
                           "format_s": _r(tm["format_s"]), "units_per_s": _r(n_class / dt)}
    detail["host_path"]["class_sample"] = tm
    del mdl, ctx
    pl.release()
    return out


def run_rank(args, rank, local_rank, world, pl=None, out=sys.stdout):
    """One rank of the benchmark.  `pl` (plumbing) is TorchRocm unless a test injects its own."""
    pl = pl or TorchRocm(rank, local_rank, world, args.oversubscribe)
    default_run = world == 1 and args.config == "uncor" and not args.model and not args.n and hasattr(pl, "release")
    hp, hp_detail = None, {"host_path": {}}
    if rank == 0 and default_run and not args.no_host_path:
        # FIRST, in the state a consumer's process is in: once this process has laid a torch tensor over a 36 GB trace (the plumbing of the checks
        # below) and then FREED that trace, the host path's copies run at 47 instead of 56.5 GB/s (40 instead of 55 through the staging buffers) --
        # tools/host_path_bisect.sh; a trace that is only touched through the C ABI, or that stays alive, does not have that effect
        try:
            hp = host_path(args, pl, hp_detail)
        except Exception as ex:
            hp = {"error": "%s: %s" % (type(ex).__name__, ex)}
    w = make_workload(args, pl, rank, world)
    elapsed, step_ms = measure(w, pl, args, args.warmup, args.steps)
    if getattr(args, "ranges_out", None) and hasattr(w, "digest") and hasattr(pl, "torch"):
        os.makedirs(args.ranges_out, exist_ok=True)
        with open(os.path.join(args.ranges_out, "rank%d.json" % rank), "w") as f:
            # (the warm-up and timed steps: pre-warm, placement and settle steps use step numbers from 500 000 up)
            json.dump({"rank": rank, "world": world, "n": w.n, "ranges": [r for r in w.ranges if r["step"] < 500_000], "digest": w.digest()}, f)
    line = None
    if rank == 0:
        from em_model_manned_bayes_amd import _lib as L
        cfg = CONFIGS[args.config]
        total = w.n * world * args.steps
        lib_version = L.lib().emgpu_version().decode()
        kernel = w.kernel_name()
        roof = roofline_of(w, step_ms, lib_version)
        wc = w.config()
        detail = {"config": wc, "roofline": roof, "configs": {}, "host_path": hp_detail["host_path"], "notes": NOTES}
        line = {
            "metric": cfg["metric"], "value": total / elapsed, "unit": cfg["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 draws and compares; f64 dediscretize; f32 values stored", "data": "synthetic",
            "config": {"workload": wc["workload"], "transition_mode": wc.get("transition_mode"), "kernel": kernel, "lib": lib_version,
                       "philox_rounds": int(L.lib().emgpu_philox_rounds()), "trace_ld": wc.get("trace_ld"), "models": wc.get("models"),
                       "launches_per_step": wc.get("launches_per_step"), "box_state": box_state(getattr(w, "telemetry", None), w),
                       "notes": "profiles/r06_bench_notes.json"},
            "roofline": compact_roofline(roof),
        }
        if isinstance(w, TerminalWorkload):
            line["config"]["track_seconds_per_encounter"] = wc.get("track_seconds_per_encounter")
        if getattr(args, "verbose_line", False):
            line["roofline"], line["config"] = roof, dict(wc, **line["config"])
        if getattr(pl, "shared", False):
            line["oversubscribed"] = True
        if args.step_gap_ms > 0.0:
            line["diagnostic"] = "idle gaps of %g ms between the launches (--step-gap-ms): value and ms_per_step include them" % args.step_gap_ms
        if world == 1 and not args.no_cpu_baseline:
            c = w.cpu_baseline(args.cpu_sample)
            detail["cpu_baseline"] = c
            line["cpu_baseline"] = compact_cpu(c)
        if default_run and not args.no_other_configs:
            if hasattr(w, "close"):
                w.close()
            w = None
            pl.release()
            line["configs"] = other_configs(args, pl, lib_version, detail)
        if hp is not None:
            line["host_path"] = hp
        sys.stderr.write("DETAIL " + json.dumps(detail) + "\n")
        if getattr(args, "detail_out", None):
            with open(args.detail_out, "w") as f:
                json.dump(dict(detail, line=line), f, indent=1)
        out.write(json.dumps(line) + "\n")
        out.flush()
    pl.finish()
    return line


def recorded_traffic(kernel_name, algorithmic_bytes, lib_version, data_dependent=False):
    """roofline.traffic: HBM bytes per launch from the PMC passes (WRITE_SIZE + 2 x FETCH_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md) of a committed profile of the SAME kernel, launch size
    AND library build (profiles/*_summary.json, produced by tools/profile_bench.sh: counters need
    their own rocprofv3 passes and cannot be read from inside this process).  The library carries a
    hash of its sources (emgpu_version()); a summary recorded from other sources does not count:
    null rather than a stale figure."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            s = json.load(open(f))
        except Exception:
            continue
        line = s.get("bench_line", {})
        theirs = line.get("roofline", {}).get("algorithmic_bytes_per_launch")
        # the same launch size: equal bytes -- or, where the bytes are a property of the sampled data (terminal: 75 + 20 B x the track-seconds
        # the batch happened to produce, which moves in the fourth digit with the step's index range), within half a percent
        same_size = theirs == algorithmic_bytes or (data_dependent and isinstance(theirs, (int, float))
                                                    and abs(theirs - algorithmic_bytes) <= 0.005 * algorithmic_bytes)
        same = (line.get("config", {}).get("kernel") == kernel_name      # the summary's own bench line ran this kernel variant ...
                and same_size                                            # ... on this launch size ...
                and line.get("config", {}).get("lib") == lib_version)     # ... from these sources
        if same and "hbm_traffic_bytes_per_launch" in s:
            best = (s["hbm_traffic_bytes_per_launch"], os.path.basename(f), s.get("pmc_per_launch", {}).get("SQ_INSTS_VALU"))
    if best is None:
        return {"traffic": None}
    out = {"traffic": best[0], "traffic_source": "profiles/" + best[1]}
    if best[2]:
        out["insts_valu"] = best[2]
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.write_notes:
        with open(args.write_notes, "w") as f:
            json.dump(NOTES, f, indent=1)
        return 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 3
    run_rank(args, rank, local_rank, world)
    return 0


if __name__ == "__main__":
    sys.exit(main())
