"""World-size-2 gloo tests of the N>1 path (no GPU here).

test_two_rank_bench_runs_the_products_rank_logic: both ranks run bench.run_rank -- the product's
make_params / step_first_index / mixed_blocks / loader / JSON line -- with only device memory and
the kernel launch replaced (tests/mp_plumbing.py answers a launch with the CPU oracle), and the
union of what the ranks sampled must equal one oracle run over the global index range.
The same property is checked for the HIP path itself on a GPU in tests/test_gpu_parity.py
(test_sharded_calls_equal_one_call, test_two_rank_processes_on_one_gpu, test_multi_device_driver).
"""
import io
import json
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, argv, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import bench
    import mp_plumbing
    args = bench.parse_args(argv)
    pl = mp_plumbing.CpuGloo(rank, world)
    buf = io.StringIO()
    bench.run_rank(args, rank, rank, world, pl=pl, out=buf)
    q.put((rank, buf.getvalue(), pl.ctx.launches, [b.a for b in pl.bufs], pl.calls))


def _run(world, argv):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _unpack(bufs, T, n):
    """the rank's trace (leading dimension padded by bench.py) -> its first n columns in user-facing shapes"""
    from em_model_manned_bayes_amd import native
    ib, iv, db, dv = bufs
    return ib.T[:n], iv.T[:n], native.unpack_dyn_bin(db, T)[:n], native.unpack_dyn_val(dv, T)[:n]


def test_two_rank_bench_runs_the_products_rank_logic(model_dir):
    import oracle as O
    from em_model_manned_bayes_amd import em_io
    n, T, world, steps, warmup = 37, 24, 2, 2, 1
    got = _run(world, ["--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--n", str(n), "--seconds", str(T), "--no-cpu-baseline"])
    line = json.loads(got[0][1])
    assert got[1][1] == ""                                    # only rank 0 prints
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == "weak"
    assert abs(line["value"] - n * world * steps / (line["ms_per_step"] * steps * 1e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["algorithmic_bytes_per_unit"] == 5 * 7 + 5 * T * 3
    # every step covers a fresh contiguous global range, rank r the r-th part of it
    for r in range(world):
        assert got[r][2] == [(0, (k * world + r) * n, n) for k in range(warmup + steps)]
    # the process group's FIRST barrier and all-reduce come before anything is timed (RCCL sets its communicator up in them: hundreds of
    # milliseconds at 8 ranks that must not land in a 0.1 s timed region); the timed region itself is barrier, 2 x steps records, barrier, max
    for r in range(world):
        calls = got[r][4]
        assert calls[:2] == ["barrier", "max"], calls
        first_record = calls.index("record")
        assert calls[first_record - 1] == "barrier" and calls[first_record: first_record + 2 * steps] == ["record"] * (2 * steps)
        assert calls[first_record + 2 * steps: first_record + 2 * steps + 2] == ["barrier", "max"], calls
    # the buffers hold the last step: ranks 0 and 1 together == one oracle run over that step's global range
    k = warmup + steps - 1
    om = O.OracleModel(O.parse_model_txt(em_io.materialize_model("uncor_1200code_v2p1", model_dir)))
    full = O.uncor_sample(om, n * world, T, 0x5EED0002, first_index=k * world * n, want_events=False)
    parts = [_unpack(g[3], T, n) for g in got]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full["init_bin"])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full["init_val"].astype(np.float32))
    assert np.array_equal(np.concatenate([p[2] for p in parts]), full["dense_bin"])
    assert np.array_equal(np.concatenate([p[3] for p in parts]), full["dense_val"].astype(np.float32))


def test_two_rank_mixed_batch_blocks(model_dir):
    """config 4 in miniature: 2 ranks x 50 trajectories, six models in contiguous blocks of the step's range;
    each rank's launches tile its shard and land in ONE trace per rank."""
    import bench
    import oracle as O
    from em_model_manned_bayes_amd import em_io, sharding
    n, T, world = 50, 16, 2
    got = _run(world, ["--gpus", "2", "--config", "mixed", "--steps", "1", "--warmup", "0", "--n", str(n), "--seconds", str(T), "--no-cpu-baseline"])
    line = json.loads(got[0][1])
    assert line["n_gpus"] == 2 and line["config"]["models"] == bench.V1P2
    total = n * world
    oms = [O.OracleModel(O.parse_model_txt(em_io.materialize_model(nm, model_dir))) for nm in bench.V1P2]
    for r in range(world):
        launches = got[r][2]
        lo, hi = sharding.shard_range(total, r, world)
        assert launches == sharding.mixed_batch_blocks(total, 6, lo, hi)
        assert sum(c for _, _, c in launches) == n and launches[0][1] == lo
        ib, iv, db, dv = _unpack(got[r][3], T, n)
        for (m, first, cnt) in launches:
            ref = O.uncor_sample(oms[m], cnt, T, 0x5EED0004, first_index=first, want_events=False)
            sl = slice(first - lo, first - lo + cnt)
            assert np.array_equal(ib[sl], ref["init_bin"]) and np.array_equal(db[sl], ref["dense_bin"])
            assert np.array_equal(dv[sl], ref["dense_val"].astype(np.float32))


def test_eight_rank_dry_run_covers_the_range_once(model_dir):
    """The 8-GPU node layout on CPU: eight gloo ranks run bench.run_rank on the mixed batch (config 4) and
    on the headline config; every step's global range is covered exactly once by disjoint rank shards, the line says
    n_gpus 8, and every rank's model blocks tile its shard."""
    import bench
    import oracle as O
    from em_model_manned_bayes_amd import em_io, sharding
    world, n, T = 8, 13, 8
    got = _run(world, ["--gpus", "8", "--config", "mixed", "--steps", "2", "--warmup", "1", "--n", str(n), "--seconds", str(T), "--no-cpu-baseline"])
    line = json.loads(got[0][1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and all(g[1] == "" for g in got[1:])
    assert abs(line["value"] - n * world * 2 / (line["ms_per_step"] * 2 * 1e-3)) < 1e-6 * line["value"]
    total = n * world
    oms = [O.OracleModel(O.parse_model_txt(em_io.materialize_model(nm, model_dir))) for nm in bench.V1P2]
    for k in range(3):
        covered = []
        for r in range(world):
            lo, hi = sharding.shard_range(total, r, world)
            want = [(m, k * total + f, c) for (m, f, c) in sharding.mixed_batch_blocks(total, 6, lo, hi)]
            mine = [b for b in got[r][2] if k * total <= b[1] < (k + 1) * total]
            assert mine == want and sum(c for _, _, c in mine) == n
            covered += [(f, f + c) for _, f, c in mine]
        covered.sort()
        assert covered[0][0] == k * total and covered[-1][1] == (k + 1) * total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))      # disjoint and gap-free
    for r in (0, 3, 7):   # the last step's trace of three ranks against the oracle, block by block
        lo, _ = sharding.shard_range(total, r, world)
        ib, iv, db, dv = _unpack(got[r][3], T, n)
        for (m, first, cnt) in [b for b in got[r][2] if b[1] >= 2 * total]:
            ref = O.uncor_sample(oms[m], cnt, T, 0x5EED0004, first_index=first, want_events=False)
            sl = slice(first - 2 * total - lo, first - 2 * total - lo + cnt)
            assert np.array_equal(db[sl], ref["dense_bin"]) and np.array_equal(dv[sl], ref["dense_val"].astype(np.float32))
    got = _run(world, ["--gpus", "8", "--steps", "1", "--warmup", "0", "--n", "5", "--seconds", "6", "--no-cpu-baseline"])
    assert json.loads(got[0][1])["n_gpus"] == 8
    assert sorted(g[2][0][1] for g in got) == [5 * r for r in range(8)] and all(g[2][0][2] == 5 for g in got)


def test_the_launcher_counts_gpus_without_a_gpu_runtime(monkeypatch, tmp_path):
    """bench.launch_ranks decides from visible_gpu_count(): sysfs, no torch / HIP in the parent."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpu_count(); "
            "assert 'torch' not in sys.modules, 'the launcher imported torch'; print(n)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    assert int(out.stdout.decode().strip()) >= 0


def test_bench_never_runs_fewer_ranks_than_asked(monkeypatch):
    """`python bench.py --gpus 2` must start two ranks or fail: no silent one-GPU run."""
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "2"]) == 3            # launcher: no GPUs here -> refuses before spawning anything
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.main(["--gpus", "2"]) == 3            # under a launcher whose world disagrees with --gpus
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    assert bench.main(["--gpus", "1"]) == 3


def test_shard_arithmetic_matches_the_library():
    from em_model_manned_bayes_amd import native, sharding
    for n_total, world in ((0, 3), (7, 8), (61, 2), (50_000_000, 8), (10, 1)):
        for r in range(world):
            assert native.shard_range(n_total, r, world) == sharding.shard_range(n_total, r, world)
    for n_total, nm, lo, hi in ((50, 6, 10, 30), (50_000_000, 6, 6_250_000, 12_500_000), (5, 6, 0, 5)):
        assert native.mixed_blocks(n_total, nm, lo, hi) == sharding.mixed_batch_blocks(n_total, nm, lo, hi)


def test_bench_host_side_helpers(tmp_path, monkeypatch):
    """bench.usable_cores honours a cgroup CPU quota; bench.recorded_traffic matches a profile summary only for the same kernel, library
    build and launch size -- exactly, or within half a percent where the size is a property of the sampled data (terminal)."""
    import bench
    cores, note = bench.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1) and "hardware threads" in note
    prof = tmp_path / "profiles"
    prof.mkdir()
    line = {"config": {"kernel": "k_x<1>", "lib": "emgpu test src:abc"}, "roofline": {"algorithmic_bytes_per_launch": 1000}}
    json.dump({"bench_line": line, "hbm_traffic_bytes_per_launch": 1234.0}, open(prof / "t1_summary.json", "w"))
    line2 = {"config": {"kernel": "k_t", "lib": "emgpu test src:abc"}, "roofline": {"algorithmic_bytes_per_launch": 5000.25}}
    json.dump({"bench_line": line2, "hbm_traffic_bytes_per_launch": 9999.0}, open(prof / "t2_summary.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.recorded_traffic("k_x<1>", 1000, "emgpu test src:abc")["traffic"] == 1234.0
    assert bench.recorded_traffic("k_x<1>", 1001, "emgpu test src:abc")["traffic"] is None        # another launch size
    assert bench.recorded_traffic("k_x<1>", 1000, "emgpu test src:def")["traffic"] is None        # another build
    assert bench.recorded_traffic("k_y<1>", 1000, "emgpu test src:abc")["traffic"] is None        # another kernel
    assert bench.recorded_traffic("k_t", 5010, "emgpu test src:abc", data_dependent=True)["traffic"] == 9999.0   # data-dependent bytes: 0.2 % apart
    assert bench.recorded_traffic("k_t", 5010, "emgpu test src:abc")["traffic"] is None           # ... only for a workload that says so
    assert bench.recorded_traffic("k_t", 5100, "emgpu test src:abc", data_dependent=True)["traffic"] is None      # 2 % apart
