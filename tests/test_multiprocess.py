"""World-size-2 gloo test of the N>1 path: ranks shard the global index range, there is no
collective on the data path, and the union of the shards equals the single-process result.
No GPU here, so the per-shard sampler is the oracle; the same property is checked for the HIP
path on one GPU in tests/test_gpu_parity.py::test_sharded_calls_equal_one_call."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, T, seed, path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from em_model_manned_bayes_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n_total, rank, world)
    om = O.OracleModel(O.parse_model_txt(path))
    r = O.uncor_sample(om, hi - lo, T, seed, first_index=lo, want_events=False)
    # the only communication: a barrier and the max-over-ranks clock, as in bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    q.put((rank, lo, hi, r["dense_bin"], r["dense_val"], r["init_val"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(model_dir):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from em_model_manned_bayes_amd import em_io
    path = em_io.materialize_model("uncor_1200code_v2p1", model_dir)
    n_total, T, seed, world = 61, 48, 0x5EED0004, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, T, seed, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == n_total
    full = O.uncor_sample(O.OracleModel(O.parse_model_txt(path)), n_total, T, seed, want_events=False)
    assert np.array_equal(np.concatenate([g[3] for g in got]), full["dense_bin"])
    assert np.array_equal(np.concatenate([g[4] for g in got]), full["dense_val"])
    assert np.array_equal(np.concatenate([g[5] for g in got]), full["init_val"])
