"""Test plumbing for bench.run_rank on a machine WITHOUT a GPU (tests/test_multiprocess.py).

bench.py's rank logic -- which global indices a rank samples at step k, how a mixed batch is cut
into (model, first_index, count) launches, where a launch lands in the rank's trace, the barrier
and max-over-ranks clock, the JSON line -- is product code and runs here unchanged, with the
product's own native.make_params / native.mixed_blocks / NativeModel.load_txt / struct layouts.
Only the two things that need a GPU are replaced: device memory (numpy arrays behind the same
data_ptr() interface) and the kernel launch itself, which this shim answers with the CPU ORACLE
writing the device-native layout.  This is test infrastructure: nothing in the product imports it.
"""
import ctypes as C
import time

import numpy as np
import torch
import torch.distributed as dist

import oracle as O
from em_model_manned_bayes_amd import native as real_native, _lib as L


class _Buf:
    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype=dtype)

    def data_ptr(self):
        return self.a.ctypes.data


_live = {}   # data_ptr -> numpy array (how the fake launch finds the rank's buffers again)


class _Ctx:
    def __init__(self):
        self.kernel = "oracle(em_uncor_sample_batch)"
        self.launches = []
        self.last = 0   # oracle calls that answered the last sample_dbn_*_device call

    def sync(self):
        pass

    def last_kernel(self):
        return self.kernel

    def last_launches(self):
        return self.last


class _Model:
    """the product's NativeModel (C++ loader) + the oracle's view of the same file"""

    def __init__(self, path):
        self.nm = real_native.NativeModel.load_txt(path)
        self.om = O.OracleModel(O.parse_model_txt(path))
        self.n_initial, self.n_dyn = self.nm.n_initial, self.nm.n_dyn
        self._h = self.nm._h

    def get_labels(self, f):
        return self.nm.get_labels(f)


class NativeShim:
    """what bench.py sees as `pl.native`"""
    make_params = staticmethod(real_native.make_params)          # product
    mixed_blocks = staticmethod(real_native.mixed_blocks)        # product (libemgpu, host only)

    class NativeModel:
        load_txt = staticmethod(lambda path: _Model(path))

    @staticmethod
    def _write(model, n, first, T, seed, per_step, col, ptrs):
        r = O.uncor_sample(model.om, n, T, seed, first_index=first, per_step=per_step, want_events=False)
        ib, iv, db, dv = (_live[ptrs[k]] for k in ("init_bin", "init_val", "dyn_bin", "dyn_val"))
        ib[:, col: col + n] = r["init_bin"].T
        iv[:, col: col + n] = r["init_val"].T.astype(np.float32)
        G4, nd = db.shape[0], db.shape[1]
        bins = np.zeros((n, G4 * 4, nd), dtype=np.uint8)
        vals = np.zeros((n, G4 * 4, nd), dtype=np.float32)
        bins[:, :T], vals[:, :T] = r["dense_bin"], r["dense_val"].astype(np.float32)
        db.view(np.uint8).reshape(G4, nd, -1, 4)[:, :, col: col + n, :] = bins.reshape(n, G4, 4, nd).transpose(1, 3, 0, 2)
        dv[:, :, col: col + n, :] = vals.reshape(n, G4, 4, nd).transpose(1, 3, 0, 2)

    @staticmethod
    def sample_dbn_device(ctx, model, p, ld=0, col_offset=0, **ptrs):
        ctx.launches.append((0, int(p.first_index), int(p.n)))
        ctx.last = 1
        NativeShim._write(model, int(p.n), int(p.first_index), int(p.sample_time), int(p.seed),
                          p.transition_mode == L.TRANSITION_PER_STEP, int(col_offset), ptrs)

    @staticmethod
    def sample_dbn_blocks_device(ctx, models, p, blocks, ld=0, col_offset=0, **ptrs):
        ctx.last = len(blocks)
        for (m, first, cnt) in blocks:
            assert int(p.first_index) <= first and first + cnt <= int(p.first_index) + int(p.n)
            ctx.launches.append((m, first, cnt))
            NativeShim._write(models[m], cnt, first, int(p.sample_time), int(p.seed), p.transition_mode == L.TRANSITION_PER_STEP,
                              int(col_offset) + first - int(p.first_index), ptrs)


class CpuGloo:
    """bench.TorchRocm's interface on gloo + host memory"""
    native = NativeShim

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.ctx = _Ctx()
        self.bufs = []
        self.calls = []     # the order of the collectives and the clock reads (what may and may not land between t0 and t1)

    def context(self):
        return self.ctx

    def empty(self, shape, dtype):
        b = _Buf(shape, {"uint8": np.uint8, "float32": np.float32, "int32": np.uint32}[dtype])
        _live[b.data_ptr()] = b.a
        self.bufs.append(b)
        return b

    def barrier(self):
        self.calls.append("barrier")
        dist.barrier()

    def event(self):
        return [0.0]

    def record(self, ev):
        self.calls.append("record")
        ev[0] = time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b[0] - a[0]) * 1e3

    def max_over_ranks(self, x):
        self.calls.append("max")
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        dist.barrier()
        dist.destroy_process_group()
