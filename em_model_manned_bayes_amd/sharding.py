"""Multi-GPU sharding of a batch of independent trajectories.

Every trajectory is keyed by its GLOBAL sample index (the Philox counter), so a batch can be cut
into contiguous index ranges, one per rank, with no collective on the data path and results that
do not depend on the number of GPUs.
"""


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of rank `rank` when n_total indices are split over `world` ranks
    (the first n_total % world ranks get one extra).  Same arithmetic as emgpu_shard_range in the
    library (tests compare the two); kept in Python so that launchers can plan without loading it."""
    n_total, rank, world = int(n_total), int(rank), int(world)
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def step_first_index(step, rank, world, n_per_rank):
    """Weak-scaling benchmark layout: step k of rank r samples indices [(k*world + r)*n, ... + n)."""
    return (int(step) * int(world) + int(rank)) * int(n_per_rank)


def mixed_batch_blocks(n_total, n_models, lo=0, hi=None):
    """Mixed-model batch (BASELINE.json configs[3]): the global index range [0, n_total) is cut into
    n_models contiguous blocks, block m sampled from model m.  Returns, for the sub-range [lo, hi)
    owned by one rank, the list of (model, first_index, count) calls to make -- each call is one
    kernel launch touching one table set."""
    hi = n_total if hi is None else hi
    out = []
    for m in range(n_models):
        a, b = shard_range(n_total, m, n_models)
        a2, b2 = max(a, lo), min(b, hi)
        if b2 > a2:
            out.append((m, a2, b2 - a2))
    return out


def run_sharded(model, n, sample_time, seed, devices=None, first_index=0, **kw):
    """One process, several GPUs: sample n trajectories of `model` (a native.NativeModel) with the batch
    split over `devices` (default: every visible device) inside one library call -- one host thread and
    one HIP stream per device (emgpu_sample_dbn_multi_host).  Same keywords and the same result dict as
    native.sample_dbn_host; the result does not depend on the number of devices."""
    from . import native
    if devices is None:
        devices = range(native.device_count())
    ctxs = [native.default_context(int(d)) for d in devices]
    if not ctxs:
        raise native.L.EmgpuError(native.L.ERR_NO_DEVICE, "no HIP device visible: the product path has no CPU fallback")
    return native.sample_dbn_host(ctxs, model, n, sample_time, seed, first_index=first_index, **kw)
