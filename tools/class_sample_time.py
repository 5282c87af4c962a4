import time, tempfile, sys
sys.path.insert(0, ".")
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import em_io
mdl = E.UncorEncounterModel(em_io.materialize_model("uncor_1200code_v2p1", tempfile.mkdtemp()))
mdl.sample(2048, 240, seed=1)
for n in (100000, 100000):
    t0 = time.perf_counter(); mdl.sample(n, 240, seed=2); dt = time.perf_counter() - t0
    print(n, "%.3f s" % dt, {k: round(v, 3) if isinstance(v, float) else v for k, v in mdl.last_sample_timing.items()})
