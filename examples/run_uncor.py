#!/usr/bin/env python3
"""examples/run_uncor.py -- the sampling part of the reference's demo script code/matlab/RUN_uncor.m
(lines 16-50) on the GPU: instantiate the model, preset the start distribution, draw samples.
The .track calls of RUN_uncor.m (:54-70) need em-core and are outside this package.

    python examples/run_uncor.py [parameters_filename]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import em_model_manned_bayes_amd as E  # noqa: E402

# Inputs (RUN_uncor.m:16-26)
parameters_filename = sys.argv[1] if len(sys.argv) > 1 else None   # default: uncor_1200only_fwse_v1p2 (UncorEncounterModel.m:26)
n_samples, sample_time, init_seed = 1, 210, 1

# Instantiate object (RUN_uncor.m:29)
mdl = E.UncorEncounterModel(parameters_filename=parameters_filename)

# Start distribution (RUN_uncor.m:35-48): G = 1 (CONUS), A = 4 (other airspace), L = 2 ([500, 1200) ft AGL)
start = [None] * mdl.n_initial
start[0], start[1], start[2] = 1, 4, 2
mdl.start = start

# Samples (RUN_uncor.m:50)
out_inits, out_events, out_samples, out_EME = mdl.sample(n_samples, sample_time, seed=init_seed)
print("labels      :", [lab.strip('"') for lab in mdl.labels_initial])
print("out_inits   :", out_inits[0])
print("out_events  : %d rows [dt var value]; first rows:\n%s" % (out_events[0].shape[0], out_events[0][:5]))
print("out_samples :", out_samples[0].shape, "(n_initial x T)")
print("out_EME     : %d control rows [t_s dh_fps dpsi_radps dv_ftpss]; first rows:\n%s" % (out_EME[0].event.shape[0], out_EME[0].event[:3]))

# The same model at scale, dense trace resident on the GPU: see bench.py
