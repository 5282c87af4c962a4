"""RUN_1_emsample.m + RUN_2_sample2track.m: sample a model into initial.txt / transition.txt, then turn the
files into 1 Hz tracks (CSV).  Usage: python examples/run_emsample_sample2track.py [model] [out_dir]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import em_io

name = sys.argv[1] if len(sys.argv) > 1 else "uncor_1200code_v2p1"
out = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp()
path = em_io.materialize_model(name, out)
fi, ft = os.path.join(out, "initial.txt"), os.path.join(out, "transition.txt")
E.em_sample(path, initial_output_filename=fi, transition_output_filename=ft, num_initial_samples=100, num_transition_samples=60,
            rng_seed=42)                                                      # RUN_1_emsample.m
is_good, T_initial = E.sample2track(path, fi, ft, out_dir_parent=os.path.join(out, "tracks"), rng_seed=42)   # RUN_2_sample2track.m
print("%d of %d tracks written under %s" % (int(is_good.sum()), is_good.size, os.path.join(out, "tracks")))
