import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
from em_model_manned_bayes_amd import em_io, native
from util import uncor_indices
tmp = tempfile.mkdtemp()
path = em_io.materialize_model("uncor_1200code_v2p1", tmp)
nm = native.NativeModel.load_txt(path); pp = O.parse_model_txt(path)
ctx = native.Context(0)
n, T, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 240, 0xABCDEF12345
first = 2**33 + 17
idx = uncor_indices(pp)
ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, first_index=first, want_events=False)
got = native.sample_dbn_host(ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=False, **idx)
print(got["kernel"])
print("bins equal", np.array_equal(got["dyn_bin"], ref["dense_bin"]))
rv = ref["dense_val"].astype(np.float32)
bad = np.argwhere(got["dyn_val"] != rv)
print("value mismatches", len(bad), "of", rv.size)
for b in bad[:20]:
    i, c, k = b
    print(i, c, k, got["dyn_val"][i, c, k], rv[i, c, k], "bin", got["dyn_bin"][i, c, k], "prev", got["dyn_val"][i, c - 1, k], rv[i, c - 1, k])
if len(bad):
    print("traj with mismatch:", np.unique(bad[:, 0])[:30], "lanes", np.unique(bad[:, 0] % 64)[:64])
    print("cols", np.unique(bad[:, 1])[:40])
bb = np.argwhere(got["dyn_bin"] != ref["dense_bin"])
print("bin mismatches", len(bb))
for b in bb[:10]:
    i, c, k = b
    print(i, c, k, got["dyn_bin"][i, c, k], ref["dense_bin"][i, c, k])
print("init bin eq", np.array_equal(got["init_bin"].astype(np.int32), ref["init_bin"]), "attempts eq", np.array_equal(got["attempts"], ref["attempts"]), "max attempts", ref["attempts"].max())
ib = np.argwhere(got["init_bin"].astype(np.int32) != ref["init_bin"])
print(ib[:10])
for i in np.unique(ib[:, 0])[:5]:
    print(i, got["init_bin"][i], ref["init_bin"][i], got["attempts"][i], ref["attempts"][i], got["init_val"][i], ref["init_val"][i])
print("2-attempt trajs:", np.nonzero(ref["attempts"] > 1)[0])
g2 = native.sample_dbn_host(ctx, nm, n, T, seed, first_index=first, want_dense=True, want_events=True, **idx)
print(g2["kernel"], "generic bins eq", np.array_equal(g2["dyn_bin"], ref["dense_bin"]), np.argwhere(g2["dyn_bin"] != ref["dense_bin"])[:5])
