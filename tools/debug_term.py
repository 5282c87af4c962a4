import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
import em_model_manned_bayes_amd as E
from em_model_manned_bayes_amd import synthetic, native
d = synthetic.write_terminal_directory(tempfile.mkdtemp())
ctx = native.Context(0)
t = E.CorTerminalModel(srcData="terminalradar", parameters_directory=d)
t.acType1, t.acType2 = sys.argv[1], sys.argv[2]
n, seed = 600, 0x5EED0005
_, samples = t.sample(n, seed=seed, ctx=ctx)
geo, mo = t._geo_rows(samples); dl = t._dyn_rows()
oms = []
for m in t._traj:
    pp = O.parse_model_txt(m.parameters_filename)
    oms.append(O.OracleModel(pp, alpha_transition=O.stay_prior_alpha(pp, 1.0)))
ref, rr = O.propagate(oms, mo, geo, seed, dl)
got, rows = native.propagate_terminal_host(ctx, [m.native for m in t._traj], geo, mo, seed, dyn_limits=dl)
bad = np.nonzero(rows != rr)[0]
print("lanes with different row counts", len(bad), bad[:10], rows[bad[:10]], rr[bad[:10]])
for L_ in bad[:3]:
    r = min(rows[L_], rr[L_])
    dlt = np.abs(got[L_, :r] - ref[L_, :r]) / (np.abs(ref[L_, :r]) + 1e-6)
    first = np.argwhere(dlt > 1e-5)
    print("lane", L_, "role", L_ & 3, "first diff at", first[:3].tolist())
    if len(first):
        s = first[0][0]
        print(" gpu", got[L_, max(s-2,0):s+2]); print(" ref", ref[L_, max(s-2,0):s+2])
