import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as O
from em_model_manned_bayes_amd import em_io, native
from util import uncor_indices
tmp = tempfile.mkdtemp()
path = em_io.materialize_model("uncor_1200code_v2p1", tmp)
nm = native.NativeModel.load_txt(path); pp = O.parse_model_txt(path)
n, T, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 240, 0xABCDEF12345
first = 2**33 + 17
idx = uncor_indices(pp)
ref = O.uncor_sample(O.OracleModel(pp), n, T, seed, first_index=first, want_events=False)
dev = torch.device("cuda", 0)
ctx = native.Context(0, stream=torch.cuda.current_stream().cuda_stream)
G4 = T // 4
outs = []
for rep in range(3):
    db = torch.full((G4, 3, n), -1, dtype=torch.int32, device=dev)
    dv = torch.full((G4, 3, n, 4), float("nan"), dtype=torch.float32, device=dev)
    ib = torch.zeros((7, n), dtype=torch.uint8, device=dev); iv = torch.zeros((7, n), dtype=torch.float32, device=dev)
    p, _ = native.make_params(n, T, seed, first_index=first, **idx)
    native.sample_dbn_device(ctx, nm, p, init_bin=ib.data_ptr(), init_val=iv.data_ptr(), dyn_bin=db.data_ptr(), dyn_val=dv.data_ptr())
    ctx.sync(); torch.cuda.synchronize()
    b = native.unpack_dyn_bin(db.cpu().numpy().view(np.uint32), T); v = native.unpack_dyn_val(dv.cpu().numpy(), T)
    outs.append((b, v))
    bb = np.argwhere(b != ref["dense_bin"]); vv = np.argwhere(v != ref["dense_val"].astype(np.float32))
    print("rep", rep, ctx.last_kernel(), "bin mism", len(bb), bb[:5].tolist(), "val mism", len(vv), vv[:5].tolist(), "unwritten words", int((db == -1).sum()), "nan vals", int(torch.isnan(dv).sum()))
    for x in bb[:5]:
        i, c, k = x
        print("   got", b[i, c, k], v[i, c, k], "ref", ref["dense_bin"][i, c, k], ref["dense_val"][i, c, k])
i = 3514
b, v = outs[0]
print("ref bins", ref["dense_bin"][i, :12, 1], "vals", ref["dense_val"][i, :12, 1])
print("gpu bins", b[i, :12, 1], "vals", v[i, :12, 1])
print("init", ref["init_bin"][i], ref["init_val"][i])
