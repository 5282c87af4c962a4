"""tools/pack_models.py -- pack the reference's BSD-2 model DATA files into models/*.npz.

Run in the build container only (reads /root/reference/model).  The packed files are data
(CPT counts, cut points, resample rates) -- see models/NOTICE.  They let the GPU box, which has
no /root/reference, benchmark and test on the real tables.
"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from em_model_manned_bayes_amd import em_io  # noqa: E402

REF = "/root/reference/model"
SKIP = set()


def main():
    files = sorted(glob.glob(os.path.join(REF, "*.txt"))) + sorted(glob.glob(os.path.join(REF, "correlated_terminal", "*", "*.txt")))
    os.makedirs(os.path.join(ROOT, "models"), exist_ok=True)
    for f in files:
        name = os.path.splitext(os.path.basename(f))[0]
        if name in SKIP:
            continue
        p = em_io.em_read(f)
        out = os.path.join(ROOT, "models", name + ".npz")
        em_io.save_npz(p, out)
        print("%-50s %8d B" % (name, os.path.getsize(out)))


if __name__ == "__main__":
    main()
