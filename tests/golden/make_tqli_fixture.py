"""Generate tests/golden/tqli_golden.json by importing the reference's own
prototype ``python_tests/tqli.py`` (numpy + scipy only) in the build container.

Run once here (``python tests/golden/make_tqli_fixture.py``); the GPU box never
sees /root/reference, it only reads the committed JSON.  The fixture is data:
the (d, e) input the reference file defines at ``python_tests/tqli.py:63-90`` and
the eigenvalues its ``tqli`` routine (``:47-60``) returns for it, plus scipy's
``eigh_tridiagonal`` answer the reference asserts against (``:93-99``).
"""
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True  # the reference tree is read-only for us: no __pycache__ next to its file

import numpy as np
from scipy import linalg

REF = "/root/reference/python_tests/tqli.py"
spec = importlib.util.spec_from_file_location("ref_tqli", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

d_in = ref.d.copy()
e_in = ref.e.copy()
d = ref.d.copy()
e = ref.e.copy()
ref.tqli(d, e)
eigs_scipy = sorted(linalg.eigh_tridiagonal(d_in, e_in[:-1], eigvals_only=True))

# a second, seeded case through the same reference routine
rng = np.random.default_rng(7)
d2_in = rng.uniform(0.5, 2.0, 16)
e2_in = np.concatenate([rng.uniform(0.1, 0.7, 15), [0.0]])
d2, e2 = d2_in.copy(), e2_in.copy()
ref.tqli(d2, e2)

out = {
    "source": "python_tests/tqli.py @ Wells-Group/pmg-dolfinx 2024_08_07",
    "cases": [
        {"d": d_in.tolist(), "e": e_in.tolist(), "tqli_d_out": d.tolist(), "sorted_eigs_scipy": list(map(float, eigs_scipy))},
        {"d": d2_in.tolist(), "e": e2_in.tolist(), "tqli_d_out": d2.tolist(),
         "sorted_eigs_scipy": list(map(float, sorted(linalg.eigh_tridiagonal(d2_in, e2_in[:-1], eigvals_only=True))))},
    ],
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tqli_golden.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
print(sorted(d.tolist()))
