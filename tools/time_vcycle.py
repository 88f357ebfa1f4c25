"""Time the config-2 V-cycle (64^3, p = 4 -> 2 -> 1, Chebyshev(3)) alone.
usage: python tools/time_vcycle.py [reps [n]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmg_dolfinx_amd as pm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = pm.PoissonHierarchy(n, (1, 2, 4), cheb_its=3)
x = H.new_vector()
x.set(0.0)
for _ in range(3):
    H.mg.apply(H.rhs[-1], x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    H.mg.apply(H.rhs[-1], x)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"n={n} policy={os.environ.get('PMG_STREAM_POLICY', 'auto')} lib={os.environ.get('PMG_AMD_LIB', 'default')} V-cycle {ms:.3f} ms  {H.fine_ndofs_owned / ms / 1e6:.3f} GDoF/s  |x| = {pm.norm(x):.12e}")
