"""Time the patched transfers of the config-2 hierarchy (64^3, p = 1, 2, 4) alone.
usage: python tools/time_transfers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmg_dolfinx_amd as pm

H = pm.PoissonHierarchy(64, (1, 2, 4), cheb_its=3)


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for i, ip in enumerate(H.interpolators):
    uc, uf = H.new_vector(i), H.new_vector(i + 1)
    uc.data.normal_()
    uf.data.normal_()
    pc, pf = H.orders[i], H.orders[i + 1]
    print(f"lib={os.environ.get('PMG_AMD_LIB', 'default')} p{pc}->p{pf}: prolong+add {timed(lambda: ip.interpolate_add(uc, uf)):7.1f} us   "
          f"restrict {timed(lambda: ip.reverse_interpolate(uf, uc)):7.1f} us")
