"""Which OpenMP thread count gives the C oracle's V-cycle its best time on this host?  (The GPU box is a
container: nproc says 128, the CPU quota may be lower.)  Prints the cgroup limits and seconds per cycle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pmg_dolfinx_amd as pm  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(path, open(path).read().strip())
    except Exception as e:
        print(path, "-", type(e).__name__)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "cpu_share()", co.cpu_share())
part = pm.BoxPartition(n)
orders = (1, 2, 4)
cl = [co.CLevel(p, 2.0, part.level(p).dofmap, part.xgeom, part.geom_dofmap, part.level(p).bc_marker) for p in orders]
ci = [co.CInterp(cl[i], cl[i + 1]) for i in range(2)]
cm = co.CMultigrid(cl, ci, [2.13, 2.36, 2.52], 3)
b = np.random.default_rng(1).standard_normal(cl[-1].ndofs)
x = np.zeros_like(b)
for t in [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64, 128]:
    co.set_num_threads(t)
    cm.apply(b, x)
    t0 = time.perf_counter()
    for _ in range(2):
        cm.apply(b, x)
    dt = (time.perf_counter() - t0) / 2
    xa = np.random.default_rng(0).standard_normal(cl[-1].ndofs)
    t0 = time.perf_counter()
    cl[-1].apply(xa)
    da = time.perf_counter() - t0
    print(f"threads {t:4d}: {dt:.3f} s per V-cycle = {cl[-1].ndofs / dt / 1e6:.1f} MDoF/s = {32.78 * (n / 64) ** 3 / dt:.1f} GB/s "
          f"algorithmic; p4 apply {da * 1e3:.0f} ms", flush=True)
