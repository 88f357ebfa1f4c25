"""Per-kernel achieved HBM GB/s against the MI355X roofline:
operator apply for p in {1..8} at ~17 M dofs (BASELINE config 4), and on the
config-2 hierarchy (64^3, p = 4 -> 2 -> 1) the BLAS-1 ops, the Chebyshev smoother,
the transfers and one V-cycle.  Timing: torch.cuda events on the launch stream,
R repetitions.  Writes gpurun_out/kernel_roofline_<tag>.md (copy it into profiles/).
usage (on the GPU box): python tools/kernel_roofline.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pmg_dolfinx_amd as pm

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
PEAK = 8000.0
rows = []


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def add(name, nbytes, t, note=""):
    gbs = nbytes / t / 1e9
    rows.append((name, nbytes / 1e6, t * 1e6, gbs, gbs / PEAK, note))
    print(f"{name:58s} {nbytes/1e6:9.1f} MB {t*1e6:9.1f} us {gbs:8.0f} GB/s {gbs/PEAK:6.3f}", flush=True)


def apply_bytes(P, ncells):
    N, U = (P + 1) ** 3, P**3
    return (52 * N + 8 + 17 * U) * ncells


# ---- config 4: apply sweep
for P, n in ((1, 256), (2, 128), (3, 85), (4, 64), (5, 51), (6, 43), (7, 36), (8, 32)):
    part = pm.BoxPartition(n)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
    x, y = pm.Vector(layout), pm.Vector(layout)
    x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
    t = timed(lambda: op(x, y), reps=10)
    add(f"operator apply p={P}, {n}^3 cells, {lv.ndofs/1e6:.1f} M dofs", apply_bytes(P, part.ncells), t,
        "model storedG, all colour launches of one application")
    del op, x, y, layout, part, lv
    torch.cuda.empty_cache()

# ---- config 2 hierarchy
H = pm.PoissonHierarchy(64, (1, 2, 4), kappa=2.0, cheb_its=3)
for lvl, P in ((2, 4), (1, 2), (0, 1)):
    L, op, sm = H.layouts[lvl], H.operators[lvl], H.smoothers[lvl]
    n = H.levels[lvl].size_local
    a, b, c = pm.Vector(L), pm.Vector(L), pm.Vector(L)
    a.data.copy_(torch.randn(n, dtype=torch.float64, device="cuda"))
    b.data.copy_(torch.randn(n, dtype=torch.float64, device="cuda"))
    add(f"axpy r = a x + y            p={P} ({n/1e6:.2f} M dofs)", 24 * n, timed(lambda: pm.axpy(c, 0.5, a, b)), "2 reads + 1 write")
    add(f"pointwise_mult              p={P}", 24 * n, timed(lambda: pm.pointwise_mult(c, a, b)), "2 reads + 1 write")
    add(f"inner_product (host value)  p={P}", 16 * n, timed(lambda: pm.inner_product(a, b)), "2 reads; includes the D2H sync")
    ncells = H.part.ncells
    k = 3
    t = timed(lambda: sm.solve(op, a, b), reps=10)
    nb = (k) * apply_bytes(P, ncells) + n * 8 * (5 + 8 * (k - 1) + 3)
    add(f"Chebyshev({k}) solve          p={P}", nb, t, f"{k} applies + fused passes (init 5, {k-1} steps x 8, final add 3)")
for i, (pc, pf) in enumerate(((1, 2), (2, 4))):
    ip = H.interpolators[i]
    Lc, Lf = H.layouts[i], H.layouts[i + 1]
    nc, nf = H.levels[i].size_local, H.levels[i + 1].size_local
    uc, uf = pm.Vector(Lc), pm.Vector(Lf)
    uc.data.copy_(torch.randn(nc, dtype=torch.float64, device="cuda"))
    uf.data.copy_(torch.randn(nf, dtype=torch.float64, device="cuda"))
    ncells = H.part.ncells
    Nc, Nf = (pc + 1) ** 3, (pf + 1) ** 3
    nb = ncells * 4 * (Nc + Nf) + 8 * (nc + 2 * nf)
    add(f"prolong + correct  p{pc}->p{pf}", nb, timed(lambda: ip.interpolate_add(uc, uf)), "both dofmaps + coarse read + fine read/write")
    nb = ncells * 4 * (Nc + Nf) + 8 * (nc + 2 * nf)
    add(f"restrict           p{pf}->p{pc}", nb, timed(lambda: ip.reverse_interpolate(uf, uc)), "both dofmaps + fine read + multiplicity + coarse write")
xv = H.new_vector()
xv.set(0.0)
t = timed(lambda: H.mg.apply(H.rhs[-1], xv), reps=10)
counts = H.mg.apply_counts()
nb = sum(c * apply_bytes(P, H.part.ncells) for c, P in zip(counts, (1, 2, 4)))
add("V-cycle p=4->2->1 (applies only counted)", nb, t, f"applies per level (coarse->fine) {counts}; vector/transfer bytes not counted")

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", f"kernel_roofline_{tag}.md")  # copy into profiles/ afterwards
with open(out, "w") as f:
    f.write(f"# Achieved HBM GB/s per kernel, round tag {tag} (MI355X, peak 8000 GB/s, copy ceiling ~6300)\n\n")
    f.write("Bytes are ALGORITHMIC (SURVEY.md 8d): compulsory traffic of the reference-faithful data structures, "
            "8 B per dof per vector pass.  Time: device events around R repetitions (`tools/kernel_roofline.py`).\n\n")
    f.write("| kernel | algorithmic MB | us | GB/s | frac of 8 TB/s | note |\n|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r[0]} | {r[1]:.1f} | {r[2]:.1f} | {r[3]:.0f} | {r[4]:.3f} | {r[5]} |\n")
print("wrote", out)
