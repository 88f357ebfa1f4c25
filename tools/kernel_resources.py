"""Register / LDS / occupancy table of the stiffness kernels as compiled for gfx950
(-Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
       "-fno-gpu-rdc", *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
       os.path.join(ROOT, "pmg-dolfinx_amd", "csrc", "laplacian.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.strip()] = v.strip()
print("| kernel | VGPRs | SGPRs | spills V/S | LDS bytes | waves/SIMD (registers) |")
print("|---|---|---|---|---|---|")
for name, r in sorted(rows.items()):
    m = re.search(r"stiffness_column_kernelILi(\d)ELb(\d)", name)
    if not m:
        continue
    mode = "affine" if m.group(2) == "1" else "stored"
    print(f"| stiffness_column_kernel<{m.group(1)}, {mode}> | {r.get('VGPRs')} | {r.get('SGPRs')} | "
          f"{r.get('VGPRs Spill')}/{r.get('SGPRs Spill')} | {r.get('LDS Size [bytes/block]')} | "
          f"{r.get('Occupancy [waves/SIMD]')} |")
