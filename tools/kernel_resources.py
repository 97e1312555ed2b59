"""Compact per-kernel resource table (VGPR / SGPR / scratch / LDS / occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage.   usage: python tools/kernel_resources.py stencil.hip [filter]"""
import os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pde_opt_amd", "csrc")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/opt/rocm/include", "-fno-slp-vectorize", "-c",
       os.path.join(CSRC, src), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
print(f"{'vgpr':>5} {'sgpr':>5} {'scr':>5} {'lds':>6} {'occ':>4}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"pdeopt::|\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if flt and flt not in n:
        continue
    print(f"{r.get('vgpr',0):>5} {r.get('sgpr',0):>5} {r.get('scratch',0):>5} {r.get('lds',0):>6} {r.get('occ',0):>4}  {n}")
