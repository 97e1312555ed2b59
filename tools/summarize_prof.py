"""Summarise rocprofv3 output directories produced by tools/profile_gpu.sh into a small text
report (per-kernel avg duration, PMC sums per launch).  usage: summarize_prof.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]

def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))

def short(name):
    name = re.sub(r"pdeopt::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:110]

print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f"{'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for r in rows[:12]:
        print(f"{int(r['Calls']):>7} {float(r['AverageNs'])/1e3:>10.2f} {float(r['MinNs'])/1e3:>10.2f} "
              f"{float(r['MaxNs'])/1e3:>10.2f} {float(r['Percentage']):>6.2f}  {short(r['Name'])}")
for f in find("trace", "*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    if rows:
        r = rows[len(rows) // 2]
        keys = [k for k in r if any(s in k for s in ("VGPR", "SGPR", "LDS", "Scratch", "Workgroup", "Grid"))]
        print("sample dispatch:", short(r.get("Kernel_Name", "")), {k: r[k] for k in keys})

for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    print(f"== {sub}: per-launch averages ==")
    for k in sorted(acc, key=lambda kk: -sum(cnt[kk].values()))[:8]:
        parts = ", ".join(f"{c}={acc[k][c]/cnt[k][c]:.4g}" for c in sorted(acc[k]))
        n = max(cnt[k].values())
        print(f"  [{n} launches] {k}\n      {parts}")
