"""Per-launch PMC averages of the dominant kernel -> profiles/pmc_<tag>.json (read by bench.py's roofline block).

    python tools/pmc_to_json.py <workload> <tag> <kernel-substring> <dir> [<dir> ...]

Every <dir> is the output directory of one `rocprofv3 --pmc ...` pass (tools/pmc_busy.sh, tools/profile_gpu.sh:
separate passes per counter group, as MI355X_MICROARCH.md prescribes).  Counters are averaged over the launches
whose kernel name contains the substring (several alternatives separated by "|": the kernels of a multi-kernel
substep; `per_kernel` then keeps each one's own averages).  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE
reports half the bytes of wide streaming loads on gfx950, both are KiB of L2 fabric-side requests (Infinity-Cache
hits included)."""
import collections, csv, glob, json, os, sys

workload, tag, sub = sys.argv[1:4]
subs = sub.split("|")
acc, cnt, names = collections.defaultdict(float), collections.Counter(), set()
kacc, kcnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter)
for d in sys.argv[4:]:
    # gpurun_out/ accumulates runs: only the NEWEST pass of every directory counts
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if any(x in r["Kernel_Name"] for x in subs):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
                short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pdeopt::", "").replace("void ", "").split("(")[0]
                names.add(short)
                kacc[short][r["Counter_Name"]] += float(r["Counter_Value"])
                kcnt[short][r["Counter_Name"]] += 1
out = {c: acc[c] / cnt[c] for c in sorted(acc)}
rec = {"kernels": sorted(names), "launches_averaged": max(cnt.values()) if cnt else 0, "counters_per_launch": out,
       "source": "rocprofv3 --pmc passes of `python bench.py --steps 1 --warmup 0` (" + ", ".join(os.path.basename(d.rstrip('/')) for d in sys.argv[4:]) + ")"}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    rec["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
if len(names) > 1:
    rec["per_kernel"] = {}
    for k in sorted(kacc):
        pk = {c: kacc[k][c] / kcnt[k][c] for c in sorted(kacc[k])}
        if "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
            pk["hbm_bytes_per_launch"] = (2 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024
        pk["launches"] = max(kcnt[k].values())
        rec["per_kernel"][k] = pk
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", f"pmc_{tag}.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
allrec[workload] = rec
json.dump(allrec, open(path, "w"), indent=1)
print(json.dumps(rec, indent=1))
