"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.  usage: python tools/kres.py <log> [substring]"""
import re, subprocess, sys
log, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cur, rows = None, []
for line in open(log, errors="replace"):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = name.replace("pdeopt::", "").split("(")[0].replace("void ", "")
    if sub in name:
        print(f"{name:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} B  sgpr-spill {r.get('SGPRs Spill','?'):>4s} vgpr-spill {r.get('VGPRs Spill','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
