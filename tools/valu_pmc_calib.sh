#!/bin/bash
# What does SQ_ACTIVE_INST_VALU read for a kernel that issues NOTHING but independent v_fma_f32 (tools/valubench.bin)?
# Calibrates "VALU busy" = 4 x SQ_ACTIVE_INST_VALU / SIMDs over GRBM_GUI_ACTIVE / 8 against a known-saturated pipe.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc -- $ROOT/tools/valubench.bin > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
files = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter file"); print(open("$OUT/run.log").read()[-1500:]); raise SystemExit
rows = collections.OrderedDict()
for r in csv.DictReader(open(files[0])):
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:40], r["Grid_Size"])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("dispatch kernel grid | insts/SIMD  busy-quadcycles x4/SIMD  cycles(GRBM/8) | busy share | cycles per inst | busy cycles per inst")
for (d, k, g), c in rows.items():
    if "SQ_INSTS_VALU" not in c: continue
    per = c["SQ_INSTS_VALU"] / 1024.0
    busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    if per < 1000: continue
    print(d, k, g, "| %.0f %.0f %.0f | %.3f | %.2f | %.2f" % (per, busy, cyc, busy / cyc, cyc / per, busy / per),
          "| wave-cycles: wait_any %.2f wait_inst %.2f active %.2f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
