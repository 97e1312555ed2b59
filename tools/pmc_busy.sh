#!/bin/bash
# one PMC pass: which pipe is busy?  usage: tools/pmc_busy.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_busy -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_busy.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $OUT/pmc_busy2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_busy2.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for d in ("pmc_busy", "pmc_busy2"):
    files = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)
    if not files:
        print(d, "no counter file"); print(open("$OUT/%s.log" % d).read()[-1500:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"][:72]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for row in csv.DictReader(open(files[0])):
        pass
    disp = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        disp[(row["Kernel_Name"][:72], row["Dispatch_Id"])] += 1
    for k in acc:
        nd = len([1 for (kk, _) in disp if kk == k])
        if any(t in k for t in ("pair", "stage", "quad", "col_reg", "row_")):
            print(k, "launches", nd, {c: "%.4g" % (v / nd) for c, v in acc[k].items()})
PY
