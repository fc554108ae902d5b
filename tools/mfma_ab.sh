#!/bin/bash
# A/B of the matrix-pipe variants (DFLO_MFMA=0 | 1) on one box with the counters that say why: kernel time (rocprofv3 --kernel-trace
# --stats of a short run), then two PMC passes per side.  usage: tools/mfma_ab.sh <tag> <bench args...>  ->  gpurun_out/r6/<tag>_mfma_ab.txt
set -u
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-secondary --no-live-traffic $*"
{
echo "# $TAG: python bench.py $*   (DFLO_MFMA=0: vector units, DFLO_MFMA=1: matrix instructions; same box, back to back)"
for mf in 0 1 0 1; do printf "DFLO_MFMA=%d  " $mf; DFLO_MFMA=$mf tools/bench_line.sh --no-secondary "$@"; done
for mf in 0 1; do
  D=/tmp/mfab_${TAG}_$mf; rm -rf $D
  ( cd /tmp && DFLO_MFMA=$mf rocprofv3 --kernel-trace --stats -d $D/trace -o t -f csv -- python $ROOT/bench.py --steps 40 --warmup 10 $ARGS ) > /dev/null 2>&1
  ( cd /tmp && DFLO_MFMA=$mf rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-include-regex stage_kernel -d $D/p1 -o p -f csv -- python $ROOT/bench.py --steps 6 --warmup 2 $ARGS ) > /dev/null 2>&1
  ( cd /tmp && DFLO_MFMA=$mf rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-include-regex stage_kernel -d $D/p2 -o p -f csv -- python $ROOT/bench.py --steps 6 --warmup 2 $ARGS ) > /dev/null 2>&1
  echo "---- DFLO_MFMA=$mf"
  python - $D <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
short = lambda n: n.split("(")[0].replace("void dflo::", "")[:64]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stage_kernel" in r["Name"]:
            print("  %-66s calls %5s  avg %9.1f us" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    print("  " + k)
    print("     " + "  ".join("%s %.4g" % (n, c[n]) for n in sorted(c)))
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0   # summed over the 8 XCDs
    if g:
        mf, va = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / g, 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / g
        print("     kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs = %.0f;  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.4f;  valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles) = %.4f;  "
              "MFMA + VALU = %.4f;  co-execution cycles / MFMA busy cycles = %.4f" % (g, mf, va, mf + va, c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1.0)))
PY
done
} 2>&1 | tee $OUT/${TAG}_mfma_ab.txt
