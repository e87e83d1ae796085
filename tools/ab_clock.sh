#!/bin/bash
# shader clock under a variant: GRBM_GUI_ACTIVE / 8 / kernel time, from one counter pass per tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
spp=$1; shift
for tag in "$@"; do
  rm -rf /tmp/clk_$tag
  PT_COUNT=0 PT_HIP_LIB=$R/build/ab/libpt_hip_$tag.so timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/clk_$tag -o c --output-format csv -- python3 $R/tools/prof_run.py frame0 $spp 2 > /tmp/clk_$tag.log 2>&1 || { tail -5 /tmp/clk_$tag.log; exit 1; }
  ms=$(grep kernel /tmp/clk_$tag.log | tail -1 | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  python3 - "$tag" "$ms" $(find /tmp/clk_$tag -name "*counter_collection.csv") <<'PY'
import csv, sys
tag, ms, f = sys.argv[1], float(sys.argv[2]), sys.argv[3]
v = {}
for r in csv.DictReader(open(f)):
    if "pt_render" in r["Kernel_Name"]:
        v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
g = v["GRBM_GUI_ACTIVE"][-1] / 8
print("%-14s kernel %.2f ms  clock %.3f GHz  VALU insts %.4g  cycles/inst/SIMD %.3f" % (tag, ms, g / ms / 1e6, v["SQ_INSTS_VALU"][-1], g * 1024 / v["SQ_INSTS_VALU"][-1]))
PY
done
