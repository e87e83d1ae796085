# Round 3: the reference kernel timed on MI355X with the reference's host pattern, + rocprofv3 stats and PMC passes over it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gpuref
run() { local log=$1; shift; timeout -k 10 300 python3 -u tools/ref_gfx950_time.py "$@" > gpurun_out/gpuref/$log 2>&1 || { tail -30 gpurun_out/gpuref/$log; exit 1; }; cat gpurun_out/gpuref/$log; }
run time_cfg1.log frame0 256 256 16 50 3 compare
run time_cfg2.log frame0 1920 1080 256 50 2 compare
run time_cfg2_nosync.log frame0 1920 1080 256 50 2 nocompare nosync
run time_cfg5.log lattice 1920 1080 64 32 2 compare
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gpuref/prof -o ref --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ref_gfx950_time.py frame0 1920 1080 256 50 1 nocompare > $GRAFT_REPO_ROOT/gpurun_out/gpuref/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/gpuref/prof.log; exit 1; }
tail -3 $GRAFT_REPO_ROOT/gpurun_out/gpuref/prof.log
cd $GRAFT_REPO_ROOT
PT_PMC_PROG=tools/ref_gfx950_time.py bash tools/pmc_passes.sh ref_gfx950 frame0 1920 1080 32 50 1 nocompare || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_ref_gfx950 pt_main > gpurun_out/gpuref/pmc_ref.md
cat gpurun_out/gpuref/pmc_ref.md
find gpurun_out/gpuref/prof -name "*stats*" | head
