set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
PT_COUNT=0 bash tools/pmc_passes.sh r02_lds_strict frame0 32 1 > gpurun_out/r02/pmc_strict.log 2>&1 || { tail -5 gpurun_out/r02/pmc_strict.log; exit 1; }
PT_MATH=fast PT_COUNT=0 bash tools/pmc_passes.sh r02_lds_fast frame0 32 1 > gpurun_out/r02/pmc_fast.log 2>&1 || { tail -5 gpurun_out/r02/pmc_fast.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/prof_b -o b --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02/prof_b.log 2>&1 || { tail -20 $R/gpurun_out/r02/prof_b.log; exit 1; }
tail -1 $R/gpurun_out/r02/prof_b.log | cut -c1-300
