set -e
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02/t2.log 2>&1 || { tail -30 gpurun_out/r02/t2.log; exit 1; }
tail -3 gpurun_out/r02/t2.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/prof_a -o a --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02/prof_a.log 2>&1 || { tail -20 $R/gpurun_out/r02/prof_a.log; exit 1; }
tail -1 $R/gpurun_out/r02/prof_a.log
cd $R
PT_MATH=fast PT_COUNT=0 bash tools/pmc_passes.sh r02_fastlds frame0 32 1
PT_PMC_EXTRA=1 PT_COUNT=0 bash tools/pmc_passes.sh r02_stream10k dense10k 8 1
PT_PMC_EXTRA=1 PT_COUNT=0 bash tools/pmc_passes.sh r02_stream100k dense100k 2 1
