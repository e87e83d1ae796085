set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02/t3.log 2>&1 || { tail -40 gpurun_out/r02/t3.log; exit 1; }
tail -3 gpurun_out/r02/t3.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r02/b3.log 2>&1 || { tail -20 gpurun_out/r02/b3.log; exit 1; }
tail -1 gpurun_out/r02/b3.log
