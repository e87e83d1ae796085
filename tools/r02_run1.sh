set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02/t1.log 2>&1 || { tail -30 gpurun_out/r02/t1.log; exit 1; }
tail -3 gpurun_out/r02/t1.log
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r02/t_noalloc.log 2>&1 || { tail -30 gpurun_out/r02/t_noalloc.log; exit 1; }
tail -3 gpurun_out/r02/t_noalloc.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > gpurun_out/r02/b1.log 2>&1 || { tail -20 gpurun_out/r02/b1.log; exit 1; }
tail -1 gpurun_out/r02/b1.log
