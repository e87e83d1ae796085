cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gpuref
timeout -k 10 200 python3 -u tools/make_gpuref_golden.py dense12_48x27 dense13_48x27 > gpurun_out/gpuref/golden2.log 2>&1 || { tail -20 gpurun_out/gpuref/golden2.log; exit 1; }
cat gpurun_out/gpuref/golden2.log
timeout -k 10 300 python3 -u tools/ref_gfx950_time.py dense10k 480 270 16 50 1 compare > gpurun_out/gpuref/time_dense10k.log 2>&1 || { tail -20 gpurun_out/gpuref/time_dense10k.log; exit 1; }
cat gpurun_out/gpuref/time_dense10k.log
timeout -k 10 300 python3 -u tools/ref_gfx950_time.py dense10k 1920 1080 4 50 1 nocompare > gpurun_out/gpuref/time_dense10k_1080p.log 2>&1 || { tail -20 gpurun_out/gpuref/time_dense10k_1080p.log; exit 1; }
cat gpurun_out/gpuref/time_dense10k_1080p.log
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; rc=$?
tail -8 gpurun_out/gputests.log
exit $rc
