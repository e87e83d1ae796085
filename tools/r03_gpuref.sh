# Round 3, first GPU call: freeze the reference kernel's own output on MI355X, measure the HIP builds against it, time it.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gpuref
timeout -k 10 400 python3 -u tools/make_gpuref_golden.py > gpurun_out/gpuref/golden.log 2>&1 || { tail -30 gpurun_out/gpuref/golden.log; exit 1; }
tail -40 gpurun_out/gpuref/golden.log
timeout -k 10 120 python3 -u tools/ref_gfx950_time.py frame0 256 256 16 50 3 compare > gpurun_out/gpuref/time_cfg1.log 2>&1 || { tail -30 gpurun_out/gpuref/time_cfg1.log; exit 1; }
cat gpurun_out/gpuref/time_cfg1.log
timeout -k 10 300 python3 -u tools/ref_gfx950_time.py frame0 1920 1080 256 50 2 compare > gpurun_out/gpuref/time_cfg2.log 2>&1 || { tail -30 gpurun_out/gpuref/time_cfg2.log; exit 1; }
cat gpurun_out/gpuref/time_cfg2.log
timeout -k 10 120 python3 -u tools/ref_gfx950_time.py frame0 1920 1080 256 50 2 nocompare nosync > gpurun_out/gpuref/time_cfg2_nosync.log 2>&1 || { tail -30 gpurun_out/gpuref/time_cfg2_nosync.log; exit 1; }
cat gpurun_out/gpuref/time_cfg2_nosync.log
