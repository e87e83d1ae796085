cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03final
PT_REFFUZZ_CASES=200000 PT_REFFUZZ_SEED=100000 PT_REFFUZZ_OUT=gpurun_out/r03final/gpuref_fuzz_200k.json timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_reference_fuzz.py -m gpu -x -q -s --timeout 980 --timeout-method thread > gpurun_out/r03final/fuzz200k.log 2>&1; rc=$?
tail -4 gpurun_out/r03final/fuzz200k.log
exit $rc
