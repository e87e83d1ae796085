cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; rc=$?
tail -25 gpurun_out/gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python3 -u tools/meandiv_walk.py > gpurun_out/meandiv_walk.log 2>&1; rc=$?
tail -4 gpurun_out/meandiv_walk.log
exit $rc
