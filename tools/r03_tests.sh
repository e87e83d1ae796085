cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1; rc=$?
tail -15 gpurun_out/gputests.log
exit $rc
