cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in 1 2 4 8; do printf "lanes %d: " $l; PT_LANES_PER_PIXEL=$l PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1; done; done
