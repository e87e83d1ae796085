cd $GRAFT_REPO_ROOT
for rep in 1 2; do for t in 0 4 8 12 20; do printf "tail %2d%%: " $t; PT_TAIL_PCT=$t PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1; done; done
PT_TAIL_PCT=0 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 1024 1 50 3840 2160 | tail -1
PT_TAIL_PCT=8 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 1024 1 50 3840 2160 | tail -1
PT_TAIL_PCT=0 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py lattice 64 2 32 | tail -1
PT_TAIL_PCT=8 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py lattice 64 2 32 | tail -1
