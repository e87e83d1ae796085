cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "4 8" "5 8" "5 15" "5 30" "5 50"; do set -- $cfg; printf "eighth: tail lanes 2^%d, %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1920 1080 8 3 8 | tail -1; done
for cfg in "4 8" "5 8" "5 4"; do set -- $cfg; printf "full: tail lanes 2^%d, %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 PT_COUNT=0 timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1; done
done
