cd $GRAFT_REPO_ROOT
export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_split.so PT_COUNT=0
for rep in 1 2; do
printf "full default:            "; timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1 || exit 1
for cfg in "4 4 4" "4 4 12" "4 5 8" "8 4 8" "8 4 20" "8 5 8" "8 5 15" "8 3 0"; do set -- $cfg
  printf "full S1=%d tail 2^%d %2d%%: " $1 $2 $3; PT_LANES_PER_PIXEL=$1 PT_TAIL_LANES=$2 PT_TAIL_PCT=$3 timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1 || exit 1
done
printf "eighth default:            "; timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1920 1080 8 3 8 | tail -1 || exit 1
for cfg in "16 5 15" "16 5 25" "32 5 0" "8 5 20" "8 4 50"; do set -- $cfg
  printf "eighth S1=%d tail 2^%d %2d%%: " $1 $2 $3; PT_LANES_PER_PIXEL=$1 PT_TAIL_LANES=$2 PT_TAIL_PCT=$3 timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1920 1080 8 3 8 | tail -1 || exit 1
done
done
