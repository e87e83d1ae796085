cd $GRAFT_REPO_ROOT
export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_t64.so PT_COUNT=0
for rep in 1 2; do
for cfg in "5 8" "6 8" "6 4" "6 15"; do set -- $cfg
  printf "640x360x2000 tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py frame0 2000 2 50 640 360 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
  printf "eighth       tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1920 1080 8 3 8 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
done
done
