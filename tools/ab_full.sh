# A/B on one box, headline frame only: tools/ab_full.sh <tag> ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for tag in "$@"; do
  export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so PT_COUNT=0
  printf "%-8s " $tag; timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1 || exit 1
done
done
