# A/B on one box: tools/ab_split.sh <tag> <tag> ...   (headline frame, one rank's eighth, main.lua's 640x360x2000)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in "$@"; do
  export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so PT_COUNT=0
  printf "%-6s full:   " $tag; timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | tail -1 || exit 1
  printf "%-6s eighth: " $tag; timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1920 1080 8 3 8 | tail -1 || exit 1
  printf "%-6s 640x360x2000: " $tag; timeout -k 5 120 python3 tools/prof_run.py frame0 2000 2 50 640 360 | tail -1 || exit 1
done
done
