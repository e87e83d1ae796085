# A/B on one box: tools/ab_lattice.sh <tag> ...   (config 5 lattice 64 spp depth 32; headline frame)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in "$@"; do
  export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so PT_COUNT=0
  printf "%-8s " $tag; timeout -k 5 120 python3 tools/prof_run.py lattice 64 3 32 | grep -o "kernel [0-9.]*" | tr '\n' ' ' || exit 1
  printf " | headline "; timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 | grep -o "kernel [0-9.]*" | tr '\n' ' ' || exit 1; echo
done
done
