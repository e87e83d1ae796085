# A/B on one box, streaming kernel: tools/ab_dense.sh <tag> ...   (10^4 spheres, 16 spp, 8 launches per process, all
# printed: the persistent kernel's time varies by +-2.5 % from launch to launch; tags ending in :fast use PT_MATH=fast)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "$@"; do
  tag=${t%:fast}; math=strict; [ "$t" != "$tag" ] && math=fast
  export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so PT_COUNT=0 PT_MATH=$math
  printf "%-12s " $t; timeout -k 5 200 python3 tools/prof_run.py dense10k 16 8 | grep -o "kernel [0-9.]*" | tr '\n' ' ' || exit 1; echo
done
done
