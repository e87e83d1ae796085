#!/bin/bash
# A/B timing on ONE box (devices differ by several percent, so only timings from the same call compare):
#   tools/ab_run.sh <spp> <tag> [<tag> ...]     each tag = build/ab/libpt_hip_<tag>.so (tools/ab_build.sh); a tag
#   ending in :fast is timed with PT_MATH=fast
cd $GRAFT_REPO_ROOT
spp=$1; shift
for rep in 1 2; do
for t in "$@"; do
  tag=${t%:fast}; math=strict; [ "$t" != "$tag" ] && math=fast
  printf "%-22s " "$t"
  PT_MATH=$math PT_COUNT=0 PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so timeout -k 5 120 python3 tools/prof_run.py frame0 $spp 2 | tail -1 || exit 1
done
done
