#!/bin/bash
# A/B timing on ONE GPU box (boxes differ by several percent, so only timings from the same call compare).
#
#   tools/ab.sh [-r REPS] <workload> <variant> [<variant> ...]
#
#   workload  headline (1080p x 256 spp) | eighth (rank 3 of 8 of it) | lua (640x360 x 2000) | cfg1 (256^2 x 16) |
#             4k (3840x2160 x 256) | lattice (config 5, 64 spp, depth 32) | dense10k (16 spp) | dense100k (2 spp) |
#             "raw: <tools/prof_run.py arguments>"
#   variant   [VAR=VALUE,VAR=VALUE@]<tag>[:fast]
#             tag = "tree" (the in-tree libpt_hip.so) or a library built by tools/ab_build.sh <tag> [-D...];
#             :fast times the relaxed build; VAR=VALUE pairs are PT_* switches for that variant only, e.g.
#             PT_LANES_PER_PIXEL=8,PT_TAIL_PCT=0@tree     PT_KERNEL=stream@tree     nosplit:fast
#
# This one script replaces round 2's ab_full/ab_split/ab_tail*/ab_lanes/ab_lattice*/ab_dense/ab_cross/ab_open.sh:
#   lanes sweep    tools/ab.sh headline PT_LANES_PER_PIXEL={1,2,4,8}@tree
#   tail sweep     tools/ab.sh headline PT_TAIL_PCT={0,3,8}@tree
#   kernel choice  for s in lattice "raw: lattice5 32 3 32"; do tools/ab.sh "$s" PT_KERNEL={lds,stream}@tree; done
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
reps=2
[ "$1" = "-r" ] && { reps=$2; shift 2; }
wl=$1; shift
case "$wl" in
  headline) args="frame0 256 2" ;;
  eighth)   args="frame0 256 3 50 1920 1080 8 3 8" ;;
  lua)      args="frame0 2000 2 50 640 360" ;;
  cfg1)     args="frame0 16 4 50 256 256" ;;
  4k)       args="frame0 256 2 50 3840 2160" ;;
  lattice)  args="lattice 64 3 32" ;;
  dense10k) args="dense10k 16 4" ;;
  dense100k) args="dense100k 2 2" ;;
  raw:*)    args="${wl#raw:}" ;;
  *) echo "unknown workload $wl"; exit 2 ;;
esac
for rep in $(seq "$reps"); do
for v in "$@"; do
  envs=""; t=$v
  case "$v" in *@*) envs=${v%@*}; t=${v##*@} ;; esac
  tag=${t%:fast}; math=strict; [ "$t" != "$tag" ] && math=fast
  lib=$PWD/pathtracer_amd/libpt_hip.so; [ "$tag" != tree ] && lib=$PWD/build/ab/libpt_hip_$tag.so
  printf "%-10s %-36s " "$wl" "$v"
  # shellcheck disable=SC2086
  env PT_COUNT=0 PT_MATH=$math PT_HIP_LIB=$lib $(echo "$envs" | tr ',' ' ') timeout -k 5 280 python3 tools/prof_run.py $args | grep -o "kernel [0-9.]* ms" | tr '\n' ' ' || exit 1
  echo
done
done
