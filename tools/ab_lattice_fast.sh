cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for tag in "$@"; do
  export PT_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/libpt_hip_$tag.so PT_COUNT=0
  printf "%-8s strict " $tag; PT_MATH=strict timeout -k 5 120 python3 tools/prof_run.py lattice 64 3 32 | grep -o "kernel [0-9.]*" | tr '\n' ' '
  printf " | fast "; PT_MATH=fast timeout -k 5 120 python3 tools/prof_run.py lattice 64 3 32 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
done
done
