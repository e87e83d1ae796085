# second-tier lanes (log2) and share on the lattice (64 spp), the 4K frame (256 spp here) and config 1, same library, one box
cd $GRAFT_REPO_ROOT
export PT_COUNT=0
for rep in 1 2; do
for cfg in "4 8" "6 3"; do set -- $cfg
  printf "lattice tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py lattice 64 3 32 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
  printf "4K x256  tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py frame0 256 2 50 3840 2160 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
  printf "1280x720 tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py frame0 256 3 50 1280 720 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
  printf "256x256x16 tail 2^%d %2d%%: " $1 $2; PT_TAIL_LANES=$1 PT_TAIL_PCT=$2 timeout -k 5 120 python3 tools/prof_run.py frame0 16 4 50 256 256 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
done
done
