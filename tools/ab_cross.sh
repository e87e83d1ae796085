# lock-step against streaming kernel on closed scenes of growing table size (the choice in pt_render), one box
cd $GRAFT_REPO_ROOT
export PT_COUNT=0
for sc in lattice lattice4 lattice5 lattice6 lattice7; do
for k in lds stream; do
  printf "%-9s %-6s " $sc $k; PT_KERNEL=$k timeout -k 5 200 python3 tools/prof_run.py $sc 32 3 32 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
done
done
