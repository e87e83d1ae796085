# lock-step against streaming kernel on small OPEN scenes (random spheres under a sky), one box
cd $GRAFT_REPO_ROOT
export PT_COUNT=0
for sc in dense12 dense16 dense24 dense32 dense48 dense64 dense100; do
for k in lds stream; do
  printf "%-9s %-6s " $sc $k; PT_KERNEL=$k timeout -k 5 200 python3 tools/prof_run.py $sc 64 3 | grep -o "kernel [0-9.]*" | tr '\n' ' '; echo
done
done
