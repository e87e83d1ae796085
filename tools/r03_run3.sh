cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 -u tools/meandiv_walk.py > gpurun_out/meandiv_walk.log 2>&1; rc=$?
tail -4 gpurun_out/meandiv_walk.log
[ $rc -eq 0 ] || exit $rc
bash tools/pmc_passes.sh r03_lattice lattice 32 1 32 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_r03_lattice pt_render > gpurun_out/pmc_r03_lattice.md
cat gpurun_out/pmc_r03_lattice.md
PT_MATH=fast bash tools/pmc_passes.sh r03_lattice_fast lattice 32 1 32 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_r03_lattice_fast pt_render > gpurun_out/pmc_r03_lattice_fast.md
cat gpurun_out/pmc_r03_lattice_fast.md
