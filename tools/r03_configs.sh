# Kernel times of the five BASELINE configs with the current kernels, both builds (tools/prof_run.py: scene spp reps depth W H world rank band)
cd $GRAFT_REPO_ROOT
run() { echo "### $* [reference-exact build]"; PT_MATH=fast timeout -k 5 280 python3 tools/prof_run.py "$@" | tail -2 || exit 1
        echo "### $* [strict build]"; PT_MATH=strict timeout -k 5 280 python3 tools/prof_run.py "$@" | tail -1 || exit 1; }
run frame0 16 3 50 256 256
run frame0 256 2
run dense10k 64 2
run dense100k 16 1
PT_ACCEL=grid run dense10k 64 2
PT_ACCEL=grid run dense100k 64 2
run frame0 1024 1 50 3840 2160
run frame0 1024 2 50 3840 2160 8 3 8
run frame0 256 2 50 1920 1080 8 3 8
run lattice 64 2 32
run frame0 2000 2 50 640 360
