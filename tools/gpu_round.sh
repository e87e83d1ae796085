#!/bin/bash
# What a round's GPU check consists of, in one call (gpurun -- 'bash tools/gpu_round.sh <tag>'): the GPU test suite, the
# no-allocation test alone in a fresh process, bench.py, its rocprofv3 kernel trace, and the PMC passes of the headline
# kernel (strict and relaxed).  Outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
tag=${1:-round}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/$tag
mkdir -p "$O"
step() { local log=$1; shift; "$@" > "$O/$log" 2>&1 || { echo "FAILED: $*"; tail -30 "$O/$log"; exit 1; }; tail -2 "$O/$log"; }
step tests.log timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread
step noalloc.log timeout -k 10 300 python3 -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread
step bench.log timeout -k 10 400 python3 bench.py --steps 5 --warmup 1
cd /tmp && export TMPDIR=/tmp
step prof.log timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/prof" -o bench --output-format csv -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras
cd "$R" || exit 1
PT_COUNT=0 step pmc_strict.log bash tools/pmc_passes.sh ${tag}_lds_strict frame0 32 1
PT_COUNT=0 PT_MATH=fast step pmc_fast.log bash tools/pmc_passes.sh ${tag}_lds_fast frame0 32 1
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_lds_strict pt_render > "$O/pmc_lds_strict.md"
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_lds_fast pt_render > "$O/pmc_lds_fast.md"
find "$O/prof" -name "*kernel_stats.csv" -exec head -6 {} \;
