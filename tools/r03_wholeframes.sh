# Whole frames of every BASELINE config at full size and sample count: the HIP builds against the LIVE reference kernel
# (tools/ref_gfx950_time.py ... compare writes gpurun_out/gpuref/time_*.json with pixels_differing_in_bits per build).
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/gpuref
run() { timeout -k 10 400 python3 -u tools/ref_gfx950_time.py "$@" 1 compare > gpurun_out/gpuref/whole_$1_$2x$3_$4.log 2>&1 || { tail -20 gpurun_out/gpuref/whole_$1_$2x$3_$4.log; exit 1; }; grep -o '^hip_[a-z]* {"kernel_ms": [0-9.]*, "pixels": [0-9]*, "pixels_differing_in_bits": [0-9]*' gpurun_out/gpuref/whole_$1_$2x$3_$4.log; }
run frame0 256 256 16 50 &&
run frame0 1920 1080 256 50 &&
run lattice 1920 1080 64 32 &&
run dense500 1920 1080 8 50 &&
run dense10k 1920 1080 64 50 &&
run frame0 3840 2160 1024 50
