#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the whole C host (scene_driver.c, renderer.c, pt_image_io.c) and the host
# half of pt_kernels.hip in ONE executable, build/san/pt_scene_driver_san; the device code is not instrumented
# (GPU sanitizers are not available on the pool).  Run it on the GPU box with tools/san_run.sh.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/san
CL=/opt/rocm/lib/llvm/bin/clang
SAN="-fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer -g"
INC="-Iinclude -Ipathtracer_amd/csrc"
for f in renderer pt_image_io scene_driver; do
	$CL -O1 $SAN -ffp-contract=off $INC -c pathtracer_amd/csrc/$f.c -o build/san/$f.o
done
H="--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 $INC -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -g"
hipcc $H -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -c pathtracer_amd/csrc/pt_kernels.hip -o build/san/k.o
hipcc $H -ffp-contract=off -c pathtracer_amd/csrc/pt_kernels_fast.hip -o build/san/kf.o
hipcc --offload-arch=gfx950 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize build/san/scene_driver.o build/san/renderer.o \
	build/san/pt_image_io.o build/san/k.o build/san/kf.o -o build/san/pt_scene_driver_san -ldl -lm -lpthread
rm -f build/san/*.o
echo built build/san/pt_scene_driver_san
