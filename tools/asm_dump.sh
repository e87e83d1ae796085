#!/bin/bash
# Disassemble both builds of the render kernels into /tmp/asm (strict.s, fast.s) and cut out one kernel each.
set -e
mkdir -p /tmp/asm
F="--offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -Ipathtracer_amd/csrc --offload-device-only -S"
hipcc $F -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt pathtracer_amd/csrc/pt_kernels.hip -o /tmp/asm/strict.s 2>/dev/null
hipcc $F -ffp-contract=off pathtracer_amd/csrc/pt_kernels_fast.hip -o /tmp/asm/fast.s 2>/dev/null
awk '/^_ZN9pt_strict20pt_render_lds_kernelILb0ELi3ELi4E/{f=1} f{print} /s_endpgm/{if(f) exit}' /tmp/asm/strict.s > /tmp/asm/lds_ft.s
awk '/^_ZN7pt_fast20pt_render_lds_kernelILb0ELi3ELi4E/{f=1} f{print} /s_endpgm/{if(f) exit}' /tmp/asm/fast.s > /tmp/asm/lds_ft_fast.s
wc -l /tmp/asm/lds_ft.s /tmp/asm/lds_ft_fast.s
