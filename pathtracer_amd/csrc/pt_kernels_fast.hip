// pt_kernels_fast.hip — the render kernels once more, as pt_fast: the reference kernel's own arithmetic, every float operation
// as the reference's OpenCL build for gfx950 performs it (pt_device.h, pt_render_kernels.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -c   (nothing may be fused that is not written so)
#define PT_NS pt_fast
#define PT_FAST_MATH 1
// register budget of the lock-step kernel: that of 7 waves per SIMD (it uses 56 VGPRs and runs 8); measured against no cap
// and a cap for 8: 112.0 / 112.9 / 112.7 ms on the headline frame, 48.7 / 48.5 / 48.2 on the lattice — inside the noise
#ifndef PT_FAST_WAVES
#define PT_FAST_WAVES 7
#endif
#define PT_LDS_WAVES PT_FAST_WAVES
#include "pt_render_kernels.h"
