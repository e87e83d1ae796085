// pt_kernels_fast.hip — the render kernels once more, with relaxed float rules (see pt_render_kernels.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fno-slp-vectorize -c
#define PT_NS pt_fast
#define PT_FAST_MATH 1
// 7 waves per SIMD (72 VGPRs): left alone the compiler keeps all nine spheres in registers (90 VGPRs, 5 waves): 127.0 -> 123.9 ms
#define PT_LDS_WAVES 7
#include "pt_render_kernels.h"
