// pt_kernels_fast.hip — the render kernels once more, with relaxed float rules (see pt_render_kernels.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -fno-slp-vectorize -c
#define PT_NS pt_fast
#define PT_FAST_MATH 1
#include "pt_render_kernels.h"
