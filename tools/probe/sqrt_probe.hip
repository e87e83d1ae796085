// Exploratory (not product): exhaustive comparison of candidate short sqrt / rcp sequences with the compiler's
// correctly rounded sqrtf on every float bit pattern, on the device.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt tools/probe/sqrt_probe.hip -o tools/probe/sqrt_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define NC 8
__device__ __forceinline__ float cand(int c, float x) {
	float s = __builtin_amdgcn_sqrtf(x);
	if(c == 0) return s;                                                    // raw
	if(c == 1) {                                                            // down only
		float sm = __uint_as_float(__float_as_uint(s) - 1u);
		float rm = __builtin_fmaf(-sm, s, x);
		return rm <= 0.0f ? sm : s;
	}
	if(c == 2) {                                                            // up only
		float sp = __uint_as_float(__float_as_uint(s) + 1u);
		float rp = __builtin_fmaf(-sp, s, x);
		return rp > 0.0f ? sp : s;
	}
	if(c == 3) {                                                            // Markstein from rsq
		float y = __builtin_amdgcn_rsqf(x);
		float g = x * y, h = 0.5f * y;
		float r = __builtin_fmaf(-g, g, x);
		return __builtin_fmaf(r, h, g);
	}
	if(c == 4) {                                                            // v_sqrt + one Newton step with rsq-derived h
		float h = 0.5f * __builtin_amdgcn_rsqf(x);
		float r = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(r, h, s);
	}
	if(c == 5) {                                                            // v_sqrt + Newton step with rcp(s)
		float h = 0.5f * __builtin_amdgcn_rcpf(s);
		float r = __builtin_fmaf(-s, s, x);
		return __builtin_fmaf(r, h, s);
	}
	if(c == 6) {                                                            // Markstein with refined h (LLVM FTZ path)
		float y = __builtin_amdgcn_rsqf(x);
		float g = x * y, h = 0.5f * y;
		float e = __builtin_fmaf(-h, g, 0.5f);
		h = __builtin_fmaf(h, e, h);
		g = __builtin_fmaf(g, e, g);
		float d = __builtin_fmaf(-g, g, x);
		return __builtin_fmaf(d, h, g);
	}
	// c == 7: rsq-based, one correction of g only
	float y = __builtin_amdgcn_rsqf(x);
	float g = x * y, h = 0.5f * y;
	float e = __builtin_fmaf(-h, g, 0.5f);
	g = __builtin_fmaf(g, e, g);
	return g;
}
__global__ void probe(unsigned long long *out) {
	// each thread: 65536 consecutive patterns; positive normal floats >= 2^-96 only
	const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) << 16;
	unsigned long long hi[NC] = {0}, lo[NC] = {0}, far[NC] = {0};
	for(uint32_t k = 0; k < 65536u; k++) {
		const uint32_t u = base + k;
		if(u < 0x0f800000u || u >= 0x7f800000u) continue;
		const float x = __uint_as_float(u);
		const uint32_t want = __float_as_uint(sqrtf(x));
#pragma unroll
		for(int c = 0; c < NC; c++) {
			const uint32_t got = __float_as_uint(cand(c, x));
			if(got == want + 1) hi[c]++;
			else if(got + 1 == want) lo[c]++;
			else if(got != want) far[c]++;
		}
	}
	for(int c = 0; c < NC; c++) {
		if(hi[c]) atomicAdd(&out[3 * c], hi[c]);
		if(lo[c]) atomicAdd(&out[3 * c + 1], lo[c]);
		if(far[c]) atomicAdd(&out[3 * c + 2], far[c]);
	}
}
int main() {
	unsigned long long *d, h[3 * NC];
	hipMalloc(&d, sizeof h);
	hipMemset(d, 0, sizeof h);
	probe<<<128, 256>>>(d);      // 32768 threads x 65536 = 2^31 patterns (positive floats)
	hipDeviceSynchronize();
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *name[NC] = {"raw v_sqrt", "down-only fix", "up-only fix", "markstein(rsq)", "v_sqrt+newton(rsq h)", "v_sqrt+newton(rcp h)", "llvm ftz path", "rsq one-step"};
	for(int c = 0; c < NC; c++) printf("%-24s one ulp above %12llu  one ulp below %12llu  further %12llu\n", name[c], h[3*c], h[3*c+1], h[3*c+2]);
	return 0;
}
