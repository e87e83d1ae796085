// Exploratory (not product): 1 / RN(sqrt(x)) from a v_rsq_f32 seed, candidate correction sequences, every x in [2^-96, 2^127].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define NC 6
__device__ __forceinline__ float step(float s, float r) { return __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r); }
__device__ __forceinline__ float cand(int c, float x, float s, float y) {
	if(c == 0) return step(s, step(s, y));
	if(c == 1) return step(s, step(s, step(s, y)));
	if(c == 2) return step(s, step(s, __uint_as_float(__float_as_uint(y) - 1u)));
	if(c == 3) return step(s, step(s, __uint_as_float(__float_as_uint(y) + 1u)));
	if(c == 4) return step(s, step(s, __builtin_amdgcn_rcpf(s)));
	// c == 5: seed refined against x instead of s first: y1 = y + y*(0.5 - 0.5*x*y*y) then two steps against s
	float h = 0.5f * y, g = x * y;
	float e = __builtin_fmaf(-h, g, 0.5f);
	float y1 = __builtin_fmaf(y, e, y);
	return step(s, y1);
}
__global__ void probe(unsigned long long *out, uint32_t *ex) {
	const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) << 16;
	unsigned long long bad[NC] = {0};
	for(uint32_t k = 0; k < 65536u; k++) {
		const uint32_t u = base + k;
		if(u < 0x0f800000u || u > 0x7f000000u) continue;
		const float x = __uint_as_float(u);
		const float want = 1.0f / sqrtf(x);
		float y = __builtin_amdgcn_rsqf(x);
		const float g = x * y, h = 0.5f * y;
		const float s = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
#pragma unroll
		for(int c = 0; c < NC; c++) {
			const float got = cand(c, x, s, y);
			if(__float_as_uint(got) != __float_as_uint(want)) {
				bad[c]++;
				if(c == 0) { unsigned int slot = atomicAdd(&ex[0], 1u); if(slot < 8) { ex[1 + 4 * slot] = u; ex[2 + 4 * slot] = __float_as_uint(got); ex[3 + 4 * slot] = __float_as_uint(want); ex[4 + 4 * slot] = __float_as_uint(y); } }
			}
		}
	}
	for(int c = 0; c < NC; c++) if(bad[c]) atomicAdd(&out[c], bad[c]);
}
int main() {
	unsigned long long *d, h[NC]; uint32_t *e, he[64];
	hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
	hipMalloc(&e, sizeof he); hipMemset(e, 0, sizeof he);
	probe<<<128, 256>>>(d, e);
	hipDeviceSynchronize();
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	hipMemcpy(he, e, sizeof he, hipMemcpyDeviceToHost);
	const char *name[NC] = {"rsq seed, 2 steps", "rsq seed, 3 steps", "rsq seed - 1ulp, 2 steps", "rsq seed + 1ulp, 2 steps", "rcp(s) seed, 2 steps", "rsq refined vs x, 1 step"};
	for(int c = 0; c < NC; c++) printf("%-28s mismatches %llu\n", name[c], h[c]);
	for(int i = 0; i < 8 && i < (int)he[0]; i++) printf("x=%08x got=%08x want=%08x rsq=%08x\n", he[1+4*i], he[2+4*i], he[3+4*i], he[4+4*i]);
	return 0;
}
