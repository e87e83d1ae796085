// Exploratory (not product): does a hand-paired packed-f32 sphere test (two spheres per v_pk_* instruction, IEEE-exact
// per component) beat the scalar one?  Same 17 operations per sphere in the reference's order.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/probe/pk_probe.hip -o tools/probe/pk_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2 __attribute__((ext_vector_type(2)));
#define NS 8
__global__ __launch_bounds__(256) void scalar_k(const float4 *__restrict__ sph, float *out, int iters) {
	float4 g[NS];
	for(int i = 0; i < NS; i++) g[i] = sph[i];
	const float4 ray = sph[NS + (threadIdx.x & 63)];
	float ox = threadIdx.x * 1e-3f, oy = ray.w, oz = -1.0f + ray.w, dx = ray.x, dy = ray.y, dz = ray.z, a = (dx * dx + dy * dy) + dz * dz, acc = 0.0f;
	for(int it = 0; it < iters; it++) {
#pragma unroll
		for(int i = 0; i < NS; i++) {
			const float x = ox - g[i].x, y = oy - g[i].y, z = oz - g[i].z;
			const float b = (x * dx + y * dy) + z * dz;
			const float c = x * x + y * y + z * z - g[i].w;
			const float disc = b * b - a * c;
			acc = fminf(acc, disc);      // keep it alive (1 op per sphere in both kernels)
		}
		ox += acc * 1e-30f; oy -= acc * 1e-30f; oz += acc * 2e-30f; dx += acc * 1e-30f; dy -= acc * 1e-30f; dz += acc * 3e-30f; a += acc * 1e-30f;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + ox;
}
__global__ __launch_bounds__(256) void packed_k(const float4 *__restrict__ sph, float *out, int iters) {
	v2 cx[NS / 2], cy[NS / 2], cz[NS / 2], r2[NS / 2];
	for(int i = 0; i < NS / 2; i++) {
		const float4 p = sph[2 * i], q = sph[2 * i + 1];
		cx[i] = (v2){p.x, q.x}; cy[i] = (v2){p.y, q.y}; cz[i] = (v2){p.z, q.z}; r2[i] = (v2){p.w, q.w};
	}
	const float4 ray = sph[NS + (threadIdx.x & 63)];
	float ox = threadIdx.x * 1e-3f, oy = ray.w, oz = -1.0f + ray.w, dx = ray.x, dy = ray.y, dz = ray.z, a = (dx * dx + dy * dy) + dz * dz, acc = 0.0f;
	for(int it = 0; it < iters; it++) {
#pragma unroll
		for(int i = 0; i < NS / 2; i++) {
			const v2 x = (v2){ox, ox} - cx[i], y = (v2){oy, oy} - cy[i], z = (v2){oz, oz} - cz[i];
			const v2 b = (x * (v2){dx, dx} + y * (v2){dy, dy}) + z * (v2){dz, dz};
			const v2 c = x * x + y * y + z * z - r2[i];
			const v2 disc = b * b - (v2){a, a} * c;
			acc = fminf(acc, disc.x);
			acc = fminf(acc, disc.y);
		}
		ox += acc * 1e-30f; oy -= acc * 1e-30f; oz += acc * 2e-30f; dx += acc * 1e-30f; dy -= acc * 1e-30f; dz += acc * 3e-30f; a += acc * 1e-30f;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc + ox;
}
int main() {
	float4 h[NS + 64];
	for(int i = 0; i < NS; i++) h[i] = make_float4(i * 1.5f, -i * 0.7f, 3.0f + i, 1.0f + 0.1f * i);
	for(int i = 0; i < 64; i++) h[NS + i] = make_float4(0.3f + 0.01f * i, 0.4f - 0.003f * i, 0.5f, 0.25f + 0.001f * i);
	float4 *d; float *o;
	(void)hipMalloc(&d, sizeof h); (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
	(void)hipMalloc(&o, 4 * 256 * 2048 * 4);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int iters = 20000;
	for(int rep = 0; rep < 3; rep++) {
		for(int which = 0; which < 2; which++) {
			(void)hipEventRecord(e0, 0);
			if(which == 0) scalar_k<<<2048, 256>>>(d, o, iters); else packed_k<<<2048, 256>>>(d, o, iters);
			(void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1);
			// 2048 blocks x 4 waves = 8192 waves = 8 per SIMD; ns per wave-iteration per SIMD:
			printf("%s: %.2f ms, %.1f ns per sphere test per wave per SIMD\n", which ? "packed" : "scalar", ms, ms * 1e6 / iters / 8 / NS);
		}
	}
	return 0;
}
