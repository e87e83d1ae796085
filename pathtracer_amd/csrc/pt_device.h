// pt_device.h — device-side arithmetic of the path tracer, shared by every kernel variant.
//
// Hand-written for gfx950.  Float semantics are the contract: strict IEEE-754 binary32
// in the reference's operation order, no FMA contraction (the build passes
// -ffp-contract=off), correctly rounded divide and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals kept.  Under those rules every
// function below is bit-identical to the CPU oracle (oracle/pt_oracle.c), which is
// itself bit-identical to the strict x86 build of the reference kernel.  Citations are
// to the reference (src/cl/*.cl) for what each function must compute.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PT_NS
#error "include through pt_render_kernels.h (PT_NS and PT_FAST_MATH select the build)"
#endif

#define PT_DEV __device__ __forceinline__

namespace PT_NS {

struct f3 { float x, y, z; };

#if PT_FAST_MATH
// ---------------------------------------------------------------------------------------------------------------------
// The relaxed build (pt_fast).  Its arithmetic is not "some" use of the freedoms the reference's build options grant
// (-cl-mad-enable -cl-no-signed-zeros -cl-fast-relaxed-math, src/c/renderer.c:13) but THE use ROCm's own OpenCL
// compiler makes of them when it builds the reference kernel for this device: every operation below is written out as
// the reference's gfx950 code object (oracle/_ref/ptref_gfx950.hsaco, `make -C oracle ref_gfx950`) performs it — which
// products are fused into which sums and in which order, where a divide became v_rcp_f32 and a product, where
// v_sqrt_f32 / v_rsq_f32 are used bare, how the compiler re-associated the camera offsets and exposure * aperture, and
// AMD's definitions of length() / normalize() / sin() / cos() from its OpenCL device library.  The translation unit is
// compiled with -ffp-contract=off: nothing is fused or re-associated that is not written so here.  Goal and test: the
// image equals the reference kernel's own output on this GPU BIT FOR BIT (tests/test_gpu_reference_kernel.py); what
// stays ours is the structure around the arithmetic (all samples in one launch, LDS tables, lanes sharing a pixel,
// wave-level culling of small spheres, the persistent streaming kernel).
// ---------------------------------------------------------------------------------------------------------------------
PT_DEV float ocl_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }      // v_sqrt_f32, bare (1 ulp)
PT_DEV float ocl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }        // v_rcp_f32
PT_DEV float ocl_rsq(float x) { return __builtin_amdgcn_rsqf(x); }        // v_rsq_f32
// x*x + y*y + z*z and a.x*b.x + a.y*b.y + a.z*b.z as the compiler fuses them: one product, two v_fmac_f32, in x, y, z order
PT_DEV float ocl_sumsq(float x, float y, float z) { return __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)); }
PT_DEV float ocl_dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
// v_sqrt_f32 with the input brought into the normal range first (the compiler's expansion of sqrt for a possibly denormal input)
PT_DEV float ocl_sqrt_scaled(float x) {
	const bool tiny = x < 0x1p-126f;
	const float r = ocl_sqrt(tiny ? __builtin_ldexpf(x, 32) : x);
	return tiny ? __builtin_ldexpf(r, -16) : r;
}
// FLT_MIN <= x < +inf in one unsigned compare (NaN, negative, zero, denormal and infinite x fail it)
PT_DEV bool ocl_plain(float x) { return __float_as_uint(x) - 0x00800000u < 0x7f000000u; }
// The wave's lanes whose x is NOT such a float, as a lane mask: ONE v_cmp_class_f32 on every class but "positive normal"
// (the same set as !ocl_plain(x); written as asm because the compiler turns the class builtin back into add + compare)
PT_DEV uint64_t ocl_odd_lanes(float x) {
	uint64_t m;
	asm("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(x), "s"(0x2ffu));
	return m;
}
// AMD's length(float3): sqrt of the fused sum of squares; operands re-scaled when that sum under- or overflows
// (the rare path: ocl_length_edge)
PT_DEV float ocl_length_edge(float x, float y, float z, float l2) {
	if(l2 >= 0x1p-126f) {
		if(l2 != __builtin_inff()) return ocl_sqrt(l2);
		return 0x1p66f * ocl_sqrt_scaled(ocl_sumsq(0x1p-66f * x, 0x1p-66f * y, 0x1p-66f * z));
	}
	return 0x1p-86f * ocl_sqrt_scaled(ocl_sumsq(0x1p86f * x, 0x1p86f * y, 0x1p86f * z));
}
// `edge` is set where the rare path was taken (the caller's scan then runs serially: closest_hit_serial)
PT_DEV float ocl_length(f3 v, bool &edge) {
	const float l2 = ocl_sumsq(v.x, v.y, v.z);
	float r = ocl_sqrt(l2);
	edge = false;
	// one wave-level test for the rare path (no exec-mask juggling around the common one)
	if(__builtin_expect(ocl_odd_lanes(l2) != 0, 0)) { if(!ocl_plain(l2)) { r = ocl_length_edge(v.x, v.y, v.z, l2); edge = true; } }
	return r;
}
PT_DEV float ocl_length(f3 v) { bool e; return ocl_length(v, e); }
// AMD's normalize(float3): v * rsq(fused sum of squares); zero vectors returned as they are, operands re-scaled when
// the sum under- or overflows (infinite components become +-1, the others 0, as the library does) — the rare path;
// returns the factor and may replace the components
PT_DEV float ocl_normalize_edge(float &x, float &y, float &z) {
	if(x == 0.0f && y == 0.0f && z == 0.0f) return 1.0f;
	float l2 = ocl_sumsq(x, y, z);
	if(!(l2 >= 0x1p-126f)) {
		x *= 0x1p86f; y *= 0x1p86f; z *= 0x1p86f;
		l2 = ocl_sumsq(x, y, z);
	} else if(l2 == __builtin_inff()) {
		x *= 0x1p-66f; y *= 0x1p-66f; z *= 0x1p-66f;
		l2 = ocl_sumsq(x, y, z);
		if(l2 == __builtin_inff()) {
			x = __builtin_copysignf(__builtin_isinf(x) ? 1.0f : 0.0f, x);
			y = __builtin_copysignf(__builtin_isinf(y) ? 1.0f : 0.0f, y);
			z = __builtin_copysignf(__builtin_isinf(z) ? 1.0f : 0.0f, z);
			l2 = ocl_sumsq(x, y, z);
		}
	}
	const bool tiny = l2 < 0x1p-126f;
	float r = ocl_rsq(tiny ? l2 * 0x1p24f : l2);
	if(tiny) r *= 0x1p12f;
	return r;
}
PT_DEV f3 ocl_normalize(f3 v) {
	const float l2 = ocl_sumsq(v.x, v.y, v.z);
	float r = ocl_rsq(l2);
	if(__builtin_expect(ocl_odd_lanes(l2) != 0, 0)) { if(!ocl_plain(l2)) r = ocl_normalize_edge(v.x, v.y, v.z); }
	f3 o; o.x = v.x * r; o.y = v.y * r; o.z = v.z * r;
	return o;
}
#endif

// The two operations whose rounding the strict build pins (correctly rounded, as the x86 oracle's) and the
// fast build relaxes to the 1-ulp hardware instructions.
#if PT_FAST_MATH
PT_DEV float pt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
PT_DEV float pt_div(float a, float b) { return a / b; }
#endif

// num / den for operands away from the edges of the format (here: 2^-40 <= den <= 2^40, 2^-80 <= num <= 2^80): the
// compiler's correctly rounded expansion — reciprocal, one Newton step, quotient, two residual corrections — without
// its v_div_scale range scaling (exact powers of two, which change nothing while no intermediate leaves the normal
// range: the smallest here is a residual, a multiple of ulp(den) * ulp(q) >= 2^-126) and without v_div_fixup (zeros,
// infinities, NaNs, denormals: excluded).  8 VALU instructions instead of 11; pt_selftest op 7 compares it with the
// compiler's divide on pseudo-random operands of that range and on its corners (tests/test_gpu_parity.py::
// test_short_divide_is_the_divide_on_its_range; op 8 is the canary: the bare num * v_rcp_f32(den) must FAIL there).
PT_DEV float div_seeded(float num, float den, float r) {
#if PT_FAST_MATH
	return num * r;
#else
	r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
	float q = num * r;
	q = __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
	return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
#endif
}
PT_DEV float div_nt(float num, float den) { return div_seeded(num, den, __builtin_amdgcn_rcpf(den)); }
// The same divide where den = a = RN(RN(sqrt(x))^2) is the scan's |d|^2 (src/cl/ray.cl:73) and y = v_rsq_f32(x) is already
// in a register (sqrt_pos_y below): y * y approximates 1 / a to about 2^-21, which the Newton step of div_seeded squares
// away — one multiply instead of the quarter-rate v_rcp_f32.  pt_selftest op 10 compares it with the compiler's divide.
PT_DEV float div_by_a(float num, float a, float y) { return div_seeded(num, a, y * y); }

PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 scl3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// OpenCL dot/length/normalize as pinned in oracle/ref_shim.cpp
PT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// Correctly rounded sqrt in FIVE instructions: y = v_rsq_f32(x) (1 ulp), g = x * y, h = y / 2, then one Markstein
// step  s = fma(fma(-g, g, x), h, g).  Identical to sqrtf() for every x with 2^-96 <= x < +inf — pt_selftest op 4
// walks every such bit pattern on the device (tools/probe/sqrt_probe.hip is where the sequence was found: the raw
// v_sqrt_f32 is one ulp low on 15 % of the floats and one ulp high on 1e5 of them, so the compiler's expansion, and
// round 1's sqrt_nt, test both neighbours with an FMA residual each: 8 instructions after the v_sqrt_f32).
// A negative or NaN x gives NaN, as sqrtf.  NOT sqrtf for x = +-0 and +inf (0 * inf: NaN) and for 0 < x < 2^-96
// (residual underflow, v_rsq_f32 flushes denormals): sqrt_pos() is for callers that cannot see those inputs or do not
// care what they give; sqrt_z() adds +-0 and +inf; sqrt_cr() is sqrtf for every input.
constexpr uint32_t PT_SQRT_TINY_BITS = 0x0f800000u;   // 2^-96
PT_DEV float sqrt_pos_y(float x, float &y) {
#if PT_FAST_MATH
	y = __builtin_amdgcn_rsqf(x);
	return __builtin_amdgcn_sqrtf(x);
#else
	y = __builtin_amdgcn_rsqf(x);
	const float g = x * y, h = 0.5f * y;
	return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
#endif
}
PT_DEV float sqrt_pos(float x) {
#if PT_FAST_MATH
	return __builtin_amdgcn_sqrtf(x);
#else
	float y;
	return sqrt_pos_y(x, y);
#endif
}

// sqrt_pos plus x = +-0 and +inf (returned as they are, as sqrtf does): sqrtf for x == +-0, every |x| >= 2^-96
// (negative: NaN), +inf and NaN.
PT_DEV float sqrt_z(float x) {
	float s = sqrt_pos(x);
#if !PT_FAST_MATH
	if(__builtin_isfpclass(x, 0x0260)) s = x;      // fcZero | fcPosInf: one v_cmp_class_f32 + one v_cndmask_b32
#endif
	return s;
}

// sqrt_pos for 1 - c*c with |c| <= 1 (random_unit_vector): the argument is 0 or at least 2^-24, and the NaN that
// sqrt_pos gives at 0 becomes the 0 it should be through one v_max_f32 (IEEE maximum: the non-NaN operand).
PT_DEV float sqrt_01(float x) {
	float s = sqrt_pos(x);
#if !PT_FAST_MATH
	asm("v_max_f32 %0, %1, 0" : "=v"(s) : "v"(s));
#endif
	return s;
}

// sqrtf() for every input: the short form where it is exact, the compiler's full expansion in the (practically never
// taken) branch for everything else: +-0, 0 < |x| < 2^-96, negative, +inf, NaN.
PT_DEV float sqrt_cr(float x) {
	float s = sqrt_pos(x);
#if PT_FAST_MATH
	return s;
#endif
	if(__builtin_expect(__float_as_uint(x) - PT_SQRT_TINY_BITS >= 0x7f800000u - PT_SQRT_TINY_BITS, 0)) s = sqrtf(x);
	return s;
}

#if PT_FAST_MATH
PT_DEV float len3(f3 a) { return ocl_length(a); }
#else
PT_DEV float len3(f3 a) { return sqrt_cr(dot3(a, a)); }
#endif

// 1.0f / x: v_rcp_f32 (1 ulp) and two FMA Newton steps, which land on the correctly rounded quotient for every x
// with 2^-126 <= |x| <= 2^126 (pt_selftest op 5 checks all of them against the compiler's correctly rounded
// divide).  rcp_nt is the bare sequence for callers that know x is in that range; rcp_cr adds the guard that sends
// everything else (0, denormals, huge, inf, NaN) to the full divide.
PT_DEV float rcp_nt(float x) {
	float r = __builtin_amdgcn_rcpf(x);
#if !PT_FAST_MATH
	r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
	r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#endif
	return r;
}
PT_DEV float rcp_cr(float x) {
	float r = rcp_nt(x);
#if !PT_FAST_MATH
	if(__builtin_expect((__float_as_uint(x) & 0x7fffffffu) - 0x00800000u > 0x7e800000u - 0x00800000u, 0)) r = 1.0f / x;
#endif
	return r;
}

// normalize() = a * (1 / length(a)) (oracle/ref_shim.cpp): the short sqrt and the short reciprocal under ONE guard,
// with ONE transcendental: the v_rsq_f32 of the sqrt, moved up by one ulp, is also the seed of the reciprocal of the
// rounded length (within 2^-22 of it; the two Newton steps of rcp_nt land on the correctly rounded quotient from there as
// they do from v_rcp_f32 — pt_selftest op 11 walks every x = dot(a, a) of the guarded range against 1.0f / sqrtf(x)).
// The extra ulp matters for the 222 values of x just under a power of four: the root is 2^k (1 - 2^-24), its reciprocal
// rounds to 2^-k (1 + 2^-23), and a seed of exactly 2^-k is a fixed point of the step (tools/probe/invlen_probe.hip).
// x in [2^-96, 2^127] puts sqrt_pos in its exact domain and the length in [2^-48, 2^63.5], well inside rcp_nt's;
// everything else (0, tiny, huge, inf, NaN) takes the compiler's correctly rounded forms.
PT_DEV float inv_len(float x) {
#if PT_FAST_MATH
	return __builtin_amdgcn_rsqf(x);
#else
	float r;
	const float s = sqrt_pos_y(x, r);
	r = __uint_as_float(__float_as_uint(r) + 1u);
	r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
	r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
	if(__builtin_expect(__float_as_uint(x) - PT_SQRT_TINY_BITS > 0x7f000000u - PT_SQRT_TINY_BITS, 0)) r = 1.0f / sqrtf(x);
	return r;
#endif
}
#if PT_FAST_MATH
PT_DEV f3 norm3(f3 a) { return ocl_normalize(a); }
#else
PT_DEV f3 norm3(f3 a) { return scl3(a, inv_len(dot3(a, a))); }
#endif

constexpr uint32_t PT_M31 = 2147483647u;   // MINSTD modulus 2^31 - 1 (src/cl/rng.cl:12)

// x mod (2^31 - 1) for any 64-bit x, by Mersenne folding (2^31 == 1 mod M): no 64-bit divide.
PT_DEV uint32_t mod_m31(uint64_t x) {
	// three 31-bit limbs: bits 0-30, 31-61, 62-63
	uint32_t s = (uint32_t)(x & PT_M31) + (uint32_t)((x >> 31) & PT_M31);   // < 2^32 - 1, no wrap
	uint32_t t = (s & PT_M31) + (s >> 31) + (uint32_t)(x >> 62);            // <= M + 1 + 3
	uint32_t u = t - PT_M31;                                                // wraps high when t < M
	return u < t ? u : t;                                                   // t >= M ? t - M : t
}

// src/cl/rng.cl:7-9 — (16807*a*b) % M * (16807*a*b) % M, left-assoc, wrapping at 2^64.
// `xb` = 16807*b mod 2^64 is the same for every work-item of a sample; the host folds it once.
PT_DEV uint32_t init_rng_2(uint32_t id, uint64_t xb) {
	uint64_t x = (uint64_t)id * xb;
	uint64_t y = (uint64_t)mod_m31(x) * x;
	return mod_m31(y);
}

// src/cl/rng.cl:11-14 — state = 16807*state % M; return (float)(state / 2147483647.0).
// The state stays below 2^31, so the product stays below 2^46 and one fold is enough.
// The f64 divide is replaced by a multiply with RN(1/M): tests/test_rng_exhaustive.py checks on the
// CPU, for every one of the 2^31 - 1 states, that the narrowed float is identical.
PT_DEV float rand_float(uint32_t &state) {
	// p = 16807 * state, folded as (p mod 2^31) + (p >> 31).  Multiplying by 2 * 16807 instead puts p >> 31 in the high
	// word of the product and (p mod 2^31) << 1 in the low word: one shift and one add instead of and, alignbit, add.
	uint64_t p2 = (uint64_t)state * 33614u;
	uint32_t t = ((uint32_t)p2 >> 1) + (uint32_t)(p2 >> 32);
	uint32_t u = t - PT_M31;
	state = u < t ? u : t;
	return (float)((double)state * (1.0 / 2147483647.0));
}

// src/cl/rng.cl:16-18 with a=-1, b=1:  -1 + r*(1 - -1)
#if PT_FAST_MATH
PT_DEV float rand_pm1(uint32_t &state) { return __builtin_fmaf(rand_float(state), 2.0f, -1.0f); }      // v_fma_f32 r, 2.0, -1.0
#else
PT_DEV float rand_pm1(uint32_t &state) { return -1.0f + rand_float(state) * 2.0f; }
#endif

// src/cl/rng.cl:20-34 — theta draw first; y >= 0 always (kept: not a uniform sphere sample).
PT_DEV f3 random_unit_vector(uint32_t &state) {
	float cosTheta = rand_pm1(state);
	float cosPhi = rand_pm1(state);
#if PT_FAST_MATH
	float sinTheta = ocl_sqrt(__builtin_fmaf(-cosTheta, cosTheta, 1.0f));      // 1 - c*c fused, bare v_sqrt_f32
	float sinPhi = ocl_sqrt(__builtin_fmaf(-cosPhi, cosPhi, 1.0f));
#else
	// 1 - c*c with |c| <= 1 is 0 or >= 2^-24 (c*c is a float below 1 or exactly 1)
	float sinTheta = sqrt_01(1.0f - cosTheta * cosTheta);
	float sinPhi = sqrt_01(1.0f - cosPhi * cosPhi);
#endif
	return mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

// src/cl/ray.cl:54-56 — in - (normal*2)*dot(in,normal)
#if PT_FAST_MATH
// in - normal * 2 * dot(in, normal) as compiled: fma(-2 * n, dot, in), the dot product fused in x, y, z order
PT_DEV f3 reflection_dir(f3 in, f3 n) {
	const float dn = ocl_dot(in, n);
	return mk3(__builtin_fmaf(-2.0f * n.x, dn, in.x), __builtin_fmaf(-2.0f * n.y, dn, in.y), __builtin_fmaf(-2.0f * n.z, dn, in.z));
}
#else
PT_DEV f3 reflection_dir(f3 in, f3 n) { return sub3(in, scl3(scl3(n, 2.0f), dot3(in, n))); }
#endif

// pow(x,5) of oracle/ref_shim.cpp = (float)pow((double)x,5.0).  Here: x^5 in f64 (x^2 exact, two roundings
// after) narrowed to f32.  Both are the CORRECTLY ROUNDED binary32 value of the exact x^5 for every float x:
// checked against an exact 128-bit integer arbiter on all 2^23 mantissas of a normal binade (the chain and
// binary32 rounding are invariant under powers of two while the result is a normal float), on every float whose
// x^5 is a binary32 denormal or rounds to zero, and across the overflow edge — tests/test_pow5_exhaustive.py
// (host libm and the chain, CPU) and tests/test_gpu_parity.py::test_pow5_is_correctly_rounded (this function,
// on the device).  No double-rounding case exists.
PT_DEV float pow5(float x) {
#if PT_FAST_MATH
	const float x2 = x * x;
	return x2 * x2 * x;
#endif
	double d = (double)x;
	double d2 = d * d;
	double d4 = d2 * d2;
	return (float)(d4 * d);
}

// src/cl/ray.cl:58-62 — Schlick, in two parts: r0 depends on the material only (for a front-face hit rel is the
// sphere's refIdx, and the host folds the same four IEEE operations into the material table once)
#if PT_FAST_MATH
// (1 - n) / (1 + n) as v_rcp_f32(1 + n) * (1 - n); pow(x, 5) as x^2, x^4, x^4 * x; r0 + (1 - r0) * p fused
PT_DEV float schlick_r0(float rel) {
	const float r = ocl_rcp(1.0f + rel) * (1.0f - rel);
	return r * r;
}
PT_DEV float reflectance(float cosTheta, float r0) { return __builtin_fmaf(1.0f - r0, pow5(1.0f - cosTheta), r0); }
#else
PT_DEV float schlick_r0(float rel) {
	float r = pt_div(1.0f - rel, 1.0f + rel);
	return r * r;
}
PT_DEV float reflectance(float cosTheta, float r0) { return r0 + (1.0f - r0) * pow5(1.0f - cosTheta); }
#endif

// src/cl/ray.cl:64-69
#if PT_FAST_MATH
// cosTheta is the caller's (the compiler computes the same fused chain twice and the values are equal)
PT_DEV f3 refraction_dir(f3 in, f3 n, float rel, float cosTheta) {
	const f3 perp = mk3(__builtin_fmaf(n.x, cosTheta, in.x) * rel, __builtin_fmaf(n.y, cosTheta, in.y) * rel, __builtin_fmaf(n.z, cosTheta, in.z) * rel);
	const float lp = ocl_length(perp);
	const float s = ocl_sqrt(fabsf(__builtin_fmaf(-lp, lp, 1.0f)));           // v_sqrt_f32 |1 - lp*lp|, fused
	return mk3(perp.x - s * n.x, perp.y - s * n.y, perp.z - s * n.z);          // products rounded, then subtracted
}
#else
PT_DEV f3 refraction_dir(f3 in, f3 n, float rel) {
	float cosTheta = fminf(dot3(scl3(in, -1.0f), n), 1.0f);
	f3 perp = scl3(add3(in, scl3(n, cosTheta)), rel);
	float lp = len3(perp);
	f3 par = scl3(n, -sqrt_z(fabsf(1.0f - lp * lp)));       // 1 - lp*lp is 0, inf, or at least 2^-24 in magnitude
	return add3(par, perp);
}
#endif

}   // namespace PT_NS
