// pt_render_kernels.h — the render kernels, compiled twice into libpt_hip.so:
//   pt_kernels.hip       PT_NS = pt_strict, PT_FAST_MATH = 0, -ffp-contract=off -fno-fast-math
//                        -fhip-fp32-correctly-rounded-divide-sqrt: bit-identical to the CPU oracle
//   pt_kernels_fast.hip  PT_NS = pt_fast,   PT_FAST_MATH = 1, -ffp-contract=off as well: the arithmetic of the
//                        reference kernel AS ROCm's OpenCL COMPILER BUILDS IT FOR gfx950 under the reference's own
//                        options (-cl-mad-enable -cl-fast-relaxed-math, src/c/renderer.c:13), written out operation by
//                        operation (pt_device.h): fused products, v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 bare, AMD's
//                        length() / normalize() / sin() / cos().  Opt-in; bit-identical to the reference kernel's own
//                        output on this GPU (tests/test_gpu_reference_kernel.py), within tolerance of the oracle
#pragma once
#include "pt_params.h"
#include "pt_device.h"

namespace PT_NS {

// Thread -> global work-item id under the row-band partition.  Returns -1 past the end.
PT_DEV int pixel_id(const KParams &p, int t) {
	if(t >= p.localPixels) return -1;
	int lrow = t / p.W, col = t - lrow * p.W;
	int lband = lrow / p.bandRows, rin = lrow - lband * p.bandRows;
	int row = (lband * p.world + p.rank) * p.bandRows + rin;
	return row < p.H ? row * p.W + col : -1;
}

// src/cl/main.cl:8-17 — the part of the camera ray that does not depend on the sample.
#if PT_FAST_MATH
// As compiled for gfx950: x / (float)W is a product with v_rcp_f32, the two offsets were re-associated to
// ((-+2 * sensorWidth) * (float)(index - size / 2)) * rcp(W) — the aspect ratio cancelled against 1 / H —, the rotation
// is three fused multiply-adds around rounded products, and sin / cos are the device library's (the same functions a HIP
// translation unit gets; launch-uniform, evaluated once per pixel here).
PT_DEV f3 camera_origin(const KParams &p, int id) {
	const int row = id / p.W, column = id - row * p.W;
	const float rcpW = ocl_rcp((float)p.W);
	const float ox = ((-2.0f * p.sensorWidth) * (float)(column - p.W / 2)) * rcpW;      // -xOffset
	const float oy = ((2.0f * p.sensorWidth) * (float)(row - p.H / 2)) * rcpW;          // yOffset
	const float oz = p.focalLength;
	const float cy = cosf(p.rotY), sy = sinf(p.rotY), cx = cosf(p.rotX), sx = sinf(p.rotX);
	const f3 r = mk3(__builtin_fmaf(ox, cy, oz * sy), __builtin_fmaf(oy, cx, -(oz * sx)), __builtin_fmaf(oy, sx, oz * cx));
	return add3(mk3(p.px, p.py, p.pz), r);
}
#else
PT_DEV f3 camera_origin(const KParams &p, int id) {
	int row = id / p.W, column = id - row * p.W;
	float aspect = (float)p.W / (float)p.H;
	float xOff = 2.0f * (float)(column - p.W / 2) / (float)p.W * p.sensorWidth;
	float yOff = 2.0f * (float)(row - p.H / 2) / (float)p.H * p.sensorWidth / aspect;
	f3 v = mk3(-xOff, yOff, p.focalLength);
	// rotate_vector as it nets out (src/cl/ray.cl:39-52)
	f3 r = mk3(v.x * p.cosRy + v.z * p.sinRy, v.y * p.cosRx - v.z * p.sinRx, v.y * p.sinRx + v.z * p.cosRx);
	return add3(mk3(p.px, p.py, p.pz), r);
}
#endif

// ---------------------------------------------------------------------------------------
// Per-path state and the pieces every kernel variant shares
// ---------------------------------------------------------------------------------------
struct Path {
	f3 o, d;          // ray (direction NOT normalised after the first bounce, src/cl/ray.cl:138)
	f3 mask;          // ray.cl:93
	uint32_t rng;     // MINSTD state (src/cl/rng.cl)
};

// src/cl/main.cl:6,19-23 — RNG seeding and the primary ray of sample k of work-item id
PT_DEV void start_sample(Path &pt, const KParams &p, int id, f3 origin, uint64_t xb) {
	const f3 pos = mk3(p.px, p.py, p.pz);
	pt.rng = init_rng_2((uint32_t)id, xb);                                               // main.cl:6
#if PT_FAST_MATH
	const f3 ruv = random_unit_vector(pt.rng);
	const f3 target = mk3(__builtin_fmaf(p.aperture, ruv.x, pos.x), __builtin_fmaf(p.aperture, ruv.y, pos.y), __builtin_fmaf(p.aperture, ruv.z, pos.z));
#else
	const f3 target = add3(pos, scl3(random_unit_vector(pt.rng), p.aperture));           // main.cl:19
#endif
	pt.d = norm3(sub3(target, origin));                                                  // main.cl:21
	pt.o = pos;                                                                          // main.cl:23
	pt.mask = mk3(1.0f, 1.0f, 1.0f);                                                     // ray.cl:93
}

// src/cl/main.cl:27-29 — exposure scale, then the running mean with frame = f (two steps, so that lanes sharing a
// pixel scale their own sample once and repeat only the mean)
#if PT_FAST_MATH
// as compiled: color * (exposure * aperture) (re-associated), then fma(image, frame, color) * v_rcp_f32(frame + 1)
PT_DEV f3 expose(f3 color, const KParams &p) { return scl3(color, p.exposure * p.aperture); }
PT_DEV f3 running_mean(f3 acc, f3 color, float f, float f1) {
	const float r = ocl_rcp(f1);
	return mk3(__builtin_fmaf(acc.x, f, color.x) * r, __builtin_fmaf(acc.y, f, color.y) * r, __builtin_fmaf(acc.z, f, color.z) * r);
}
#else
PT_DEV f3 expose(f3 color, const KParams &p) { return scl3(scl3(color, p.exposure), p.aperture); }       // main.cl:27
PT_DEV f3 running_mean(f3 acc, f3 color, float f, float f1) {                                             // main.cl:29
	return mk3(pt_div(acc.x * f + color.x, f1), pt_div(acc.y * f + color.y, f1), pt_div(acc.z * f + color.z, f1));
}
#endif
PT_DEV f3 accumulate(f3 acc, f3 color, const KParams &p, int frame) {
	return running_mean(acc, expose(color, p), (float)frame, (float)(frame + 1));
}

// The running mean when S >= 4 lanes share a pixel (pt_render_lds_kernel): x / f1 as q = x * r, q' = fma(fma(-q, f1, x), r, q)
// with r = RN(1 / f1).  q' is the correctly rounded quotient for EVERY significand of x and EVERY significand of f1:
// both forms are invariant under powers of two in x and in f1 as long as nothing under- or overflows, so the 2^23
// integer divisors in [2^23, 2^24) times the 2^23 significands of one binade are all cases, and tools/meandiv_walk.py
// walked all 2^46 of them on the device's own arithmetic (pt_selftest op 12; 0 mismatches,
// profiles/r03_meandiv_walk.json; tools/probe/meandiv_probe.c is the CPU walk of f1 = 1 .. 65536).  No under- or
// overflow is what mean_safe() guarantees: with every colour component 0 or in [2^-40, 2^40] the mean stays in
// [2^-64, 2^40], x in [2^-64, 2^65], and the residual is a multiple of ulp(q) >= 2^-111.
#if PT_FAST_MATH
PT_DEV bool mean_safe(f3) { return true; }      // the relaxed mean_step IS the relaxed running_mean: nothing to guard
#else
PT_DEV bool mean_safe(f3 c) {
	const uint32_t x = __float_as_uint(c.x), y = __float_as_uint(c.y), z = __float_as_uint(c.z);
	const uint32_t hi = max(max(x, y), z), lo = min(min(x - 1u, y - 1u), z - 1u);      // 0 - 1 wraps to the top: 0 passes
	return hi <= 0x53800000u && lo >= 0x2B7FFFFFu;                                     // negative, NaN, inf: hi fails
}
#endif
#if PT_FAST_MATH
PT_DEV float mean_step(float acc, float c, float f, float f1, float r) { return __builtin_fmaf(acc, f, c) * r; }      // r = v_rcp_f32(f1)
#else
PT_DEV float mean_step(float acc, float c, float f, float f1, float r) {               // main.cl:29, one channel
	const float x = acc * f + c;
	const float q = x * r;
	return __builtin_fmaf(__builtin_fmaf(-q, f1, x), r, q);
}
#endif

#if PT_FAST_MATH
// src/cl/ray.cl:117-171 as the reference's gfx950 code object computes it (see pt_device.h).  `g.w` is not used here.
// LOCKSTEP: where the offset point is formed (same value either way; each kernel keeps the place that is faster for it:
// headline frame 107.8 -> 106.6 ms with the later place, 10^4 spheres on the streaming kernel 148.8 -> 150.2 ms).
// fetchC(): the sphere's matC row (pt_params.h), fetched only by a dielectric hit (LDS record + 48 / the global table).
template <bool LOCKSTEP, typename FetchC, typename StoreM>
PT_DEV bool shade_hit(Path &pt, float t, const float4 g, const float4 mA, const float4 mB, f3 &color, FetchC fetchC, StoreM storeM) {
	const int type = __float_as_int(mA.w);
	const f3 mcol = mk3(mA.x, mA.y, mA.z);
	const f3 d = pt.d;
	const f3 hitPos = mk3(__builtin_fmaf(d.x, t, pt.o.x), __builtin_fmaf(d.y, t, pt.o.y), __builtin_fmaf(d.z, t, pt.o.z));      // ray.cl:117, fused
	f3 normal = ocl_normalize(sub3(hitPos, mk3(g.x, g.y, g.z)));                                                                // ray.cl:118
	f3 offPos = hitPos;
	if(!LOCKSTEP) {
		// above the material branches, for every lane with a hit (the empty asm pins it there: see below)
		offPos = mk3(__builtin_fmaf(-0.0001f, normal.x, hitPos.x), __builtin_fmaf(-0.0001f, normal.y, hitPos.y),
		             __builtin_fmaf(-0.0001f, normal.z, hitPos.z));                                                                 // ray.cl:119, fused, outward normal
		asm volatile("" :: "v"(offPos.x), "v"(offPos.y), "v"(offPos.z));
	}
	const bool front = 0.0f >= ocl_dot(d, normal);                                                                              // ray.cl:122 as !(dot > 0)
	// ray.cl:126 turns the normal of a back face round.  Only the dielectric goes on from a back face (below), so the three
	// selects are made there and not for every hit.
	if(type <= 0) {
		if(front && type == 0) { color = mul3(mcol, pt.mask); return true; }
		return false;
	}
	if(!front && type != 3) return false;
	// The offset point, for the lanes that go on only and straight into the registers the loop carries the origin in: it is
	// formed HERE (the empty asm pins it: the compiler would sink it below the material branches, keep the normal alive for
	// it and copy the normal for the `base` that the branches overwrite), and the hit point is kept alive past it (the asm
	// reads it), so that the three FMAs are three-address and need no copy into the origin's registers afterwards.
	if(LOCKSTEP) {
		offPos = mk3(__builtin_fmaf(-0.0001f, normal.x, hitPos.x), __builtin_fmaf(-0.0001f, normal.y, hitPos.y),
		             __builtin_fmaf(-0.0001f, normal.z, hitPos.z));                                                                 // ray.cl:119, fused, outward normal
		asm volatile("" :: "v"(offPos.x), "v"(offPos.y), "v"(offPos.z), "v"(hitPos.x), "v"(hitPos.y), "v"(hitPos.z));
		pt.o = offPos;
	}
	// The three scattering materials end in `direction = base + random_unit_vector * scale` with the same two RNG draws;
	// ONE shared copy of the draw, so that a wave whose lanes hit different materials runs it once.  The reference's
	// lambertian `normal + ruv` (a plain add) is fma(ruv, 1.0f, normal) bit for bit; the others are fused in its code
	// object as written here.
	const f3 mc = mul3(mcol, pt.mask);                                                   // color * mask, every material
	f3 base = normal, mask = mc;                                                         // ray.cl:136-141
	const float scale = mB.y;      // the table holds 1 for the lambertian, the fuzzyness for the others (pt_scene_upload)
	if(type != 1) {
		if(type == 2) base = reflection_dir(d, normal);                                  // ray.cl:145
		else {                                                                           // ray.cl:150-171
			if(!front) normal = mk3(-normal.x, -normal.y, -normal.z);                    // ray.cl:126
			const float cosTheta = fminf(1.0f, __builtin_fmaf(-d.z, normal.z, __builtin_fmaf(-d.y, normal.y, (-d.x) * normal.x)));
			// 1 / refIdx = v_rcp_f32(refIdx) (ray.cl:156) and Schlick's r0 of the relative index (ray.cl:59-60) depend on the
			// sphere and the face only: derived once per scene with these very instructions (derive_materials), read here
			const float4 mC = fetchC();
			const float sinTheta = ocl_sqrt(__builtin_fmaf(-cosTheta, cosTheta, 1.0f));
			const float rel = front ? mB.z : mC.x;
			const float r0 = front ? mC.y : mC.z;
			bool reflect = 1.0f < sinTheta * rel;
			if(!reflect) reflect = !(reflectance(cosTheta, r0) <= rand_float(pt.rng));      // short-circuit draw, ray.cl:163
			base = reflect ? reflection_dir(d, normal) : refraction_dir(d, normal, rel, cosTheta);
		}
		// mask * (1 - tint) + mask * color * tint as compiled: fma(mask, 1 - tint, (color * mask) * tint)
		const float tint = mB.x, omt = 1.0f - tint;
		mask = mk3(__builtin_fmaf(pt.mask.x, omt, mc.x * tint), __builtin_fmaf(pt.mask.y, omt, mc.y * tint), __builtin_fmaf(pt.mask.z, omt, mc.z * tint));
	}
	const f3 ruv = random_unit_vector(pt.rng);
	const f3 dir = mk3(__builtin_fmaf(ruv.x, scale, base.x), __builtin_fmaf(ruv.y, scale, base.y), __builtin_fmaf(ruv.z, scale, base.z));
	// `base` is kept alive past these three FMAs (the empty asm reads it): with a dead addend the compiler forms them in place
	// on `base` (tied v_fmac) and then copies the result into the registers the loop carries d in; with a live one they
	// are three-address v_fma straight into those registers (three copies fewer per bounce: 108.1 -> 106.9 ms)
	asm volatile("" :: "v"(base.x), "v"(base.y), "v"(base.z));
	pt.d = dir;
	if(!LOCKSTEP) pt.o = offPos;
	storeM(mask);
	return false;
}
#else
// src/cl/ray.cl:117-171 — everything after the closest hit is known.  Returns true when the path
// ends here (front face of a light), with its colour in `color`.  (LOCKSTEP: used by the relaxed form only.)
template <bool LOCKSTEP, typename FetchC, typename StoreM>
PT_DEV bool shade_hit(Path &pt, float t, const float4 g, const float4 mA, const float4 mB, f3 &color, FetchC, StoreM storeM) {
	const int type = __float_as_int(mA.w);
	const f3 mcol = mk3(mA.x, mA.y, mA.z);
	const f3 d = pt.d;
	const f3 hitPos = add3(pt.o, scl3(d, t));                                            // ray.cl:117
	f3 normal = norm3(sub3(hitPos, mk3(g.x, g.y, g.z)));                                 // ray.cl:118
	const f3 offPos = sub3(hitPos, scl3(normal, 0.0001f));                               // ray.cl:119 (outward normal)
	asm volatile("" :: "v"(offPos.x), "v"(offPos.y), "v"(offPos.z));                     // computed here, not below the branches (see the relaxed form above)
	const bool front = !(dot3(d, normal) > 0.0f);                                        // ray.cl:122-128; the back face's normal is
	                                                                                     // turned round where it is used: the dielectric
	if(type <= 0) {                                                                      // ray.cl:130-134
		if(front && type == 0) { color = mul3(pt.mask, mcol); return true; }
		// back face of a light, or a material code none of ray.cl:130-171's branches matches (the table holds -1
		// for those, pt_scene_upload): the ray is left as it is, nothing is drawn from the RNG
		return false;
	}
	if(!front && type != 3) return false;                                                // ray.cl:137,144: only the dielectric scatters from a back face
	// The three scattering materials all end in `direction = base + random_unit_vector * scale` with the SAME two
	// RNG draws; only `base`, `scale` and the mask rule differ.  One shared copy of the draw instead of one per
	// branch: a wave whose lanes hit different materials runs it once (x * 1.0f is x, bit for bit, so the
	// lambertian `normal + random_unit_vector` of ray.cl:138 is unchanged).
	f3 base = normal, mask = mul3(pt.mask, mcol);                                        // ray.cl:138,140
	const float scale = mB.y;      // 1 for the lambertian, the fuzzyness for the others (pt_scene_upload), ray.cl:145,165,168
	if(type != 1) {
		mask = add3(scl3(pt.mask, 1.0f - mB.x), scl3(mask, mB.x));                       // ray.cl:147,170
		if(type == 2) base = reflection_dir(d, normal);                                  // ray.cl:145
		else {                                                                           // ray.cl:150-171
			if(!front) normal = scl3(normal, -1.0f);                                     // ray.cl:126
			float rel = mB.z, r0 = mB.w;                                                 // mB.w: Schlick r0 of refIdx, folded on the host
			if(!front) { rel = pt_div(1.0f, mB.z); r0 = schlick_r0(rel); }               // ray.cl:152
			const float cosTheta = fminf(dot3(scl3(d, -1.0f), normal), 1.0f);
			// 1 - c*c is 0 or at least 2^-24 in magnitude.  sqrt_pos: NaN where sqrtf gives NaN (negative) AND at 0 — either
			// way the comparison below is false, exactly as with sqrtf's 0 (rel * 0 is 0 or NaN, never > 1)
			const float sinTheta = sqrt_pos(1.0f - cosTheta * cosTheta);
			bool reflect = rel * sinTheta > 1.0f;
			if(!reflect) reflect = reflectance(cosTheta, r0) > rand_float(pt.rng);      // short-circuit draw, ray.cl:163
			base = reflect ? reflection_dir(d, normal) : refraction_dir(d, normal, rel);
		}
	}
	pt.d = add3(base, scl3(random_unit_vector(pt.rng), scale));
	pt.o = offPos;
	storeM(mask);
	return false;
}
#endif

// ---------------------------------------------------------------------------------------
// Closest hit (src/cl/ray.cl:101-110 with hit_sphere, :71-87)
// ---------------------------------------------------------------------------------------
struct Hit { bool hit; float t; int idx; };

#if !PT_FAST_MATH
// The reference scan, as written: sqrt and divide per candidate, strict `<` on the rounded distances.
template <typename Tab>
PT_DEV Hit closest_hit_reference(const Tab geom, int n, f3 o, f3 d, float a) {
	Hit h; h.hit = false; h.t = 0.0f; h.idx = 0;
	for(int i = 0; i < n; i++) {
		const float4 g = geom[i];
		const f3 oc = sub3(o, mk3(g.x, g.y, g.z));
		const float bHalf = dot3(oc, d);
		const float c = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z - g.w;
		const float disc = bHalf * bHalf - a * c;
		if(!(disc < 0.0f)) {
			const float t = (-bHalf - sqrtf(disc)) / a;
			if(t > 0.0f && (!h.hit || t < h.t)) { h.hit = true; h.t = t; h.idx = i; }
		}

	}
	return h;
}
#endif

// Branch-free scan with ONE divide per ray instead of one per candidate.  Same result as
// closest_hit_reference for every input; the argument:
//  * num_i = -bHalf_i - sqrt(disc_i) is computed exactly as the reference does (sqrt_nt == sqrtf
//    unless 0 < |disc| < 2^-96; the scan keeps the smallest |disc| and raises `slow` below that).
//    A negative or NaN disc gives a NaN num.
//  * With 2^-40 <= a <= 2^40 and 2^-80 <= num <= 2^80 the quotient t = num/a neither underflows nor
//    overflows, so t > 0 <=> num > 0, and rounding is monotonic: num_i < num_j => t_i <= t_j.
//    A NaN or non-positive num is no hit on both sides.  Outside those ranges: `slow`.
//  * Positive floats order like their bit patterns, so key = bits(num) puts every valid num below
//    KEY_NONE = bits(+inf) and negatives, NaNs and +inf at or above it.  num == +0 (t == 0: no hit in
//    the reference) is the one invalid value that sorts first; it then wins the scan, is seen at the
//    end (best == 0) and raises `slow`.  The scan keeps the smallest key with the lowest index (strict <), and `prev`, the
//    smallest key among LOWER indices (the running minimum just before the last update).
//  * The reference's winner is the lowest index among the spheres whose ROUNDED t is minimal.  That
//    is our index unless a lower-indexed sphere with a larger num rounds to the same t; by
//    monotonicity the smallest such num is `prev`, and if prev > num * (1 + 2^-21) the two quotients
//    are more than an ulp apart.  Otherwise: `slow`.
//  `slow` lanes re-run the reference scan (whole wave skips it when no lane asks).
constexpr uint32_t KEY_NONE = 0x7f800000u;   // bits(+inf)
constexpr uint32_t KEY_NONE_FAST = 0x7f800001u;   // relaxed build: keys are bits(t), +inf included among the hits

// src/cl/ray.cl:72-76 — bHalf and the discriminant, in the reference's operation order
PT_DEV void sphere_disc(const float4 g, f3 o, f3 d, float a, float &bHalf, float &disc) {
	const f3 oc = sub3(o, mk3(g.x, g.y, g.z));
#if PT_FAST_MATH
	// As compiled (g.w is the RADIUS in this build's tables): b = d.x*oc.x, +oc.y*d.y, +oc.z*d.z fused in that order; the sum
	// of squares in the order y, x, z; r*r fused into r*r - |oc|^2 = -c; disc = fma(b, b, RN(a * -c)).
	bHalf = __builtin_fmaf(oc.z, d.z, __builtin_fmaf(oc.y, d.y, d.x * oc.x));
	const float q = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.x, oc.x, oc.y * oc.y));
	const float negc = __builtin_fmaf(g.w, g.w, -q);
	disc = __builtin_fmaf(bHalf, bHalf, a * negc);
#else
	bHalf = dot3(oc, d);
	const float c = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z - g.w;
	disc = bHalf * bHalf - a * c;
#endif
}

// smallest |disc| seen, as one v_min_f32 with the |x| source modifier (fminf(fabsf()) costs two extra
// canonicalising v_max); a NaN disc leaves it unchanged
PT_DEV void track_tiny(float disc, float &tiny) {
#if !PT_FAST_MATH
	asm("v_min_f32 %0, %0, |%1|" : "+v"(tiny) : "v"(disc));
#endif
}
PT_DEV void track_tiny2(float discA, float discB, float &tiny) {
#if !PT_FAST_MATH
	asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(tiny) : "v"(discA), "v"(discB));
#endif
}

PT_DEV void candidate_key(float bHalf, float disc, int i, uint32_t &best, uint32_t &prev, int &idx, float ra) {
#if PT_FAST_MATH
	// relaxed build: the candidate's own distance, t = v_rcp_f32(a) * (-sqrt(disc) - b) as the reference's code object
	// forms it, is the key — the strict `<` of ray.cl:104 on the very floats the reference compares, lowest index first.
	// A negative disc gives a NaN t, which is no hit (ray.cl:78).
	const float tf = ra * (-__builtin_amdgcn_sqrtf(disc) - bHalf);
	// key = bits(t): every t > 0 up to +inf sorts below KEY_NONE_FAST, negative and NaN t above it.  t == +0 (no hit,
	// ray.cl:82) sorts FIRST: it wins, finish_scan sees best == 0 and the lane re-runs the scan serially
	// (closest_hit_serial / closest_hit_wave) — one compare per ray instead of one subtraction per candidate.
	const uint32_t keyf = __float_as_uint(tf);
	const bool updf = keyf < best;
	idx = updf ? i : idx;
	best = updf ? keyf : best;
	return;
#endif
	const float num = -bHalf - sqrt_pos(disc);      // disc == 0 or inf: NaN here, and the lane is `slow` (track_tiny) or has no hit
	const uint32_t key = __float_as_uint(num);
	const bool upd = key < best;
	prev = upd ? best : prev;
	idx = upd ? i : idx;
	best = upd ? key : best;
}

PT_DEV void candidate_update(float bHalf, float disc, int i, uint32_t &best, uint32_t &prev, int &idx, float &tiny, float ra) {
	track_tiny(disc, tiny);
	candidate_key(bHalf, disc, i, best, prev, idx, ra);
}

// `ra`: v_rcp_f32(a), used by the relaxed build only (its keys are the distances themselves)
PT_DEV void test_sphere(const float4 g, int i, f3 o, f3 d, float a, uint32_t &best, uint32_t &prev, int &idx, float &tiny, float ra) {
	float bHalf, disc;
	sphere_disc(g, o, d, a, bHalf, disc);
	candidate_update(bHalf, disc, i, best, prev, idx, tiny, ra);
}

#if PT_FAST_MATH
// The reference's loop as its code object runs it, one sphere after the other (ray.cl:101-110): the relaxed build's
// fallback for the one case its key scan does not decide (a distance of exactly +0, see candidate_key).
template <typename Tab>
PT_DEV Hit closest_hit_serial(const Tab geom, int n, f3 o, f3 d, float a, float ra) {
	Hit h; h.hit = false; h.t = 0.0f; h.idx = 0;
	for(int i = 0; i < n; i++) {
		float bHalf, disc;
		sphere_disc(geom[i], o, d, a, bHalf, disc);
		if(0.0f <= disc) {
			// the comparisons as the reference's finite-math-only code object makes them: a NaN distance passes both
			// (`distance > 0` is !(0 >= t), `newDistance < distance` is !(t >= distance)), and a NaN `distance` is
			// overwritten by whatever hit comes next
			const float t = ra * (-ocl_sqrt(disc) - bHalf);
			if(!(0.0f >= t) && (!h.hit || !(t >= h.t))) { h.hit = true; h.t = t; h.idx = i; }
		}
	}
	return h;
}
#endif

// "This lane can hit the sphere" as a sign bit: disc >= 0 and bHalf < 0, i.e. sign(disc) clear and sign(bHalf) set, in
// full-rate bit operations (the float form, min(disc, -bHalf) >= 0 over four spheres, is four v_min, three v_max and a
// compare, all half rate).  What the sign test leaves out beyond that form is no hit in the reference either:
// bHalf == +0 gives num = -sqrt(disc) <= 0; a NaN disc gives a NaN t (ray.cl:78,82 are both false); and disc is never
// -0 (bHalf * bHalf is +0 or larger, and x - y rounds to -0 only for x == -0).
PT_DEV int can_hit(float bHalf, float disc) { return (int)(~__float_as_uint(disc) & __float_as_uint(bHalf)); }

// Scan of one LDS-staged chunk for tables too large to unroll against (dense scenes): most spheres are
// missed by every ray of the wave, so the sqrt/select part runs only for a group of four in which some
// lane can hit something.  A sphere can only be hit if disc >= 0 and bHalf < 0 (then -bHalf > sqrt(disc)
// is possible): can_hit(); what is skipped is no hit in the
// reference either (disc < 0: ray.cl:78; bHalf >= 0: t <= 0).  count4 is a multiple of 4; the chunk is
// padded with spheres of r^2 = -inf, whose disc is -inf.
PT_DEV void scan_chunk(const float4 *buf, int count4, int base, f3 o, f3 d, float a, float ra, uint32_t &best, uint32_t &prev, int &idx, float &tiny) {
	for(int i = 0; i < count4; i += 4) {
		const float4 g0 = buf[i], g1 = buf[i + 1], g2 = buf[i + 2], g3 = buf[i + 3];
		float b0, b1, b2, b3, d0, d1, d2, d3;
		sphere_disc(g0, o, d, a, b0, d0);
		sphere_disc(g1, o, d, a, b1, d1);
		sphere_disc(g2, o, d, a, b2, d2);
		sphere_disc(g3, o, d, a, b3, d3);
		const int m0 = can_hit(b0, d0), m1 = can_hit(b1, d1), m2 = can_hit(b2, d2), m3 = can_hit(b3, d3);
		if(__builtin_amdgcn_ballot_w64((m0 | m1 | m2 | m3) < 0) != 0) {
			if(__builtin_amdgcn_ballot_w64(m0 < 0) != 0) candidate_update(b0, d0, base + i, best, prev, idx, tiny, ra);
			if(__builtin_amdgcn_ballot_w64(m1 < 0) != 0) candidate_update(b1, d1, base + i + 1, best, prev, idx, tiny, ra);
			if(__builtin_amdgcn_ballot_w64(m2 < 0) != 0) candidate_update(b2, d2, base + i + 2, best, prev, idx, tiny, ra);
			if(__builtin_amdgcn_ballot_w64(m3 < 0) != 0) candidate_update(b3, d3, base + i + 3, best, prev, idx, tiny, ra);
		}
	}
}

// Start value of the scan's small-|disc| tracker: folds the lower bound on `a` into the same compare (finish_scan).
PT_DEV float scan_tiny_init(float a) {
#if PT_FAST_MATH
	return 0.0f;
#else
	return a * 0x1p-56f;
#endif
}

// End of a scan: the single divide, and the checks that send a lane to the reference scan (`slow`).
PT_DEV Hit finish_scan(float a, float rinv, uint32_t best, uint32_t prev, int idx, float tiny, bool &slow) {
	Hit h;
	h.hit = best < KEY_NONE;
	h.idx = idx;
#if PT_FAST_MATH
	h.hit = best < KEY_NONE_FAST;
	h.t = __uint_as_float(best);      // the key IS the winner's distance (candidate_key)
	slow = best == 0u;                // a distance of +0 is no hit in the reference: the serial scan decides
	return h;
#endif
	const float num = __uint_as_float(best);
	const float pnum = __uint_as_float(prev);                // +inf when there is none
	h.t = div_seeded(num, a, rinv);        // rinv: 1 / a to 2^-21 or better; exact inside the ranges checked below, outside them the lane is `slow` or has no hit
	// `tiny` starts at a * 2^-56 (scan_tiny_init) and then takes the smallest |disc|: below 2^-96 means a < 2^-40, or
	// some |disc| under sqrt_nt's domain (an exactly zero disc lands here too: rare, and exact either way)
	slow = tiny < 0x1p-96f;
	slow |= !(a <= 0x1p40f);
	slow |= h.hit && (!(num >= 0x1p-80f && num <= 0x1p80f) || pnum <= num * (1.0f + 0x1p-21f));
	return h;
}


// The reference scan for ONE ray by the whole wave: lane L tests spheres L, L+64, ... with the
// reference's own arithmetic (sqrt and divide per sphere), then a lexicographic (t, index) minimum over
// the lanes gives the smallest rounded t at the lowest index — what ray.cl:101-110 computes serially.
// o, d, a must be wave-uniform.  Used by the streaming kernel for lanes that ask for the exact scan and
// whenever few lanes of a wave are tracing (the tail of a render): 64 spheres per step instead of one.
PT_DEV Hit closest_hit_wave(const float4 *__restrict__ geom, int n, f3 o, f3 d, float a, int lane) {
	uint32_t bk = 0x7f800000u;      // keys are bits(t) - 1; every t > 0 up to +inf is below this
	int bi = 0;
	for(int i = lane; i < n; i += 64) {
		float bHalf, disc;
		sphere_disc(geom[i], o, d, a, bHalf, disc);
#if PT_FAST_MATH
		const float t = ocl_rcp(a) * (-ocl_sqrt(disc) - bHalf);      // as the reference's code object forms it (candidate_key)
#else
		const float t = (-bHalf - sqrtf(disc)) / a;      // disc < 0 (ray.cl:78): sqrtf gives NaN, t is NaN, no hit
#endif
		const uint32_t key = __float_as_uint(t) - 1u;
		if(key < bk) { bk = key; bi = i; }
	}
	for(int off = 32; off > 0; off >>= 1) {
		const uint32_t ok = (uint32_t)__shfl_xor((int)bk, off, 64);
		const int oi = __shfl_xor(bi, off, 64);
		if(ok < bk || (ok == bk && oi < bi)) { bk = ok; bi = oi; }
	}
	Hit h;
	h.hit = bk < 0x7f800000u;
	h.t = __uint_as_float(bk + 1u);
	h.idx = bi;
	return h;
}

// LDS image of the tables for the lock-step kernel: one 48-byte record {geom, matA, matB} per sphere, three records
// and one int4 of their LDS addresses per 160-byte trio, the n % 3 last records after the trios:
//   trio t at 160 t:  record 3t | record 3t+1 | record 3t+2 | {addr(3t), addr(3t+1), addr(3t+2), -}
// The scan walks it with ONE address register per trip and constant offsets; the winner is tracked by its record
// address, which arrives in a VGPR through the (otherwise idle) LDS pipe instead of a v_mov from the scalar loop
// counter per sphere, and the shading reads geometry and materials at that address + 0/16/32 with no address
// arithmetic.  The kernel is bound by the number of VALU instructions issued (profiles/r01_ubench_valu_issue_cost.md).
#if PT_FAST_MATH
// relaxed build: a fourth row per record, matC (pt_params.h), read only by a dielectric hit
#define PT_REC_BYTES 64
#define PT_TRIO_BYTES 208
#else
#define PT_REC_BYTES 48
#define PT_TRIO_BYTES 160
#endif
typedef float pt_v4f __attribute__((ext_vector_type(4)));
typedef int pt_v4i __attribute__((ext_vector_type(4)));
// ds_read_b128 at a 32-bit LDS address (+ a constant offset the compiler folds into the instruction)
PT_DEV float4 lds_f4(uint32_t addr) {
	const pt_v4f v = *(const __attribute__((address_space(3))) pt_v4f *)(uintptr_t)addr;
	return make_float4(v.x, v.y, v.z, v.w);
}
PT_DEV int4 lds_i4(uint32_t addr) {
	const pt_v4i v = *(const __attribute__((address_space(3))) pt_v4i *)(uintptr_t)addr;
	return make_int4(v.x, v.y, v.z, v.w);
}

PT_DEV uint32_t lds_record_offset(int i) {
	const int t = i / 3;
	return (uint32_t)(PT_TRIO_BYTES * t + PT_REC_BYTES * (i - 3 * t));
}
inline size_t lds_table_bytes(int n) { return (size_t)PT_TRIO_BYTES * (size_t)(n / 3) + (size_t)PT_REC_BYTES * (size_t)(n % 3); }

struct LdsGeom {      // geom[i] for closest_hit_reference
	uint32_t base;
	PT_DEV float4 operator[](int i) const { return lds_f4(base + lds_record_offset(i)); }
};

// Wave-level "can any lane hit this sphere": a hit needs disc >= 0 and bHalf < 0 (then -bHalf > sqrt(disc) is
// possible); min(disc, -bHalf) >= 0 is a superset of that, and what it rules out is no hit in the reference either
// (disc < 0: ray.cl:78; bHalf >= 0: t <= 0, ray.cl:82).  One v_min with a source modifier, one compare, one scalar branch.
PT_DEV bool wave_can_hit(float disc, float bHalf) {
	float m;
	asm("v_min_f32 %0, %1, -%2" : "=v"(m) : "v"(disc), "v"(bHalf));
	return __builtin_amdgcn_ballot_w64(m >= 0.0f) != 0;
}

// Returns Hit.idx = LDS address of the winner's record.  `a` may be wrong when dot(d, d) is 0, inf or below 2^-96 (the
// caller uses the unguarded sqrt_pos_y); it is then NaN, inf or below 2^-40: scan_tiny_init() starts the small-|disc|
// tracker below 2^-96 or finish_scan()'s a <= 2^40 test fails, the reference scan is asked for, and that gets the exact value.
// `skipMask` bit t: trio t holds a sphere much smaller than the largest of the table (set by the host); such a sphere
// is rarely a candidate for any of the 64 rays of a wave, and its sqrt/select part is entered only if it is.
// One trio of the lock-step scan (closest_hit_lds)
template <bool SKIP>
PT_DEV void scan_trio(uint32_t addr, bool skipBit, f3 o, f3 d, float a, float ra, uint32_t &best, uint32_t &prev, int &idx, float &tiny) {
	const float4 g0 = lds_f4(addr), g1 = lds_f4(addr + PT_REC_BYTES), g2 = lds_f4(addr + 2 * PT_REC_BYTES);
	const int4 ii = lds_i4(addr + 3 * PT_REC_BYTES);
	float b0, b1, b2, d0, d1, d2;
	sphere_disc(g0, o, d, a, b0, d0);
	sphere_disc(g1, o, d, a, b1, d1);
	sphere_disc(g2, o, d, a, b2, d2);
	if(SKIP && skipBit) {
		// (the empty asm keeps the compiler from flattening the branch into "compute the root anyway and select": it
		// did that to the relaxed build's short update, 65.9 -> 68.5 ms on the 39-sphere lattice)
		if(wave_can_hit(d0, b0)) { asm volatile(""); candidate_update(b0, d0, ii.x, best, prev, idx, tiny, ra); }
		if(wave_can_hit(d1, b1)) { asm volatile(""); candidate_update(b1, d1, ii.y, best, prev, idx, tiny, ra); }
		if(wave_can_hit(d2, b2)) { asm volatile(""); candidate_update(b2, d2, ii.z, best, prev, idx, tiny, ra); }
	} else {
		track_tiny2(d0, d1, tiny);
		track_tiny(d2, tiny);
		candidate_key(b0, d0, ii.x, best, prev, idx, ra);
		candidate_key(b1, d1, ii.y, best, prev, idx, ra);
		candidate_key(b2, d2, ii.z, best, prev, idx, ra);
	}
}

// `ya`: strict build v_rsq_f32(dot(d, d)) (its square seeds the one divide); relaxed build v_rcp_f32(a), handed to every
// candidate (candidate_key).
// NT > 0: the table has exactly NT trios and no remainder (n == 3 * NT) and the small-sphere mask is the compile-time
// constant MASK: the scan is straight-line code, which spares the loop-carried register copies (4 per trio) and the
// merges of the two forms of a trio, and lets the compiler keep the first trio in registers across bounces (the
// reference's nine-sphere scene: 165.9 -> 162 ms).  NT == 0: any table; MASK < 0 is then -1 (small-sphere tests by the
// run-time mask) or -2 (none).
template <int NT, int MASK>
PT_DEV Hit closest_hit_lds(uint32_t base, int n, unsigned long long skipMask, f3 o, f3 d, float a, float ya, bool edge = false) {
#if PT_FAST_MATH
	uint32_t best = KEY_NONE_FAST, prev = KEY_NONE;
#else
	uint32_t best = KEY_NONE, prev = KEY_NONE;
#endif
	float tiny = scan_tiny_init(a);
	int idx = 0;
	uint32_t addr = base;
	int i = 0;
	constexpr bool SKIP = NT > 0 ? MASK != 0 : MASK == -1;
	if(NT > 0) {
		skipMask = (unsigned long long)MASK;      // a compile-time constant from here on
#pragma unroll
		for(int t = 0; t < NT; t++) scan_trio<SKIP>(base + PT_TRIO_BYTES * t, SKIP && ((skipMask >> t) & 1u), o, d, a, ya, best, prev, idx, tiny);
		i = n;
	} else if(SKIP) {
		// the mask as RUNS of trios of one form, each run a loop of its own: a loop body that holds both forms pays for
		// merging them (register copies at every trio): 39-sphere lattice 68.75 -> 64.0 ms
		int left = n / 3;
		while(left > 0) {
			int run = skipMask ? __builtin_ctzll(skipMask) : 64;               // trios without a small sphere
			run = run < left ? run : left;
			for(int j = 0; j < run; j++, addr += PT_TRIO_BYTES) scan_trio<false>(addr, false, o, d, a, ya, best, prev, idx, tiny);
			left -= run;
			skipMask = run < 64 ? skipMask >> run : 0ull;
			run = ~skipMask ? __builtin_ctzll(~skipMask) : 64;                    // trios with one
			run = run < left ? run : left;
			for(int j = 0; j < run; j++, addr += PT_TRIO_BYTES) scan_trio<true>(addr, true, o, d, a, ya, best, prev, idx, tiny);
			left -= run;
			skipMask = run < 64 ? skipMask >> run : 0ull;
		}
		i = 3 * (n / 3);
	} else {
		for(; i + 3 <= n; i += 3, addr += PT_TRIO_BYTES) scan_trio<false>(addr, false, o, d, a, ya, best, prev, idx, tiny);
	}
	for(; i < n; i++, addr += PT_REC_BYTES) test_sphere(lds_f4(addr), (int)addr, o, d, a, best, prev, idx, tiny, ya);
	bool slow;
	Hit h = finish_scan(a, ya * ya, best, prev, idx, tiny, slow);      // ya = v_rsq_f32(dot(d, d)): ya^2 seeds 1 / a (div_by_a)
#if PT_FAST_MATH
	slow |= edge;      // |d|^2 outside the normal range: distances may be NaN, which only the serial scan orders as the reference does
#endif
	if(__builtin_expect(slow, 0)) {
		const LdsGeom tab = {base};
#if PT_FAST_MATH
		h = closest_hit_serial(tab, n, o, d, a, ya);
#else
		const float ln = len3(d);
		h = closest_hit_reference(tab, n, o, d, ln * ln);
#endif
		h.idx = (int)(base + lds_record_offset(h.idx));
	}
	return h;
}

// ---------------------------------------------------------------------------------------
// Lock-step kernel (small tables, staged whole in LDS): one thread per pixel, all samples in-kernel,
// lock-step sample loop, geometry and material tables staged in LDS and read back as
// wave-uniform (geometry) or per-lane (material of the winning sphere) ds_reads, so the
// arithmetic runs on VGPR operands at full VALU rate.
// ---------------------------------------------------------------------------------------
#ifndef PT_LDS_WAVES
#define PT_LDS_BOUNDS __launch_bounds__(256)
#else
#define PT_LDS_BOUNDS __launch_bounds__(256, PT_LDS_WAVES)
#endif
template <bool COUNT, int NT, int MASK>
__global__ PT_LDS_BOUNDS void pt_render_lds_kernel(float4 *__restrict__ image, const KParams p, const SceneTables sc,
                                                             const uint64_t *__restrict__ seedmul,
                                                             unsigned long long *__restrict__ bounceCounter) {
	extern __shared__ float4 lds[];
	const uint32_t ldsBase = (uint32_t)(uintptr_t)lds;      // low half of the flat address = LDS address
	for(int i = (int)threadIdx.x; i < p.n; i += (int)blockDim.x) {
		const uint32_t off = lds_record_offset(i);
		float4 *const rec = (float4 *)((char *)lds + off);
		rec[0] = sc.geom[i];
		rec[1] = sc.matA[i];
		rec[2] = sc.matB[i];
#if PT_FAST_MATH
		rec[3] = sc.matC[i];
#endif
		if(i % 3 == 0 && i + 3 <= p.n) {
			const int a0 = (int)(ldsBase + off);
			*(int4 *)((char *)lds + off + 3 * PT_REC_BYTES) = make_int4(a0, a0 + PT_REC_BYTES, a0 + 2 * PT_REC_BYTES, 0);
		}
	}
	__syncthreads();
	// Sample-interleaved lanes: a pixel is served by S = 1 << p.logS adjacent lanes, lane `sub` tracing the
	// samples k = sub (mod S).  The samples of a pixel are independent paths (src/cl/main.cl:6: the RNG
	// stream depends on (id, seed_k) only); what is serial is the running mean, so after every round of S
	// samples the group accumulates the S colours IN ORDER k, k+1, ... (S >= 4: one channel per lane, see
	// meanSlot below; S = 2: every lane all three, collected by shuffles).  Bit-identical to S = 1; it only
	// multiplies the number of waves for images that cannot fill the chip on their own (a 640x360 frame,
	// or one GPU's bands of a 1080p frame split 8 ways, is ~4000 waves for 8192 wave slots).
	// Two tiers of a launch: the first p.tierPixels pixels of the rank at 2^logS lanes each, the rest — the blocks that
	// are dispatched LAST — at 2^logS2 >= 2^logS lanes each.  A block lasts nFrames / S rounds, and the end of a launch
	// is spent waiting for the last blocks while the chip drains: shorter blocks there cut that tail (pt_render).
	// A block never straddles the tiers (tierThreads is a multiple of the block size).
	int thread = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	int logS = p.logS, pixelBase = 0;
	if(thread >= p.tierThreads) { thread -= p.tierThreads; logS = p.logS2; pixelBase = p.tierPixels; }
	const int S = 1 << logS;
	const int sub = thread & (S - 1);
	const int groupBase = (int)(threadIdx.x & 63) & ~(S - 1);
	const int id = pixel_id(p, pixelBase + (thread >> logS));
	unsigned int myBounces = 0;
	// S >= 4: the ordered mean is split by channel.  After a round every lane leaves {colour, r = 1 / (its frame + 1)}
	// in meanSlot (wave-private rows: a group never spans waves), then lane `sub` of a group runs channel min(sub, 2)
	// through the round's S steps at 5 instructions each (mean_step) instead of all three channels at ~10 each with
	// an IEEE divide per step; lanes 0..2 of the group hold the pixel's mean.  One launch-long switch per wave
	// (`split`): it falls back to the plain form for good the first time a lane sees a value outside mean_safe().
	__shared__ float4 meanSlot[2 * 256];
	// The path's mask lives in LDS between bounce iterations (one private 16-byte slot per lane; volatile accesses, so
	// that the compiler does not forward the store to the next iteration's load and rebuild the register value): as a
	// loop-carried register value it cost six register copies per iteration around the divergent material branches (the
	// structuriser's merge blocks carried it in registers of their own), while an LDS read and a write cost no vector
	// instruction and the LDS pipe is a third busy: headline 104.9 -> 103.3 ms, the strict build 153.1 -> 149.8 ms.  It is
	// read at the top of an iteration and written where a scattering hit forms the new one (shade_hit's storeM).
	__shared__ float4 maskSlot[256];
	const uint32_t maskAddr = (uint32_t)(uintptr_t)&maskSlot[threadIdx.x];
#define PT_MASK_LOAD(m) do { const pt_v4f v_ = *(volatile __attribute__((address_space(3))) pt_v4f *)(uintptr_t)maskAddr; (m) = mk3(v_.x, v_.y, v_.z); } while(0)
#define PT_MASK_STORE(m) do { pt_v4f v_; v_.x = (m).x; v_.y = (m).y; v_.z = (m).z; v_.w = 0.0f; *(volatile __attribute__((address_space(3))) pt_v4f *)(uintptr_t)maskAddr = v_; } while(0)
	if(id >= 0) {
		const f3 origin = camera_origin(p, id);
		f3 acc = mk3(0.0f, 0.0f, 0.0f);
		if(p.firstFrame > 0) {
			float4 v = image[id];
			acc = mk3(v.x, v.y, v.z);
		}
		const int ch = sub < 2 ? sub : 2;
		bool split = S >= 4 && __ballot(!mean_safe(acc)) == 0;
		float accCh = ch == 0 ? acc.x : ch == 1 ? acc.y : acc.z;
		const float2 *const rd = (const float2 *)meanSlot + ((int)(threadIdx.x & ~(unsigned)(S - 1)) * 4 + ch);
		for(int k0 = 0; k0 < p.nFrames; k0 += S) {
			const int k = k0 + sub;
			f3 color = mk3(0.0f, 0.0f, 0.0f);                                            // ray.cl:92
			if(k < p.nFrames) {
				Path pt;
				start_sample(pt, p, id, origin, seedmul[k]);
				PT_MASK_STORE(pt.mask);
				for(int j = 0; j < p.maxDepth; j++) {                                    // ray.cl:96
					if(COUNT) myBounces++;
					PT_MASK_LOAD(pt.mask);
#if PT_FAST_MATH
					// as compiled: length(d) = v_sqrt_f32 of the fused sum of squares, a = len * len, 1 / a = v_rcp_f32(a)
					bool edge;
					const float ln = ocl_length(pt.d, edge);
					const float a = ln * ln, ya = ocl_rcp(a);
#else
					float ya;
					const float ln = sqrt_pos_y(dot3(pt.d, pt.d), ya);                   // see closest_hit_lds()
					const float a = ln * ln;                                             // ray.cl:73 pow(length(d),2)
#endif
#if PT_FAST_MATH
					const Hit h = closest_hit_lds<NT, MASK>(ldsBase, p.n, p.trioSkipMask, pt.o, pt.d, a, ya, edge);
#else
					const Hit h = closest_hit_lds<NT, MASK>(ldsBase, p.n, p.trioSkipMask, pt.o, pt.d, a, ya);
#endif
					if(!h.hit) { color = mul3(pt.mask, mk3(p.bgx, p.bgy, p.bgz)); break; }   // ray.cl:112-115
					const uint32_t rec = (uint32_t)h.idx;
					if(shade_hit<true>(pt, h.t, lds_f4(rec), lds_f4(rec + 16), lds_f4(rec + 32), color, [&]() { return lds_f4(rec + 48); }, [&](f3 m) { PT_MASK_STORE(m); })) break;
				}
			}
			if(S == 1) acc = accumulate(acc, color, p, p.firstFrame + k0);
			else {
				const int round = p.nFrames - k0 < S ? p.nFrames - k0 : S;
				color = expose(color, p);
				float f = (float)(p.firstFrame + k0);      // frame numbers stay below 2^24 when S > 1 (pt_render): f + 1.0f is exact
				if(split && __ballot(!mean_safe(color)) != 0) {
					split = false;
					acc = mk3(__shfl(accCh, groupBase, 64), __shfl(accCh, groupBase + 1, 64), __shfl(accCh, groupBase + 2, 64));
				}
				if(split) {
					const float r = pt_div(1.0f, f + (float)(sub + 1));
					meanSlot[2 * threadIdx.x] = make_float4(color.x, r, color.y, r);
					meanSlot[2 * threadIdx.x + 1] = make_float4(color.z, r, 0.0f, 0.0f);
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
					for(int s = 0; s < round; s++, f += 1.0f) {
						const float2 cr = rd[4 * s];
						accCh = mean_step(accCh, cr.x, f, f + 1.0f, cr.y);
					}
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
				} else
					for(int s = 0; s < round; s++, f += 1.0f) {
						const f3 c = mk3(__shfl(color.x, groupBase + s, 64), __shfl(color.y, groupBase + s, 64), __shfl(color.z, groupBase + s, 64));
						acc = running_mean(acc, c, f, f + 1.0f);
					}
			}
		}
		if(split) acc = mk3(__shfl(accCh, groupBase, 64), __shfl(accCh, groupBase + 1, 64), __shfl(accCh, groupBase + 2, 64));
		if(COUNT && S > 1) {
			unsigned int v = myBounces, tot = 0;
			for(int s = 0; s < S; s++) tot += (unsigned int)__shfl((int)v, groupBase + s, 64);
			if(sub == 0) image[id] = make_float4(acc.x, acc.y, acc.z, (float)tot);
		} else if(sub == 0) image[id] = make_float4(acc.x, acc.y, acc.z, COUNT ? (float)myBounces : 0.0f);
	}
	if(COUNT) {
		unsigned int v = myBounces;
		for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
		if((threadIdx.x & 63) == 0 && v) atomicAdd(bounceCounter, (unsigned long long)v);
	}
}

// ---------------------------------------------------------------------------------------
// Closest hit through the uniform grid (opt-in extension; the reference scans linearly).
// Every sphere that is tested is tested with the reference's own arithmetic (sphere_disc, sqrtf, divide)
// and the winner is the lexicographic minimum of (rounded t, index) — what the linear scan of
// src/cl/ray.cl:101-110 returns — so the result is the reference's whenever the reference's winner is
// among the spheres tested.  It is: the ray is walked cell by cell (3D-DDA) in order of increasing t, a
// sphere is listed in every cell its bounding box, inflated by a tenth of a cell, overlaps — far more
// than the rounding error of the walk or of a grazing hit's t — and the walk only stops after finishing a
// cell whose exit lies beyond the best t found, with a relative margin of 1e-4.  Spheres outside the grid
// (big[]) are tested first, for every ray.  tests/test_gpu_grid.py checks bit-identity with the linear
// scan on dense scenes.
// ---------------------------------------------------------------------------------------
PT_DEV void grid_test(const float4 *__restrict__ geom, int i, f3 o, f3 d, float a, uint32_t &bk, int &bi) {
	float bHalf, disc;
	sphere_disc(geom[i], o, d, a, bHalf, disc);
	if(!(disc < 0.0f)) {                                               // ray.cl:78
#if PT_FAST_MATH
		const float t = ocl_rcp(a) * (-ocl_sqrt(disc) - bHalf);
#else
		const float t = pt_div(-bHalf - sqrt_cr(disc), a);             // ray.cl:80 (sqrt_cr == sqrtf, pt_device.h)
#endif
		const uint32_t key = __float_as_uint(t) - 1u;                  // t > 0 (ray.cl:82) <=> key < bits(+inf)
		if(key < bk || (key == bk && i < bi)) { bk = key; bi = i; }
	}
}

PT_DEV Hit closest_hit_grid(const GridTables &G, const float4 *__restrict__ geom, f3 o, f3 d, float a) {
	uint32_t bk = 0x7f800000u;
	int bi = 0x7fffffff;
	for(int k = 0; k < G.nBig; k++) grid_test(geom, G.big[k], o, d, a, bk, bi);
	// slab test against the grid box, in the ray's own parameter (d is not normalised)
	float t0 = 0.0f, t1 = __builtin_inff();
	bool inside = true;
	const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
	const float lo[3] = {G.minx, G.miny, G.minz}, hi[3] = {G.maxx, G.maxy, G.maxz};
	float inv[3];
#pragma unroll
	for(int ax = 0; ax < 3; ax++) {
		if(dd[ax] != 0.0f) {
			inv[ax] = 1.0f / dd[ax];
			const float ta = (lo[ax] - oo[ax]) * inv[ax], tb = (hi[ax] - oo[ax]) * inv[ax];
			t0 = fmaxf(t0, fminf(ta, tb));
			t1 = fminf(t1, fmaxf(ta, tb));
		} else {
			inv[ax] = __builtin_inff();
			if(!(oo[ax] >= lo[ax] && oo[ax] <= hi[ax])) inside = false;
		}
	}
	// the box is already padded by a cell-tenth beyond every listed sphere; keep rays that graze it
	if(inside && t0 <= t1 * (1.0f + 1e-4f) + 1e-30f && (G.gx | G.gy | G.gz) != 0) {
		const float ts = t0 * (1.0f - 1e-5f);
		const int g[3] = {G.gx, G.gy, G.gz};
		int ic[3], step[3];
		float tMax[3], tDelta[3];
#pragma unroll
		for(int ax = 0; ax < 3; ax++) {
			const float pos = oo[ax] + dd[ax] * ts;
			int c = (int)floorf((pos - lo[ax]) * G.invCell);
			c = c < 0 ? 0 : (c >= g[ax] ? g[ax] - 1 : c);
			ic[ax] = c;
			if(dd[ax] > 0.0f) {
				step[ax] = 1;
				tMax[ax] = (lo[ax] + (float)(c + 1) * G.cell - oo[ax]) * inv[ax];
				tDelta[ax] = G.cell * inv[ax];
			} else if(dd[ax] < 0.0f) {
				step[ax] = -1;
				tMax[ax] = (lo[ax] + (float)c * G.cell - oo[ax]) * inv[ax];
				tDelta[ax] = -G.cell * inv[ax];
			} else {
				step[ax] = 0;
				tMax[ax] = __builtin_inff();
				tDelta[ax] = __builtin_inff();
			}
		}
		// a walk visits at most gx + gy + gz cells
		for(int guard = g[0] + g[1] + g[2] + 3; guard > 0; guard--) {
			const int cellIdx = (ic[2] * g[1] + ic[1]) * g[0] + ic[0];
			const int cb = G.cellStart[cellIdx], ce = G.cellStart[cellIdx + 1];
			for(int k = cb; k < ce; k++) grid_test(geom, G.items[k], o, d, a, bk, bi);
			const float tExit = fminf(tMax[0], fminf(tMax[1], tMax[2]));
			if(bk < 0x7f800000u && __uint_as_float(bk + 1u) < tExit * (1.0f - 1e-4f)) break;
			if(tMax[0] <= tMax[1] && tMax[0] <= tMax[2]) {
				ic[0] += step[0]; tMax[0] += tDelta[0];
				if(ic[0] < 0 || ic[0] >= g[0]) break;
			} else if(tMax[1] <= tMax[2]) {
				ic[1] += step[1]; tMax[1] += tDelta[1];
				if(ic[1] < 0 || ic[1] >= g[1]) break;
			} else {
				ic[2] += step[2]; tMax[2] += tDelta[2];
				if(ic[2] < 0 || ic[2] >= g[2]) break;
			}
		}
	}
	Hit h;
	h.hit = bk < 0x7f800000u;
	h.t = __uint_as_float(bk + 1u);
	h.idx = h.hit ? bi : 0;
	return h;
}

// ---------------------------------------------------------------------------------------
// Streaming kernel (tables that do not fit in LDS: BASELINE config 3, 10^4-10^5 spheres).
// One wave per block, persistent: waves pull pixels from a device-wide queue until it is empty.
//  * A lane owns ONE pixel at a time and runs its samples in order k = 0, 1, ... (the running mean
//    of src/cl/main.cl:29 is a serial chain per pixel), but the moment a path ends the lane starts
//    its next sample, and the moment a pixel is finished the lane takes the next pixel from the
//    queue: the idle lanes are balloted, ONE atomicAdd per wave reserves as many pixels as there are
//    idle lanes and mbcnt hands them out.  Open scenes end paths after 1-3 bounces and the work per
//    pixel varies several-fold across the image; lock-step lanes would idle most of the time.
//  * Each bounce of the wave streams the geometry table through a double-buffered, wave-private LDS
//    chunk (the next chunk's global load is in flight while this one is scanned by scan_chunk()):
//    no block barriers, waves drift freely.
// Legal because a sample's RNG stream is a pure function of (id, seed_k) (src/cl/main.cl:6).
// ---------------------------------------------------------------------------------------
#define PT_CHUNK 128
#define PT_COOP_MAX_LANES 16   // at or below this many tracing lanes the wave scans ray by ray, 64 spheres per step
// strict build: r^2 = -inf, disc = -inf, never a candidate.  Relaxed build (w is the radius): -c = +inf, disc = +inf,
// t = rcp(a) * (-inf - b) = -inf: no hit either.
#define PT_PAD_SPHERE make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff())

template <bool COUNT, bool GRID>
__global__ __launch_bounds__(64) void pt_render_stream_kernel(float4 *__restrict__ image, const KParams p, const SceneTables sc,
                                                               const uint64_t *__restrict__ seedmul,
                                                               unsigned long long *__restrict__ bounceCounter,
                                                               unsigned int *__restrict__ queue, const GridTables grid) {
	__shared__ float4 sChunk[2][PT_CHUNK];
	const int lane = (int)threadIdx.x;
	const int nchunks = (p.n + PT_CHUNK - 1) / PT_CHUNK;
	unsigned int myBounces = 0;
	unsigned int pixelStart = 0;
	bool has = false;          // this lane owns a pixel
	bool exhausted = false;    // wave-uniform: the queue has no more pixels
	int id = 0, k = 0, j = 0;
	Path pt;
	pt.o = pt.d = pt.mask = mk3(0.0f, 0.0f, 0.0f);
	pt.rng = 0;
	f3 origin = mk3(0.0f, 0.0f, 0.0f), acc = mk3(0.0f, 0.0f, 0.0f);
	if(!GRID && nchunks == 1) {
		sChunk[0][lane] = lane < p.n ? sc.geom[lane] : PT_PAD_SPHERE;
		sChunk[0][lane + 64] = lane + 64 < p.n ? sc.geom[lane + 64] : PT_PAD_SPHERE;
		__syncthreads();
	}
	for(;;) {
		// ---- refill: idle lanes take the next pixels of the queue
		const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has);
		if(idle != 0 && !exhausted) {
			const unsigned int cnt = (unsigned int)__popcll(idle);
			unsigned int base = 0;
			if(lane == __ffsll((long long)idle) - 1) base = atomicAdd(queue, cnt);
			base = (unsigned int)__builtin_amdgcn_readlane((int)base, __ffsll((long long)idle) - 1);
			exhausted = base + cnt >= (unsigned int)p.localPixels;
			if(!has) {
				const unsigned int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)idle, 0u));
				const unsigned int t = base + below;
				const int nid = t < (unsigned int)p.localPixels ? pixel_id(p, (int)t) : -1;
				if(nid >= 0) {
					has = true;
					pixelStart = myBounces;
					id = nid; k = 0; j = 0;
					origin = camera_origin(p, id);
					acc = mk3(0.0f, 0.0f, 0.0f);
					if(p.firstFrame > 0) {
						float4 v = image[id];
						acc = mk3(v.x, v.y, v.z);
					}
					start_sample(pt, p, id, origin, seedmul[0]);
				}
			}
		}
		if(__builtin_amdgcn_ballot_w64(has) == 0) {
			if(exhausted) break;
			continue;      // every pixel this wave just drew was out of range: draw again
		}
		// ---- one bounce of every lane that owns a pixel
		const bool tracing = has && j < p.maxDepth;                                      // ray.cl:96
		float a = 1.0f, rcpA = 1.0f;
		bool edge = false;      // relaxed build: |d|^2 outside the normal range (see closest_hit_serial)
		if(tracing) {
			if(COUNT) myBounces++;
#if PT_FAST_MATH
			const float ln = ocl_length(pt.d, edge);
#else
			const float ln = len3(pt.d);
#endif
			a = ln * ln;                                                                 // ray.cl:73
			rcpA = __builtin_amdgcn_rcpf(a);      // strict build: seed of finish_scan's divide; relaxed build: the reference's 1 / a
		}
		Hit h;
		h.hit = false; h.t = 0.0f; h.idx = 0;
		const unsigned long long tmask = __builtin_amdgcn_ballot_w64(tracing);
		unsigned long long coop = tmask;      // lanes whose ray the whole wave scans together
		if(GRID) {
			// opt-in: every lane walks the grid for its own ray
			if(tracing) h = closest_hit_grid(grid, sc.geom, pt.o, pt.d, a);
			coop = 0;
		} else if(__popcll(tmask) > PT_COOP_MAX_LANES) {
			// ray-parallel: every lane scans the whole table for its own ray
#if PT_FAST_MATH
			uint32_t best = KEY_NONE_FAST, prev = KEY_NONE;
#else
			uint32_t best = KEY_NONE, prev = KEY_NONE;
#endif
			float tiny = scan_tiny_init(a);
			int idx = 0;
			if(nchunks > 1) {      // a table of one chunk was staged once, before the loop
				sChunk[0][lane] = lane < p.n ? sc.geom[lane] : PT_PAD_SPHERE;
				sChunk[0][lane + 64] = lane + 64 < p.n ? sc.geom[lane + 64] : PT_PAD_SPHERE;
				__syncthreads();
			}
			for(int c = 0; c < nchunks; c++) {
				float4 r0 = PT_PAD_SPHERE, r1 = PT_PAD_SPHERE;
				const bool more = c + 1 < nchunks;
				if(more) {
					const int i0 = (c + 1) * PT_CHUNK + lane, i1 = i0 + 64;
					if(i0 < p.n) r0 = sc.geom[i0];
					if(i1 < p.n) r1 = sc.geom[i1];
				}
				if(tracing) {
					const int left = p.n - c * PT_CHUNK;
					const int count4 = left >= PT_CHUNK ? PT_CHUNK : ((left + 3) & ~3);
					scan_chunk(sChunk[c & 1], count4, c * PT_CHUNK, pt.o, pt.d, a, rcpA, best, prev, idx, tiny);
				}
				if(more) {
					sChunk[(c + 1) & 1][lane] = r0;
					sChunk[(c + 1) & 1][lane + 64] = r1;
				}
				if(nchunks > 1) __syncthreads();
			}
			bool slow = false;
			if(tracing) h = finish_scan(a, rcpA, best, prev, idx, tiny, slow);
			coop = __builtin_amdgcn_ballot_w64(slow);
		}
		// sphere-parallel: the wave takes the remaining rays one at a time, with the reference's arithmetic
		while(coop != 0) {
			const int L = __ffsll((long long)coop) - 1;
			coop &= coop - 1;
			const f3 ro = mk3(__shfl(pt.o.x, L, 64), __shfl(pt.o.y, L, 64), __shfl(pt.o.z, L, 64));
			const f3 rd = mk3(__shfl(pt.d.x, L, 64), __shfl(pt.d.y, L, 64), __shfl(pt.d.z, L, 64));
			const float ra = __shfl(a, L, 64);
			const Hit hw = closest_hit_wave(sc.geom, p.n, ro, rd, ra, lane);
			if(lane == L) h = hw;
		}
#if PT_FAST_MATH
		// distances that may be NaN are ordered by the reference's own sequential rule only (whatever scan ran above)
		if(__builtin_expect(__builtin_amdgcn_ballot_w64(tracing && edge) != 0, 0)) { if(tracing && edge) h = closest_hit_serial(sc.geom, p.n, pt.o, pt.d, a, rcpA); }
#endif
		bool ended = false;
		f3 color = mk3(0.0f, 0.0f, 0.0f);                                                // ray.cl:92,174
		if(tracing) {
			if(!h.hit) { color = mul3(pt.mask, mk3(p.bgx, p.bgy, p.bgz)); ended = true; }   // ray.cl:112-115
			else ended = shade_hit<false>(pt, h.t, sc.geom[h.idx], sc.matA[h.idx], sc.matB[h.idx], color, [&]() { return sc.matC[h.idx]; }, [&](f3 m) { pt.mask = m; });
			j++;
		}
		if(has && !ended && j >= p.maxDepth) ended = true;                               // loop exhausted: black
		if(ended) {
			acc = accumulate(acc, color, p, p.firstFrame + k);
			k++;
			j = 0;
			if(k < p.nFrames) start_sample(pt, p, id, origin, seedmul[k]);
			else {
				image[id] = make_float4(acc.x, acc.y, acc.z, COUNT ? (float)(myBounces - pixelStart) : 0.0f);
				has = false;
			}
		}
	}
	if(COUNT) {
		unsigned int v = myBounces;
		for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
		if(lane == 0 && v) atomicAdd(bounceCounter, (unsigned long long)v);
	}
}


size_t lds_bytes(int n) { return lds_table_bytes(n); }

#if PT_FAST_MATH
// matC (pt_params.h): per sphere, what a dielectric hit needs beyond the reference's own fields, computed with the
// instructions shade_hit() itself used per hit (v_rcp_f32 for 1 / refIdx, schlick_r0() for either face)
__global__ __launch_bounds__(256) void pt_derive_materials_kernel(const float4 *__restrict__ matB, float4 *__restrict__ matC, int n) {
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if(i >= n) return;
	const float refIdx = matB[i].z;
	const float rinv = ocl_rcp(refIdx);
	matC[i] = make_float4(rinv, schlick_r0(refIdx), schlick_r0(rinv), 0.0f);
}
void derive_materials(const float4 *matB, float4 *matC, int n, hipStream_t stream) {
	if(n > 0) hipLaunchKernelGGL(pt_derive_materials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, matB, matC, n);
}
#endif

// Launch one of the kernels of this build (error reporting: hipGetLastError() in the caller).
void launch_render(const LaunchArgs &la) {
	const dim3 block(256), grid(la.gridPixels), sgrid(la.gridStream);
	if(la.kernel == PT_K_LDS) {
		// instances: with/without the in-kernel bounce count; the generic scan loop with/without the small-sphere tests;
		// and, for a table of exactly nine spheres (the reference's scene), the unrolled scan for each of the 8 masks
		typedef void (*kern_t)(float4 *, const KParams, const SceneTables, const uint64_t *, unsigned long long *);
		static const kern_t nine[2][8] = {
			{pt_render_lds_kernel<false, 3, 0>, pt_render_lds_kernel<false, 3, 1>, pt_render_lds_kernel<false, 3, 2>, pt_render_lds_kernel<false, 3, 3>,
			 pt_render_lds_kernel<false, 3, 4>, pt_render_lds_kernel<false, 3, 5>, pt_render_lds_kernel<false, 3, 6>, pt_render_lds_kernel<false, 3, 7>},
			{pt_render_lds_kernel<true, 3, 0>, pt_render_lds_kernel<true, 3, 1>, pt_render_lds_kernel<true, 3, 2>, pt_render_lds_kernel<true, 3, 3>,
			 pt_render_lds_kernel<true, 3, 4>, pt_render_lds_kernel<true, 3, 5>, pt_render_lds_kernel<true, 3, 6>, pt_render_lds_kernel<true, 3, 7>}};
		kern_t k;
		if(la.p.n == 9 && !la.genericScan) k = nine[la.count ? 1 : 0][(unsigned)(la.p.trioSkipMask & 7u)];
		else if(la.p.trioSkipMask) k = la.count ? pt_render_lds_kernel<true, 0, -1> : pt_render_lds_kernel<false, 0, -1>;
		else k = la.count ? pt_render_lds_kernel<true, 0, -2> : pt_render_lds_kernel<false, 0, -2>;
		hipLaunchKernelGGL(k, grid, block, la.ldsBytes, la.stream, la.image, la.p, la.sc, la.seedmul, la.bounceCounter);
	} else {
		const bool g = la.kernel == PT_K_GRID;
		if(la.count && g) hipLaunchKernelGGL((pt_render_stream_kernel<true, true>), sgrid, dim3(64), 0, la.stream, la.image, la.p, la.sc, la.seedmul, la.bounceCounter, la.queue, la.grid);
		else if(la.count) hipLaunchKernelGGL((pt_render_stream_kernel<true, false>), sgrid, dim3(64), 0, la.stream, la.image, la.p, la.sc, la.seedmul, la.bounceCounter, la.queue, la.grid);
		else if(g) hipLaunchKernelGGL((pt_render_stream_kernel<false, true>), sgrid, dim3(64), 0, la.stream, la.image, la.p, la.sc, la.seedmul, la.bounceCounter, la.queue, la.grid);
		else hipLaunchKernelGGL((pt_render_stream_kernel<false, false>), sgrid, dim3(64), 0, la.stream, la.image, la.p, la.sc, la.seedmul, la.bounceCounter, la.queue, la.grid);
	}
}

}   // namespace PT_NS
