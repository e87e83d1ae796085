// TEST INFRASTRUCTURE — not part of the product path.
//
// Alternative OpenCL builtin definitions for the x86 build of the reference kernel (oracle/_ref), compiled three
// times (-DPT_SHIM_VARIANT=1, 2, 3) and linked with the kernel object built WITH THE REFERENCE'S OWN FLAGS
// (-cl-mad-enable -cl-no-signed-zeros -cl-fast-relaxed-math, /root/reference/src/c/renderer.c:13).
//
// Why: the reference pins no OpenCL runtime and asks for relaxed math, so the roundings of
// sin/cos/sqrt/pow/dot/length/normalize are whatever the runtime that happens to run it does.  oracle/ref_shim.cpp is
// ONE legal choice (the plainest IEEE one); these are three others, modelled on what conforming runtimes do under
// -cl-fast-relaxed-math.  Together the builds span an ENVELOPE of legal reference outputs; the parity tolerance of
// this repo is stated against the whole envelope, not against our own choice.
//
//   variant 1 "fma/rsqrt":  dot and length accumulate with FMAs (z first), normalize multiplies by a reciprocal
//                           square root (native_rsqrt style, correctly rounded 1/sqrt), pow(x, y) = exp2(y * log2(x))
//                           evaluated in binary32
//   variant 2 "1 ulp":      the plain definitions of ref_shim.cpp with every sqrt, sin, cos, pow result moved by
//                           -1, 0 or +1 ulp, chosen by a hash of the argument bits (a 1-ulp-accurate native unit;
//                           OpenCL's relaxed-math bounds are looser than this)
//   variant 3 "gpu style":  dot and length accumulate with FMAs (x first), length(v) = x * rsqrt(x) with x = dot(v, v),
//                           normalize through the same rsqrt, sqrt(x) = x * rsqrt(x), pow through exp2/log2 in binary32,
//                           sin/cos evaluated in binary32 after a double-precision 1 / (2 pi) range reduction
//
// No reference source here.  Same build rules as ref_shim.cpp (same clang as the kernel object, -ffp-contract=off so
// an FMA is where one is written and nowhere else).
#ifndef PT_SHIM_VARIANT
#error "compile with -DPT_SHIM_VARIANT=1, 2 or 3"
#endif

typedef float float3 __attribute__((ext_vector_type(3)));

extern "C" {
float sqrtf(float);
float fabsf(float);
float fminf(float, float);
float sinf(float);
float cosf(float);
float exp2f(float);
float log2f(float);
float fmaf(float, float, float);
double pow(double, double);
double sqrt(double);
double sin(double);
double cos(double);
}

thread_local unsigned long ptref_global_id = 0;
unsigned long get_global_id(unsigned int) { return ptref_global_id; }

static inline unsigned f2u(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
static inline float u2f(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

// -1, 0 or +1 ulp (each a third of the time), a fixed function of the argument
static inline float nudge(float r, float arg) {
	if(!(r == r) || r == 0.0f || f2u(r) << 1 >= 0xff000000u) return r;      // NaN, 0, inf: left alone
	unsigned h = f2u(arg) * 2654435761u;
	h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
	const unsigned k = h % 3u;
	return k == 0 ? r : u2f(f2u(r) + (k == 1 ? 1u : ~0u));
}

static inline float rsqrt_cr(float x) { return (float)(1.0 / sqrt((double)x)); }

float fabs(float x) { return fabsf(x); }
float fmin(float a, float b) { return fminf(a, b); }

#if PT_SHIM_VARIANT == 1
float cos(float x) { return cosf(x); }
float sin(float x) { return sinf(x); }
float sqrt(float x) { return sqrtf(x); }
float pow(float x, float y) { return exp2f(y * log2f(x)); }
float dot(float3 a, float3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
float length(float3 v) { return sqrtf(fmaf(v.x, v.x, fmaf(v.y, v.y, v.z * v.z))); }
float3 normalize(float3 v) {
	const float inv = rsqrt_cr(fmaf(v.x, v.x, fmaf(v.y, v.y, v.z * v.z)));
	float3 r; r.x = v.x * inv; r.y = v.y * inv; r.z = v.z * inv; return r;
}
#elif PT_SHIM_VARIANT == 2
float cos(float x) { return nudge(cosf(x), x); }
float sin(float x) { return nudge(sinf(x), x); }
float sqrt(float x) { return nudge(sqrtf(x), x); }
float pow(float x, float y) { return nudge((float)pow((double)x, (double)y), x); }
float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
float length(float3 v) { const float x = (v.x * v.x + v.y * v.y) + v.z * v.z; return nudge(sqrtf(x), x); }
float3 normalize(float3 v) {
	const float x = (v.x * v.x + v.y * v.y) + v.z * v.z;
	const float inv = nudge(1.0f / sqrtf(x), x);
	float3 r; r.x = v.x * inv; r.y = v.y * inv; r.z = v.z * inv; return r;
}
#elif PT_SHIM_VARIANT == 3
static inline float sincos_f32(float x, int want_cos) {
	// range reduction in double, evaluation rounded to binary32 twice (argument and result)
	const double t = (double)x * 0.15915494309189535;          // x / (2 pi)
	const double fr = t - (double)(long long)t;
	const float a = (float)(fr * 6.283185307179586);
	return want_cos ? (float)cos((double)a) : (float)sin((double)a);
}
float cos(float x) { return sincos_f32(x, 1); }
float sin(float x) { return sincos_f32(x, 0); }
float sqrt(float x) { return x == 0.0f ? x : x * rsqrt_cr(x); }
float pow(float x, float y) { return exp2f(y * log2f(x)); }
float dot(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
float length(float3 v) { const float x = fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)); return x == 0.0f ? x : x * rsqrt_cr(x); }
float3 normalize(float3 v) {
	const float inv = rsqrt_cr(fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
	float3 r; r.x = v.x * inv; r.y = v.y * inv; r.z = v.z * inv; return r;
}
#else
#error "unknown PT_SHIM_VARIANT"
#endif
