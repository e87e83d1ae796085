// TEST INFRASTRUCTURE — not part of the product path.
//
// OpenCL C builtin shim for the x86 build of the reference kernel (oracle/_ref).
//
// The reference kernel (/root/reference/src/cl/main.cl, compiled where it lies by
// oracle/Makefile with `clang -x cl`) leaves ten OpenCL builtins undefined.  An
// OpenCL runtime would supply them; none exists in this container, so this file
// does, with the plainest IEEE-754 definitions an OpenCL CPU runtime could use.
// These definitions ARE the float semantics the oracle (oracle/pt_oracle.c) and
// the HIP kernels (pathtracer_amd/csrc/pt_kernels.hip) restate op for op:
//
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z      left to right, no FMA
//   length(v)     = sqrtf(dot(v,v))                     correctly rounded sqrt
//   normalize(v)  = v * (1.0f / length(v))              one correctly rounded divide
//   pow(x,y)      = (float)pow((double)x,(double)y)     correctly rounded to f32 (up to
//                                                       double rounding, p ~ 2^-29)
//   sqrt/fabs/fmin/sin/cos = sqrtf/fabsf/fminf/sinf/cosf of the host libm
//
// This file contains no reference source; it must be compiled with the same clang
// as the kernel object so the C++ manglings of the ext_vector_type overloads match
// (_Z3dotDv3_fS_, _Z6lengthDv3_f, _Z9normalizeDv3_f, ...), and with
// -ffp-contract=off so the definitions above mean what they say.

typedef float float3 __attribute__((ext_vector_type(3)));

extern "C" {
float sqrtf(float);
float fabsf(float);
float fminf(float, float);
float sinf(float);
float cosf(float);
double pow(double, double);
}

// Global work-item id of the work-item being emulated on this thread; set by
// oracle/ref_driver.cpp before every call of the kernel body.
thread_local unsigned long ptref_global_id = 0;

unsigned long get_global_id(unsigned int) { return ptref_global_id; }

float cos(float x) { return cosf(x); }
float sin(float x) { return sinf(x); }
float sqrt(float x) { return sqrtf(x); }
float fabs(float x) { return fabsf(x); }
float fmin(float a, float b) { return fminf(a, b); }
float pow(float x, float y) { return (float)pow((double)x, (double)y); }

float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
float length(float3 v) { return sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z); }
float3 normalize(float3 v) {
	float inv = 1.0f / sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
	float3 r;
	r.x = v.x * inv;
	r.y = v.y * inv;
	r.z = v.z * inv;
	return r;
}
