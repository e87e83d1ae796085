// TEST INFRASTRUCTURE — not part of the product path.
//
// Driver for the x86 build of the reference kernel (oracle/_ref/libptref_*.so).
//
// oracle/Makefile compiles /root/reference/src/cl/main.cl *where it lies* into an
// object (`clang -x cl -Dmain=pt_main`); this file links against
// that object and plays the part of the reference's host loop
// (/root/reference/src/c/renderer.c:153-262): for every sample it takes the next
// seed and runs the kernel body once per work-item.  It exposes a plain C ABI
// (pointers and sizes only) that tests/ and tools/make_golden.py load with ctypes.
//
// The struct declarations below restate the *layouts* of the kernel's argument
// types (/root/reference/src/cl/ray.cl:1-37: 16/32/48/80/48-byte structs) so that
// the SysV calling convention of `__clang_ocl_kern_imp_pt_main` matches; it must be
// compiled with the same clang as the kernel object.

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float float3 __attribute__((ext_vector_type(3)));
typedef int int2 __attribute__((ext_vector_type(2)));

struct Ray { float3 origin; float3 direction; };
struct ImageInfo { int2 size; int maxRayDepth; };
struct SceneInfo { float3 color; int sphereCount; };
struct Material { int type; float3 color; float tint; float fuzzyness; float refIdx; };
struct Sphere { float3 center; float radius; Material material; };
struct Camera { float3 pos; float3 rot; float sensorWidth; float focalLength; float aperture; float exposure; };

static_assert(sizeof(ImageInfo) == 16, "ImageInfo layout");
static_assert(sizeof(SceneInfo) == 32, "SceneInfo layout");
static_assert(sizeof(Material) == 48, "Material layout");
static_assert(sizeof(Sphere) == 80, "Sphere layout");
static_assert(sizeof(Camera) == 48, "Camera layout");

// Symbols of the reference kernel object (C linkage: OpenCL C does not mangle).
extern "C" {
unsigned long init_rng_1(unsigned long a);
unsigned long init_rng_2(unsigned long a, unsigned long b);
float rand_float(unsigned long *rng);
float rand_float_in_range(float a, float b, unsigned long *rng);
float3 random_unit_vector(unsigned long *rng);
float3 rotate_vector(float3 vec, float3 rot);
float3 reflection_dir(float3 in, float3 normal);
float reflectance(float cosTheta, float relativeRefIdx);
float3 refraction_dir(float3 in, float3 normal, float relativeRefIdx);
int hit_sphere(Ray ray, Sphere sphere, float *distance);
float3 cast_ray(SceneInfo sceneInfo, Sphere *sphereList, Ray firstRay, int maxDepth, unsigned long *rng);
// The kernel body with the ordinary C calling convention (the `pt_main` symbol is
// the spir_kernel stub and must not be called from C).
void __clang_ocl_kern_imp_pt_main(float3 *image, ImageInfo imageInfo, SceneInfo sceneInfo, Camera camera,
                                  Sphere *sphereList, int frame, unsigned long seed);
}

extern thread_local unsigned long ptref_global_id;  // oracle/ref_shim.cpp

static float3 f3(const float *p) { float3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; }
static void st3(float *p, float3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// Interchange sphere record shared by oracle/, tests/ and the product's C ABI:
// 12 floats = cx cy cz radius | type(as float 0..3) colr colg colb | tint fuzz refIdx pad
static Sphere *make_spheres(const float *s, int n) {
	Sphere *out = (Sphere *)aligned_alloc(16, sizeof(Sphere) * (size_t)(n > 0 ? n : 1));
	memset(out, 0, sizeof(Sphere) * (size_t)(n > 0 ? n : 1));
	for(int i = 0; i < n; i++) {
		const float *r = s + 12 * (size_t)i;
		out[i].center = f3(r);
		out[i].radius = r[3];
		out[i].material.type = (int)r[4];
		out[i].material.color = f3(r + 5);
		out[i].material.tint = r[8];
		out[i].material.fuzzyness = r[9];
		out[i].material.refIdx = r[10];
	}
	return out;
}

struct Job {
	float3 *image; ImageInfo ii; SceneInfo si; Camera cam; Sphere *spheres;
	const uint64_t *seeds; int first_frame, nframes; long id_begin, id_end;
};

static void *worker(void *arg) {
	Job *j = (Job *)arg;
	// Work-items are independent (RNG state is a function of (id, seed), the only
	// write is the work-item's own pixel), so iterating samples inside the pixel
	// loop gives the same image as the reference's sample-major launch order.
	for(long id = j->id_begin; id < j->id_end; id++) {
		ptref_global_id = (unsigned long)id;
		for(int k = 0; k < j->nframes; k++)
			__clang_ocl_kern_imp_pt_main(j->image, j->ii, j->si, j->cam, j->spheres, j->first_frame + k, j->seeds[k]);
	}
	return NULL;
}

extern "C" {

uint64_t ptref_init_rng_1(uint64_t a) { return init_rng_1(a); }
uint64_t ptref_init_rng_2(uint64_t a, uint64_t b) { return init_rng_2(a, b); }
float ptref_rand_float(uint64_t *state) { unsigned long s = *state; float f = rand_float(&s); *state = s; return f; }
float ptref_rand_float_in_range(float a, float b, uint64_t *state) { unsigned long s = *state; float f = rand_float_in_range(a, b, &s); *state = s; return f; }
void ptref_random_unit_vector(uint64_t *state, float *out) { unsigned long s = *state; st3(out, random_unit_vector(&s)); *state = s; }
void ptref_rotate_vector(const float *vec, const float *rot, float *out) { st3(out, rotate_vector(f3(vec), f3(rot))); }
void ptref_reflection_dir(const float *in, const float *normal, float *out) { st3(out, reflection_dir(f3(in), f3(normal))); }
float ptref_reflectance(float cosTheta, float rel) { return reflectance(cosTheta, rel); }
void ptref_refraction_dir(const float *in, const float *normal, float rel, float *out) { st3(out, refraction_dir(f3(in), f3(normal), rel)); }
int ptref_hit_sphere(const float *origin, const float *dir, const float *center, float radius, float *t) {
	Ray r; r.origin = f3(origin); r.direction = f3(dir);
	Sphere s; memset(&s, 0, sizeof s); s.center = f3(center); s.radius = radius;
	return hit_sphere(r, s, t);
}
// One path: returns the colour cast_ray yields for the given first ray.
void ptref_cast_ray(const float *bg, const float *spheres, int n, const float *origin, const float *dir,
                    int maxDepth, uint64_t *state, float *out) {
	Sphere *sl = make_spheres(spheres, n);
	SceneInfo si; si.color = f3(bg); si.sphereCount = n;
	Ray r; r.origin = f3(origin); r.direction = f3(dir);
	unsigned long s = *state;
	st3(out, cast_ray(si, sl, r, maxDepth, &s));
	*state = s;
	free(sl);
}

// image: W*H*4 floats (the reference's padded float3 accumulator, 16 B per pixel;
// the 4th lane is left alone).  cam: pos3 rot3 sensorWidth focalLength aperture exposure.
// Runs frames first_frame .. first_frame+nframes-1 with seeds[0..nframes-1] on
// work-items id_begin <= id < id_end.
int ptref_render(float *image, int W, int H, int maxDepth, const float *bg, const float *cam,
                 const float *spheres, int n, const uint64_t *seeds, int first_frame, int nframes,
                 long id_begin, long id_end, int nthreads) {
	if(((uintptr_t)image & 15) != 0) return -1;
	Sphere *sl = make_spheres(spheres, n);
	Job base;
	base.image = (float3 *)image;
	base.ii.size.x = W; base.ii.size.y = H; base.ii.maxRayDepth = maxDepth;
	base.si.color = f3(bg); base.si.sphereCount = n;
	base.cam.pos = f3(cam); base.cam.rot = f3(cam + 3);
	base.cam.sensorWidth = cam[6]; base.cam.focalLength = cam[7]; base.cam.aperture = cam[8]; base.cam.exposure = cam[9];
	base.spheres = sl; base.seeds = seeds; base.first_frame = first_frame; base.nframes = nframes;
	if(nthreads < 1) nthreads = 1;
	if(nthreads > 256) nthreads = 256;
	long total = id_end - id_begin;
	// interleaved chunks of 256 work-items keep the threads balanced
	Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)nthreads);
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
	long per = (total + nthreads - 1) / nthreads;
	for(int t = 0; t < nthreads; t++) {
		jobs[t] = base;
		jobs[t].id_begin = id_begin + per * t;
		jobs[t].id_end = jobs[t].id_begin + per < id_end ? jobs[t].id_begin + per : id_end;
		if(jobs[t].id_begin > id_end) jobs[t].id_begin = id_end;
		pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	for(int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
	free(jobs); free(th); free(sl);
	return 0;
}

}  // extern "C"
