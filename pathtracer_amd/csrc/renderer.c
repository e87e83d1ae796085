/* renderer.c — thin C host behind the reference's renderer API (include/renderer.h), calling the
 * MI355X backend through its C ABI (include/pt_hip.h) where the reference called OpenCL.
 *
 * Written from the API's documented behaviour (src/c/renderer.c:60-326 is what each function must
 * do); the structure is ours: the scene is flattened to the backend's interchange records, ALL
 * samples of a render() go down in one call (the reference enqueues and clFinish-es per sample,
 * src/c/renderer.c:170-251), gamma/clamp/8-bit conversion run on the GPU and only 3 bytes per
 * pixel come back, and with PT_NUM_GPUS > 1 the image is cut into row bands over several GPUs.
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "pt_hip.h"
#include "pt_image_io.h"
#include "renderer.h"

#ifdef PT_HAVE_STB
/* the reference's vendored encoder, found on the include path of a drop-in build */
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"
#endif

#define PT_MAX_GPUS 8
#define PT_BAND_ROWS 8

typedef struct pt_backend {
	int n;
	pt_ctx *ctx[PT_MAX_GPUS];
} pt_backend;

/* printf + append to log.txt in the CWD, like the reference's msg() (include/k_util.h:96-113) */
static void host_msg(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
	fflush(stdout);
	FILE *f = fopen("log.txt", "a");
	if(f) {
		va_start(ap, fmt);
		vfprintf(f, fmt, ap);
		va_end(ap);
		fclose(f);
	}
}

static double now_sec(void) {
	struct timeval tv;
	gettimeofday(&tv, NULL);
	return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

/* The per-sample seed stream.  The reference takes `seed = rand()` per sample in a process that
 * never calls srand() (src/c/renderer.c:172): sample k of the process uses the k-th value of glibc's
 * default-seeded TYPE_3 additive-feedback generator, continuing across render() calls.  The HIP
 * runtime draws from libc's rand() state itself while it initialises (measured on the GPU box: the first
 * rand() after hipSetDevice is no longer 1804289383), so calling rand() here would NOT give the
 * reference's seeds.  This is that generator restated (r[i] = r[i-3] + r[i-31] mod 2^32 over a
 * 31-word ring seeded by the MINSTD/Schrage recurrence from 1, first 310 outputs discarded, output
 * r >> 1), private to the renderer so nothing else in the process can disturb it. */
static struct { uint32_t ring[31]; int f, r, ready; } g_seed;

static void seed_stream_init(void) {
	int32_t word = 1;
	g_seed.ring[0] = (uint32_t)word;
	for(int i = 1; i < 31; i++) {
		long hi = word / 127773, lo = word % 127773;
		long w = 16807 * lo - 2836 * hi;
		if(w < 0) w += 2147483647;
		word = (int32_t)w;
		g_seed.ring[i] = (uint32_t)word;
	}
	g_seed.f = 3; g_seed.r = 0; g_seed.ready = 1;
	for(int i = 0; i < 310; i++) {
		g_seed.ring[g_seed.f] += g_seed.ring[g_seed.r];
		g_seed.f = (g_seed.f + 1) % 31; g_seed.r = (g_seed.r + 1) % 31;
	}
}

static uint32_t seed_stream_next(void) {
	if(!g_seed.ready) seed_stream_init();
	uint32_t v = (g_seed.ring[g_seed.f] += g_seed.ring[g_seed.r]);
	g_seed.f = (g_seed.f + 1) % 31; g_seed.r = (g_seed.r + 1) % 31;
	return v >> 1;
}

/* test hook (not part of renderer.h): next value of the seed stream */
unsigned int pt_seed_stream_next(void) { return seed_stream_next(); }

static void backend_destroy(pt_backend *b) {
	if(!b) return;
	for(int i = 0; i < b->n; i++) pt_ctx_destroy(b->ctx[i]);
	free(b);
}

Renderer *create_renderer() {
	Renderer *renderer = calloc(1, sizeof(Renderer));
	if(!renderer) { host_msg("Failed to allocate renderer\n"); exit(1); }
	renderer->sceneInfo.sphereCount = 0;
	renderer->sphereList = NULL;

	int devices = 0;
	if(pt_device_count(&devices) != PT_OK || devices < 1) {
		/* the reference exits when its OpenCL program cannot be built (src/c/renderer.c:84-91);
		 * there is no CPU fallback here either */
		host_msg("Failed to find a HIP device: %s\n", pt_last_error());
		exit(1);
	}
	int want = 1;
	const char *env = getenv("PT_NUM_GPUS");
	if(env && atoi(env) > 0) want = atoi(env);
	if(want > devices) want = devices;
	if(want > PT_MAX_GPUS) want = PT_MAX_GPUS;

	pt_backend *b = calloc(1, sizeof(pt_backend));
	if(!b) { host_msg("Failed to allocate backend\n"); exit(1); }
	for(int i = 0; i < want; i++) {
		if(pt_ctx_create(i, &b->ctx[i]) != PT_OK) {
			host_msg("Failed to create backend context on device %d: %s\n", i, pt_last_error());
			exit(1);
		}
		b->n = i + 1;
	}
	renderer->backend = b;
	return renderer;
}

void set_image_properties(Renderer *renderer, int width, int height, int maxRayDepth) {
	renderer->imageInfo.size.x = width;
	renderer->imageInfo.size.y = height;
	renderer->imageInfo.maxRayDepth = maxRayDepth;
	pt_backend *b = renderer->backend;
	/* the reference leaks the old buffer when called twice (src/c/renderer.c:103); pt_image_alloc frees it */
	for(int i = 0; b && i < b->n; i++)
		if(pt_image_alloc(b->ctx[i], width, height) != PT_OK)
			host_msg("set_image_properties: %s\n", pt_last_error());
}

void clear_scene(Renderer *renderer) {
	renderer->sceneInfo.sphereCount = 0;
	free(renderer->sphereList);
	renderer->sphereList = NULL;
}

void set_background_color(Renderer *renderer, float r, float g, float b) {
	renderer->sceneInfo.color = (pt_float3){r, g, b, 0};
}

Material create_lambertian_material(float r, float g, float b) {
	return (Material){1, (pt_float3){r, g, b, 0}, 0, 0, 0};
}

Material create_metal_material(float r, float g, float b, float tint, float fuzzyness) {
	return (Material){2, (pt_float3){r, g, b, 0}, tint, fuzzyness, 0};
}

Material create_dielectric_material(float r, float g, float b, float tint, float fuzzyness, float refIdx) {
	return (Material){3, (pt_float3){r, g, b, 0}, tint, fuzzyness, refIdx};
}

Material create_light_source_material(float r, float g, float b) {
	return (Material){0, (pt_float3){r, g, b, 0}, 0, 0, 0};
}

void add_sphere(Renderer *renderer, float x, float y, float z, float radius, Material material) {
	int count = renderer->sceneInfo.sphereCount + 1;
	Sphere *grown = realloc(renderer->sphereList, sizeof(Sphere) * (size_t)count);
	if(!grown) { host_msg("add_sphere: out of memory\n"); return; }
	renderer->sphereList = grown;
	Sphere *s = &grown[count - 1];
	memset(s, 0, sizeof *s);
	s->center = (pt_float3){x, y, z, 0};
	s->radius = radius;
	s->material = material;
	renderer->sceneInfo.sphereCount = count;
}

void set_camera_properties(Renderer *renderer, float x, float y, float z, float rotX, float rotY, float rotZ,
                           float sensorWidth, float focalLength, float aperture, float exposure) {
	renderer->camera.pos = (pt_float3){x, y, z, 0};
	renderer->camera.rot = (pt_float3){rotX, rotY, rotZ, 0};
	renderer->camera.sensorWidth = sensorWidth;
	renderer->camera.focalLength = focalLength;
	renderer->camera.aperture = aperture;
	renderer->camera.exposure = exposure;
}

/* Sphere list -> interchange records (include/pt_hip.h) */
static float *flatten_scene(const Renderer *r) {
	int n = r->sceneInfo.sphereCount;
	float *out = malloc(sizeof(float) * PT_SPHERE_FLOATS * (size_t)(n > 0 ? n : 1));
	if(!out) return NULL;
	for(int i = 0; i < n; i++) {
		const Sphere *s = &r->sphereList[i];
		float *o = out + PT_SPHERE_FLOATS * (size_t)i;
		o[0] = s->center.x; o[1] = s->center.y; o[2] = s->center.z; o[3] = s->radius;
		o[4] = (float)s->material.type;
		o[5] = s->material.color.x; o[6] = s->material.color.y; o[7] = s->material.color.z;
		o[8] = s->material.tint; o[9] = s->material.fuzzyness; o[10] = s->material.refIdx; o[11] = 0;
	}
	return out;
}

static Image *render_failed(Renderer *renderer, const char *what) {
	host_msg("%s: %s\n", what, pt_last_error());
	destroy_renderer(renderer);   /* as src/c/renderer.c:244-249 */
	return NULL;
}

Image *render(Renderer *renderer, int samples, int verbose) {
	pt_backend *b = renderer->backend;
	if(!b || b->n < 1) return NULL;
	const int W = renderer->imageInfo.size.x, H = renderer->imageInfo.size.y;
	const int n = renderer->sceneInfo.sphereCount;
	if(samples < 0) samples = 0;

	float *flat = flatten_scene(renderer);
	if(!flat) return NULL;
	for(int i = 0; i < b->n; i++)
		if(pt_scene_upload(b->ctx[i], flat, n) != PT_OK) { free(flat); return render_failed(renderer, "scene upload"); }
	free(flat);

	/* one seed per sample, in order (src/c/renderer.c:172); the stream continues across calls */
	uint64_t *seeds = malloc(sizeof(uint64_t) * (size_t)(samples > 0 ? samples : 1));
	if(!seeds) return NULL;
	for(int k = 0; k < samples; k++) seeds[k] = (uint64_t)seed_stream_next();

	pt_render_args a;
	memset(&a, 0, sizeof a);
	a.width = W; a.height = H; a.max_depth = renderer->imageInfo.maxRayDepth;
	a.background[0] = renderer->sceneInfo.color.x; a.background[1] = renderer->sceneInfo.color.y; a.background[2] = renderer->sceneInfo.color.z;
	a.cam_pos[0] = renderer->camera.pos.x; a.cam_pos[1] = renderer->camera.pos.y; a.cam_pos[2] = renderer->camera.pos.z;
	a.cam_rot[0] = renderer->camera.rot.x; a.cam_rot[1] = renderer->camera.rot.y; a.cam_rot[2] = renderer->camera.rot.z;
	a.sensor_width = renderer->camera.sensorWidth; a.focal_length = renderer->camera.focalLength;
	a.aperture = renderer->camera.aperture; a.exposure = renderer->camera.exposure;
	a.world = b->n; a.band_rows = PT_BAND_ROWS;
	a.async = 1;

	double start = now_sec();
	/* quiet: everything in one launch.  verbose: chunks of 10 samples so the reference's progress
	 * line (every 10th sample, src/c/renderer.c:254-261) still appears; the running mean continues
	 * across chunks bit for bit. */
	int chunk = verbose ? 10 : samples;
	if(samples == 0) {
		/* the reference's loop does not run; its buffer keeps whatever it held — ours is defined: zeros */
		for(int i = 0; i < b->n; i++) if(pt_image_alloc(b->ctx[i], W, H) != PT_OK) { free(seeds); return render_failed(renderer, "image"); }
	}
	for(int done = 0; done < samples; done += chunk) {
		int nfr = samples - done < chunk ? samples - done : chunk;
		a.seeds = seeds + done; a.first_frame = done; a.n_frames = nfr;
		for(int i = 0; i < b->n; i++) {
			a.rank = i;
			if(pt_render(b->ctx[i], &a) != PT_OK) { free(seeds); return render_failed(renderer, "render"); }
		}
		for(int i = 0; i < b->n; i++)
			if(pt_sync(b->ctx[i]) != PT_OK) { free(seeds); return render_failed(renderer, "render"); }
		if(verbose) {
			int frame = done + nfr;
			double per = (now_sec() - start) / frame;
			double et = (samples - frame) * per;
			int hh = (int)(et / 3600), mm = (int)(et / 60) - hh * 60, ss = (int)et - hh * 3600 - mm * 60;
			host_msg("%d/%d samples (%f%%), ET: %02d:%02d:%02d\r", frame, samples, (float)frame / samples * 100, hh, mm, ss);
		}
	}
	free(seeds);
	if(verbose) printf("\n");

	/* gamma + clamp + 8-bit on the device (src/c/renderer.c:275-278), 3 bytes per pixel back */
	Image *image = malloc(sizeof(Image));
	if(!image) return NULL;
	image->width = W; image->height = H;
	image->data = malloc(3 * (size_t)W * (size_t)H);
	if(!image->data) { free(image); return NULL; }
	if(b->n == 1) {
		if(pt_image_read_u8(b->ctx[0], NULL, 0, 0, image->data) != PT_OK) { destroy_image(image); return render_failed(renderer, "read-back"); }
	} else {
		unsigned char *tmp = malloc(3 * (size_t)W * (size_t)H);
		if(!tmp) { destroy_image(image); return NULL; }
		for(int i = 0; i < b->n; i++) {
			if(pt_image_read_u8(b->ctx[i], NULL, 0, 0, tmp) != PT_OK) { free(tmp); destroy_image(image); return render_failed(renderer, "read-back"); }
			for(int row = 0; row < H; row++)
				if((row / PT_BAND_ROWS) % b->n == i)
					memcpy(image->data + 3 * (size_t)W * row, tmp + 3 * (size_t)W * row, 3 * (size_t)W);
		}
		free(tmp);
	}
	return image;
}

void render_to_file(Renderer *renderer, int samples, char *fileName, int verbose) {
	Image *image = render(renderer, samples, verbose);
	if(!image) return;   /* the reference dereferences NULL here (src/c/renderer.c:288-290); we do not */
	write_image(image, fileName);
	destroy_image(image);
}

void destroy_renderer(Renderer *renderer) {
	if(!renderer) return;
	backend_destroy(renderer->backend);
	renderer->backend = NULL;
	/* like the reference, the Renderer itself and its sphere list stay with the caller */
}

static const char *filename_ext(const char *filename) {
	const char *dot = strrchr(filename, '.');
	if(!dot || dot == filename) return "";
	return dot + 1;
}

void write_image(Image *image, char *fileName) {
	const char *format = filename_ext(fileName);
#ifdef PT_HAVE_STB
	if(strcmp(format, "png") == 0) stbi_write_png(fileName, image->width, image->height, 3, image->data, 0);
	else if(strcmp(format, "bmp") == 0) stbi_write_bmp(fileName, image->width, image->height, 3, image->data);
	else if(strcmp(format, "jpg") == 0 || strcmp(format, "jpeg") == 0) stbi_write_jpg(fileName, image->width, image->height, 3, image->data, 95);
	else if(strcmp(format, "tga") == 0) stbi_write_tga(fileName, image->width, image->height, 3, image->data);
#else
	if(strcmp(format, "png") == 0) pt_write_png(fileName, image->width, image->height, image->data);
	else if(strcmp(format, "bmp") == 0) pt_write_bmp(fileName, image->width, image->height, image->data);
	else if(strcmp(format, "jpg") == 0 || strcmp(format, "jpeg") == 0)
		host_msg("write_image: built without stb_image_write.h, no JPEG encoder (rebuild with STB_DIR=<reference>/include)\n");
	else if(strcmp(format, "tga") == 0) pt_write_tga(fileName, image->width, image->height, image->data);
#endif
}

void destroy_image(Image *image) {
	if(!image) return;
	free(image->data);
	free(image);
}
