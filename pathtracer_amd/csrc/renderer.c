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
#include <pthread.h>
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

#define PT_MAX_PIPE 4
#define PT_PIPE_DEPTH 2   /* frames queued per GPU before render_to_file() blocks */

/* One render_to_file() call, detached from the Renderer: everything the frame needs is copied at
 * call time (scene, camera, its seeds in stream order), so the caller may go on editing the scene. */
typedef struct pt_job {
	struct pt_job *next;
	float *flat; int n;
	uint64_t *seeds; int samples;
	pt_render_args args;
	char *fileName;
} pt_job;

struct pt_encpool;
typedef struct pt_worker {
	pthread_t thread;
	pthread_mutex_t lock;
	pthread_cond_t wake, room;
	pt_job *head, *tail;
	int pending;          /* queued + running */
	int quit, started;
	pt_ctx *ctx;
	int imgW, imgH;
	struct pt_encpool *enc;
} pt_worker;

/* encoder pool of the frame pipeline: PNG/JPG encoding is host work that outlasts the render of a
 * small frame several times, so finished 8-bit images go to PT_ENCODERS threads */
#define PT_ENCODERS 4
#define PT_ENCODE_QUEUE 8
typedef struct pt_encjob { struct pt_encjob *next; Image image; char *fileName; } pt_encjob;
typedef struct pt_encpool {
	pthread_t thread[PT_ENCODERS];
	pthread_mutex_t lock;
	pthread_cond_t wake, room;
	pt_encjob *head, *tail;
	int pending, quit, started;
} pt_encpool;

typedef struct pt_backend {
	int n;
	pt_ctx *ctx[PT_MAX_GPUS];
	int dev[PT_MAX_GPUS];                       /* HIP device of each rank (all 0 with PT_SHARE_GPU) */
	int imgW[PT_MAX_GPUS], imgH[PT_MAX_GPUS];   /* size of each context's accumulator */
	/* frame pipeline (PT_PIPELINE=k, k = 1..PT_MAX_PIPE): k worker threads per GPU, each with its own
	 * context and stream, frames dealt round-robin.  k = 2 lets two small frames share one GPU: a
	 * 640x360 frame is 3600 waves, under half of the 8192 an MI355X holds. */
	int pipeline;
	long frameNo;
	pt_worker worker[PT_MAX_GPUS * PT_MAX_PIPE];
	pt_ctx *extra[PT_MAX_GPUS * PT_MAX_PIPE];   /* contexts of the workers beyond the first per GPU */
	pt_encpool enc;
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

static void pipeline_drain(pt_backend *b);
static void pipeline_stop(pt_backend *b);

static void backend_destroy(pt_backend *b) {
	if(!b) return;
	pipeline_stop(b);
	for(int i = 0; i < PT_MAX_GPUS * PT_MAX_PIPE; i++) pt_ctx_destroy(b->extra[i]);
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
	/* PT_SHARE_GPU=1 (debugging aid): PT_NUM_GPUS ranks even on fewer GPUs, ALL on GPU 0 — the band partition, the
	 * per-rank launches and pt_combine_bands' merge run as they would on n GPUs (the merge as device copies, not RCCL) */
	const char *share = getenv("PT_SHARE_GPU");
	const int shared = share && atoi(share) > 0;
	if(want > devices && !shared) want = devices;
	if(want > PT_MAX_GPUS) want = PT_MAX_GPUS;

	pt_backend *b = calloc(1, sizeof(pt_backend));
	if(!b) { host_msg("Failed to allocate backend\n"); exit(1); }
	for(int i = 0; i < want; i++) {
		b->dev[i] = shared ? 0 : i;
		if(pt_ctx_create(b->dev[i], &b->ctx[i]) != PT_OK) {
			host_msg("Failed to create backend context on device %d: %s\n", i, pt_last_error());
			exit(1);
		}
		b->n = i + 1;
	}
	const char *pipe = getenv("PT_PIPELINE");
	b->pipeline = pipe ? atoi(pipe) : 0;
	if(b->pipeline < 0) b->pipeline = 0;
	if(b->pipeline > PT_MAX_PIPE) b->pipeline = PT_MAX_PIPE;
	renderer->backend = b;
	return renderer;
}

void set_image_properties(Renderer *renderer, int width, int height, int maxRayDepth) {
	renderer->imageInfo.size.x = width;
	renderer->imageInfo.size.y = height;
	renderer->imageInfo.maxRayDepth = maxRayDepth;
	pt_backend *b = renderer->backend;
	if(b) pipeline_drain(b);
	/* the reference leaks the old buffer when called twice (src/c/renderer.c:103); pt_image_alloc frees it */
	for(int i = 0; b && i < b->n; i++) {
		if(pt_image_alloc(b->ctx[i], width, height) != PT_OK)
			host_msg("set_image_properties: %s\n", pt_last_error());
		else { b->imgW[i] = width; b->imgH[i] = height; }
	}
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

/* kernel arguments 1..3 of the reference's launch (src/c/renderer.c:160-162) */
static void fill_render_args(pt_render_args *a, const Renderer *renderer) {
	memset(a, 0, sizeof *a);
	a->width = renderer->imageInfo.size.x; a->height = renderer->imageInfo.size.y;
	a->max_depth = renderer->imageInfo.maxRayDepth;
	a->background[0] = renderer->sceneInfo.color.x; a->background[1] = renderer->sceneInfo.color.y; a->background[2] = renderer->sceneInfo.color.z;
	a->cam_pos[0] = renderer->camera.pos.x; a->cam_pos[1] = renderer->camera.pos.y; a->cam_pos[2] = renderer->camera.pos.z;
	a->cam_rot[0] = renderer->camera.rot.x; a->cam_rot[1] = renderer->camera.rot.y; a->cam_rot[2] = renderer->camera.rot.z;
	a->sensor_width = renderer->camera.sensorWidth; a->focal_length = renderer->camera.focalLength;
	a->aperture = renderer->camera.aperture; a->exposure = renderer->camera.exposure;
	/* The drop-in renders with the reference's own arithmetic: the float operations of its kernel as its OpenCL build for
	 * this GPU performs them (-cl-fast-relaxed-math, src/c/renderer.c:13; pt_hip.h `fast_math`), so that the image is the
	 * one the reference produces on this device, bit for bit.  PT_MATH=strict selects the IEEE build (bit-identical to
	 * the CPU oracle) instead. */
	a->fast_math = 1;
}

static Image *render_failed(Renderer *renderer, const char *what) {
	host_msg("%s: %s\n", what, pt_last_error());
	destroy_renderer(renderer);   /* as src/c/renderer.c:244-249 */
	return NULL;
}

Image *render(Renderer *renderer, int samples, int verbose) {
	pt_backend *b = renderer->backend;
	if(!b || b->n < 1) return NULL;
	pipeline_drain(b);   /* frames queued by render_to_file() own the contexts until they are done */
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

	/* a pipelined frame of another size may have resized a context's accumulator */
	for(int i = 0; i < b->n; i++)
		if(b->imgW[i] != W || b->imgH[i] != H) {
			if(pt_image_alloc(b->ctx[i], W, H) != PT_OK) { free(seeds); return render_failed(renderer, "image"); }
			b->imgW[i] = W; b->imgH[i] = H;
		}
	pt_render_args a;
	fill_render_args(&a, renderer);
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
	/* several GPUs: the other ranks' bands travel into rank 0's accumulator over xGMI (one packed RCCL message per rank,
	 * peer copies where RCCL cannot be set up: pt_combine_bands), then ONE conversion and ONE 3-byte-per-pixel
	 * read-back, as with a single GPU */
	if(b->n > 1 && samples > 0) {
		if(pt_combine_bands(b->ctx, b->n, 0, PT_BAND_ROWS) != PT_OK) { destroy_image(image); return render_failed(renderer, "combine"); }
		static int told;
		if(!told) { told = 1; host_msg("multi-GPU combine over %d ranks: %s\n", b->n, pt_combine_transport()); }
	}
	if(pt_image_read_u8(b->ctx[0], NULL, 0, 0, image->data) != PT_OK) { destroy_image(image); return render_failed(renderer, "read-back"); }
	return image;
}

/* ---- frame pipeline (scripts/main.lua:25-98 renders 301 frames back to back) ----
 * With PT_PIPELINE=1 render_to_file() only QUEUES the frame: a worker thread per GPU uploads the copied
 * scene, renders, reads the 8-bit image back and encodes the file while the caller builds the next frame,
 * and with several GPUs (PT_NUM_GPUS) whole frames are dealt round-robin instead of cutting each frame
 * into bands.  Seeds are drawn here, at call time, so every frame gets exactly the seeds it would get
 * without the pipeline.  Files are complete when destroy_renderer(), render() or set_image_properties()
 * returns.  Off by default: the file is then on disk when render_to_file() returns, as in the reference. */
static void *encoder_main(void *arg) {
	pt_encpool *e = arg;
	for(;;) {
		pthread_mutex_lock(&e->lock);
		while(!e->head && !e->quit) pthread_cond_wait(&e->wake, &e->lock);
		pt_encjob *j = e->head;
		if(!j) { pthread_mutex_unlock(&e->lock); return NULL; }
		e->head = j->next;
		if(!e->head) e->tail = NULL;
		pthread_mutex_unlock(&e->lock);
		write_image(&j->image, j->fileName);
		free(j->image.data); free(j->fileName); free(j);
		pthread_mutex_lock(&e->lock);
		e->pending--;
		pthread_cond_broadcast(&e->room);
		pthread_mutex_unlock(&e->lock);
	}
}

static void encode_submit(pt_encpool *e, Image *image, const char *fileName) {
	pt_encjob *j = calloc(1, sizeof *j);
	char *name = strdup(fileName);
	if(!j || !name) {   /* out of memory: encode here */
		write_image(image, (char *)fileName);
		free(image->data); free(j); free(name);
		return;
	}
	j->image = *image; j->fileName = name;
	pthread_mutex_lock(&e->lock);
	while(e->pending >= PT_ENCODE_QUEUE) pthread_cond_wait(&e->room, &e->lock);
	if(e->tail) e->tail->next = j; else e->head = j;
	e->tail = j;
	e->pending++;
	pthread_cond_signal(&e->wake);
	pthread_mutex_unlock(&e->lock);
}

static void job_free(pt_job *j) {
	if(!j) return;
	free(j->flat); free(j->seeds); free(j->fileName); free(j);
}

static void job_run(pt_worker *w, pt_job *j) {
	const int W = j->args.width, H = j->args.height;
	if(pt_scene_upload(w->ctx, j->flat, j->n) != PT_OK) { host_msg("frame %s: scene upload: %s\n", j->fileName, pt_last_error()); return; }
	if(w->imgW != W || w->imgH != H) {
		if(pt_image_alloc(w->ctx, W, H) != PT_OK) { host_msg("frame %s: image: %s\n", j->fileName, pt_last_error()); return; }
		w->imgW = W; w->imgH = H;
	}
	j->args.seeds = j->seeds; j->args.first_frame = 0; j->args.n_frames = j->samples;
	j->args.rank = 0; j->args.world = 1; j->args.band_rows = PT_BAND_ROWS; j->args.async = 0;
	if(j->samples == 0) { if(pt_image_alloc(w->ctx, W, H) != PT_OK) return; }
	else if(pt_render(w->ctx, &j->args) != PT_OK) { host_msg("frame %s: render: %s\n", j->fileName, pt_last_error()); return; }
	Image image;
	image.width = W; image.height = H;
	image.data = malloc(3 * (size_t)W * (size_t)H);
	if(!image.data) return;
	if(pt_image_read_u8(w->ctx, NULL, 0, 0, image.data) != PT_OK) {
		host_msg("frame %s: read-back: %s\n", j->fileName, pt_last_error());
		free(image.data);
		return;
	}
	encode_submit(w->enc, &image, j->fileName);   /* takes ownership of image.data */
}

static void *worker_main(void *arg) {
	pt_worker *w = arg;
	for(;;) {
		pthread_mutex_lock(&w->lock);
		while(!w->head && !w->quit) pthread_cond_wait(&w->wake, &w->lock);
		pt_job *j = w->head;
		if(!j) { pthread_mutex_unlock(&w->lock); return NULL; }
		w->head = j->next;
		if(!w->head) w->tail = NULL;
		pthread_mutex_unlock(&w->lock);
		job_run(w, j);
		job_free(j);
		pthread_mutex_lock(&w->lock);
		w->pending--;
		pthread_cond_broadcast(&w->room);
		pthread_mutex_unlock(&w->lock);
	}
}

static void pipeline_drain(pt_backend *b) {
	for(int i = 0; i < PT_MAX_GPUS * PT_MAX_PIPE; i++) {
		pt_worker *w = &b->worker[i];
		if(!w->started) continue;
		pthread_mutex_lock(&w->lock);
		while(w->pending > 0) pthread_cond_wait(&w->room, &w->lock);
		pthread_mutex_unlock(&w->lock);
		/* the worker may have resized its accumulator; render() re-allocates on mismatch below */
		if(i < b->n) { b->imgW[i] = w->imgW; b->imgH[i] = w->imgH; }
	}
	if(b->enc.started) {
		pthread_mutex_lock(&b->enc.lock);
		while(b->enc.pending > 0) pthread_cond_wait(&b->enc.room, &b->enc.lock);
		pthread_mutex_unlock(&b->enc.lock);
	}
}

static void pipeline_stop(pt_backend *b) {
	pipeline_drain(b);
	for(int i = 0; i < PT_MAX_GPUS * PT_MAX_PIPE; i++) {
		pt_worker *w = &b->worker[i];
		if(!w->started) continue;
		pthread_mutex_lock(&w->lock);
		w->quit = 1;
		pthread_cond_broadcast(&w->wake);
		pthread_mutex_unlock(&w->lock);
		pthread_join(w->thread, NULL);
		pthread_mutex_destroy(&w->lock); pthread_cond_destroy(&w->wake); pthread_cond_destroy(&w->room);
		w->started = 0;
	}
	if(b->enc.started) {
		pthread_mutex_lock(&b->enc.lock);
		b->enc.quit = 1;
		pthread_cond_broadcast(&b->enc.wake);
		pthread_mutex_unlock(&b->enc.lock);
		for(int i = 0; i < b->enc.started; i++) pthread_join(b->enc.thread[i], NULL);
		pthread_mutex_destroy(&b->enc.lock); pthread_cond_destroy(&b->enc.wake); pthread_cond_destroy(&b->enc.room);
		b->enc.started = 0;
	}
}

static int pipeline_submit(Renderer *renderer, int samples, const char *fileName) {
	pt_backend *b = renderer->backend;
	pt_job *j = calloc(1, sizeof *j);
	if(!j) return 0;
	if(samples < 0) samples = 0;
	j->n = renderer->sceneInfo.sphereCount;
	j->flat = flatten_scene(renderer);
	j->samples = samples;
	j->seeds = malloc(sizeof(uint64_t) * (size_t)(samples > 0 ? samples : 1));
	j->fileName = strdup(fileName);
	if(!j->flat || !j->seeds || !j->fileName) { job_free(j); return 0; }
	fill_render_args(&j->args, renderer);
	/* everything that can fail comes first; the frame's seeds are drawn last, just before it is queued: a submit that
	 * fails falls back to render(), which draws them itself, and the seed stream must not have moved twice */
	if(!b->enc.started) {
		pthread_mutex_init(&b->enc.lock, NULL); pthread_cond_init(&b->enc.wake, NULL); pthread_cond_init(&b->enc.room, NULL);
		b->enc.head = b->enc.tail = NULL; b->enc.pending = 0; b->enc.quit = 0;
		for(int i = 0; i < PT_ENCODERS; i++)
			if(pthread_create(&b->enc.thread[i], NULL, encoder_main, &b->enc) == 0) b->enc.started++;
		if(!b->enc.started) { job_free(j); return 0; }
	}
	const int g = (int)(b->frameNo++ % (b->n * b->pipeline));   /* worker g drives GPU g % n */
	pt_worker *w = &b->worker[g];
	if(!w->started) {
		if(g < b->n) { w->ctx = b->ctx[g]; w->imgW = b->imgW[g]; w->imgH = b->imgH[g]; }
		else {
			if(!b->extra[g] && pt_ctx_create(b->dev[g % b->n], &b->extra[g]) != PT_OK) {
				host_msg("frame pipeline: %s\n", pt_last_error());
				job_free(j);
				return 0;
			}
			w->ctx = b->extra[g]; w->imgW = w->imgH = 0;
		}
		w->enc = &b->enc;
		pthread_mutex_init(&w->lock, NULL); pthread_cond_init(&w->wake, NULL); pthread_cond_init(&w->room, NULL);
		w->head = w->tail = NULL; w->pending = 0; w->quit = 0;
		if(pthread_create(&w->thread, NULL, worker_main, w) != 0) { job_free(j); return 0; }
		w->started = 1;
	}
	for(int k = 0; k < samples; k++) j->seeds[k] = (uint64_t)seed_stream_next();   /* call order = frame order */
	pthread_mutex_lock(&w->lock);
	while(w->pending >= PT_PIPE_DEPTH) pthread_cond_wait(&w->room, &w->lock);
	if(w->tail) w->tail->next = j; else w->head = j;
	w->tail = j;
	w->pending++;
	pthread_cond_signal(&w->wake);
	pthread_mutex_unlock(&w->lock);
	return 1;
}

void render_to_file(Renderer *renderer, int samples, char *fileName, int verbose) {
	pt_backend *b = renderer->backend;
	if(b && b->pipeline && b->n >= 1 && !verbose && pipeline_submit(renderer, samples, fileName)) return;
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
