/* scene_driver.c — issues, in C, the renderer-API calls that the reference's scene script
 * (scripts/main.lua:1-103) issues for animation frames first..first+count-1, so the drop-in
 * boundary can be exercised where Lua is not installed (it is not in the build image).
 *
 *   pt_scene_driver [first=0] [count=1] [width=640] [height=360] [samples=2000] [depth=50] [pattern=clip/frame%d.png]
 *
 * Lua numbers are doubles narrowed to float at the binding (src/c/lua_wrapper.c:166-188); the
 * (float) casts below do the same narrowing.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>

#include "renderer.h"

static double now_sec(void) {
	struct timeval tv;
	gettimeofday(&tv, NULL);
	return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

int main(int argc, char **argv) {
	int first = argc > 1 ? atoi(argv[1]) : 0;
	int count = argc > 2 ? atoi(argv[2]) : 1;
	int width = argc > 3 ? atoi(argv[3]) : 640;
	int height = argc > 4 ? atoi(argv[4]) : 360;
	int samples = argc > 5 ? atoi(argv[5]) : 2000;
	int maxDepth = argc > 6 ? atoi(argv[6]) : 50;
	const char *pattern = argc > 7 ? argv[7] : "clip/frame%d.png";
	const double wallRadius = 10000, frames = 300, pi = 3.14159265358979323846;

	const double t0 = now_sec();
	Renderer *r = create_renderer();
	const double t1 = now_sec();
	set_image_properties(r, width, height, maxDepth);
	set_background_color(r, 0, 0, 0);

	Material lightSource = create_light_source_material(10, 10, 10);
	Material white = create_lambertian_material(1, 1, 1);
	Material red = create_lambertian_material(1, 0, 0);
	Material green = create_lambertian_material(0, 1, 0);
	Material mirror = create_metal_material(0, 0, 0, 0, 0);
	Material greenGlass = create_dielectric_material(0, 1, 0, 0.8f, 0, 0.95f);

	for(int i = first; i < first + count; i++) {
		printf("frame %d\n", i);
		clear_scene(r);
		add_sphere(r, 0, (float)(wallRadius + 6), -10, (float)wallRadius, white);
		add_sphere(r, 0, (float)-(wallRadius + 6), -10, (float)wallRadius, white);
		add_sphere(r, 0, 0, (float)-(wallRadius + 20), (float)wallRadius, white);
		add_sphere(r, 0, 0, (float)(wallRadius + 20), (float)wallRadius, white);
		add_sphere(r, (float)(wallRadius + 6), 0, -5, (float)wallRadius, green);
		add_sphere(r, (float)-(wallRadius + 6), 0, -5, (float)wallRadius, red);
		add_sphere(r, -1, 0, -10, 1, greenGlass);

		const double cx = -1, cy = 0, cz = -10, ph = 2 * pi * (i / frames);
		add_sphere(r, (float)(cx + 2 * cos(ph)), (float)(cy + sin(ph)), (float)(cz + 2 * sin(ph)), 0.5f, lightSource);
		const double ph2 = -4 * pi * (i / frames);
		add_sphere(r, (float)(cx + 3.5 * cos(ph2)), (float)(cy + 0), (float)(cz + 3.5 * sin(ph2)), 1, mirror);

		const double startOffset = 3 * pi / 4;
		set_camera_properties(r, (float)(cx + 0), (float)(cy + 5 * cos(ph + startOffset)), (float)(cz + 5 * sin(ph + startOffset)),
		                      (float)(ph - pi / 2 + startOffset), 0, 0, 15, 20, 0.001f, 1000);

		char name[512];
		snprintf(name, sizeof name, pattern, i);
		render_to_file(r, samples, name, 0);
	}
	const double t2 = now_sec();
	destroy_renderer(r);
	const double t3 = now_sec();
	if(getenv("PT_DRIVER_TIMING"))
		fprintf(stderr, "create_renderer %.3f s, %d frames queued/rendered %.3f s, destroy_renderer (drains the pipeline) %.3f s\n",
		        t1 - t0, count, t2 - t1, t3 - t2);
	return 0;
}
