// pt_params.h — plain structs shared by the host side of the C ABI (pt_kernels.hip) and the two builds of
// the render kernels (pt_render_kernels.h compiled as pt_strict in pt_kernels.hip and as pt_fast in
// pt_kernels_fast.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------
// Kernel parameters (by value: land in SGPRs — the reference passes its three structs by
// value too, src/c/renderer.c:160-162)
// ---------------------------------------------------------------------------------------
#define PT_LDS_MAX_SPHERES 512       // what the lock-step kernel can hold: 27 KB of LDS per 256-thread block (PT_KERNEL=lds)
#define PT_LOCKSTEP_MAX_SPHERES 12   // what it is chosen for (pt_render) whatever the scene ...
#define PT_LOCKSTEP_MAX_CLOSED 160   // ... and up to here when the camera sits in a closed scene (long paths, no idle lanes)

struct KParams {
	int W, H, maxDepth, n;
	float bgx, bgy, bgz;
	float px, py, pz;                   // Camera.pos
	float cosRx, sinRx, cosRy, sinRy;   // cosf/sinf of Camera.rot.x / .y, folded on the host (rot.z is dead, src/cl/ray.cl:42-43)
	float rotX, rotY;                   // the angles themselves: the relaxed build evaluates the device library's sin / cos, as the reference does
	float sensorWidth, focalLength, aperture, exposure;
	int firstFrame, nFrames;
	int rank, world, bandRows;          // row-band partition (pt_hip.h)
	int localPixels;                    // pixels of this rank
	int logS;                           // LDS kernel: 2^logS lanes share a pixel (sample-interleaved)
	int logS2, tierPixels, tierThreads; // LDS kernel: pixels from tierPixels on (threads from tierThreads on) use 2^logS2 lanes
	unsigned long long trioSkipMask;    // LDS kernel: bit t = trio t of the table holds a small sphere (closest_hit_lds)
};

// Device scene tables (SoA).  geom[i] = {cx, cy, cz, r*r} for the strict build, {cx, cy, cz, r} for the relaxed one (the
// reference's code object fuses r*r into r*r - |oc|^2); matA[i] = {colr, colg, colb, type bits};
// matB[i] = {tint, factor of random_unit_vector() in the new direction (fuzz; 1 for the lambertian), refIdx,
// ((1 - refIdx) / (1 + refIdx))^2}.  Relaxed build only: matC[i] = {1 / refIdx, Schlick r0 of refIdx, Schlick r0 of
// 1 / refIdx, -}: what a dielectric hit used to compute per hit (two v_rcp_f32 and five more operations), derived ONCE per
// scene by a kernel of that build (pt_fast::derive_materials), i.e. by the same instructions on the same GPU.
struct SceneTables {
	const float4 *geom;
	const float4 *matA;
	const float4 *matB;
	const float4 *matC;
};


// Uniform grid over the small spheres of the scene (opt-in acceleration structure, no reference counterpart:
// src/cl/ray.cl:101-110 is a linear scan).  Cell (x, y, z) lists the spheres whose inflated bounding box
// overlaps it: items[cellStart[cell] .. cellStart[cell + 1]).  Spheres too large for the grid are in big[]
// and are tested for every ray.
struct GridTables {
	const int *cellStart;
	const int *items;
	const int *big;
	int nBig;
	int gx, gy, gz;
	float minx, miny, minz;
	float maxx, maxy, maxz;
	float cell, invCell;
};

// One render launch, as pt_render() hands it to a kernel build.
struct LaunchArgs {
	float4 *image;
	KParams p;
	SceneTables sc;
	const uint64_t *seedmul;
	unsigned long long *bounceCounter;
	unsigned int *queue;          // zeroed by the caller when `kernel` is the streaming kernel
	hipStream_t stream;
	bool count;                   // the COUNT variants: bounce-loop iterations, per pixel and in total
	int kernel;                   // PT_K_*
	unsigned gridPixels;          // blocks of 256 threads over localPixels << logS
	unsigned gridStream;          // one-wave blocks of the streaming kernel
	size_t ldsBytes;
	bool genericScan;             // debugging aid (PT_GENERIC_SCAN): the scan loop even where an unrolled instance exists
	GridTables grid;              // PT_K_GRID only
};
enum { PT_K_LDS = 0, PT_K_STREAM = 1, PT_K_GRID = 3 };

// bytes of LDS the lock-step kernel of a build needs for a table of n spheres (the builds' records differ in size)
namespace pt_strict { void launch_render(const LaunchArgs &la); size_t lds_bytes(int n); }
namespace pt_fast { void launch_render(const LaunchArgs &la); size_t lds_bytes(int n); void derive_materials(const float4 *matB, float4 *matC, int n, hipStream_t stream); }
