"""Parity cases shared by the golden-fixture generator, the oracle tests and the GPU tests."""
from pathtracer_amd import scenes

# name -> (scene factory, W, H, max_depth, spp)
CASES = {
    "frame0_64x36": (lambda: scenes.main_lua_frame(0), 64, 36, 50, 16),
    "frame0_256": (lambda: scenes.main_lua_frame(0), 256, 256, 50, 16),     # BASELINE config 1
    "frame77_96x54": (lambda: scenes.main_lua_frame(77), 96, 54, 50, 8),
    "frame211_80x45": (lambda: scenes.main_lua_frame(211), 80, 45, 50, 8),
    "lattice_96x54": (scenes.glass_lattice, 96, 54, 32, 8),                 # config 5 scene, small
    "dense500_96x54": (lambda: scenes.dense_random(500), 96, 54, 50, 4),    # config 3 generator, small
    "dense3000_64x36": (lambda: scenes.dense_random(3000), 64, 36, 50, 2),
    "degenerate_64x48": (scenes.degenerate, 64, 48, 50, 8),                 # ties, duplicates, tangent / concentric spheres
    "on_surface_48x36": (scenes.on_surface, 48, 36, 50, 4),                 # a hit distance of exactly +0 on every primary ray
    "dense12_48x27": (lambda: scenes.dense_random(12, seed=5), 48, 27, 50, 4),     # largest open-scene table on the lock-step kernel (PT_LOCKSTEP_MAX_SPHERES)
    "dense13_48x27": (lambda: scenes.dense_random(13, seed=5), 48, 27, 50, 4),     # first size on the streaming kernel
    "dense16_48x27": (lambda: scenes.dense_random(16, seed=3), 48, 27, 50, 4),     # five trios + one record; a quarter chunk of the streaming scan
    "dense17_48x27": (lambda: scenes.dense_random(17, seed=3), 48, 27, 50, 4),     # five trios + two records
    "dense512_48x27": (lambda: scenes.dense_random(512, seed=7), 48, 27, 50, 3),   # 4 chunks of 128, the last one full
    "dense513_48x27": (lambda: scenes.dense_random(513, seed=7), 48, 27, 50, 3),   # 5 chunks, the last one a single sphere
    "empty_33x17": (scenes.empty, 33, 17, 50, 3),                           # no spheres, ragged size
    "light_64x48": (lambda: scenes.material_micro("light"), 64, 48, 50, 8),
    "lambert_64x48": (lambda: scenes.material_micro("lambert"), 64, 48, 50, 8),
    "metal_64x48": (lambda: scenes.material_micro("metal"), 64, 48, 50, 8),
    "dielectric_64x48": (lambda: scenes.material_micro("dielectric"), 64, 48, 50, 8),
    "dielectric_low_64x48": (lambda: scenes.material_micro("dielectric_low"), 64, 48, 50, 8),
    "depth0_16x16": (lambda: scenes.main_lua_frame(0), 16, 16, 0, 2),       # maxDepth 0: loop never runs -> black
    "depth1_32x32": (lambda: scenes.main_lua_frame(3), 32, 32, 1, 4),
    # other units: hit parameters and discriminants outside the ranges of the kernels' short arithmetic (range guards,
    # reference-scan fallback), near overflow, and past it (inf / NaN everywhere)
    "frame0_scale_m45_48x27": (lambda: scenes.scaled(scenes.main_lua_frame(0), -45), 48, 27, 50, 4),
    "frame0_scale_m60_48x27": (lambda: scenes.scaled(scenes.main_lua_frame(0), -60), 48, 27, 50, 4),
    "frame0_scale_p50_48x27": (lambda: scenes.scaled(scenes.main_lua_frame(0), 50), 48, 27, 50, 4),
    "frame0_scale_p57_48x27": (lambda: scenes.scaled(scenes.main_lua_frame(0), 57), 48, 27, 50, 4),
    "micro_scale_m40_48x36": (lambda: scenes.scaled(scenes.material_micro("dielectric"), -40), 48, 36, 50, 4),
    "micro_scale_p40_48x36": (lambda: scenes.scaled(scenes.material_micro("metal"), 40), 48, 36, 50, 4),
    # directions far from unit length (a = |d|^2 outside the scan's range, and overflowing), an absurd refraction index
    "metal_fuzz25_48x36": (lambda: scenes.material_micro("metal_fuzz25"), 48, 36, 50, 4),
    "metal_fuzz70_48x36": (lambda: scenes.material_micro("metal_fuzz70"), 48, 36, 50, 4),
    "dielectric_idx30_48x36": (lambda: scenes.material_micro("dielectric_idx30"), 48, 36, 50, 4),
    # material codes outside 0..3: no branch of ray.cl:130-171 matches, no RNG draw, black after maxDepth re-hits
    "unknown7_48x36": (lambda: scenes.material_micro("unknown7"), 48, 36, 50, 4),
    "unknown_neg_48x36": (lambda: scenes.material_micro("unknown_neg"), 48, 36, 50, 4),
    "one_pixel": (lambda: scenes.main_lua_frame(0), 1, 1, 50, 4),           # id 0: RNG stuck at 0
}
