// Instruction-throughput microbenchmark for gfx950 (debug tool, not product code).
// Each kernel runs REPS x 16 independent instances of one instruction per wave; grid fills the chip
// with 8 waves per SIMD.  Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REPS 4096
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline float tof(float x) { return x; }
__device__ inline float tof(double x) { return (float)x; }
__device__ inline float tof(v2f x) { return x.x + x.y; }

#define KERNEL(name, decl, body)                                                    \
	__global__ __launch_bounds__(256) void name(float *out, int reps) {             \
		decl;                                                                       \
		for(int r = 0; r < reps; r++) {                                             \
			asm volatile(body body body body body body body body                    \
			             body body body body body body body body                    \
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "s"(0x41a7)); \
		}                                                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = tof(a0) + tof(a1) + tof(a2) + tof(a3);      \
	}

#define F32DECL float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.0001f, b1 = 0.9999f
#define F64DECL double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.0001, b1 = 0.9999
#define PKDECL v2f a0 = {1.0f * threadIdx.x, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, b0 = {1.0001f, 1.0f}, b1 = {0.9999f, 1.0f}
#define MADDECL double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; unsigned b0 = threadIdx.x * 977u, b1 = b0 + 12345u

KERNEL(k_mul_f32, F32DECL, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5\n")
KERNEL(k_fma_f32, F32DECL, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %5, %4\n")
KERNEL(k_sqrt_f32, F32DECL, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
KERNEL(k_rcp_f32, F32DECL, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_cndmask, F32DECL, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n")
KERNEL(k_cmp_f32, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %5\n")
KERNEL(k_cmp_sgpr, F32DECL, "v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %5\n v_cmp_lt_f32 s[24:25], %2, %4\n v_cmp_lt_f32 s[26:27], %3, %5\n")
KERNEL(k_min_f32, F32DECL, "v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %5\n")
KERNEL(k_add_u32, F32DECL, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5\n")
KERNEL(k_mul_lo_u32, F32DECL, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %5\n")
KERNEL(k_mul_hi_u32, F32DECL, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %5\n")
KERNEL(k_cvt_f32_u32, F32DECL, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL(k_mul_f64, F64DECL, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %5\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %5\n")
KERNEL(k_fma_f64, F64DECL, "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %5, %4\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %5, %4\n")
KERNEL(k_mad_u64_u32, MADDECL, "v_mad_u64_u32 %0, vcc, %4, %6, 0\n v_mad_u64_u32 %1, vcc, %5, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %6, 0\n v_mad_u64_u32 %3, vcc, %5, %6, 0\n")
KERNEL(k_cvt_f64_u32, MADDECL, "v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %5\n")
#define CVTDECL float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; double b0 = 1.0001 * threadIdx.x, b1 = 0.9999
KERNEL(k_cvt_f32_f64, CVTDECL, "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %4\n v_cvt_f32_f64 %3, %5\n")
KERNEL(k_pk_mul_f32, PKDECL, "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %5\n")
KERNEL(k_pk_add_f32, PKDECL, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n")
KERNEL(k_pk_fma_f32, PKDECL, "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %5, %4\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %5, %4\n")
KERNEL(k_readlane, F32DECL, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n")
KERNEL(k_salu, F32DECL, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")


KERNEL(k_cndmask_nodep, F32DECL, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n")
KERNEL(k_cndmask_sgpr, F32DECL, "v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %5, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[22:23]\n v_cndmask_b32 %3, %3, %5, s[22:23]\n")
KERNEL(k_cmp_cnd_pair, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc\n")
KERNEL(k_add_f32, F32DECL, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(k_max_f32, F32DECL, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %5\n")
KERNEL(k_and_b32, F32DECL, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %5\n")
KERNEL(k_mov_b32, F32DECL, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n")
KERNEL(k_lshl_add, F32DECL, "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %5\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %5\n")
KERNEL(k_min_u32, F32DECL, "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %5\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %5\n")
KERNEL(k_min3_f32, F32DECL, "v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %5, %4\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %5, %4\n")
KERNEL(k_med3_f32, F32DECL, "v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %5, %4\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %5, %4\n")
KERNEL(k_mul_legacy, F32DECL, "v_mul_legacy_f32 %0, %0, %4\n v_mul_legacy_f32 %1, %1, %5\n v_mul_legacy_f32 %2, %2, %4\n v_mul_legacy_f32 %3, %3, %5\n")
KERNEL(k_rsq_f32, F32DECL, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n")
KERNEL(k_div_scale, F32DECL, "v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %5, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %5, %3\n")
KERNEL(k_div_fmas, F32DECL, "v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %5, %4\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %5, %4\n")
KERNEL(k_div_fixup, F32DECL, "v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %5, %4\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %5, %4\n")
KERNEL(k_cmp_class, F32DECL, "v_cmp_class_f32 vcc, %0, %4\n v_cmp_class_f32 vcc, %1, %5\n v_cmp_class_f32 vcc, %2, %4\n v_cmp_class_f32 vcc, %3, %5\n")
KERNEL(k_mul_sgpr, F32DECL, "v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3\n")
KERNEL(k_fma_sgpr, F32DECL, "v_fma_f32 %0, s20, %0, %4\n v_fma_f32 %1, s21, %1, %5\n v_fma_f32 %2, s22, %2, %4\n v_fma_f32 %3, s23, %3, %5\n")
KERNEL(k_sub_sgpr, F32DECL, "v_subrev_f32 %0, s20, %0\n v_subrev_f32 %1, s21, %1\n v_subrev_f32 %2, s22, %2\n v_subrev_f32 %3, s23, %3\n")

KERNEL(k_ashr_i32, F32DECL, "v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3\n ")
KERNEL(k_lshr_b32, F32DECL, "v_lshrrev_b32 %0, 31, %0\n v_lshrrev_b32 %1, 31, %1\n v_lshrrev_b32 %2, 31, %2\n v_lshrrev_b32 %3, 31, %3\n ")
KERNEL(k_lshl_b32, F32DECL, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n ")
KERNEL(k_sub_u32, F32DECL, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %5\n ")
KERNEL(k_xor_b32, F32DECL, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %5\n ")
KERNEL(k_or_b32, F32DECL, "v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %5\n ")
KERNEL(k_bfe_u32, F32DECL, "v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5\n ")
KERNEL(k_add3_u32, F32DECL, "v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %5, %5\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %5, %5\n ")
KERNEL(k_and_or_b32, F32DECL, "v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %5, %5\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %5, %5\n ")
KERNEL(k_fmac_f32, F32DECL, "v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %5, %5\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %5, %5\n ")
KERNEL(k_add_f32_neg, F32DECL, "v_add_f32 %0, -%0, %4\n v_add_f32 %1, -%1, %5\n v_add_f32 %2, -%2, %4\n v_add_f32 %3, -%3, %5\n ")
KERNEL(k_sub_f32, F32DECL, "v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %5\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %5\n ")
KERNEL(k_add_f32_abs, F32DECL, "v_add_f32 %0, |%0|, %4\n v_add_f32 %1, |%1|, %5\n v_add_f32 %2, |%2|, %4\n v_add_f32 %3, |%3|, %5\n ")
KERNEL(k_addc_co, F32DECL, "v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n ")
KERNEL(k_subb_co, F32DECL, "v_subb_co_u32 %0, vcc, %0, %4, vcc\n v_subb_co_u32 %1, vcc, %1, %5, vcc\n v_subb_co_u32 %2, vcc, %2, %4, vcc\n v_subb_co_u32 %3, vcc, %3, %5, vcc\n ")
KERNEL(k_add_co, F32DECL, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %5\n ")
KERNEL(k_max_u32, F32DECL, "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %5\n ")
KERNEL(k_med3_u32, F32DECL, "v_med3_u32 %0, %0, %4, %4\n v_med3_u32 %1, %1, %5, %5\n v_med3_u32 %2, %2, %4, %4\n v_med3_u32 %3, %3, %5, %5\n ")
KERNEL(k_min_i32, F32DECL, "v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %5\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %5\n ")
KERNEL(k_bfi_b32, F32DECL, "v_bfi_b32 %0, %0, %4, %4\n v_bfi_b32 %1, %1, %5, %5\n v_bfi_b32 %2, %2, %4, %4\n v_bfi_b32 %3, %3, %5, %5\n ")
KERNEL(k_perm_b32, F32DECL, "v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %5, %5\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %5, %5\n ")
KERNEL(k_mul_u32_u24, F32DECL, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %5\n ")
KERNEL(k_mad_u32_u24, F32DECL, "v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %5, %5\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %5, %5\n ")
KERNEL(k_xad_u32, F32DECL, "v_xad_u32 %0, %0, %4, %4\n v_xad_u32 %1, %1, %5, %5\n v_xad_u32 %2, %2, %4, %4\n v_xad_u32 %3, %3, %5, %5\n ")
KERNEL(k_cmp_lt_u32, F32DECL, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %5\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %5\n ")
KERNEL(k_sqrt_add_mix, F32DECL, "v_sqrt_f32 %0, %0\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_sqrt_f32 %1, %1\n v_add_f32 %1, %1, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %1, %1, %5\n v_sqrt_f32 %2, %2\n v_add_f32 %2, %2, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %2, %2, %4\n v_sqrt_f32 %3, %3\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %5\n ")
KERNEL(k_cmp_add_mix, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_add_f32 %0, %0, %4\n v_cmp_lt_f32 vcc, %1, %5\n v_add_f32 %1, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_add_f32 %2, %2, %4\n v_cmp_lt_f32 vcc, %3, %5\n v_add_f32 %3, %3, %5\n ")
KERNEL(k_ldexp_f32, F32DECL, "v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %5\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %5\n ")
KERNEL(k_frexp_mant, F32DECL, "v_frexp_mant_f32 %0, %0\n v_frexp_mant_f32 %1, %1\n v_frexp_mant_f32 %2, %2\n v_frexp_mant_f32 %3, %3\n ")

KERNEL(m_cmp_add3, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_cmp2_add2, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_add_f32 %1, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_cmp3_add1, F32DECL, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_sqrt_add3, F32DECL, "v_sqrt_f32 %0, %0\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_sqrt_cmp_add2, F32DECL, "v_sqrt_f32 %0, %0\n v_cmp_lt_f32 vcc, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_sqrt_cmp3, F32DECL, "v_sqrt_f32 %0, %0\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %5\n")
KERNEL(m_min_cnd_mul_fma, F32DECL, "v_min_f32 %0, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_mul_f32 %2, %2, %4\n v_fma_f32 %3, %3, %5, %4\n")
KERNEL(m_fma3_cmp, F32DECL, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %4, %5\n v_cmp_lt_f32 vcc, %3, %5\n")
KERNEL(m_min_add3, F32DECL, "v_min_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_cnd_add3, F32DECL, "v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %1, %1, %5\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n")
KERNEL(m_add_mul, F32DECL, "v_add_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_mul_f32 %3, %3, %5\n")


// mixes of packed-f32 and scalar-f32 instructions (do the two pipes overlap?)
#define MIXKERNEL(name, body)                                                       \
	__global__ __launch_bounds__(256) void name(float *out, int reps) {             \
		v2f p0 = {1.0f * threadIdx.x, 2.0f}, p1 = p0 + 1.0f, q0 = {1.0001f, 1.0f}, q1 = {0.9999f, 1.0f};   \
		float a0 = threadIdx.x, a1 = a0 + 1, b0 = 1.0001f, b1 = 0.9999f;             \
		for(int r = 0; r < reps; r++) {                                             \
			asm volatile(body body body body body body body body                    \
			             body body body body body body body body                    \
			             : "+v"(p0), "+v"(p1), "+v"(a0), "+v"(a1) : "v"(q0), "v"(q1), "v"(b0), "v"(b1)); \
		}                                                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = tof(p0) + tof(p1) + a0 + a1;   \
	}
MIXKERNEL(x_pk2_add2, "v_pk_mul_f32 %0, %0, %4\n v_add_f32 %2, %2, %6\n v_pk_mul_f32 %1, %1, %5\n v_add_f32 %3, %3, %7\n")
MIXKERNEL(x_pk1_add3, "v_pk_mul_f32 %0, %0, %4\n v_add_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n v_add_f32 %2, %2, %7\n")
MIXKERNEL(x_pk3_add1, "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %0, %0, %5\n v_add_f32 %3, %3, %7\n")
MIXKERNEL(x_pk2_cmp2, "v_pk_mul_f32 %0, %0, %4\n v_cmp_lt_f32 vcc, %2, %6\n v_pk_mul_f32 %1, %1, %5\n v_cmp_lt_f32 vcc, %3, %7\n")
MIXKERNEL(x_pkfma2_add2, "v_pk_fma_f32 %0, %0, %4, %5\n v_add_f32 %2, %2, %6\n v_pk_fma_f32 %1, %1, %5, %4\n v_add_f32 %3, %3, %7\n")
MIXKERNEL(x_sqrt_pk3, "v_sqrt_f32 %2, %2\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %0, %0, %5\n")


// LDS read throughput: wave-uniform (broadcast) and per-lane ds_read_b128 / b64 / b32, 64 reads per loop trip
#define LDSKERNEL(name, INS, ADDR)                                                  \
	__global__ __launch_bounds__(256) void name(float *out, int reps) {             \
		__shared__ float4 buf[1024];                                                \
		for(int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4(i, 1, 2, 3); \
		__syncthreads();                                                            \
		unsigned addr = (unsigned)(size_t)buf + (ADDR);                             \
		float acc = 0;                                                              \
		for(int r = 0; r < reps; r++) {                                             \
			float4 v0, v1, v2, v3;                                                  \
			for(int k = 0; k < 16; k++) {                                           \
				asm volatile(INS " %0, %4\n" INS " %1, %4 offset:16\n" INS " %2, %4 offset:32\n" INS " %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n" \
				             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr));  \
				acc += v0.x + v1.x + v2.x + v3.x;                                   \
			}                                                                       \
		}                                                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                           \
	}
LDSKERNEL(l_b128_uniform, "ds_read_b128", 0)
LDSKERNEL(l_b128_perlane, "ds_read_b128", (threadIdx.x & 63) * 64)

KERNEL(t_sqrt_add7, F32DECL, "v_sqrt_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n ")
KERNEL(t_sqrt_add15, F32DECL, "v_sqrt_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n ")
KERNEL(t_sqrt_add31, F32DECL, "v_sqrt_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %1, %1, %4\n ")
KERNEL(t_sqrt3_add13, F32DECL, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %3, %3, %4\n ")
KERNEL(t_sqrt2_add30, F32DECL, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %5\n ")

typedef void (*kfn)(float *, int);
struct Entry { const char *name; kfn fn; };
#define E(n) {#n, n}

int main(int argc, char **argv) {
	Entry es[] = {E(k_mul_f32), E(k_fma_f32), E(k_sqrt_f32), E(k_rcp_f32), E(k_cndmask), E(k_cmp_f32), E(k_cmp_sgpr), E(k_min_f32),
	              E(k_add_u32), E(k_mul_lo_u32), E(k_mul_hi_u32), E(k_cvt_f32_u32), E(k_mul_f64), E(k_fma_f64), E(k_mad_u64_u32),
	              E(k_cvt_f64_u32), E(k_cvt_f32_f64), E(k_pk_mul_f32), E(k_pk_add_f32), E(k_pk_fma_f32), E(k_readlane), E(k_salu), E(k_cndmask_nodep), E(k_cndmask_sgpr), E(k_cmp_cnd_pair), E(k_add_f32), E(k_max_f32), E(k_and_b32), E(k_mov_b32), E(k_lshl_add), E(k_min_u32), E(k_min3_f32), E(k_med3_f32), E(k_mul_legacy), E(k_rsq_f32), E(k_div_scale), E(k_div_fmas), E(k_div_fixup), E(k_cmp_class), E(k_mul_sgpr), E(k_fma_sgpr), E(k_sub_sgpr), E(t_sqrt_add7), E(t_sqrt_add15), E(t_sqrt_add31), E(t_sqrt3_add13), E(t_sqrt2_add30), E(l_b128_uniform), E(l_b128_perlane), E(x_pk2_add2), E(x_pk1_add3), E(x_pk3_add1), E(x_pk2_cmp2), E(x_pkfma2_add2), E(x_sqrt_pk3), E(m_cmp_add3), E(m_cmp2_add2), E(m_cmp3_add1), E(m_sqrt_add3), E(m_sqrt_cmp_add2), E(m_sqrt_cmp3), E(m_min_cnd_mul_fma), E(m_fma3_cmp), E(m_min_add3), E(m_cnd_add3), E(m_add_mul), E(k_ashr_i32), E(k_lshr_b32), E(k_lshl_b32), E(k_sub_u32), E(k_xor_b32), E(k_or_b32), E(k_bfe_u32), E(k_add3_u32), E(k_and_or_b32), E(k_fmac_f32), E(k_add_f32_neg), E(k_sub_f32), E(k_add_f32_abs), E(k_addc_co), E(k_subb_co), E(k_add_co), E(k_max_u32), E(k_med3_u32), E(k_min_i32), E(k_bfi_b32), E(k_perm_b32), E(k_mul_u32_u24), E(k_mad_u32_u24), E(k_xad_u32), E(k_cmp_lt_u32), E(k_sqrt_add_mix), E(k_cmp_add_mix), E(k_ldexp_f32), E(k_frexp_mant)};
	int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 8;
	int blocks = 256 * wavesPerSimd;   // 256 CUs x (wavesPerSimd*4 waves / 4 waves per block)
	float *out;
	hipMalloc(&out, sizeof(float) * 256 * blocks);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	printf("waves/SIMD %d, blocks %d\n", wavesPerSimd, blocks);
	for(auto &e : es) {
		e.fn<<<blocks, 256>>>(out, 64);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		e.fn<<<blocks, 256>>>(out, REPS);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		double inst_per_simd = (double)REPS * 64 * wavesPerSimd;   // 64 instr per loop iteration
		double ns_per_inst = ms * 1e6 / inst_per_simd;
		printf("%-16s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cyc at 2.4 GHz)\n", e.name, ms, ns_per_inst, ns_per_inst * 2.4);
	}
	return 0;
}
