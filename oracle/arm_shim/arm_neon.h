// oracle/arm_shim/arm_neon.h -- TEST INFRASTRUCTURE ONLY.  Just enough NEON names for the reference's kernels/matmul.h to PARSE under -DQM_ARM on an x86 host
// (its static inline my_vdotq_s32 helper, kernels/matmul.h:30-49, is never called by the one function the oracle pins: the QM_ARM branch of
// naive_mat_mul_int4, kernels/matmul_int4.cc:50-76, which is plain scalar C++).  Declarations only: nothing here is ever linked or executed.
#pragma once
#include <stdint.h>
struct int8x8_t { int8_t v[8]; };
struct int8x16_t { int8_t v[16]; };
struct int16x4_t { int16_t v[4]; };
struct int16x8_t { int16_t v[8]; };
struct int32x4_t { int32_t v[4]; };
int8x8_t vget_low_s8(int8x16_t);
int8x8_t vget_high_s8(int8x16_t);
int16x8_t vmull_s8(int8x8_t, int8x8_t);
int16x4_t vget_low_s16(int16x8_t);
int16x4_t vget_high_s16(int16x8_t);
int32x4_t vaddl_s16(int16x4_t, int16x4_t);
int32x4_t vaddq_s32(int32x4_t, int32x4_t);
int32x4_t vdotq_s32(int32x4_t, int8x16_t, int8x16_t);
