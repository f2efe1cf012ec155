/* cuda_fp16.h -- see cuda.h in this directory (test-only host emulation).  `half`: binary16 storage; every intrinsic is one correctly
 * rounded operation.  Sums, differences and products of two halves are exact in double, so one rounding of the double result is the
 * correctly rounded half result; a quotient of two 11-bit significands rounded to double and then to half is not a double-rounding
 * case either (53 >= 2 * 11 + 2); the fused multiply-add is exact integer arithmetic (orc_hfma). */
#ifndef TCE_ORACLE_CUDA_FP16_EMUL_H
#define TCE_ORACLE_CUDA_FP16_EMUL_H
#include "cuda.h"
extern "C" {
uint16_t orc_f64_to_f16(double v);
float orc_f16_to_f32(uint16_t h);
uint16_t orc_hfma(uint16_t a, uint16_t b, uint16_t c);
}
struct half {
    uint16_t x;
    half() = default;
    half(float f) : x(orc_f64_to_f16((double)f)) {}
    half(double f) : x(orc_f64_to_f16(f)) {}
    half(int i) : x(orc_f64_to_f16((double)i)) {}
    operator float() const { return orc_f16_to_f32(x); }
};
typedef half __half;
struct half2 {
    half x, y;
};
static inline half tce_emul_bits(uint16_t b) {
    half h;
    h.x = b;
    return h;
}
static inline float __half2float(half a) { return orc_f16_to_f32(a.x); }
static inline half __float2half(float f) { return half(f); }
static inline half __hadd(half a, half b) { return half((double)__half2float(a) + (double)__half2float(b)); }
static inline half __hsub(half a, half b) { return half((double)__half2float(a) - (double)__half2float(b)); }
static inline half __hmul(half a, half b) { return half((double)__half2float(a) * (double)__half2float(b)); }
static inline half __hdiv(half a, half b) { return half((double)__half2float(a) / (double)__half2float(b)); }
static inline half __hneg(half a) { return tce_emul_bits((uint16_t)(a.x ^ 0x8000u)); }
static inline half __hfma(half a, half b, half c) { return tce_emul_bits(orc_hfma(a.x, b.x, c.x)); }
static inline bool __hgt(half a, half b) { return __half2float(a) > __half2float(b); }
static inline bool __hlt(half a, half b) { return __half2float(a) < __half2float(b); }
static inline half __hmax(half a, half b) { return __hgt(a, b) ? a : b; }
static inline half2 __hadd2(half2 a, half2 b) { return half2{__hadd(a.x, b.x), __hadd(a.y, b.y)}; }
static inline half hexp(half a) { return half(expf(__half2float(a))); }
template <typename T> static inline T __ldg(const T *p) { return *p; }
#endif
