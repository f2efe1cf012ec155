/*
 * tce_oracle.c -- CPU ORACLE for the quantized-matmul hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the arithmetic of TinyChatEngine's
 * kernels/matmul.h surface (reference @ 2024_08_07).  It is the checker the GPU
 * parity tests compare against.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product path
 * (tinychatengine_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity pin: every function here is checked bit-for-bit (fp32 / int8 paths) or
 * exactly (software-fp16 path) against the reference's own sources compiled
 * from /root/reference into oracle/_ref/libtce_ref.so (see oracle/Makefile and
 * tests/test_oracle_vs_ref.py) and against the committed vectors in
 * tests/golden/ that were produced by that library (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (NO -Ofast, NO -mfma: the
 * int8 epilogue must round the multiply and the add separately -- SURVEY App. B).
 *
 * Each function cites the reference file:line it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* IEEE binary16 helpers (round-to-nearest-even), independent of any library */
/* ------------------------------------------------------------------------- */

/* double -> binary16 with a single correct RNE rounding (no double rounding). */
ORC_API uint16_t orc_f64_to_f16(double v) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    uint16_t sign = (uint16_t)((bits >> 48) & 0x8000u);
    int64_t exp = (int64_t)((bits >> 52) & 0x7FF);
    uint64_t man = bits & 0xFFFFFFFFFFFFFull;
    if (exp == 0x7FF) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    }
    if (exp == 0 && man == 0) return sign;
    int64_t e = exp - 1023; /* unbiased */
    if (e > 15) return (uint16_t)(sign | 0x7C00u);
    /* significand with hidden bit, 53 bits */
    uint64_t sig = (exp ? (1ull << 52) : 0) | man;
    int shift; /* how many low bits of sig are dropped */
    int64_t he; /* half biased exponent field */
    if (e >= -14) {
        shift = 42; /* keep 11 bits (1 hidden + 10) */
        he = e + 15;
    } else {
        /* subnormal half: value = m * 2^-24, m in [0,1023] */
        shift = (int)(42 + (-14 - e));
        he = 0;
        if (shift > 63) return sign; /* underflow to zero (|v| < 2^-35) */
    }
    uint64_t kept = sig >> shift;
    uint64_t rem = sig & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (kept & 1))) kept++;
    uint32_t out;
    if (he == 0) {
        out = (uint32_t)kept; /* may carry into exponent field = smallest normal, correct */
    } else {
        out = (uint32_t)(((uint64_t)he << 10) + (kept - 1024)); /* carry propagates into exponent */
    }
    if (out >= 0x7C00u) out = 0x7C00u;
    return (uint16_t)(sign | out);
}

ORC_API uint16_t orc_f32_to_f16(float v) { return orc_f64_to_f16((double)v); }

ORC_API float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            man &= 0x3FF;
            out = sign | (uint32_t)((127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        out = sign | 0x7F800000u | (man << 13);
    } else {
        out = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &out, 4);
    return f;
}

ORC_API void orc_f32_to_f16_array(const float *src, uint16_t *dst, int64_t n) {
    for (int64_t i = 0; i < n; i++) dst[i] = orc_f32_to_f16(src[i]);
}
ORC_API void orc_f16_to_f32_array(const uint16_t *src, float *dst, int64_t n) {
    for (int64_t i = 0; i < n; i++) dst[i] = orc_f16_to_f32(src[i]);
}

/* ------------------------------------------------------------------------- */
/* Weight-format spec: group quantizer + the three packings used on the path  */
/* ------------------------------------------------------------------------- */

/* llm/tools/quantize_methods.py:9-21 (python) == llm/src/nn_modules/cuda/utils.cu:162-178 */
ORC_API int orc_zeros_width(int in_features, int group_size) {
    int mult;
    if (group_size >= 128) mult = 1;
    else if (group_size == 64) mult = 2;
    else if (group_size == 32) mult = 4;
    else return -1;
    int w = (in_features / group_size + 7) / 8;
    return ((w + mult - 1) / mult) * mult;
}

/*
 * Group quantizer shared by quantize_row_q4_5 / q4_6 (quantize_methods.py:323-339,
 * 392-411): per group of G consecutive values along IC (the flat array is
 * reshaped (nb, G)), d = x[argmax|x|] / -8 (signed), id = 1/d (0 if d == 0),
 * code = trunc(clip(x*id + 8.5, 0, 15)).  All arithmetic in float32 like numpy.
 * codes: uint8 [N*K] (one code per byte), d_out: float32 [N*K/G].
 */
ORC_API void orc_group_quantize(const float *w, int64_t n_elems, int G, uint8_t *codes, float *d_out) {
    int64_t nb = n_elems / G;
    for (int64_t b = 0; b < nb; b++) {
        const float *x = w + b * G;
        int imax = 0;
        float amax = fabsf(x[0]);
        for (int i = 1; i < G; i++) { /* np.argmax: first maximal element */
            float a = fabsf(x[i]);
            if (a > amax) { amax = a; imax = i; }
        }
        float d = x[imax] / -8.0f;
        float id = (d == 0.0f) ? 0.0f : 1.0f / d;
        d_out[b] = d;
        for (int i = 0; i < G; i++) {
            float t = x[i] * id;
            t = t + 8.5f;
            if (t < 0.0f) t = 0.0f;
            if (t > 15.0f) t = 15.0f;
            codes[b * G + i] = (uint8_t)(int32_t)t; /* astype(int32): truncation */
        }
    }
}

/*
 * q4_6 packing = what Linear_half_int4 loads (quantize_methods.py:413-440,
 * llm/include/ops/linear.h:188-210).
 *   qweight  uint32 [N][K/8]   nibble i of word j = code[n][8j+i]
 *   scales   fp16   [N][zw*8]  first K/G valid, rest 0
 *   zeros    uint32 [N][zw]    every nibble = 8
 */
ORC_API void orc_pack_q4_6(const uint8_t *codes, const float *d, int N, int K, int G, uint32_t *qweight,
                           uint16_t *scales, uint32_t *zeros) {
    int zw = orc_zeros_width(K, G);
    int sfw = zw * 8;
    int ng = K / G;
    for (int n = 0; n < N; n++) {
        for (int j = 0; j < K / 8; j++) {
            uint32_t word = 0;
            for (int i = 0; i < 8; i++) word |= (uint32_t)(codes[(int64_t)n * K + 8 * j + i] & 0xF) << (4 * i);
            qweight[(int64_t)n * (K / 8) + j] = word;
        }
        for (int g = 0; g < sfw; g++)
            scales[(int64_t)n * sfw + g] = (g < ng) ? orc_f32_to_f16(d[(int64_t)n * ng + g]) : 0;
        for (int z = 0; z < zw; z++) zeros[(int64_t)n * zw + z] = 0x88888888u;
    }
}

/*
 * q4_5 packing = AWQ GEMM layout consumed by naive_mat_mul_fp16_int4
 * (quantize_methods.py:341-366, kernels/cuda/matmul_int4.cu:19-39).
 *   qweight uint32 [K][N/8]  nibbles 0..7 of word j hold n = 8j + {0,2,4,6,1,3,5,7}
 *   scales  fp16   [K/G][N]
 *   zeros   uint32 [K/G][N/8] = 0x88888888
 */
ORC_API void orc_pack_q4_5(const uint8_t *codes, const float *d, int N, int K, int G, uint32_t *qweight,
                           uint16_t *scales, uint32_t *zeros) {
    static const int order[8] = {0, 2, 4, 6, 1, 3, 5, 7};
    int ng = K / G;
    for (int k = 0; k < K; k++)
        for (int j = 0; j < N / 8; j++) {
            uint32_t word = 0;
            for (int i = 0; i < 8; i++)
                word |= (uint32_t)(codes[(int64_t)(8 * j + order[i]) * K + k] & 0xF) << (4 * i);
            qweight[(int64_t)k * (N / 8) + j] = word;
        }
    for (int g = 0; g < ng; g++)
        for (int n = 0; n < N; n++) scales[(int64_t)g * N + n] = orc_f32_to_f16(d[(int64_t)n * ng + g]);
    for (int64_t i = 0; i < (int64_t)ng * (N / 8); i++) zeros[i] = 0x88888888u;
}

/* sequential packing (q4_0 style, quantize_methods.py:28-76 / kernels/matmul_int4.cc:116-120):
 *   uint8 [N][K/2], byte b of a row = code[2b] | code[2b+1] << 4 */
ORC_API void orc_pack_sequential(const uint8_t *codes, int N, int K, uint8_t *packed) {
    for (int64_t i = 0; i < (int64_t)N * K / 2; i++) packed[i] = (uint8_t)((codes[2 * i] & 0xF) | (codes[2 * i + 1] << 4));
}

/* inverse of the q4_6 weight packing: uint32 [N][K/8] -> one code per byte [N][K] */
ORC_API void orc_unpack_q4_6(const uint32_t *qweight, int N, int K, uint8_t *codes) {
    for (int64_t w = 0; w < (int64_t)N * (K / 8); w++) {
        uint32_t word = qweight[w];
        for (int i = 0; i < 8; i++) codes[w * 8 + i] = (uint8_t)((word >> (4 * i)) & 0xF);
    }
}

/* ------------------------------------------------------------------------- */
/* W4 paths                                                                   */
/* ------------------------------------------------------------------------- */

/*
 * MatmulOperator::naive_mat_mul_int4, generic (#else) branch --
 * kernels/matmul_int4.cc:105-127.  A f32 [M][K]; B uint8 [N][K/2] sequential
 * nibbles; scales f32 [N][K/G]; one float zero point; strictly sequential fp32
 * accumulation over k; (q - z) * s is rounded once, then a*deq, then the add.
 */
ORC_API void orc_naive_mat_mul_int4(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales,
                                    float zero_point, float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += G) {
                float s = scales[((int64_t)j * K + k) / G];
                const float *a = A + (int64_t)i * K + k;
                const uint8_t *b = B + (int64_t)j * (K / 2) + k / 2;
                for (int q = 0; q < G / 2; q++) {
                    uint8_t p = b[q];
                    float d0 = ((float)(p & 0x0F) - zero_point) * s;
                    float d1 = ((float)(p >> 4) - zero_point) * s;
                    float t0 = a[2 * q] * d0;
                    acc = acc + t0;
                    float t1 = a[2 * q + 1] * d1;
                    acc = acc + t1;
                }
            }
            C[(int64_t)i * N + j] = acc;
        }
}

/*
 * MatmulOperator::naive_mat_mul_int4, QM_x86 branch -- kernels/matmul_int4.cc:78-104.  Weights in the x86 interleave
 * (quantize_row_q4_3, llm/tools/quantize_methods.py:232-240): per 64 weights, byte e holds code[e] in the low and
 * code[32+e] in the high nibble; zero point hard-coded 8; the accumulation alternates x[e]*w[e], x[32+e]*w[32+e].
 * Only meaningful for block_size == 32 (the branch steps k by 2*block_size but consumes block_size bytes).
 * This is the reference the AVX W4A8 fast path is compared with in the reference's own test (test_ops.cc:648-653).
 */
ORC_API int orc_naive_mat_mul_int4_x86(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales,
                                       float *C) {
    if (G != 32 || K % 64 != 0) return -1;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += 2 * G) {
                float s = scales[((int64_t)j * K + k) / G];
                float s1 = scales[((int64_t)j * K + k) / G + 1];
                const uint8_t *b = B + (int64_t)j * (K / 2) + k / 2;
                const float *x = A + (int64_t)i * K + k;
                for (int e = 0; e < 32; e++) {
                    uint8_t p = b[e];
                    float d0 = (float)((double)(p & 0x0F) - 8.0) * s;
                    float d1 = (float)((double)(p >> 4) - 8.0) * s1;
                    float t0 = x[e] * d0;
                    acc = acc + t0;
                    float t1 = x[32 + e] * d1;
                    acc = acc + t1;
                }
            }
            C[(int64_t)i * N + j] = acc;
        }
    return 0;
}

/*
 * MatmulOperator::naive_mat_mul_int4, QM_ARM branch -- kernels/matmul_int4.cc:50-76.  Per group of G weights (G / 2 bytes), per run of 16 bytes: byte e holds a
 * code in its low nibble that meets x[e] and one in its high nibble that meets x[16 + e]; the activation pointer advances by 16 per run (NOT by the 32 weights the run
 * holds -- for G = 32, the ARM models' group size, a group is one run and the pointer restarts at the next group: consistent; for larger groups the runs overlap,
 * restated as written).  Zero point hard-coded 8.
 */
ORC_API int orc_naive_mat_mul_int4_arm(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales, float *C) {
    if (G % 32 != 0 || K % G != 0) return -1;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += G) {
                float s = scales[((int64_t)j * K + k) / G];
                const uint8_t *b = B + (int64_t)j * (K / 2) + k / 2;
                const float *x = A + (int64_t)i * K + k;
                for (int qi = 0; qi < G / 2; qi += 16)
                    for (int qj = 0; qj < 16; qj++) {
                        uint8_t p = b[qi + qj];
                        float d0 = (float)((double)(p & 0x0F) - 8.0) * s;
                        float d1 = (float)((double)(p >> 4) - 8.0) * s;
                        float t0 = *x * d0;
                        acc = acc + t0;
                        float t1 = x[16] * d1;
                        acc = acc + t1;
                        x++;
                    }
            }
            C[(int64_t)i * N + j] = acc;
        }
    return 0;
}

/*
 * MatmulOperator::naive_mat_mul_int4, QM_METAL branch -- kernels/matmul_int4.cc:16-49.  Per 4 bytes: the four low nibbles are weights 0..3, the four high nibbles
 * weights 4..7 of eight consecutive k; all eight dequantised first, then accumulated in k order.  Zero point hard-coded 8.
 */
ORC_API int orc_naive_mat_mul_int4_metal(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales, float *C) {
    if (G % 8 != 0 || K % G != 0) return -1;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += G) {
                float s = scales[((int64_t)j * K + k) / G];
                const uint8_t *b = B + (int64_t)j * (K / 2) + k / 2;
                const float *x = A + (int64_t)i * K + k;
                for (int qi = 0; qi < G / 2; qi += 4) {
                    float d[8];
                    for (int e = 0; e < 4; e++) {
                        d[e] = (float)((double)(b[qi + e] & 0x0F) - 8.0) * s;
                        d[4 + e] = (float)((double)(b[qi + e] >> 4) - 8.0) * s;
                    }
                    for (int e = 0; e < 8; e++) {
                        float t = *x++ * d[e];
                        acc = acc + t;
                    }
                }
            }
            C[(int64_t)i * N + j] = acc;
        }
    return 0;
}

/* naive_mat_mul_int4_with_offset -- kernels/matmul_int4.cc:133-165 (deq = (q-z)*s + o). */
ORC_API void orc_naive_mat_mul_int4_with_offset(int M, int N, int K, int G, const float *A, const uint8_t *B,
                                                const float *scales, const float *offset, float zero_point, float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += G) {
                float s = scales[((int64_t)j * K + k) / G];
                float o = offset[((int64_t)j * K + k) / G];
                const float *a = A + (int64_t)i * K + k;
                const uint8_t *b = B + (int64_t)j * (K / 2) + k / 2;
                for (int q = 0; q < G / 2; q++) {
                    uint8_t p = b[q];
                    float m0 = ((float)(p & 0x0F) - zero_point) * s;
                    float d0 = m0 + o;
                    float m1 = ((float)(p >> 4) - zero_point) * s;
                    float d1 = m1 + o;
                    float t0 = a[2 * q] * d0;
                    acc = acc + t0;
                    float t1 = a[2 * q + 1] * d1;
                    acc = acc + t1;
                }
            }
            C[(int64_t)i * N + j] = acc;
        }
}

/* mat_mul_accelerator_int4_fast (ref backend) -- kernels/ref/matmul_ref_int4.cc:11-38: deq = q*s + o (no zero
 * point), block 32 only.  The function is driven entirely by B.row (`b_row`): k runs over [0, b_row), the weight row
 * stride is b_row BYTES and the scale row stride b_row/16.  Its only caller passes b_row = K/2
 * (llm/src/ops/linear.cc:138-139), so as called it contracts over the FIRST HALF of K only -- restated as is
 * (this backend function is not on the GPU path: the CUDA build stubs it, kernels/cuda/gemv_cuda.cu:262-264). */
ORC_API int orc_ref_int4_fast(int M, int N, int K, int G, int b_row, const float *A, const uint8_t *B,
                              const float *scales, const float *offset, float *C) {
    if (G != 32) return -1;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < b_row; k += G) {
                float s = scales[(int64_t)j * (b_row / 16) + k / 32];
                float o = offset[(int64_t)j * (b_row / 16) + k / 32];
                const uint8_t *b = B + (int64_t)j * b_row + k / 2;
                const float *a = A + (int64_t)i * K + k;
                for (int q = 0; q < G / 2; q++) {
                    uint8_t p = b[q];
                    float m0 = (float)(p & 0x0F) * s;
                    float d0 = m0 + o;
                    float m1 = (float)(p >> 4) * s;
                    float d1 = m1 + o;
                    float t0 = a[2 * q] * d0;
                    acc = acc + t0;
                    float t1 = a[2 * q + 1] * d1;
                    acc = acc + t1;
                }
            }
            C[(int64_t)i * N + j] = acc;
        }
    return 0;
}

/*
 * W4A16 GEMV on the q4_6 layout -- the math of gemv_kernel_g128 / g64
 * (kernels/cuda/gemv_cuda.cu:140-194, 68-123) restated with the accumulation
 * order of the kernels/ref-class oracle (naive_mat_mul_int4): inputs and scales
 * are fp16 promoted to fp32, the zero point is READ per group from the packed
 * zeros (gemv_cuda.cu:159,166), deq = s * (q - z) in fp32, accumulate in fp32
 * sequentially over k, final result stored both as fp32 and as __float2half (RNE,
 * gemv_cuda.cu:192).  With all zero nibbles == 8 the fp32 result is bit-identical
 * to orc_naive_mat_mul_int4 on the unpacked codes (tests pin this).
 * zeros row stride = orc_zeros_width(K,G); scales row stride = 8x that.
 */
ORC_API void orc_w4a16_gemv_q4_6(int M, int N, int K, int G, const uint16_t *A, const uint32_t *qweight,
                                 const uint16_t *scales, const uint32_t *zeros, float *C32, uint16_t *C16) {
    int zw = orc_zeros_width(K, G);
    int sfw = zw * 8;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k += G) {
                int g = k / G;
                float s = orc_f16_to_f32(scales[(int64_t)j * sfw + g]);
                float z = (float)((zeros[(int64_t)j * zw + g / 8] >> (4 * (g % 8))) & 0xF);
                for (int kk = k; kk < k + G; kk++) {
                    uint32_t word = qweight[(int64_t)j * (K / 8) + kk / 8];
                    float q = (float)((word >> (4 * (kk % 8))) & 0xF);
                    float deq = (q - z) * s;
                    float t = orc_f16_to_f32(A[(int64_t)i * K + kk]) * deq;
                    acc = acc + t;
                }
            }
            if (C32) C32[(int64_t)i * N + j] = acc;
            if (C16) C16[(int64_t)i * N + j] = orc_f32_to_f16(acc);
        }
}

/*
 * MatmulOperator::naive_mat_mul_fp16_int4 -- kernels/cuda/matmul_int4.cu:8-48.
 * AWQ GEMM layout (q4_5), zero point fixed 8, EVERY operation in software
 * binary16 with RNE per operation (half_float 2.2.0 without
 * HALF_ARITHMETIC_TYPE: llm/half-2.2.0/include/half.hpp:296,386-387):
 *   weight = h( h(q - 8) * s );  acc = h( acc + h(input * weight) ).
 * Products of two halfs are exact in double; sums of two halfs are exact in
 * double; so one orc_f64_to_f16 per op reproduces the correctly rounded result.
 */
ORC_API void orc_naive_mat_mul_fp16_int4(int M, int N, int K, int G, const uint16_t *A, const uint32_t *qweight,
                                         const uint16_t *scales, uint16_t *C) {
    static const int shift_of[8] = {0, 16, 4, 20, 8, 24, 12, 28}; /* j%8 -> bit offset (order 0 2 4 6 1 3 5 7) */
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            uint16_t acc = 0;
            for (int k = 0; k < K; k++) {
                double s = (double)orc_f16_to_f32(scales[(int64_t)(k / G) * N + j]);
                double in = (double)orc_f16_to_f32(A[(int64_t)i * K + k]);
                uint32_t word = qweight[(int64_t)k * (N / 8) + j / 8];
                double q = (double)((word >> shift_of[j % 8]) & 0xF);
                uint16_t qz = orc_f64_to_f16(q - 8.0);
                uint16_t wgt = orc_f64_to_f16((double)orc_f16_to_f32(qz) * s);
                uint16_t prod = orc_f64_to_f16(in * (double)orc_f16_to_f32(wgt));
                acc = orc_f64_to_f16((double)orc_f16_to_f32(acc) + (double)orc_f16_to_f32(prod));
            }
            C[(int64_t)i * N + j] = acc;
        }
}

/* ------------------------------------------------------------------------- */
/* W8A8 (SmoothQuant) paths -- kernels/ref/matmul_ref_int8.cc                */
/* ------------------------------------------------------------------------- */

static inline int32_t orc_dot_i8(const int8_t *a, const int8_t *b, int k) {
    int32_t acc = 0;
    for (int t = 0; t < k; t++) acc += (int32_t)a[t] * (int32_t)b[t];
    return acc;
}

static inline int8_t orc_requant(float v, int qmin, int qmax) {
    /* (int32_t)std::round(v); MAX(.,q_min); MIN(.,q_max); (int8_t) -- matmul_ref_int8.cc:29-32 */
    int32_t r = (int32_t)roundf(v);
    if (r < qmin) r = qmin;
    if (r > qmax) r = qmax;
    return (int8_t)r;
}

/* int8_ref_matmul -- matmul_ref_int8.cc:11-35: int8 bias, alpha, beta; B is [N][K]. */
ORC_API void orc_int8_matmul_bias_i8(int M, int N, int K, const int8_t *A, const int8_t *B, const int8_t *bias,
                                     float alpha, float beta, int qmin, int qmax, int8_t *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + (int64_t)j * K, K);
            float t = (float)acc * alpha;
            float u = (float)bias[j] * beta;
            float v = t + u;
            C[(int64_t)i * N + j] = orc_requant(v, qmin, qmax);
        }
}

/* int8_ref_matmul_nobias -- matmul_ref_int8.cc:37-61 */
ORC_API void orc_int8_matmul_nobias_i8(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha, int qmin,
                                       int qmax, int8_t *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + (int64_t)j * K, K);
            float t = (float)acc * alpha;
            C[(int64_t)i * N + j] = orc_requant(t, qmin, qmax);
        }
}

/* int8_ref_matmul_nobias_batch -- matmul_ref_int8.cc:63-87: row i of A uses its own B[i] ([M][N][K]). */
ORC_API void orc_int8_matmul_nobias_batch_i8(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha,
                                             int qmin, int qmax, int8_t *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + ((int64_t)i * N + j) * K, K);
            float t = (float)acc * alpha;
            C[(int64_t)i * N + j] = orc_requant(t, qmin, qmax);
        }
}

/* int8_ref_matmul_bfp32_ofp32 -- matmul_ref_int8.cc:89-111: fp32 bias added after the alpha scale. */
ORC_API void orc_int8_matmul_bias_f32(int M, int N, int K, const int8_t *A, const int8_t *B, const float *bias,
                                      float alpha, float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + (int64_t)j * K, K);
            float t = (float)acc * alpha;
            C[(int64_t)i * N + j] = t + bias[j];
        }
}

/* int8_ref_matmul_nobias_ofp32 -- matmul_ref_int8.cc:113-135 */
ORC_API void orc_int8_matmul_nobias_f32(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha, float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + (int64_t)j * K, K);
            C[(int64_t)i * N + j] = (float)acc * alpha;
        }
}

/* int8_ref_matmul_nobias_ofp32_batch -- matmul_ref_int8.cc:137-159 */
ORC_API void orc_int8_matmul_nobias_batch_f32(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha,
                                              float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = orc_dot_i8(A + (int64_t)i * K, B + ((int64_t)i * N + j) * K, K);
            C[(int64_t)i * N + j] = (float)acc * alpha;
        }
}

/*
 * MatmulOperator::naive_mat_mul_int8 -- kernels/matmul_int8.cc:8-30.  Zero-point
 * form, B UNtransposed [K][N], truncating cast of acc * (A_sc*B_sc/C_sc).
 */
ORC_API void orc_naive_mat_mul_int8(int M, int N, int K, const int8_t *A, const int8_t *B, int32_t A_zp, int32_t C_zp,
                                    float A_sc, float B_sc, float C_sc, int qmin, int qmax, int8_t *C) {
    float t0 = A_sc * B_sc;
    float eff = t0 / C_sc;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            int32_t acc = 0;
            for (int k = 0; k < K; k++) acc += ((int32_t)A[(int64_t)i * K + k] - A_zp) * (int32_t)B[(int64_t)k * N + j];
            float f = (float)acc * eff;
            int32_t r = (int32_t)f;
            r -= C_zp;
            if (r < qmin) r = qmin;
            if (r > qmax) r = qmax;
            C[(int64_t)i * N + j] = (int8_t)r;
        }
}

/* ------------------------------------------------------------------------- */
/* fp32 helpers on the same operator surface                                  */
/* ------------------------------------------------------------------------- */

/* mat_mul_transposed (kernels/matmul_imp.cc:23-35) == fp32_ref_matmul (kernels/ref/matmul_ref_fp32.cc:11-30):
 * C = A . B^T, B [N][K], sequential fp32 accumulate.  bias may be NULL
 * (…_fastover_column_bias, kernels/avx/matmul_avx_fp32.cc:93-117 adds bias[j] after the sum). */
ORC_API void orc_fp32_matmul_transposed(int M, int N, int K, const float *A, const float *B, const float *bias, float *C) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) {
                float t = A[(int64_t)i * K + k] * B[(int64_t)j * K + k];
                acc = acc + t;
            }
            if (bias) acc = acc + bias[j];
            C[(int64_t)i * N + j] = acc;
        }
}

/* ------------------------------------------------------------------------- */
/* element-wise glue behind the W4A16 linears of a decoder layer */
/* ------------------------------------------------------------------------- */

/* add_half (llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:12-18): c[i] = __hadd(a[i], b[i]), one binary16 rounding
 * (the float sum of two halves is exact).  add_half and SiLuMul_half are pinned against the reference's kernel sources run
 * through oracle/cuda_emul/ (tests/test_oracle_glue.py), and so is orc_rmsnorm_half below (its kernel needs warp shuffles and
 * __syncthreads: the emulation runs that block's threads concurrently). */
ORC_API void orc_add_half(const uint16_t *a, const uint16_t *b, uint16_t *c, int64_t n) {
    for (int64_t i = 0; i < n; i++) c[i] = orc_f32_to_f16(orc_f16_to_f32(a[i]) + orc_f16_to_f32(b[i]));
}

/* SiLuMul_half (Int4llamaDecoderLayer.cu:20-30): a[i] = __hmul(__hmul(v, __hdiv(1, __hadd(1, hexp(__hneg(v))))), b[i]),
 * every operation rounded to binary16.  The float product / sum of two halves is exact, so each is one rounding; the
 * quotient is formed in double (no double-rounding case for an 11-bit result).  hexp: CUDA documents round-to-nearest
 * of the exponential; restated as the C library's expf rounded to half.  A device exponential that differs from expf
 * in the last float bit changes the half result only when that float sits on a rounding boundary (~1 element in 4000,
 * by one half ulp of e, which moves the final product by at most one ulp): the GPU test allows exactly that. */
ORC_API void orc_silu_mul_half(const uint16_t *a, const uint16_t *b, uint16_t *out, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const float v = orc_f16_to_f32(a[i]);
        const uint16_t e = orc_f32_to_f16(expf(-v));
        const uint16_t d = orc_f32_to_f16(1.0f + orc_f16_to_f32(e));
        const uint16_t r = orc_f64_to_f16(1.0 / (double)orc_f16_to_f32(d));
        const uint16_t sv = orc_f32_to_f16(v * orc_f16_to_f32(r));
        out[i] = orc_f32_to_f16(orc_f16_to_f32(sv) * orc_f16_to_f32(b[i]));
    }
}

/* generalT5LayerNorm, the kernel LlamaRMSNorm_cuda::forward launches (llm/src/ops/cuda/LlamaRMSNorm.cu:68-115):
 *   out[i] = half( clamp( (float(x[i]) * rsqrtf(sum_j x[j]^2 / n + eps)) * gamma[i] ) ),  clamp to +-(65504 - 1000)
 * (clamp_inf_for_half, llm/include/ops/cuda/reduction.cuh:76-81).  The sum is formed as the reference forms it: blockDim
 * = min(n, 1024) / 2 threads (1024 / 2 when n % 32 != 0), thread t adds its elements t, t + blockDim, ... in order, then
 * the butterfly of warpReduceSum over the 32 lanes of each warp and once more over the warps (reduction.cuh:38-67).
 * rsqrtf is an approximate instruction there (2 ulp); 1 / sqrtf is used here and on the GPU side of the tests. */
ORC_API void orc_rmsnorm_half(const uint16_t *x, const float *gamma, uint16_t *out, int m, int n, float eps) {
    int bd = (n < 1024 ? n : 1024);
    if (n % 32 != 0) bd = 1024;
    bd /= 2;
    if (bd < 1) bd = 1;
    for (int r = 0; r < m; r++) {
        const uint16_t *xr = x + (int64_t)r * n;
        float part[1024];
        for (int t = 0; t < 1024; t++) part[t] = 0.0f;
        for (int t = 0; t < bd; t++) {
            float s = 0.0f;
            for (int i = t; i < n; i += bd) {
                const float d = orc_f16_to_f32(xr[i]);
                s = s + d * d;
            }
            part[t] = s;
        }
        const int nwarp = (bd + 31) / 32;
        float shared[32];
        for (int w = 0; w < 32; w++) shared[w] = 0.0f;
        for (int w = 0; w < nwarp; w++) {  /* warpReduceSum: val += shfl_xor(val, mask) for mask = 16, 8, 4, 2, 1 */
            float v[32];
            for (int l = 0; l < 32; l++) v[l] = (w * 32 + l < bd) ? part[w * 32 + l] : 0.0f;
            for (int mask = 16; mask > 0; mask >>= 1) {
                float nv[32];
                for (int l = 0; l < 32; l++) nv[l] = v[l] + v[l ^ mask];
                for (int l = 0; l < 32; l++) v[l] = nv[l];
            }
            shared[w] = v[0];
        }
        float v[32];
        for (int l = 0; l < 32; l++) v[l] = ((float)l < (float)bd / 32.f) ? shared[l] : 0.0f;
        for (int mask = 16; mask > 0; mask >>= 1) {
            float nv[32];
            for (int l = 0; l < 32; l++) nv[l] = v[l] + v[l ^ mask];
            for (int l = 0; l < 32; l++) v[l] = nv[l];
        }
        const float rs = 1.0f / sqrtf(v[0] / (float)n + eps);
        for (int i = 0; i < n; i++) {
            float f = (orc_f16_to_f32(xr[i]) * rs) * gamma[i];
            f = f > 0.0f ? (f < 65504.F - 1000 ? f : 65504.F - 1000) : (f > -65504.F + 1000 ? f : -65504.F + 1000);
            out[(int64_t)r * n + i] = orc_f32_to_f16(f);
        }
    }
}

/* LayerNormQ::forward (llm/src/ops/LayerNormQ.cc:12-52), the op in front of every W8A8 linear of the OPT path: fp32 in,
 * int8 out, eps = 1e-5, every sum sequential in fp32, out = (int8) round( (v - mean) / std * w + b ) (std::round: half
 * away from zero; no clamp in the reference -- values are assumed to fit).  The narrowing is done through int32 here,
 * which is what the reference's static_cast does for in-range values.  Pinned against the reference's own LayerNormQ.cc compiled
 * into oracle/_ref/glue_harness (tests/test_oracle_glue.py). */
ORC_API void orc_layernorm_q(const float *x, const float *w, const float *b, int8_t *out, int m, int n) {
    const float eps = 0.00001;
    for (int r = 0; r < m; r++) {
        const float *xr = x + (int64_t)r * n;
        float mean = 0;
        for (int k = 0; k < n; k++) mean += xr[k];
        mean /= (float)n;
        float sq = 0;
        for (int k = 0; k < n; k++) {
            const float d = xr[k] - mean;
            const float p = d * d;
            sq += p;
        }
        const float var = sq / (float)n;
        const float std_dev = sqrtf(var + eps);
        for (int k = 0; k < n; k++) {
            const float t = (xr[k] - mean) / std_dev;
            const float u = t * w[k];
            const float f = u + b[k];
            out[(int64_t)r * n + k] = (int8_t)(int32_t)roundf(f);
        }
    }
}

/* The element-wise steps between the two int8 BMMs of the reference's OPT attention, restated operation by operation:
 *   batch_Add   (llm/src/ops/batch_add.cc:13-19):   v = s[i][j][k] + mask[0][j][k]
 *   softmax     (llm/src/ops/softmax.cc:11-36):     max_value starts from input.m_data[0] -- element [0][0][0] of the WHOLE tensor, not of the
 *               row (:13) -- and the caller runs it IN PLACE (attn_probs wraps attn_weights' array, Int8OPTAttention.cc:258-260), so from the
 *               second row on that element already holds row (0, 0)'s first PROBABILITY, not its score; sum += std::exp(v - max) for k ascending (fp32); the quotient std::exp(v - max) / (sum + 1e-10) is a
 *               DOUBLE division (the literal 1e-10 promotes the divisor, :31) rounded to float on assignment
 *   int8 probs  (llm/src/nn_modules/Int8OPTAttention.cc:264-267): static_cast<int8_t>(std::round(p * 127)), the product in fp32
 * Pinned against the reference's own batch_add.cc and softmax.cc compiled into oracle/_ref/glue_harness (tests/test_oracle_glue.py). */
ORC_API void orc_opt_softmax_q(const float *scores, const float *mask, int8_t *probs, int heads, int sq, int tgz) {
    float first = scores[0] + mask[0];  /* m_data[0] as row (0, 0) sees it; overwritten by that row's first output (in place) */
    for (int i = 0; i < heads; i++)
        for (int j = 0; j < sq; j++) {
            const float *s = scores + ((int64_t)i * sq + j) * tgz;
            const float *m = mask + (int64_t)j * tgz;
            float max_value = first;
            float sum = 0;
            for (int k = 0; k < tgz; k++) {
                const float value = s[k] + m[k];
                if (value > max_value) max_value = value;
            }
            for (int k = 0; k < tgz; k++) {
                const float value = s[k] + m[k];
                sum += expf(value - max_value);
            }
            for (int k = 0; k < tgz; k++) {
                const float value = s[k] + m[k];
                const float final_v = (float)((double)expf(value - max_value) / ((double)sum + 1e-10));
                const float scaled = final_v * 127;
                probs[((int64_t)i * sq + j) * tgz + k] = (int8_t)roundf(scaled);
                if (i == 0 && j == 0 && k == 0) first = final_v;
            }
        }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* Attention ops either side of the int4 linears.  The reference implements these only as CUDA      */
/* kernels (llm/src/ops/cuda/BMM_F16T.cu, softmax.cu, RotaryPosEmb.cu), which cannot run on a device here; what       */
/* follows restates their arithmetic operation by operation.  PINNED (round 2) against those kernel SOURCES executed  */
/* on the CPU through the host emulation in oracle/cuda_emul/ (`make glue`, tests/test_oracle_glue.py: bit for bit) -- */
/* which fixes structure and order of operations; the binary16 primitives are this file's (the fused multiply-add is  */
/* exact integer arithmetic, checked against exact rational arithmetic in tests/test_oracle.py), and hexp is modelled */
/* on BOTH sides as the C library's expf rounded to binary16 (CUDA's hexp is an approximation of its own).            */
/* ------------------------------------------------------------------------------------------------------------------ */

/* value of a finite binary16 as m * 2^e with integer m (|m| < 2^11) */
static void orc_half_parts(uint16_t h, int64_t *m, int *e) {
    const int exp = (h >> 10) & 0x1F;
    const int64_t man = h & 0x3FF;
    int64_t mm;
    int ee;
    if (exp == 0) { mm = man; ee = -24; }
    else { mm = man | 0x400; ee = exp - 25; }
    *m = (h & 0x8000u) ? -mm : mm;
    *e = ee;
}

/* round (sign, magnitude * 2^-72) to binary16, nearest even; magnitude is an exact integer */
static uint16_t orc_round_i128_to_half(int neg, unsigned __int128 mag) {
    const uint16_t sign = neg ? 0x8000u : 0;
    if (mag == 0) return sign;
    int msb = 127;
    while (!((mag >> msb) & 1)) msb--;
    /* value = mag * 2^-72; the leading bit has weight 2^(msb - 72) */
    int e = msb - 72;
    int drop; /* low bits to drop so that 11 bits (normal) or fewer (subnormal) remain */
    if (e >= -14) drop = msb - 10;
    else drop = (-24) + 72; /* subnormal: keep multiples of 2^-24 */
    unsigned __int128 kept, rem, half;
    if (drop <= 0) {
        kept = mag << (-drop);
        rem = 0;
        half = 1;
    } else {
        kept = mag >> drop;
        rem = mag & ((((unsigned __int128)1) << drop) - 1);
        half = ((unsigned __int128)1) << (drop - 1);
    }
    if (rem > half || (rem == half && (kept & 1))) kept++;
    if (e >= -14) {
        if (kept == 2048) { kept = 1024; e++; }
        if (e > 15) return (uint16_t)(sign | 0x7C00u);
        return (uint16_t)(sign | (uint16_t)((e + 15) << 10) | (uint16_t)(kept - 1024));
    }
    return (uint16_t)(sign | (uint16_t)kept); /* kept <= 1024: 1024 is the smallest normal, encoded correctly */
}

/* __hfma(a, b, c): a * b + c with ONE rounding (finite inputs; inf / nan are not produced by the tests' data) */
ORC_API uint16_t orc_hfma(uint16_t a, uint16_t b, uint16_t c) {
    int64_t ma, mb, mc;
    int ea, eb, ec;
    orc_half_parts(a, &ma, &ea);
    orc_half_parts(b, &mb, &eb);
    orc_half_parts(c, &mc, &ec);
    /* everything in units of 2^-72: products have e >= -48, addends e >= -24, so all shifts are >= 0 and < 104 bits */
    __int128 p = (__int128)(ma * mb);
    p = p * ((__int128)1 << (ea + eb + 72));
    __int128 q = (__int128)mc * ((__int128)1 << (ec + 72));
    __int128 s = p + q;
    if (s == 0) {
        /* exact zero: +0 unless both terms are negative zeros (RNE) */
        const int pneg = ((a ^ b) & 0x8000u) != 0, cneg = (c & 0x8000u) != 0;
        return (pneg && cneg) ? 0x8000u : 0;
    }
    const int neg = s < 0;
    return orc_round_i128_to_half(neg, (unsigned __int128)(neg ? -s : s));
}

static uint16_t orc_hop(double v) { return orc_f64_to_f16(v); } /* one binary16 rounding of an exactly computed double */

/* BMM_F16T::forward -> mat_mul_transposed_cuda (llm/src/ops/cuda/BMM_F16T.cu:28-45): per batch b, C[i][j] =
 * __hmul(alpha, acc) with acc = 0; acc = __hfma(A[i][k], B[j][k], acc) for k ascending.  A [batch][M][K], B [batch][N][K],
 * C [batch][M][N], all binary16.  Both attention products of Int4llamaAttention use it (qk with alpha = 1/sqrt(head_dim) as
 * stored, pv with alpha = 1 on the transposed V: Int4llamaAttention.cu:185, 211). */
ORC_API void orc_bmm_f16t(int batch, int M, int N, int K, const uint16_t *A, const uint16_t *B, uint16_t *C, uint16_t alpha) {
    const double al = (double)orc_f16_to_f32(alpha);
    for (int b = 0; b < batch; b++)
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) {
                const uint16_t *a = A + ((int64_t)b * M + i) * K;
                const uint16_t *w = B + ((int64_t)b * N + j) * K;
                uint16_t acc = 0;
                for (int k = 0; k < K; k++) acc = orc_hfma(a[k], w[k], acc);
                C[((int64_t)b * M + i) * N + j] = orc_hop(al * (double)orc_f16_to_f32(acc)); /* 11 x 11 bits: exact in double */
            }
}

/* softmax_cuda (llm/src/ops/cuda/softmax.cu:4-40): per row, max by comparison from -65504; sum = __hadd(sum,
 * hexp(__hsub(x, max))) for k ascending; out = __hdiv(hexp(__hsub(x, max)), sum). */
ORC_API void orc_softmax_half(int64_t rows, int n, const uint16_t *x, uint16_t *out) {
    for (int64_t r = 0; r < rows; r++) {
        const uint16_t *xr = x + r * n;
        float mx = -65504.0f;
        for (int k = 0; k < n; k++) {
            const float v = orc_f16_to_f32(xr[k]);
            mx = mx > v ? mx : v;
        }
        uint16_t sum = 0;
        for (int k = 0; k < n; k++) {
            const uint16_t d = orc_hop((double)orc_f16_to_f32(xr[k]) - (double)mx);
            const uint16_t e = orc_f32_to_f16(expf(orc_f16_to_f32(d)));
            sum = orc_hop((double)orc_f16_to_f32(sum) + (double)orc_f16_to_f32(e));
        }
        for (int k = 0; k < n; k++) {
            const uint16_t d = orc_hop((double)orc_f16_to_f32(xr[k]) - (double)mx);
            const uint16_t e = orc_f32_to_f16(expf(orc_f16_to_f32(d)));
            out[r * n + k] = orc_hop((double)orc_f16_to_f32(e) / (double)orc_f16_to_f32(sum));
        }
    }
}

/* RotaryPosEmb_cuda_forward (llm/src/ops/cuda/RotaryPosEmb.cu:4-34), in place on q and k [heads][len][hd]; cos / sin
 * [positions][hd]: x'[j] = __hfma(x[j], cos[p][j], __hmul(rot[j], sin[p][j])) with rot = (-x[hd/2:], x[:hd/2]), p = i +
 * start_idx.  PARITY UNPINNED like the operators above (CUDA-only in the reference). */
ORC_API void orc_rope_half(uint16_t *q, uint16_t *k, const uint16_t *cosv, const uint16_t *sinv, int heads, int len, int hd, int start_idx) {
    uint16_t buf[512];
    const int hp = hd / 2;
    for (int t = 0; t < 2; t++) {
        uint16_t *x = t ? k : q;
        if (!x) continue;
        for (int b = 0; b < heads; b++)
            for (int i = 0; i < len; i++) {
                uint16_t *r = x + ((int64_t)b * len + i) * hd;
                const uint16_t *c = cosv + (int64_t)(i + start_idx) * hd, *s = sinv + (int64_t)(i + start_idx) * hd;
                for (int j = 0; j < hp; j++) buf[j] = (uint16_t)(r[j + hp] ^ 0x8000u); /* __hneg */
                for (int j = hp; j < hd; j++) buf[j] = r[j - hp];
                for (int j = 0; j < hd; j++) {
                    const uint16_t m = orc_hop((double)orc_f16_to_f32(buf[j]) * (double)orc_f16_to_f32(s[j])); /* __hmul: exact product, one rounding */
                    buf[j] = orc_hfma(r[j], c[j], m);
                }
                for (int j = 0; j < hd; j++) r[j] = buf[j];
            }
    }
}
