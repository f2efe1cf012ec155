/*
 * tce_matmul.h -- C ABI of libtce_hip.so: the MI355X (gfx950 / CDNA4) implementation of
 * TinyChatEngine's quantized-matmul hot path (reference: kernels/matmul.h @ 2024_08_07).
 *
 * This is the drop-in boundary.  Every entry point takes plain pointers and sizes (no C++
 * or torch types), is ASYNCHRONOUS (returns after enqueueing on `stream`; NULL = the HIP
 * null stream, which is what the reference uses -- kernels/cuda/gemv_cuda.cu:237), never
 * allocates, frees or retains caller memory, and returns 0 or a negative TCE_ERR_* code
 * (it never throws or exits; the C++ adapter in tinychatengine_amd/adapter/ maps failures to
 * the reference's printf+exit(1) behaviour -- kernels/cuda/gemv_cuda.cu:254-256).
 *
 * All data pointers must be device-accessible (hipMalloc or hipMallocManaged memory).
 *
 * Which reference symbol each entry point replaces:
 *
 *   tce_w4a16_forward          <- matmul::MatmulOperator::gemv_forward_cuda
 *                                 (kernels/cuda/gemv_cuda.cu:213-260; kernels gemv_kernel_g128 :140-194,
 *                                  gemv_kernel_g64 :68-123).  Serves every M like the reference does
 *                                 (grid.z = M there); here M <= 2 runs the bandwidth-bound GEMV kernels,
 *                                 3 <= M <= 16 (group 128) the small-batch kernel that streams the weights once for
 *                                 all rows -- also 17 <= M <= 128 in 16-row slices while N is too small to fill the
 *                                 chip with GEMM tiles --, larger M the MFMA GEMM kernels (groups of 128, and from
 *                                 M = 17 also 64 and 32 when K % 128 == 0); smaller M with another group size stays
 *                                 on the GEMV kernels.
 *   tce_w4a16_forward_group    <- several gemv_forward_cuda calls that read the same activation
 *                                 (fused q/k/v: llm/src/nn_modules/cuda/Int4llamaAttention.cu:125;
 *                                  gate+up: Int4llamaDecoderLayer.cu:96-99) issued as ONE launch.
 *   tce_w4a16_awq_fp16acc      <- MatmulOperator::naive_mat_mul_fp16_int4 (kernels/cuda/matmul_int4.cu:8-48)
 *   tce_w4a16_gemm_awq         <- MatmulOperator::gemm_forward_cuda* (declared kernels/matmul.h:140-145,
 *                                 never defined in the reference; AWQ q4_5 layout, fp32 accumulate here)
 *   tce_w8a8_matmul            <- the eight int8 methods of kernels/ref/matmul_ref_int8.cc:161-192
 *                                 (int8_ref_matmul{,_nobias,_nobias_batch,_bfp32_ofp32,_nobias_ofp32,
 *                                  _nobias_ofp32_batch}), selected by the descriptor's bias/out kinds.
 *   tce_plan_*                 <- the host loop that issues one decode token's linears
 *                                 (llm/src/nn_modules/cuda/Int4llamaDecoder.cu:84-103) captured once into a
 *                                 hipGraph and replayed; new on MI355X (launch latency ~ kernel time).
 */
#ifndef TCE_MATMUL_H
#define TCE_MATMUL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCE_API __attribute__((visibility("default")))

#define TCE_VERSION 113 /* 0.1.13: tce_w8a8_describe_dispatch; 0.1.12: tce_w4a16_forward_independent (up to TCE_MAX_INDEPENDENT decode linears with their own activations and K as one launch: the sharded block); 0.1.11: tce_w4a16_gemm_scratch_faults (a k-cut exchange that gives up stores NaN and poisons its counter: loud, sticky), TCE_PLAN_TAGGED on packed copies runs the int8-contraction token kernel (tce_plan_is_chained = 4), TCE_DESC_V2_MAX_BYTES; 0.1.10: tce_attention_decode_step_deferred_f16 + tce_w4a16_forward_deferred_attention (the attention combine in o_proj's prologue); size-prefixed descriptors (tce_w4a16_desc_v2 / tce_w8a8_desc_v2 + the *_v2 entry points; the plain ones stay), TCE_ERR_RCCL, the tuning setters act on the CALLING THREAD only; 0.1.9: tce_w4a16_check_zero_point_8_async, tce_host_alloc / tce_host_free (the adapter no longer synchronises); 0.1.8: per-family tuning setters (tce_attention_set_tuning, tce_w8a8_set_tuning); 0.1.7: decode on the pre-packed copy (int8 contraction), tce_w4a16_set_gemv_i8; 0.1.6: TCE_PLAN_TUNED; 0.1.5: tce_opt_attention_decode; 0.1.4: tce_attention_prefill_f16 (0.1.3: tce_attention_decode_step_gqa_f16, TCE_PLAN_OVERLAPPED; 0.1.2: tce_w4a16_desc.scratch; 0.1.1: .prepacked, tce_w4a16_prepack*) */

/* error codes (return values) */
#define TCE_OK 0
#define TCE_ERR_BAD_ARG (-1)           /* null pointer / non-positive size */
#define TCE_ERR_UNSUPPORTED_GROUP (-2) /* group size not in {32,64,128}: reference prints "Unsupported group size" and exits */
#define TCE_ERR_UNSUPPORTED_SHAPE (-3) /* K % 32 != 0, N % 8 != 0 for the AWQ layout, ... */
#define TCE_ERR_HIP (-4)               /* a HIP runtime call failed; see tce_last_error() */
#define TCE_ERR_UNSUPPORTED_KIND (-5)  /* bias/out kind combination that has no reference counterpart */
#define TCE_ERR_RCCL (-6)              /* an RCCL call failed (ncclAllGather ...): tce_last_error() carries ncclGetErrorString's text -- NOT a HIP error */

/* M at or below which tce_w4a16_forward never uses the prefill GEMM (GEMV or small-batch kernels) */
#define TCE_W4A16_GEMV_MAX_M 8

/*
 * W4A16 on the reference's q4_6 ("CUDA GEMV") layout -- llm/tools/quantize_methods.py:370-442:
 *   A        fp16  [M][lda]           activations, row-major           (matmul_params.A.half_data_ptr)
 *   qweight  u32   [N][K/8]           nibble i of word j = code[n][8j+i] (matmul_params.B.int32_data_ptr)
 *   scales   fp16  [N][scales_stride] first K/G entries valid           (matmul_params.half_scales)
 *   zeros    u32   [N][zeros_stride]  nibble g%8 of word g/8            (matmul_params.int32_zero_point)
 *   C        fp16  [M][ldc]                                            (matmul_params.C.half_data_ptr)
 *   C[m][n] = fp16( sum_k fp32(s[n][k/G]) * (q[n][k] - z[n][k/G]) * fp32(A[m][k]) ), fp32 accumulate.
 * Strides of 0 select the reference's defaults: lda = K, ldc = N,
 * zeros_stride = calculate_zeros_width(K,G) (llm/src/nn_modules/cuda/utils.cu:162-178), scales_stride = 8x that.
 * Like the reference, B.row / B.column are NOT part of the contract (gemv_cuda.cu:217-221 ignores them).
 */
typedef struct tce_w4a16_desc {
    int32_t M, N, K;
    int32_t group_size; /* QK: 128 (reference CUDA default, llm/include/common.h:18), 64 or 32 */
    const void *A;
    const void *qweight;
    const void *scales;
    const void *zeros;
    void *C;
    int32_t lda, ldc;                     /* elements; 0 = dense */
    int32_t scales_stride, zeros_stride;  /* elements / words; 0 = reference default */
    int32_t flags;                        /* TCE_W4_* */
    int32_t reserved;
    const void *rmsnorm_gamma;            /* NULL = none.  fp32 [K]: A is the UN-normalised hidden state and the kernel stages
                                             RMSNorm(A) * gamma (generalT5LayerNorm arithmetic, see tce_rmsnorm_half); M = 1 */
    float rmsnorm_eps;
    int32_t reserved2;
    const void *prepacked;                /* NULL = none.  The q4_mfma copy of this linear's weights built by tce_w4a16_prepack
                                             (same N, K, group size): the prefill GEMM for large M AND the decode kernel for M <= 4
                                             (round 4: csrc/w4a16_gemv_i8.hip) read it instead of qweight / scales / zeros (which must
                                             still be valid: every other M, and the fused RMSNorm prologue, use them) */
    void *scratch;                        /* NULL = none.  tce_w4a16_gemm_scratch_bytes() bytes of device memory, 256-byte aligned, its first 4096
                                             bytes ZEROED once by the caller (every call leaves them zero): lets the pre-packed GEMM cut the k range of
                                             a launch with few tiles (M = 512 at N = 4096 is 128 tiles for 256 CUs) across workgroups
                                             and add the partial tiles in a fixed order.  One scratch area per stream that runs such
                                             calls concurrently; calls on one stream may share it.  (0.1.10: a range cut in TWO runs is a
                                             directed hand-off -- run 1 waits, for a bounded number of polls, for the tile of run 0, which is
                                             dispatched before it on the same XCD's queue.  A wait that runs out -- never seen -- is LOUD since
                                             0.1.11: the tile is stored as NaN, the tile's counter word is poisoned so that every later call
                                             meeting it stores NaN too, and the last word of the first 4096 bytes counts the faults:
                                             tce_w4a16_gemm_scratch_faults() reads it; zeroing the first 4096 bytes recovers the area.) */
} tce_w4a16_desc;

/* flags */
#define TCE_W4_FORCE_GEMV 1  /* use the GEMV kernel even when M > TCE_W4A16_GEMV_MAX_M (weights re-streamed per 4 rows) */
#define TCE_W4_FORCE_GEMM 2  /* use the MFMA GEMM kernel even for small M */
#define TCE_W4_ZERO_POINT_IS_8 4 /* the caller vouches that every 4-bit zero point of this linear is 8 (`zeros` all
                                    0x88888888 -- what the reference quantizer always writes, quantize_methods.py:436-440):
                                    the GEMV then does not stream the zeros.  Results are identical when the promise holds. */

/* Fused decode epilogues: the element-wise kernels the reference launches right behind these linears,
 * with the same fp16 arithmetic, applied while the result is still in registers.
 *   TCE_W4_SILU_MUL_PAIRS  replaces gate_proj + up_proj + SiLuMul_half (Int4llamaDecoderLayer.cu:20-30, 96-102): the
 *       linear holds the two projections' rows INTERLEAVED (row 2n = gate row n, row 2n+1 = up row n; a load-time row
 *       permutation, like the reference's offline qkv merge), N is even, and
 *           C[m][n] = hmul( hmul(g, hdiv(1, hadd(1, hexp(hneg(g))))), u ),  g = fp16(y[m][2n]), u = fp16(y[m][2n+1])
 *       with every operation rounded to fp16 as in the reference kernel.  C is [M][N/2] (ldc 0 = N/2).  Decode batches (M <= 8) run it in the GEMV
 *       kernels' epilogue; a batch of M > 128 rows with a `prepacked` copy in the 128-row GEMM's (the prompt path); anything else on the GEMV kernel,
 *       four rows per pass.
 *   TCE_W4_ADD_TO_C        replaces o_proj / down_proj + add_half (Int4llamaDecoderLayer.cu:12-18, 86-88, 107-108):
 *           C[m][n] = hadd(C[m][n], fp16(y[m][n]))   (C holds the residual on entry, like residual_add there). */
#define TCE_W4_SILU_MUL_PAIRS 8
#define TCE_W4_ADD_TO_C 16

TCE_API int tce_w4a16_forward(const tce_w4a16_desc *d, void *stream);

/* Size-prefixed form (0.1.10): a host compiled against THIS header keeps working when a later library appends fields to the descriptor, and a later host talks to
 * this library as long as it leaves the fields this library does not know at zero.  `struct_size` = sizeof(tce_w4a16_desc_v2) as the CALLER compiled it; the library
 * copies min(struct_size, its own size) bytes into a zeroed descriptor of its own and refuses a size below the 0.1.10 layout (TCE_ERR_BAD_ARG) or non-zero bytes
 * beyond what it knows.  `struct_size` above TCE_DESC_V2_MAX_BYTES (an uninitialised descriptor) and a non-zero `reserved0` are refused too (TCE_ERR_BAD_ARG).
 * Same semantics as tce_w4a16_forward(&v2->desc, stream). */
#define TCE_DESC_V2_MAX_BYTES 4096
typedef struct tce_w4a16_desc_v2 {
    uint32_t struct_size;
    uint32_t reserved0;
    tce_w4a16_desc desc;
    /* fields of later versions are appended here */
} tce_w4a16_desc_v2;
TCE_API int tce_w4a16_forward_v2(const tce_w4a16_desc_v2 *d, void *stream);

/* Round 4: o_proj / down_proj + residual add + the RMSNorm that FOLLOWS, one launch (decode, M = 1) -- replaces `linear; add_half; LlamaRMSNorm` of
 * Int4llamaDecoderLayer.cu:86-99 (post_attention_layernorm) and :107-108 + :78 of the next layer (input_layernorm):
 *   C[n]  = hadd(C[n], fp16(y[n]))                                           as TCE_W4_ADD_TO_C (which `d` must carry; C is the whole residual row: ldc 0 or N)
 *   xn[n] = half( clamp( (float(C[n]) * rs) * gamma[n] ) ),  rs = 1 / sqrt(mean_n C[n]^2 + eps)   generalT5LayerNorm on the UPDATED row: the bits of tce_rmsnorm_half
 * so that the next linears (q/k/v, gate/up) are plain launches on xn instead of carrying the prologue -- which every one of their workgroups repeats.  Every workgroup
 * writes its 16 values and their sums of squares through to memory and counts itself in; the one that arrives last normalises the row.  Needs `d->prepacked` (the
 * int8-contraction kernel), group 128, N % 8 == 0, N <= 16384.  `workspace`: tce_w4a16_residual_rmsnorm_workspace_bytes() bytes, 16-byte aligned, ZEROED once by the
 * caller (every launch leaves its counter zero); one per stream that runs such launches concurrently.
 * Measured (round 4): bit-exact and SLOWER than what it replaces on this part -- +3.8 us per launch (write-through acknowledgements, the last workgroup's
 * serial pass) against the +1.7 / +5.5 us the fused prologues cost q/k/v and gate/up: a whole token 1.65 against 1.41 ms.  Offered for hosts that must count launches. */
TCE_API size_t tce_w4a16_residual_rmsnorm_workspace_bytes(void);
TCE_API int tce_w4a16_forward_residual_rmsnorm(const tce_w4a16_desc *d, const float *gamma, float eps, void *xn_out, void *workspace, void *stream);

/* The same two element-wise operations as stand-alone kernels (for hosts that keep the reference's launch structure):
 *   tce_add_half:       c[i] = hadd(a[i], b[i])                                   (add_half, Int4llamaDecoderLayer.cu:12-18)
 *   tce_silu_mul_half:  a[i] = hmul(hmul(a[i], hdiv(1, hadd(1, hexp(-a[i])))), b[i])  (SiLuMul_half, :20-30)
 * n halves; pointers 16-byte aligned; c may alias a or b. */
/* The two fp16 operators between the q/k/v and the o linears of the reference's Llama attention, with its
 * arithmetic (see csrc/attention_ops.hip):
 *   tce_bmm_f16t     <- BMM_F16T::forward / mat_mul_transposed_cuda (llm/src/ops/cuda/BMM_F16T.cu:28-76):
 *                       A fp16 [batch][M][K], B fp16 [batch][N][K], C fp16 [batch][M][N];
 *                       C = hmul(alpha, acc), acc = hfma(A[i][k], B[j][k], acc) for k ascending (binary16 accumulation);
 *                       alpha as binary16 bits (what alpha_half.bin holds)
 *   tce_softmax_half <- softmax_cuda (llm/src/ops/cuda/softmax.cu:4-40): rows of n binary16 values */
TCE_API int tce_bmm_f16t(const void *A, const void *B, void *C, int batch, int M, int N, int K, unsigned short alpha_half_bits, void *stream);
TCE_API int tce_softmax_half(const void *x, void *out, long long rows, int n, void *stream);
/* RotaryPosEmb_cuda_forward (llm/src/ops/cuda/RotaryPosEmb.cu:4-34), in place on q and / or k fp16 [heads][len][head_dim] (NULL skips one);
 * cos / sin tables fp16 [positions][head_dim]; row i uses position i + start_idx:  x'[j] = hfma(x[j], cos[j], hmul(rot[j], sin[j])),
 * rot = (-x[head_dim/2:], x[:head_dim/2]). */
TCE_API int tce_rope_half(void *q, void *k, const void *cos_table, const void *sin_table, int heads, int len, int head_dim, int start_idx, void *stream);
/* One decode step (one query row per head) of Int4llamaAttention's qk_bmm -> batch_Add(mask) -> check_inf_half -> softmax ->
 * pv_bmm (llm/src/nn_modules/cuda/Int4llamaAttention.cu:184-211) as ONE launch, every operation and every order kept (bit-identical
 * to tce_bmm_f16t + hadd + tce_softmax_half + tce_bmm_f16t): q fp16 [heads][head_dim], K [heads][keys][head_dim],
 * Vt [heads][head_dim][keys] (the transposed values the reference also keeps), mask fp16 [keys] or NULL, out [heads][head_dim]. */
TCE_API int tce_attention_decode_f16(const void *q, const void *K, const void *Vt, const void *mask, void *out, int heads, int keys, int head_dim,
                                     unsigned short alpha_half_bits, void *stream);
/* One decode step of the reference's Llama attention block between the fused q/k/v linear and o_proj as ONE bandwidth-bound launch
 * (csrc/attention_fast.hip): replaces shape_qkv, RotaryPosEmb_cuda_forward, the KV append, qk_bmm, batch_Add, check_inf_half, softmax,
 * transpose_1_2idx, pv_bmm and unshape (llm/src/nn_modules/cuda/Int4llamaAttention.cu:130-217).
 *   qkv        fp16 [3][heads][head_dim]   the fused projection's output row (q | k | v, head-major: what qkv_proj.forward writes)
 *   k_cache, v_cache  fp16 [heads][max_keys][head_dim]   fixed-capacity caches; rows [0, pos) hold the past, row `pos` is WRITTEN by this
 *              call (the rotated key -- bit-identical to what the reference appends -- and the value)
 *   cos_table / sin_table  fp16 [positions][head_dim] (RotaryPosEmb's tables) or both NULL: no rotation
 *   mask       fp16 [pos + 1] additive, or NULL;  alpha as binary16 bits;  out fp16 [heads][head_dim] (= o_proj's input row)
 *   workspace  tce_attention_decode_workspace_bytes(heads, max_keys, head_dim) bytes, ZEROED once by the caller before the first use
 * Scores, softmax and the weighted sum run in fp32 over key chunks spread across the chip (the bit-exact binary16-chain form of the
 * same block is tce_attention_decode_f16): results agree with that form within 2e-3 * max|out| per head.  head_dim == 128. */
TCE_API size_t tce_attention_decode_workspace_bytes(int heads, int max_keys, int head_dim);
/* How tce_attention_decode_step_f16 would cut a context of `keys` keys (= pos + 1), as text: "chunks=C keys-per-chunk=K waves=W workgroups=G
 * combine=yes|no" -- one chunk per head and no combine up to 320 keys, four chunks up to 1024, then eight chunks of at most 512 keys.
 * Launches nothing, makes no HIP call. */
TCE_API int tce_attention_decode_describe(int heads, int keys, char *buf, int buf_len);
TCE_API int tce_attention_decode_step_f16(const void *qkv, void *k_cache, void *v_cache, const void *cos_table, const void *sin_table, const void *mask,
                                          void *out, void *workspace, int heads, int head_dim, int max_keys, int pos, unsigned short alpha_half_bits,
                                          void *stream);
/* The same step for grouped-query attention (Llama-3-8B: 32 query heads over 8 key / value heads, llm/include/model.h:83): query head i reads
 * key / value head i / (heads / kv_heads), the reference's `repeat` (llm/src/nn_modules/non_cuda/Int4llamaAttention.cc:166-185) without the copy.
 *   qkv  fp16 [heads + 2 * kv_heads][head_dim]: the query heads, then the key heads, then the value heads (the fused projection's row)
 *   k_cache, v_cache  fp16 [kv_heads][max_keys][head_dim];  out fp16 [heads][head_dim]
 *   workspace  tce_attention_decode_workspace_bytes(heads, max_keys, head_dim) bytes, zeroed once
 * Any heads % kv_heads == 0 (kv_heads == heads is tce_attention_decode_step_f16).  A workgroup owns (query head, chunk of keys) and the
 * heads / kv_heads workgroups of a key / value head read the same cache rows (from HBM once; the fused form -- one workgroup streaming a
 * chunk once for 2 or 4 query heads -- exists behind tce_w4a16_set_debug_mode(2922 / 2924) and measured slower, see csrc/attention_fast.hip).  Rows of the caches at and beyond `pos` may hold anything on entry (uninitialised
 * memory included): they are never weighted into the result. */
TCE_API int tce_attention_decode_step_gqa_f16(const void *qkv, void *k_cache, void *v_cache, const void *cos_table, const void *sin_table, const void *mask,
                                              void *out, void *workspace, int heads, int kv_heads, int head_dim, int max_keys, int pos,
                                              unsigned short alpha_half_bits, void *stream);
TCE_API int tce_attention_decode_describe_gqa(int heads, int kv_heads, int keys, char *buf, int buf_len);
/* The same step with the position read ON THE DEVICE: `pos_device` (int32, device memory) holds the token's position when the kernel runs, `pos_bound` >= every
 * value it will hold while this launch is replayed (the grid and the chunk length are cut for pos_bound; workgroups of chunks past the actual context leave at
 * once).  A launch captured into a hipGraph is then replayable token after token -- the host (or a one-thread kernel in the same graph) advances the word -- where
 * the by-value entry points bake the position into the graph node.  pos_device == NULL: pos_bound is the position (the by-value form).  `mask`, if given, must
 * cover pos_bound + 1 keys. */
TCE_API int tce_attention_decode_step_pos_f16(const void *qkv, void *k_cache, void *v_cache, const void *cos_table, const void *sin_table, const void *mask,
                                              void *out, void *workspace, int heads, int kv_heads, int head_dim, int max_keys, const int32_t *pos_device,
                                              int pos_bound, unsigned short alpha_half_bits, void *stream);

/* Round 5: the attention step WITHOUT its cross-workgroup combine, and the linear that consumes it (o_proj) doing the combine in its prologue.
 * A decode step over more than ~320 keys cuts every head's keys into 4 or 8 chunks; each chunk's workgroup ends with a partial online-softmax state (M, L, O[128]) and
 * the LAST one to arrive combines them -- acknowledged write-through stores, a counter, coherent re-reads: 2.3 - 3.2 us of a 7 - 10 us launch.
 * tce_attention_decode_step_deferred_f16 is tce_attention_decode_step_pos_f16 that stops at plain stores of the partial states and describes them in *info;
 * tce_w4a16_forward_deferred_attention is tce_w4a16_forward for the linear that reads the step's output row (d->A = the step's `out`; M = 1, d->prepacked, groups of
 * 128, K = heads * 128 a multiple of 1024, no RMSNorm prologue; TCE_W4_ADD_TO_C allowed): every workgroup forms the row from the partial states with the operations, and in
 * the order, of the step's own combine and rounds it to binary16 -- the SAME row, bit for bit -- while its weights are in flight.  The kernel boundary between the two launches
 * is the only ordering needed.  info->slots == 1 (a short context, or a cut beyond 8 slots): nothing was deferred, `out` is final, and the second call is the plain
 * tce_w4a16_forward.  With the position on the device the number of LIVE slots is recomputed by the consumer from the same word (when one is live the step wrote `out`
 * itself).  Both calls must see the same pos_device / pos_bound.  `workspace` as for the other entry points (tce_attention_decode_workspace_bytes). */
typedef struct tce_attention_deferred {
    int32_t slots;   /* chunk slots per query head: 1 = nothing deferred */
    int32_t chunk;   /* keys per chunk */
    int32_t heads;   /* query heads */
    int32_t stride;  /* floats per partial state: M, L, two unused, O[128] */
    const float *part; /* [heads][slots][stride], inside `workspace` */
} tce_attention_deferred;
TCE_API int tce_attention_decode_step_deferred_f16(const void *qkv, void *k_cache, void *v_cache, const void *cos_table, const void *sin_table, const void *mask,
                                                   void *out, void *workspace, int heads, int kv_heads, int head_dim, int max_keys, const int32_t *pos_device, int pos_bound,
                                                   unsigned short alpha_half_bits, tce_attention_deferred *info, void *stream);
TCE_API int tce_w4a16_forward_deferred_attention(const tce_w4a16_desc *d, const tce_attention_deferred *info, const int32_t *pos_device, int pos_bound, void *stream);
/* The same block for m > 1 NEW rows -- a prompt, or a chunk of one on top of pos cached keys (Int4llamaAttention.cu:116-229 with sqlen > 1) -- as two
 * launches (csrc/attention_prefill.hip): rotation of q and the new keys with the reference's binary16 arithmetic + the KV append (rows pos .. pos + m - 1;
 * bit-identical to what m decode steps append), then one pass over the keys per (query head, 64 query rows): scores and the weighted sum of V on the
 * matrix pipe with fp32 accumulation, online softmax in fp32.
 *   qkv        fp16 [m][ld_qkv]: per row the query heads, the key heads, the value heads (what the fused projection writes for M = m);
 *              ld_qkv = 0: (heads + 2 kv_heads) * head_dim
 *   k_cache, v_cache  fp16 [kv_heads][max_keys][head_dim] as for the decode step (pos + m <= max_keys); cos / sin tables [positions][head_dim] or both NULL
 *   mask       fp16 [m][ld_mask] additive (the reference's attention_mask: 0 / the lowest half), or NULL;  ld_mask = 0: pos + m
 *   causal     non-zero: row r sees keys 0 .. pos + r (in addition to the mask, if any) -- key tiles behind a block's diagonal are skipped
 *   out        fp16 [m][ld_out]: row r, columns head * head_dim .. (= o_proj's input rows; ld_out = 0: heads * head_dim)
 *   workspace  tce_attention_prefill_workspace_bytes(heads, m, head_dim) bytes (the rotated queries; no zeroing needed)
 * Not the reference's binary16 accumulation chains (tce_bmm_f16t + tce_softmax_half are those, bit for bit): agrees with a float64 evaluation within
 * 2e-3 * max|out| per head and row.  head_dim == 128.  Cache rows at and beyond pos + m may hold anything. */
TCE_API size_t tce_attention_prefill_workspace_bytes(int heads, int m, int head_dim);
TCE_API int tce_attention_prefill_f16(const void *qkv, int ld_qkv, void *k_cache, void *v_cache, const void *cos_table, const void *sin_table, const void *mask,
                                      int ld_mask, int causal, void *out, int ld_out, void *workspace, int heads, int kv_heads, int head_dim, int max_keys,
                                      int pos, int m, unsigned short alpha_half_bits, void *stream);
/* Reads [ptr, ptr + bytes) with at most `workgroups` workgroups (0 = as many as the range needs) and discards the data: the
 * range then sits in the memory-side cache (256 MiB) for the launch that needs it.  Meant for a side stream / graph branch
 * next to the launch BEFORE that one (no reference counterpart: cudaMallocManaged prefetching is the closest idea). */
TCE_API int tce_prefetch(const void *ptr, long long bytes, int workgroups, void *stream);
TCE_API int tce_add_half(const void *a, const void *b, void *c, long long n, void *stream);
/* RMSNorm as the reference's CUDA build computes it (generalT5LayerNorm, llm/src/ops/cuda/LlamaRMSNorm.cu:68-115):
 *   out[r][i] = half( clamp( (float(x[r][i]) * rs_r) * gamma[i] ) ),  rs_r = 1 / sqrt(mean_i x[r][i]^2 + eps), fp32,
 *   clamp to +-(65504 - 1000).  x, out fp16 [m][n]; gamma fp32 [n]; n % 8 == 0.
 * A descriptor with rmsnorm_gamma set has that normalisation applied to its (un-normalised) activation while it is staged
 * -- input_layernorm + q/k/v, post_attention_layernorm + gate/up (Int4llamaDecoderLayer.cu:78, 92-99) as one launch each;
 * M = 1 (decode) only; the linears of a group share gamma and eps like they share A.  Works in tce_w4a16_forward,
 * tce_w4a16_forward_group and plans.  tce_w4a16_forward_group_rmsnorm is the same call with gamma / eps passed separately
 * (the descriptors' own rmsnorm fields are ignored). */
TCE_API int tce_rmsnorm_half(const void *x, const float *gamma, void *out, int m, int n, float eps, void *stream);
TCE_API int tce_w4a16_forward_group_rmsnorm(const tce_w4a16_desc *descs, int count, const float *gamma, float eps, void *stream);
TCE_API int tce_silu_mul_half(void *a, const void *b, long long n, void *stream);

/* Load-time helper for TCE_W4_ZERO_POINT_IS_8: returns 1 if all `n_words` packed zero-point words are 0x88888888, 0 if
 * not, negative on error.  Synchronous; call it once per weight tensor (weights are immutable after loading). */
TCE_API int tce_w4a16_check_zero_point_8(const void *zeros, long long n_words);
/* The same check WITHOUT a synchronisation (round 5; what the C++ adapter uses inside gemv_forward_cuda, whose contract is "asynchronous, never syncs" --
 * kernels/cuda/gemv_cuda.cu:237-251): enqueues one kernel on `stream` that writes 1 (every word is 0x88888888) or 2 (not) to *verdict when it has run.
 * `verdict` must be a word the device can write and the host can read without a copy -- tce_host_alloc memory -- and should be 0 on entry; the caller polls it
 * with plain loads.  Returns after the launch. */
TCE_API int tce_w4a16_check_zero_point_8_async(const void *zeros, long long n_words, int *verdict, void *stream);

/* Load-time re-layout for the prefill GEMM ("offline pre-swizzle to an MFMA-friendly tile layout"; no reference
 * counterpart -- the reference re-runs its GEMV M times).  tce_w4a16_prepack reads qweight / scales / zeros (+ strides) of `d`
 * (q4_6 as loaded from weight_int4.bin / scaling_factor_int4.bin / zero_point_int4.bin; A, C, M are ignored) and writes the
 * q4_mfma copy (csrc/w4a16_mfma_layout.hpp: 16-row x 128-k tiles in MFMA fragment order, nibbles ordered for a 9-instruction
 * exact unpack, per-group effective scales and zero-point constants) into `packed`, which must hold
 * tce_w4a16_prepack_bytes(N, K, group_size) bytes of device memory (0 = shape not supported: K % 128 != 0).  Asynchronous on
 * `stream`; once per weight tensor.  A descriptor whose `prepacked` points at that copy lets tce_w4a16_forward run the 128-row
 * MFMA kernel (csrc/w4a16_gemm_pk.hip) for M > 128 (where its cost model beats the 64-row kernel's); results stay within the W4A16 tolerance of every other path. */
TCE_API size_t tce_w4a16_prepack_bytes(int N, int K, int group_size);
TCE_API size_t tce_w4a16_gemm_scratch_bytes(void); /* size of tce_w4a16_desc.scratch (32 MiB + 4 KiB) */
/* (0.1.11) Exchanges between workgroups on `scratch` that gave up waiting, counted since the area was last zeroed: synchronises `stream`, copies one word.  0 is the only
 * value ever observed; non-zero means some outputs computed with this area since hold NaN (never a plausible wrong number) and the area stays poisoned until its first
 * 4096 bytes are zeroed again.  A host that keeps a scratch area for a long time calls this where it synchronises anyway (the adapter: at teardown and on request). */
TCE_API int tce_w4a16_gemm_scratch_faults(const void *scratch, void *stream, uint32_t *faults);
TCE_API int tce_w4a16_prepack(const tce_w4a16_desc *d, void *packed, void *stream);

/* count (<= TCE_MAX_GROUP) linears with identical M, K, group_size and A/lda, one launch (GEMV path only). */
#define TCE_MAX_GROUP 4
TCE_API int tce_w4a16_forward_group(const tce_w4a16_desc *descs, int count, void *stream);
/* (0.1.12) count (<= TCE_MAX_INDEPENDENT) decode linears that share NOTHING -- each its own activation, K, N, epilogue flags -- as ONE launch.  For the column-sharded
 * model (the one-gather-per-block form): a rank's q / k / v / o / gate / up / down shards of a block read replicated inputs and do not depend on each
 * other; one by one they are launches of 1-6 MB at 8 ranks, each paying a full launch boundary.  No ordering among the linears of the call is implied or provided (a
 * linear that reads another one's output belongs in a later call).  Every output is bit-identical to the same descriptor through tce_w4a16_forward.
 * One launch when every linear has M = 1, group_size 128, K <= 16384, a packed copy (`prepacked`) and no RMSNorm prologue (csrc/w4a16_gemv_i8.hip, the mixed launch);
 * otherwise the linears are issued one after the other on `stream` -- same results.  *launches (may be null) receives the number of kernel launches made. */
#define TCE_MAX_INDEPENDENT 8
TCE_API int tce_w4a16_forward_independent(const tce_w4a16_desc *descs, int count, int *launches, void *stream);
/* How tce_w4a16_forward_independent would run `descs` (no launch, no HIP call): "gemv-i8-mixed waves=<per workgroup> workgroups=<n>" -- one launch -- or
 * "one-by-one launches=<count>". */
TCE_API int tce_w4a16_describe_independent(const tce_w4a16_desc *descs, int count, char *buf, int buf_len);

/*
 * AWQ "CUDA GEMM" (q4_5) layout -- quantize_methods.py:299-368, kernels/cuda/matmul_int4.cu:19-39:
 *   qweight u32 [K][N/8], nibbles 0..7 of word j hold n = 8j + {0,2,4,6,1,3,5,7}; scales fp16 [K/G][N];
 *   zero point fixed 8.  tce_w4a16_awq_fp16acc reproduces naive_mat_mul_fp16_int4 bit-for-bit
 *   (every operation rounded to binary16, sequential over k).  tce_w4a16_gemm_awq is the fast path
 *   (fp32 accumulate): `workspace` must hold tce_w4a16_awq_workspace_bytes(N,K,G) bytes and receives the
 *   q4_6 re-layout of the weights (pass the same workspace again with repack=0 to skip the re-layout).
 */
TCE_API int tce_w4a16_awq_fp16acc(int M, int N, int K, int group_size, const void *A, const void *qweight,
                                  const void *scales, void *C, void *stream);
TCE_API size_t tce_w4a16_awq_workspace_bytes(int N, int K, int group_size);
TCE_API int tce_w4a16_gemm_awq(int M, int N, int K, int group_size, const void *A, const void *qweight,
                               const void *scales, void *C, void *workspace, int repack, void *stream);

/*
 * W8A8 (SmoothQuant) -- kernels/ref/matmul_ref_int8.cc.  A int8 [M][K]; B int8 [N][K] (K contiguous;
 * matmul_params.B.row = K, B.column = N -- llm/src/ops/W8A8B8O8Linear.cc:47-50); acc int32 exact.
 *   out_kind TCE_OUT_INT8: C int8 [M][N] = clamp( (int32) round( (float)acc*alpha [+ (float)bias_i8[n]*beta] ), q_min, q_max )
 *            (round = half away from zero; multiply, multiply, add each rounded separately -- :29-32)
 *   out_kind TCE_OUT_FP32: C fp32 [M][N] = (float)acc*alpha [+ bias_f32[n]]          (:108, :132)
 *   b_per_row != 0: row i of A multiplies its own B_i, B laid out [M][N][K] (the *_batch variants, :79, :153); batch must be 1 (TCE_ERR_UNSUPPORTED_KIND otherwise)
 *   batch > 1: `batch` independent problems at element strides strideA/strideB/strideC (the per-head loop of
 *            llm/src/ops/BMM_S8T_S8N_F32T.cc:45-59 / BMM_S8T_S8N_S8T.cc:47-60 as one launch).
 */
#define TCE_BIAS_NONE 0
#define TCE_BIAS_INT8 1
#define TCE_BIAS_FP32 2
#define TCE_OUT_INT8 0
#define TCE_OUT_FP32 1

typedef struct tce_w8a8_desc {
    int32_t M, N, K;
    int32_t batch;        /* >= 1 */
    const void *A;
    const void *B;
    const void *bias;     /* int8 [N] or fp32 [N] or NULL */
    void *C;
    int64_t strideA, strideB, strideC; /* elements between consecutive batch entries (ignored when batch == 1) */
    float alpha, beta;
    int32_t q_min, q_max; /* C.qparams.q_min/q_max: -128..127, or 0..127 for the fused-ReLU linear (W8A8B8O8LinearReLU.cc:32) */
    int32_t bias_kind, out_kind;
    int32_t b_per_row;
    int32_t accumulate;   /* TCE_OUT_FP32 only: C[m][n] = C[m][n] + result, one more rounding -- the residual add the reference issues behind out_proj / fc2
                             (`add`, llm/src/nn_modules/Int8OPTDecoderLayer.cc:14-22, 39, 54) in the same launch; bit-exact against the two-step form */
    int32_t lda, ldb, ldc; /* row strides in elements, 0 = dense (K, K, N): a head's 64-column slice of a [rows][heads * 64] projection output is an
                             operand as it lies (strideA = 64, lda = heads * 64), a cache with room for max_keys keys likewise; b_per_row with
                             batch == 1 and strideB != 0: the per-row B_m are strideB elements apart */
    int32_t reserved2;
} tce_w8a8_desc;

TCE_API int tce_w8a8_matmul(const tce_w8a8_desc *d, void *stream);
/* size-prefixed form (0.1.10), as tce_w4a16_desc_v2 */
typedef struct tce_w8a8_desc_v2 {
    uint32_t struct_size;
    uint32_t reserved0;
    tce_w8a8_desc desc;
    void *scratch;  /* NULL = none.  tce_w8a8_scratch_bytes() bytes of device memory, 256-byte aligned, its first 4096 bytes ZEROED once by the caller (every call leaves
                       them zero): lets a launch with few 64 x 64 tiles and a long k chain (OPT's fc2 at prefill: 512 x 768 x 3072 is 96 tiles of 48 steps on 256 CUs) cut
                       the chain into runs on several workgroups; the int32 partial tiles are added exactly, in any order -- the result stays bit-exact.  One area per
                       stream that runs such calls concurrently */
} tce_w8a8_desc_v2;
TCE_API int tce_w8a8_matmul_v2(const tce_w8a8_desc_v2 *d, void *stream);
TCE_API size_t tce_w8a8_scratch_bytes(void);

/* The element-wise steps between the two int8 BMMs of the reference's OPT attention (llm/src/nn_modules/Int8OPTAttention.cc:254-268), as ONE launch:
 *   batch_Add (llm/src/ops/batch_add.cc:3-24): s[h][j][k] + mask[j][k];  softmax over k (llm/src/ops/softmax.cc:5-40: the running maximum starts
 *   from `m_data[0]` -- element [0][0][0] of the tensor the reference normalises IN PLACE, i.e. the masked score for row (0, 0) and that row's first
 *   probability for every later row --, exponentials summed in k order in fp32, the quotient formed in double: exp / (sum + 1e-10));
 *   attn_probs_int8 = (int8) std::round(p * 127)  (:266).
 * scores fp32 [heads][sq][tgz] (the qk BMM's output), mask fp32 [sq][tgz], probs int8 [heads][sq][ld_probs] (ld_probs 0 = tgz; a multiple of 16
 * lets the pv BMM take its vector path).  Same operations in the same order; the device's expf may differ from the host's in the last bit. */
TCE_API int tce_opt_softmax_q(const float *scores, const float *mask, void *probs, int heads, int sq, int tgz, int ld_probs, void *stream);
/* The KV append of Int8OPTAttention::forward (:205-236; the reference copies the whole past into the other cache buffer per token): the new
 * key / value rows k, v int8 [sq][heads * hd] (k_proj / v_proj outputs as they lie) go to rows [pos, pos + sq) of k_cache [heads][max_keys][hd]
 * and, TRANSPOSED (the pv BMM's operand, :271-273 without the per-token transpose of the whole cache), to columns [pos, pos + sq) of
 * vt_cache [heads][hd][max_keys]. */
TCE_API int tce_opt_kv_append(const void *k, const void *v, void *k_cache, void *vt_cache, int heads, int hd, int sq, int pos, int max_keys, void *stream);
/* The whole OPT attention of a DECODE step (m <= 8 new rows) between the q/k/v projections and out_proj as ONE launch (csrc/opt_attention.hip): the KV append
 * of tce_opt_kv_append, the qk BMM (BMM_S8T_S8N_F32T.cc:12-62), tce_opt_softmax_q and the pv BMM (BMM_S8T_S8N_S8T.cc:12-63, clamp -128 .. 127) -- int32 dot
 * products and the same floating-point operations in the same order, so `out` and both caches are bit-identical to the four separate launches.
 *   q, k_new, v_new  int8 [m][ld]: the projections' output rows (head h: columns h * hd ..; ld 0 = heads * hd)
 *   k_cache int8 [heads][max_keys][hd], vt_cache int8 [heads][hd][max_keys]: rows / columns pos .. pos + m - 1 are written
 *   mask fp32 [m][pos + m] additive;  out int8 [m][ld]: head h writes its hd columns (out_proj's input rows)
 * hd == 64 (OPT-125M / 1.3B) or 128 (OPT-6.7B), m <= 8, ld and max_keys multiples of 16. */
TCE_API int tce_opt_attention_decode(const void *q, const void *k_new, const void *v_new, void *k_cache, void *vt_cache, const float *mask, void *out, int heads,
                                     int head_dim, int m, int pos, int max_keys, int ld, float alpha_qk, float alpha_pv, void *stream);

/* LayerNormQ::forward (llm/src/ops/LayerNormQ.cc:12-52), the op in front of the W8A8 linears: x fp32 [m][n],
 * weight / bias fp32 [n], out int8 [m][n] = (int8) round((x - mean) / sqrt(var + 1e-5) * weight + bias), sums sequential
 * in fp32 exactly like the reference loop: BIT-EXACT.  n % 4 == 0, n <= 8192. */
TCE_API int tce_layernorm_q(const float *x, const float *weight, const float *bias, void *out, int m, int n, void *stream);

/* LayerNormQ and the int8 linears that read its output as ONE launch, for decode: replaces
 *   self_attn_layer_norm.forward + q_proj / k_proj / v_proj.forward   (llm/src/nn_modules/Int8OPTAttention.cc:186-201 behind LayerNormQ.cc:12-52)
 *   final_layer_norm.forward + fc1.forward                             (LayerNormQ + W8A8B8O8LinearReLU)
 * x fp32 [m][k] (m <= 8), ln_weight / ln_bias fp32 [k]; `linears[i]` (count <= TCE_MAX_GROUP) describe the consumers exactly as for
 * tce_w8a8_matmul with M = m, K = k, batch = 1, b_per_row = 0 -- their `A` is ignored (it is the normalised row, which never
 * leaves the chip unless ln_out, int8 [m][k], is given).  BIT-EXACT against tce_layernorm_q followed by tce_w8a8_matmul (and so
 * against the reference).  k % 16 == 0. */
TCE_API int tce_layernorm_q_w8a8_group(const float *x, const float *ln_weight, const float *ln_bias, int m, int k, const tce_w8a8_desc *linears,
                                       int count, void *ln_out, void *stream);

/* ---- replayable plan: a fixed sequence of W4A16 launches captured into one hipGraph ---- */
typedef struct tce_plan tce_plan;
/* group_sizes[i] consecutive descriptors form launch i (1 = tce_w4a16_forward, >1 = tce_w4a16_forward_group). */
TCE_API int tce_plan_create(const tce_w4a16_desc *descs, const int32_t *group_sizes, int n_launches, tce_plan **out);
/* TCE_PLAN_TAGGED (TCE_PLAN_CHAINED is accepted as a synonym): ONE persistent kernel walks the whole launch list -- no kernel
 * boundaries, no barriers.  Every output of a launch is ALSO written as one 32-bit word (token tag << 16 | fp16 bits,
 * device-scope store) into a shadow vector the plan owns, and a launch whose activation vector is (a 16-byte-aligned slice
 * of) an output of an earlier launch of the plan polls those words -- the poll is its activation read; its first weight
 * tiles are requested before it starts to wait.  A launch whose activations come from outside the plan reads them at once.
 * What is ordered is therefore the plan's DATA FLOW (and, transitively, everything in front of it), not the launch list as
 * such: launches that do not feed each other may overlap.  Outputs are bit-identical to the stream-ordered plan.
 * Round 6: when the linears carry packed copies (tce_w4a16_prepack) and TCE_W4_ZERO_POINT_IS_8, the walk runs on the int8-contraction body of the decode GEMV
 * (csrc/w4a16_gemv_i8_token.hip; tce_plan_is_chained = 4): the list's longest prefix of M = 1, group-128, K <= 15360 launches goes into ONE kernel, what is left (e.g. a
 * linear with general zero points) follows it as ordinary launches of the same graph.  Bit-identical to the stream-ordered plan; measured level with it (0.976 vs 0.962 ms
 * per Llama-3-8B token, profiles/r6/i8_token_kernel.md) -- opt-in.
 * Not taken (the plan is then built stream-ordered; tce_plan_is_chained tells: 0 stream-ordered, 2 token kernel on the fp16 body, 4 on the int8 body): a launch that
 * is not an M = 1 GEMV the persistent kernel takes; an activation vector that straddles two outputs or starts at an odd
 * 16-byte offset; a launch list in which a launch overwrites memory that an earlier launch reads un-tagged (activations from
 * outside the plan, the old value of TCE_W4_ADD_TO_C) or also writes WITHOUT being downstream of that launch in the data
 * flow -- stream order would protect such a hazard by position, polls do not.  tce_plan_status synchronises the device and returns TCE_ERR_HIP
 * if a wait ever timed out (~0.3 s; cannot happen unless the device is shared with a kernel that never ends).
 * tce_plan_geometry reports what the token kernel runs with (rows per row group, ring depth, waves per workgroup, workgroups).
 * Measured (MI355X, Llama-2-7B-shaped token): the primitive is cheap -- 1.8 us per bare hand-off against 2.0 us for a kernel
 * boundary and 5.0 us for round 1's arrival-counter barrier (scripts/probes/handoff_probe.hip) -- but the token is NOT faster
 * than the stream-ordered graph (1.40 vs 0.995 ms): opt-in, and bench.py picks whichever is faster. */
#define TCE_PLAN_CHAINED 1
#define TCE_PLAN_TAGGED 2
/* TCE_PLAN_OVERLAPPED: the same tagged data flow, but every launch stays a kernel of its own; the launches are issued on two (tunable)
 * alternating graph branches with NO edge between consecutive launches, so launch j+1's workgroups are dispatched, request their first
 * weight steps and poll their activation words while launch j is still running (csrc/w4a16_gemv_ovl.hip).  A launch's grid is capped at
 * 1 / branches of what the chip holds of that kernel, so waiting workgroups cannot keep their producers off the chip.  Same acceptance
 * rules as TCE_PLAN_TAGGED (plus: no fused RMSNorm prologue yet); tce_plan_is_chained returns 3.  Outputs bit-identical to the
 * stream-ordered plan. */
#define TCE_PLAN_OVERLAPPED 4
/* TCE_PLAN_TUNED (stream-ordered plans): the geometry of every decode (M = 1) launch is chosen at plan creation by timing the compiled candidates on this
 * device -- the WHOLE launch list is captured and replayed with one group of same-shaped launches at a time on each compiled candidate, a candidate stays
 * only if the whole plan gets 0.7 % faster; outputs are redirected to a scratch buffer (the caller's buffers are not written).  Costs a second or two, once; results are bit-identical to the
 * untuned plan's (only geometries that keep every row's summation order are candidates). */
#define TCE_PLAN_TUNED 8
/* (0.1.12) TCE_PLAN_INDEPENDENT: every group of the list is a set of linears that share NOTHING (tce_w4a16_forward_independent: group sizes up to
 * TCE_MAX_INDEPENDENT, own activation / K / N / flags per linear) -- a rank's shards of one transformer block in the one-gather-per-block form.  Stream-ordered plan;
 * not combinable with the other flags. */
#define TCE_PLAN_INDEPENDENT 16
TCE_API int tce_plan_create_ex(const tce_w4a16_desc *descs, const int32_t *group_sizes, int n_launches, int flags, tce_plan **out);
TCE_API int tce_plan_is_chained(const tce_plan *plan);
TCE_API int tce_plan_geometry(const tce_plan *plan, int *rows, int *depth, int *waves, int *workgroups);
/* TCE_PLAN_TUNED plans: the geometry that won the timing for launch `launch` -- (rows per wave, waves along N, waves splitting K, pipeline depth) of the
 * row-block GEMV -- or all zero where the dispatcher's own choice stayed (also for untuned plans); depth + 200: the launch forms its activation sums
 * by every wave instead of once per workgroup (the default for M = 1); + 1000 / + 2000: its issue order was forced to activations-first / weights-first (the other knobs the timing tries). */
TCE_API int tce_plan_launch_geometry(const tce_plan *plan, int launch, int *rows, int *waves_n, int *waves_k, int *depth);
TCE_API int tce_plan_status(tce_plan *plan);
TCE_API int tce_plan_launch(tce_plan *plan, void *stream);
TCE_API int tce_plan_n_launches(const tce_plan *plan);
TCE_API void tce_plan_destroy(tce_plan *plan);

/* ---- multi-GPU: column shards + peer-write all-gather (no reference counterpart, the reference is single-device) ----
 * One process per GPU.  Every linear is sharded by OUTPUT ROWS: rank r of P computes rows [r*N/P, (r+1)*N/P) from the replicated
 * activation; tce_w4a16_shard fills the shard's descriptor (in q4_6 a row range is one contiguous byte range of qweight, scales and
 * zeros; N % P == 0 and (N / P) % 16 == 0, the reference wrapper's own divisibility rule, linear.cu:16-17); A, C, M, flags and the
 * strides are copied, C must then be pointed at the rank's slice buffer [M][N/P].
 * The slices are joined by tce_allgather_f16: at decode sizes (1-4 KB per rank) a latency-bound exchange, done as ONE small kernel
 * per call that writes the slice into every rank's window over xGMI, publishes an epoch flag, waits for the other ranks' flags
 * and copies the complete vector to `dst_full` (csrc/comm.hip).  The call is asynchronous, stream-ordered and capturable into a
 * hipGraph (epochs are counted on the device).  Every rank must issue the same sequence of calls per slot.
 *   tce_comm_create   rank's window for vectors of up to max_vector_elems halves, `slots` independent exchange slots
 *   tce_comm_export / tce_comm_connect   the host all-gathers the 64-byte handles (MPI, torch.distributed, a file ...) and hands every
 *                     rank the table [world][64]; tce_comm_connect_local instead when all ranks live in one process (tests)
 *                     (ONE host thread driving several devices, the reference's host shape -- Int4llamaForCausalLM.cu:40-44: create each rank's
 *                     communicator with that rank's device current; tce_comm_connect_local enables peer access between the devices, and
 *                     tce_allgather_f16 / tce_comm_status / tce_comm_reset make the communicator's device current for their own calls and restore
 *                     the caller's; the stream passed to tce_allgather_f16 must be a stream of the communicator's device)
 *   tce_comm_status   synchronous: 1 if a wait ever timed out (a rank never arrived within the bound: 2 s, tce_comm_set_timeout_ms), 0 if not,
 *                     negative on error; a flagged communicator's later exchanges do not wait any more (their outputs are void): one lost exchange
 *                     costs one bound, not one bound each.  A host that uses the outputs checks the status (per token, or per batch of tokens).
 *   tce_comm_reset    re-arms a flagged communicator.  Every rank calls it after the host made sure no exchange is in flight anywhere
 *                     (barrier + device synchronisation); the per-slot epochs are kept -- every rank's gather kernels ran to their end.
 *   tce_comm_export refuses a window that is not fine-grained memory (the flags are polled across the link; TCE_ERR_UNSUPPORTED_SHAPE).
 * Prefill (M >= 17) moves megabytes per exchange: use RCCL there (ncclAllGather; this repository's host code does, through
 * torch.distributed) -- the peer-write kernel is a single workgroup. */
typedef struct tce_comm tce_comm;
#define TCE_COMM_HANDLE_BYTES 64
#define TCE_COMM_MAX_RANKS 8
TCE_API int tce_w4a16_shard(const tce_w4a16_desc *full, int rank, int world, tce_w4a16_desc *shard);
TCE_API int tce_comm_create(int rank, int world, int max_vector_elems, int slots, tce_comm **out);
TCE_API int tce_comm_export(tce_comm *comm, void *handle_out /* 64 bytes */);
TCE_API int tce_comm_connect(tce_comm *comm, const void *handles /* [world][64], rank order */);
TCE_API int tce_comm_connect_local(tce_comm *comm, tce_comm *const *all_ranks /* [world] */);
TCE_API int tce_allgather_f16(tce_comm *comm, int slot, const void *src_slice, void *dst_full, int n_total, void *stream);
/* (0.1.12) tce_w4a16_forward_independent with the all-gather of ONE of its linears INSIDE the launch (the one-gather-per-block form as one launch per block:
 * layers + 1 launches per token).  Linear `gathered` of the call is this rank's slice (its N rows; M = 1) of a vector of N * world halves: besides storing the slice at
 * its own C, the tiles that compute it write their 16 values straight into every rank's window, the tile that arrives last publishes this rank's flag, waits for the
 * other ranks' flags and copies the complete vector to `dst_full` -- tce_allgather_f16's protocol, window, buffers, flags and epochs (the two calls may alternate on one
 * slot), without its launch.  Same contract: every rank issues the same sequence of exchanges per slot; a rank that never arrives flags the communicator after the bound
 * (tce_comm_status).  The other linears of the call are not delayed by the exchange (one wave of the launch waits).
 * One launch under tce_w4a16_forward_independent's conditions (+ the gathered linear has no SiLU-mul-pair epilogue, N * world % 8 == 0 and <= 16384, the vector fits the window);
 * otherwise tce_w4a16_forward_independent followed by tce_allgather_f16 -- same results.  *launches (may be null): kernel launches made. */
TCE_API int tce_w4a16_forward_independent_gather(const tce_w4a16_desc *descs, int count, int gathered, tce_comm *comm, int slot, void *dst_full, int *launches, void *stream);
/* Round 4: RCCL behind the same communicator, for exchanges beyond the latency regime (north star: "a single RCCL all-gather over xGMI per transformer block";
 * SURVEY 8e sizes a prompt's exchange at 0.65-1.97 MB per rank).  librccl is opened on first use (dlopen); the communicator is built from an opaque
 * TCE_RCCL_ID_BYTES blob that rank 0 obtains (tce_comm_rccl_unique_id) and the host hands to every rank exactly like the IPC handles (tce_comm_rccl_init is
 * collective: every rank calls it; one rank per device -- RCCL refuses two ranks on one GPU).  tce_allgather_f16 then dispatches by size: slices of at most
 * 64 KiB that fit the window -> the peer-write kernel (one launch, graph-capturable); larger -> ncclAllGather on `stream`.
 *   tce_allgather_rows_f16: the column-sharded outputs of M > 1 rows (a sharded prompt): src fp16 [M][n_total / world] (this rank's columns), dst fp16 [M][ldd]
 *   (ldd = 0: n_total) on every rank.  M = 1 is tce_allgather_f16.  M > 1: the ranks' whole blocks are gathered rank-major into `workspace`
 *   (tce_allgather_rows_workspace_bytes(M, n_total) bytes, 16-byte aligned; peer-write kernel or RCCL by size) and one kernel lays the rows side by side. */
#define TCE_RCCL_ID_BYTES 128
TCE_API int tce_comm_rccl_unique_id(void *id_out /* TCE_RCCL_ID_BYTES */);
TCE_API int tce_comm_rccl_init(tce_comm *comm, const void *id /* TCE_RCCL_ID_BYTES */);
TCE_API size_t tce_allgather_rows_workspace_bytes(int M, int n_total);
TCE_API int tce_allgather_rows_f16(tce_comm *comm, int slot, const void *src, void *dst, int M, int n_total, int ldd, void *workspace, void *stream);
TCE_API int tce_comm_status(tce_comm *comm);
TCE_API int tce_comm_set_timeout_ms(tce_comm *comm, int milliseconds); /* 1 .. 600000; takes effect for exchanges enqueued (or captured) afterwards */
TCE_API int tce_comm_reset(tce_comm *comm);
TCE_API int tce_comm_device(const tce_comm *comm);
TCE_API void tce_comm_destroy(tce_comm *comm);

/* ---- memory / sync helpers: what the L2 wrappers need from the runtime ----
 * tce_malloc(managed=1) / tce_free replace allocate_aligned_memory_gpu / free_aligned_memory_gpu
 * (llm/src/nn_modules/cuda/utils.cu:92-103, cudaMallocManaged there: model files are read straight into that memory,
 * llm/include/common.h:111-120).  managed=0 gives plain device memory (preferred on MI355X; fill it with tce_memcpy).
 * tce_synchronize(NULL) is the per-forward device sync (Int4llamaForCausalLM.cu:40-44 / tests' cudaDeviceSynchronize). */
#define TCE_MEMCPY_H2D 0
#define TCE_MEMCPY_D2H 1
#define TCE_MEMCPY_D2D 2
TCE_API int tce_malloc(void **ptr, size_t bytes, int managed);
TCE_API int tce_free(void *ptr);
/* Pinned, device-mapped, coherent host memory (hipHostMalloc): small words the device writes and the host polls (see tce_w4a16_check_zero_point_8_async). */
TCE_API int tce_host_alloc(void **ptr, size_t bytes);
TCE_API int tce_host_free(void *ptr);
TCE_API int tce_memcpy(void *dst, const void *src, size_t bytes, int kind, void *stream);
TCE_API int tce_synchronize(void *stream);
TCE_API int tce_device_count(void);

/* ---- introspection / tuning (not part of the reference surface) ---- */
TCE_API int tce_version(void);
TCE_API const char *tce_last_error(void);
TCE_API const char *tce_build_info(void);
/* Drops the HIP runtime's sticky last error (and this library's message).  For a host that recovers from a failure of
 * its own -- e.g. a stream capture that was invalidated: the runtime keeps reporting hipErrorStreamCaptureInvalidated as the
 * "last error", and the next launch's check here would return it as TCE_ERR_HIP.  Returns the HIP error code it dropped. */
TCE_API int tce_reset_last_error(void);
/* Which kernel family (and, for the GEMM, which tile and form) tce_w4a16_forward would run for this descriptor, as text:
 * "gemv passes=P kernel=row-block|persistent" | "gemv-i8 rows-per-pass=R group=G" | "small-batch slices=S" | "gemm-pk tile=128xC quartets=Q group=G" |
 * "gemm-dma tile=RxC quartets=Q group=G" | "gemm tile=RxC".  Launches nothing and
 * makes no HIP call (works without a GPU); a shape the chosen GEMM form cannot hold in LDS still falls back at launch time. */
TCE_API int tce_w4a16_describe_dispatch(const tce_w4a16_desc *d, char *buf, int buf_len);
/* (0.1.13) The same for tce_w8a8_matmul / tce_w8a8_matmul_v2 (with_scratch != 0: as if tce_w8a8_desc_v2.scratch were given): "w8a8 wave-per-column rows=M" |
 * "w8a8 wave-per-output (a B per row of A)" | "w8a8 tile=128xC quartets=Q" | "w8a8 k-slice tile=RxC waves=W workgroups=G" (the whole tile in every wave, the k-steps
 * dealt to the waves) | "w8a8 tile=64x64 deep-pipeline quartets=Q" | "w8a8 tile=64x64 quartets=Q [kcut=S]" | "w8a8 tile=32x64 quartets=Q" | "w8a8 generic ...".
 * Operand pointers enter through their 16-byte alignment only (null pointers: aligned).  Launches nothing, makes no HIP call. */
TCE_API int tce_w8a8_describe_dispatch(const tce_w8a8_desc *d, int with_scratch, char *buf, int buf_len);
/* Tuning and diagnostics entry points (forced kernel forms for parity tests and sweeps, per-family setters, the debug buffer): include/tce_tuning.h -- exported by
 * this library, never needed by a host, not part of the operator boundary. */
/* Algorithmic HBM bytes of one tce_w4a16_forward call: N*K/2 + 2*N*K/G + N*K/(2G) + 2*M*K + 2*M*N. */
TCE_API int64_t tce_w4a16_algorithmic_bytes(int M, int N, int K, int group_size);

#ifdef __cplusplus
}
#endif
#endif /* TCE_MATMUL_H */
