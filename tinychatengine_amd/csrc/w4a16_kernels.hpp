// w4a16_kernels.hpp -- host-visible declarations shared by the W4A16 translation units.
#pragma once
#include "tce_common.hpp"

namespace tce {

// One linear of a (possibly grouped) GEMV launch.
struct GemvSeg {
    const uint4_t *qweight;  // u32 [N][K/8] viewed as 16-byte chunks: [N][K/32]
    const half_t *scales;    // fp16 [N][scales_stride]
    const unsigned *zeros;   // u32 [N][zeros_stride]
    half_t *C;               // fp16 [M][ldc]
    int N, ldc, scales_stride, zeros_stride;
    int bytes_w, bytes_s, bytes_z;  // extents for the buffer descriptors (each < 2 GiB)
    int block_begin;         // first blockIdx.x of this linear
    int epilogue;            // TCE_W4_SILU_MUL_PAIRS / TCE_W4_ADD_TO_C bits of the descriptor
};

struct GemvArgs {
    const half_t *A;  // fp16 [M][lda]
    int lda, M, K, log2g;
    int nseg;
    unsigned long long *dbg;  // MODE 2 (timestamps) only
    int zeros_are_8;          // every linear of the launch carries TCE_W4_ZERO_POINT_IS_8
    const float *gamma;       // non-null: stage RMSNorm(A) * gamma instead of A (fused prologue, M = 1)
    float eps;
    int shared_xsum;          // the per-chunk activation sums are computed once per workgroup (table in LDS) instead of by every wave in every step
    GemvSeg seg[TCE_MAX_GROUP];
};

// (rows per wave, waves along N, waves splitting K, pipeline depth in steps) -- every tuple listed here is compiled
// for MB in {1,2,4}; tce_w4a16_set_gemv_config() can force any of them, the dispatcher picks by shape otherwise.
#define TCE_GEMV_VARIANTS(X) \
    X(1, 4, 1, 1)            \
    X(2, 4, 1, 1)            \
    X(4, 4, 1, 1)            \
    X(1, 4, 1, 2)            \
    X(2, 4, 1, 2)            \
    X(4, 4, 1, 2)            \
    X(2, 4, 1, 3)            \
    X(3, 4, 1, 1)            \
    X(3, 4, 1, 2)            \
    X(2, 8, 1, 2)            \
    X(4, 8, 1, 1)            \
    X(1, 2, 2, 1)            \
    X(2, 2, 2, 1)            \
    X(2, 2, 2, 2)            \
    X(1, 2, 4, 1)

bool gemv_variant_exists(int rows, int wn, int wk, int depth);
void set_gemv_debug_mode(int mode);
void set_gemv_order(int force);
void set_w8a8_deep(int d);          // W8A8 64 x 64 tile with 8 k-steps in flight: 0 the rule, 1 / 2 / 4 quartets forced, 9 off
void set_w8a8_rows32(int r);        // W8A8 32 x 64 tiles (round 6): 0 the rule, 1 forced, 2 off
void set_w8a8_kslice(int code);    // W8A8, the whole tile in every wave (round 6): 0 the rule, 1 off, else a forced form (w8a8_gemm.hip)
void set_w8a8_big(int b);           // W8A8 128-row tiles: 0 the rule, 1 / 2 forced (128 / 64 columns), 9 off
void set_lnq_stamps(void *p);
void set_lnq_form(int f);           // 1: the workgroup-per-8-rows form of the LayerNormQ + W8A8 launch at every k
void set_gemv_shared_xsum(int on);  // 0 the rule (on for M = 1), 1 on, 2 off
void set_gemv_debug_buffer(void *p);

// persistent form, w4a16_gemv_stream.hip
// One GEMV launch (up to TCE_MAX_GROUP linears sharing the activation) as the persistent kernels read it.
struct StreamLaunch {
    const half_t *A;  // fp16 [K]
    int K, log2g;
    int nseg;
    int n_rg;                    // row groups over all linears of the launch
    GemvSeg seg[TCE_MAX_GROUP];  // block_begin = first row group of the linear
    const float *gamma;          // non-null: stage RMSNorm(A) * gamma instead of A (generalT5LayerNorm arithmetic)
    float eps;
    // tagged token plans only (w4a16_gemv_stream.hip): the activation as (tag << 16 | fp16 bits) words written by the launch of
    // the same plan that produces it (null: A comes from outside the plan, read at once), and where this launch's outputs go in
    // that form (per linear; null: nobody inside the plan reads them)
    const unsigned *A_tag;
    unsigned *C_tag[TCE_MAX_GROUP];
};
void set_gemv_stream_config(int rows, int nw, int depth);
void set_gemv_stream_debug(int mode, void *buf);
bool gemv_stream_supports(const tce_w4a16_desc *descs, int count);
int launch_w4a16_gemv_stream(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err,
                             const float *gamma = nullptr, float eps = 0.f);
// token plans: a list of launches walked by ONE persistent kernel, the data flow between them ordered by tagged output words
struct TokenPlan;
// mode 0: the token kernel (one persistent kernel walks the list); mode 1: one kernel per launch on alternating graph branches, ordered
// by the same tagged words (w4a16_gemv_ovl.hip)
int token_plan_create(const tce_w4a16_desc *descs, const int32_t *groups, int n_launches, TokenPlan **out, hipError_t *hip_err, int mode = 0);
int token_plan_mode(const TokenPlan *tp);
// w4a16_gemv_ovl.hip: the per-launch kernel of overlapped plans
struct OvlGeom {
    int wn, xb, depth, rows_per_block, grid, per_cu;
    bool z8;
    size_t lds;
};
int ovl_chains();
int ovl_geometry(const tce_w4a16_desc *descs, int count, int cus, int chains, OvlGeom *g, hipError_t *hip_err);
hipError_t ovl_enqueue(const StreamLaunch *L, const OvlGeom &g, const unsigned *epoch, unsigned *status, int launch_index, hipStream_t stream);
void set_gemv_ovl_config(int depth, int chains);
void set_gemv_ovl_stamps(void *buf);
int token_plan_enqueue(TokenPlan *tp, hipStream_t stream, hipError_t *hip_err);
int token_plan_status(TokenPlan *tp, unsigned *status, hipError_t *hip_err);
void token_plan_geometry(const TokenPlan *tp, int *rows, int *depth, int *waves, int *blocks);
void token_plan_destroy(TokenPlan *tp);
int launch_w4a16_gemv(const tce_w4a16_desc *descs, int count, int forced_rows, int forced_wn, int forced_wk,
                      int forced_depth, hipStream_t stream, hipError_t *hip_err, const float *gamma = nullptr, float eps = 0.f);

// w4a16_gemv_i8.hip: the decode GEMV (M <= 4) as an int8 contraction on the pre-packed copy
bool gemv_i8_supports(const tce_w4a16_desc *descs, int count, bool with_norm = false);
int gemv_i8_rows_per_pass(int M, int K, int group_size);
void set_gemv_i8_mode(int mode, int rows);  // mode 0 automatic (taken wherever a packed copy comes with the descriptors), 1 off; rows 0 the rule, 1 / 2 tiles per wave
struct I8ResidualNorm {  // tce_w4a16_forward_residual_rmsnorm: the RMSNorm that follows the residual add, produced by the same launch
    const float *gamma;
    float eps;
    void *xn_out;
    void *workspace;  // float [2048] + unsigned counter (zero between launches)
};
struct AttnDeferred;
int launch_w4a16_gemv_i8(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err, const float *gamma = nullptr, float eps = 0.f, const I8ResidualNorm *rn = nullptr,
                         const AttnDeferred *comb = nullptr, const int *comb_pos_dev = nullptr, int comb_pos = 0);

// the mixed launch (round 6): up to TCE_MAX_INDEPENDENT decode linears with their own activations and K, one launch; bit-identical to the linears issued one by one
struct PeerGatherEpi;  // (below, with the communicator)
bool gemv_i8_mixed_supports(const tce_w4a16_desc *descs, int count);
void gemv_i8_mixed_geometry(const tce_w4a16_desc *descs, int count, int *waves, int *workgroups);
void set_gemv_i8_mixed_waves(int w);  // tuning: 0 the rule
int launch_w4a16_gemv_i8_mixed(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err, const PeerGatherEpi *gather = nullptr, int gathered = -1);
// w4a16_gemv_i8_token.hip (round 6): a prefix of a launch list as ONE persistent kernel on the int8-contraction body, the data flow ordered by tagged output words
struct I8TokenPlan;
int i8_token_plan_create(const tce_w4a16_desc *descs, const int32_t *groups, int n_launches, I8TokenPlan **out, int *n_taken, hipError_t *hip_err);
int i8_token_plan_enqueue(I8TokenPlan *tp, hipStream_t stream, hipError_t *hip_err);
int i8_token_plan_status(I8TokenPlan *tp, unsigned *status, hipError_t *hip_err);
int i8_token_plan_stages(const I8TokenPlan *tp);
int i8_token_plan_blocks(const I8TokenPlan *tp);
void i8_token_plan_destroy(I8TokenPlan *tp);
void set_i8_token_stamps(void *buf);  // non-null: plans built from now on record [workgroup][stage][8] wall-clock stamps there (scripts/token_timeline.py)
void set_i8_token_max_units(int u);   // a stage with more units (16-row tile x 1024-k chunk) per workgroup ends the prefix the kernel takes (default 512)
void set_i8_token_mode(int mode);     // 0: tagged plans take this kernel where the list allows (default), 1: never (round 2's token kernel on the fp16 body)

// MFMA GEMM on the q4_6 layout (prefill).  m_tiles x n_tiles 16x16 MFMA tiles per wave, 4 waves along N.
#define TCE_GEMM_VARIANTS(X) \
    X(8, 1)                  \
    X(8, 2)                  \
    X(4, 2)                  \
    X(4, 1)                  \
    X(2, 2)                  \
    X(4, 4)                  \
    X(2, 4)
bool gemm_variant_exists(int m_tiles, int n_tiles);
void set_gemm_xcd_rows(int xm);
void set_gemm_dma_xcd_rows(int xm);
float gemm_dma_estimate_us(int M, int N, int K);
void gemm_dma_describe(int M, int N, bool g128, int *mt, int *nt, int *ks);  // the tile and form launch_w4a16_gemm_dma would pick  // the GEMM dispatcher's cost model (rounds x k-blocks x us per k-block)
void set_gemm_dma_mode(int mode);
void set_w8a8_ksplit(int ks);  // w8a8_gemm.hip tuning: wave quartets per tile (0 = automatic)  // timing experiments, see w4a16_gemm_dma.hip
// w4a16_gemm_dma.hip: same contract as launch_w4a16_gemm with an explicit tile; TCE_ERR_UNSUPPORTED_SHAPE when the shape
// (alignment, table size) does not fit, so the caller can fall back
int launch_w4a16_gemm_dma(const tce_w4a16_desc &d, int m_tiles, int n_tiles, hipStream_t stream, hipError_t *hip_err);  // tuning: XCD grid of xm x (8 / xm) over (row blocks x column blocks)
int launch_w4a16_gemm(const tce_w4a16_desc &d, int forced_mt, int forced_nt, hipStream_t stream, hipError_t *hip_err);

// w4a16_gemm_pk.hip: the 128-row GEMM on pre-packed weights (q4_mfma, w4a16_mfma_layout.hpp)
size_t prepack_bytes(int N, int K, int G);
int launch_w4a16_prepack(const tce_w4a16_desc &d, void *out, hipStream_t stream, hipError_t *hip_err);
int launch_w4a16_gemm_pk(const tce_w4a16_desc &d, const void *packed, hipStream_t stream, hipError_t *hip_err);
void set_gemm_pk256_auto(int on);  // 0: the dispatcher never picks the 256-row forms by itself
void set_gemm_pk_wide_auto(int on);
void set_gemm_pk_form16_auto(int on);
void set_gemm_pk_handoff_delta(int d);
void set_gemm_pk_prio(int on);
void set_gemm_pk_handoff(int on);    // 0: a two-run k cut exchanges through the last arriver (A/B), 1: run 0 hands its tile to run 1  // 1: the dispatcher may pick the wide forms (128 rows x 64 columns per wave)
float gemm_pk_estimate_us(int M, int N, int K, int *form_out, bool has_scratch = false, int *split_out = nullptr, int group_size = 128, bool zero_point_8 = false);
size_t gemm_pk_scratch_bytes();
void set_gemm_pk_split(int s);
void set_gemm_pk_mode(int ks, int xm);
void set_gemm_pk_ablation(int abl);  // timing experiments (results meaningless): see w4a16_gemm_pk_kernel  // tuning: forced wave quartets per tile / XCD rows (0 = automatic)

int check_zero_point_8(const void *zeros, long long n_words, hipError_t *hip_err);
int check_zero_point_8_async(const void *zeros, long long n_words, int *verdict, hipStream_t stream, hipError_t *hip_err);  // one launch, no synchronisation: *verdict (host-mapped) = 1 / 2

// AWQ (q4_5) helpers
int launch_awq_fp16acc(int M, int N, int K, int G, const void *A, const void *qweight, const void *scales, void *C,
                       hipStream_t stream, hipError_t *hip_err);
int launch_awq_repack(int N, int K, int G, const void *qweight, const void *scales, void *workspace, hipStream_t stream,
                      hipError_t *hip_err);

// small batches, 2 <= M <= 16 (w4a16_skinny.hip)
bool skinny_supports(const tce_w4a16_desc &d);
void set_skinny_config(int ks);
void set_skinny_max_m(int m);  // largest M the small-batch kernel takes (16-row slices of the batch on gridDim.y)  // tuning: waves per 16-row tile, 0 = automatic
int launch_w4a16_skinny(const tce_w4a16_desc &d, hipStream_t stream, hipError_t *hip_err);

// element-wise glue of the decoder layer (glue.hip)
int launch_layernorm_q(const float *x, const float *w, const float *b, void *out, int m, int n, hipStream_t stream, hipError_t *hip_err);
// w8a8_lnq_fused.hip
int launch_lnq_w8a8_group(const float *x, const float *ln_w, const float *ln_b, int m, int k, const tce_w8a8_desc *lin, int count, void *ln_out,
                          hipStream_t stream, hipError_t *hip_err);
int launch_opt_softmax_q(const float *scores, const float *mask, void *probs, int heads, int sq, int tgz, int ldp, hipStream_t stream, hipError_t *hip_err);
int launch_opt_kv_append(const void *k, const void *v, void *kc, void *vt, int heads, int hd, int sq, int pos, int max_keys, hipStream_t stream, hipError_t *hip_err);
int launch_rmsnorm_half(const void *x, const float *gamma, void *out, int m, int n, float eps, hipStream_t stream, hipError_t *hip_err);
// attention_ops.hip
int launch_bmm_f16t(const void *A, const void *B, void *C, int batch, int M, int N, int K, unsigned short alpha_bits, hipStream_t stream, hipError_t *hip_err);
int launch_attention_decode(const void *q, const void *K, const void *Vt, const void *mask, void *out, int heads, int t, int hd, unsigned short alpha_bits,
                            hipStream_t stream, hipError_t *hip_err);
// comm.hip
struct Comm;
int comm_create(int rank, int world, int max_vector_elems, int slots, Comm **out, hipError_t *he);
int comm_export(Comm *c, void *handle64, hipError_t *he);
int comm_connect_ipc(Comm *c, const void *handles, hipError_t *he);
int comm_connect_local(Comm *c, Comm *const *all, hipError_t *he);
int comm_set_timeout_ms(Comm *c, int ms);
int comm_reset(Comm *c, hipError_t *he);
int comm_device(const Comm *c);
int comm_status(Comm *c, hipError_t *he);
void comm_destroy(Comm *c);
int launch_allgather_f16(Comm *c, int slot, const void *src_slice, void *dst_full, int n_total, hipStream_t stream, hipError_t *he);
int comm_world_of(const Comm *c);
// the peer-write exchange of ONE linear's output slice inside the launch that computes it (w4a16_gemv_i8.hip, the mixed launch; round 6): what that kernel's epilogue
// needs of the communicator -- same window, buffers, flags and epochs as allgather_peer_kernel, so the two forms can alternate on one slot
constexpr int kCommMaxRanks = 8, kCommFlagStride = 16;  // the flag table of a window: [slots][2 parities][kCommMaxRanks] words, kCommFlagStride words apart (comm.hip)
struct PeerGatherEpi {
    unsigned char *peer[8];  // every rank's window as mapped here
    unsigned *epochs;        // [slots] exchanges completed per slot, the status word at [slots], the arrival counters at [slots + 1 + slot]
    void *dst;               // the complete vector on this rank
    int rank, world, slot, slots;
    unsigned slice_elems;    // halves per rank
    size_t vec_bytes, flags_off;
    unsigned long long timeout_ticks;
};
int comm_peer_gather_epi(Comm *c, int slot, void *dst_full, int n_total, PeerGatherEpi *out);  // the checks of launch_allgather_f16, no launch
int comm_rccl_unique_id(void *id128);
int comm_rccl_init(Comm *c, const void *id128);
bool comm_has_rccl(const Comm *c);
const char *comm_rccl_last_error();  // text of this thread's last failed RCCL call
bool comm_peer_regime(const Comm *c, int n_total);
int launch_allgather_rccl(Comm *c, const void *src_slice, void *dst_full, size_t n_per_rank, hipStream_t stream);
size_t allgather_rows_workspace_bytes(int M, int n_total);
int launch_allgather_rows_f16(Comm *c, int slot, const void *src, void *dst, int M, int n_total, int ldd, void *workspace, hipStream_t stream, hipError_t *he);
// attention_fast.hip
void set_attention_fast_target(int wgs);
void set_attention_fast_waves(int nw);
void set_attention_fast_fuse(int r);
void set_attention_fast_probe_no_combine(int on);  // timing experiment: partial states stored, no combine, `out` not written
void describe_attention_decode_fast(int heads, int keys, int *chunk, int *chunks, int *waves, int kv_heads = 0);
size_t attention_decode_workspace_bytes(int heads, int max_keys, int hd);
// attention_prefill.hip: m > 1 new rows (rotation + append + causal / masked attention over pos + m keys), two launches
// opt_attention.hip: KV append + qk + mask / softmax / int8 + pv of the SmoothQuant OPT attention for m <= 8 new rows, one launch
int launch_opt_attention_decode(const void *q, const void *kn, const void *vn, void *kc, void *vtc, const float *mask, void *out, int heads, int hd, int m, int pos,
                                int max_keys, int ld, float a_qk, float a_pv, hipStream_t stream, hipError_t *hip_err);
size_t attention_prefill_workspace_bytes(int heads, int m, int hd);
void set_attention_prefill_waves(int w);  // 0 automatic, 4 / 8 forced
int launch_attention_prefill(const void *qkv, int ld_qkv, void *kc, void *vc, const void *cosv, const void *sinv, const void *mask, int ld_mask, int causal,
                             void *out, int ld_out, void *workspace, int heads, int kv_heads, int max_keys, int pos, int m, float alpha, hipStream_t stream,
                             hipError_t *hip_err);
// what a deferred attention step leaves for its consumer (tce_attention_deferred of the C ABI, field for field)
struct AttnDeferred {
    int slots;   // chunk slots of the launch's grid: 1 = nothing deferred (`out` is final); 2 .. kAttnDeferMaxSlots = partial states per (query head, slot)
    int chunk;   // keys per chunk: slot i of a head is live when i * chunk < position + 1
    int heads;   // query heads
    int stride;  // floats per partial state: M, L, -, -, O[128]
    const float *part;
};
constexpr int kAttnDeferMaxSlots = 8;
int launch_attention_decode_fast(const void *qkv, void *kc, void *vc, const void *cosv, const void *sinv, const void *mask, void *out, void *workspace,
                                 int heads, int kv_heads, int hd, int max_keys, int pos, unsigned short alpha_bits, hipStream_t stream, hipError_t *hip_err,
                                 const int *pos_dev = nullptr, AttnDeferred *deferred = nullptr);
int launch_rope_half(void *q, void *k, const void *cosv, const void *sinv, int heads, int len, int hd, int start_idx, hipStream_t stream, hipError_t *hip_err);
int launch_softmax_half(const void *x, void *out, long long rows, int n, hipStream_t stream, hipError_t *hip_err);
int launch_prefetch(const void *ptr, long long bytes, int workgroups, hipStream_t stream, hipError_t *hip_err);
int launch_add_half(const void *a, const void *b, void *c, long long n, hipStream_t stream, hipError_t *hip_err);
int launch_silu_mul_half(void *a, const void *b, long long n, hipStream_t stream, hipError_t *hip_err);

}  // namespace tce
