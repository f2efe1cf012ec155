/* tce_tuning.h -- tuning and diagnostics entry points of libtce_hip.so (round 6: split from tce_matmul.h).
 *
 * NOT part of the operator boundary: a host that drops this library in behind the reference's MatmulOperator never calls anything declared here.
 * These calls select among kernels the dispatcher can reach anyway (forced forms: the parity tests hold every form to the oracle / to each other bit for bit; the
 * sweeps under scripts/ time them) and point diagnostics at a buffer.  Every setting acts on the CALLING THREAD's launches only and computes the same results as the
 * default -- except the settings documented as diagnostic instantiations (outputs meaningless), which exist in the lab build only:
 * `python -m tinychatengine_amd.build --lab` -> libtce_hip_lab.so (same sources, -DTCE_LAB); the product library refuses them with TCE_ERR_BAD_ARG. */
#ifndef TCE_TUNING_H
#define TCE_TUNING_H
#include "tce_matmul.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Force a GEMV kernel + launch geometry for every subsequent call from this process (all 0 = automatic).
 *   waves_k >= 1: the workgroup-per-row-block kernel: rows_per_wave in {1,2,4}, waves_n x waves_k waves per workgroup
 *                 (waves_k of them split K), depth = weight steps kept in flight per wave;
 *   waves_k == 0: the persistent stream kernel: rows_per_wave in {1,2} rows per unit, waves_n (1..16) waves per
 *                 workgroup (one workgroup per CU), depth in {0 = auto, 2, 3} units in flight.
 * TCE_ERR_BAD_ARG if that variant was not compiled. */
TCE_API int tce_w4a16_set_gemv_config(int rows_per_wave, int waves_n, int waves_k, int depth);
/* The decode kernel on pre-packed copies (csrc/w4a16_gemv_i8.hip: M <= 4 rows as an exact int8 contraction on the matrix pipe; taken by tce_w4a16_forward /
 * _forward_group / plans whenever every descriptor of the launch carries `prepacked`, K % 128 == 0 and no fused RMSNorm prologue is asked for; group sizes 64 / 32:
 * M <= 2 / M = 1).  mode 0 = that rule, 1 = off (the fp16 GEMV kernels on the q4_6 arrays take those launches: A/B runs); rows = 16-row tiles per wave for the
 * M = 1, K <= 8192 launches: 0 = the rule (one; two with the RMSNorm prologue where one leaves a short second generation of workgroups), 1, 2.  A row's arithmetic depends on K, the group size and the rows per pass only -- never on N or on
 * `rows`: column shards and grouped launches are bit-identical to the plain launch.  Per host thread (0.1.10: every tuning setter of this header acts on the calling thread's launches only -- two host threads driving
 * two devices or streams may force different kernels without a lock); results do not depend on it beyond the kernel family. */
TCE_API int tce_w4a16_set_gemv_i8(int mode, int tiles_per_wave);
/* Tuning / diagnostics switch for the sweeps under scripts/ (per host thread since 0.1.10, never needed by a host):
 *   0..4     GEMV kernels, M = 1: 0 normal; 1 stream the weights only (no unpack, no dot products: the memory-side ceiling
 *            of the access pattern, outputs meaningless); 2 normal math plus per-wave timestamps into the debug buffer;
 *            3 / 4 further timing variants of the persistent kernel (w4a16_gemv_stream.hip)
 *   10..12   row-block GEMV, M = 1, issue order: 10 the rule (activations first when the grid is one generation of workgroups, weights
 *            first otherwise), 11 activations first always, 12 weights first always (scripts/gemv_order_ab.py)
 *   20..30   small-batch kernel: 20 automatic, 21 / 22 / 24 / 28 waves per tile, 30 shared-activation form, 29 off
 *   40..48   GEMM XCD grid rows: 40 automatic, 41 / 42 / 44 / 48 forced
 *   50..52   LDS-DMA GEMM wave quartets per tile: 50 automatic, 51 one, 52 two
 *   60..69   pre-packed 128-row GEMM: 60 automatic, 61 / 62 / 63 / 64 forced form (128x128 tile with one quartet / two quartets splitting K / 128x256 tile with two quartets side by side / 128x128 with the k range cut across workgroups when a scratch area is given; taken for every M), 69 off
 *   600+a    pre-packed GEMM, one quartet, with parts of its loop switched off (a: 1 rescale, 2 unpack, 4 fragment reads, 8 MFMAs,
 *            16 activation DMAs, 32 barriers; only the combinations scripts/gemm_pk_ablation.py uses are compiled); outputs meaningless
 *   70..74   W8A8 wave quartets per tile: 70 automatic, 71 / 72 / 74 forced
 *   1000+m   largest M the small-batch kernel takes (default 1128 = 128; 1016 restricts it to M <= 16)
 *   640..644 pre-packed GEMM with the k range cut across workgroups: runs per cut tile forced (640: the cost model's choice)
 *   2900+w   fast attention step: waves per workgroup, w in {4, 8, 16} (2900: the default, 4)
 *   2920+r   fast attention step, grouped queries: query heads per workgroup, r in {1, 2, 4} (2920: the rule, 1)
 *   3000+g   fast attention step: workgroups the key range is cut for (3000: the fitted per-context rule, the default)
 * Round 5 (the full list is the dispatch in csrc/tce_capi.hip, one commented `if` per range):
 *   66..69, 672..674, 2669, 690 / 691   pre-packed GEMM, 256-row wave tiles: whole tiles / k range cut / the tile shared by two quartets / 256 x 256 tiles; the dispatcher may pick them (691) or not (690)
 *   51000..51016                        the mixed decode launch (tce_w4a16_forward_independent): waves per workgroup forced (51000: the rule -- the width that launches the fewest waves)
 *   2676, 6916 / 6917                   pre-packed GEMM, 128 x 128 tiles with two quartets AND the k range handed off between two workgroups (round 6, form 16): forced; offered or not
 *   2670..2675, 2682..2684, 692 / 693   pre-packed GEMM, the wide forms (128 rows x 64 / 48 columns per wave) on 128 x 256 / 192 / 512 tiles, their k range cut in 2 / 3 / 4; offered (693) or not (692)
 *   694 / 695, 6950+d                   a k range cut in two runs: both meet at the counter (694) / run 0 hands its tile to run 1 (695, the default); run 0 shorter by d k-blocks (default 2)
 *   696 / 697 / 698, 6972..6974         the two waves of a SIMD at different priorities: off / on / the launcher's rule (default); level 3 / 1 / chosen by slot parity
 *   2600+a, 26000+a                     the 256-row / wide form with parts of the loop switched off (as 600+a; 128 / 256: where the refill is issued); outputs meaningless
 *   7700 / 7701, 7702 / 7703, 7710+u    TCE_PLAN_TAGGED on packed copies: the int8 token kernel where the list allows (7700, default) / never (7701); plans built from now on record per-stage
 *                                       wall-clock stamps in the debug buffer (7702) / stop (7703); a stage with more than u units per workgroup ends the prefix the kernel takes (7710: 512)
 *   (6262 / 6263 of round 5 are gone: the two-quartet forms are offered for every group size again -- isa_lint.py RULE 1, profiles/r6/pk_lost_lanes_rule.md)
 *   170..179, 180..188                  W8A8: the 64 x 64 tile with 8 k-steps in flight (quartets forced / off); a tile's k-steps cut across workgroups (180 the rule, 181 off, 182.. runs)
 *   190..192, 19000 / 19001 / 19304..   W8A8: 32 x 64 tiles (190 the rule, 191 forced, 192 off); the whole tile in every wave, the k-steps dealt to the waves (round 6: 19000 the rule,
 *                                       19001 off, 19304 / 19404 / 19904 the 32 x 48 / 32 x 64 / 64 x 64 tile forced wherever the 64 x 64 kernel would run)
 *   2700..2899, 2950..2968, 2930 / 2931 prefill attention: block pairing, waves x row tiles; fast attention step without its combine (2931: timing only, the output is NOT written)
 *   15000 / 15001, 50000+..             plans: graph replay / eager issue of a stream-ordered plan; overlapped plans' branches and ring slots
 * Every setting computes correct results except GEMV modes 1, 3, 4, the "switched off" ablations (600+a, 2600+a, 26000+a), 83 and 2931. */
TCE_API int tce_w4a16_set_debug_mode(int mode);
/* The same knobs per kernel family, by name (round 4: one numbered mode space for every family had already produced an A/B that compared a setting with itself, and
 * a mode of one family landing in another's range).  Per host thread (0.1.10), for tuning sweeps and tests; 0 everywhere = the fitted rules.  Every setting computes the same results.
 *   tce_attention_set_tuning: the fast decode attention step -- waves per workgroup (0 | 4 | 8 | 16), workgroups the key range is cut for (0 | 32..8192),
 *                             query heads per workgroup for grouped queries (0 | 1 | 2 | 4)
 *   tce_w8a8_set_tuning:      wave quartets per 64 x 64 tile (0 | 1 | 2 | 4); the 128-row tiles (0 the rule | 1 / 2: forced with 128 / 64 columns | 3 / 4: the same with two
 *                             quartets | 9 off); the 64 x 64 tile with eight k-steps in flight (0 the rule | 1 / 2 / 4: forced with that many quartets | 9 off) */
TCE_API int tce_attention_set_tuning(int waves_per_workgroup, int workgroups, int heads_per_workgroup);
TCE_API int tce_w8a8_set_tuning(int quartets_per_tile, int big_tiles, int deep_pipeline);
/* mode 2: every wave writes {start, x staged, math done, end} (100 MHz wall clock, 4 x u64 per wave) to this device buffer */
TCE_API int tce_w4a16_set_debug_buffer(void *device_buffer);
/* Force an MFMA GEMM tile (m_tiles x n_tiles of 16x16 per wave); 0,0 = automatic. */
TCE_API int tce_w4a16_set_gemm_config(int m_tiles, int n_tiles);
/* Enumerate the compiled kernel variants (for tuning sweeps / tests): returns 0 and fills the outputs, or
 * TCE_ERR_BAD_ARG when idx is past the end. */
TCE_API int tce_w4a16_gemv_variant(int idx, int *rows_per_wave, int *waves_n, int *waves_k, int *depth);
TCE_API int tce_w4a16_gemm_variant(int idx, int *m_tiles, int *n_tiles);

#ifdef __cplusplus
}
#endif
#endif /* TCE_TUNING_H */
