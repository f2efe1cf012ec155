// w4a16_mfma_layout.hpp -- index arithmetic of the pre-packed ("q4_mfma") weight format and of the LDS activation image used by
// the 128-row MFMA GEMM (w4a16_gemm_pk.hip).  Plain functions, usable on the host (tests/host/test_mfma_layout.cc checks that
// what the prepack kernel writes, what the DMA writes and what the fragment reads fetch agree for every (n, k) and (row, k)).
//
// Source format (the reference's q4_6, llm/tools/quantize_methods.py:370-442): qweight u32 [N][K/8], nibble i of word j = code of
// k = 8j + i; scales fp16 [N][zw*8]; zeros u32 [N][zw].  It stays what the library accepts at its boundary; the packed copy is
// built once per weight tensor at load time (tce_w4a16_prepack; SURVEY section 8f rank 2 "offline pre-swizzle to an MFMA-friendly
// tile layout") and only the prefill GEMM reads it.
//
// q4_mfma, group size G in {128, 64, 32}, K % 128 == 0, NT16 = ceil(N / 16) row tiles, NKB = K / 128 k-blocks:
//   words   u32 [NT16][NKB][64 lanes][4]     lane = q * 16 + n16 (q = k-quarter of an MFMA step, n16 = row within the tile);
//                                            word s of that lane holds the 8 codes k = kb*128 + 32*s + 8*q + e (e = 0..7) of row
//                                            jt*16 + n16 -- exactly the B fragment v_mfma_f32_16x16x32_f16 wants from this lane in
//                                            step s, so a wave's load of one (tile, k-block) is one contiguous KiB;
//                                            nibble order inside a word: [e0 e2 e4 e6 e1 e3 e5 e7], so that the four mask
//                                            extractions  w & 0x000F000F, w & 0x00F000F0, (w>>8) & 0x000F000F, (w>>8) & 0x00F000F0
//                                            yield the halves (e0,e1), (e2,e3), (e4,e5), (e6,e7): natural k order, 9 VALU per word
//   consts  {f32 e, u32 zc} [NT16][K/G][16]  per (row, group): e = the EFFECTIVE scale of group g (the fp16 scale as fp32, or -- for a
//                                            group whose scale is 0 -- the previous group's effective scale (for leading zero-scale
//                                            groups: the row's first non-zero scale; 1.0 for an all-zero row);
//                                            such a group's codes are rewritten to its zero point, so it contributes exactly 0
//                                            either way); zc = packed halves (-(1024 + z), -(64 + z)): the exact zero-point removal
//                                            constants of the two nibble positions.
//                                            The kernel keeps its accumulator in units of the CURRENT group's scale: before a group's
//                                            MFMAs accumulate into it, acc <- acc * (e_prev / e_g) (one v_rcp-free fp32 divide per
//                                            group per lane, computed from two consecutive consts), and multiplies by e of the last
//                                            group once at the end:  sum_g s_g * blk_g  ==  e_last * (((blk_0 r_1 + blk_1) r_2 + ...)
//   last    f32 [NT16][16]                   e of the row's last group (written by the prepack for tools and tests; the kernel carries
//                                            the value it needs in a register: a k-split wave quartet ends on ITS last group)
//   dscales fp16 [NT16][K/G][16]             (round 4) the fp16 group scales as loaded, tile-major: the 16 rows of a tile side by side per group -- what the
//                                            decode kernel on this copy (w4a16_gemv_i8.hip) reads: 8 bytes per lane = the four rows of its MFMA output registers
//   dzeros  u32 [NT16][K/G][2]               (round 4) the 4-bit zero points of the same 16 rows: nibble n16 % 8 of word n16 / 8 (read only without TCE_W4_ZERO_POINT_IS_8)
//   rows past N (N % 16 != 0) are zero-filled tile rows: codes = 8, zc for z = 8, e = 1, last = 0, dscale = 0, dzero = 8.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TCE_HD __host__ __device__ inline
#else
#define TCE_HD inline
#endif

namespace tce {
namespace pk {

constexpr int kBM = 128;         // rows of the activation tile (= rows per wave)
constexpr int kHalfK = 64;       // k per half-stage
constexpr int kHalfBytes = kBM * kHalfK * 2;  // 16 KiB: one half-stage of activations

TCE_HD int nt16(int N) { return (N + 15) / 16; }
TCE_HD size_t words_bytes(int N, int K) { return (size_t)nt16(N) * (K / 128) * 64 * 4 * 4; }
TCE_HD size_t consts_bytes(int N, int K, int G) { return (size_t)nt16(N) * (K / G) * 16 * 8; }
TCE_HD size_t last_bytes(int N) { return (size_t)nt16(N) * 16 * 4; }
// one allocation: [words | consts | last | dscales | dzeros], each part 256-byte aligned
TCE_HD size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
TCE_HD size_t consts_offset(int N, int K) { return align256(words_bytes(N, K)); }
TCE_HD size_t last_offset(int N, int K, int G) { return consts_offset(N, K) + align256(consts_bytes(N, K, G)); }
TCE_HD size_t dscales_bytes(int N, int K, int G) { return (size_t)nt16(N) * (K / G) * 16 * 2; }
TCE_HD size_t dzeros_bytes(int N, int K, int G) { return (size_t)nt16(N) * (K / G) * 8; }
TCE_HD size_t dscales_offset(int N, int K, int G) { return last_offset(N, K, G) + align256(last_bytes(N)); }
TCE_HD size_t dzeros_offset(int N, int K, int G) { return dscales_offset(N, K, G) + align256(dscales_bytes(N, K, G)); }
TCE_HD size_t total_bytes(int N, int K, int G) { return dzeros_offset(N, K, G) + align256(dzeros_bytes(N, K, G)); }

// ---- packed words ----
// position of code (n, k): index of the u32 word in `words`, and the nibble (0..7) inside it
TCE_HD size_t word_index(int n, int k, int K) {
    const int jt = n >> 4, n16 = n & 15, kb = k >> 7, kk = k & 127;
    const int s = kk >> 5, q = (kk >> 3) & 3;
    return (((size_t)jt * (K >> 7) + kb) * 64 + (q * 16 + n16)) * 4 + s;
}
TCE_HD int nibble_index(int k) {
    const int e = k & 7;
    return (e >> 1) + 4 * (e & 1);  // e0 e2 e4 e6 -> nibbles 0..3, e1 e3 e5 e7 -> nibbles 4..7
}
// the unpack of the kernel, as integers: the 8 codes of a packed word in natural order e = 0..7
TCE_HD void unpack_word_codes(uint32_t w, int out[8]) {
    const uint32_t sh = w >> 8;
    const uint32_t d0 = w & 0x000F000Fu, d1 = w & 0x00F000F0u, d2 = sh & 0x000F000Fu, d3 = sh & 0x00F000F0u;
    out[0] = d0 & 0xFFFF;          out[1] = d0 >> 16;
    out[2] = (d1 & 0xFFFF) >> 4;   out[3] = d1 >> 20;
    out[4] = d2 & 0xFFFF;          out[5] = d2 >> 16;
    out[6] = (d3 & 0xFFFF) >> 4;   out[7] = d3 >> 20;
}
TCE_HD size_t const_index(int n, int g, int K, int G) { return ((size_t)(n >> 4) * (K / G) + g) * 16 + (n & 15); }
// zc word: low half -(1024 + z) = 0xE400 | z, high half -(64 + z) = 0xD400 | z << 4 (ulp of [64,128) is 1/16)
TCE_HD uint32_t zc_word(unsigned z) { return ((0xD400u | (z << 4)) << 16) | (0xE400u | z); }

// ---- LDS activation image: half-stages of 128 rows x 64 halves (128 bytes per row, 8 pieces of 16 bytes) ----
// piece p (k = 8p .. 8p+7 of the half-block) of row r sits at position p ^ swz(r): the 16 lanes that one ds_read_b128 serves
// together (MI355X_MICROARCH.md, LDS table) then touch 16 distinct 16-byte slots of the 256-byte bank row.
TCE_HD int swz(int row) { return (row >> 1) & 7; }
TCE_HD int lds_piece_offset(int row, int p) { return row * 128 + ((p ^ swz(row)) << 4); }
// DMA: one global_load_lds_dwordx4 of a wave writes 1 KiB lane-linear (wave-uniform base + lane * 16) = 8 rows.
// With `nw` waves feeding a ring (4, or 8 when two wave quartets share one activation tile), instruction `ii` (0 .. 16/nw - 1) of
// wave `w` (0 .. nw-1) fills rows (ii * nw + w) * 8 .. + 7; lane l writes row + (l >> 3), position l & 7, and therefore FETCHES
// the piece that belongs there.
TCE_HD int dma_row(int w, int ii, int lane, int nw = 4) { return (ii * nw + w) * 8 + (lane >> 3); }
TCE_HD int dma_src_piece(int row, int lane) { return (lane & 7) ^ swz(row); }
TCE_HD int dma_lds_base(int w, int ii, int nw = 4) { return (ii * nw + w) * 1024; }
// fragment read of lane (n16, q), m-tile i, local step sl (0 / 1) of a half-stage: 16 bytes = k 32*sl + 8*q .. + 7 of row 16 i + n16
TCE_HD int frag_offset(int i, int n16, int q, int sl) { return lds_piece_offset(i * 16 + n16, sl * 4 + q); }

}  // namespace pk
}  // namespace tce
