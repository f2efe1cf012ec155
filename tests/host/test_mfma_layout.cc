// Host check of tinychatengine_amd/csrc/w4a16_mfma_layout.hpp (compiled and run by tests/test_layout_host.py with g++):
// the packed weight format is a bijection, the kernel's mask unpack returns natural k order, the LDS image written by the DMA
// mapping is what the fragment reads fetch, and every ds_read_b128 lane group is bank-conflict free.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "w4a16_mfma_layout.hpp"

using namespace tce::pk;

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);   \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main() {
    // ---- 1. packed words: bijection + unpack order ----
    for (int N : {16, 40, 4096 / 16}) {
        const int K = 384;
        const int NP = nt16(N) * 16;
        std::vector<uint32_t> words(words_bytes(N, K) / 4, 0);
        std::vector<int> hits(words.size() * 8, 0);
        auto code = [&](int n, int k) { return (n * 7 + k * 3 + (k >> 5)) & 15; };
        for (int n = 0; n < NP; ++n)
            for (int k = 0; k < K; ++k) {
                const size_t wi = word_index(n, k, K);
                CHECK(wi < words.size());
                const int nb = nibble_index(k);
                hits[wi * 8 + nb]++;
                words[wi] |= (uint32_t)code(n, k) << (4 * nb);
            }
        for (int h : hits) CHECK(h == 1);
        // a lane's word s of (tile, k-block) must unpack to k = kb*128 + 32 s + 8 q + e, e ascending
        for (int jt = 0; jt < nt16(N); ++jt)
            for (int kb = 0; kb < K / 128; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int q = lane >> 4, n16 = lane & 15;
                        const uint32_t w = words[(((size_t)jt * (K / 128) + kb) * 64 + lane) * 4 + s];
                        int c[8];
                        unpack_word_codes(w, c);
                        for (int e = 0; e < 8; ++e) CHECK(c[e] == code(jt * 16 + n16, kb * 128 + 32 * s + 8 * q + e));
                    }
    }
    // zero-point constants: halves -(1024 + z) and -(64 + z)
    for (unsigned z = 0; z < 16; ++z) {
        const uint32_t w = zc_word(z);
        CHECK((w & 0xFFFF) == (0xE400u | z));          // sign | exp of 1024 | z            -> -(1024 + z)
        CHECK((w >> 16) == (0xD400u | (z << 4)));      // sign | exp of 64   | z * 2^4 ulps -> -(64 + z)
    }
    // ---- 2. LDS image: DMA writes vs fragment reads ----
    struct Piece { int row, p; };
    std::vector<Piece> lds;
    for (int nw : {8, 4}) {  // two quartets sharing one ring, one quartet per ring (checked last: `lds` is reused below)
        lds.assign(kHalfBytes / 16, Piece{-1, -1});
        for (int w = 0; w < nw; ++w)
            for (int ii = 0; ii < 16 / nw; ++ii)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = dma_row(w, ii, lane, nw);
                    const int off = dma_lds_base(w, ii, nw) + lane * 16;
                    CHECK(off % 16 == 0 && off < kHalfBytes);
                    CHECK(lds[off / 16].row == -1);
                    lds[off / 16] = Piece{row, dma_src_piece(row, lane)};
                }
        for (auto &pc : lds) CHECK(pc.row >= 0);
    }
    // every DMA instruction reads whole 128-byte rows (8 lanes = the 8 pieces of one row, permuted)
    for (int w = 0; w < 4; ++w)
        for (int ii = 0; ii < 4; ++ii)
            for (int r8 = 0; r8 < 8; ++r8) {
                std::set<int> ps;
                for (int l = 0; l < 8; ++l) ps.insert(dma_src_piece(dma_row(w, ii, r8 * 8 + l), r8 * 8 + l));
                CHECK(ps.size() == 8);
            }
    for (int i = 0; i < 8; ++i)
        for (int n16 = 0; n16 < 16; ++n16)
            for (int q = 0; q < 4; ++q)
                for (int sl = 0; sl < 2; ++sl) {
                    const int off = frag_offset(i, n16, q, sl);
                    CHECK(off % 16 == 0 && off < kHalfBytes);
                    CHECK(lds[off / 16].row == i * 16 + n16);
                    CHECK(lds[off / 16].p == sl * 4 + q);
                }
    // ---- 3. bank conflicts of the fragment reads: the four ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS table) ----
    const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                               {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                               {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                               {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int i = 0; i < 8; ++i)
        for (int sl = 0; sl < 2; ++sl)
            for (auto &grp : groups) {
                std::set<int> slots;
                for (int lane : grp) slots.insert((frag_offset(i, lane & 15, lane >> 4, sl) / 16) % 16);
                CHECK(slots.size() == 16);
            }
    // sizes
    CHECK(words_bytes(4096, 4096) == (size_t)4096 * 4096 / 2);
    CHECK(total_bytes(4096, 4096, 128) % 256 == 0);
    std::printf("mfma layout ok\n");
    return 0;
}
