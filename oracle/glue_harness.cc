// glue_harness.cc -- TEST INFRASTRUCTURE (oracle/; built by `make -C oracle glue` into oracle/_ref/glue_harness, only where
// /root/reference exists).  Runs the REFERENCE's own CUDA kernel sources for the ops either side of the W4A16 linears on the CPU --
// llm/src/ops/cuda/softmax.cu (softmax_cuda), BMM_F16T.cu (BMM_F16T::forward -> mat_mul_transposed_cuda), RotaryPosEmb.cu
// (RotaryPosEmb_cuda_forward) and the add_half / SiLuMul_half kernels of llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu -- through the
// host emulation in oracle/cuda_emul/ (one thread after the other; binary16 intrinsics as single correctly rounded operations), with
// the launch geometry of their call sites, so that tests/test_oracle_glue.py can pin oracle/tce_oracle.c's restatements
// (orc_softmax_half, orc_bmm_f16t, orc_rope_half, orc_add_half, orc_silu_mul_half) against them bit for bit.
//   glue_harness add   n a.bin b.bin out.bin
//   glue_harness silu  n a.bin b.bin out.bin                      (out = the kernel's in-place result in a)
//   glue_harness softmax x y z in.bin out.bin
//   glue_harness bmm   batch m n k alpha_bits a.bin w.bin out.bin
//   glue_harness rope  heads len hd start positions q.bin k.bin cos.bin sin.bin q_out.bin k_out.bin
//   glue_harness gemv  m n k x.bin qweight.bin scales.bin zeros.bin out.bin   (matmul::MatmulOperator::gemv_forward_cuda -> gemv_kernel_g128,
//                                                                 kernels/cuda/gemv_cuda.cu:140-260 -- THE hot-path kernel of the reference, from its
//                                                                 own source, block (32, 4) as concurrent threads, __shfl_down_sync warp reduction)
//   glue_harness lnq   m n x_f32.bin w_f32.bin b_f32.bin out_i8.bin   (LayerNormQ::forward, llm/src/ops/LayerNormQ.cc:12-52 -- host code in the
//                                                                 reference, compiled as it is: pins orc_layernorm_q)
//   glue_harness optsm heads sq tgz scores_f32.bin mask_f32.bin out_i8.bin   (batch_Add -> softmax -> the int8 conversion of Int8OPTAttention.cc:254-268:
//                                                                 llm/src/ops/batch_add.cc and softmax.cc compiled as they are: pins orc_opt_softmax_q)
//   glue_harness rmsnorm m n eps x.bin gamma_f32.bin out.bin     (LlamaRMSNorm_cuda::forward -> generalT5LayerNorm: warp shuffles and
//                                                                 __syncthreads, so its block runs as concurrent OS threads)
// Nothing of this is part of the product.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "operators.h"

thread_local tce_emul_idx blockIdx, threadIdx;
thread_local dim3 blockDim, gridDim;
tce_emul::block_state *tce_emul::g_block = nullptr;

// the reference's kernels (defined in the sources named above; no header declares them)
__global__ void softmax_cuda(Matrix3D<half> input, Matrix3D<half> output);
__global__ void RotaryPosEmb_cuda_forward(Matrix3D<half> query, Matrix3D<half> key, Matrix3D<half> cos, Matrix3D<half> sin, int start_idx, int len);
__global__ void add_half(Matrix3D<float16_t> a, Matrix3D<float16_t> b, Matrix3D<float16_t> c);
__global__ void SiLuMul_half(Matrix3D<float16_t> a, Matrix3D<float16_t> b);

static std::vector<half> rd(const char *path, size_t n) {
    std::vector<half> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), 2, n, f) != n) {
        fprintf(stderr, "glue_harness: cannot read %zu halves from %s\n", n, path);
        exit(2);
    }
    fclose(f);
    return v;
}
static void wr(const char *path, const std::vector<half> &v) {
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(v.data(), 2, v.size(), f) != v.size()) {
        fprintf(stderr, "glue_harness: cannot write %s\n", path);
        exit(2);
    }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 2) return 64;
    const std::string op = argv[1];
    auto I = [&](int i) { return atoi(argv[i]); };
    if (op == "add" && argc == 6) {  // Int4llamaDecoderLayer.cu:86-88: 1024 threads per block, ceil(n / 1024) blocks
        const int n = I(2);
        auto a = rd(argv[3], n), b = rd(argv[4], n);
        std::vector<half> c(n);
        Matrix3D<float16_t> A(a.data(), 1, 1, n), B(b.data(), 1, 1, n), Cm(c.data(), 1, 1, n);
        tce_emul::launch(tce_emul::cfg((n + 1023) / 1024, 1024), [&] { add_half(A, B, Cm); });
        wr(argv[5], c);
        return 0;
    }
    if (op == "silu" && argc == 6) {  // Int4llamaDecoderLayer.cu:100-102
        const int n = I(2);
        auto a = rd(argv[3], n), b = rd(argv[4], n);
        Matrix3D<float16_t> A(a.data(), 1, 1, n), B(b.data(), 1, 1, n);
        tce_emul::launch(tce_emul::cfg((n + 1023) / 1024, 1024), [&] { SiLuMul_half(A, B); });
        wr(argv[5], a);
        return 0;
    }
    if (op == "softmax" && argc == 7) {  // Int4llamaAttention.cu:196-199: block (64, 16), grid over (heads, rows)
        const int x = I(2), y = I(3), z = I(4);
        auto in = rd(argv[5], (size_t)x * y * z);
        std::vector<half> out((size_t)x * y * z);
        Matrix3D<half> Im(in.data(), x, y, z), Om(out.data(), x, y, z);
        dim3 block(64, 16), grid((x + 63) / 64, (y + 15) / 16);
        tce_emul::launch(tce_emul::cfg(grid, block), [&] { softmax_cuda(Im, Om); });
        wr(argv[6], out);
        return 0;
    }
    if (op == "bmm" && argc == 10) {  // BMM_F16T::forward (BMM_F16T.cu:50-78), its own grid / block
        const int bsz = I(2), m = I(3), n = I(4), k = I(5);
        half alpha;
        alpha.x = (uint16_t)I(6);
        auto a = rd(argv[7], (size_t)bsz * m * k), w = rd(argv[8], (size_t)bsz * n * k);
        std::vector<half> c((size_t)bsz * m * n);
        Matrix3D<half> Am(a.data(), bsz, m, k), Wm(w.data(), bsz, n, k), Cm(c.data(), bsz, m, n);
        BMM_F16T bmm(alpha);
        bmm.forward(Am, Wm, Cm);
        wr(argv[9], c);
        return 0;
    }
    if (op == "rope" && argc == 13) {  // Int4llamaAttention.cu:157-159: grid (heads), block (len)
        const int heads = I(2), len = I(3), hd = I(4), start = I(5), positions = I(6);
        auto q = rd(argv[7], (size_t)heads * len * hd), k = rd(argv[8], (size_t)heads * len * hd);
        auto cs = rd(argv[9], (size_t)positions * hd), sn = rd(argv[10], (size_t)positions * hd);
        Matrix3D<half> Q(q.data(), heads, len, hd), K(k.data(), heads, len, hd), Cs(cs.data(), 1, positions, hd), Sn(sn.data(), 1, positions, hd);
        tce_emul::launch(tce_emul::cfg(dim3(heads, 1, 1), dim3(len, 1, 1)), [&] { RotaryPosEmb_cuda_forward(Q, K, Cs, Sn, start, len); });
        wr(argv[11], q);
        wr(argv[12], k);
        return 0;
    }
    if (op == "gemv" && argc == 10) {
        const int m = I(2), n = I(3), k = I(4);
        const int zw = (k / 128 + 7) / 8;  // calculate_zeros_width (llm/src/nn_modules/cuda/utils.cu:162-178)
        auto x = rd(argv[5], (size_t)m * k);
        std::vector<uint32_t> qw((size_t)n * (k / 8)), zp((size_t)n * zw);
        auto rdw = [&](const char *path, void *dst, size_t bytes) {
            FILE *f = fopen(path, "rb");
            if (!f || fread(dst, 1, bytes, f) != bytes) {
                fprintf(stderr, "glue_harness: cannot read %zu bytes from %s\n", bytes, path);
                exit(2);
            }
            fclose(f);
        };
        rdw(argv[6], qw.data(), qw.size() * 4);
        auto sc = rd(argv[7], (size_t)n * zw * 8);
        rdw(argv[8], zp.data(), zp.size() * 4);
        std::vector<half> out((size_t)m * n);
        struct matmul_params p;
        memset(&p, 0, sizeof(p));
        p.A.row = m; p.A.column = k; p.A.half_data_ptr = x.data();
        p.B.row = k; p.B.column = n; p.B.int32_data_ptr = reinterpret_cast<int32_t *>(qw.data());
        p.C.row = m; p.C.column = n; p.C.half_data_ptr = out.data();
        p.half_scales = sc.data();
        p.int32_zero_point = reinterpret_cast<int *>(zp.data());
        p.block_size = 128;
        matmul::MatmulOperator opr;
        opr.gemv_forward_cuda(&p);
        wr(argv[9], out);
        return 0;
    }
    if (op == "lnq" && argc == 8) {
        const int m = I(2), n = I(3);
        auto rdf = [&](const char *path, size_t cnt) {
            std::vector<float> v(cnt);
            FILE *f = fopen(path, "rb");
            if (!f || fread(v.data(), 4, cnt, f) != cnt) exit(2);
            fclose(f);
            return v;
        };
        auto x = rdf(argv[4], (size_t)m * n), w = rdf(argv[5], n), b = rdf(argv[6], n);
        std::vector<int8_t> out((size_t)m * n);
        LayerNormQ_params params;
        params.weight = Matrix3D<float>(w.data(), 1, 1, n);
        params.bias = Matrix3D<float>(b.data(), 1, 1, n);
        LayerNormQ ln(params);
        Matrix3D<float> X(x.data(), 1, m, n);
        Matrix3D<int8_t> O(out.data(), 1, m, n);
        ln.forward(X, O);
        FILE *f = fopen(argv[7], "wb");
        if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) return 2;
        fclose(f);
        return 0;
    }
    if (op == "optsm" && argc == 8) {  // batch_Add (batch_add.cc) -> softmax (softmax.cc), the reference's own sources, then Int8OPTAttention.cc:264-267's loop
        const int heads = I(2), sq = I(3), tgz = I(4);
        auto rdf = [&](const char *path, size_t cnt) {
            std::vector<float> v(cnt);
            FILE *f = fopen(path, "rb");
            if (!f || fread(v.data(), 4, cnt, f) != cnt) exit(2);
            fclose(f);
            return v;
        };
        auto sc = rdf(argv[5], (size_t)heads * sq * tgz), mk = rdf(argv[6], (size_t)sq * tgz);
        Matrix3D<float> attn_weights(sc.data(), heads, sq, tgz), mask(mk.data(), 1, sq, tgz);
        batch_Add(attn_weights, mask, attn_weights);
        Matrix3D<float> attn_probs(sc.data(), heads, sq, tgz);
        softmax(attn_weights, attn_probs, 2);
        std::vector<int8_t> out(sc.size());
        const int len = attn_probs.length();
        for (int i = 0; i < len; i++) out[i] = static_cast<int8_t>(std::round(attn_probs.m_data[i] * 127));  // Int8OPTAttention.cc:266
        FILE *f = fopen(argv[7], "wb");
        if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) return 2;
        fclose(f);
        return 0;
    }
    if (op == "rmsnorm" && argc == 8) {  // LlamaRMSNorm_cuda::forward (LlamaRMSNorm.cu:96-115), its own grid / block
        const int m = I(2), n = I(3);
        const float eps = (float)atof(argv[4]);
        auto x = rd(argv[5], (size_t)m * n);
        std::vector<float> gamma(n);
        FILE *f = fopen(argv[6], "rb");
        if (!f || fread(gamma.data(), 4, n, f) != (size_t)n) return 2;
        fclose(f);
        std::vector<half> out((size_t)m * n);
        Matrix3D<half> X(x.data(), 1, m, n), O(out.data(), 1, m, n);
        LlamaRMSNorm_cuda norm(Matrix3D<float>(gamma.data(), 1, 1, n));
        norm.forward(X, O, eps);
        wr(argv[7], out);
        return 0;
    }
    fprintf(stderr, "glue_harness: bad arguments\n");
    return 64;
}
