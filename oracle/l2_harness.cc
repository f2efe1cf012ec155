// l2_harness.cc -- TEST INFRASTRUCTURE (oracle/): the link-and-run proof of the drop-in.
//
// oracle/Makefile (target `l2link`) compiles the reference's OWN, UNMODIFIED level-2 callers
//     llm/src/ops/cuda/linear.cu            Linear_half_int4::forward, Linear_FP16_int4_ref::forward_ref   (:5-40, :43-77)
//     llm/src/ops/W8A8B8O8Linear.cc         W8A8B8O8Linear::forward, load_W8A8B8O8Linear_params            (:5-78)
//     llm/src/ops/W8A8B8O8LinearReLU.cc     W8A8B8O8LinearReLU::forward                                    (:7-78)
//     llm/src/ops/W8A8BFP32OFP32Linear.cc   W8A8BFP32OFP32Linear::forward                                  (:6-73)
//     llm/src/ops/BMM_S8T_S8N_F32T.cc       BMM_S8T_S8N_F32T::forward                                      (:12-63)
//     llm/src/ops/BMM_S8T_S8N_S8T.cc        BMM_S8T_S8N_S8T::forward                                       (:12-62)
//     llm/src/utils.cc                      read_to_array<T>                                               (:15-30)
// from /root/reference where they lie (as host C++, -DQM_CUDA, the <cuda*.h> names served by oracle/cuda_shim/), compiles
// tinychatengine_amd/adapter/matmul_operator_hip.cc against the reference's own kernels/matmul.h
// (-DTCE_ADAPTER_USE_REFERENCE_HEADER) and links all of it with this file into oracle/_ref/l2_harness.  Every
// `op.<member>(&params)` those objects make therefore resolves to the HIP adapter -- no member is re-typed here.
//
// This file is the part of a QM_HIP build that INTEGRATION.md section 2.4 asks the maintainer to supply: the two allocator
// templates of llm/src/nn_modules/cuda/utils.cu:92-103 on top of tce_malloc / tce_free, calculate_zeros_width
// (utils.cu:157-178, host arithmetic restated), the NUM_THREAD global the application files define, and a main() that plays
// the role of llm/tests/cuda/test_ops.cu / non_cuda/test_ops.cc: build the op from files, forward, write the output.
// tests/test_l2_link.py writes the files (quantize.py / numpy) and compares the outputs with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "operators.h"
#include "utils.h"

#include "tce_matmul.h"

int NUM_THREAD = 8;  // defined by the reference's application files (e.g. llm/application/chat.cc); CPU backends only read it

// ---- llm/src/nn_modules/cuda/utils.cu:92-103, 157-178 for a HIP build ----
template <typename T>
void allocate_aligned_memory_gpu(T *&ptr, size_t size) {
    void *p = nullptr;
    if (tce_malloc(&p, size, /*managed=*/1) != TCE_OK) throw std::runtime_error(tce_last_error());
    ptr = static_cast<T *>(p);
}
template <typename T>
void free_aligned_memory_gpu(T *&ptr) {
    if (ptr) tce_free(ptr);
    ptr = nullptr;
}
template void allocate_aligned_memory_gpu(float16_t *&, size_t);
template void allocate_aligned_memory_gpu(naive_float16_t *&, size_t);
template void allocate_aligned_memory_gpu(int *&, size_t);
template void allocate_aligned_memory_gpu(int8_t *&, size_t);
template void allocate_aligned_memory_gpu(float *&, size_t);

int make_divisible_c(int c, int divisor) { return (c + divisor - 1) / divisor; }
int calculate_zeros_width(int in_features, int group_size, int pack_num) {
    int mult;
    if (group_size >= 128) mult = 1;
    else if (group_size == 64) mult = 2;
    else if (group_size == 32) mult = 4;
    else throw std::runtime_error("The group_size of calculate_zeros_width should be 128, 64 or 32.");
    return make_divisible_c(make_divisible_c(in_features / group_size, pack_num), mult) * mult;
}

namespace {

template <typename T>
T *dev_alloc(size_t n) {
    T *p = nullptr;
    allocate_aligned_memory_gpu(p, n * sizeof(T));
    return p;
}

template <typename T>
void read_file(const std::string &path, T *dst, size_t n) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.read(reinterpret_cast<char *>(dst), (std::streamsize)(n * sizeof(T)));
    if ((size_t)f.gcount() != n * sizeof(T)) throw std::runtime_error("short read: " + path);
}

template <typename T>
void write_file(const std::string &path, const T *src, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(src), (std::streamsize)(n * sizeof(T)));
    if (!f) throw std::runtime_error("cannot write " + path);
}

void sync() {
    if (tce_synchronize(nullptr) != TCE_OK) throw std::runtime_error(tce_last_error());
}

// Linear_half_int4 (llm/include/ops/linear.h:215-247): its own constructor allocates scale / zero_point through
// allocate_aligned_memory_gpu and loads the three q4_6 files; the weight buffer is the caller's, as in
// Int4llamaDecoderLayer.cu:41-66.
int run_w4a16(const std::string &dir, int M, int N, int K) {
    int *w = dev_alloc<int>((size_t)N * K / 8);
    Linear_half_int4 op(Matrix3D<int>(w, 1, N, K / 8), dir);
    float16_t *x = dev_alloc<float16_t>((size_t)M * K), *y = dev_alloc<float16_t>((size_t)M * N);
    read_file(dir + "/x.bin", x, (size_t)M * K);
    std::memset(y, 0xff, (size_t)M * N * sizeof(float16_t));  // NaN pattern: unwritten outputs show
    Matrix3D<float16_t> X(x, 1, M, K), Y(y, 1, M, N);
    op.forward(X, Y);
    op.forward(X, Y);  // second call: the adapter's per-tensor caches are warm
    sync();
    write_file(dir + "/out.bin", y, (size_t)M * N);
    return 0;
}

// kind 0 W8A8B8O8Linear, 1 W8A8B8O8LinearReLU, 2 W8A8BFP32OFP32Linear (llm/tests/non_cuda/test_ops.cc:177-345)
int run_w8a8(const std::string &dir, int kind, int B, int M, int N, int K) {
    int8_t *w = dev_alloc<int8_t>((size_t)N * K), *x = dev_alloc<int8_t>((size_t)B * M * K);
    read_file(dir + "/x.bin", x, (size_t)B * M * K);
    Matrix3D<int8_t> X(x, B, M, K), W(w, 1, N, K);
    if (kind == 2) {
        float *bias = dev_alloc<float>((size_t)N), *y = dev_alloc<float>((size_t)B * M * N);
        struct W8A8BFP32OFP32Linear_params p;
        p.weight = W;
        p.bias = Matrix3D<float>(bias, 1, 1, N);
        W8A8BFP32OFP32Linear op(p);
        load_W8A8BFP32OFP32Linear_params(op, dir);
        Matrix3D<float> Y(y, B, M, N);
        op.forward(X, Y);
        sync();
        write_file(dir + "/out.bin", y, (size_t)B * M * N);
        return 0;
    }
    int8_t *bias = dev_alloc<int8_t>((size_t)N), *y = dev_alloc<int8_t>((size_t)B * M * N);
    Matrix3D<int8_t> Y(y, B, M, N);
    if (kind == 0) {
        struct W8A8B8O8Linear_params p;
        p.weight = W;
        p.bias = Matrix3D<int8_t>(bias, 1, 1, N);
        W8A8B8O8Linear op(p);
        load_W8A8B8O8Linear_params(op, dir);
        op.forward(X, Y);
    } else {
        struct W8A8B8O8LinearReLU_params p;
        p.weight = W;
        p.bias_int8 = Matrix3D<int8_t>(bias, 1, 1, N);
        W8A8B8O8LinearReLU op(p);
        load_W8A8B8O8LinearReLU_params(op, dir);
        op.forward(X, Y);
    }
    sync();
    write_file(dir + "/out.bin", y, (size_t)B * M * N);
    return 0;
}

// BMM_S8T_S8N_F32T / BMM_S8T_S8N_S8T (test_ops.cc:380-473): x [b][m][k], weight [b][n][k]; m == 1 && b > 1 takes the *_batch members
int run_bmm(const std::string &dir, bool fp32_out, int B, int M, int N, int K) {
    int8_t *x = dev_alloc<int8_t>((size_t)B * M * K), *w = dev_alloc<int8_t>((size_t)B * N * K);
    read_file(dir + "/x.bin", x, (size_t)B * M * K);
    read_file(dir + "/weight.bin", w, (size_t)B * N * K);
    Matrix3D<int8_t> X(x, B, M, K), W(w, B, N, K);
    if (fp32_out) {
        float *y = dev_alloc<float>((size_t)B * M * N);
        struct BMM_S8T_S8N_F32T_params p;
        p.alpha = 0;
        BMM_S8T_S8N_F32T op(p);
        load_BMM_S8T_S8N_F32T(op, dir);
        Matrix3D<float> Y(y, B, M, N);
        op.forward(X, W, Y);
        sync();
        write_file(dir + "/out.bin", y, (size_t)B * M * N);
    } else {
        int8_t *y = dev_alloc<int8_t>((size_t)B * M * N);
        struct BMM_S8T_S8N_S8T_params p;
        p.alpha = 0;
        BMM_S8T_S8N_S8T op(p);
        load_BMM_S8T_S8N_S8T(op, dir);
        Matrix3D<int8_t> Y(y, B, M, N);
        op.forward(X, W, Y);
        sync();
        write_file(dir + "/out.bin", y, (size_t)B * M * N);
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    try {
        const std::string cmd = argc > 1 ? argv[1] : "";
        if (cmd == "symbols") {  // proves the dynamic link resolves without touching a device
            std::printf("l2_harness linked: sizeof(matmul_params)=%zu tce_version=%d\n", sizeof(struct matmul_params), tce_version());
            return 0;
        }
        auto I = [&](int i) { return std::atoi(argv[i]); };
        if (cmd == "w4a16" && argc == 6) return run_w4a16(argv[2], I(3), I(4), I(5));
        if ((cmd == "w8a8" || cmd == "w8a8relu" || cmd == "w8a8fp32") && argc == 7)
            return run_w8a8(argv[2], cmd == "w8a8" ? 0 : (cmd == "w8a8relu" ? 1 : 2), I(3), I(4), I(5), I(6));
        if ((cmd == "bmm_f32" || cmd == "bmm_s8") && argc == 7) return run_bmm(argv[2], cmd == "bmm_f32", I(3), I(4), I(5), I(6));
        std::fprintf(stderr, "usage: l2_harness symbols | w4a16 DIR M N K | w8a8|w8a8relu|w8a8fp32 DIR B M N K | bmm_f32|bmm_s8 DIR B M N K\n");
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "l2_harness: %s\n", e.what());
        return 1;
    } catch (const char *e) {  // common.h:115 throws a string literal on file errors
        std::fprintf(stderr, "l2_harness: %s\n", e);
        return 1;
    }
}
