// adapter_bench.cc -- the DROP-IN path on the clock (round 5; VERDICT r4 "missing #3").
//
// Issues decode tokens exactly the way the reference's host does: one `matmul::MatmulOperator::gemv_forward_cuda(&params)` per linear -- the fused qkv
// projection, o_proj, gate_proj, up_proj, down_proj, lm_head -- with a `matmul_params` filled on the stack the way Linear_half_int4::forward fills it
// (llm/src/ops/cuda/linear.cu:5-40), on the NULL stream, eagerly, one launch after the other, with the glue between them in the reference's order
// (llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:73-115: input_layernorm -> attention block -> add_half -> post_attention_layernorm -> gate, up ->
// SiLuMul_half -> down -> add_half; Int4llamaAttention.cu:125,220: qkv_proj, o_proj; Int4llamaForCausalLM.cu:33: lm_head).  The glue is this library's
// C-ABI counterpart of each reference kernel (tce_rmsnorm_half, tce_add_half, tce_silu_mul_half) and ONE launch for the attention block between qkv_proj and
// o_proj (tce_attention_decode_step_gqa_f16: INTEGRATION.md 2.6).  11 launches per block + final norm + lm_head.
//
// Three legs, same launch list, same buffers:
//   adapter      through libtce_matmul_operator.so (the mangled MatmulOperator member: per-tensor look-up, packed copy, zero-point verdict)      <- the number asked for
//   capi_eager   tce_w4a16_forward called directly with descriptors that already carry the packed copy (what the adapter adds = adapter - capi_eager)
//   capi_graph   the same direct calls captured once into a hipGraph and replayed (what a host that adopts graphs gets; the launch structure is unchanged)
// Per leg: `warmup` tokens, then `tokens` tokens; wall-clock (issue + final synchronise), HIP events on the null stream around the same span, and the host time
// of the issue loop alone (-> host microseconds per call).  Prints one JSON object.
//
// Weights are synthetic (random codes, scales 0.003 * U(0.5, 1.5), zero point 8): every layer has its OWN buffers (3.9 GB for Llama-3-8B -> a token streams from
// HBM, not from the 256 MB cache), filled from one host image per shape.  Timed at a fixed context (pos = keys - 1), like bench.py's whole-token leg.
//
// This is a measurement tool, not part of the product: it links the HIP runtime directly for events and graph capture.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tce_matmul.h"
#include "tce_matmul_operator.h"

namespace {

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            std::printf("%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);  \
            std::exit(2);                                                                           \
        }                                                                                           \
    } while (0)
#define TCECHK(x)                                                                              \
    do {                                                                                       \
        int r_ = (x);                                                                          \
        if (r_ != TCE_OK) {                                                                    \
            std::printf("%s failed: %s (%d) (%s:%d)\n", #x, tce_last_error(), r_, __FILE__, __LINE__); \
            std::exit(2);                                                                      \
        }                                                                                      \
    } while (0)

struct Shape {
    const char *name;
    int hidden, heads, kv_heads, ffn, vocab, layers;
};
const Shape kShapes[] = {
    {"llama3-8b", 4096, 32, 8, 14336, 128256, 32},   // llm/include/model.h:83
    {"llama2-7b", 4096, 32, 32, 11008, 32000, 32},   // llm/include/model.h:71 (the shapes BASELINE.json spells)
    {"tiny", 256, 2, 2, 512, 1024, 2},
};

uint64_t g_rng = 0x9E3779B97F4A7C15ull;
inline uint64_t rnd() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return g_rng;
}
uint16_t f2h(float f) {  // round to nearest even, normals only (enough for synthetic data)
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    int e = (int)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (e <= 0) return (uint16_t)s;
    if (e >= 31) return (uint16_t)(s | 0x7BFF);
    uint32_t h = (uint32_t)e << 10 | (m >> 13);
    if ((m & 0x1FFF) > 0x1000 || ((m & 0x1FFF) == 0x1000 && (h & 1))) ++h;
    return (uint16_t)(s | h);
}

int zeros_width(int K, int G) {  // calculate_zeros_width, llm/src/nn_modules/cuda/utils.cu:162-178
    const int mult = G >= 128 ? 1 : (G == 64 ? 2 : 4);
    return (((K / G + 7) / 8) + mult - 1) / mult * mult;
}

struct Linear {  // what Linear_half_int4 holds (llm/include/ops/linear.h:215-218)
    int N = 0, K = 0;
    int32_t *weight = nullptr;
    float16_t *scale = nullptr;
    int *zero_point = nullptr;
    void *packed = nullptr;  // capi legs: tce_w4a16_prepack's copy, built by this tool
};

struct HostImage {
    int N, K;
    std::vector<int32_t> w;
    std::vector<uint16_t> s;
    std::vector<int> z;
};
HostImage make_image(int N, int K, int G) {
    HostImage im;
    im.N = N;
    im.K = K;
    const int zw = zeros_width(K, G);
    im.w.resize((size_t)N * (K / 8));
    for (size_t i = 0; i + 1 < im.w.size(); i += 2) {
        const uint64_t r = rnd();
        im.w[i] = (int32_t)r;
        im.w[i + 1] = (int32_t)(r >> 32);
    }
    if (im.w.size() & 1) im.w.back() = (int32_t)rnd();
    im.s.assign((size_t)N * zw * 8, 0);
    for (int n = 0; n < N; ++n)
        for (int g = 0; g < K / G; ++g) im.s[(size_t)n * zw * 8 + g] = f2h(0.003f * (0.5f + (float)(rnd() & 0xFFFF) / 65536.0f));
    im.z.assign((size_t)N * zw, (int)0x88888888u);
    return im;
}
template <typename T>
T *dev_alloc(size_t n) {
    void *p = nullptr;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    return static_cast<T *>(p);
}
Linear upload(const HostImage &im, int G, bool pack) {
    Linear l;
    l.N = im.N;
    l.K = im.K;
    l.weight = dev_alloc<int32_t>(im.w.size() + 1);  // (the reference over-allocates down_proj by a byte too, Int4llamaDecoderLayer.cu:63)
    l.scale = dev_alloc<float16_t>(im.s.size());
    l.zero_point = dev_alloc<int>(im.z.size());
    HIPCHK(hipMemcpy(l.weight, im.w.data(), im.w.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(l.scale, im.s.data(), im.s.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(l.zero_point, im.z.data(), im.z.size() * 4, hipMemcpyHostToDevice));
    if (pack) {
        const size_t need = tce_w4a16_prepack_bytes(l.N, l.K, G);
        if (need) {
            HIPCHK(hipMalloc(&l.packed, need));
            tce_w4a16_desc d;
            std::memset(&d, 0, sizeof(d));
            d.M = 1;
            d.N = l.N;
            d.K = l.K;
            d.group_size = G;
            d.qweight = l.weight;
            d.scales = l.scale;
            d.zeros = l.zero_point;
            TCECHK(tce_w4a16_prepack(&d, l.packed, nullptr));
        }
    }
    return l;
}

struct Layer {
    Linear qkv, o, gate, up, down;
    float *gamma_in, *gamma_post;
    float16_t *k_cache, *v_cache;
};

struct Model {
    Shape sh;
    int G = 128, max_keys = 0, pos = 0;
    std::vector<Layer> layers;
    Linear lm_head;
    float *gamma_final;
    float16_t *cos_t, *sin_t;
    void *attn_ws;
    void *scratch;
    // activations (static arrays sized once, like the reference's: Int4llamaDecoderLayer.cu:34-39)
    float16_t *x, *xn, *qkv_out, *attn_out, *o_out, *resid, *xn2, *g, *u, *d, *logits;
};

enum class Leg { Adapter, CapiEager, CapiGraph };

// Linear_half_int4::forward (llm/src/ops/cuda/linear.cu:5-40), M = 1
inline void linear_forward_adapter(const Linear &l, float16_t *x, float16_t *out, int G) {
    struct matmul_params params;
    params.A.row = 1;
    params.A.column = l.K;
    params.A.half_data_ptr = x;
    params.B.row = l.K / 8;
    params.B.column = l.N;
    params.B.int32_data_ptr = l.weight;
    params.C.row = 1;
    params.C.column = l.N;
    params.C.half_data_ptr = out;
    params.opt_params.num_thread = 8;
    params.half_scales = l.scale;
    params.int32_zero_point = l.zero_point;
    params.block_size = G;
    matmul::MatmulOperator op = matmul::MatmulOperator();
    op.gemv_forward_cuda(&params);
}
inline void linear_forward_capi(const Linear &l, float16_t *x, float16_t *out, int G, void *scratch, hipStream_t st) {
    tce_w4a16_desc d;
    std::memset(&d, 0, sizeof(d));
    d.M = 1;
    d.N = l.N;
    d.K = l.K;
    d.group_size = G;
    d.A = x;
    d.qweight = l.weight;
    d.scales = l.scale;
    d.zeros = l.zero_point;
    d.C = out;
    d.flags = TCE_W4_ZERO_POINT_IS_8;
    d.prepacked = l.packed;
    d.scratch = l.packed ? scratch : nullptr;
    TCECHK(tce_w4a16_forward(&d, st));
}

int g_calls = 0;  // launches issued by the last issue_token

void issue_token(Model &m, Leg leg, hipStream_t st) {
    const Shape &s = m.sh;
    const int H = s.hidden;
    const unsigned short alpha = f2h(1.0f / std::sqrt(128.0f));
    int calls = 0;
    auto lin = [&](const Linear &l, float16_t *x, float16_t *out) {
        if (leg == Leg::Adapter) linear_forward_adapter(l, x, out, m.G);
        else linear_forward_capi(l, x, out, m.G, m.scratch, st);
        ++calls;
    };
    float16_t *x = m.x;
    for (Layer &L : m.layers) {
        TCECHK(tce_rmsnorm_half(x, L.gamma_in, m.xn, 1, H, 1e-5f, st));                                                       // input_layernorm
        lin(L.qkv, m.xn, m.qkv_out);                                                                                          // qkv_proj
        TCECHK(tce_attention_decode_step_gqa_f16(m.qkv_out, L.k_cache, L.v_cache, m.cos_t, m.sin_t, nullptr, m.attn_out, m.attn_ws, s.heads, s.kv_heads, 128,
                                                 m.max_keys, m.pos, alpha, st));                                              // shape_qkv .. unshape
        lin(L.o, m.attn_out, m.o_out);                                                                                        // o_proj
        TCECHK(tce_add_half(x, m.o_out, m.resid, H, st));                                                                     // add_half
        TCECHK(tce_rmsnorm_half(m.resid, L.gamma_post, m.xn2, 1, H, 1e-5f, st));                                              // post_attention_layernorm
        lin(L.gate, m.xn2, m.g);                                                                                              // gate_proj
        lin(L.up, m.xn2, m.u);                                                                                                // up_proj
        TCECHK(tce_silu_mul_half(m.g, m.u, s.ffn, st));                                                                       // SiLuMul_half
        lin(L.down, m.g, m.d);                                                                                                // down_proj
        TCECHK(tce_add_half(m.resid, m.d, m.resid, H, st));                                                                   // add_half (in place, as the reference)
        x = m.resid;
        calls += 6;
    }
    TCECHK(tce_rmsnorm_half(x, m.gamma_final, m.xn, 1, H, 1e-5f, st));
    lin(m.lm_head, m.xn, m.logits);
    g_calls = calls + 1;
}

struct Result {
    double wall_ms_per_token, event_ms_per_token, host_issue_us_per_call;
    int launches;
};

Result run_leg(Model &m, Leg leg, int warmup, int tokens) {
    hipStream_t st = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    if (leg == Leg::CapiGraph) {
        HIPCHK(hipStreamCreate(&cap));
        issue_token(m, Leg::CapiEager, cap);  // (first-use set-up outside the capture)
        HIPCHK(hipStreamSynchronize(cap));
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        issue_token(m, Leg::CapiEager, cap);
        HIPCHK(hipStreamEndCapture(cap, &graph));
        HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIPCHK(hipGraphDestroy(graph));
        st = cap;
    }
    auto one = [&] {
        if (leg == Leg::CapiGraph) HIPCHK(hipGraphLaunch(exec, st));
        else issue_token(m, leg, st);
    };
    for (int i = 0; i < warmup; ++i) one();
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < tokens; ++i) one();
    HIPCHK(hipEventRecord(e1, st));
    const auto t1 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipDeviceSynchronize());
    const auto t2 = std::chrono::steady_clock::now();
    float ev_ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ev_ms, e0, e1));
    Result r;
    r.wall_ms_per_token = std::chrono::duration<double, std::milli>(t2 - t0).count() / tokens;
    r.event_ms_per_token = ev_ms / tokens;
    r.launches = g_calls;
    r.host_issue_us_per_call = std::chrono::duration<double, std::micro>(t1 - t0).count() / ((double)tokens * (leg == Leg::CapiGraph ? 1 : g_calls));
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    if (exec) HIPCHK(hipGraphExecDestroy(exec));
    if (cap) HIPCHK(hipStreamDestroy(cap));
    return r;
}

void print_result(const char *name, const Result &r, bool last) {
    std::printf("\"%s\": {\"tokens_per_s\": %.1f, \"ms_per_token_wall\": %.4f, \"ms_per_token_events\": %.4f, \"host_us_per_call\": %.3f, \"launches_per_token\": %d}%s", name,
                1e3 / r.wall_ms_per_token, r.wall_ms_per_token, r.event_ms_per_token, r.host_issue_us_per_call, r.launches, last ? "" : ", ");
}

}  // namespace

int main(int argc, char **argv) {
    std::string model = "llama3-8b", legs = "adapter,capi_eager,capi_graph";
    int tokens = 200, warmup = 20, keys = 512, layers_override = 0;
    bool prepare = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&] { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--model") model = next();
        else if (a == "--tokens") tokens = std::atoi(next());
        else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--keys") keys = std::atoi(next());
        else if (a == "--layers") layers_override = std::atoi(next());
        else if (a == "--legs") legs = next();
        else if (a == "--prepare") prepare = true;  // call tce_adapter_prepare per linear at load time (the optional hook) instead of packing inside the first token
        else {
            std::printf("usage: %s [--model llama3-8b|llama2-7b|tiny] [--tokens 200] [--warmup 20] [--keys 512] [--layers N] [--legs adapter,capi_eager,capi_graph] [--prepare]\n", argv[0]);
            return 2;
        }
    }
    const Shape *sp = nullptr;
    for (const Shape &s : kShapes)
        if (model == s.name) sp = &s;
    if (!sp) {
        std::printf("unknown model %s\n", model.c_str());
        return 2;
    }
    Model m;
    m.sh = *sp;
    if (layers_override > 0) m.sh.layers = layers_override;
    const Shape &s = m.sh;
    const int G = m.G, H = s.hidden;
    m.max_keys = ((keys + 63) / 64) * 64;
    m.pos = keys - 1;
    const bool need_capi = legs.find("capi") != std::string::npos;

    const int qkv_n = (s.heads + 2 * s.kv_heads) * 128;
    const HostImage im_qkv = make_image(qkv_n, H, G), im_o = make_image(H, H, G), im_gate = make_image(s.ffn, H, G), im_down = make_image(H, s.ffn, G),
                    im_lm = make_image(s.vocab, H, G);
    std::vector<float> ones((size_t)std::max(H, 1), 1.0f);
    auto gamma = [&] {
        float *p = dev_alloc<float>(H);
        HIPCHK(hipMemcpy(p, ones.data(), (size_t)H * 4, hipMemcpyHostToDevice));
        return p;
    };
    const size_t kv_elems = (size_t)s.kv_heads * m.max_keys * 128;
    std::vector<uint16_t> kv_host(kv_elems);
    for (auto &v : kv_host) v = f2h(((float)(rnd() & 0xFFFF) / 65536.0f - 0.5f) * 0.5f);
    for (int li = 0; li < s.layers; ++li) {
        Layer L;
        L.qkv = upload(im_qkv, G, need_capi);
        L.o = upload(im_o, G, need_capi);
        L.gate = upload(im_gate, G, need_capi);
        L.up = upload(im_gate, G, need_capi);
        L.down = upload(im_down, G, need_capi);
        L.gamma_in = gamma();
        L.gamma_post = gamma();
        L.k_cache = dev_alloc<float16_t>(kv_elems);
        L.v_cache = dev_alloc<float16_t>(kv_elems);
        HIPCHK(hipMemcpy(L.k_cache, kv_host.data(), kv_elems * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(L.v_cache, kv_host.data(), kv_elems * 2, hipMemcpyHostToDevice));
        m.layers.push_back(L);
    }
    m.lm_head = upload(im_lm, G, need_capi);
    m.gamma_final = gamma();
    {  // RotaryPosEmb's tables (llm/src/ops/cuda/RotaryPosEmb.cu): [positions][128]
        std::vector<uint16_t> c((size_t)m.max_keys * 128), sn((size_t)m.max_keys * 128);
        for (int p = 0; p < m.max_keys; ++p)
            for (int i = 0; i < 128; ++i) {
                const double inv = std::pow(10000.0, -(double)((i % 64) * 2) / 128.0);
                c[(size_t)p * 128 + i] = f2h((float)std::cos(p * inv));
                sn[(size_t)p * 128 + i] = f2h((float)std::sin(p * inv));
            }
        m.cos_t = dev_alloc<float16_t>(c.size());
        m.sin_t = dev_alloc<float16_t>(c.size());
        HIPCHK(hipMemcpy(m.cos_t, c.data(), c.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(m.sin_t, sn.data(), sn.size() * 2, hipMemcpyHostToDevice));
    }
    const size_t ws = tce_attention_decode_workspace_bytes(s.heads, m.max_keys, 128);
    HIPCHK(hipMalloc(&m.attn_ws, ws));
    HIPCHK(hipMemset(m.attn_ws, 0, ws));
    const size_t sb = tce_w4a16_gemm_scratch_bytes();
    HIPCHK(hipMalloc(&m.scratch, sb));
    HIPCHK(hipMemset(m.scratch, 0, 4096));
    auto act = [&](size_t n) {
        float16_t *p = dev_alloc<float16_t>(n);
        HIPCHK(hipMemset(p, 0, n * 2));
        return p;
    };
    m.x = act(H);
    m.xn = act(H);
    m.qkv_out = act(qkv_n);
    m.attn_out = act(H);
    m.o_out = act(H);
    m.resid = act(H);
    m.xn2 = act(H);
    m.g = act(s.ffn);
    m.u = act(s.ffn);
    m.d = act(H);
    m.logits = act(s.vocab);
    {
        std::vector<uint16_t> xh(H);
        for (auto &v : xh) v = f2h(((float)(rnd() & 0xFFFF) / 65536.0f - 0.5f) * 2.0f);
        HIPCHK(hipMemcpy(m.x, xh.data(), (size_t)H * 2, hipMemcpyHostToDevice));
    }
    HIPCHK(hipDeviceSynchronize());

    double first_token_ms = -1, prepare_ms = -1;
    if (legs.find("adapter") != std::string::npos) {
        if (prepare) {
            const auto t0 = std::chrono::steady_clock::now();
            for (Layer &L : m.layers)
                for (const Linear *l : {&L.qkv, &L.o, &L.gate, &L.up, &L.down}) tce_adapter_prepare(l->weight, l->scale, l->zero_point, l->N, l->K, G);
            tce_adapter_prepare(m.lm_head.weight, m.lm_head.scale, m.lm_head.zero_point, m.lm_head.N, m.lm_head.K, G);
            HIPCHK(hipDeviceSynchronize());
            prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        const auto t0 = std::chrono::steady_clock::now();
        issue_token(m, Leg::Adapter, nullptr);  // the first token: packed copies are built here unless --prepare did it
        HIPCHK(hipDeviceSynchronize());
        first_token_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    std::printf("{\"model\": \"%s\", \"layers\": %d, \"keys\": %d, \"tokens\": %d, \"warmup\": %d, ", s.name, s.layers, keys, tokens, warmup);
    if (first_token_ms >= 0) std::printf("\"adapter_first_token_ms\": %.2f, \"adapter_prepare_ms\": %.2f, \"adapter_device_bytes\": %lld, ", first_token_ms, prepare_ms, tce_adapter_device_bytes());
    Result ra{}, rg{};
    bool have_a = false, have_g = false;
    if (legs.find("adapter") != std::string::npos) {
        ra = run_leg(m, Leg::Adapter, warmup, tokens);
        have_a = true;
        print_result("adapter", ra, false);
    }
    if (legs.find("capi_eager") != std::string::npos) print_result("capi_eager", run_leg(m, Leg::CapiEager, warmup, tokens), false);
    if (legs.find("capi_graph") != std::string::npos) {
        rg = run_leg(m, Leg::CapiGraph, warmup, tokens);
        have_g = true;
        print_result("capi_graph", rg, false);
    }
    // sanity: the token's logits are finite numbers (a NaN would mean a broken data flow, and NaN arithmetic can time differently)
    std::vector<uint16_t> lg(s.vocab);
    HIPCHK(hipMemcpy(lg.data(), m.logits, (size_t)s.vocab * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (uint16_t h : lg) bad += (h & 0x7C00) == 0x7C00;
    if (have_a && have_g) std::printf("\"adapter_over_graph\": %.3f, ", rg.wall_ms_per_token / ra.wall_ms_per_token);
    std::printf("\"non_finite_logits\": %d}\n", bad);
    return bad ? 1 : 0;
}
