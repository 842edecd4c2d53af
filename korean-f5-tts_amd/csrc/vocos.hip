// Vocos (mel -> waveform) on gfx950: ConvNeXt backbone + ISTFT head, all f32 (the reference feeds the vocoder fp32,
// infer/utils_infer.py:699-703).  Every contraction is an exact-f32 MFMA GEMM (gemm.h):
//   embed Conv1d(C, dim, k=7)  = im2col (7*C columns) x W[dim, 7*C]
//   pwconv1 / pwconv2 / head   = nn.Linear
//   inverse real DFT + window  = S[T, 2F(+pad)] x Basis[n_fft, 2F(+pad)]^T   (Basis host-computed in float64 -> f32,
//                                hann window and 1/n_fft folded in; exact restatement of torch.istft's irfft * window)
// followed by overlap-add / window-envelope normalisation / centre trim (torch.istft(center=True)).
#include <map>
#include <string>
#include <vector>

#include "elementwise.h"
#include "gemm_dispatch.h"
#include "internal.h"

using namespace f5;
#define fail f5_fail

namespace {
struct VBlock {
    float *dwk, *dwb, *lnw, *lnb, *w1, *b1, *w2, *b2, *gamma;
};
struct VT {
    float* p = nullptr;
    std::vector<int64_t> shape;
};
}  // namespace

struct f5_vocos {
    f5_vocos_config cfg{};
    std::map<std::string, VT> raw;
    std::vector<void*> owned;
    bool finalized = false;
    int F = 0, K2 = 0, kemb = 0, head_n = 0;
    float *emb_w = nullptr, *emb_b = nullptr, *n0w = nullptr, *n0b = nullptr, *fnw = nullptr, *fnb = nullptr;
    float *head_w = nullptr, *head_b = nullptr, *hann = nullptr, *basis = nullptr;
    std::vector<VBlock> blocks;
    Arena arena;
    ~f5_vocos() {
        for (auto& kv : raw)
            if (kv.second.p) (void)hipFree(kv.second.p);
        for (void* p : owned) (void)hipFree(p);
    }
};

extern "C" int f5_vocos_create(const f5_vocos_config* c, f5_vocos** out) {
    if (!c || !out) return fail(F5_EINVAL, "f5_vocos_create: null argument");
    if (c->dim % 4 || c->dim > 2048 || c->intermediate_dim % 4 || c->input_channels % 4 || c->n_fft % 4 ||
        c->hop_length <= 0 || c->n_fft % c->hop_length)
        return fail(F5_EINVAL, "f5_vocos_create: unsupported dimensions");
    f5_vocos* v = new f5_vocos();
    v->cfg = *c;
    v->F = c->n_fft / 2 + 1;
    v->K2 = round_up(2 * v->F, 32);            // whole 128-byte K-tiles (f32) for the LDS-DMA GEMM
    v->kemb = round_up(7 * c->input_channels, 32);
    v->head_n = round_up(c->n_fft + 2, 4);
    *out = v;
    return F5_OK;
}
extern "C" int f5_vocos_destroy(f5_vocos* v) {
    if (v) {
        (void)hipDeviceSynchronize();
        delete v;
    }
    return F5_OK;
}
extern "C" int f5_vocos_load_weight(f5_vocos* v, const char* name, const void* dev, const int64_t* shape, int32_t ndim,
                                    f5_stream stream) {
    if (!v || !name || !dev || ndim < 0 || ndim > 4) return fail(F5_EINVAL, "f5_vocos_load_weight: bad arguments");
    VT t;
    t.shape.assign(shape, shape + ndim);
    size_t n = 1;
    for (auto d : t.shape) n *= (size_t)d;
    auto it = v->raw.find(name);
    if (it != v->raw.end()) {
        (void)hipFree(it->second.p);
        v->raw.erase(it);
    }
    HIPCHK(hipMalloc((void**)&t.p, std::max<size_t>(n * 4, 16)));
    HIPCHK(hipMemcpyAsync(t.p, dev, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    v->raw[name] = t;
    v->finalized = false;
    return F5_OK;
}

static int vneed(f5_vocos* v, const std::string& n, std::vector<int64_t> shape, const VT** out) {
    auto it = v->raw.find(n);
    if (it == v->raw.end()) return fail(F5_ESTATE, "missing vocos weight '%s'", n.c_str());
    if (it->second.shape != shape) return fail(F5_EINVAL, "vocos weight '%s' has the wrong shape", n.c_str());
    *out = &it->second;
    return F5_OK;
}
static int vcopy(f5_vocos* v, hipStream_t s, const std::string& n, std::vector<int64_t> shape, float** out) {
    const VT* t = nullptr;
    CHK(vneed(v, n, shape, &t));
    size_t cnt = 1;
    for (auto d : shape) cnt *= (size_t)d;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(cnt * 4, 16)));
    v->owned.push_back(p);
    HIPCHK(hipMemcpyAsync(p, t->p, cnt * 4, hipMemcpyDeviceToDevice, s));
    *out = (float*)p;
    return F5_OK;
}

extern "C" int f5_vocos_finalize(f5_vocos* v, f5_stream stream) {
    if (!v) return fail(F5_EINVAL, "null vocos");
    hipStream_t s = (hipStream_t)stream;
    const f5_vocos_config& c = v->cfg;
    const int C = c.input_channels, D = c.dim, I = c.intermediate_dim;
    for (void* p : v->owned) (void)hipFree(p);
    v->owned.clear();
    // embed conv [D, C, 7] -> [D, 7*C]
    const VT* t = nullptr;
    CHK(vneed(v, "backbone.embed.weight", {D, C, 7}, &t));
    void* p = nullptr;
    {
        Scratch<float> tmp;  // [D, 7*C] tap-major, then zero-padded to kemb columns
        HIPCHK(tmp.alloc((size_t)D * 7 * C));
        hipLaunchKernelGGL((permute_last2_kernel<float>), dim3(ew_blocks((long)D * 7 * C)), dim3(256), 0, s, t->p, tmp.p,
                           (long)D, C, 7);
        HIPCHK(hipMalloc(&p, (size_t)D * v->kemb * 4));
        v->owned.push_back(p);
        v->emb_w = (float*)p;
        hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(ew_blocks((long)D * v->kemb)), dim3(256), 0, s, tmp.p, 7 * C, D,
                           7 * C, v->emb_w, v->kemb, D);
        KCHK();
        HIPCHK(hipStreamSynchronize(s));
    }
    CHK(vcopy(v, s, "backbone.embed.bias", {D}, &v->emb_b));
    CHK(vcopy(v, s, "backbone.norm.weight", {D}, &v->n0w));
    CHK(vcopy(v, s, "backbone.norm.bias", {D}, &v->n0b));
    CHK(vcopy(v, s, "backbone.final_layer_norm.weight", {D}, &v->fnw));
    CHK(vcopy(v, s, "backbone.final_layer_norm.bias", {D}, &v->fnb));
    v->blocks.resize(c.num_layers);
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string pf = "backbone.convnext." + std::to_string(i);
        VBlock& b = v->blocks[i];
        CHK(vneed(v, pf + ".dwconv.weight", {D, 1, 7}, &t));
        HIPCHK(hipMalloc(&p, (size_t)7 * D * 4));
        v->owned.push_back(p);
        b.dwk = (float*)p;
        hipLaunchKernelGGL((permute_last2_kernel<float>), dim3(ew_blocks(7L * D)), dim3(256), 0, s, t->p, b.dwk, 1L, D, 7);
        KCHK();
        CHK(vcopy(v, s, pf + ".dwconv.bias", {D}, &b.dwb));
        CHK(vcopy(v, s, pf + ".norm.weight", {D}, &b.lnw));
        CHK(vcopy(v, s, pf + ".norm.bias", {D}, &b.lnb));
        CHK(vcopy(v, s, pf + ".pwconv1.weight", {I, D}, &b.w1));
        CHK(vcopy(v, s, pf + ".pwconv1.bias", {I}, &b.b1));
        CHK(vcopy(v, s, pf + ".pwconv2.weight", {D, I}, &b.w2));
        CHK(vcopy(v, s, pf + ".pwconv2.bias", {D}, &b.b2));
        CHK(vcopy(v, s, pf + ".gamma", {D}, &b.gamma));
    }
    // head: [n_fft + 2, D] padded to head_n rows (zero rows / zero bias)
    CHK(vneed(v, "head.out.weight", {c.n_fft + 2, D}, &t));
    HIPCHK(hipMalloc(&p, (size_t)v->head_n * D * 4));
    v->owned.push_back(p);
    v->head_w = (float*)p;
    hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(ew_blocks((long)v->head_n * D)), dim3(256), 0, s, t->p, D, c.n_fft + 2,
                       D, v->head_w, D, v->head_n);
    KCHK();
    CHK(vneed(v, "head.out.bias", {c.n_fft + 2}, &t));
    HIPCHK(hipMalloc(&p, (size_t)v->head_n * 4));
    v->owned.push_back(p);
    v->head_b = (float*)p;
    hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(1), dim3(256), 0, s, t->p, c.n_fft + 2, 1, c.n_fft + 2, v->head_b,
                       v->head_n, 1);
    KCHK();
    CHK(vcopy(v, s, "aux.hann", {c.n_fft}, &v->hann));
    CHK(vcopy(v, s, "aux.idft_basis", {c.n_fft, v->K2}, &v->basis));
    HIPCHK(hipStreamSynchronize(s));
    for (auto& kv : v->raw)
        if (kv.second.p) (void)hipFree(kv.second.p);
    v->raw.clear();
    v->finalized = true;
    return F5_OK;
}

extern "C" int f5_vocos_decode(f5_vocos* v, const float* mel, int32_t B, int32_t T, float* wav, f5_stream stream) {
    if (!v) return fail(F5_EINVAL, "f5_vocos_decode: null argument");
    return f5_vocos_decode_strided(v, mel, B, T, (int64_t)v->cfg.input_channels * T, T, 1, wav, stream);
}

extern "C" int f5_vocos_decode_strided(f5_vocos* v, const float* mel, int32_t B, int32_t T, int64_t stride_b, int64_t stride_c,
                                       int64_t stride_t, float* wav, f5_stream stream) {
    if (!v || !mel || !wav) return fail(F5_EINVAL, "f5_vocos_decode: null argument");
    if (!v->finalized) return fail(F5_ESTATE, "f5_vocos_finalize has not been called");
    if (B <= 0 || T < 2) return fail(F5_EINVAL, "f5_vocos_decode: need B >= 1 and T >= 2 frames");
    hipStream_t s = (hipStream_t)stream;
    const f5_vocos_config& c = v->cfg;
    const int C = c.input_channels, D = c.dim, I = c.intermediate_dim, nfft = c.n_fft;
    const long R = (long)B * T;
    // workspace
    auto plan = [&](Arena& a, float** col, float** x, float** t1, float** h, float** hd, float** S, float** fr) {
        a.reset();
        *col = a.take<float>((size_t)R * v->kemb);
        *x = a.take<float>((size_t)R * D);
        *t1 = a.take<float>((size_t)R * D);
        *h = a.take<float>((size_t)R * I);
        *hd = a.take<float>((size_t)R * v->head_n);
        *S = a.take<float>((size_t)R * v->K2);
        *fr = a.take<float>((size_t)R * nfft);
        return align_up(a.off, 256) + 256;
    };
    float *col, *x, *t1, *h, *hd, *S, *fr;
    Arena dry;
    const size_t need_b = plan(dry, &col, &x, &t1, &h, &hd, &S, &fr);
    if (need_b > v->arena.cap) {
        HIPCHK(hipDeviceSynchronize());
        if (v->arena.base) (void)hipFree(v->arena.base);
        v->arena.base = nullptr;
        v->arena.cap = 0;
        HIPCHK(hipMalloc((void**)&v->arena.base, need_b));
        v->arena.cap = need_b;
    }
    (void)plan(v->arena, &col, &x, &t1, &h, &hd, &S, &fr);

    hipLaunchKernelGGL(im2col7_kernel, dim3(ew_blocks(R * v->kemb)), dim3(256), 0, s, mel, (long)stride_b, (long)stride_c,
                       (long)stride_t, col, B, C, T, v->kemb);
    KCHK();
    HIPCHK(launch_gemm<float>(s, col, v->kemb, v->emb_w, v->kemb, (int)R, D, v->kemb, EpiStore<float>{t1, D, v->emb_b, F5_ACT_NONE}));
    hipLaunchKernelGGL((layernorm_kernel<float>), dim3((R + 3) / 4), dim3(256), 0, s, t1, D, x, D, (int)R, D, 1e-6f, v->n0w,
                       v->n0b, 0, 0, 0, Prefetch{});
    KCHK();
    for (auto& b : v->blocks) {
        hipLaunchKernelGGL(dwconv7_ln_kernel, dim3((R + 3) / 4), dim3(256), 0, s, x, t1, b.dwk, b.dwb, b.lnw, b.lnb, B, T, D,
                           (const int*)nullptr, 1e-6f);
        KCHK();
        HIPCHK(launch_gemm<float>(s, t1, D, b.w1, D, (int)R, I, D, EpiStore<float>{h, I, b.b1, F5_ACT_GELU_ERF}));
        // x = x + gamma * (pwconv2(h) + bias)   (layer scale == a gate vector shared by every row)
        HIPCHK(launch_gemm<float>(s, h, I, b.w2, I, (int)R, D, I, EpiGateRes{x, x, D, b.b2, b.gamma, 0, (int)R + 1, nullptr}));
    }
    hipLaunchKernelGGL((layernorm_kernel<float>), dim3((R + 3) / 4), dim3(256), 0, s, x, D, t1, D, (int)R, D, 1e-6f, v->fnw,
                       v->fnb, 0, 0, 0, Prefetch{});
    KCHK();
    HIPCHK(launch_gemm<float>(s, t1, D, v->head_w, D, (int)R, v->head_n, D, EpiStore<float>{hd, v->head_n, v->head_b, F5_ACT_NONE}));
    hipLaunchKernelGGL(istft_spec_kernel, dim3(ew_blocks(R * v->K2)), dim3(256), 0, s, hd, v->head_n, S, v->K2, R, v->F);
    KCHK();
    HIPCHK(launch_gemm<float>(s, S, v->K2, v->basis, v->K2, (int)R, nfft, v->K2, EpiStore<float>{fr, nfft, nullptr, F5_ACT_NONE}));
    hipLaunchKernelGGL(istft_ola_kernel, dim3(ew_blocks((long)B * (T - 1) * c.hop_length)), dim3(256), 0, s, fr, v->hann, wav,
                       B, T, nfft, c.hop_length);
    KCHK();
    return F5_OK;
}
