// Prompt mel front-end: wav -> log-mel, the reference's MelSpec with mel_spec_type="vocos"
// (model/modules.py:78-146: torchaudio MelSpectrogram(n_fft=1024, win=1024, hop=256, n_mels=100, power=1, center=True,
// norm=None) -> clamp(min=1e-5).log()).
//   STFT  = reflect pad + framing as a STRIDED VIEW (row t = wav_pad[t*hop .. t*hop + n_fft), lda = hop: no im2col) x
//           windowed DFT basis [2F(+pad), n_fft] on the exact-f32 MFMA GEMM
//   |S|   = sqrt(re^2 + im^2)                               (elementwise)
//   mel   = |S| [T, F(+pad)] x filterbank [n_mels, F(+pad)]^T with a log(max(., 1e-5)) epilogue
// Output is [B, T, n_mels] -- the layout CFM.sample wants after its permute(0, 2, 1) (cfm.py:106-108).
// mel_spec_type="bigvgan" (modules.py:33-75, get_bigvgan_mel_spectrogram) is the same pipeline with reflect padding
// (n_fft - hop) / 2 and center=False (T = (nw + 2 pad - n_fft) / hop + 1), |S| = sqrt(re^2 + im^2 + 1e-9) and librosa's
// slaney filterbank as the loaded table: f5_mel_forward_ex(pad, eps).
#include <map>
#include <string>
#include <vector>

#include "elementwise.h"
#include "gemm_dispatch.h"
#include "internal.h"

using namespace f5;
#define fail f5_fail

// out rows have stride `ls` >= nw + 2*pad (a multiple of 4 floats so that every row stays 16-byte aligned)
static __global__ void reflect_pad_kernel(const float* __restrict__ wav, float* __restrict__ out, int B, int nw, int pad, int ls) {
    const long total = (long)B * ls;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / ls);
        int j = (int)(i % ls) - pad;
        float v = 0.f;
        if (j < nw + pad) {
            if (j < 0) j = -j;                   // reflect (no edge repeat), torch pad_mode="reflect"
            if (j >= nw) j = 2 * (nw - 1) - j;
            v = wav[(size_t)b * nw + j];
        }
        out[i] = v;
    }
}
// spec [T, lds] with re in [0, F), im in [F, 2F)  ->  mag [T, ldm], columns >= F zeroed
static __global__ void magnitude_kernel(const float* __restrict__ spec, int lds, float* __restrict__ mag, int ldm, long T, int F,
                                        float eps) {
    const long total = T * ldm;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldm);
        const long t = i / ldm;
        float v = 0.f;
        if (c < F) {
            const float re = spec[t * lds + c], im = spec[t * lds + F + c];
            v = sqrtf(re * re + im * im + eps);
        }
        mag[i] = v;
    }
}

struct f5_mel {
    int n_fft = 0, hop = 0, n_mels = 0, F = 0, ns = 0, kf = 0;
    float *basis = nullptr, *fb = nullptr;
    Arena arena;
    ~f5_mel() {
        if (basis) (void)hipFree(basis);
        if (fb) (void)hipFree(fb);
    }
};

extern "C" int f5_mel_create(int32_t n_fft, int32_t hop, int32_t n_mels, f5_mel** out) {
    if (!out || n_fft <= 0 || hop <= 0 || n_mels <= 0 || (n_fft % 32) || (hop % 4) || (n_mels % 4))
        return fail(F5_EINVAL, "f5_mel_create: need n_fft %% 32 == 0, hop %% 4 == 0, n_mels %% 4 == 0");
    f5_mel* m = new f5_mel();
    m->n_fft = n_fft; m->hop = hop; m->n_mels = n_mels;
    m->F = n_fft / 2 + 1;
    m->ns = round_up(2 * m->F, 4);     // spectrum row: [re | im | pad]
    m->kf = round_up(m->F, 32);        // magnitude row padded to whole f32 K-tiles
    *out = m;
    return F5_OK;
}
extern "C" int f5_mel_destroy(f5_mel* m) {
    if (m) {
        (void)hipDeviceSynchronize();
        delete m;
    }
    return F5_OK;
}
// name = "aux.dft_basis" f32[ns, n_fft] (rows: w*cos for f < F, then w*sin, zero pad) or "aux.mel_fb" f32[n_mels, kf]
extern "C" int f5_mel_load(f5_mel* m, const char* name, const void* dev, const int64_t* shape, int32_t ndim, f5_stream stream) {
    if (!m || !name || !dev || ndim != 2) return fail(F5_EINVAL, "f5_mel_load: bad arguments");
    const std::string n(name);
    float** dst = nullptr;
    if (n == "aux.dft_basis") {
        if (shape[0] != m->ns || shape[1] != m->n_fft) return fail(F5_EINVAL, "aux.dft_basis must be [%d, %d]", m->ns, m->n_fft);
        dst = &m->basis;
    } else if (n == "aux.mel_fb") {
        if (shape[0] != m->n_mels || shape[1] != m->kf) return fail(F5_EINVAL, "aux.mel_fb must be [%d, %d]", m->n_mels, m->kf);
        dst = &m->fb;
    } else {
        return fail(F5_EINVAL, "f5_mel_load: unknown tensor '%s'", name);
    }
    const size_t bytes = (size_t)shape[0] * shape[1] * 4;
    if (*dst) (void)hipFree(*dst);
    HIPCHK(hipMalloc((void**)dst, bytes));
    HIPCHK(hipMemcpyAsync(*dst, dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return F5_OK;
}
// wav f32[B, nw] -> out f32[B, T, n_mels], T = nw / hop + 1 (center=True)
extern "C" int f5_mel_forward(f5_mel* m, const float* wav, int32_t B, int32_t nw, float* out, f5_stream stream) {
    if (!m) return fail(F5_EINVAL, "f5_mel_forward: bad arguments");
    return f5_mel_forward_ex(m, wav, B, nw, m->n_fft / 2, 0.0f, out, stream);
}
// general form: reflect padding `pad` on both sides, frames at multiples of hop inside the padded signal
// (T = (nw + 2 pad - n_fft) / hop + 1), magnitude sqrt(re^2 + im^2 + mag_eps)
extern "C" int f5_mel_forward_ex(f5_mel* m, const float* wav, int32_t B, int32_t nw, int32_t pad, float mag_eps, float* out,
                                 f5_stream stream) {
    if (!m || !wav || !out || B <= 0 || pad < 0) return fail(F5_EINVAL, "f5_mel_forward: bad arguments");
    if (!m->basis || !m->fb) return fail(F5_ESTATE, "f5_mel_forward: aux.dft_basis / aux.mel_fb not loaded");
    if (nw <= pad || nw + 2 * pad < m->n_fft) return fail(F5_EINVAL, "f5_mel_forward: too few samples for the reflect padding / one frame");
    hipStream_t s = (hipStream_t)stream;
    const int T = (nw + 2 * pad - m->n_fft) / m->hop + 1, Lp = round_up(nw + 2 * pad, 4);
    auto plan = [&](Arena& a, float** wp, float** spec, float** mag) {
        a.reset();
        *wp = a.take<float>((size_t)B * Lp + m->n_fft);
        *spec = a.take<float>((size_t)T * m->ns);
        *mag = a.take<float>((size_t)T * m->kf);
        return align_up(a.off, 256) + 256;
    };
    float *wp, *spec, *mag;
    Arena dry;
    const size_t need_b = plan(dry, &wp, &spec, &mag);
    if (need_b > m->arena.cap) {
        HIPCHK(hipDeviceSynchronize());
        if (m->arena.base) (void)hipFree(m->arena.base);
        m->arena.base = nullptr;
        m->arena.cap = 0;
        HIPCHK(hipMalloc((void**)&m->arena.base, need_b));
        HIPCHK(hipMemset(m->arena.base, 0, need_b));
        m->arena.cap = need_b;
    }
    (void)plan(m->arena, &wp, &spec, &mag);
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(ew_blocks((long)B * Lp)), dim3(256), 0, s, wav, wp, B, nw, pad, Lp);
    KCHK();
    for (int b = 0; b < B; ++b) {
        // frames are a strided view of the padded signal: row t starts at t*hop
        HIPCHK(launch_gemm<float>(s, wp + (size_t)b * Lp, m->hop, m->basis, m->n_fft, T, m->ns, m->n_fft,
                                  EpiStore<float>{spec, m->ns, nullptr, F5_ACT_NONE}));
        hipLaunchKernelGGL(magnitude_kernel, dim3(ew_blocks((long)T * m->kf)), dim3(256), 0, s, spec, m->ns, mag, m->kf, (long)T, m->F,
                           mag_eps);
        KCHK();
        HIPCHK(launch_gemm<float>(s, mag, m->kf, m->fb, m->kf, T, m->n_mels, m->kf,
                                  EpiStore<float>{out + (size_t)b * T * m->n_mels, m->n_mels, nullptr, F5_ACT_LOGCLAMP}));
    }
    return F5_OK;
}
