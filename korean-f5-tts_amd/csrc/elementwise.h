// Bandwidth-bound kernels of the F5-TTS path (norms, packing, Euler/CFG update, text-encoder and Vocos glue).
// All of them stream rows with 16-byte accesses, one wave (64 lanes) per row where a row reduction is needed
// (wave shuffles only, no LDS), and keep statistics in fp32.
#pragma once
#include "f5_common.h"

namespace f5 {

// ------------------------------------------------------------------------------------------- LayerNorm
// out[r, :] = LN(x[r, :]; eps) * A + B with either
//   modulate: A = 1 + scale[b(r), :], B = shift[b(r), :]   (AdaLayerNorm, modules.py:314-320,335-341,680-693)
//   affine  : A = weight, B = bias                         (nn.LayerNorm(dim), modules.py:259; Vocos norms)
// One wave per row; D <= 2048, D % 4 == 0.  Two-pass statistics in registers (mean, then centred variance).
//
// Prefetch: up to two byte ranges (the weights of the GEMMs that follow) are read and discarded, spread over the grid,
// requested AFTER the row so that they never delay it.  The 370 MB of block weights do not fit the 256 MB Infinity
// Cache, so every GEMM of a step otherwise starts with HBM-latency misses on its W tiles; this launch is latency-bound
// and has the memory system to itself.
struct Prefetch { const char* p0 = nullptr; size_t n0 = 0; const char* p1 = nullptr; size_t n1 = 0; };
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, TO* __restrict__ out,
                                                        int ldo, int R, int D, float eps, const float* __restrict__ A,
                                                        const float* __restrict__ Bv, int vec_stride,
                                                        int rows_per_batch, int modulate, Prefetch pf,
                                                        const int* __restrict__ m_limit = nullptr, int planar = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R || (m_limit && r >= *m_limit)) return;   // (m_limit: rows present in a packed batch, RowPack)
    const float* xr = x + (size_t)r * ldx;
    float4 v[8], ga[8], gb[8];
    // the modulation / affine vectors do not depend on the statistics: request them together with the row so that the
    // kernel is ONE memory round trip deep, not two
    const size_t vb = (size_t)(rows_per_batch > 0 ? r / rows_per_batch : 0) * vec_stride;
    const float a0 = modulate ? 0.f : 1.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
        ga[i] = (A && c < D) ? *reinterpret_cast<const float4*>(A + vb + c) : make_float4(a0, a0, a0, a0);
        gb[i] = (Bv && c < D) ? *reinterpret_cast<const float4*>(Bv + vb + c) : make_float4(0, 0, 0, 0);
    }
    u32x4 pfv[8] = {};
    if (pf.n0 | pf.n1) {
        const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t off = (gtid + (size_t)(i & 3) * gsz) * 16;   // four 16-byte chunks per thread and range
            const char* base = i < 4 ? pf.p0 : pf.p1;
            const size_t n = i < 4 ? pf.n0 : pf.n1;
            if (off + 16 <= n) pfv[i] = *reinterpret_cast<const u32x4*>(base + off);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < D) {
            float4 a = ga[i];
            const float4 b = gb[i];
            if (modulate) { a.x += 1.f; a.y += 1.f; a.z += 1.f; a.w += 1.f; }
            store4_at(out + (size_t)r * ldo, c, planar, (v[i].x - mean) * rstd * a.x + b.x, (v[i].y - mean) * rstd * a.y + b.y,
                      (v[i].z - mean) * rstd * a.z + b.z, (v[i].w - mean) * rstd * a.w + b.w);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(pfv[i]));   // keep the prefetch loads; nothing uses their data
}

// x_transformers.RMSNorm as used by UNetT (unett.py:154,168,185): F.normalize(x, dim=-1) * sqrt(D) * g
template <typename TO>
__global__ __launch_bounds__(256) void xrmsnorm_kernel(const float* __restrict__ x, int ldx, TO* __restrict__ out, int ldo,
                                                       int R, int D, const float* __restrict__ gvec, int planar = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* xr = x + (size_t)r * ldx;
    float4 v[8];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    const float sc = sqrtf((float)D) / nrm;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < D) {
            const float4 gg = *reinterpret_cast<const float4*>(gvec + c);
            store4_at(out + (size_t)r * ldo, c, planar, v[i].x * sc * gg.x, v[i].y * sc * gg.y, v[i].z * sc * gg.z, v[i].w * sc * gg.w);
        }
    }
}

// --------------------------------------------------------------------------------- input-embedding operand
// A[b', n, :] = [ x[b, n, :mel] | cond[b, n, :mel] (0 for the uncond half / drop_audio_cond) | text[b', n, :Dt] ]
// (dit.py:135-138, cfg_infer packing dit.py:296-305).  b' in [0, Bp); rows b' >= B are the uncond half.
// rowmap (RowPack, may be null): output row r is (b', n) = rowmap[r] for r < *row_limit instead of r = b' N + n; rows
// whose position lies past the sample's own length (the <= 3 alignment rows of a packed utterance) are written as zero.
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                  const float* __restrict__ text_c, const float* __restrict__ text_u,
                                  T* __restrict__ out, int ldo, int B, int Bp, int N, int mel, int Dt, int drop_cond_first,
                                  const int2* __restrict__ rowmap = nullptr, const int* __restrict__ row_limit = nullptr,
                                  const int* __restrict__ lens = nullptr) {
    const int K4 = (2 * mel + Dt) / 4;
    const long total = (rowmap ? (long)*row_limit : (long)Bp * N) * K4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % K4) * 4;
        const long row = idx / K4;
        int n = (int)(row % N), bp = (int)(row / N);
        if (rowmap) { const int2 m = rowmap[row]; bp = m.x; n = m.y; }
        const bool un = bp >= B;
        const int b = un ? bp - B : bp;
        float4 v;
        if (rowmap && n >= lens[b]) v = make_float4(0, 0, 0, 0);
        else if (c < mel) v = *reinterpret_cast<const float4*>(x + ((size_t)b * N + n) * mel + c);
        else if (c < 2 * mel)
            v = (un || drop_cond_first) ? make_float4(0, 0, 0, 0)
                                        : *reinterpret_cast<const float4*>(cond + ((size_t)b * N + n) * mel + (c - mel));
        else
            v = *reinterpret_cast<const float4*>((un ? text_u : text_c) + ((size_t)b * N + n) * Dt + (c - 2 * mel));
        store4(out + (size_t)row * ldo + c, v.x, v.y, v.z, v.w);
    }
}

// ------------------------------------------------------------------------------------------ UNetT glue
// x[b', 0, :] = t_emb[b' or shared]; x[b', 1 + n, :] = h[b', n, :]      (unett.py:244: cat([t[:, None], x], dim=1))
static __global__ void unett_assemble_kernel(const float* __restrict__ h, const float* __restrict__ temb, int temb_stride,
                                             float* __restrict__ x, int Bp, int N, int D) {
    const int d4 = D / 4;
    const long total = (long)Bp * (N + 1) * d4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4);
        const long row = i / d4;
        const int n = (int)(row % (N + 1)), b = (int)(row / (N + 1));
        const float4 v = n == 0 ? reinterpret_cast<const float4*>(temb + (size_t)b * temb_stride)[c]
                                : reinterpret_cast<const float4*>(h + ((size_t)b * N + (n - 1)) * D)[c];
        reinterpret_cast<float4*>(x)[i] = v;
    }
}
// out[r, :] = [x[r, :] | skip[r, :]] converted to T   (unett.py:266: cat((x, skip), dim=-1))
// planar (T = float, F5_PREC_F16X3): the row is written pre-split (store4_planar) -- the A operand of a MODE 5 / ping-pong GEMM
template <typename T>
__global__ void cat2_kernel(const float* __restrict__ x, const float* __restrict__ skip, T* __restrict__ out, long rows, int D, int planar = 0) {
    const int d4 = D / 4;
    const long total = rows * 2 * d4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (2 * d4));
        const long r = i / (2 * d4);
        const float4 v = c < d4 ? reinterpret_cast<const float4*>(x + r * D)[c]
                                : reinterpret_cast<const float4*>(skip + r * D)[c - d4];
        store4_at(out + r * 2 * D, c * 4, planar, v.x, v.y, v.z, v.w);
    }
}
// pred[b', n, :] = pred_all[b', 1 + n, :]     (unett.py:278: norm_out(x)[:, 1:, :])
static __global__ void strip_first_token_kernel(const float* __restrict__ in, float* __restrict__ out, int Bp, int N, int C) {
    const int c4 = C / 4;
    const long total = (long)Bp * N * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        const long row = i / c4;
        const int n = (int)(row % N), b = (int)(row / N);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(in + ((size_t)b * (N + 1) + n + 1) * C)[c];
    }
}

// ------------------------------------------------------------------------------------- CFG + Euler update
// y += dt * (p + (p - u) * cfg)   (cfm.py:190-191 + fixed-grid Euler, f5_tts_trtllm.py:360-369); also emits the new
// state into the trajectory slot.  pred holds [cond half ; uncond half] when cfg is on.
static __global__ void euler_cfg_kernel(float* __restrict__ y, const float* __restrict__ pred, long half_elems,
                                        const float* __restrict__ tgrid, int step, float cfg, int use_cfg,
                                        float* __restrict__ traj_slot) {
    const float dt = tgrid[step + 1] - tgrid[step];   // f32 subtraction, as torch does on the f32 grid (cfm.py:211-218)
    const long n4 = half_elems / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 yy = reinterpret_cast<float4*>(y)[i];
        const float4 p = reinterpret_cast<const float4*>(pred)[i];
        float4 v = p;
        if (use_cfg) {
            const float4 u = reinterpret_cast<const float4*>(pred + half_elems)[i];
            v.x = p.x + (p.x - u.x) * cfg; v.y = p.y + (p.y - u.y) * cfg;
            v.z = p.z + (p.z - u.z) * cfg; v.w = p.w + (p.w - u.w) * cfg;
        }
        yy.x += dt * v.x; yy.y += dt * v.y; yy.z += dt * v.z; yy.w += dt * v.w;
        reinterpret_cast<float4*>(y)[i] = yy;
        if (traj_slot) reinterpret_cast<float4*>(traj_slot)[i] = yy;
    }
}

// The same update when the backbone ran on PACKED rows (RowPack): pred[r, :] with r = row_start[b'] + n for n < lens[b];
// frames past a sample's own length keep their value (they influence nothing when attn_mask_enabled, and are not computed).
static __global__ void euler_cfg_packed_kernel(float* __restrict__ y, const float* __restrict__ pred, int B, int N, int mel,
                                               const int* __restrict__ row_start, const int* __restrict__ lens,
                                               const float* __restrict__ tgrid, int step, float cfg, int use_cfg,
                                               float* __restrict__ traj_slot) {
    const float dt = tgrid[step + 1] - tgrid[step];
    const int c4n = mel / 4;
    const long total = (long)B * N * c4n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n);
        const long fr = i / c4n;
        const int n = (int)(fr % N), b = (int)(fr / N);
        float4 yy = reinterpret_cast<float4*>(y)[i];
        if (n < lens[b]) {
            const float4 p = reinterpret_cast<const float4*>(pred + ((size_t)row_start[b] + n) * mel)[c];
            float4 v = p;
            if (use_cfg) {
                const float4 u = reinterpret_cast<const float4*>(pred + ((size_t)row_start[B + b] + n) * mel)[c];
                v.x = p.x + (p.x - u.x) * cfg; v.y = p.y + (p.y - u.y) * cfg;
                v.z = p.z + (p.z - u.z) * cfg; v.w = p.w + (p.w - u.w) * cfg;
            }
            yy.x += dt * v.x; yy.y += dt * v.y; yy.z += dt * v.z; yy.w += dt * v.w;
            reinterpret_cast<float4*>(y)[i] = yy;
        }
        if (traj_slot) reinterpret_cast<float4*>(traj_slot)[i] = yy;
    }
}
// rowmap[row_start[b'] + n] = (b', n) for n < row_start[b' + 1] - row_start[b']   (one block per batch row)
static __global__ void fill_rowmap_kernel(const int* __restrict__ row_start, int2* __restrict__ rowmap) {
    const int bp = blockIdx.x, r0 = row_start[bp], cnt = row_start[bp + 1] - r0;
    for (int n = threadIdx.x; n < cnt; n += blockDim.x) rowmap[r0 + n] = make_int2(bp, n);
}

// out = where(mask, a, b) row-wise: cfm.py:151-153 (step_cond) and :221-223 (final out)
static __global__ void select_rows_kernel(const float* a, const float* bsrc, const unsigned char* __restrict__ rowmask,
                                   float* out, long rows, int C) {  // out may alias bsrc (in-place select)
    const int c4 = C / 4;
    const long total = rows * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const float* src = rowmask[r] ? a : bsrc;
        float4 v = make_float4(0, 0, 0, 0);
        if (src) v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ------------------------------------------------------------------------------------ time-step features
// feat[s, j] = sin(arg), feat[s, 128 + j] = cos(arg), arg = (1000 * t[s]) * freq[j]   (modules.py:157-164)
static __global__ void time_sinus_kernel(const float* __restrict__ t, const float* __restrict__ freq, float* __restrict__ feat,
                                  int S, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * half) return;
    const int s = i / half, j = i - s * half;
    const float arg = (1000.0f * t[s]) * freq[j];
    feat[(size_t)s * 2 * half + j] = sinf(arg);
    feat[(size_t)s * 2 * half + half + j] = cosf(arg);
}

static __global__ void act_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int act) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = apply_act(in[i], act);
}

// ------------------------------------------------------------------------------------------ text encoder
// h[b, n, :] = E[id] + pos[n]  with id = text[b, n] + 1 (0 = filler, also for n >= nt), drop_text -> id = 0
// (dit.py:87-101).  Rows n >= lens[b] are written as zero; with mask_padding filler rows are zeroed (dit.py:104-105).
static __global__ void text_embed_kernel(const long long* __restrict__ text, int nt, const float* __restrict__ E,
                                  const float* __restrict__ pos, float* __restrict__ out, int B, int N, int Dt,
                                  const int* __restrict__ lens, int drop_text, int mask_padding, int add_pos,
                                  int pos_clamp) {
    const int d4 = Dt / 4;
    const long total = (long)B * N * d4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % d4) * 4;
        const long row = idx / d4;
        const int n = (int)(row % N), b = (int)(row / N);
        const int len = lens ? lens[b] : N;
        long long id = (n < nt) ? text[(size_t)b * nt + n] + 1 : 0;
        const bool filler = (id == 0);
        if (drop_text) id = 0;
        float4 v = make_float4(0, 0, 0, 0);
        if (n < len && !(mask_padding && filler)) {
            v = *reinterpret_cast<const float4*>(E + (size_t)id * Dt + c);
            if (add_pos) {
                const int pn = n < pos_clamp ? n : pos_clamp - 1;
                const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)pn * Dt + c);
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
        }
        reinterpret_cast<float4*>(out)[idx] = v;
    }
}

// zero the rows whose (shifted) token id is the filler (text_mask_padding=True, dit.py:106-108)
static __global__ void zero_filler_rows_kernel(const long long* __restrict__ text, int nt, float* __restrict__ h, int B, int N,
                                        int C) {
    const int c4 = C / 4;
    const long total = (long)B * N * c4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / c4;
        const int n = (int)(row % N), b = (int)(row / N);
        const long long id = (n < nt) ? text[(size_t)b * nt + n] + 1 : 0;
        if (id == 0) reinterpret_cast<float4*>(h)[idx] = make_float4(0, 0, 0, 0);
    }
}

// Depthwise Conv1d(k=7, pad=3) over the sequence + LayerNorm(C, affine, eps)   (ConvNeXt blocks: modules.py:256-270,
// Vocos ConvNeXtBlock).  x, out: [B, N, C] fp32; wk: [7][C] (repacked), one wave per token.  Tokens outside
// [0, lens[b]) read as zero (each sample is convolved at its own length, dit.py:247-258).
static __global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const float* __restrict__ wk, const float* __restrict__ wb,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                         int B, int N, int C, const int* __restrict__ lens, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= (long)B * N) return;
    const int n = (int)(row % N), b = (int)(row / N);
    const int len = lens ? lens[b] : N;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        v[i] = make_float4(0, 0, 0, 0);
        if (c < C) {
            float4 a = *reinterpret_cast<const float4*>(wb + c);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int nn = n + k - 3;
                if (nn >= 0 && nn < len) {
                    const float4 xv = *reinterpret_cast<const float4*>(x + ((size_t)b * N + nn) * C + c);
                    const float4 w = *reinterpret_cast<const float4*>(wk + (size_t)k * C + c);
                    a.x += w.x * xv.x; a.y += w.y * xv.y; a.z += w.z * xv.z; a.w += w.w * xv.w;
                }
            }
            v[i] = a;
            s += (a.x + a.y) + (a.z + a.w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < C) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < C) {
            const float4 w = *reinterpret_cast<const float4*>(lnw + c), bi = *reinterpret_cast<const float4*>(lnb + c);
            *reinterpret_cast<float4*>(out + (size_t)row * C + c) =
                make_float4((v[i].x - mean) * rstd * w.x + bi.x, (v[i].y - mean) * rstd * w.y + bi.y,
                            (v[i].z - mean) * rstd * w.z + bi.z, (v[i].w - mean) * rstd * w.w + bi.w);
        }
    }
}

// GRN (modules.py:231-240): Gx[b, c] = ||h[b, :len, c]||_2 over the SEQUENCE; Nx = Gx / (mean_c Gx + 1e-6);
// h <- gamma * (h * Nx) + beta + h.   Deterministic two-stage reduction: P partial sums per (b, c), combined in order.
constexpr int GRN_P = 16;
static __global__ void grn_partial_kernel(const float* __restrict__ h, float* __restrict__ part, int N, int C,
                                   const int* __restrict__ lens) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y, b = blockIdx.z;
    if (c >= C) return;
    const int len = lens ? lens[b] : N;
    float s = 0.f;
    for (int n = p; n < len; n += GRN_P) {
        const float v = h[((size_t)b * N + n) * C + c];
        s += v * v;
    }
    part[((size_t)b * GRN_P + p) * C + c] = s;
}
static __global__ __launch_bounds__(256) void grn_apply_kernel(float* __restrict__ h, const float* __restrict__ part,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int N, int C, const int* __restrict__ lens, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* gx = reinterpret_cast<float*>(smem_raw);  // [C] then [4] wave partials
    const int b = blockIdx.y, tid = threadIdx.x;
    float loc = 0.f;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < GRN_P; ++p) s += part[((size_t)b * GRN_P + p) * C + c];
        const float gv = sqrtf(s);
        gx[c] = gv;
        loc += gv;
    }
    loc = wave_sum(loc);
    if ((tid & 63) == 0) gx[C + (tid >> 6)] = loc;
    __syncthreads();
    const float mean = (gx[C] + gx[C + 1] + gx[C + 2] + gx[C + 3]) / (float)C;
    const float inv = 1.0f / (mean + 1e-6f);
    const int len = lens ? lens[b] : N;
    const int n0 = blockIdx.x * rows_per_block;
    const int c4n = C / 4;
    for (int i = tid; i < rows_per_block * c4n; i += 256) {
        const int n = n0 + i / c4n, c = (i % c4n) * 4;
        if (n >= len) continue;
        float4* ptr = reinterpret_cast<float4*>(h + ((size_t)b * N + n) * C + c);
        float4 v = *ptr;
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
        v.x = ga.x * (v.x * (gx[c] * inv)) + be.x + v.x;
        v.y = ga.y * (v.y * (gx[c + 1] * inv)) + be.y + v.y;
        v.z = ga.z * (v.z * (gx[c + 2] * inv)) + be.z + v.z;
        v.w = ga.w * (v.w * (gx[c + 3] * inv)) + be.w + v.w;
        *ptr = v;
    }
}

// zero rows n >= lens[b] of [B, N, C]
static __global__ void zero_tail_rows_kernel(float* __restrict__ h, int B, int N, int C, const int* __restrict__ lens) {
    const int c4 = C / 4;
    const long total = (long)B * N * c4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / c4;
        const int n = (int)(row % N), b = (int)(row / N);
        if (n >= lens[b]) reinterpret_cast<float4*>(h)[idx] = make_float4(0, 0, 0, 0);
    }
}

// zipvoice-style late average upsampling of the text embedding (dit.py:54-84; text_embedding_average_upsampling=True): the VALID
// tokens of a sample (original id != filler, position < the sample's own length) are repeated to fill its audio length, the
// last `remainder` tokens once more than the others; rows past the length and samples without a valid token are zero.
// One workgroup per sample; `in` -> `out` (different buffers).
static __global__ __launch_bounds__(256) void text_avg_upsample_kernel(const long long* __restrict__ text, int nt, const float* __restrict__ in,
                                                                        float* __restrict__ out, int N, int Dt, const int* __restrict__ lens) {
    extern __shared__ int valid_idx[];                    // [N]: positions of the valid tokens, in order
    __shared__ int n_valid;
    const int b = blockIdx.x;
    const int alen = lens ? min(lens[b], N) : N;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int n = 0; n < alen && n < nt; ++n)
            if (text[(size_t)b * nt + n] + 1 != 0) valid_idx[c++] = n;
        n_valid = c;
    }
    __syncthreads();
    const int tl = n_valid, d4 = Dt / 4;
    const int base = tl > 0 ? alen / tl : 0, rem = tl > 0 ? alen % tl : 0;
    const int split = (tl - rem) * base;                  // output rows fed by the tokens that repeat `base` times
    for (long idx = threadIdx.x; idx < (long)N * d4; idx += blockDim.x) {
        const int n = (int)(idx / d4), c = (int)(idx % d4);
        float4 v = make_float4(0, 0, 0, 0);
        if (n < alen && tl > 0) {
            const int j = n < split ? n / base : (tl - rem) + (n - split) / (base + 1);
            v = reinterpret_cast<const float4*>(in + ((size_t)b * N + valid_idx[j]) * Dt)[c];
        }
        reinterpret_cast<float4*>(out + ((size_t)b * N + n) * Dt)[c] = v;
    }
}

// qk_norm = "rms_norm" (modules.py:397-404,481-497): q and k, as the QKV epilogue left them WITHOUT rotary and scale
// ([Bp, H, N, 64] each), get RMSNorm(64, eps) with their own gain, then the rotary embedding on the first pe_heads heads, then q its
// softmax scale -- in place.  16 lanes per (batch row, head, position): 4 consecutive dims per lane = two rotary pairs.
template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(T* __restrict__ q, T* __restrict__ k, const float* __restrict__ gq,
                                                          const float* __restrict__ gk, const float* __restrict__ rope_cos,
                                                          const float* __restrict__ rope_sin, long rows, int N, int H, int pe_heads,
                                                          float q_scale, float eps) {
    const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long r = gid >> 4;                              // (b * H + h) * N + pos
    if (r >= rows) return;
    const int d = (int)(gid & 15) * 4;
    const int pos = (int)(r % N), h = (int)((r / N) % H);
    float2 cs = make_float2(1, 1), sn = make_float2(0, 0);
    const bool rot = h < pe_heads;
    if (rot) {
        cs = *reinterpret_cast<const float2*>(rope_cos + (size_t)pos * 32 + (d >> 1));
        sn = *reinterpret_cast<const float2*>(rope_sin + (size_t)pos * 32 + (d >> 1));
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        T* p = (which ? k : q) + r * 64 + d;
        const float4 g = *reinterpret_cast<const float4*>((which ? gk : gq) + d);
        float a0 = to_f32(p[0]), a1 = to_f32(p[1]), a2 = to_f32(p[2]), a3 = to_f32(p[3]);
        float ss = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
        const float rs = rsqrtf(ss * (1.0f / 64.0f) + eps);
        a0 = a0 * rs * g.x; a1 = a1 * rs * g.y; a2 = a2 * rs * g.z; a3 = a3 * rs * g.w;
        if (rot) {
            const float r0 = a0 * cs.x - a1 * sn.x, r1 = a1 * cs.x + a0 * sn.x;
            const float r2 = a2 * cs.y - a3 * sn.y, r3 = a3 * cs.y + a2 * sn.y;
            a0 = r0; a1 = r1; a2 = r2; a3 = r3;
        }
        const float sc = which ? 1.0f : q_scale;
        store4(p, a0 * sc, a1 * sc, a2 * sc, a3 * sc);
    }
}

// --------------------------------------------------------------------------------------- weight repacking
// Conv1d weight [R = out channels][A = in channels per group][31 taps] (torch) -> implicit-GEMM operand
// out[r][tap * A + a], rows zero-padded to ld_out (a whole number of 128-byte K-tiles, convpos.h)
template <typename T>
__global__ void conv_pack_kernel(const float* __restrict__ in, T* __restrict__ out, long R, int A, int Bd, int ld_out) {
    const long total = R * ld_out;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % ld_out);
        const long r = i / ld_out;
        const int a = k % A, bd = k / A;
        out[i] = bd < Bd ? from_f32<T>(in[(r * A + a) * Bd + bd]) : from_f32<T>(0.f);
    }
}

// in [R][A][Bd] fp32 -> out [R][Bd][A] (T)
template <typename T>
__global__ void permute_last2_kernel(const float* __restrict__ in, T* __restrict__ out, long R, int A, int Bd) {
    const long total = R * A * Bd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const long t = i / A;
        const int bd = (int)(t % Bd);
        const long r = t / Bd;
        out[i] = from_f32<T>(in[(r * A + a) * Bd + bd]);
    }
}
// 2-D copy with conversion and zero padding: out[r][c] = c < cols ? in[r][c] : 0, r < rows_out (rows >= rows_in zero)
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ in, int ld_in, int rows_in, int cols, T* __restrict__ out,
                                int ld_out, int rows_out) {
    const long total = (long)rows_out * ld_out;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld_out);
        const long r = i / ld_out;
        out[i] = from_f32<T>((r < rows_in && c < cols) ? in[r * ld_in + c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ Vocos
// A[b*T + t][k*C + ci] = mel[b][ci][t + k - 3]  (Conv1d(C, dim, k=7, pad=3) as a GEMM); columns >= 7*C zero (K padding).
// mel is read through element strides (sb, sc, st), so the [B, T, C] mel that sample() returns can be decoded through
// the reference's `vocoder.decode(mel.permute(0, 2, 1))` view without a transposing copy.
static __global__ void im2col7_kernel(const float* __restrict__ mel, long sb, long sc, long st, float* __restrict__ A, int B, int C,
                                      int T, int ld) {
    const long total = (long)B * T * ld;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % ld);
        const long r = i / ld;
        const int t = (int)(r % T), b = (int)(r / T);
        float v = 0.f;
        if (col < 7 * C) {
            const int k = col / C, ci = col - k * C;
            const int tt = t + k - 3;
            if (tt >= 0 && tt < T) v = mel[b * sb + ci * sc + tt * st];
        }
        A[i] = v;
    }
}
// ISTFT head (export_vocoder_to_onnx.py:51-59): h[r, 0:F] = log-magnitude, h[r, F:2F] = phase;
// S[r, f] = min(exp(m), 100) * cos(p), S[r, F + f] = ... * sin(p); columns >= 2F zeroed (K padding of the iDFT GEMM).
static __global__ void istft_spec_kernel(const float* __restrict__ h, int ldh, float* __restrict__ S, int lds, long R, int F) {
    const long total = R * lds;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % lds);
        const long r = i / lds;
        float v = 0.f;
        if (c < 2 * F) {
            const int f = c < F ? c : c - F;
            const float mag = fminf(expf(h[r * ldh + f]), 100.0f);
            const float p = h[r * ldh + F + f];
            v = c < F ? mag * cosf(p) : mag * sinf(p);
        }
        S[i] = v;
    }
}
// overlap-add of windowed frames + window-envelope normalisation + centre trim = torch.istft(center=True)
// frames [B*T, nfft] (already multiplied by the synthesis window), wav [B, (T-1)*hop]
static __global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win, float* __restrict__ wav,
                                 int B, int T, int nfft, int hop) {
    const int L = (T - 1) * hop;
    const long total = (long)B * L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / L);
        const int pos = (int)(i % L) + nfft / 2;
        int t1 = pos / hop;
        if (t1 > T - 1) t1 = T - 1;
        int t0 = (pos - nfft + hop) / hop;  // smallest t with t*hop + nfft > pos
        if (pos - nfft + 1 <= 0) t0 = 0;
        float acc = 0.f, env = 0.f;
        for (int t = t0; t <= t1; ++t) {
            const int j = pos - t * hop;
            if (j >= 0 && j < nfft) {
                acc += frames[((size_t)b * T + t) * nfft + j];
                env += win[j] * win[j];
            }
        }
        wav[i] = acc / env;
    }
}

// In place: W f32 [rows, ld] (ld % 32 == 0) -> per 32-element block the 128 bytes the split-operand GEMM reads (gemm2.h MODE 3):
// 16-byte chunk g = f16 hi of k = 4g..4g+3, 16+4g..16+4g+3; chunk 4 + g = f16 lo (w - hi) of the same k.  One thread per block.
static __global__ void split_planar_kernel(float* __restrict__ w, long blocks) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < blocks; i += (long)gridDim.x * blockDim.x) {
        float4* p = reinterpret_cast<float4*>(w + i * 32);
        float v[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 t = p[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        f16_t hi[32], lo[32];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float x = v[(s < 4 ? 4 * g : 16 + 4 * g - 4) + s];
                const f16_t h = (f16_t)x;
                hi[g * 8 + s] = h;
                lo[g * 8 + s] = (f16_t)(x - (float)h);
            }
        f16x8* o = reinterpret_cast<f16x8*>(w + i * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x8 a, b;
#pragma unroll
            for (int s = 0; s < 8; ++s) { a[s] = hi[g * 8 + s]; b[s] = lo[g * 8 + s]; }
            o[g] = a;
            o[4 + g] = b;
        }
    }
}

// Diagnostic (F5_X3_ABLATE): out = (float)(f16)in -- an f32 operand that is not stored split, as plain f16 would see it
static __global__ void round_f16_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)(f16_t)in[i];
}
// The rotary table in the order the QKV epilogue's accumulator lanes read it (EpiQKV, gemm.h): out[pos][g][j] = {cos, cos, sin, sin} of
// the pairs (j*8 + g*2, j*8 + g*2 + 1) from the [maxpos][32] tables (modules.py:398-424 / x_transformers RotaryEmbedding values).
static __global__ __launch_bounds__(256) void rope_frag_kernel(const float* __restrict__ cs, const float* __restrict__ sn, float* __restrict__ out,
                                                        long maxpos) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < maxpos * 16; i += (long)gridDim.x * blockDim.x) {   // one (pos, g, j)
        const long pos = i >> 4;
        const int g = (int)(i >> 2) & 3, j = (int)i & 3, pair = j * 8 + g * 2;
        const float2 c = *reinterpret_cast<const float2*>(cs + pos * 32 + pair), s2 = *reinterpret_cast<const float2*>(sn + pos * 32 + pair);
        *reinterpret_cast<float4*>(out + i * 4) = make_float4(c.x, c.y, s2.x, s2.y);
    }
}

// Diagnostic (F5_X3_ABLATE, tools/x3_ablate.py): zeroes the lo halves (chunks 4..7 of every 128-byte block) of a split-planar
// operand in place, which turns the three-product f16x3 contraction into the plain f16 one (a_hi w_hi) for that operand.
static __global__ void zero_lo_planar_kernel(float* __restrict__ w, long blocks) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < blocks * 4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(w + (i >> 2) * 32)[4 + (i & 3)] = make_float4(0.f, 0.f, 0.f, 0.f);
}

inline int ew_blocks(long total, int per_block = 256, int cap = 4096) {
    long b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace f5
