/* f5_hip.h -- C ABI of libf5hip.so, the MI355X (gfx950) engine for the F5-TTS inference hot path.
 *
 * Plain C: opaque handles, raw device pointers, sizes and a hipStream_t passed as void*.  No torch / C++ types.
 * Every entry point returns 0 on success or a negative F5_E* code; f5_last_error() returns the message of the last
 * failure on the calling thread.  All work is enqueued on the caller's stream; the library never synchronises the
 * device except where a function's comment says so.  Device pointers are borrowed for the duration of the call
 * (weights are copied / repacked into engine-owned memory by f5_load_weight + f5_finalize).
 *
 * Each function names the reference interface it replaces (paths relative to the reference's src/f5_tts/):
 *   f5_sample        <- CFM.sample's ODE solve                       model/cfm.py:151-223 (odeint call :218)
 *   f5_dit_forward   <- DiT.forward / UNetT.forward                  model/backbones/dit.py:278-329, unett.py:217-280
 *   f5_text_embed    <- TextEmbedding.forward (+ per-sample loop)    model/backbones/dit.py:86-115,244-258
 *   f5_vocos_decode  <- vocoder.decode(mel)                          infer/utils_infer.py:702-703 (third-party vocos)
 *   f5_bigvgan_forward <- vocoder(mel) (third-party BigVGAN v2)      infer/utils_infer.py:138-152,705
 *   f5_mel_forward   <- MelSpec.forward (vocos / bigvgan type)       model/modules.py:33-146
 *   f5_load_weight   <- load_checkpoint's state-dict assignment      infer/utils_infer.py:242-286
 * The reference-side binding (ctypes) is shown in INTEGRATION.md.
 */
#ifndef F5_HIP_H
#define F5_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F5_OK 0
#define F5_EINVAL (-1)   /* bad argument / shape / unsupported configuration */
#define F5_EHIP (-2)     /* a HIP runtime call failed */
#define F5_ESTATE (-3)   /* call order violated (e.g. compute before f5_finalize) */
#define F5_ENOMEM (-4)

#define F5_PREC_F32 0    /* exact-f32 MFMA everywhere ("parity" mode) */
#define F5_PREC_BF16 1   /* bf16 MFMA operands, f32 accumulate, f32 residual stream / ODE state / norms */
#define F5_PREC_F16 2    /* fp16 MFMA operands (the reference's own GPU dtype, infer/utils_infer.py:243-251; same MFMA rate as
                            bf16, 3 more mantissa bits), f32 accumulate, f32 residual stream / ODE state / norms */
#define F5_PREC_F16X3 3  /* f32 data flow (activations, attention, norms as F5_PREC_F32); each GEMM operand is split into two fp16
                            halves (hi + lo, 22 bits) and the product takes three fp16 MFMAs: f32-level results; the backbone GEMMs run
                            at ~2.5-3x the f32 MFMA rate, a C2 utterance in 0.41x the f32 time.  The two attention products
                            use the hi halves only (plain fp16 products: measured harmless; F5_X3_ATTN_SPLIT=1 for three).
                            |activation| < 65504 as for F5_PREC_F16 */

#define F5_PREC_F16P 4   /* F5_PREC_F16 in the transformer blocks (fp16 MFMA operands, f32 accumulate / residual / norms) with the
                            model's input and output layers -- input projection, conv position embedding, final norm + output
                            projection -- as split-fp16 products on f32 operands (as F5_PREC_F16X3): the rounding of the ODE state
                            entering and of the flow prediction leaving the backbone is what dominates F5_PREC_F16's error (DESIGN.md
                            section 3, tools/x3_ablate.py); meets the 1e-3 parity bar at ~F5_PREC_F16 speed */

#define F5_OPT_QK_RMSNORM 1        /* qk_norm="rms_norm" (modules.py:397-404,481-484): weights ...attn.q_norm.weight / k_norm.weight [64] */
#define F5_OPT_LONG_SKIP 2         /* long_skip_connection=True (dit.py:205,313-324): long_skip_connection.weight [D, 2D] */
#define F5_OPT_TEXT_AVG_UPSAMPLE 4 /* text_embedding_average_upsampling=True (dit.py:54-84; needs text_mask_padding) */

#define F5_BACKBONE_DIT 0
#define F5_BACKBONE_UNETT 1

typedef struct f5_engine f5_engine;
typedef struct f5_vocos f5_vocos;
typedef void* f5_stream; /* hipStream_t */

/* Arch of the backbone: the keyword arguments of DiT(...) / UNetT(...) (dit.py:147-168, unett.py:107-128). */
typedef struct f5_config {
    int32_t backbone;          /* F5_BACKBONE_* */
    int32_t precision;         /* F5_PREC_* */
    int32_t dim;               /* model width D: 256, 512, 768 or 1024 */
    int32_t depth;
    int32_t heads;
    int32_t dim_head;          /* must be 64 */
    int32_t ff_dim;            /* int(dim * ff_mult) */
    int32_t text_dim;
    int32_t conv_layers;       /* ConvNeXt-V2 text blocks */
    int32_t pe_attn_head;      /* heads that receive rotary; <0 = all (pe_attn_head=None) */
    int32_t text_mask_padding; /* dit.py:39,90-91,104-108 */
    int32_t attn_mask_enabled; /* modules.py:501-506 */
    int32_t text_num_embeds;   /* constructor argument; the table has text_num_embeds + 1 rows */
    int32_t mel_dim;           /* 100 */
    int32_t max_pos;           /* rows of the rotary table given as aux.rope_cos/sin (>= longest sequence) */
    int32_t options;           /* F5_OPT_* bits: the DiT constructor options no shipped config switches on */
    int32_t reserved[4];
} f5_config;

const char* f5_last_error(void);
/* Build id string ("f5hip <n> gfx950"). */
const char* f5_version(void);

int f5_create(const f5_config* cfg, f5_engine** out);
int f5_destroy(f5_engine* e);

/* Copies one tensor (fp32, contiguous, device memory) into the engine under its reference state-dict name, e.g.
 * "transformer_blocks.3.attn.to_q.weight" (names: model/backbones/dit.py, unett.py; convert_checkpoint.py:129-145).
 * Host-computed constant tables travel the same way under reserved names:
 *   aux.rope_cos / aux.rope_sin [max_pos, 32]   rotary angles n * 10000^(-2j/64)          (x_transformers RotaryEmbedding)
 *   aux.time_freqs [128]                         exp(-k ln(1e4)/127)                       (modules.py:159-161)
 *   aux.text_pos [P, text_dim]                   precompute_freqs_cis(text_dim, P)         (modules.py:202-213)   */
int f5_load_weight(f5_engine* e, const char* name, const void* dev_f32, const int64_t* shape, int32_t ndim,
                   f5_stream stream);
/* Checks that every tensor the arch needs is present and repacks (fused QKV, stacked AdaLN, conv tap-major, bf16).
 * Synchronises the stream once. */
int f5_finalize(f5_engine* e, f5_stream stream);

/* text i64[B, nt] (device, padded with -1) -> out f32[B, N, text_dim] (device).  lens (HOST int32[B] or NULL): each
 * sample is embedded at its own length and zero padded to N, as dit.py:247-258 does when an audio mask is given. */
int f5_text_embed(f5_engine* e, const int64_t* text, int32_t B, int32_t nt, const int32_t* lens_host, int32_t N,
                  int32_t drop_text, float* out, f5_stream stream);

/* One backbone forward (dit.py:278-329).  x, cond f32[B, N, mel]; text i64[B, nt]; time HOST f32[B];
 * lens HOST int32[B] or NULL (= the reference's mask=None).  cfg_infer != 0 packs cond + uncond: out f32[2B, N, mel],
 * else out f32[B, N, mel] with the given drop flags.  Text embeddings are recomputed on every call (cache=False). */
int f5_dit_forward(f5_engine* e, const float* x, const float* cond, const int64_t* text, int32_t nt,
                   const float* time_host, const int32_t* lens_host, int32_t B, int32_t N, int32_t cfg_infer,
                   int32_t drop_audio_cond, int32_t drop_text, float* out, f5_stream stream);

/* The ODE solve of CFM.sample (cfm.py:151-223): step_cond = where(cond_mask, cond, 0); `steps` Euler steps over the
 * HOST time grid t[steps + 1] with classifier-free guidance (cfg_strength < 1e-5 -> single conditional forward);
 * out = where(cond_mask, cond, y_final).
 *   cond f32[B, cond_frames, mel] (cond_frames <= N; the frames cond_frames .. N-1 read as zero: the reference's
 *   F.pad(cond, (0, 0, 0, N - cond_seq_len)), cfm.py:145; cond_frames = 0 is no_ref_audio), cond_mask u8[B, N], y0 f32[B, N, mel],
 *   text i64[B, nt], lens HOST int32[B] = per-sample durations or NULL when B == 1 (cfm.py:155-158),
 *   out f32[B, N, mel], traj f32[steps + 1, B, N, mel] or NULL.
 * The text embeddings are computed once per call (the reference's per-sample() cache, dit.py:244-269).
 * Frames past a sample's own length: with attn_mask_enabled and B > 1 the backbone runs on the valid rows only (RowPack), and those
 * frames keep y0's value in `out` / `traj` (the reference integrates them to values nobody reads, cfm.py:160-191); F5_PACK_ROWS=0
 * computes them as the reference does.  With attn_mask_enabled = 0 (every shipped config) all frames match the reference. */
int f5_sample(f5_engine* e, const float* cond, int32_t cond_frames, const uint8_t* cond_mask, const float* y0,
              const int64_t* text, int32_t nt, const float* t_host, int32_t steps, float cfg_strength,
              const int32_t* lens_host, int32_t B, int32_t N, float* out, float* traj, f5_stream stream);

/* Pre-sizes the activation arena (otherwise it grows on first use, which calls hipMalloc inside f5_sample). */
int f5_reserve(f5_engine* e, int32_t max_batch, int32_t max_frames, int32_t max_steps);

/* ------------------------------------------------------------------------------------------------ Vocos */
typedef struct f5_vocos_config {
    int32_t input_channels; /* 100 */
    int32_t dim;            /* 512 */
    int32_t intermediate_dim; /* 1536 */
    int32_t num_layers;     /* 8 */
    int32_t n_fft;          /* 1024 */
    int32_t hop_length;     /* 256 */
    int32_t reserved[4];
} f5_vocos_config;

int f5_vocos_create(const f5_vocos_config* cfg, f5_vocos** out);
int f5_vocos_destroy(f5_vocos* v);
/* names: backbone.embed.weight, backbone.convnext.N.{dwconv,norm,pwconv1,pwconv2}.{weight,bias}, ...gamma, head.out.*;
 * aux.hann [n_fft], aux.idft_basis [n_fft, 2*(n_fft/2+1) rounded up to a multiple of 32] (window folded in). */
int f5_vocos_load_weight(f5_vocos* v, const char* name, const void* dev_f32, const int64_t* shape, int32_t ndim,
                         f5_stream stream);
int f5_vocos_finalize(f5_vocos* v, f5_stream stream);
/* mel f32[B, C, T] -> wav f32[B, (T - 1) * hop]   (Vocos.decode: backbone -> ISTFTHead, padding="center") */
int f5_vocos_decode(f5_vocos* v, const float* mel, int32_t B, int32_t T, float* wav, f5_stream stream);
/* The same with mel addressed as mel[b * stride_b + c * stride_c + t * stride_t] (element strides): the callers'
 * `vocoder.decode(generated.permute(0, 2, 1))` (utils_infer.py:702-703) passes a transposed VIEW of sample()'s [B, T, C]
 * output, which is decoded in place instead of through a transposing copy. */
int f5_vocos_decode_strided(f5_vocos* v, const float* mel, int32_t B, int32_t T, int64_t stride_b, int64_t stride_c,
                            int64_t stride_t, float* wav, f5_stream stream);

/* ---------------------------------------------------------------------------------------------- BigVGAN
 * The vocoder of mel_spec_type="bigvgan" (infer/utils_infer.py:138-152: bigvgan.BigVGAN.from_pretrained(
 * "nvidia/bigvgan_v2_24khz_100band_256x"), remove_weight_norm(); called as vocoder(mel[B, 100, T]) -> wav[B, 1, 256 T]
 * at :705).  The reference takes it from an un-vendored submodule: the architecture is restated from the published
 * BigVGAN v2 (see csrc/bigvgan.hip; checker: tests/test_bigvgan.py) -- parity unpinned. */
typedef struct f5_bigvgan f5_bigvgan;
typedef struct f5_bigvgan_config {
    int32_t num_mels;                 /* 100 */
    int32_t upsample_initial_channel; /* 1536; halves at every upsample stage */
    int32_t num_upsamples;            /* 6 */
    int32_t upsample_rates[8];        /* 4, 4, 2, 2, 2, 2 */
    int32_t upsample_kernel_sizes[8]; /* 8, 8, 4, 4, 4, 4 */
    int32_t num_kernels;              /* 3 AMP blocks per stage ... */
    int32_t resblock_kernel_sizes[4]; /* ... with kernels 3, 7, 11 */
    int32_t num_dilations;            /* 3 (conv, conv) pairs per AMP block ... */
    int32_t resblock_dilations[4];    /* ... with dilations 1, 3, 5 on the first conv of each pair */
    int32_t use_tanh_at_final;        /* 0: clamp(-1, 1) */
    int32_t use_bias_at_final;        /* 0 */
    int32_t precision;                /* F5_PREC_F32 (0) or F5_PREC_F16X3: the wide stages' convolutions as split-f16 products */
    int32_t reserved[3];
} f5_bigvgan_config;
int f5_bigvgan_create(const f5_bigvgan_config* cfg, f5_bigvgan** out);
int f5_bigvgan_destroy(f5_bigvgan* v);
/* names (after remove_weight_norm): conv_pre.{weight,bias}, ups.N.0.{weight,bias}, resblocks.N.convs1.M.{weight,bias},
 * resblocks.N.convs2.M.{weight,bias}, resblocks.N.activations.A.act.{alpha,beta}, activation_post.act.{alpha,beta},
 * conv_post.weight (+ .bias); host-computed aux.up_filter / aux.down_filter [12] (kaiser-sinc filters of Activation1d). */
int f5_bigvgan_load_weight(f5_bigvgan* v, const char* name, const void* dev_f32, const int64_t* shape, int32_t ndim,
                           f5_stream stream);
int f5_bigvgan_finalize(f5_bigvgan* v, f5_stream stream);
/* mel addressed as mel[b * stride_b + c * stride_c + t * stride_t] (element strides) -> wav f32[B, T * prod(rates)] */
int f5_bigvgan_forward(f5_bigvgan* v, const float* mel, int32_t B, int32_t T, int64_t stride_b, int64_t stride_c,
                       int64_t stride_t, float* wav, f5_stream stream);

/* ------------------------------------------------------------------------------------- prompt mel front-end
 * MelSpec.forward, mel_spec_type="vocos" (model/modules.py:78-146): wav f32[B, nw] -> log-mel f32[B, T, n_mels],
 * T = nw / hop + 1.  Constant tables are host-computed and loaded once:
 *   aux.dft_basis f32[round_up(2*(n_fft/2+1), 4), n_fft]  rows w*cos(2 pi f j / n_fft) for f <= n_fft/2, then w*sin(..)
 *   aux.mel_fb    f32[n_mels, round_up(n_fft/2+1, 32)]    HTK mel filterbank, norm=None (torchaudio melscale_fbanks)   */
typedef struct f5_mel f5_mel;
int f5_mel_create(int32_t n_fft, int32_t hop_length, int32_t n_mels, f5_mel** out);
int f5_mel_destroy(f5_mel* m);
int f5_mel_load(f5_mel* m, const char* name, const void* dev_f32, const int64_t* shape, int32_t ndim, f5_stream stream);
int f5_mel_forward(f5_mel* m, const float* wav, int32_t B, int32_t nw, float* out, f5_stream stream);
/* General form (mel_spec_type="bigvgan", model/modules.py:33-75: pad = (n_fft - hop) / 2, center=False, mag_eps = 1e-9,
 * aux.mel_fb = librosa's slaney filterbank): reflect padding `pad` on both sides, T = (nw + 2 pad - n_fft) / hop + 1 frames,
 * |S| = sqrt(re^2 + im^2 + mag_eps).  f5_mel_forward is pad = n_fft / 2, mag_eps = 0. */
int f5_mel_forward_ex(f5_mel* m, const float* wav, int32_t B, int32_t nw, int32_t pad, float mag_eps, float* out,
                      f5_stream stream);

/* ----------------------------------------------------------------------------- kernel-level entry points
 * Used by tests/ (parity of each kernel against a torch fp32 restatement) and by the micro-benchmarks.  fp32 in/out;
 * the operands are converted to the requested MFMA precision internally.  They allocate scratch and synchronise. */
int f5k_gemm(int32_t prec, const float* A, const float* W, const float* bias, int32_t act, float* out, int32_t M,
             int32_t N, int32_t K, int32_t tile_m, int32_t tile_n, f5_stream stream);
/* q, k, v f32[Bp, H, N, 64] (q unscaled) -> out f32[Bp, N, H*64]; kv_lens HOST int32[Bp] or NULL */
int f5k_attention(int32_t prec, const float* q, const float* k, const float* v, const int32_t* kv_lens_host,
                  float* out, int32_t Bp, int32_t H, int32_t N, f5_stream stream);
/* x f32[Bp, N, D]; w f32[D, D/16, 31]; y = mish(conv(x) + bias) (+ res); lens HOST int32[Bp] or NULL */
int f5k_convpos(int32_t prec, const float* x, const float* w, const float* bias, const float* res,
                const int32_t* lens_host, float* y, int32_t Bp, int32_t N, int32_t D, f5_stream stream);
/* LayerNorm(no affine, eps) * (1 + scale[b]) + shift[b]; x f32[R, D], scale/shift f32[R / rows_per_batch, D] */
int f5k_layernorm_mod(const float* x, const float* scale, const float* shift, float* out, int32_t R, int32_t D,
                      int32_t rows_per_batch, float eps, f5_stream stream);
/* repeated-launch timing of one GEMM shape (for bench/roofline): returns average microseconds per launch */
int f5k_gemm_time(int32_t prec, int32_t M, int32_t N, int32_t K, int32_t tile_m, int32_t tile_n, int32_t iters,
                  float* avg_us, f5_stream stream);

/* Per-launch HIP-event timing of the last f5_sample call when enabled: one (start, stop) event pair per kernel launch
 * on the launch stream, summed by kernel class; flops = algorithmic FLOPs of the bracketed launches (MFMA classes).
 * Off by default.  classes: 0 gemm, 1 attention, 2 layernorm, 3 convpos, 4 pack/euler/select, 5 text encoder
 * (whole sub-graph), 6 time-MLP + AdaLN precompute (whole sub-graph). */
int f5_profile_enable(f5_engine* e, int32_t on);
int f5_profile_read(f5_engine* e, float* ms_by_class, int32_t* launches_by_class, double* flops_by_class,
                    int32_t nclass);

#ifdef __cplusplus
}
#endif
#endif /* F5_HIP_H */
