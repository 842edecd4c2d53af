// Per-precision body of the engine (template definitions; included by engine_bf16.hip / engine_f16.hip / engine_f32.hip,
// each of which instantiates EngineOps<T> for one operand type).
#pragma once
#include "engine_types.h"

// --------------------------------------------------------------------------------------------- finalize
static int need(const WeightStore& ws, const std::string& n, std::initializer_list<int64_t> shape, const Tensor** out) {
    const Tensor* t = ws.get(n);
    if (!t) return fail(F5_ESTATE, "missing weight '%s'", n.c_str());
    std::vector<int64_t> want(shape);
    if (t->shape != want) {
        std::string got, exp;
        for (auto s : t->shape) got += std::to_string(s) + ",";
        for (auto s : want) exp += std::to_string(s) + ",";
        return fail(F5_EINVAL, "weight '%s' has shape [%s], expected [%s]", n.c_str(), got.c_str(), exp.c_str());
    }
    *out = t;
    return F5_OK;
}

// F5_PREC_F16X3: the backbone's GEMM weights (BB) are stored split into f16 hi / lo halves (elementwise.h split_planar_kernel)
template <typename T, bool BB> static int maybe_split_weight(f5_engine* e, hipStream_t s, T* w, size_t elems) {
    if constexpr (BB && std::is_same_v<T, float>) {
        if (e->split16 || e->io_split) {
            hipLaunchKernelGGL(split_planar_kernel, dim3(ew_blocks((long)(elems / 32))), dim3(256), 0, s, w, (long)(elems / 32));
            HIPCHK(hipGetLastError());
        }
    }
    return F5_OK;
}
// the conv position embedding runs split too where its K blocks stay inside one tap (32 | channels per group: dim 512, 1024)
static inline bool conv_split(const f5_engine* e) { return e->split16 && convpos_can_split(e->cfg.dim); }
// every backbone GEMM goes through here (the time / text paths call launch_gemm<float> directly: always plain f32)
template <typename T, typename Epi>
static hipError_t egemm(const f5_engine* e, hipStream_t s, const T* A, int lda, const T* W, int ldw, int M, int N, int K, const Epi& epi,
                        int force_cfg = -1, const int* m_limit = nullptr, int m_hint = 0, bool a_planar = false) {
    // a_planar: the A operand was written pre-split by its producer (store4_planar: LayerNorm, attention, the GELU epilogue)
    const int split = (std::is_same_v<T, float> && e->split16) ? (a_planar ? 2 : 1) : 0;
    return launch_gemm<T>(s, A, lda, W, ldw, M, N, K, epi, force_cfg, m_limit, m_hint, GemmConv{}, split);
}

// W [N, K] f32 -> T [N, round_up(K, 8)]
template <typename T, bool BB = false>
static int pack_linear(f5_engine* e, hipStream_t s, const std::string& wname, const std::string& bname, int N, int K,
                       LinW<T>* L, int n_pad = 0) {
    const Tensor *w = nullptr, *b = nullptr;
    CHK(need(e->ws, wname, {N, K}, &w));
    const int Np = n_pad ? n_pad : N;
    L->N = Np;
    L->K = K;
    L->ldw = round_up(K, 64);
    CHK(dev_alloc(e, &L->w, (size_t)Np * L->ldw));
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)Np * L->ldw)), dim3(256), 0, s, w->p, K, N, K, L->w,
                       L->ldw, Np);
    L->b = nullptr;
    if (!bname.empty()) {
        CHK(need(e->ws, bname, {N}, &b));
        CHK(dev_alloc(e, &L->b, (size_t)Np));
        hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(ew_blocks(Np)), dim3(256), 0, s, b->p, N, 1, N, L->b, Np, 1);
    }
    HIPCHK(hipGetLastError());
    return maybe_split_weight<T, BB>(e, s, L->w, (size_t)Np * L->ldw);
}

// concatenates several [Ni, K] linears row-wise into one [sum Ni, ldw] operand (+ bias)
template <typename T, bool BB = false>
static int pack_concat(f5_engine* e, hipStream_t s, const std::vector<std::string>& pfx, int Ni, int K, LinW<T>* L,
                       bool bias = true) {
    const int n = (int)pfx.size();
    L->N = n * Ni;
    L->K = K;
    L->ldw = round_up(K, 64);
    CHK(dev_alloc(e, &L->w, (size_t)L->N * L->ldw));
    L->b = nullptr;
    if (bias) CHK(dev_alloc(e, &L->b, (size_t)L->N));
    for (int i = 0; i < n; ++i) {
        const Tensor *w = nullptr, *b = nullptr;
        CHK(need(e->ws, pfx[i] + ".weight", {Ni, K}, &w));
        hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(ew_blocks((long)Ni * L->ldw)), dim3(256), 0, s, w->p, K, Ni, K,
                           L->w + (size_t)i * Ni * L->ldw, L->ldw, Ni);
        if (bias) {
            CHK(need(e->ws, pfx[i] + ".bias", {Ni}, &b));
            HIPCHK(hipMemcpyAsync(L->b + (size_t)i * Ni, b->p, (size_t)Ni * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    HIPCHK(hipGetLastError());
    return maybe_split_weight<T, BB>(e, s, L->w, (size_t)L->N * L->ldw);
}

// the backbone's own GEMM weights (split in F5_PREC_F16X3)
template <typename T, typename... Args> static int pack_linear_bb(Args&&... args) { return pack_linear<T, true>(std::forward<Args>(args)...); }
template <typename T, typename... Args> static int pack_concat_bb(Args&&... args) { return pack_concat<T, true>(std::forward<Args>(args)...); }

static int copy_vec(f5_engine* e, hipStream_t s, const std::string& name, std::initializer_list<int64_t> shape,
                    float** out) {
    const Tensor* t = nullptr;
    CHK(need(e->ws, name, shape, &t));
    CHK(dev_alloc(e, out, t->numel()));
    HIPCHK(hipMemcpyAsync(*out, t->p, t->numel() * sizeof(float), hipMemcpyDeviceToDevice, s));
    return F5_OK;
}

template <typename T> static int finalize_t(f5_engine* e, Packed<T>& P, hipStream_t s) {
    const f5_config& c = e->cfg;
    const int D = c.dim, Dt = c.text_dim, F = c.ff_dim, inner = e->inner, mel = c.mel_dim;
    const bool dit = c.backbone == F5_BACKBONE_DIT;
    // aux tables
    const Tensor* t = nullptr;
    if (!(t = e->ws.get("aux.rope_cos")) || t->shape.size() != 2 || t->shape[1] != 32 || t->shape[0] < 1)
        return fail(F5_ESTATE, "missing/invalid aux.rope_cos [max_pos, 32]");
    const int64_t maxpos = t->shape[0];
    e->cfg.max_pos = (int)maxpos;
    CHK(copy_vec(e, s, "aux.rope_cos", {maxpos, 32}, &P.rope_cos));
    CHK(copy_vec(e, s, "aux.rope_sin", {maxpos, 32}, &P.rope_sin));
    CHK(dev_alloc(e, &P.rope_frag, (size_t)maxpos * 64));
    hipLaunchKernelGGL(rope_frag_kernel, dim3(ew_blocks(maxpos * 16)), dim3(256), 0, s, P.rope_cos, P.rope_sin, P.rope_frag, (long)maxpos);
    KCHK();
    CHK(copy_vec(e, s, "aux.time_freqs", {128}, &P.time_freqs));
    // time MLP
    CHK(pack_linear<float>(e, s, "time_embed.time_mlp.0.weight", "time_embed.time_mlp.0.bias", D, 256, &P.time0));
    CHK(pack_linear<float>(e, s, "time_embed.time_mlp.2.weight", "time_embed.time_mlp.2.bias", D, D, &P.time2));
    // text encoder
    CHK(copy_vec(e, s, "text_embed.text_embed.weight", {c.text_num_embeds + 1, Dt}, &P.E));
    if (c.conv_layers > 0) {
        if (!(t = e->ws.get("aux.text_pos")) || t->shape.size() != 2 || t->shape[1] != Dt)
            return fail(F5_ESTATE, "missing/invalid aux.text_pos [P, text_dim]");
        P.text_pos_rows = (int)t->shape[0];
        CHK(copy_vec(e, s, "aux.text_pos", {t->shape[0], Dt}, &P.text_pos));
    }
    P.tblocks.resize(c.conv_layers);
    for (int i = 0; i < c.conv_layers; ++i) {
        const std::string p = "text_embed.text_blocks." + std::to_string(i);
        TextBlockW& tb = P.tblocks[i];
        const Tensor* dw = nullptr;
        CHK(need(e->ws, p + ".dwconv.weight", {Dt, 1, 7}, &dw));
        CHK(dev_alloc(e, &tb.dwk, (size_t)7 * Dt));
        hipLaunchKernelGGL((permute_last2_kernel<float>), dim3(ew_blocks(7L * Dt)), dim3(256), 0, s, dw->p, tb.dwk, 1L, Dt, 7);
        CHK(copy_vec(e, s, p + ".dwconv.bias", {Dt}, &tb.dwb));
        CHK(copy_vec(e, s, p + ".norm.weight", {Dt}, &tb.lnw));
        CHK(copy_vec(e, s, p + ".norm.bias", {Dt}, &tb.lnb));
        CHK(copy_vec(e, s, p + ".grn.gamma", {1, 1, 2 * Dt}, &tb.gamma));
        CHK(copy_vec(e, s, p + ".grn.beta", {1, 1, 2 * Dt}, &tb.beta));
        CHK(pack_linear<float>(e, s, p + ".pwconv1.weight", p + ".pwconv1.bias", 2 * Dt, Dt, &tb.pw1));
        CHK(pack_linear<float>(e, s, p + ".pwconv2.weight", p + ".pwconv2.bias", Dt, 2 * Dt, &tb.pw2));
    }
    // input embedding
    CHK(pack_linear_bb<T>(e, s, "input_embed.proj.weight", "input_embed.proj.bias", D, e->kin, &P.in_proj));
    auto zlo_w = [&](T* wp, size_t elems) {   // diagnostic F5_X3_ABLATE: plain f16 weights (lo halves of the split layout zeroed)
        if constexpr (std::is_same_v<T, float>)
            hipLaunchKernelGGL(zero_lo_planar_kernel, dim3(ew_blocks((long)elems / 8)), dim3(256), 0, s, wp, (long)(elems / 32));
    };
    if (e->x3_ablate & 64) zlo_w(P.in_proj.w, (size_t)P.in_proj.N * P.in_proj.ldw);
    const int cpg = D / 16;
    P.conv_kp = round_up(31 * cpg, GEMM_ROW_BYTES / (int)sizeof(T));   // whole K-tiles, zero padded (convpos.h)
    for (int j = 0; j < 2; ++j) {
        const std::string p = "input_embed.conv_pos_embed.conv1d." + std::to_string(j * 2);
        const Tensor* w = nullptr;
        CHK(need(e->ws, p + ".weight", {D, cpg, 31}, &w));
        CHK(dev_alloc(e, &P.conv_w[j], (size_t)D * P.conv_kp));
        hipLaunchKernelGGL((conv_pack_kernel<T>), dim3(ew_blocks((long)D * P.conv_kp)), dim3(256), 0, s, w->p,
                           P.conv_w[j], (long)D, cpg, 31, P.conv_kp);
        if (conv_split(e)) CHK((maybe_split_weight<T, true>(e, s, P.conv_w[j], (size_t)D * P.conv_kp)));   // F5_PREC_F16X3 (convpos.h SPLIT)
        if (conv_split(e) && (e->x3_ablate & 128)) zlo_w(P.conv_w[j], (size_t)D * P.conv_kp);
        CHK(copy_vec(e, s, p + ".bias", {D}, &P.conv_b[j]));
    }
    // transformer
    P.blocks.resize(c.depth);
    std::vector<std::string> modp;
    for (int i = 0; i < c.depth; ++i) {
        BlockW<T>& b = P.blocks[i];
        const std::string p = dit ? "transformer_blocks." + std::to_string(i) : "layers." + std::to_string(i);
        const std::string at = dit ? p + ".attn" : p + ".2";
        const std::string ff = dit ? p + ".ff" : p + ".4";
        CHK(pack_concat_bb<T>(e, s, std::vector<std::string>{at + ".to_q", at + ".to_k", at + ".to_v"}, inner, D, &b.qkv));
        CHK(pack_linear_bb<T>(e, s, at + ".to_out.0.weight", at + ".to_out.0.bias", D, inner, &b.out));
        CHK(pack_linear_bb<T>(e, s, ff + ".ff.0.0.weight", ff + ".ff.0.0.bias", F, D, &b.ff1));
        CHK(pack_linear_bb<T>(e, s, ff + ".ff.2.weight", ff + ".ff.2.bias", D, F, &b.ff2));
        if constexpr (std::is_same_v<T, float>) {   // diagnostic F5_X3_ABLATE: plain f16 weights (lo halves zeroed) for a class
            auto zlo = [&](LinW<T>& L) {
                const long blocks = (long)L.N * L.ldw / 32;
                hipLaunchKernelGGL(zero_lo_planar_kernel, dim3(ew_blocks(blocks * 4)), dim3(256), 0, s, L.w, blocks);
            };
            if (e->x3_ablate & 1) zlo(b.qkv);
            if (e->x3_ablate & 8) zlo(b.out);
            if (e->x3_ablate & 16) zlo(b.ff1);
            if (e->x3_ablate & 32) zlo(b.ff2);
            HIPCHK(hipGetLastError());
        }
        if (c.options & F5_OPT_QK_RMSNORM) {
            CHK(copy_vec(e, s, at + ".q_norm.weight", {64}, &b.qn));
            CHK(copy_vec(e, s, at + ".k_norm.weight", {64}, &b.kn));
        }
        if (dit) {
            modp.push_back(p + ".attn_norm.linear");
        } else {
            CHK(copy_vec(e, s, p + ".1.g", {D}, &b.norm1_g));
            CHK(copy_vec(e, s, p + ".3.g", {D}, &b.norm2_g));
            if (i >= c.depth / 2 && e->ws.get(p + ".0.weight")) {
                CHK(pack_linear_bb<T>(e, s, p + ".0.weight", "", D, 2 * D, &b.skip));
                if constexpr (std::is_same_v<T, float>) {
                    if (e->x3_ablate & 512) {   // diagnostic: the skip projection with plain f16 weights
                        const long blocks = (long)b.skip.N * b.skip.ldw / 32;
                        hipLaunchKernelGGL(zero_lo_planar_kernel, dim3(ew_blocks(blocks * 4)), dim3(256), 0, s, b.skip.w, blocks);
                        HIPCHK(hipGetLastError());
                    }
                }
            }
        }
    }
    if (dit) {
        // stacked AdaLN: rows [l*6D, (l+1)*6D) = layer l (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp);
        // last 2D rows = norm_out (scale, shift).  Always f32.
        LinW<float>& M = P.mod;
        M.N = e->modN;
        M.K = D;
        M.ldw = D;
        CHK(dev_alloc(e, &M.w, (size_t)M.N * D));
        CHK(dev_alloc(e, &M.b, (size_t)M.N));
        for (int i = 0; i <= c.depth; ++i) {
            const bool last = i == c.depth;
            const std::string p = last ? std::string("norm_out.linear") : modp[i];
            const int rows = last ? 2 * D : 6 * D;
            const Tensor *w = nullptr, *b = nullptr;
            CHK(need(e->ws, p + ".weight", {rows, D}, &w));
            CHK(need(e->ws, p + ".bias", {rows}, &b));
            HIPCHK(hipMemcpyAsync(M.w + (size_t)i * 6 * D * D, w->p, (size_t)rows * D * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
            HIPCHK(hipMemcpyAsync(M.b + (size_t)i * 6 * D, b->p, (size_t)rows * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    } else {
        CHK(copy_vec(e, s, "norm_out.g", {D}, &P.norm_out_g));
    }
    CHK(pack_linear_bb<T>(e, s, "proj_out.weight", "proj_out.bias", mel, D, &P.proj_out));
    if (c.options & F5_OPT_LONG_SKIP) {
        CHK(pack_linear_bb<T>(e, s, "long_skip_connection.weight", "", D, 2 * D, &P.long_skip));
        if (e->io_split) CHK((pack_linear<float, true>(e, s, "long_skip_connection.weight", "", D, 2 * D, &P.long_skip_f)));
    }
    if (e->io_split) {   // F5_PREC_F16P: the input / output layers once more, as split-planar f32 operands
        CHK((pack_linear<float, true>(e, s, "input_embed.proj.weight", "input_embed.proj.bias", D, e->kin, &P.in_proj_f)));
        CHK((pack_linear<float, true>(e, s, "proj_out.weight", "proj_out.bias", mel, D, &P.proj_out_f)));
        P.conv_kp_f = round_up(31 * cpg, GEMM_ROW_BYTES / (int)sizeof(float));
        for (int j = 0; j < 2; ++j) {
            const std::string p = "input_embed.conv_pos_embed.conv1d." + std::to_string(j * 2);
            const Tensor* w = nullptr;
            CHK(need(e->ws, p + ".weight", {D, cpg, 31}, &w));
            CHK(dev_alloc(e, &P.conv_w_f[j], (size_t)D * P.conv_kp_f));
            hipLaunchKernelGGL((conv_pack_kernel<float>), dim3(ew_blocks((long)D * P.conv_kp_f)), dim3(256), 0, s, w->p,
                               P.conv_w_f[j], (long)D, cpg, 31, P.conv_kp_f);
            // (dims whose 32-deep K blocks straddle taps keep the exact-f32 MFMA kernel: convpos_can_split)
            if (convpos_can_split(D)) CHK((maybe_split_weight<float, true>(e, s, P.conv_w_f[j], (size_t)D * P.conv_kp_f)));
        }
    }
    if (e->x3_ablate & 256) zlo_w(P.proj_out.w, (size_t)P.proj_out.N * P.proj_out.ldw);
    HIPCHK(hipGetLastError());
    return F5_OK;
}
template <typename T> static size_t carve_into(const f5_engine* e, Arena& a, Work<T>& w, int B, int N, int S) {
    const size_t NT = (size_t)std::max(e->res_nt, 1);
    const f5_config& c = e->cfg;
    a.reset();
    const size_t Bp = 2 * (size_t)B, D = c.dim, Dt = c.text_dim, F = c.ff_dim, mel = c.mel_dim;
    const size_t Nt = c.backbone == F5_BACKBONE_UNETT ? N + 1 : N;  // UNetT prepends the time token
    const size_t rows = Bp * Nt;
    const size_t SS = (size_t)std::max(S + 1, (int)Bp + 1);
    w.Npad = round_up((int)Nt, 64);
    w.tdev = a.take<float>(SS);
    w.feat = a.take<float>(SS * 256);
    w.th = a.take<float>(SS * D);
    w.temb = a.take<float>(SS * D);
    w.st = a.take<float>(SS * D);
    w.mod = a.take<float>(SS * e->modN);
    w.lens = a.take<int>(Bp + 16);
    w.lens_plain = a.take<int>(Bp + 16);
    w.row_start = a.take<int>(Bp + (size_t)B + 16);
    w.rowmap = a.take<int2>(Bp * (size_t)round_up(N, 4));
    w.step_cond = a.take<float>((size_t)B * N * mel);
    w.text_c = a.take<float>((size_t)B * N * Dt);
    w.text_u = a.take<float>((size_t)B * N * Dt);
    w.tx_a = a.take<float>((size_t)B * N * Dt);
    w.tx_b = a.take<float>((size_t)B * N * Dt);
    w.tx_h1 = a.take<float>((size_t)B * N * 2 * Dt);
    w.grn_part = a.take<float>((size_t)B * GRN_P * 2 * Dt);
    w.uc = a.take<float>((size_t)N * Dt);
    w.acat = a.take<T>(Bp * N * e->kin_pad * (e->io_split ? sizeof(float) / sizeof(T) : 1));   // (F5_PREC_F16P packs the input rows as f32)
    w.h = a.take<float>(rows * D);
    w.c1 = a.take<float>(rows * D);
    w.x = a.take<float>(rows * D);
    w.pred = a.take<float>(Bp * N * mel);
    w.xn = a.take<T>(rows * D);
    w.q = a.take<T>(rows * e->inner);
    w.k = a.take<T>(rows * e->inner);
    w.ao = a.take<T>(rows * e->inner);
    w.vt = a.take<T>(Bp * c.heads * 64 * w.Npad);
    w.ffh = a.take<T>(rows * F);
    w.in_cond = a.take<float>((size_t)B * N * mel);
    w.y = a.take<float>((size_t)B * N * mel);
    w.out_buf = a.take<float>((size_t)B * N * mel);
    w.traj_buf = a.take<float>((size_t)(S + 1) * B * N * mel);
    w.in_mask = a.take<unsigned char>((size_t)B * N + 16);
    w.in_text = a.take<long long>((size_t)B * NT + 2);
    w.cat2 = nullptr;
    w.skips = nullptr;
    w.pred_all = nullptr;
    if (c.options & F5_OPT_LONG_SKIP) w.cat2 = a.take<T>(rows * 2 * D * (e->io_split ? sizeof(float) / sizeof(T) : 1));
    if (c.backbone == F5_BACKBONE_UNETT) {
        w.cat2 = a.take<T>(rows * 2 * D);
        w.skips = a.take<float>(rows * D * (c.depth / 2));
        w.pred_all = a.take<float>(rows * mel);
    }
    return align_up(a.off, 256) + 256;
}
template <typename T> static void carve(f5_engine* e, Work<T>& w, int B, int N, int S) {
    (void)carve_into<T>(e, e->arena, w, B, N, S);
}

// ---------------------------------------------------------------------------------------------- sub-graphs

template <typename T> static Packed<T>& packed(f5_engine* e);
template <> inline Packed<float>& packed<float>(f5_engine* e) { return e->pf; }
template <> inline Packed<bf16_t>& packed<bf16_t>(f5_engine* e) { return e->pb; }
template <> inline Packed<f16_t>& packed<f16_t>(f5_engine* e) { return e->ph; }

// time features -> t_emb [S, D], silu(t_emb), and (DiT) all AdaLN vectors mod [S, modN]
template <typename T> static int run_time_path(f5_engine* e, Work<T>& w, int S, hipStream_t s) {
    Packed<T>& P = packed<T>(e);
    const int D = e->cfg.dim;
    e->prof.begin(PC_TIME, s);
    hipLaunchKernelGGL(time_sinus_kernel, dim3((S * 128 + 255) / 256), dim3(256), 0, s, w.tdev, P.time_freqs, w.feat, S, 128);
    KCHK();
    HIPCHK(launch_gemm<float>(s, w.feat, 256, P.time0.w, P.time0.ldw, S, D, 256,
                              EpiStore<float>{w.th, D, P.time0.b, F5_ACT_SILU}));
    HIPCHK(launch_gemm<float>(s, w.th, D, P.time2.w, P.time2.ldw, S, D, D, EpiStore<float>{w.temb, D, P.time2.b, F5_ACT_NONE}));
    if (e->cfg.backbone == F5_BACKBONE_DIT) {
        hipLaunchKernelGGL(act_kernel, dim3(ew_blocks((long)S * D)), dim3(256), 0, s, w.temb, w.st, (long)S * D, F5_ACT_SILU);
        KCHK();
        HIPCHK(launch_gemm<float>(s, w.st, D, P.mod.w, P.mod.ldw, S, e->modN, D,
                                  EpiStore<float>{w.mod, e->modN, P.mod.b, F5_ACT_NONE}));
    }
    e->prof.end(s);
    return F5_OK;
}

// text ids -> text embedding [B, N, Dt] (dit.py:86-115 + per-sample lengths dit.py:247-258)
template <typename T>
static int run_text_embed(f5_engine* e, Work<T>& w, const int64_t* text, int B, int nt, const int* lens_dev, int N,
                          int drop_text, float* out, hipStream_t s) {
    Packed<T>& P = packed<T>(e);
    const f5_config& c = e->cfg;
    const int Dt = c.text_dim;
    const bool dit = c.backbone == F5_BACKBONE_DIT;
    const long rows = (long)B * N;
    e->prof.begin(PC_TEXT, s);
    if (c.conv_layers > 0 && dit && N > P.text_pos_rows) return fail(F5_EINVAL, "N=%d exceeds aux.text_pos rows", N);
    float* cur = c.conv_layers > 0 ? w.tx_a : out;  // block input/output (updated in place); last block writes `out`
    const int pos_rows = P.text_pos_rows > 0 ? P.text_pos_rows : 1;
    hipLaunchKernelGGL(text_embed_kernel, dim3(ew_blocks(rows * Dt / 4)), dim3(256), 0, s, (const long long*)text, nt, P.E,
                       P.text_pos, cur, B, N, Dt, lens_dev, drop_text, c.text_mask_padding && c.conv_layers > 0,
                       c.conv_layers > 0, dit ? 1 << 30 : pos_rows);
    KCHK();
    for (int i = 0; i < c.conv_layers; ++i) {
        TextBlockW& tb = P.tblocks[i];
        float* nxt = (i == c.conv_layers - 1) ? out : cur;
        hipLaunchKernelGGL(dwconv7_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, cur, w.tx_b, tb.dwk, tb.dwb, tb.lnw,
                           tb.lnb, B, N, Dt, lens_dev, 1e-6f);
        KCHK();
        HIPCHK(launch_gemm<float>(s, w.tx_b, Dt, tb.pw1.w, tb.pw1.ldw, (int)rows, 2 * Dt, Dt,
                                  EpiStore<float>{w.tx_h1, 2 * Dt, tb.pw1.b, F5_ACT_GELU_ERF}));
        hipLaunchKernelGGL(grn_partial_kernel, dim3((2 * Dt + 255) / 256, GRN_P, B), dim3(256), 0, s, w.tx_h1, w.grn_part,
                           N, 2 * Dt, lens_dev);
        KCHK();
        const int rpb = 16;
        hipLaunchKernelGGL(grn_apply_kernel, dim3((N + rpb - 1) / rpb, B), dim3(256), (2 * Dt + 8) * sizeof(float), s,
                           w.tx_h1, w.grn_part, tb.gamma, tb.beta, N, 2 * Dt, lens_dev, rpb);
        KCHK();
        // residual add: nxt = cur + pwconv2(h1)
        HIPCHK(launch_gemm<float>(s, w.tx_h1, 2 * Dt, tb.pw2.w, tb.pw2.ldw, (int)rows, Dt, 2 * Dt,
                                  EpiGateRes{nxt, cur, Dt, tb.pw2.b, nullptr, 0, N, nullptr}));
        if (c.text_mask_padding) {
            hipLaunchKernelGGL(zero_filler_rows_kernel, dim3(ew_blocks(rows * Dt / 4)), dim3(256), 0, s,
                               (const long long*)text, nt, nxt, B, N, Dt);
            KCHK();
        }
        cur = nxt;
    }
    if (lens_dev && c.conv_layers > 0) {
        hipLaunchKernelGGL(zero_tail_rows_kernel, dim3(ew_blocks(rows * Dt / 4)), dim3(256), 0, s, out, B, N, Dt, lens_dev);
        KCHK();
    }
    if (c.options & F5_OPT_TEXT_AVG_UPSAMPLE) {   // dit.py:112-113: after the text blocks; the mask is that of the ORIGINAL ids
        hipLaunchKernelGGL(text_avg_upsample_kernel, dim3(B), dim3(256), (size_t)N * sizeof(int), s, (const long long*)text, nt, out, w.tx_b, N, Dt,
                           lens_dev);
        KCHK();
        HIPCHK(hipMemcpyAsync(out, w.tx_b, (size_t)rows * Dt * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    e->prof.end(s);
    return F5_OK;
}

// One DiT forward over Bp packed rows (dit.py:297-327).  mod_row: AdaLN vectors of this step; mod_stride: distance
// between batch rows' vectors (0 when every row shares the time step, as in sample()).
template <typename T>
static int run_dit_forward(f5_engine* e, Work<T>& w, const float* y, const float* cond, int B, int Bp, int N,
                           const float* mod_row, int mod_stride, const int* lens_dev, int drop_cond_first,
                           const float* text_first, const float* text_second, hipStream_t s, const RowPack& pk = RowPack{}) {
    Packed<T>& P = packed<T>(e);
    const f5_config& c = e->cfg;
    const int D = c.dim, F = c.ff_dim, inner = e->inner, mel = c.mel_dim, H = c.heads;
    const int rows = Bp * N;                        // launch geometry: the padded batch (pk: rows present = *pk.rows_dev)
    const int* ml = pk.rows_dev;                    // device row count of a packed batch, or null
    const int mh = pk ? (int)pk.rows_host : 0;      // rows expected (the call's own lengths): tile choice only
    const int* gate_lens = pk ? nullptr : lens_dev; // (a packed batch has no padded rows to leave untouched)
    Prof& pr = e->prof;
    const double rows_fl = pk ? pk.rows_host : (double)rows;
    auto gflops = [&](double n, double k) { return 2.0 * rows_fl * n * k; };
    const double conv_fl = 2.0 * rows_fl * D * (D / 16) * 31;
    if (e->io_split) {
        // F5_PREC_F16P: [y, cond, text] packed as f32 rows; input projection and conv position embedding as split-f16 products on f32
        // operands (what the f16 blocks then see is x, the f32 residual stream, exactly as in F5_PREC_F16X3)
        float* acat_f = reinterpret_cast<float*>(w.acat);
        pr.begin(PC_MISC, s);
        hipLaunchKernelGGL((pack_input_kernel<float>), dim3(ew_blocks((long)rows * e->kin / 4)), dim3(256), 0, s, y, cond,
                           text_first, text_second, acat_f, e->kin_pad, B, Bp, N, mel, c.text_dim, drop_cond_first, pk.rowmap, ml,
                           lens_dev);
        KCHK();
        pr.end(s);
        pr.begin(PC_GEMM, s, gflops(D, e->kin));
        HIPCHK(launch_gemm<float>(s, acat_f, e->kin_pad, P.in_proj_f.w, P.in_proj_f.ldw, rows, D, e->kin_pad,
                                  EpiStore<float>{w.h, D, P.in_proj_f.b, F5_ACT_NONE}, -1, ml, mh, GemmConv{}, 1));
        pr.end(s);
        const bool cs = convpos_can_split(D);
        pr.begin(PC_CONV, s, conv_fl);
        HIPCHK(launch_convpos<float>(s, w.h, P.conv_w_f[0], P.conv_kp_f, P.conv_b[0], nullptr, w.c1, Bp, N, D, lens_dev, B, pk.row_start, cs));
        pr.end(s);
        pr.begin(PC_CONV, s, conv_fl);
        HIPCHK(launch_convpos<float>(s, w.c1, P.conv_w_f[1], P.conv_kp_f, P.conv_b[1], w.h, w.x, Bp, N, D, lens_dev, B, pk.row_start, cs));
        pr.end(s);
    } else {
    // input embedding
    pr.begin(PC_MISC, s);
    hipLaunchKernelGGL((pack_input_kernel<T>), dim3(ew_blocks((long)rows * e->kin / 4)), dim3(256), 0, s, y, cond,
                       text_first, text_second, w.acat, e->kin_pad, B, Bp, N, mel, c.text_dim, drop_cond_first, pk.rowmap, ml,
                       lens_dev);
    KCHK();
    pr.end(s);
    auto ablate_round = [&](int bit, const float* src, float* dst, long n) -> const float* {   // diagnostic F5_X3_ABLATE: an unsplit f32 operand as f16 sees it
        if (!(e->x3_ablate & bit)) return src;
        hipLaunchKernelGGL(round_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, src, dst, n);
        return dst;
    };
    if constexpr (std::is_same_v<T, float>) ablate_round(64, w.acat, w.acat, (long)rows * e->kin_pad);
    pr.begin(PC_GEMM, s, gflops(D, e->kin));
    HIPCHK(egemm<T>(e, s, w.acat, e->kin_pad, P.in_proj.w, P.in_proj.ldw, rows, D, e->kin_pad,
                          EpiStore<float>{w.h, D, P.in_proj.b, F5_ACT_NONE}, -1, ml, mh));
    pr.end(s);
    pr.begin(PC_CONV, s, conv_fl);
    // (ablation: conv 1 reads a rounded COPY of h -- w.x is free until conv 2 writes it -- because h itself is conv 2's f32 residual)
    const float* conv1_in = ablate_round(128, w.h, w.x, (long)rows * D);
    HIPCHK(launch_convpos<T>(s, conv1_in, P.conv_w[0], P.conv_kp, P.conv_b[0], nullptr, w.c1, Bp, N, D, lens_dev, B, pk.row_start, conv_split(e)));
    pr.end(s);
    ablate_round(128, w.c1, w.c1, (long)rows * D);
    pr.begin(PC_CONV, s, conv_fl);
    HIPCHK(launch_convpos<T>(s, w.c1, P.conv_w[1], P.conv_kp, P.conv_b[1], w.h, w.x, Bp, N, D, lens_dev, B, pk.row_start, conv_split(e)));
    pr.end(s);
    }
    const bool long_skip = (c.options & F5_OPT_LONG_SKIP) != 0;
    if (long_skip)   // residual = x (dit.py:313-314); h is free once the conv embedding has used it as its own residual
        HIPCHK(hipMemcpyAsync(w.h, w.x, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    const bool qk_norm = (c.options & F5_OPT_QK_RMSNORM) != 0;
    const int pe_heads = c.pe_attn_head < 0 ? H : c.pe_attn_head;
    const int* attn_lens = (c.attn_mask_enabled && lens_dev) ? lens_dev : nullptr;
    const int pl = e->split16 ? 1 : 0;   // F5_PREC_F16X3: xn / ao / ffh are written pre-split (the A operands of the block GEMMs)
    auto ablate_a = [&](int bit, T* a, int k) {   // diagnostic F5_X3_ABLATE: the class's A operand as plain f16 (lo halves zeroed)
        if constexpr (std::is_same_v<T, float>) {
            if (e->x3_ablate & bit) {
                const long blocks = (long)rows * k / 32;
                hipLaunchKernelGGL(zero_lo_planar_kernel, dim3(ew_blocks(blocks * 4)), dim3(256), 0, s, a, blocks);
            }
        }
    };
    // weight prefetch from the LayerNorm launches (see layernorm_kernel): only where the GEMMs are latency-bound
    const bool wpf = rows <= 4096 && !(getenv("F5_WEIGHT_PREFETCH") && getenv("F5_WEIGHT_PREFETCH")[0] == '0');
    for (int l = 0; l < c.depth; ++l) {
        BlockW<T>& bw = P.blocks[l];
        const float* m = mod_row + (size_t)l * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((layernorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.xn, D, rows, D, 1e-6f,
                           m + D, m, mod_stride, N, 1,
                           wpf ? Prefetch{(const char*)bw.qkv.w, (size_t)3 * inner * bw.qkv.ldw * sizeof(T),
                                          (const char*)bw.out.w, (size_t)D * bw.out.ldw * sizeof(T)} : Prefetch{}, ml, pl);
        KCHK();
        pr.end(s);
        ablate_a(1, w.xn, D);
        pr.begin(PC_GEMM, s, gflops(3 * inner, D));
        // F5_PREC_F16X3 with plain-f16 attention products (x3_attn_hi == 3): q / k / v^T leave the QKV epilogue as f16 and the fast
        // 16-bit attention kernel writes its output pre-split for the out-projection (the f32 buffers are reused at half their size)
        bool attn16 = false;
        if constexpr (std::is_same_v<T, float>) {
            attn16 = e->split16 && e->x3_attn_hi == 3 && !qk_norm;
            if (attn16)
                HIPCHK(egemm<T>(e, s, w.xn, D, bw.qkv.w, bw.qkv.ldw, rows, 3 * inner, D,
                                EpiQKV<f16_t>{reinterpret_cast<f16_t*>(w.q), reinterpret_cast<f16_t*>(w.k), reinterpret_cast<f16_t*>(w.vt),
                                              bw.qkv.b, P.rope_frag, N, w.Npad, H, pe_heads, attention_q_scale<f16_t>(), pk.rowmap},
                                -1, ml, mh, pl));
        }
        if (!attn16) {
        HIPCHK(egemm<T>(e, s, w.xn, D, bw.qkv.w, bw.qkv.ldw, rows, 3 * inner, D,
                              EpiQKV<T>{w.q, w.k, w.vt, bw.qkv.b, P.rope_frag, N, w.Npad, H, qk_norm ? 0 : pe_heads,
                                        qk_norm ? 1.0f : attention_q_scale<T>(), pk.rowmap},
                              -1, ml, mh, pl));
        }
        pr.end(s);
        if (qk_norm) {   // RMSNorm on q / k comes BEFORE the rotary embedding (modules.py:481-497): the epilogue left both raw
            const long qrows = (long)Bp * H * N;
            pr.begin(PC_MISC, s);
            hipLaunchKernelGGL((qknorm_rope_kernel<T>), dim3((unsigned)((qrows * 16 + 255) / 256)), dim3(256), 0, s, w.q, w.k, bw.qn, bw.kn,
                               P.rope_cos, P.rope_sin, qrows, N, H, pe_heads, attention_q_scale<T>(), 1e-6f);
            KCHK();
            pr.end(s);
        }
        pr.begin(PC_ATTN, s, 4.0 * H * 64 * (pk ? pk.sq_host : (double)Bp * N * N));
        if constexpr (std::is_same_v<T, float>) {
            if (attn16)
                HIPCHK(launch_attention_v2<f16_t>(s, reinterpret_cast<const f16_t*>(w.q), reinterpret_cast<const f16_t*>(w.k),
                                                  reinterpret_cast<const f16_t*>(w.vt), nullptr, Bp, H, N, w.Npad, attn_lens, B, lens_dev,
                                                  pk.row_start, reinterpret_cast<float*>(w.ao)));
        }
        if (!attn16) {
        HIPCHK(launch_attention_any(s, w.q, w.k, w.vt, w.ao, Bp, H, N, w.Npad, attn_lens, B, lens_dev, pk.row_start, e->split16, pl,
                                    e->x3_attn_hi));
        }
        pr.end(s);
        ablate_a(8, w.ao, inner);
        pr.begin(PC_GEMM, s, gflops(D, inner));
        HIPCHK(egemm<T>(e, s, w.ao, inner, bw.out.w, bw.out.ldw, rows, D, inner,
                              EpiGateRes{w.x, w.x, D, bw.out.b, m + 2 * D, mod_stride, N, gate_lens}, -1, ml, mh, pl));
        pr.end(s);
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((layernorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.xn, D, rows, D, 1e-6f,
                           m + 4 * D, m + 3 * D, mod_stride, N, 1,
                           wpf ? Prefetch{(const char*)bw.ff1.w, (size_t)F * bw.ff1.ldw * sizeof(T),
                                          (const char*)bw.ff2.w, (size_t)D * bw.ff2.ldw * sizeof(T)} : Prefetch{}, ml, pl);
        KCHK();
        pr.end(s);
        ablate_a(16, w.xn, D);
        pr.begin(PC_GEMM, s, gflops(F, D));
        HIPCHK(egemm<T>(e, s, w.xn, D, bw.ff1.w, bw.ff1.ldw, rows, F, D, EpiStore<T>{w.ffh, F, bw.ff1.b, F5_ACT_GELU_TANH, pl}, -1, ml, mh, pl));
        pr.end(s);
        ablate_a(32, w.ffh, F);
        pr.begin(PC_GEMM, s, gflops(D, F));
        HIPCHK(egemm<T>(e, s, w.ffh, F, bw.ff2.w, bw.ff2.ldw, rows, D, F,
                              EpiGateRes{w.x, w.x, D, bw.ff2.b, m + 5 * D, mod_stride, N, nullptr}, -1, ml, mh, pl));
        pr.end(s);
    }
    if (long_skip) {   // x = long_skip_connection(cat((x, residual), dim=-1))   (dit.py:323-324)
        pr.begin(PC_MISC, s);
        if (e->io_split) {
            float* cat_f = reinterpret_cast<float*>(w.cat2);
            hipLaunchKernelGGL((cat2_kernel<float>), dim3(ew_blocks((long)rows * 2 * D / 4)), dim3(256), 0, s, w.x, w.h, cat_f, (long)rows, D);
            KCHK();
            pr.end(s);
            pr.begin(PC_GEMM, s, gflops(D, 2 * D));
            HIPCHK(launch_gemm<float>(s, cat_f, 2 * D, P.long_skip_f.w, P.long_skip_f.ldw, rows, D, 2 * D,
                                      EpiStore<float>{w.x, D, nullptr, F5_ACT_NONE}, -1, ml, mh, GemmConv{}, 1));
        } else {
            hipLaunchKernelGGL((cat2_kernel<T>), dim3(ew_blocks((long)rows * 2 * D / 4)), dim3(256), 0, s, w.x, w.h, w.cat2, (long)rows, D);
            KCHK();
            pr.end(s);
            pr.begin(PC_GEMM, s, gflops(D, 2 * D));
            HIPCHK(egemm<T>(e, s, w.cat2, 2 * D, P.long_skip.w, P.long_skip.ldw, rows, D, 2 * D, EpiStore<float>{w.x, D, nullptr, F5_ACT_NONE},
                            -1, ml, mh));
        }
        pr.end(s);
    }
    const float* mf = mod_row + (size_t)c.depth * 6 * D;  // (scale, shift)
    if (e->io_split) {   // F5_PREC_F16P: final norm to f32 (pre-split rows in c1, free since the conv embedding) + split output projection
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((layernorm_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.c1, D, rows, D, 1e-6f, mf,
                           mf + D, mod_stride, N, 1, Prefetch{}, ml, 1);
        KCHK();
        pr.end(s);
        pr.begin(PC_GEMM, s, gflops(mel, D));
        HIPCHK(launch_gemm<float>(s, w.c1, D, P.proj_out_f.w, P.proj_out_f.ldw, rows, mel, D,
                                  EpiStore<float>{w.pred, mel, P.proj_out_f.b, F5_ACT_NONE}, -1, ml, mh, GemmConv{}, 2));
        pr.end(s);
        return F5_OK;
    }
    pr.begin(PC_LN, s);
    hipLaunchKernelGGL((layernorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.xn, D, rows, D, 1e-6f, mf,
                       mf + D, mod_stride, N, 1, Prefetch{}, ml, pl);
    KCHK();
    pr.end(s);
    ablate_a(256, w.xn, D);
    pr.begin(PC_GEMM, s, gflops(mel, D));
    HIPCHK(egemm<T>(e, s, w.xn, D, P.proj_out.w, P.proj_out.ldw, rows, mel, D,
                          EpiStore<float>{w.pred, mel, P.proj_out.b, F5_ACT_NONE}, -1, ml, mh, pl));
    pr.end(s);
    return F5_OK;
}

// One UNetT forward over Bp packed rows (unett.py:217-280): time token prepended (N + 1 tokens per row), concat skip
// connections, x_transformers RMSNorm, no AdaLN.  temb: time embedding rows ([1, D] shared when temb_stride == 0).
template <typename T>
static int run_unett_forward(f5_engine* e, Work<T>& w, const float* y, const float* cond, int B, int Bp, int N,
                             const float* temb, int temb_stride, const int* lens_dev, int drop_cond_first,
                             const float* text_first, const float* text_second, hipStream_t s) {
    Packed<T>& P = packed<T>(e);
    const f5_config& c = e->cfg;
    const int D = c.dim, F = c.ff_dim, inner = e->inner, mel = c.mel_dim, H = c.heads;
    const int Nt = N + 1;
    const int rows_in = Bp * N, rows = Bp * Nt;
    Prof& pr = e->prof;
    auto gfl = [&](double r, double n, double k) { return 2.0 * r * n * k; };
    const double conv_fl = 2.0 * rows_in * D * (D / 16) * 31;
    float* emb = reinterpret_cast<float*>(w.cat2);  // [rows, 2D] of T >= [rows_in, D] f32; free until the first concat; conv input and output must not alias
    if (e->io_split) {   // F5_PREC_F16P: input projection + conv position embedding as split-f16 products on f32 operands (run_dit_forward)
        float* acat_f = reinterpret_cast<float*>(w.acat);
        pr.begin(PC_MISC, s);
        hipLaunchKernelGGL((pack_input_kernel<float>), dim3(ew_blocks((long)rows_in * e->kin / 4)), dim3(256), 0, s, y, cond,
                           text_first, text_second, acat_f, e->kin_pad, B, Bp, N, mel, c.text_dim, drop_cond_first);
        KCHK();
        pr.end(s);
        pr.begin(PC_GEMM, s, gfl(rows_in, D, e->kin));
        HIPCHK(launch_gemm<float>(s, acat_f, e->kin_pad, P.in_proj_f.w, P.in_proj_f.ldw, rows_in, D, e->kin_pad,
                                  EpiStore<float>{w.h, D, P.in_proj_f.b, F5_ACT_NONE}, -1, nullptr, 0, GemmConv{}, 1));
        pr.end(s);
        const bool cs = convpos_can_split(D);
        pr.begin(PC_CONV, s, conv_fl);
        HIPCHK(launch_convpos<float>(s, w.h, P.conv_w_f[0], P.conv_kp_f, P.conv_b[0], nullptr, w.c1, Bp, N, D, nullptr, B, nullptr, cs));
        pr.end(s);
        pr.begin(PC_CONV, s, conv_fl);
        HIPCHK(launch_convpos<float>(s, w.c1, P.conv_w_f[1], P.conv_kp_f, P.conv_b[1], w.h, emb, Bp, N, D, nullptr, B, nullptr, cs));
        pr.end(s);
    } else {
    pr.begin(PC_MISC, s);
    hipLaunchKernelGGL((pack_input_kernel<T>), dim3(ew_blocks((long)rows_in * e->kin / 4)), dim3(256), 0, s, y, cond,
                       text_first, text_second, w.acat, e->kin_pad, B, Bp, N, mel, c.text_dim, drop_cond_first);
    KCHK();
    pr.end(s);
    pr.begin(PC_GEMM, s, gfl(rows_in, D, e->kin));
    HIPCHK(egemm<T>(e, s, w.acat, e->kin_pad, P.in_proj.w, P.in_proj.ldw, rows_in, D, e->kin_pad,
                          EpiStore<float>{w.h, D, P.in_proj.b, F5_ACT_NONE}));
    pr.end(s);
    pr.begin(PC_CONV, s, conv_fl);   // unett.py:99-100: conv_pos_embed is called WITHOUT a mask
    HIPCHK(launch_convpos<T>(s, w.h, P.conv_w[0], P.conv_kp, P.conv_b[0], nullptr, w.c1, Bp, N, D, nullptr, B, nullptr, conv_split(e)));
    pr.end(s);
    pr.begin(PC_CONV, s, conv_fl);
    HIPCHK(launch_convpos<T>(s, w.c1, P.conv_w[1], P.conv_kp, P.conv_b[1], w.h, emb, Bp, N, D, nullptr, B, nullptr, conv_split(e)));
    pr.end(s);
    }
    pr.begin(PC_MISC, s);
    // The skip stack needs no copies: the stream entering block l < depth/2 is WRITTEN into skip slot l (by the assemble kernel / the
    // FF2 epilogue of block l - 1), the block's first residual update reads it there and writes w.x, and nothing writes the slot again.
    const int half = c.depth / 2;
    auto slot = [&](int l) { return w.skips + (size_t)l * rows * D; };
    hipLaunchKernelGGL(unett_assemble_kernel, dim3(ew_blocks((long)rows * D / 4)), dim3(256), 0, s, emb, temb, temb_stride,
                       half > 0 ? slot(0) : w.x, Bp, N, D);
    KCHK();
    pr.end(s);
    const int pe_heads = c.pe_attn_head < 0 ? H : c.pe_attn_head;
    const int* attn_lens = (c.attn_mask_enabled && lens_dev) ? lens_dev : nullptr;   // lens_dev holds len + 1 for UNetT
    const int pl = e->split16 ? 1 : 0;   // F5_PREC_F16X3: xn / ao / ffh written pre-split (as in run_dit_forward)
    auto ablate_a = [&](int bit, T* a, int k) {   // diagnostic F5_X3_ABLATE (run_dit_forward): the class's A operand as plain f16
        if constexpr (std::is_same_v<T, float>) {
            if (e->x3_ablate & bit) {
                const long blocks = (long)rows * k / 32;
                hipLaunchKernelGGL(zero_lo_planar_kernel, dim3(ew_blocks(blocks * 4)), dim3(256), 0, s, a, blocks);
            }
        }
    };
    for (int l = 0; l < c.depth; ++l) {
        BlockW<T>& bw = P.blocks[l];
        const float* x_in = l < half ? slot(l) : w.x;            // the stream entering the block (= skips.append(x), unett.py:258-259)
        float* x_out = l + 1 < half ? slot(l + 1) : w.x;         // where the block leaves it
        if (l >= half) {
            const float* skip = w.skips + (size_t)(c.depth - 1 - l) * rows * D;   // LIFO (skips.pop())
            pr.begin(PC_MISC, s);
            hipLaunchKernelGGL((cat2_kernel<T>), dim3(ew_blocks((long)rows * 2 * D / 4)), dim3(256), 0, s, w.x, skip, w.cat2,
                               (long)rows, D, pl);   // (F5_PREC_F16X3: pre-split, so that the projection runs on pre-split operands too)
            KCHK();
            pr.end(s);
            ablate_a(512, w.cat2, 2 * D);   // diagnostic: ... and its A operand as plain f16
            pr.begin(PC_GEMM, s, gfl(rows, D, 2 * D));
            HIPCHK(egemm<T>(e, s, w.cat2, 2 * D, bw.skip.w, bw.skip.ldw, rows, D, 2 * D,
                                  EpiStore<float>{w.x, D, nullptr, F5_ACT_NONE}, -1, nullptr, 0, pl));
            pr.end(s);
        }
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((xrmsnorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, x_in, D, w.xn, D, rows, D, bw.norm1_g, pl);
        KCHK();
        pr.end(s);
        ablate_a(1, w.xn, D);
        pr.begin(PC_GEMM, s, gfl(rows, 3 * inner, D));
        bool attn16 = false;   // (run_dit_forward)
        if constexpr (std::is_same_v<T, float>) {
            attn16 = e->split16 && e->x3_attn_hi == 3;
            if (attn16)
                HIPCHK(egemm<T>(e, s, w.xn, D, bw.qkv.w, bw.qkv.ldw, rows, 3 * inner, D,
                                EpiQKV<f16_t>{reinterpret_cast<f16_t*>(w.q), reinterpret_cast<f16_t*>(w.k), reinterpret_cast<f16_t*>(w.vt),
                                              bw.qkv.b, P.rope_frag, Nt, w.Npad, H, pe_heads, attention_q_scale<f16_t>()},
                                -1, nullptr, 0, pl));
        }
        if (!attn16) {
        HIPCHK(egemm<T>(e, s, w.xn, D, bw.qkv.w, bw.qkv.ldw, rows, 3 * inner, D,
                              EpiQKV<T>{w.q, w.k, w.vt, bw.qkv.b, P.rope_frag, Nt, w.Npad, H, pe_heads, attention_q_scale<T>()},
                              -1, nullptr, 0, pl));
        }
        pr.end(s);
        pr.begin(PC_ATTN, s, 4.0 * Bp * H * (double)Nt * Nt * 64);
        if constexpr (std::is_same_v<T, float>) {
            if (attn16)
                HIPCHK(launch_attention_v2<f16_t>(s, reinterpret_cast<const f16_t*>(w.q), reinterpret_cast<const f16_t*>(w.k),
                                                  reinterpret_cast<const f16_t*>(w.vt), nullptr, Bp, H, Nt, w.Npad, attn_lens, B, lens_dev,
                                                  nullptr, reinterpret_cast<float*>(w.ao)));
        }
        if (!attn16) {
        HIPCHK(launch_attention_any(s, w.q, w.k, w.vt, w.ao, Bp, H, Nt, w.Npad, attn_lens, B, lens_dev, nullptr, e->split16, pl,
                                    e->x3_attn_hi));
        }
        pr.end(s);
        ablate_a(8, w.ao, inner);
        pr.begin(PC_GEMM, s, gfl(rows, D, inner));
        HIPCHK(egemm<T>(e, s, w.ao, inner, bw.out.w, bw.out.ldw, rows, D, inner,
                              EpiGateRes{w.x, x_in, D, bw.out.b, nullptr, 0, Nt, lens_dev}, -1, nullptr, 0, pl));
        pr.end(s);
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((xrmsnorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.xn, D, rows, D, bw.norm2_g, pl);
        KCHK();
        pr.end(s);
        ablate_a(16, w.xn, D);
        pr.begin(PC_GEMM, s, gfl(rows, F, D));
        HIPCHK(egemm<T>(e, s, w.xn, D, bw.ff1.w, bw.ff1.ldw, rows, F, D, EpiStore<T>{w.ffh, F, bw.ff1.b, F5_ACT_GELU_TANH, pl}, -1, nullptr, 0, pl));
        pr.end(s);
        ablate_a(32, w.ffh, F);
        pr.begin(PC_GEMM, s, gfl(rows, D, F));
        HIPCHK(egemm<T>(e, s, w.ffh, F, bw.ff2.w, bw.ff2.ldw, rows, D, F, EpiGateRes{x_out, w.x, D, bw.ff2.b, nullptr, 0, Nt, nullptr}, -1, nullptr, 0, pl));
        pr.end(s);
    }
    if (e->io_split) {   // F5_PREC_F16P: final norm to pre-split f32 rows (in the skip stack's first slot: every skip has been popped) + split projection
        float* xn_f = w.skips;
        pr.begin(PC_LN, s);
        hipLaunchKernelGGL((xrmsnorm_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, xn_f, D, rows, D, P.norm_out_g, 1);
        KCHK();
        pr.end(s);
        pr.begin(PC_GEMM, s, gfl(rows, mel, D));
        HIPCHK(launch_gemm<float>(s, xn_f, D, P.proj_out_f.w, P.proj_out_f.ldw, rows, mel, D,
                                  EpiStore<float>{w.pred_all, mel, P.proj_out_f.b, F5_ACT_NONE}, -1, nullptr, 0, GemmConv{}, 2));
        pr.end(s);
    } else {
    pr.begin(PC_LN, s);
    hipLaunchKernelGGL((xrmsnorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, w.x, D, w.xn, D, rows, D, P.norm_out_g, pl);
    KCHK();
    pr.end(s);
    pr.begin(PC_GEMM, s, gfl(rows, mel, D));
    HIPCHK(egemm<T>(e, s, w.xn, D, P.proj_out.w, P.proj_out.ldw, rows, mel, D, EpiStore<float>{w.pred_all, mel, P.proj_out.b, F5_ACT_NONE}, -1, nullptr, 0, pl));
    pr.end(s);
    }
    pr.begin(PC_MISC, s);
    hipLaunchKernelGGL(strip_first_token_kernel, dim3(ew_blocks((long)rows_in * mel / 4)), dim3(256), 0, s, w.pred_all, w.pred, Bp, N, mel);
    KCHK();
    pr.end(s);
    return F5_OK;
}

// dispatches one backbone forward; `step_row` selects the time step's vectors inside the per-call tables
template <typename T>
static int run_backbone(f5_engine* e, Work<T>& w, const float* y, const float* cond, int B, int Bp, int N, int step_row,
                        int per_row_time, const int* lens_dev, int drop_cond_first, const float* text_first,
                        const float* text_second, hipStream_t s, const RowPack& pk = RowPack{}) {
    if (e->cfg.backbone == F5_BACKBONE_DIT)
        return run_dit_forward<T>(e, w, y, cond, B, Bp, N, w.mod + (size_t)step_row * e->modN, per_row_time ? e->modN : 0,
                                  lens_dev, drop_cond_first, text_first, text_second, s, pk);
    return run_unett_forward<T>(e, w, y, cond, B, Bp, N, w.temb + (size_t)step_row * e->cfg.dim, per_row_time ? e->cfg.dim : 0,
                                lens_dev, drop_cond_first, text_first, text_second, s);
}
// lens bookkeeping: uploads per-sample lengths (duplicated for the uncond half) through pinned staging
template <typename T>
static int upload_small(f5_engine* e, Work<T>& w, const float* t_host, int nT, const int32_t* lens_host, int B,
                        hipStream_t s, int chunk = 0, int halves = 2) {
    if (chunk <= 0 || chunk > B) chunk = B;
    const size_t bytes = (size_t)nT * 4 + (size_t)3 * B * 4 + ((size_t)3 * B + 16) * 4 + 64;
    char* hb = nullptr;
    int slot = 0;
    CHK(e->stage.acquire(bytes, &hb, &slot));
    float* th = reinterpret_cast<float*>(hb);
    int* lh = reinterpret_cast<int*>(hb + (size_t)nT * 4);
    if (nT > 0) {
        memcpy(th, t_host, (size_t)nT * 4);
        HIPCHK(hipMemcpyAsync(w.tdev, th, (size_t)nT * 4, hipMemcpyHostToDevice, s));
    }
    if (lens_host) {
        const int add = e->cfg.backbone == F5_BACKBONE_UNETT ? 1 : 0;  // UNetT masks are left-padded for the time token
        for (int u0 = 0; u0 < B; u0 += chunk) {   // chunk-major: [cond lens of the chunk][uncond lens of the chunk]
            const int bc = std::min(chunk, B - u0);
            for (int i = 0; i < bc; ++i) lh[2 * u0 + i] = lh[2 * u0 + bc + i] = lens_host[u0 + i] + add;
        }
        for (int i = 0; i < B; ++i) lh[2 * B + i] = lens_host[i] + add;
        HIPCHK(hipMemcpyAsync(w.lens, lh, (size_t)2 * B * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(w.lens_plain, lh + 2 * B, (size_t)B * 4, hipMemcpyHostToDevice, s));
        // RowPack tables: per chunk, `halves` x Bc batch rows (cond half, then uncond half), each rounded up to 4 rows;
        // chunk c's table starts at halves * u0 + c (every chunk has one entry more than it has batch rows)
        int* rs = lh + 3 * B;
        e->pack_rows_host.clear();
        e->pack_sq_host.clear();
        int cidx = 0, nrs = 0;
        for (int u0 = 0; u0 < B; u0 += chunk, ++cidx) {
            const int bc = std::min(chunk, B - u0);
            int* t = rs + halves * u0 + cidx;
            double sq = 0;
            t[0] = 0;
            for (int k = 0; k < halves * bc; ++k) {
                const int len = lens_host[u0 + k % bc];
                t[k + 1] = t[k] + round_up(len, 4);
                sq += (double)len * len;
            }
            e->pack_rows_host.push_back((double)t[halves * bc]);
            e->pack_sq_host.push_back(sq);
            nrs = halves * u0 + cidx + halves * bc + 1;
        }
        HIPCHK(hipMemcpyAsync(w.row_start, rs, (size_t)nrs * 4, hipMemcpyHostToDevice, s));
    }
    return e->stage.release(slot, s);
}
template <typename T>
static int text_embed_impl(f5_engine* e, const int64_t* text, int B, int nt, const int32_t* lens_host, int N,
                           int drop_text, float* out, hipStream_t s) {
    CHK(ensure_arena(e, B, N, 1));
    Work<T> w;
    carve<T>(e, w, e->res_B, e->res_N, e->res_S);
    CHK(upload_small<T>(e, w, nullptr, 0, lens_host, B, s));
    const bool per_sample = lens_host && e->cfg.backbone == F5_BACKBONE_DIT;  // unett.py embeds at the padded length
    return run_text_embed<T>(e, w, text, B, nt, per_sample ? w.lens_plain : nullptr, N, drop_text, out, s);
}
template <typename T>
static int forward_impl(f5_engine* e, const float* x, const float* cond, const int64_t* text, int nt,
                        const float* time_host, const int32_t* lens_host, int B, int N, int cfg_infer,
                        int drop_audio_cond, int drop_text, float* out, hipStream_t s) {
    const int Bp = cfg_infer ? 2 * B : B;
    CHK(ensure_arena(e, B, N, Bp));
    Work<T> w;
    carve<T>(e, w, e->res_B, e->res_N, e->res_S);
    std::vector<float> tt(Bp);
    for (int i = 0; i < Bp; ++i) tt[i] = time_host[i % B];
    CHK(upload_small<T>(e, w, tt.data(), Bp, lens_host, B, s));
    const int* lens_dev = lens_host ? w.lens : nullptr;
    CHK(run_time_path<T>(e, w, Bp, s));
    // UNetT embeds text at the padded length for every sample (unett.py:196-215), DiT at each sample's own length
    const int* tlens = (e->cfg.backbone == F5_BACKBONE_DIT && lens_host) ? w.lens_plain : nullptr;
    if (cfg_infer) {
        CHK(run_text_embed<T>(e, w, text, B, nt, tlens, N, 0, w.text_c, s));
        CHK(run_text_embed<T>(e, w, text, B, nt, tlens, N, 1, w.text_u, s));
        CHK(run_backbone<T>(e, w, x, cond, B, Bp, N, 0, 1, lens_dev, 0, w.text_c, w.text_u, s));
    } else {
        CHK(run_text_embed<T>(e, w, text, B, nt, tlens, N, drop_text, w.text_c, s));
        CHK(run_backbone<T>(e, w, x, cond, B, Bp, N, 0, 1, lens_dev, drop_audio_cond, w.text_c, w.text_c, s));
    }
    HIPCHK(hipMemcpyAsync(out, w.pred, (size_t)Bp * N * e->cfg.mel_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    return F5_OK;
}
// the second (unconditional) half of every [2B ...] activation buffer, as a Work of its own
template <typename T> static Work<T> second_half(const f5_engine* e, const Work<T>& w, int B, int N) {
    const f5_config& c = e->cfg;
    const size_t rows = (size_t)B * N;
    Work<T> h = w;
    h.acat = w.acat + rows * e->kin_pad;
    h.h = w.h + rows * c.dim;
    h.c1 = w.c1 + rows * c.dim;
    h.x = w.x + rows * c.dim;
    h.pred = w.pred + rows * c.mel_dim;
    h.xn = w.xn + rows * c.dim;
    h.q = w.q + rows * e->inner;
    h.k = w.k + rows * e->inner;
    h.ao = w.ao + rows * e->inner;
    h.vt = w.vt + (size_t)B * c.heads * 64 * w.Npad;
    h.ffh = w.ffh + rows * c.ff_dim;
    return h;
}
// The unconditional text embedding (every token replaced by the filler) of a single utterance depends only on the weights
// and N -- unless text_mask_padding zeroes the rows whose ORIGINAL token is the filler (dit.py:90-91,104-108: the mask is
// taken before drop_text), which makes it a function of the call's text: no cache then.
static bool uc_cacheable(const f5_engine* e, int B, bool has_lens) {
    return B == 1 && !has_lens && !(e->cfg.text_mask_padding && e->cfg.conv_layers > 0) && !(e->cfg.options & F5_OPT_TEXT_AVG_UPSAMPLE);
}

// the stream-ordered body of sample(): everything between "inputs are in the arena" and "outputs are in the arena"
template <typename T>
static int sample_body(f5_engine* e, Work<T>& w, int nt, int steps, float cfg_strength, bool has_lens, int B, int N,
                       bool want_traj, hipStream_t s) {
    const f5_config& c = e->cfg;
    const int mel = c.mel_dim;
    const bool use_cfg = !(cfg_strength < 1e-5f);
    const int Bp = use_cfg ? 2 * B : B;
    const int* lens_dev = has_lens ? w.lens : nullptr;
    const long half = (long)B * N * mel;
    const int64_t* text = reinterpret_cast<const int64_t*>(w.in_text);
    // step_cond = where(cond_mask, cond, 0)   (cfm.py:151-153)
    e->prof.begin(PC_MISC, s);
    hipLaunchKernelGGL(select_rows_kernel, dim3(ew_blocks(half / 4)), dim3(256), 0, s, w.in_cond, (const float*)nullptr,
                       w.in_mask, w.step_cond, (long)B * N, mel);
    KCHK();
    e->prof.end(s);
    CHK(run_time_path<T>(e, w, steps, s));  // features of t[0..steps-1]
    const int* tlens = (c.backbone == F5_BACKBONE_DIT && has_lens) ? w.lens_plain : nullptr;
    CHK(run_text_embed<T>(e, w, text, B, nt, tlens, N, 0, w.text_c, s));
    if (use_cfg) {
        const size_t ucn = (size_t)N * c.text_dim;
        if (uc_cacheable(e, B, has_lens) && e->uc_N == N) {
            HIPCHK(hipMemcpyAsync(w.text_u, w.uc, ucn * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else {
            CHK(run_text_embed<T>(e, w, text, B, nt, tlens, N, 1, w.text_u, s));
            if (uc_cacheable(e, B, has_lens)) {   // (never reached under capture: a cache miss always runs eagerly)
                HIPCHK(hipMemcpyAsync(w.uc, w.text_u, ucn * sizeof(float), hipMemcpyDeviceToDevice, s));
                e->uc_N = N;
            }
        }
    }
    if (want_traj) HIPCHK(hipMemcpyAsync(w.traj_buf, w.y, half * sizeof(float), hipMemcpyDeviceToDevice, s));
    const bool split = use_cfg && c.backbone == F5_BACKBONE_DIT && !e->prof.on && split_cfg_enabled(e);
    Work<T> w2 = w;
    if (split) {
        w2 = second_half<T>(e, w, B, N);
        if (!e->side_stream) HIPCHK(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
        if (!e->ev_fork) HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        if (!e->ev_join) HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    const int chunk = e->cur_chunk;   // (chunk_utts, decided by sample_impl, which laid w.lens / w.row_start out for this size)
    // packed variable-length batch (RowPack): the row tables of every chunk, built once from the uploaded prefix sums
    const bool pack = has_lens && pack_rows_enabled(e);
    const int halves = use_cfg ? 2 : 1;
    auto chunk_pack = [&](int u0, int cidx, int bc) {
        RowPack pk;
        if (!pack) return pk;
        pk.row_start = w.row_start + halves * u0 + cidx;
        pk.rowmap = w.rowmap + (size_t)2 * u0 * round_up(N, 4);
        pk.rows_dev = pk.row_start + halves * bc;
        if ((size_t)cidx < e->pack_rows_host.size()) { pk.rows_host = e->pack_rows_host[cidx]; pk.sq_host = e->pack_sq_host[cidx]; }
        return pk;
    };
    if (pack) {
        int cidx = 0;
        for (int u0 = 0; u0 < B; u0 += chunk, ++cidx) {
            const int bc = std::min(chunk, B - u0);
            const RowPack pk = chunk_pack(u0, cidx, bc);
            hipLaunchKernelGGL(fill_rowmap_kernel, dim3(halves * bc), dim3(256), 0, s, pk.row_start, const_cast<int2*>(pk.rowmap));
            KCHK();
        }
    }
    for (int i = 0; i < steps; ++i) {
        if (split) {
            hipStream_t s1 = e->side_stream;
            HIPCHK(hipEventRecord(e->ev_fork, s));            // y of this step (and, first time, the text embeddings) ready
            HIPCHK(hipStreamWaitEvent(s1, e->ev_fork, 0));
            // conditional chain on s, unconditional chain (cond dropped, filler text) on the side stream
            CHK(run_backbone<T>(e, w, w.y, w.step_cond, B, B, N, i, 0, lens_dev, 0, w.text_c, w.text_c, s));
            CHK(run_backbone<T>(e, w2, w.y, w.step_cond, B, B, N, i, 0, lens_dev, 1, w.text_u, w.text_u, s1));
            HIPCHK(hipEventRecord(e->ev_join, s1));
            HIPCHK(hipStreamWaitEvent(s, e->ev_join, 0));
            e->prof.begin(PC_MISC, s);
            hipLaunchKernelGGL(euler_cfg_kernel, dim3(ew_blocks(half / 4)), dim3(256), 0, s, w.y, w.pred, half, w.tdev, i,
                               cfg_strength, use_cfg ? 1 : 0, want_traj ? w.traj_buf + (size_t)(i + 1) * half : nullptr);
            KCHK();
            e->prof.end(s);
            continue;
        }
        int cidx = 0;
        for (int u0 = 0; u0 < B; u0 += chunk, ++cidx) {
            const int bc = std::min(chunk, B - u0);
            const size_t yo = (size_t)u0 * N * mel, to = (size_t)u0 * N * c.text_dim;
            const long half_c = (long)bc * N * mel;
            const RowPack pk = chunk_pack(u0, cidx, bc);
            CHK(run_backbone<T>(e, w, w.y + yo, w.step_cond + yo, bc, use_cfg ? 2 * bc : bc, N, i, 0,
                                lens_dev ? lens_dev + 2 * u0 : nullptr, 0, w.text_c + to,
                                use_cfg ? w.text_u + to : w.text_c + to, s, pk));
            e->prof.begin(PC_MISC, s);
            float* slot = want_traj ? w.traj_buf + (size_t)(i + 1) * half + yo : nullptr;
            if (pk)
                hipLaunchKernelGGL(euler_cfg_packed_kernel, dim3(ew_blocks(half_c / 4)), dim3(256), 0, s, w.y + yo, w.pred, bc, N,
                                   mel, pk.row_start, lens_dev + 2 * u0, w.tdev, i, cfg_strength, use_cfg ? 1 : 0, slot);
            else
                hipLaunchKernelGGL(euler_cfg_kernel, dim3(ew_blocks(half_c / 4)), dim3(256), 0, s, w.y + yo, w.pred, half_c, w.tdev,
                                   i, cfg_strength, use_cfg ? 1 : 0, slot);
            KCHK();
            e->prof.end(s);
        }
    }
    // out = where(cond_mask, cond, y)   (cfm.py:221-223)
    e->prof.begin(PC_MISC, s);
    hipLaunchKernelGGL(select_rows_kernel, dim3(ew_blocks(half / 4)), dim3(256), 0, s, w.in_cond, (const float*)w.y, w.in_mask,
                       w.out_buf, (long)B * N, mel);
    KCHK();
    e->prof.end(s);
    return F5_OK;
}
template <typename T>
static int sample_impl(f5_engine* e, const float* cond, int cond_frames, const uint8_t* cond_mask, const float* y0, const int64_t* text,
                       int nt, const float* t_host, int steps, float cfg_strength, const int32_t* lens_host, int B, int N,
                       float* out, float* traj, hipStream_t s) {
    const int mel = e->cfg.mel_dim;
    const long half = (long)B * N * mel;
    if (nt > e->res_nt) {   // the text staging buffer is part of the arena plan
        e->res_nt = nt;
        e->clear_graphs();
    }
    CHK(ensure_arena(e, B, N, steps));
    Work<T> w;
    carve<T>(e, w, e->res_B, e->res_N, e->res_S);
    // ---- inputs -> arena (eager, on the caller's stream)
    e->cur_chunk = chunk_utts(e, B, N, !(cfg_strength < 1e-5f), lens_host);
    CHK(upload_small<T>(e, w, t_host, steps + 1, lens_host, B, s, e->cur_chunk, cfg_strength < 1e-5f ? 1 : 2));
    if (cond_frames < N) HIPCHK(hipMemsetAsync(w.in_cond, 0, half * sizeof(float), s));   // F.pad(cond, ..., N - cond_seq_len) (cfm.py:145)
    if (cond_frames > 0)
        HIPCHK(hipMemcpy2DAsync(w.in_cond, (size_t)N * mel * sizeof(float), cond, (size_t)cond_frames * mel * sizeof(float),
                                (size_t)cond_frames * mel * sizeof(float), B, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(w.y, y0, half * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(w.in_mask, cond_mask, (size_t)B * N, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(w.in_text, text, (size_t)B * nt * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    // ---- body: replay a captured graph when this signature has been seen, else run eagerly (and remember it)
    const bool use_cfg = !(cfg_strength < 1e-5f);
    const bool uc_ok = use_cfg && uc_cacheable(e, B, lens_host != nullptr);
    const bool uc_hit = uc_ok && e->uc_N == N;
    const bool uc_store = uc_ok && !uc_hit;
    unsigned cfg_bits;
    memcpy(&cfg_bits, &cfg_strength, 4);
    char kb[160];
    snprintf(kb, sizeof(kb), "%d|%d|%d|%d|%08x|%d|%d|%d", B, N, nt, steps, cfg_bits, lens_host ? 1 : 0, traj ? 1 : 0,
             e->cur_chunk);
    const std::string base_key(kb);
    const std::string key = base_key + (uc_hit ? "|uc" : "|nouc");
    bool done = false;
    static const bool trace = getenv("F5_TRACE") && getenv("F5_TRACE")[0] == '1';   // diagnostic: which path a call takes
    const auto t_body = std::chrono::steady_clock::now();
    auto trace_done = [&](const char* how) {
        if (trace)
            fprintf(stderr, "libf5hip: sample %s [%s]: host %.3f ms\n", key.c_str(), how,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_body).count());
    };
    if (graphs_enabled(e) && !e->prof.on && !uc_store) {
        for (auto& g : e->graphs)
            if (g.key == key) {
                HIPCHK(hipGraphLaunch(g.exec, s));
                done = true;
                trace_done("graph replay");
                break;
            }
        const bool is_warm = std::find(e->warm.begin(), e->warm.end(), base_key) != e->warm.end();
        if (!done && is_warm) {
            if (!e->cap_stream) HIPCHK(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
                const int rc = sample_body<T>(e, w, nt, steps, cfg_strength, lens_host != nullptr, B, N, traj != nullptr, e->cap_stream);
                const hipError_t ce = hipStreamEndCapture(e->cap_stream, &graph);
                if (rc == F5_OK && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    if (e->graphs.size() >= 16) {
                        (void)hipGraphExecDestroy(e->graphs.front().exec);
                        (void)hipGraphDestroy(e->graphs.front().graph);
                        e->graphs.erase(e->graphs.begin());
                    }
                    e->graphs.push_back({key, graph, exec});
                    HIPCHK(hipGraphLaunch(exec, s));
                    done = true;
                    trace_done("graph capture + instantiate + launch");
                } else {
                    // capture is an optimisation: fall back to eager launches, but say so (the wall time doubles on a busy host)
                    fprintf(stderr, "libf5hip: HIP graph capture of sample() failed (body rc %d, end-capture: %s, last: %s); "
                                    "continuing with eager launches\n", rc, hipGetErrorString(ce), hipGetErrorString(hipGetLastError()));
                    if (graph) (void)hipGraphDestroy(graph);
                    e->graphs_on = 0;
                }
            } else {
                fprintf(stderr, "libf5hip: hipStreamBeginCapture failed (%s); continuing with eager launches\n",
                        hipGetErrorString(hipGetLastError()));
                e->graphs_on = 0;
            }
        }
    }
    if (!done) {
        CHK(sample_body<T>(e, w, nt, steps, cfg_strength, lens_host != nullptr, B, N, traj != nullptr, s));
        trace_done("eager launches");
        if (std::find(e->warm.begin(), e->warm.end(), base_key) == e->warm.end()) e->warm.push_back(base_key);
    }
    // ---- outputs -> caller
    HIPCHK(hipMemcpyAsync(out, w.out_buf, half * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (traj) HIPCHK(hipMemcpyAsync(traj, w.traj_buf, (size_t)(steps + 1) * half * sizeof(float), hipMemcpyDeviceToDevice, s));
    return F5_OK;
}

// ------------------------------------------------------------------------------------------ EngineOps<T>
template <typename T> int EngineOps<T>::finalize(f5_engine* e, hipStream_t s) { return finalize_t<T>(e, packed<T>(e), s); }
template <typename T> size_t EngineOps<T>::plan_bytes(const f5_engine* e, int B, int N, int S) {
    Arena dry;  // base == nullptr: measures only
    Work<T> w;
    return carve_into<T>(e, dry, w, B, N, S);
}
template <typename T>
int EngineOps<T>::text_embed(f5_engine* e, const int64_t* text, int B, int nt, const int32_t* lens_host, int N, int drop_text,
                             float* out, hipStream_t s) {
    return text_embed_impl<T>(e, text, B, nt, lens_host, N, drop_text, out, s);
}
template <typename T>
int EngineOps<T>::forward(f5_engine* e, const float* x, const float* cond, const int64_t* text, int nt, const float* time_host,
                          const int32_t* lens_host, int B, int N, int cfg_infer, int drop_audio_cond, int drop_text, float* out,
                          hipStream_t s) {
    return forward_impl<T>(e, x, cond, text, nt, time_host, lens_host, B, N, cfg_infer, drop_audio_cond, drop_text, out, s);
}
template <typename T>
int EngineOps<T>::sample(f5_engine* e, const float* cond, int cond_frames, const uint8_t* cond_mask, const float* y0, const int64_t* text, int nt,
                         const float* t_host, int steps, float cfg_strength, const int32_t* lens_host, int B, int N, float* out,
                         float* traj, hipStream_t s) {
    return sample_impl<T>(e, cond, cond_frames, cond_mask, y0, text, nt, t_host, steps, cfg_strength, lens_host, B, N, out, traj, s);
}
