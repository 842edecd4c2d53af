// Engine state shared by the translation units of libf5hip.so: containers, packed weights, the f5_engine handle, the
// per-call workspace and the per-precision entry points (EngineOps<T>, instantiated once per operand type in
// engine_bf16.hip / engine_f16.hip / engine_f32.hip so that the precisions compile in parallel; F5_PREC_F16X3 is the float
// instantiation with f5_engine::split16 set).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "internal.h"
#include "attn2.h"
#include "convpos.h"
#include "elementwise.h"
#include "gemm_dispatch.h"

using namespace f5;
#define fail f5_fail
// ------------------------------------------------------------------------------------------- containers
struct Tensor {
    float* p = nullptr;
    std::vector<int64_t> shape;
    size_t numel() const {
        size_t n = 1;
        for (auto s : shape) n *= (size_t)s;
        return n;
    }
};

struct WeightStore {
    std::map<std::string, Tensor> t;
    ~WeightStore() {
        for (auto& kv : t)
            if (kv.second.p) (void)hipFree(kv.second.p);
    }
    int put(const char* name, const void* dev, const int64_t* shape, int ndim, hipStream_t s) {
        if (!name || !dev || ndim < 0 || ndim > 4) return fail(F5_EINVAL, "f5_load_weight: bad arguments");
        Tensor T;
        T.shape.assign(shape, shape + ndim);
        const size_t bytes = T.numel() * sizeof(float);
        auto it = t.find(name);
        if (it != t.end()) {
            (void)hipFree(it->second.p);
            t.erase(it);
        }
        HIPCHK(hipMalloc((void**)&T.p, bytes ? bytes : 4));
        HIPCHK(hipMemcpyAsync(T.p, dev, bytes, hipMemcpyDeviceToDevice, s));
        t[name] = T;
        return F5_OK;
    }
    const Tensor* get(const std::string& n) const {
        auto it = t.find(n);
        return it == t.end() ? nullptr : &it->second;
    }
};

// per-launch HIP-event profiler (off by default): one (start, stop) event pair per bracket on the launch stream
struct Prof {
    enum { MAXEV = 65536 };
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    std::vector<double> flops;
    int used = 0;
    ~Prof() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    void clear() {
        used = 0;
        cls.clear();
        flops.clear();
    }
    void begin(int c, hipStream_t s, double fl = 0.0) {
        if (!on || used + 2 > MAXEV) return;
        while ((int)ev.size() < used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev.push_back(e);
        }
        (void)hipEventRecord(ev[used], s);
        cls.push_back(c);
        flops.push_back(fl);
    }
    void end(hipStream_t s) {
        if (!on || used + 2 > MAXEV || (int)ev.size() < used + 2 || (int)cls.size() * 2 != used + 2) return;
        (void)hipEventRecord(ev[used + 1], s);
        used += 2;
    }
};
enum { PC_GEMM = 0, PC_ATTN = 1, PC_LN = 2, PC_CONV = 3, PC_MISC = 4, PC_TEXT = 5, PC_TIME = 6 };

// Host staging (pinned) for small per-call scalars (time grid, lengths): a ring of slots, each guarded by an event
// recorded after its last async copy, so that consecutive calls never synchronise the stream.
struct Staging {
    enum { NSLOT = 8 };
    char* host[NSLOT] = {};
    size_t cap[NSLOT] = {};
    hipEvent_t ev[NSLOT] = {};
    bool used[NSLOT] = {};
    int next = 0;
    ~Staging() {
        for (int i = 0; i < NSLOT; ++i) {
            if (host[i]) (void)hipHostFree(host[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
    int acquire(size_t bytes, char** out, int* slot) {
        const int i = next;
        next = (next + 1) % NSLOT;
        if (!ev[i]) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        if (used[i]) HIPCHK(hipEventSynchronize(ev[i]));  // the copies issued from this slot NSLOT calls ago are done
        if (bytes > cap[i]) {
            if (host[i]) (void)hipHostFree(host[i]);
            host[i] = nullptr;
            cap[i] = 0;
            HIPCHK(hipHostMalloc((void**)&host[i], std::max<size_t>(bytes, 4096), hipHostMallocDefault));
            cap[i] = std::max<size_t>(bytes, 4096);
        }
        *out = host[i];
        *slot = i;
        return F5_OK;
    }
    int release(int slot, hipStream_t s) {
        HIPCHK(hipEventRecord(ev[slot], s));
        used[slot] = true;
        return F5_OK;
    }
};

// --------------------------------------------------------------------------------------- packed weights
template <typename T> struct LinW {
    T* w = nullptr;      // [N, ldw]
    float* b = nullptr;  // [N] or null
    int N = 0, K = 0, ldw = 0;
};

template <typename T> struct BlockW {
    LinW<T> qkv, out, ff1, ff2, skip;  // skip: UNetT concat projection (no bias)
    float* norm1_g = nullptr;          // UNetT RMSNorm gains
    float* norm2_g = nullptr;
    float *qn = nullptr, *kn = nullptr;   // F5_OPT_QK_RMSNORM: gains of the RMSNorm on q / k [64]
};

struct TextBlockW {
    float *dwk = nullptr, *dwb = nullptr, *lnw = nullptr, *lnb = nullptr, *gamma = nullptr, *beta = nullptr;
    LinW<float> pw1, pw2;
};

template <typename T> struct Packed {
    LinW<float> time0, time2, mod;  // mod: stacked AdaLN linears [(6*depth+2)*D, D]
    float* E = nullptr;             // text embedding table
    std::vector<TextBlockW> tblocks;
    LinW<T> in_proj;
    T* conv_w[2] = {nullptr, nullptr};
    float* conv_b[2] = {nullptr, nullptr};
    int conv_kp = 0;
    std::vector<BlockW<T>> blocks;
    LinW<T> proj_out;
    LinW<T> long_skip;             // F5_OPT_LONG_SKIP: Linear(2 D -> D, no bias) over [x | residual] (dit.py:205,323-324)
    LinW<float> long_skip_f;       // ... and its split-planar f32 copy for F5_PREC_F16P (the operand is the raw stream)
    // F5_PREC_F16P: split-planar f32 copies of the input / output layers' weights (gemm2.h MODE 3 / 5, convpos.h SPLIT)
    LinW<float> in_proj_f, proj_out_f;
    float* conv_w_f[2] = {nullptr, nullptr};
    int conv_kp_f = 0;
    float* norm_out_g = nullptr;  // UNetT
    // aux tables
    float *rope_cos = nullptr, *rope_sin = nullptr, *time_freqs = nullptr, *text_pos = nullptr;
    float* rope_frag = nullptr;   // rope_cos / rope_sin in the QKV epilogue's fragment order (rope_frag_kernel)
    int text_pos_rows = 0;
};

struct f5_engine {
    f5_config cfg{};
    int inner = 0, kin = 0, kin_pad = 0, modN = 0;
    bool io_split = false;     // F5_PREC_F16P: the f16 engine with its input / output layers as split-f16 products on f32 operands
    bool split16 = false;      // F5_PREC_F16X3: the f32 engine with the backbone GEMMs on the f16 pipe (gemm2.h MODE 3)
    // Diagnostic (F5_X3_ABLATE=<bitmask>, F5_PREC_F16X3 only; tools/x3_ablate.py): contraction classes run with plain f16 products
    // (hi x hi only) instead of the three split products: 1 QKV, 2 attention K Q^T, 4 attention V^T P^T, 8 out-proj, 16 FF1, 32 FF2,
    // 64 input projection, 128 conv position embedding, 256 output projection
    int x3_ablate = 0;
    // F5_PREC_F16X3 attention: both products (K Q^T, V^T P^T) as PLAIN f16 products on the split operands' hi halves -- measured harmless
    // (tools/x3_ablate.py: DiT C2 1.10e-5 -> 1.01e-5, E2-TTS UNetT 1.7e-5 -> 5.2e-5; every GEMM class costs ~1e-3 there) and 7-9 % of
    // the f16x3 step.  F5_X3_ATTN_SPLIT=1 restores the three split products (and lets F5_X3_ABLATE bits 2 / 4 select).
    int x3_attn_hi = 3;
    WeightStore ws;
    std::vector<void*> owned;  // packed buffers
    Packed<float> pf;
    Packed<bf16_t> pb;
    Packed<f16_t> ph;
    bool finalized = false;
    Arena arena;
    Staging stage;
    Prof prof;
    int res_B = 0, res_N = 0, res_S = 0;
    // The unconditional text embedding (all filler tokens) depends only on the weights and the length: cache it across
    // sample() calls for the single-utterance case (the reference recomputes it every call, dit.py:244-269).
    // The buffer lives in the arena (Work::uc), so its lifetime and the invalidation of captured graphs follow every
    // other captured pointer; uc_N = -1 whenever the arena moves or the weights change.
    int uc_N = -1;
    // HIP graphs of whole sample() bodies (one per problem signature): ~2600 launches per utterance become one
    // hipGraphLaunch, so the host (shared, sometimes slow) can never be the bottleneck of the ODE loop.
    struct GraphEntry { std::string key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    std::vector<std::string> warm;      // signatures (without cache state) that have run eagerly once
    hipStream_t cap_stream = nullptr;
    int graphs_on = -1;                 // -1: read F5_HIP_GRAPH from the environment on first use
    // The conditional and unconditional halves of a CFG forward are independent until the Euler update: run them as two
    // concurrent kernel chains (second stream, fork/join events) so that one chain's launch ramps / drains overlap
    // the other chain's main loops.  Opt-in with F5_SPLIT_CFG=1 (see split_cfg_enabled).
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int split_cfg = -1;
    int res_nt = 0;
    int cur_chunk = 0;   // utterances per backbone call of the sample() in progress (chunk_utts, decided once per call)
    // packed variable-length batches (RowPack): per chunk of the last upload, rows present and sum of squared lengths
    // (host copies, used for the profiler's FLOP counts only) and the switch (F5_PACK_ROWS, default on)
    std::vector<double> pack_rows_host, pack_sq_host;
    int pack_rows = -1;
    void clear_graphs() {
        for (auto& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
        graphs.clear();
        warm.clear();
    }
    ~f5_engine() {
        clear_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (void* p : owned) (void)hipFree(p);
    }
};

template <typename U> static int dev_alloc(f5_engine* e, U** out, size_t n) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n * sizeof(U), 16)));
    e->owned.push_back(p);
    *out = reinterpret_cast<U*>(p);
    return F5_OK;
}
// ---------------------------------------------------------------------------------------------- workspace
template <typename T> struct Work {
    // per call
    float *tdev, *feat, *th, *temb, *st, *mod;
    int* lens;        // per-sample lengths, chunk-major: for each chunk of Bc utterances [Bc values][the same Bc values]
    int* lens_plain;  // the same lengths once, in utterance order (text encoder)
    int* row_start;   // RowPack: per chunk [2 Bc + 1] first packed row of every batch row (cond half, uncond half), last = rows
    int2* rowmap;     // RowPack: per chunk, packed row -> (batch row, position)
    float *step_cond, *text_c, *text_u, *tx_a, *tx_b, *tx_h1, *grn_part;
    float* uc;        // cached unconditional text embedding [res_N, text_dim] (f5_engine::uc_N)
    unsigned char* dummy;
    // per forward
    T* acat;
    float *h, *c1, *x, *pred;
    T *xn, *q, *k, *vt, *ao, *ffh;
    T* cat2;         // UNetT concat buffer [rows, 2D]
    float* skips;    // UNetT skip stack
    float* pred_all; // UNetT proj_out over N+1 tokens
    // sample(): engine-owned copies of the call's inputs / outputs so that the captured graph has stable pointers
    float *in_cond, *y, *out_buf, *traj_buf;
    unsigned char* in_mask;
    long long* in_text;
    int Npad;
};

// Row packing of a padded batch.  With attn_mask_enabled (modules.py:501-506) the frames past a sample's own length
// influence nothing -- the text is embedded per sample (dit.py:247-258), the conv position embedding masks them
// (modules.py:187-192), they are masked as attention keys and zeroed as attention queries (modules.py:540-542) and every
// other op is row-wise -- so the backbone runs on the valid rows only, the engine-side equivalent of the reference's
// unpad_input + flash_attn_varlen_func (modules.py:510-531) and of the TRT runtime's remove_input_padding
// (f5_tts_trtllm.py:448-456).  Batch row b' owns rows row_start[b'] .. row_start[b'+1]-1 (its length rounded up to a
// multiple of 4, so that the transposed V^T stores stay 8-byte aligned); q / k / V^T keep their padded per-batch-row layout,
// so the attention kernel is unchanged apart from where it writes its output rows.  All launch geometry stays that of
// the PADDED batch and every kernel reads the row count from row_start[Bp] on the device: a captured graph does not
// depend on the lengths.  ODE state, cond and the trajectory stay padded; frames past a sample's length keep their
// initial value there (the reference lets them drift: nobody reads them).
struct RowPack {
    const int* row_start = nullptr;   // device int[Bp + 1]
    const int2* rowmap = nullptr;     // device int2[rows]
    const int* rows_dev = nullptr;    // = row_start + Bp
    double rows_host = 0, sq_host = 0;
    explicit operator bool() const { return row_start != nullptr; }
};

// ------------------------------------------------------------------------------------ shared host helpers (engine.hip)
int ensure_arena(f5_engine* e, int B, int N, int S);
int chunk_utts(f5_engine* e, int B, int N, bool use_cfg, const int32_t* lens_host = nullptr);
bool split_cfg_enabled(f5_engine* e);
bool graphs_enabled(f5_engine* e);
bool pack_rows_enabled(f5_engine* e);

// Entry points of one operand precision T (float: exact-f32 MFMA; bf16_t / f16_t: 16-bit MFMA operands, f32 accumulate).
template <typename T> struct EngineOps {
    static int finalize(f5_engine* e, hipStream_t s);
    static size_t plan_bytes(const f5_engine* e, int B, int N, int S);
    static int text_embed(f5_engine* e, const int64_t* text, int B, int nt, const int32_t* lens_host, int N, int drop_text,
                          float* out, hipStream_t s);
    static int forward(f5_engine* e, const float* x, const float* cond, const int64_t* text, int nt, const float* time_host,
                       const int32_t* lens_host, int B, int N, int cfg_infer, int drop_audio_cond, int drop_text, float* out,
                       hipStream_t s);
    static int sample(f5_engine* e, const float* cond, int cond_frames, const uint8_t* cond_mask, const float* y0, const int64_t* text, int nt,
                      const float* t_host, int steps, float cfg_strength, const int32_t* lens_host, int B, int N, float* out,
                      float* traj, hipStream_t s);
};
extern template struct EngineOps<float>;
extern template struct EngineOps<bf16_t>;
extern template struct EngineOps<f16_t>;
