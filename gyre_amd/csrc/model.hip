// libgyre_hip model runtime: weight store, workspace arena, UNet2DCondition / AutoencoderKL
// forward graphs over the hand-written kernels, and the C ABI declared in include/gyre_hip.h.
//
// Topology restates the third-party modules the reference drives
//   gyre/pipeline/unet/core.py:262-274       unet(latents, t, encoder_hidden_states=...).sample
//   gyre/pipeline/unified_pipeline.py:309    vae.encode(image).latent_dist
//   gyre/pipeline/unified_pipeline.py:1531   vae.decode(latents).sample
// with hyper-parameters from gyre/ldm_config/v1-inference.yaml:29-64; weights are addressed by
// their diffusers state-dict keys (gyre/manager.py:1068-1112, gyre/ckpt_utils.py:259-285).
#include "model_impl.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local int64_t g_launches = 0;
void gyre_set_error(const std::string& msg) { g_err = msg; }
int64_t& gyre_launch_counter() { return g_launches; }
extern "C" const char* gyre_last_error(void) { return g_err.c_str(); }
extern "C" int gyre_abi_version(void) { return GYRE_ABI_VERSION; }
extern "C" int gyre_storage_dtype(void) { return GYRE_STORAGE_DTYPE; }      // GYRE_BF16 (libgyre_hip.so) or GYRE_F16 (libgyre_hip_f16.so)
extern "C" int64_t gyre_last_launch_count(void) { return g_launches; }

// ------------------------------------------------------------------------------------------
// per-launch timing
// ------------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int kclass; double flops, bytes; };
struct ProfState {
    unsigned long long mask = 0;  // bit per GyreKernelClass (KC_COUNT <= 64)
    std::vector<ProfRec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
};
thread_local ProfState g_prof;
static_assert(KC_COUNT <= 64, "profiler mask is 64 bits wide");
}  // namespace
GyreProfScope::GyreProfScope(int kclass, hipStream_t st, double flops, double bytes) : st_(st) {
    if (!(g_prof.mask >> kclass & 1ull)) return;
    hipEvent_t a, b;
    if (!g_prof.pool.empty()) { a = g_prof.pool.back().first; b = g_prof.pool.back().second; g_prof.pool.pop_back(); }
    else { if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return; }
    (void)hipEventRecord(a, st);
    slot = (int)g_prof.recs.size();
    g_prof.recs.push_back({a, b, kclass, flops, bytes});
}
GyreProfScope::~GyreProfScope() { stop(); }
void GyreProfScope::stop() {
    if (slot >= 0) (void)hipEventRecord(g_prof.recs[slot].b, st_);
    slot = -1;
}
static const char* kclass_name(int k) {
    static const char* n[KC_COUNT] = {
        "k_gemm<128, 128, 2, 2, 1", "k_gemm<256, 64, 4, 1, 1", "k_gemm<64, 64, 2, 2, 1",
        "k_gemm<128, 128, 2, 2, 0", "k_gemm<256, 64, 4, 1, 0", "k_gemm<64, 64, 2, 2, 0",
        "k_gemm8<256, 320, 4, 2, 1", "k_gemm8<128, 320, 2, 4, 1", "k_gemm8<256, 256, 4, 2, 1", "k_gemm8<128, 256, 2, 4, 1",
        "k_gemm8<256, 320, 4, 2, 0", "k_gemm8<128, 320, 2, 4, 0", "k_gemm8<256, 256, 4, 2, 0", "k_gemm8<128, 256, 2, 4, 0",
        "k_gemm8<128, 160, 4, 2, 1", "k_gemm8<256, 128, 4, 2, 1", "k_conv_out", "(unused 3)", "k_gemm8<128, 160, 4, 2, 0",
        "k_gemm4s<192, 320, 2, 2, 1", "k_gemm4s<192, 320, 2, 2, 0", "k_gemm4s<256, 256, 2, 2, 1", "k_gemm4s<256, 256, 2, 2, 0",
        "k_gemm4s<128, 320, 2, 2, 1", "k_gemm4s<128, 320, 2, 2, 0", "k_gemm4s<128, 256, 2, 2, 1", "k_gemm4s<128, 256, 2, 2, 0", "k_gemm4s<256, 320, 4, 2, 1", "k_gemm4s<256, 320, 4, 2, 0",
        "k_attn", "k_gn_partial+k_gn_finalize", "k_gn_apply", "k_layernorm", "other", "k_attn_bwd", "k_gn_bwd+k_ln_bwd",
        "k_splitk_reduce", "k_gemm_ar", "k_gemm_sm", "k_xattn"};
    return (k >= 0 && k < KC_COUNT) ? n[k] : "?";
}


// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int gyre_unet_create(const gyre_unet_cfg* cfg, int device, gyre_unet** out) {
    if (!cfg || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GYRE_HIP_CHECK(hipSetDevice(device));
    auto* h = new gyre_unet();
    h->cfg = *cfg; h->store.device = device;
    int rc = h->build();
    if (rc) { delete h; return rc; }
    GYRE_HIP_CHECK(hipDeviceSynchronize());  // creation only: zero-fills of the weight buffers are complete
    *out = h;
    return 0;
}
void gyre_unet_destroy(gyre_unet* h) { delete h; }
int gyre_unet_num_params(const gyre_unet* h) { return h ? (int)h->store.params.size() : 0; }
const char* gyre_unet_param_key(const gyre_unet* h, int i) {
    return (h && i >= 0 && i < (int)h->store.params.size()) ? h->store.params[i]->key.c_str() : nullptr;
}
int gyre_unet_set_weight(gyre_unet* h, const char* key, const void* p, int dtype, const int64_t* shape, int ndim, void* st) {
    if (!h || !key || !p || !shape) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    h->finalized = false;
    h->invalidate_contexts();
    return h->store.set_weight(key, p, dtype, shape, ndim, (hipStream_t)st);
}
int gyre_unet_finalize(gyre_unet* h, void* st) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null handle");
    TRY(h->store.finalize());
    h->finalized = true;
    return 0;
}
size_t gyre_unet_workspace_bytes(gyre_unet* h, int B, int H, int W, int S) {
    if (!h) return 0;
    // the maximum over the forms a forward of this shape can take: text context passed or cached, CFG halves sharing their
    // prefix or not (gyre_unet_hint_cfg_pairs) - the hints arrive after the caller sized its workspace.
    // The module shell asks before EVERY forward: the answer is memoised per (shape, ToMe, context form, planner state of the
    // calling thread) - two to four dry passes of the whole graph otherwise - and a dry pass leaves the handle's hints alone.
    const bool cached_too = h->cur().valid && h->cur().B == B && h->cur().S == S;
    const std::array<long, 7> key{B, H, W, S, h->ex.tome_r, cached_too ? 1 : 0, gemm_planner_state()};
    auto it = h->ws_memo.find(key);
    if (it != h->ws_memo.end()) return it->second;
    const bool hu = h->hint_uniform_t, hp = h->hint_cfg_pairs;
    size_t peak = 0;
    bool ok = true;
    for (int cached = 0; ok && cached <= (cached_too ? 1 : 0); ++cached)
        for (int pairs = 0; ok && pairs <= 1; ++pairs) {
            if (h->run(true, nullptr, nullptr, 0, nullptr, nullptr, 0, B, H, W, S, nullptr, 0, nullptr, 0, nullptr, cached != 0, nullptr, 0, 0,
                       nullptr, nullptr, 0, pairs))
                ok = false;
            else
                peak = std::max(peak, h->ex.arena.peak);
        }
    h->hint_uniform_t = hu; h->hint_cfg_pairs = hp;     // (a pending input-gradient pair IS dropped by a dry pass: it shares the executor)
    if (!ok) return 0;
    if (h->ws_memo.size() > 64) h->ws_memo.clear();
    h->ws_memo[key] = peak;
    return peak;
}
int gyre_unet_set_tome(gyre_unet* h, int r) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null handle");
    if (r < 0) GYRE_FAIL(GYRE_ERR_INVALID, "tome: r must be >= 0");
    h->ex.tome_r = r;
    return 0;
}
int gyre_unet_set_tiling(gyre_unet* h, int mode) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 3) GYRE_FAIL(GYRE_ERR_INVALID, "tiling: 0 = off, 1 = x, 2 = y, 3 = both");
    h->ex.tiling = mode;
    h->ws_memo.clear();
    return 0;
}
int gyre_vae_set_tiling(gyre_vae* h, int mode) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null handle");
    if (mode < 0 || mode > 3) GYRE_FAIL(GYRE_ERR_INVALID, "tiling: 0 = off, 1 = x, 2 = y, 3 = both");
    h->ex.tiling = mode;
    return 0;
}
size_t gyre_op_groupnorm_bwd_workspace(int B, int HW, int C, int groups) { return gn_bwd_workspace_bytes(B, HW, C, groups); }
int gyre_op_groupnorm_bwd(void* st, const void* x, const void* x2, int C1, int B, int HW, int C, int groups, const float* gamma,
                          const float* beta, float eps, int silu, const void* dy, const void* addend, void* ws, size_t wsb,
                          void* dx, void* dx2) {
    if (!x || !dy || !ws || !dx || (C1 < C && (!x2 || !dx2))) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (wsb < gn_bwd_workspace_bytes(B, HW, C, groups)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "groupnorm_bwd: workspace too small");
    GnBwdParams p;
    p.x = (const bf16_t*)x; p.x2 = C1 < C ? (const bf16_t*)x2 : nullptr; p.C1 = C1; p.B = B; p.HW = HW; p.C = C; p.G = groups;
    p.gamma = gamma; p.beta = beta; p.eps = eps; p.silu = silu; p.dy = (const bf16_t*)dy; p.addend = (const bf16_t*)addend;
    p.dx = (bf16_t*)dx; p.dx2 = (bf16_t*)dx2;
    return launch_groupnorm_bwd((hipStream_t)st, p, ws);
}
int gyre_op_layernorm_bwd(void* st, const void* x, const void* dy, int M, int C, const float* gamma, float eps, const void* addend,
                          void* dx) {
    return launch_layernorm_bwd((hipStream_t)st, (const bf16_t*)x, (const bf16_t*)dy, M, C, gamma, eps, (const bf16_t*)addend,
                                (bf16_t*)dx);
}
int gyre_op_geglu_bwd(void* st, const void* pre, const void* dy, int M, int F, void* dpre) {
    return launch_geglu_bwd((hipStream_t)st, (const bf16_t*)pre, (const bf16_t*)dy, (size_t)M, F, (bf16_t*)dpre);
}
static inline size_t al256_(size_t v) { return (v + 255) & ~(size_t)255; }
size_t gyre_op_attention_bwd_workspace(int B, int heads, int Nq, int Nk, int D) {
    const size_t C = (size_t)heads * D, lk = (size_t)(Nk + 31) / 32 * 32, lq = (size_t)(Nq + 31) / 32 * 32;
    return attn_bwd_stats_bytes(B, heads, Nq) + al256_(B * C * lk * 2) + 2 * al256_(B * C * lq * 2);
}
int gyre_op_attention_bwd(void* st_, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                          const void* d_o, int lddo, int B, int heads, int Nq, int Nk, int D, int k_prescaled, void* ws, size_t wsb,
                          void* dq, int lddq, void* dk, int lddk, void* dv, int lddv) {
    if (!q || !k || !v || !o || !d_o || !ws || !dq || ((dk == nullptr) != (dv == nullptr))) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (wsb < gyre_op_attention_bwd_workspace(B, heads, Nq, Nk, D)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "attention_bwd: workspace too small");
    hipStream_t st = (hipStream_t)st_;
    const int C = heads * D, lk = (Nk + 31) / 32 * 32, lq = (Nq + 31) / 32 * 32;
    char* w = (char*)ws;
    AttnBwdParams a{};
    a.stats = w; w += attn_bwd_stats_bytes(B, heads, Nq);
    bf16_t* kt = (bf16_t*)w; w += al256_((size_t)B * C * lk * 2);
    bf16_t* qt = (bf16_t*)w; w += al256_((size_t)B * C * lq * 2);
    bf16_t* dot = (bf16_t*)w;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
    a.o = (const bf16_t*)o; a.ldo = ldo; a.d_o = (const bf16_t*)d_o; a.lddo = lddo;
    a.kt = kt; a.ldkt = lk; a.qt = qt; a.d_ot = dot; a.ldqt = lq;
    a.dq = (bf16_t*)dq; a.lddq = lddq; a.dk = (bf16_t*)dk; a.lddk = lddk; a.dv = (bf16_t*)dv; a.lddv = lddv;
    a.B = B; a.H = heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.k_prescaled = k_prescaled;
    const bool need_t = attn_bwd_needs_transposes(D);
    if (need_t) TRY(launch_transpose(st, a.k, ldk, Nk, C, kt, lk, B, (size_t)Nk * ldk, (size_t)C * lk));
    if (dk && need_t) {
        TRY(launch_transpose(st, a.q, ldq, Nq, C, qt, lq, B, (size_t)Nq * ldq, (size_t)C * lq));
        TRY(launch_transpose(st, a.d_o, lddo, Nq, C, dot, lq, B, (size_t)Nq * lddo, (size_t)C * lq));
    }
    return launch_attention_bwd(st, a);
}
size_t gyre_op_tome_workspace(int B, int N, int C) { return tome_workspace_bytes(B, N, C); }
int gyre_op_tome_merge(void* st, const void* k, int ldk, const void* v, int ldv, int B, int N, int C, int r, void* ws, size_t wsb,
                       void* k_out, void* vt_out, int ldvt, int32_t* order_out, int32_t* node_idx_out) {
    if (!k || !v || !ws || !k_out || !vt_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    TomeParams p;
    p.k = (const bf16_t*)k; p.ldk = ldk; p.v = (const bf16_t*)v; p.ldv = ldv; p.B = B; p.N = N; p.C = C; p.r = r;
    p.k_out = (bf16_t*)k_out; p.vt_out = (bf16_t*)vt_out; p.ldvt = ldvt; p.ws = ws; p.ws_bytes = wsb;
    p.order_out = order_out; p.node_idx_out = node_idx_out;
    return launch_tome_merge((hipStream_t)st, p);
}

int gyre_unet_set_context(gyre_unet* h, void* st, const void* ctx, int cdt, int B, int S) {
    if (!h || !ctx) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    if (cdt < 0 || cdt > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    return h->set_context((hipStream_t)st, ctx, cdt, B, S, 0);
}
int gyre_unet_set_context_slot(gyre_unet* h, void* st, const void* ctx, int cdt, int B, int S, int slot) {
    if (!h || !ctx) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    if (cdt < 0 || cdt > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    return h->set_context((hipStream_t)st, ctx, cdt, B, S, slot);
}
int gyre_unet_select_context(gyre_unet* h, int slot) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (slot < 0 || slot >= GYRE_CTX_SLOTS) GYRE_FAIL(GYRE_ERR_INVALID, "unet: context slot out of range");
    if (!h->ctx_slots[slot].valid) GYRE_FAIL(GYRE_ERR_INVALID, "unet: context slot holds no projected context (set_weight invalidates all)");
    h->cur_slot = slot;
    return 0;
}
int gyre_unet_debug_tap(gyre_unet* h, const char* name, float* out_nchw_f32, size_t out_bytes) {
    if (!h || !name || !out_nchw_f32) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    h->taps[name] = {out_nchw_f32, out_bytes};
    return 0;
}
int gyre_unet_forward_ex(gyre_unet* h, void* st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B,
                         int H, int W, int S, void* ws, size_t wsb, void* out, int odt, const float* temb_add) {
    if (!h || !x || !t || !ws || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    if (xdt < 0 || xdt > 2 || cdt < 0 || cdt > 2 || odt < 0 || odt > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return h->run(false, (hipStream_t)st, x, xdt, t, ctx, cdt, B, H, W, S, ws, wsb, out, odt, temb_add, ctx == nullptr);
}
int gyre_unet_hint_cfg_pairs(gyre_unet* h, int on) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    h->hint_cfg_pairs = on != 0;
    return 0;
}
int gyre_unet_hint_uniform_timestep(gyre_unet* h, int on) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    h->hint_uniform_t = on != 0;
    return 0;
}
int gyre_unet_forward_ctrl(gyre_unet* h, void* st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B,
                           int H, int W, int S, void* ws, size_t wsb, void* out, int odt, const float* temb_add,
                           const void* const* down_res, int n_down_res, int rdt, const void* mid_res,
                           const void* const* adapter_states, int n_adapter_states) {
    if (!h || !x || !t || !ws || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    for (int d : {xdt, cdt, odt, rdt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    if (n_down_res < 0 || (n_down_res > 0 && !down_res) || n_adapter_states < 0 || (n_adapter_states > 0 && !adapter_states))
        GYRE_FAIL(GYRE_ERR_INVALID, "bad residual list");
    g_launches = 0;
    return h->run(false, (hipStream_t)st, x, xdt, t, ctx, cdt, B, H, W, S, ws, wsb, out, odt, temb_add, ctx == nullptr,
                  n_down_res ? down_res : nullptr, n_down_res, rdt, mid_res, n_adapter_states ? adapter_states : nullptr,
                  n_adapter_states);
}
int gyre_unet_forward(gyre_unet* h, void* st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B,
                      int H, int W, int S, void* ws, size_t wsb, void* out, int odt) {
    return gyre_unet_forward_ex(h, st, x, xdt, t, ctx, cdt, B, H, W, S, ws, wsb, out, odt, nullptr);
}

size_t gyre_unet_vjp_workspace_bytes(gyre_unet* h, int B, int H, int W, int S) {
    if (!h) return 0;
    int rc = gyre_unet_run_vjp(*h, true, nullptr, nullptr, 0, nullptr, nullptr, 0, B, H, W, S, nullptr, 0, nullptr, 0, nullptr, 0,
                               nullptr, 0, nullptr);
    return rc ? 0 : h->ex.arena.peak;
}
int gyre_unet_vjp(gyre_unet* h, void* st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B, int H, int W,
                  int S, const void* d_eps, int ddt, void* ws, size_t wsb, void* eps_out, int odt, void* dx_out, int dxdt,
                  const float* temb_add) {
    if (!h || !x || !t || !ctx || !d_eps || !ws || !eps_out || !dx_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    if (h->ex.tiling) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "the input-gradient sweep has no circular (tiling) convolutions");
    for (int d : {xdt, cdt, ddt, odt, dxdt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return gyre_unet_run_vjp(*h, false, (hipStream_t)st, x, xdt, t, ctx, cdt, B, H, W, S, d_eps, ddt, ws, wsb, eps_out, odt, dx_out,
                             dxdt, temb_add);
}

int gyre_unet_vjp_begin(gyre_unet* h, void* st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B, int H, int W,
                        int S, void* ws, size_t wsb, void* eps_out, int odt, const float* temb_add) {
    if (!h || !x || !t || !ctx || !ws || !eps_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!h->finalized) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "gyre_unet_finalize has not succeeded");
    if (h->ex.tiling) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "the input-gradient sweep has no circular (tiling) convolutions");
    for (int d : {xdt, cdt, odt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return gyre_unet_vjp_forward(*h, false, (hipStream_t)st, x, xdt, t, ctx, cdt, B, H, W, S, ws, wsb, eps_out, odt, temb_add);
}
int gyre_unet_vjp_pending(gyre_unet* h) { return h && h->vjp.valid ? 1 : 0; }
int gyre_unet_vjp_finish(gyre_unet* h, void* st, const void* d_eps, int ddt, void* dx_out, int dxdt) {
    if (!h || !d_eps || !dx_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    for (int d : {ddt, dxdt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return gyre_unet_vjp_reverse(*h, (hipStream_t)st, d_eps, ddt, dx_out, dxdt, 0, h->vjp.B);
}
int gyre_unet_vjp_finish_range(gyre_unet* h, void* st, const void* d_eps, int ddt, void* dx_out, int dxdt, int b0, int nb) {
    if (!h || !d_eps || !dx_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    for (int d : {ddt, dxdt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    if (h->vjp.valid && (b0 < 0 || nb < 1 || b0 + nb > h->vjp.B)) GYRE_FAIL(GYRE_ERR_INVALID, "vjp_finish_range: samples outside the pending batch");
    g_launches = 0;
    return gyre_unet_vjp_reverse(*h, (hipStream_t)st, d_eps, ddt, dx_out, dxdt, b0, nb);
}

int gyre_vae_create(const gyre_vae_cfg* cfg, int device, gyre_vae** out) {
    if (!cfg || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GYRE_HIP_CHECK(hipSetDevice(device));
    auto* h = new gyre_vae();
    h->cfg = *cfg; h->store.device = device;
    int rc = h->build();
    if (rc) { delete h; return rc; }
    GYRE_HIP_CHECK(hipDeviceSynchronize());
    *out = h;
    return 0;
}
void gyre_vae_destroy(gyre_vae* h) { delete h; }
int gyre_vae_num_params(const gyre_vae* h) { return h ? (int)h->store.params.size() : 0; }
const char* gyre_vae_param_key(const gyre_vae* h, int i) {
    return (h && i >= 0 && i < (int)h->store.params.size()) ? h->store.params[i]->key.c_str() : nullptr;
}
int gyre_vae_set_weight(gyre_vae* h, const char* key, const void* p, int dtype, const int64_t* shape, int ndim, void* st) {
    if (!h || !key || !p || !shape) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return h->store.set_weight(key, p, dtype, shape, ndim, (hipStream_t)st);
}
// encoder-only / decoder-only use is allowed: finalize checks everything, the run paths only need their half
int gyre_vae_finalize(gyre_vae* h, void* st) {
    if (!h) GYRE_FAIL(GYRE_ERR_INVALID, "null handle");
    return h->store.finalize();
}
size_t gyre_vae_workspace_bytes(gyre_vae* h, int B, int H, int W, int decode) {
    if (!h) return 0;
    int rc = decode ? h->run_decode(true, nullptr, nullptr, 0, B, H, W, nullptr, 0, nullptr, 0)
                    : h->run_encode(true, nullptr, nullptr, 0, B, H, W, nullptr, 0, nullptr, 0);
    return rc ? 0 : h->ex.arena.peak;
}
int gyre_vae_encode(gyre_vae* h, void* st, const void* img, int idt, int B, int H, int W, void* ws, size_t wsb,
                    void* out, int odt) {
    if (!h || !img || !ws || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (idt < 0 || idt > 2 || odt < 0 || odt > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return h->run_encode(false, (hipStream_t)st, img, idt, B, H, W, ws, wsb, out, odt);
}
int gyre_vae_decode(gyre_vae* h, void* st, const void* z, int idt, int B, int hl, int wl, void* ws, size_t wsb,
                    void* out, int odt) {
    if (!h || !z || !ws || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (idt < 0 || idt > 2 || odt < 0 || odt > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return h->run_decode(false, (hipStream_t)st, z, idt, B, hl, wl, ws, wsb, out, odt);
}

// ---- per-launch timing (bench.py roofline leg) -------------------------------------------------
size_t gyre_vae_decode_vjp_workspace_bytes(gyre_vae* h, int B, int hl, int wl) {
    if (!h) return 0;
    int rc = gyre_vae_run_decode_vjp(*h, true, nullptr, nullptr, 0, B, hl, wl, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0);
    return rc ? 0 : h->ex.arena.peak;
}
int gyre_vae_decode_vjp(gyre_vae* h, void* st, const void* z, int zdt, int B, int hl, int wl, const void* d_img, int ddt, void* ws,
                        size_t wsb, void* img_out, int odt, void* dz_out, int dzdt) {
    if (!h || !z || !d_img || !ws || !img_out || !dz_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (h->ex.tiling) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "the input-gradient sweep has no circular (tiling) convolutions");
    for (int d : {zdt, ddt, odt, dzdt}) if (d < 0 || d > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
    g_launches = 0;
    return gyre_vae_run_decode_vjp(*h, false, (hipStream_t)st, z, zdt, B, hl, wl, d_img, ddt, ws, wsb, img_out, odt, dz_out, dzdt);
}

int gyre_prof_set_mask(unsigned long long mask) { g_prof.mask = mask; return 0; }
int gyre_prof_num_classes(void) { return KC_COUNT; }
const char* gyre_prof_class_name(int k) { return kclass_name(k); }
// Collects (and clears) the records of this thread: per class launches, total ms, algorithmic flops / bytes.
// The caller must have synchronised the stream.  Arrays have KC_COUNT entries.
int gyre_prof_collect(int64_t* launches, double* ms, double* flops, double* bytes) {
    for (int k = 0; k < KC_COUNT; ++k) { launches[k] = 0; ms[k] = 0; flops[k] = 0; bytes[k] = 0; }
    static const bool dump = getenv("GYRE_PROF_DUMP") != nullptr;     // dev aid: one stderr line per timed launch
    for (auto& r : g_prof.recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            launches[r.kclass]++; ms[r.kclass] += t; flops[r.kclass] += r.flops; bytes[r.kclass] += r.bytes;
            if (dump) fprintf(stderr, "GYRE_PROF %s | %.0f MFLOP | %.0f KB | %.2f us\n", kclass_name(r.kclass), r.flops / 1e6, r.bytes / 1e3, t * 1e3);
        }
        g_prof.pool.emplace_back(r.a, r.b);
    }
    g_prof.recs.clear();
    return 0;
}

int gyre_set_batch_invariant(int n) { return gemm_set_batch_invariant(n); }
int gyre_get_batch_invariant(void) { return gemm_get_batch_invariant(); }

// ---- single operators -----------------------------------------------------------------------
size_t gyre_op_groupnorm_workspace(int B, int HW, int C, int groups) { return gn_workspace_bytes(B, HW, C, groups); }
int gyre_op_groupnorm(void* st, const void* x, const void* x2, int C1, int B, int HW, int C, int groups,
                      const float* gamma, const float* beta, float eps, int silu, void* ws, size_t wsb, void* y) {
    if (!x || !gamma || !beta || !ws || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (wsb < gn_workspace_bytes(B, HW, C, groups)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "groupnorm workspace too small");
    GnParams p;
    p.x = (const bf16_t*)x; p.x2 = x2 ? (const bf16_t*)x2 : (const bf16_t*)x; p.C1 = x2 ? C1 : C;
    p.B = B; p.HW = HW; p.C = C; p.G = groups; p.gamma = gamma; p.beta = beta; p.eps = eps; p.silu = silu;
    p.nchunks = gn_pick_chunks(B, HW, C);
    p.partial = (float*)ws;
    p.scale_shift = (float*)((char*)ws + align_up((size_t)B * p.nchunks * groups * 2 * sizeof(float), 256));
    p.y = (bf16_t*)y;
    if (gn_use_small(HW, C, p.C1, groups)) return launch_groupnorm_small((hipStream_t)st, p);
    TRY(launch_groupnorm_stats((hipStream_t)st, p));
    return launch_groupnorm_apply((hipStream_t)st, p);
}
// GroupNorm whose statistics come from the producers of x / x2 (gyre_op_conv3x3_colstats / gyre_op_linear_colstats): no
// statistics pass, one launch.
int gyre_op_groupnorm_colstats(void* st, const void* x, const void* x2, int C1, int B, int HW, int C, int groups,
                               const float* gamma, const float* beta, float eps, int silu, const float* cs_x, int cs_x_chunks,
                               const float* cs_x2, int cs_x2_chunks, int unit, void* ws, size_t wsb, void* y) {
    if (!x || !gamma || !beta || !ws || !y || !cs_x) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (wsb < gn_workspace_bytes(B, HW, C, groups)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "groupnorm workspace too small");
    GnParams p;
    p.x = (const bf16_t*)x; p.x2 = x2 ? (const bf16_t*)x2 : (const bf16_t*)x; p.C1 = x2 ? C1 : C;
    p.B = B; p.HW = HW; p.C = C; p.G = groups; p.gamma = gamma; p.beta = beta; p.eps = eps; p.silu = silu;
    p.nchunks = gn_pick_chunks(B, HW, C);
    p.partial = (float*)ws;
    p.scale_shift = (float*)((char*)ws + align_up((size_t)B * p.nchunks * groups * 2 * sizeof(float), 256));
    p.y = (bf16_t*)y;
    p.cs_x = cs_x; p.cs_x_chunks = cs_x_chunks; p.cs_x2 = x2 ? cs_x2 : nullptr; p.cs_x2_chunks = x2 ? cs_x2_chunks : 0; p.cs_unit = unit;
    if (!gn_accepts_colstats(p)) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "groupnorm: this shape does not take producer statistics");
    return launch_groupnorm_apply((hipStream_t)st, p);
}
int gyre_op_layernorm(void* st, const void* x, int M, int C, const float* g, const float* b, float eps, void* y) {
    if (!x || !g || !b || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_layernorm((hipStream_t)st, (const bf16_t*)x, M, C, g, b, eps, (bf16_t*)y);
}
int gyre_op_linear(void* st, const void* x, int M, int K, const void* w, int N, const float* bias, const void* residual,
                   int geglu, void* y) {
    if (!x || !w || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = K; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w; p.K = K;
    p.N = geglu ? 2 * N : N; p.M = M; p.bias = bias; p.residual = (const bf16_t*)residual; p.ldr = N; p.geglu = geglu;
    p.out = y; p.ldc = N; p.out_mode = OUT_BF16;
    return launch_gemm((hipStream_t)st, p);
}
int gyre_op_linear_t(void* st, const void* x, int M, int K, const void* w, int N, const float* bias, int tokens, int ldt,
                     void* y) {
    if (!x || !w || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = K; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w; p.K = K; p.N = N; p.M = M;
    p.bias = bias; p.out = y; p.out_mode = OUT_BF16_T; p.tokens_per_batch = tokens; p.ldt = ldt;
    return launch_gemm((hipStream_t)st, p);
}
int gyre_op_conv3x3(void* st, const void* x, int B, int Hi, int Wi, int Cin, const void* w, int Cout, const float* bias,
                    const void* residual, int stride, int ups, int asym, void* y) {
    if (!x || !w || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    int pad = asym ? 0 : 1;
    int Hin = ups ? 2 * Hi : Hi, Win = ups ? 2 * Wi : Wi;
    int Ho = (Hin + (pad ? 2 : 1) - 3) / stride + 1, Wo = (Win + (pad ? 2 : 1) - 3) / stride + 1;
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = Cin; p.mode = GEMM_CONV3; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
    p.stride = stride; p.pad = pad; p.ups = ups; p.W = (const bf16_t*)w; p.K = 9 * Cin; p.N = Cout; p.M = B * Ho * Wo; p.samples = B;
    p.bias = bias; p.residual = (const bf16_t*)residual; p.ldr = Cout; p.rows_per_sample = Ho * Wo;
    p.out = y; p.ldc = Cout; p.out_mode = OUT_BF16;
    return launch_gemm((hipStream_t)st, p);
}
// y = conv3x3(x; stride 1, pad 1) + (sx | sx2) w_sc^T + bias + bias_sc: a resnet's conv2 with its 1x1 shortcut folded in as extra K steps
// (GemmParams::sc_*).  sx [B][H][W][C1], sx2 [B][H][W][C2] or NULL (C2 = 0), w_sc [Cout][C1 + C2].  ws: Cout * (9 Cin + C1 + C2) * 2 + Cout * 4
// bytes (+ 512 for alignment).  GYRE_ERR_UNSUPPORTED where the planner's kernel for the shape does not know the form.
int gyre_op_conv3x3_shortcut(void* st, const void* x, int B, int H, int W, int Cin, const void* w, int Cout, const float* bias,
                             const void* sx, int C1, const void* sx2, int C2, const void* w_sc, const float* bias_sc, void* ws,
                             size_t ws_bytes, void* y) {
    if (!x || !w || !y || !sx || !w_sc || !ws || (C2 > 0 && !sx2)) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = Cin; p.mode = GEMM_CONV3; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Ho = H; p.Wo = W;
    p.stride = 1; p.pad = 1; p.N = Cout; p.M = B * H * W; p.samples = B; p.rows_per_sample = H * W;
    p.out = y; p.ldc = Cout; p.out_mode = OUT_BF16;
    p.sc_A = (const bf16_t*)sx; p.sc_lda = C1; p.sc_C1 = C1; p.sc_K = C1 + C2;
    if (C2 > 0) { p.sc_A2 = (const bf16_t*)sx2; p.sc_lda2 = C2; }
    p.K = 9 * Cin + p.sc_K;
    if (!gemm_conv_shortcut_ok(p)) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "conv3x3_shortcut: the planner's kernel for this shape does not fold the shortcut");
    const size_t wbytes = align_up((size_t)Cout * p.K * 2, 256);
    if (ws_bytes < wbytes + align_up((size_t)Cout * 4, 256)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "conv3x3_shortcut: workspace too small");
    bf16_t* wcat = (bf16_t*)ws;
    float* bcat = (float*)((char*)ws + wbytes);
    TRY(launch_concat_rows((hipStream_t)st, (const bf16_t*)w, 9 * Cin, (const bf16_t*)w_sc, p.sc_K, Cout, wcat));
    if (hipMemsetAsync(bcat, 0, (size_t)Cout * 4, (hipStream_t)st) != hipSuccess) GYRE_FAIL(GYRE_ERR_HIP, "memset");
    if (bias) TRY(launch_add_f32((hipStream_t)st, bcat, bias, (size_t)Cout));
    if (bias_sc) TRY(launch_add_f32((hipStream_t)st, bcat, bias_sc, (size_t)Cout));
    p.W = wcat; p.bias = bcat;
    return launch_gemm((hipStream_t)st, p);
}
int gyre_op_conv3x3_nchw(void* st, const void* x, int B, int H, int W, int Cin, const void* w, int Cout, const float* bias, void* y,
                         int y_dtype, int force_tiles) {
    if (!x || !w || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (!force_tiles && conv_out_supports(Cin, Cout))
        return launch_conv_out((hipStream_t)st, (const bf16_t*)x, B, H, W, Cin, (const bf16_t*)w, bias, Cout, y, y_dtype);
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = Cin; p.mode = GEMM_CONV3; p.Hi = H; p.Wi = W; p.Cin = Cin; p.Ho = H; p.Wo = W;
    p.stride = 1; p.pad = 1; p.W = (const bf16_t*)w; p.K = 9 * Cin; p.N = Cout; p.M = B * H * W; p.samples = B;
    p.bias = bias; p.rows_per_sample = H * W; p.out = y; p.out_mode = OUT_NCHW; p.out_dtype = y_dtype;
    return launch_gemm((hipStream_t)st, p);
}
// The same operators leaving the GroupNorm statistics of their output (GemmParams::colstat_out): stats_out
// [M / rows][N / unit][2] f32 with rows = *rows_out, the row-block size of the kernel the planner picks (<= 0 and
// GYRE_ERR_UNSUPPORTED when that kernel cannot).  ws: split-K slab space (gyre_op_gemm_splitk_bytes; may be null when 0).
static int op_colstats_go(hipStream_t st, GemmParams& p, int unit, int rps, float* stats_out, size_t stats_bytes, void* ws,
                          size_t ws_bytes, int* rows_out) {
    if (unit <= 0 || rps <= 0 || p.M <= 0 || p.N <= 0 || p.M % rps || p.N % unit)
        GYRE_FAIL(GYRE_ERR_INVALID, "colstats: unit and rows per sample must be positive and divide N and M");
    p.colstat_unit = unit; p.rows_per_sample = rps;
    const int rows = gemm_colstat_rows(p);
    if (rows_out) *rows_out = rows;
    if (rows <= 0) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "colstats: the planner's kernel for this shape cannot emit column statistics");
    if (stats_bytes < (size_t)(p.M / rows) * (p.N / unit) * 2 * sizeof(float)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "colstats: stats buffer too small");
    GemmPlan pl = gemm_plan(p);
    if (pl.ws_bytes) {
        if (!ws || ws_bytes < pl.ws_bytes) GYRE_FAIL(GYRE_ERR_WORKSPACE, "colstats: split-K workspace too small");
        p.splitk_ws = (float*)ws; p.splitk_ws_bytes = ws_bytes;
    }
    p.colstat_out = stats_out;
    return launch_gemm(st, p);
}
size_t gyre_op_gemm_splitk_bytes(int conv, int M, int N, int K, int B) {
    if (M <= 0 || N <= 0 || K <= 0 || (conv && K % 9)) return 0;
    GemmParams p;
    p.mode = conv ? GEMM_CONV3 : GEMM_LINEAR; p.M = M; p.N = N; p.K = K; p.Cin = conv ? K / 9 : 0; p.lda = conv ? K / 9 : K;
    p.ldc = N; p.ldr = N; p.samples = B; p.out_mode = OUT_BF16;
    if (conv) { p.Ho = p.Wo = p.Hi = p.Wi = (int)sqrt((double)(M / (B > 0 ? B : 1))); }
    return gemm_plan(p).ws_bytes;
}
int gyre_op_conv3x3_colstats(void* st, const void* x, int B, int Hi, int Wi, int Cin, const void* w, int Cout, const float* bias,
                             const void* residual, int stride, int ups, int unit, void* y, float* stats_out, size_t stats_bytes,
                             void* ws, size_t ws_bytes, int* rows_out) {
    if (!x || !w || !y || !stats_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2) || unit <= 0)
        GYRE_FAIL(GYRE_ERR_INVALID, "conv3x3_colstats: sizes and unit must be positive, stride 1 or 2");
    const int Hin = ups ? 2 * Hi : Hi, Win = ups ? 2 * Wi : Wi;
    const int Ho = (Hin + 2 - 3) / stride + 1, Wo = (Win + 2 - 3) / stride + 1;
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = Cin; p.mode = GEMM_CONV3; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo;
    p.stride = stride; p.pad = 1; p.ups = ups; p.W = (const bf16_t*)w; p.K = 9 * Cin; p.N = Cout; p.M = B * Ho * Wo; p.samples = B;
    p.bias = bias; p.residual = (const bf16_t*)residual; p.ldr = Cout;
    p.out = y; p.ldc = Cout; p.out_mode = OUT_BF16;
    return op_colstats_go((hipStream_t)st, p, unit, Ho * Wo, stats_out, stats_bytes, ws, ws_bytes, rows_out);
}
int gyre_op_linear_colstats(void* st, const void* x, int M, int K, const void* w, int N, const float* bias, const void* residual,
                            int rows_per_sample, int unit, void* y, float* stats_out, size_t stats_bytes, void* ws, size_t ws_bytes,
                            int* rows_out) {
    if (!x || !w || !y || !stats_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (M <= 0 || K <= 0 || N <= 0 || rows_per_sample <= 0 || unit <= 0 || M % rows_per_sample || N % unit)
        GYRE_FAIL(GYRE_ERR_INVALID, "linear_colstats: M, K, N, rows_per_sample and unit must be positive, rows_per_sample must divide M and unit N");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = K; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w; p.K = K;
    p.N = N; p.M = M; p.bias = bias; p.residual = (const bf16_t*)residual; p.ldr = N; p.samples = M / rows_per_sample;
    p.out = y; p.ldc = N; p.out_mode = OUT_BF16;
    return op_colstats_go((hipStream_t)st, p, unit, rows_per_sample, stats_out, stats_bytes, ws, ws_bytes, rows_out);
}
int gyre_op_repack_conv_weight(void* st, const float* w, int Cout, int Cin, int KH, int KW, int Cin_pad, void* out) {
    if (!w || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_repack_conv((hipStream_t)st, w, 0, Cout, Cin, KH, KW, Cin_pad, (bf16_t*)out);
}
int gyre_op_repack_linear_weight(void* st, const float* w, int O, int I, int geglu, void* out) {
    if (!w || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_repack_linear((hipStream_t)st, w, 0, O, I, geglu, (bf16_t*)out);
}
int gyre_op_repack_bias(void* st, const float* b, int n, int geglu, float* out) {
    if (!b || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_cast_f32((hipStream_t)st, b, 0, (size_t)n, geglu, out);
}
int gyre_op_attention(void* st, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int B, int heads,
                      int Nq, int Nk, int D, void* o, int ldo) {
    if (!q || !k || !vt || !o) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    AttnParams a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.vt = (const bf16_t*)vt; a.ldvt = ldvt;
    a.o = (bf16_t*)o; a.ldo = ldo; a.B = B; a.H = heads; a.Nq = Nq; a.Nk = Nk; a.D = D;
    return launch_attention((hipStream_t)st, a);
}
int gyre_op_qkv(void* st, const void* x, int M, int C, const void* w_qkv, int tokens, void* qk_out, void* vt_out, int ldt) {
    if (!x || !w_qkv || !qk_out || !vt_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (tokens <= 0 || M % tokens) GYRE_FAIL(GYRE_ERR_INVALID, "M must be a multiple of tokens");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = C; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w_qkv; p.K = C; p.N = 3 * C; p.M = M;
    p.samples = M / tokens;
    p.out = qk_out; p.ldc = 2 * C; p.out_mode = OUT_BF16;
    p.vt_out = (bf16_t*)vt_out; p.vt_col0 = 2 * C; p.tokens_per_batch = tokens; p.ldt = ldt;
    return launch_gemm((hipStream_t)st, p);
}
// LayerNorm folded into the consuming GEMM (kernels.h GemmParams::ln_colsum).  qkv_tokens > 0: w holds [3K][K] rows Q | K | V,
// y receives Q | K ([M][2K]) and vt_out V^T as in gyre_op_qkv; else a plain / GEGLU linear as in gyre_op_linear.
size_t gyre_op_ln_linear_workspace(int w_rows, int K, int M) {
    return align_up((size_t)w_rows * K * 2, 256) + 2 * align_up((size_t)w_rows * 4, 256) + align_up((size_t)M * 8, 256);
}
// y = x @ w^T (+bias) (+residual) as gyre_op_linear, and per row the partial (sum, sum of squares) of the rounded outputs of
// every N tile: stats_out [parts][M][2] with parts = gyre_op_linear_rowstats_parts(M, K, N, residual != 0) (0: not available
// for this shape).  Input of gyre_op_ln_linear's row_parts.
int gyre_op_linear_rowstats_parts(int M, int K, int N, int has_residual) {
    GemmParams p;
    p.lda = K; p.mode = GEMM_LINEAR; p.K = K; p.N = N; p.M = M; p.ldr = N; p.ldc = N; p.out_mode = OUT_BF16;
    if (has_residual) p.residual = (const bf16_t*)(uintptr_t)256;      // shape query only: any aligned non-null value
    return gemm_rowstat_parts(p);
}
int gyre_op_linear_rowstats(void* st, const void* x, int M, int K, const void* w, int N, const float* bias, const void* residual,
                            void* y, float* stats_out) {
    if (!x || !w || !y || !stats_out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = K; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w; p.K = K;
    p.N = N; p.M = M; p.bias = bias; p.residual = (const bf16_t*)residual; p.ldr = N;
    p.out = y; p.ldc = N; p.out_mode = OUT_BF16;
    if (gemm_rowstat_parts(p) <= 0) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "linear_rowstats: the planner's kernel for this shape cannot emit row statistics");
    p.rowstat_out = stats_out;
    return launch_gemm((hipStream_t)st, p);
}
int gyre_op_ln_linear(void* st, const void* x, int M, int K, const float* gamma, const float* beta, float eps, const void* w,
                      int N, const float* bias, int geglu, int qkv_tokens, void* vt_out, int ldt, const float* row_parts,
                      int n_parts, void* ws, size_t ws_bytes, void* y) {
    if (!x || !w || !y || !gamma || !beta || !ws) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    GemmParams p;
    p.A = (const bf16_t*)x; p.lda = K; p.mode = GEMM_LINEAR; p.W = (const bf16_t*)w; p.K = K; p.M = M; p.bias = bias;
    p.out = y; p.out_mode = OUT_BF16;
    if (qkv_tokens > 0) {
        if (!vt_out || M % qkv_tokens || N != 3 * K || geglu) GYRE_FAIL(GYRE_ERR_INVALID, "bad fused Q|K|V arguments");
        p.N = 3 * K; p.ldc = 2 * K; p.samples = M / qkv_tokens;
        p.vt_out = (bf16_t*)vt_out; p.vt_col0 = 2 * K; p.tokens_per_batch = qkv_tokens; p.ldt = ldt;
    } else {
        p.N = geglu ? 2 * N : N; p.ldc = N; p.geglu = geglu;
    }
    if (ws_bytes < gyre_op_ln_linear_workspace(p.N, K, M)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "ln_linear: workspace too small");
    if (!gemm_ln_fusable(p)) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "ln_linear: the planner's kernel for this shape cannot fold the LayerNorm");
    bf16_t* wf = (bf16_t*)ws;
    float* cs = (float*)((char*)ws + align_up((size_t)p.N * K * 2, 256));
    float* bb = (float*)((char*)cs + align_up((size_t)p.N * 4, 256));
    TRY(launch_ln_fold((hipStream_t)st, p.W, p.N, K, gamma, beta, bias, wf, cs, bb));
    if (row_parts && n_parts > 0) {      // partial sums left by the GEMM that produced x (gyre_op_linear_rowstats)
        p.ln_parts = row_parts; p.ln_nparts = n_parts; p.ln_eps = eps;
    } else {
        float* stats = (float*)((char*)bb + align_up((size_t)p.N * 4, 256));
        TRY(launch_layernorm_stats((hipStream_t)st, p.A, M, K, eps, stats));
        p.ln_stats = stats;
    }
    p.W = wf; p.bias = bb; p.ln_colsum = cs;
    return launch_gemm((hipStream_t)st, p);
}
int gyre_op_attention_ex(void* st, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, int B,
                         int heads, int Nq, int Nk, int D, void* o, int ldo, int k_prescaled) {
    if (!q || !k || !vt || !o) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    AttnParams a;
    a.q = (const bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.vt = (const bf16_t*)vt; a.ldvt = ldvt;
    a.o = (bf16_t*)o; a.ldo = ldo; a.B = B; a.H = heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.k_prescaled = k_prescaled ? 1 : 0;
    return launch_attention((hipStream_t)st, a);
}
// The cross-attention block of a BasicTransformerBlock as the fused kernel runs it (kernels_xattn.hip):
//   out = softmax(LayerNorm(x) Wq^T . K^T) V Wo^T + bo + x, row_stats[m] = (sum, sum of squares) of the rounded out[m][:]
// x [M][C] (rows before the LayerNorm, M = B * tokens), wq / wo [C][C] repacked, k [B][Nk][C] PRESCALED by log2(e) / sqrt(C / heads),
// vt [B][C][ldvt] (V transposed, ldvt >= Nk rounded up to 8).  ws: gyre_op_ln_linear_workspace(C, C, M) bytes.  GYRE_ERR_UNSUPPORTED
// outside the kernel's domain (xattn_supports: C = 320, 8 heads, Nk <= 80, tokens % 128 == 0, M / 128 >= 256).
int gyre_op_cross_attention_block(void* st, const void* x, int M, int tokens, int C, int heads, const float* gamma, const float* beta,
                                  float eps, const void* wq, const void* k, const void* vt, int Nk, int ldvt, const void* wo,
                                  const float* bo, void* ws, size_t ws_bytes, void* out, float* row_stats) {
    if (!x || !gamma || !beta || !wq || !k || !vt || !wo || !ws || !out) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    if (tokens <= 0 || M % tokens) GYRE_FAIL(GYRE_ERR_INVALID, "cross_attention_block: M must be a multiple of tokens");
    if (!xattn_supports(C, heads, tokens, Nk, M)) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "cross_attention_block: shape outside the fused kernel's domain");
    if (ws_bytes < gyre_op_ln_linear_workspace(C, C, M)) GYRE_FAIL(GYRE_ERR_WORKSPACE, "cross_attention_block: workspace too small");
    bf16_t* wf = (bf16_t*)ws;
    float* cs = (float*)((char*)ws + align_up((size_t)C * C * 2, 256));
    float* bb = (float*)((char*)cs + align_up((size_t)C * 4, 256));
    float* stats = (float*)((char*)bb + align_up((size_t)C * 4, 256));
    TRY(launch_ln_fold((hipStream_t)st, (const bf16_t*)wq, C, C, gamma, beta, nullptr, wf, cs, bb));
    TRY(launch_layernorm_stats((hipStream_t)st, (const bf16_t*)x, M, C, eps, stats));
    XattnParams xp;
    xp.x = (const bf16_t*)x; xp.ldx = C; xp.wq = wf; xp.q_colsum = cs; xp.q_bias = bb;
    xp.ln_parts = nullptr; xp.ln_nparts = 0; xp.ln_stats = stats; xp.ln_eps = eps;
    xp.k = (const bf16_t*)k; xp.vt = (const bf16_t*)vt; xp.ldvt = ldvt; xp.wo = (const bf16_t*)wo; xp.bo = bo;
    xp.out = (bf16_t*)out; xp.ldo = C; xp.rowstat_out = row_stats; xp.M = M; xp.rows_per_sample = tokens; xp.Nk = Nk; xp.heads = heads;
    return launch_xattn((hipStream_t)st, xp, C);
}
int gyre_op_nchw_to_nhwc(void* st, const void* x, int dtype, int B, int C, int HW, int Cpad, void* y) {
    if (!x || !y) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_nchw_to_nhwc((hipStream_t)st, x, dtype, B, C, HW, Cpad, (bf16_t*)y);
}
int gyre_op_copy_probe(void* st, const void* src, void* dst, size_t bytes) {
    if (!src || !dst) GYRE_FAIL(GYRE_ERR_INVALID, "null argument");
    return launch_copy_probe((hipStream_t)st, src, dst, bytes);
}

}  // extern "C"
