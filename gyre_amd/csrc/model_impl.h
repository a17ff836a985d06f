// Model runtime internals shared by model.hip (forward graphs + C ABI) and model_vjp.hip (input-gradient graphs):
// workspace arena, weight store, per-layer weight records, the op executor and the UNet / VAE graph objects.
#pragma once
#include "../../include/gyre_hip.h"
#include "kernels.h"
#include <cstdlib>
#include <cstdio>
#include <cmath>

#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#define TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int pad8(int c) { return (c + 7) / 8 * 8; }

// ------------------------------------------------------------------------------------------
// workspace arena: first-fit free list over a caller-provided buffer.  The same call sequence
// gives the same offsets, so a dry run (no launches) yields the exact peak requirement.
// ------------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr;
    size_t cap = 0, peak = 0;
    bool dry = false;
    std::map<size_t, size_t> free_;  // offset -> size
    void reset(char* b, size_t c, bool d) {
        base = b; cap = c; dry = d; peak = 0;
        free_.clear();
        free_[0] = d ? ((size_t)1 << 60) : c;
    }
    // returns offset or (size_t)-1
    size_t alloc(size_t bytes) {
        bytes = align_up(bytes ? bytes : 1, 256);
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= bytes) {
                size_t off = it->first, rem = it->second - bytes;
                free_.erase(it);
                if (rem) free_[off + bytes] = rem;
                peak = std::max(peak, off + bytes);
                return off;
            }
        }
        return (size_t)-1;
    }
    void release(size_t off, size_t bytes) {
        bytes = align_up(bytes ? bytes : 1, 256);
        auto it = free_.emplace(off, bytes).first;
        auto nx = std::next(it);
        if (nx != free_.end() && it->first + it->second == nx->first) { it->second += nx->second; free_.erase(nx); }
        if (it != free_.begin()) {
            auto pv = std::prev(it);
            if (pv->first + pv->second == it->first) { pv->second += it->second; free_.erase(it); }
        }
    }
};

struct Tn {  // NHWC bf16 activation (or a raw byte buffer when C == 0)
    bf16_t* p = nullptr; size_t off = (size_t)-1, bytes = 0;
    int B = 0, H = 0, W = 0, C = 0;
    // GroupNorm statistics its producer left (GemmParams::colstat_out): [B][cs_chunks][C / cs_unit][2] floats in the arena,
    // released with the tensor; cs_chunks == 0: none (or invalidated by an in-place update of the tensor)
    float* cs = nullptr; size_t cs_off = (size_t)-1, cs_bytes = 0; int cs_chunks = 0, cs_unit = 0;
    int rows() const { return B * H * W; }
    bool valid() const { return off != (size_t)-1; }
};

// ------------------------------------------------------------------------------------------
// weight store
// ------------------------------------------------------------------------------------------
enum ParamKind { PK_CONV3 = 0, PK_MAT = 1, PK_VEC = 2, PK_MAT_GEGLU = 3, PK_VEC_GEGLU = 4 };
struct Param {
    std::string key;
    std::vector<int64_t> shape;  // PyTorch shape expected from the caller
    int kind = PK_VEC;
    int o_pad = 0, i_pad = 0;    // padded out / in channels of the repacked matrix
    void* dev = nullptr;         // destination (inside `owner` allocation)
    float scale = 1.f;           // folded into the values in fp32 before the bf16 rounding (attention K scale)
    bool set = false;
};

struct Store {
    int device = 0;
    std::vector<std::unique_ptr<Param>> params;
    std::unordered_map<std::string, Param*> by_key;
    std::vector<void*> allocs;
    // transposed copies of weight matrices for the input-gradient pass (model_vjp.hip), made on first use and refreshed when
    // any weight was set since (per-request LoRA re-uploads): key = forward weight pointer
    struct WtEntry { bf16_t* p = nullptr; size_t bytes = 0; uint64_t version = 0; };
    std::unordered_map<const void*, WtEntry> wt_cache;
    uint64_t weights_version = 1;
    // gamma-folded copies of the weights that follow a LayerNorm (GemmParams::ln_colsum), same lifetime rules
    struct LnEntry { bf16_t* wf = nullptr; float* cs = nullptr; float* bb = nullptr; size_t elems = 0; uint64_t version = 0; };
    std::unordered_map<const void*, LnEntry> ln_cache;
    LnEntry* ln_lookup(const void* w, int N, int K, bool* fresh) {
        LnEntry& en = ln_cache[w];
        const size_t elems = (size_t)N * K;
        if (!en.wf || en.elems < elems) {
            const size_t wbytes = (elems * 2 + 255) / 256 * 256, vbytes = ((size_t)N * 4 + 255) / 256 * 256;
            char* base = (char*)dmalloc(wbytes + 2 * vbytes, false);
            if (!base) return nullptr;
            en.wf = (bf16_t*)base; en.cs = (float*)(base + wbytes); en.bb = (float*)(base + wbytes + vbytes);
            en.elems = elems; en.version = 0;
        }
        *fresh = en.version == weights_version;
        en.version = weights_version;
        return &en;
    }
    // fragment-ordered copies of the weights the A-resident GEMM kernel reads (GemmParams::w_packed), keyed by the row-major
    // matrix they were made from (a plain weight or its LayerNorm-folded copy); same lifetime rules
    std::unordered_map<const void*, WtEntry> ar_cache;
    std::unordered_map<const void*, int> ar_kind;      // which kernel's order the copy is in (tile config 30 / 31)
    void* ar_lookup(const void* w, size_t bytes, bool* fresh, int kind = 30) {
        WtEntry& en = ar_cache[w];
        if (!en.p || en.bytes < bytes) {
            en.p = (bf16_t*)dmalloc(bytes, false);
            en.bytes = bytes; en.version = 0;
        }
        int& k = ar_kind[w];
        *fresh = en.p && en.version == weights_version && k == kind;
        if (en.p) { en.version = weights_version; k = kind; }
        return en.p;
    }
    // blocked copies of the weights the LDS-DMA tile kernels read (GemmParams::W_blk), keyed like the packed copies above
    // Memory: one extra copy (N * K * 2 bytes) of every weight the planner routes to an LDS-DMA tile with K > 1024 - nearly all 3x3
    // convs and FF layers: ~1.6 GB per SD1.5 UNet handle, ~0.15 GB per VAE handle, per device-slot replica.  The copy is
    // an optimisation only: when it cannot be allocated the launch reads the row-major weights (nullptr here, prep_blk goes on).
    std::unordered_map<const void*, WtEntry> blk_cache;
    bf16_t* blk_lookup(const void* w, size_t bytes, bool* fresh) {
        WtEntry& en = blk_cache[w];
        if (!en.p || en.bytes < bytes) {
            if (en.p) dfree(en.p);                               // superseded (smaller) buffer: released now, not at Store destruction
            en.p = (bf16_t*)dmalloc(bytes, false);
            en.bytes = en.p ? bytes : 0; en.version = 0;
        }
        *fresh = en.p && en.version == weights_version;
        if (en.p) en.version = weights_version;
        return en.p;
    }
    // release one dmalloc'ed buffer early (after the device has finished with it: a superseded cache copy is only replaced between
    // calls of its handle, but kernels of the previous call may still be queued on the stream)
    void dfree(void* ptr) {
        for (size_t i = 0; i < allocs.size(); ++i)
            if (allocs[i] == ptr) { allocs[i] = allocs.back(); allocs.pop_back(); (void)hipDeviceSynchronize(); (void)hipFree(ptr); return; }
    }
    // conv2 rows followed by the 1x1 shortcut's rows, and the sum of the two biases, for the folded form (GemmParams::sc_*): keyed by
    // the conv weight, made on first use, refreshed after any weight change like the other derived copies (+0.4 GB per SD1.5 UNet handle)
    struct ScEntry { bf16_t* w = nullptr; float* b = nullptr; size_t elems = 0; uint64_t version = 0; };
    std::unordered_map<const void*, ScEntry> sc_cache;
    ScEntry* sc_lookup(const void* w, int N, int Kcat, bool* fresh) {
        ScEntry& en = sc_cache[w];
        const size_t elems = (size_t)N * Kcat;
        if (!en.w || en.elems < elems) {
            const size_t wbytes = (elems * 2 + 255) / 256 * 256;
            char* base = (char*)dmalloc(wbytes + ((size_t)N * 4 + 255) / 256 * 256, false);
            if (!base) return nullptr;
            en.w = (bf16_t*)base; en.b = (float*)(base + wbytes); en.elems = elems; en.version = 0;
        }
        *fresh = en.version == weights_version;
        en.version = weights_version;
        return &en;
    }
    ~Store() { for (void* a : allocs) (void)hipFree(a); }
    // returns the cached buffer for `w` (allocating `bytes` on first use) and whether its content is current
    bf16_t* wt_lookup(const void* w, size_t bytes, bool* fresh) {
        WtEntry& en = wt_cache[w];
        if (!en.p || en.bytes < bytes) {
            en.p = (bf16_t*)dmalloc(bytes, false);
            en.bytes = bytes; en.version = 0;
        }
        *fresh = en.p && en.version == weights_version;
        if (en.p) en.version = weights_version;
        return en.p;
    }
    void* dmalloc(size_t bytes, bool zero) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        if (zero) (void)hipMemset(p, 0, bytes);
        allocs.push_back(p);
        return p;
    }
    Param* add(const std::string& key, std::vector<int64_t> shape, int kind, void* dev, int o_pad = 0, int i_pad = 0) {
        auto p = std::make_unique<Param>();
        p->key = key; p->shape = std::move(shape); p->kind = kind; p->dev = dev; p->o_pad = o_pad; p->i_pad = i_pad;
        Param* raw = p.get();
        by_key[key] = raw;
        params.push_back(std::move(p));
        return raw;
    }
    // --- convenience creators: allocate + register -------------------------------------------------
    bf16_t* conv3(const std::string& pfx, int O, int I, float** bias) {
        int ip = pad8(I), op = pad8(O);  // zero rows up to a multiple of 8: the consumer's C % 8 == 0
        bf16_t* w = (bf16_t*)dmalloc((size_t)op * 9 * ip * 2, true);
        add(pfx + ".weight", {O, I, 3, 3}, PK_CONV3, w, op, ip);
        float* b = (float*)dmalloc((size_t)pad8(O) * 4, true);
        add(pfx + ".bias", {O}, PK_VEC, b);
        *bias = b;
        return w;
    }
    // linear or 1x1 conv; rows padded to a multiple of 8 with zeros so the output tensor can feed
    // kernels that need C % 8 == 0
    bf16_t* mat(const std::string& pfx, int O, int I, bool conv1x1, bool has_bias, float** bias, int kind = PK_MAT) {
        int ip = pad8(I), op = pad8(O);
        bf16_t* w = (bf16_t*)dmalloc((size_t)op * ip * 2, true);
        std::vector<int64_t> shp = conv1x1 ? std::vector<int64_t>{O, I, 1, 1} : std::vector<int64_t>{O, I};
        add(pfx + ".weight", shp, kind, w, op, ip);
        if (bias) *bias = nullptr;
        if (has_bias) {
            float* b = (float*)dmalloc((size_t)op * 4, true);
            add(pfx + ".bias", {O}, kind == PK_MAT_GEGLU ? PK_VEC_GEGLU : PK_VEC, b);
            *bias = b;
        }
        return w;
    }
    void vec(const std::string& key, int n, float** out) {
        float* b = (float*)dmalloc((size_t)n * 4, true);
        add(key, {n}, PK_VEC, b);
        *out = b;
    }
    int set_weight(const char* key, const void* src, int dtype, const int64_t* shape, int ndim, hipStream_t st) {
        auto it = by_key.find(key);
        if (it == by_key.end()) GYRE_FAIL(GYRE_ERR_KEY, std::string("unknown weight key: ") + key);
        Param& p = *it->second;
        if (dtype < 0 || dtype > 2) GYRE_FAIL(GYRE_ERR_INVALID, "bad dtype");
        bool ok = (int)p.shape.size() == ndim;
        for (int i = 0; ok && i < ndim; ++i) ok = p.shape[i] == shape[i];
        if (!ok) {
            std::string m = std::string("shape mismatch for ") + key + ": expected [";
            for (auto d : p.shape) m += std::to_string(d) + ",";
            m += "] got [";
            for (int i = 0; i < ndim; ++i) m += std::to_string(shape[i]) + ",";
            GYRE_FAIL(GYRE_ERR_KEY, m + "]");
        }
        int O = (int)p.shape[0];
        switch (p.kind) {
            case PK_CONV3: TRY(launch_repack_conv(st, src, dtype, O, (int)p.shape[1], 3, 3, p.i_pad, (bf16_t*)p.dev)); break;
            case PK_MAT: case PK_MAT_GEGLU: {
                int I = (int)p.shape[1];
                // [O][I] -> [O][i_pad] (reuse the conv repack with a 1x1 window for the padding case)
                if (p.i_pad != I || p.kind == PK_MAT) {
                    if (p.kind == PK_MAT_GEGLU) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "geglu weight with padded K");
                    TRY(launch_repack_conv(st, src, dtype, O, I, 1, 1, p.i_pad, (bf16_t*)p.dev, p.scale));
                } else {
                    TRY(launch_repack_linear(st, src, dtype, O, I, 1, (bf16_t*)p.dev));
                }
                break;
            }
            case PK_VEC: TRY(launch_cast_f32(st, src, dtype, (size_t)O, 0, (float*)p.dev, p.scale)); break;
            case PK_VEC_GEGLU: TRY(launch_cast_f32(st, src, dtype, (size_t)O, 1, (float*)p.dev)); break;
        }
        p.set = true;
        ++weights_version;
        return 0;
    }
    int finalize() {
        for (auto& p : params)
            if (!p->set) GYRE_FAIL(GYRE_ERR_INCOMPLETE, "weight not set: " + p->key);
        return 0;
    }
};

// ------------------------------------------------------------------------------------------
// layer weights
// ------------------------------------------------------------------------------------------
struct ResW {
    int cin = 0, cout = 0;
    float *n1g, *n1b, *n2g, *n2b;
    bf16_t *c1w, *c2w, *scw = nullptr;
    float *c1b, *c2b, *scb = nullptr;
    int temb_off = -1;  // column offset into the batched time_emb_proj output
};
struct AttnW {
    int c = 0, heads = 1, kv_dim = 0;
    bf16_t *wqk = nullptr, *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr;
    int k_prescaled = 0;         // to_k weights carry the softmax scale (UNet attention; not the VAE block)
    bool qkv_fused = false;      // wqk holds [3C][C] = Q | K | V rows (UNet self-attention)
    float *bqk = nullptr, *bq = nullptr, *bk = nullptr, *bv = nullptr, *bo = nullptr;
};
struct TBlockW {
    float *ln1g, *ln1b, *ln2g, *ln2b, *ln3g, *ln3b;
    AttnW a1, a2;
    bf16_t *ff1, *ff2;
    float *ff1b, *ff2b;
};
struct TransW {
    int c = 0, heads = 1;
    float *ng, *nb;
    bf16_t *pin, *pout;
    float *pinb, *poutb;
    std::vector<TBlockW> blocks;
};
struct ConvW { bf16_t* w = nullptr; float* b = nullptr; int cin = 0, cout = 0; };

// ------------------------------------------------------------------------------------------
// execution context: arena + stream + dry-run switch; every op allocates its output
// ------------------------------------------------------------------------------------------
struct CtxKV { bf16_t* k; bf16_t* vt; };
// Activations a forward pass keeps for the input-gradient pass (model_vjp.hip) instead of returning them to the arena
// ao: attention output before the out projection.  ToMe self-attention also keeps what the merge produced - Q|K rows, merged K / V rows
// and the matching (order, destination list) - so the reverse sweep neither re-projects nor re-matches (round 6: 0.2 ms per 64 x 64
// block of the guided step; 288 GB of HBM make the recomputation's memory saving worthless)
struct MhaSave { Tn ao, qk, km, vm, idx; int idx_B = 0, idx_b0 = 0; };      // idx_B: batch the index arrays were laid out for; idx_b0: first sample of a narrowed sweep
struct ResSave { Tn h1; };                                   // conv1 output (input of the second GroupNorm)
struct TBlockSave { Tn h0, n1, h1, n2, h2, n3; MhaSave a1, a2; };   // block input, LN outputs, residual stream after each attention
struct TransSave { std::vector<TBlockSave> blocks; Tn hlast; };     // hlast: residual stream entering proj_out
struct Exec {
    Arena arena;
    hipStream_t st = nullptr;
    Store* store = nullptr;   // owner's weight store (transposed-weight cache of the input-gradient pass); null for bare op calls
    int groups = 32;
    std::string fail;
    // cross-attention K / V^T of the current text context, projected once per request (gyre_unet_set_context)
    const std::vector<CtxKV>* ctx_cache = nullptr;
    size_t ctx_layer = 0;
    int batch = 0;   // samples in the current call (planner hint, see gemm_set_batch_invariant)
    int cs_unit = 0; // channels per GroupNorm-statistics unit the producers emit (0 = producers emit none; set per call)
    int tome_r = 0;  // ToMe: keys / values merged per self-attention (0 = off; reference option "tome", nonfree/tome_unet.py)
    int tiling = 0;  // circular conv padding: bit 0 along x, bit 1 along y (reference option "tiling", unified_pipeline.py:1671-1712)

    bool dry() const { return arena.dry; }
    int alloc(Tn& t, int B, int H, int W, int C, size_t elt = 2) {
        t.B = B; t.H = H; t.W = W; t.C = C;
        // a NEW tensor: whatever statistics record the variable carried belongs to the tensor it named before (a reused
        // variable that kept it would release that tensor's - possibly still live - statistics buffer when freed)
        t.cs = nullptr; t.cs_off = (size_t)-1; t.cs_bytes = 0; t.cs_chunks = 0; t.cs_unit = 0;
        t.bytes = (size_t)B * H * W * C * elt;
        t.off = arena.alloc(t.bytes);
        if (t.off == (size_t)-1 || (!arena.dry && t.off + t.bytes > arena.cap))
            GYRE_FAIL(GYRE_ERR_WORKSPACE, "workspace too small: need " + std::to_string(t.bytes) + " B at offset " +
                      std::to_string((long long)t.off) + ", capacity " + std::to_string(arena.cap) + " B (tensor " +
                      std::to_string(B) + "x" + std::to_string(H) + "x" + std::to_string(W) + "x" + std::to_string(C) + ")");
        t.p = arena.dry ? nullptr : (bf16_t*)(arena.base + t.off);
        return 0;
    }
    int alloc_raw(Tn& t, size_t bytes) { return alloc(t, 1, 1, 1, (int)((bytes + 1) / 2)); }
    void free(Tn& t) {
        if (t.valid()) { arena.release(t.off, t.bytes); t.off = (size_t)-1; t.p = nullptr; }
        if (t.cs_off != (size_t)-1) { arena.release(t.cs_off, t.cs_bytes); t.cs_off = (size_t)-1; t.cs = nullptr; t.cs_chunks = 0; }
    }
    // Ask the launch described by `p` (output tensor y, `rps` rows per sample) to leave the GroupNorm statistics of y
    // (GemmParams::colstat_out).  Only where the consumer takes the streaming two-kernel GroupNorm (large maps) and the planner's
    // kernel for the shape can; otherwise y simply carries none and its GroupNorm runs its own statistics pass.
    int attach_colstats(GemmParams& p, Tn& y, int rps) {
        if (!cs_unit || !store || rps <= 256 || (p.rowbias && p.rows_per_sample != rps)) return 0;
        if (!p.samples) p.samples = batch;
        GemmParams q = p;
        q.colstat_unit = cs_unit; q.rows_per_sample = rps;
        const int rows = gemm_colstat_rows(q);
        if (rows <= 0 || rps / rows > 64) return 0;        // the apply prologue of every workgroup re-reads all chunks
        y.cs_chunks = rps / rows; y.cs_unit = cs_unit;
        y.cs_bytes = (size_t)y.B * y.cs_chunks * (p.N / cs_unit) * 2 * sizeof(float);
        y.cs_off = arena.alloc(y.cs_bytes);
        if (y.cs_off == (size_t)-1 || (!arena.dry && y.cs_off + y.cs_bytes > arena.cap))
            GYRE_FAIL(GYRE_ERR_WORKSPACE, "workspace too small (GroupNorm statistics)");
        y.cs = arena.dry ? nullptr : (float*)(arena.base + y.cs_off);
        p.colstat_unit = cs_unit; p.rows_per_sample = rps; p.colstat_out = y.cs;
        return 0;
    }

    // y = x written twice along the batch axis (B -> 2B); the producer's GroupNorm statistics follow (they are per sample)
    int dup_batch(const Tn& x, Tn& y) {
        TRY(alloc(y, 2 * x.B, x.H, x.W, x.C));
        if (x.cs_chunks > 0) {
            y.cs_chunks = x.cs_chunks; y.cs_unit = x.cs_unit; y.cs_bytes = 2 * x.cs_bytes;
            y.cs_off = arena.alloc(y.cs_bytes);
            if (y.cs_off == (size_t)-1 || (!arena.dry && y.cs_off + y.cs_bytes > arena.cap))
                GYRE_FAIL(GYRE_ERR_WORKSPACE, "workspace too small (GroupNorm statistics)");
            y.cs = arena.dry ? nullptr : (float*)(arena.base + y.cs_off);
        }
        if (dry()) return 0;
        TRY(launch_dup_batch(st, x.p, y.p, 1, x.bytes));
        if (x.cs_chunks > 0) TRY(launch_dup_batch(st, x.cs, y.cs, 1, x.cs_bytes));
        return 0;
    }
    // GroupNorm (+SiLU) of x (optionally concatenated with x2 along C)
    int groupnorm(const Tn& x, const Tn* x2, const float* g, const float* b, float eps, int silu, Tn& y) {
        int C = x.C + (x2 ? x2->C : 0);
        int HW = x.H * x.W;
        Tn ws;
        TRY(alloc_raw(ws, gn_workspace_bytes(x.B, HW, C, groups)));
        TRY(alloc(y, x.B, x.H, x.W, C));
        if (!dry()) {
            GnParams p;
            p.x = x.p; p.x2 = x2 ? x2->p : x.p; p.C1 = x.C; p.B = x.B; p.HW = HW; p.C = C; p.G = groups;
            p.gamma = g; p.beta = b; p.eps = eps; p.silu = silu;
            p.nchunks = gn_pick_chunks(x.B, HW, C);
            p.partial = (float*)ws.p;
            p.scale_shift = (float*)((char*)ws.p + align_up((size_t)x.B * p.nchunks * groups * 2 * sizeof(float), 256));
            p.y = y.p;
            if (gn_use_small(HW, C, x.C, groups)) {
                TRY(launch_groupnorm_small(st, p));
            } else {
                // statistics straight from the producers of x (and of the skip tensor x2), when both left them
                if (x.cs_chunks > 0 && (!x2 || (x2->cs_chunks > 0 && x2->cs_unit == x.cs_unit))) {
                    p.cs_unit = x.cs_unit;
                    if (gn_accepts_colstats(p)) {
                        p.cs_x = x.cs; p.cs_x_chunks = x.cs_chunks;
                        if (x2) { p.cs_x2 = x2->cs; p.cs_x2_chunks = x2->cs_chunks; }
                    }
                }
                if (!p.cs_x) TRY(launch_groupnorm_stats(st, p));
                TRY(launch_groupnorm_apply(st, p));
            }
        }
        free(ws);
        return 0;
    }
    // runs one GEMM launch, giving it split-K slab space from the arena when the planner wants it
    // A-resident kernel (tile config 30): the packed weight copy, made per handle on first use (and after any weight change)
    int prep_ar(GemmParams& p) {
        if (dry()) return 0;
        const int cfg = gemm_plan(p).cfg;
        if (cfg != 30) return 0;
        bool fresh = false;
        void* pk = store ? store->ar_lookup(p.W, gemm_ar_packed_bytes(p.N, p.K), &fresh, cfg) : nullptr;
        if (!pk) GYRE_FAIL(GYRE_ERR_HIP, "cannot allocate the packed weight copy of the A-resident GEMM");
        if (!fresh) TRY(launch_ar_pack(st, p.W, p.N, p.K, pk));
        p.w_packed = pk;
        return 0;
    }
    // blocked weight copy for the LDS-DMA tile kernels (GemmParams::W_blk), made per handle on first use (and after any weight change)
    int prep_blk(GemmParams& p) {
        if (dry() || !store || p.W_blk || !gemm_w_block_wanted(p)) return 0;
        bool fresh = false;
        bf16_t* b = store->blk_lookup(p.W, (size_t)p.N * p.K * 2, &fresh);
        if (!b) { (void)hipGetLastError(); return 0; }           // no memory for the copy: the kernels read the row-major weights (W_blk stays null)
        if (!fresh) TRY(launch_w_block(st, p.W, p.N, p.K, b));
        p.W_blk = b;
        return 0;
    }
    int run_gemm(GemmParams& p) {
        if (!p.samples) p.samples = batch;
        GemmPlan pl = gemm_plan(p);
        TRY(prep_ar(p));
        TRY(prep_blk(p));
        Tn ws;
        if (pl.ws_bytes) {
            TRY(alloc_raw(ws, pl.ws_bytes));
            p.splitk_ws = (float*)ws.p; p.splitk_ws_bytes = pl.ws_bytes;
        }
        int rc = dry() ? 0 : launch_gemm(st, p);
        free(ws);
        return rc;
    }
    // 3x3 conv; x2 = second channel source (skip concat), rowbias = per-sample channel bias (temb)
    // ups: nearest-neighbour upsampling fused into the gather.  (Hup, Wup) = size of the upsampled image, by default
    // 2x; diffusers resizes to the skip connection's size when the latent is not a multiple of 2^levels
    // (UNet2DConditionModel forward_upsample_size), which for sizes 2H-1 is the 2x image minus its last row / column.
    // sc_x (| sc_skip) with sc_w / sc_b: a 1x1 shortcut over that (concatenated) image folded into this convolution as extra K steps
    // (GemmParams::sc_*); the caller has asked conv3_shortcut_foldable() first
    int conv3(const Tn& x, const ConvW& w, int stride, int pad, int ups, const float* rowbias, int ld_rowbias,
              const Tn* residual, Tn& y, int Hup = 0, int Wup = 0, const Tn* sc_x = nullptr, const Tn* sc_skip = nullptr,
              const bf16_t* sc_w = nullptr, const float* sc_b = nullptr) {
        if (ups && ((Hup && (Hup > 2 * x.H || Hup < 2 * x.H - 1)) || (Wup && (Wup > 2 * x.W || Wup < 2 * x.W - 1))))
            GYRE_FAIL(GYRE_ERR_INVALID, "upsample target must be 2x or 2x-1 of the input");
        int Hin = ups ? (Hup ? Hup : 2 * x.H) : x.H, Win = ups ? (Wup ? Wup : 2 * x.W) : x.W;
        int Ho = (Hin + (pad ? 2 : 1) - 3) / stride + 1, Wo = (Win + (pad ? 2 : 1) - 3) / stride + 1;
        TRY(alloc(y, x.B, Ho, Wo, pad8(w.cout)));
        GemmParams p;
        p.A = x.p; p.lda = x.C; p.mode = GEMM_CONV3;
        p.Hi = x.H; p.Wi = x.W; p.Cin = x.C; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = pad; p.ups = ups;
        p.Hup = ups ? Hin : 0; p.Wup = ups ? Win : 0;
        p.W = w.w; p.K = 9 * x.C; p.N = pad8(w.cout); p.M = x.B * Ho * Wo;
        p.bias = w.b; p.rowbias = rowbias; p.rows_per_sample = Ho * Wo; p.ld_rowbias = ld_rowbias;
        if (residual) { p.residual = residual->p; p.ldr = residual->C; }
        p.out = y.p; p.ldc = y.C; p.out_mode = OUT_BF16;
        p.wrap = pad ? tiling : 0;                 // a module's OWN padding turns circular; the VAE's explicit (0,1,0,1) F.pad stays zeros
        if (sc_x) {
            fill_shortcut(p, *sc_x, sc_skip);
            if (!dry()) {
                bool fresh = false;
                Store::ScEntry* e = store->sc_lookup(w.w, p.N, p.K, &fresh);
                if (!e) GYRE_FAIL(GYRE_ERR_HIP, "cannot allocate the folded conv + shortcut weights");
                if (!fresh) {
                    TRY(launch_concat_rows(st, w.w, 9 * x.C, sc_w, p.sc_K, p.N, e->w));
                    GYRE_HIP_CHECK(hipMemcpyAsync(e->b, w.b, (size_t)p.N * sizeof(float), hipMemcpyDeviceToDevice, st));
                    if (sc_b) TRY(launch_add_f32(st, e->b, sc_b, (size_t)p.N));
                }
                p.W = e->w; p.bias = e->b;
            }
        }
        TRY(attach_colstats(p, y, Ho * Wo));       // every conv output of the UNet feeds a GroupNorm
        return run_gemm(p);
    }
    void fill_shortcut(GemmParams& p, const Tn& sx, const Tn* sskip) const {
        p.sc_A = sx.p; p.sc_lda = sx.C; p.sc_C1 = sx.C; p.sc_K = sx.C;
        if (sskip) { p.sc_A2 = sskip->p; p.sc_lda2 = sskip->C; p.sc_K = sx.C + sskip->C; }
        p.K = 9 * p.Cin + p.sc_K;
    }
    // would launch_gemm run conv2 (3x3, stride 1, pad 1 over `b`) with the 1x1 shortcut over sx (| sskip) folded in?  (the pipelined
    // 256x320 tile knows the form; inference path with a weight store; tuning bit 7 of gyre_debug_gemm_ablation = off)
    bool conv3_shortcut_foldable(const Tn& b, int cout, const Tn& sx, const Tn* sskip) {
        if (!store || tiling || b.C % 64 || sx.C % 64 || (sskip && sskip->C % 64) || sx.H != b.H || sx.W != b.W) return false;
        GemmParams p;
        p.A = b.p; p.lda = b.C; p.mode = GEMM_CONV3;
        p.Hi = b.H; p.Wi = b.W; p.Cin = b.C; p.Ho = b.H; p.Wo = b.W; p.stride = 1; p.pad = 1;
        p.N = pad8(cout); p.M = b.B * b.H * b.W; p.rows_per_sample = b.H * b.W; p.samples = batch;
        p.out = (void*)(uintptr_t)256; p.ldc = p.N; p.out_mode = OUT_BF16;
        if (dry()) p.A = (const bf16_t*)(uintptr_t)256;
        fill_shortcut(p, sx, sskip);
        if (dry()) { p.sc_A = (const bf16_t*)(uintptr_t)256; if (sskip) p.sc_A2 = (const bf16_t*)(uintptr_t)256; }
        return gemm_conv_shortcut_ok(p);
    }
    // final 3x3 conv straight to the caller's NCHW buffer
    int conv3_nchw(const Tn& x, const ConvW& w, void* out, int out_dtype) {
        if (dry()) return 0;
        // the dedicated kernel (kernels_conv_out.hip): zero padding only - a tiling request keeps the tile kernels' circular gather
        if (!tiling && conv_out_supports(x.C, w.cout))
            return launch_conv_out(st, x.p, x.B, x.H, x.W, x.C, w.w, w.b, w.cout, out, out_dtype);
        GemmParams p;
        p.A = x.p; p.lda = x.C; p.mode = GEMM_CONV3;
        p.Hi = x.H; p.Wi = x.W; p.Cin = x.C; p.Ho = x.H; p.Wo = x.W; p.stride = 1; p.pad = 1;
        p.W = w.w; p.K = 9 * x.C; p.N = w.cout; p.M = x.rows();
        p.bias = w.b; p.rows_per_sample = x.H * x.W;
        p.out = out; p.out_mode = OUT_NCHW; p.out_dtype = out_dtype; p.wrap = tiling;
        return launch_gemm(st, p);
    }
    // y[M][N] = x[M][K] (|| x2) @ w^T + bias (+ residual); geglu halves N
    // Per-row partial statistics of a linear layer's output, left by its epilogue for the folded LayerNorm of the next GEMM
    // (GemmParams::rowstat_out -> ln_parts).  nparts == 0: the producer could not (4-wave tile, split K, ...): the consumer
    // runs the separate statistics pass.
    struct RowStatBuf { Tn t; int nparts = 0; };
    // rs: the caller wants the row statistics of y (and frees rs->t after the consumer ran)
    // cs_for: the output tensor (y == cs_for->p) feeds a GroupNorm: ask the launch for its statistics (attach_colstats)
    int linear(const bf16_t* x, int lda, const bf16_t* x2, int lda2, int C1, int M, int K, const bf16_t* w, int N,
               const float* bias, const bf16_t* residual, int ldr, int geglu, bf16_t* y, int ldc, RowStatBuf* rs = nullptr,
               Tn* cs_for = nullptr) {
        GemmParams p;
        p.A = x; p.lda = lda; p.A2 = x2; p.lda2 = lda2; p.C1 = C1; p.mode = GEMM_LINEAR;
        p.W = w; p.K = K; p.N = N; p.M = M; p.bias = bias; p.residual = residual; p.ldr = ldr; p.geglu = geglu;
        p.out = y; p.ldc = ldc; p.out_mode = OUT_BF16;
        p.ar_ok = store != nullptr; p.no_ar = cs_for != nullptr;      // (the A-resident kernel leaves no column statistics)
        if (cs_for && !rs) TRY(attach_colstats(p, *cs_for, cs_for->H * cs_for->W));
        if (rs) {
            if (!p.samples) p.samples = batch;
            rs->nparts = store ? gemm_rowstat_parts(p) : 0;
            if (rs->nparts > 0) {
                TRY(alloc_raw(rs->t, (size_t)rs->nparts * M * 2 * sizeof(float)));
                p.rowstat_out = (float*)rs->t.p;
                if (dry()) p.rowstat_out = nullptr;
            }
        }
        return run_gemm(p);
    }
    int linear_t(const bf16_t* x, int lda, int M, int K, const bf16_t* w, int N, const float* bias, int tokens, int ldt,
                 bf16_t* y) {
        if (dry()) return 0;
        GemmParams p;
        p.A = x; p.lda = lda; p.mode = GEMM_LINEAR; p.W = w; p.K = K; p.N = N; p.M = M; p.bias = bias;
        p.out = y; p.out_mode = OUT_BF16_T; p.tokens_per_batch = tokens; p.ldt = ldt;
        return launch_gemm(st, p);
    }
    // y = Linear(GroupNorm(x)) with the (affine-only) norm folded into per-sample weights (launch_gn_fold): the normalised tensor
    // is never written.  Taken at the large maps, where the apply pass costs more than B copies of the C x C weights (64 x 64:
    // 84 MB against 3.3 MB; 32 x 32: 42 against 13; 16 x 16 would be 10 against 52) and the planner's kernel reads per-sample
    // weights; *folded tells the caller whether it happened.
    int gn_fold_linear(const Tn& x, const float* g, const float* b, float eps, const bf16_t* w, const float* bias, int N, Tn& y,
                       RowStatBuf* rs, bool* folded) {
        *folded = false;
        const int C = x.C, HW = x.H * x.W, M = x.rows();
        // (worth it where the per-sample weights are far smaller than the tensor: at the 32x32 level the fold kernel, B x 640 x 640
        // weights, takes as long as the apply pass it replaces - measured 20.9 against 16.2 us; at 64x64 8 against 19 us)
        if (!store || HW < 8 * N || gn_use_small(HW, C, C, groups) || C % groups) return 0;
        GemmParams p;
        p.A = x.p; p.lda = C; p.mode = GEMM_LINEAR; p.W = w; p.K = C; p.N = N; p.M = M; p.samples = batch;
        p.out = y.p; p.ldc = y.C; p.out_mode = OUT_BF16; p.rows_per_sample = HW; p.ld_rowbias = N;
        p.rowbias = (const float*)(uintptr_t)256;              // (planning only: any aligned non-null value)
        if (!gemm_per_sample_w_ok(p)) return 0;
        Tn ws, wf, bfv;
        TRY(alloc_raw(ws, gn_workspace_bytes(x.B, HW, C, groups)));
        TRY(alloc_raw(wf, (size_t)x.B * N * C * 2));
        TRY(alloc_raw(bfv, (size_t)x.B * N * sizeof(float)));
        if (rs) {
            rs->nparts = gemm_rowstat_parts(p);
            if (rs->nparts > 0) TRY(alloc_raw(rs->t, (size_t)rs->nparts * M * 2 * sizeof(float)));
        }
        if (!dry()) {
            GnParams gp;
            gp.x = x.p; gp.x2 = x.p; gp.C1 = C; gp.B = x.B; gp.HW = HW; gp.C = C; gp.G = groups;
            gp.gamma = g; gp.beta = b; gp.eps = eps; gp.silu = 0;
            gp.nchunks = gn_pick_chunks(x.B, HW, C);
            gp.partial = (float*)ws.p;
            gp.scale_shift = (float*)((char*)ws.p + align_up((size_t)x.B * gp.nchunks * groups * 2 * sizeof(float), 256));
            if (x.cs_chunks > 0) {
                gp.cs_unit = x.cs_unit;
                if (gn_accepts_colstats(gp)) { gp.cs_x = x.cs; gp.cs_x_chunks = x.cs_chunks; }
            }
            if (!gp.cs_x) TRY(launch_groupnorm_stats(st, gp));      // partials (the fold kernel finishes them)
            TRY(launch_gn_fold(st, gp, w, bias, N, wf.p, (float*)bfv.p));
            p.W = wf.p; p.w_sample_stride = (size_t)N * C; p.bias = nullptr; p.rowbias = (const float*)bfv.p;
            if (rs && rs->nparts > 0) p.rowstat_out = (float*)rs->t.p;
            TRY(launch_gemm(st, p));
        }
        free(ws); free(wf); free(bfv);
        *folded = true;
        return 0;
    }
    int layernorm(const Tn& x, const float* g, const float* b, Tn& y) {
        TRY(alloc(y, x.B, x.H, x.W, x.C));
        if (dry()) return 0;
        return launch_layernorm(st, x.p, x.rows(), x.C, g, b, 1e-5f, y.p);
    }
    // LayerNorm folded into the GEMM that consumes it (kernels.h GemmParams::ln_colsum): `p` describes the product of the RAW
    // rows with the layer's own weights / bias; on return it points at the gamma-folded copies (made on first use per handle,
    // refreshed after any gyre_*_set_weight) and the launch normalises on the fly - the normalised tensor never exists.
    struct LnFold { const float* g; const float* b; const RowStatBuf* have = nullptr; };
    bool ln_fusable(const GemmParams& p) const { return store && gemm_ln_fusable(p); }
    // `stats`: arena scratch for the per-row statistics (one streaming pass over the rows); freed by the caller after the launch
    int ln_fold_into(GemmParams& p, const LnFold& ln, Tn& stats) {
        if (ln.have && ln.have->nparts > 0) {           // the producer of these rows left their partial sums
            p.ln_parts = (const float*)ln.have->t.p; p.ln_nparts = ln.have->nparts; p.ln_eps = 1e-5f;
        } else {
            TRY(alloc_raw(stats, (size_t)p.M * 2 * sizeof(float)));
            if (!dry()) {
                TRY(launch_layernorm_stats(st, p.A, p.M, p.K, 1e-5f, (float*)stats.p));
                p.ln_stats = (const float*)stats.p;
            }
        }
        if (dry()) return 0;
        bool fresh = false;
        Store::LnEntry* e = store->ln_lookup(p.W, p.N, p.K, &fresh);
        if (!e) GYRE_FAIL(GYRE_ERR_HIP, "cannot allocate the LayerNorm-folded weight copy");
        if (!fresh) TRY(launch_ln_fold(st, p.W, p.N, p.K, ln.g, ln.b, p.bias, e->wf, e->cs, e->bb));
        p.W = e->wf; p.bias = e->bb; p.ln_colsum = e->cs;
        return 0;
    }
    // y = LayerNorm(x) @ w^T + bias (geglu halves N): one launch when the planner's kernel can fold the norm, else two
    int ln_linear(const Tn& x, const LnFold& ln, const bf16_t* w, int N, const float* bias, int geglu, bf16_t* y, int ldc) {
        GemmParams p;
        p.A = x.p; p.lda = x.C; p.mode = GEMM_LINEAR; p.W = w; p.K = x.C; p.N = N; p.M = x.rows(); p.bias = bias; p.geglu = geglu;
        p.out = y; p.ldc = ldc; p.out_mode = OUT_BF16; p.samples = batch; p.ar_ok = store != nullptr;
        if (ln_fusable(p)) {
            Tn stats;
            TRY(ln_fold_into(p, ln, stats));
            int rc = run_gemm(p);
            free(stats);
            return rc;
        }
        Tn n;
        TRY(layernorm(x, ln.g, ln.b, n));
        p.A = n.p;
        int rc = run_gemm(p);
        free(n);
        return rc;
    }
    // multi-head attention of tokens x against kv source (self: kv == nullptr); out = proj(attn) + residual
    // ln != nullptr: xq_in holds the rows BEFORE the block's LayerNorm; the projections fold it where they can
    int mha(const Tn& xq_in, bool cross, const bf16_t* kvsrc, int kv_rows_per_batch, int kv_dim, const AttnW& w,
            const Tn& residual, Tn& out, MhaSave* sv = nullptr, const LnFold* ln = nullptr, RowStatBuf* rs_out = nullptr) {
        const int B = xq_in.B, Nq = xq_in.H * xq_in.W, C = w.c, D = C / w.heads;
        Tn q, k, vt, ao, nrm;
        Tn xq = xq_in;
        auto normalise = [&]() -> int {          // fallback: the separate LayerNorm pass
            if (!ln) return 0;
            TRY(layernorm(xq_in, ln->g, ln->b, nrm));
            xq = nrm; ln = nullptr;
            return 0;
        };
        const bf16_t *qp, *kp, *vtp = nullptr; int ldq, ldk, Nk, ldvt;
        Tn km, vrow, tws;
        const int tr = (!cross && tome_r > 0 && Nq % 16 == 0) ? tome_effective_r(Nq, tome_r) : 0;
        if (!cross && tr > 0) {
            TRY(normalise());
            // ToMe (nonfree/tome_unet.py:138-182): K and V of a self-attention are projected row-major, the r most
            // redundant even-position keys are averaged into their best odd-position match (values follow the same
            // assignment), and the attention runs against N - r keys.  Queries are untouched.
            Nk = Nq - tr; ldvt = (Nk + 7) / 8 * 8;
            TRY(alloc(q, B, xq.H, xq.W, 2 * C));
            TRY(linear(xq.p, C, nullptr, 0, 0, B * Nq, C, w.wqk, 2 * C, w.bqk, nullptr, 0, 0, q.p, 2 * C));
            TRY(alloc(vrow, B, xq.H, xq.W, C));
            TRY(linear(xq.p, C, nullptr, 0, 0, B * Nq, C, w.wv, C, w.bv, nullptr, 0, 0, vrow.p, C));
            TRY(alloc(km, B, Nk, 1, C));
            TRY(alloc(vt, B, C, 1, ldvt));
            Tn vm, idx;
            if (sv) {
                TRY(alloc(vm, B, Nk, 1, C));
                TRY(alloc_raw(idx, (size_t)3 * B * (Nq / 2) * 4 + 768));
            }
            TRY(alloc_raw(tws, tome_workspace_bytes(B, Nq, C)));
            if (!dry()) {
                TomeParams tp;
                tp.k = q.p + C; tp.ldk = 2 * C; tp.v = vrow.p; tp.ldv = C; tp.B = B; tp.N = Nq; tp.C = C; tp.r = tr;
                tp.k_out = km.p; tp.vt_out = vt.p; tp.ldvt = ldvt; tp.ws = tws.p; tp.ws_bytes = tws.bytes;
                if (sv) {
                    int* order = (int*)idx.p;
                    tp.vrows_out = vm.p; tp.order_out = order; tp.dstlist_out = order + (((size_t)B * (Nq / 2) + 63) & ~(size_t)63);
                }
                TRY(launch_tome_merge(st, tp));
            }
            free(tws); free(vrow);
            if (sv) { sv->vm = vm; sv->idx = idx; sv->idx_B = B; sv->idx_b0 = 0; }
            qp = q.p; kp = km.p; vtp = vt.p; ldq = 2 * C; ldk = C;
        } else if (!cross) {  // self attention: fused Q|K projection, V projected straight into V^T
            Nk = Nq; ldvt = (Nk + 7) / 8 * 8;
            TRY(alloc(q, B, xq.H, xq.W, 2 * C));
            TRY(alloc(vt, B, C, 1, ldvt));
            bool fused = false;
            if (w.qkv_fused && !w.bqk && !w.bv && Nq % 8 == 0) {
                // Q | K | V in one launch: the V tiles write V^T through the transposing epilogue (needs an 8-wave
                // tile config whose wave tiles line up with the V columns; else two launches as before)
                GemmParams p;
                p.A = xq.p; p.lda = C; p.mode = GEMM_LINEAR; p.W = w.wqk; p.K = C; p.N = 3 * C; p.M = B * Nq; p.samples = B;
                p.out = q.p; p.ldc = 2 * C; p.out_mode = OUT_BF16;
                p.vt_out = vt.p; p.vt_col0 = 2 * C; p.tokens_per_batch = Nq; p.ldt = ldvt;
                p.ar_ok = store != nullptr;
                if (dry()) { p.out = (void*)(uintptr_t)256; p.vt_out = (bf16_t*)(uintptr_t)256; p.A = (const bf16_t*)(uintptr_t)256; }  // (planning: aligned non-null)
                GemmPlan pl = gemm_plan(p);
                // (config 30 = the A-resident kernel: its 64-row weight tiles line up with the V columns whenever 2 C % 64 == 0)
                const int tn = pl.cfg == 4 ? 160 : (pl.cfg == 5 || pl.cfg == 8) ? 80 : pl.cfg == 6 ? 128 : (pl.cfg == 7 || pl.cfg == 30 || pl.cfg == 32) ? 64 : 0;
                if (tn && pl.splits == 1 && (2 * C) % tn == 0) {
                    fused = true;
                    Tn stats;
                    if (ln && ln_fusable(p)) {
                        TRY(ln_fold_into(p, *ln, stats));
                        ln = nullptr;
                    } else {
                        TRY(normalise());
                        p.A = xq.p;
                    }
                    if (!dry()) {
                        TRY(prep_ar(p));
                        TRY(prep_blk(p));
                        TRY(launch_gemm(st, p));
                    }
                    free(stats);
                }
            }
            if (!fused) {
                TRY(normalise());
                TRY(linear(xq.p, C, nullptr, 0, 0, B * Nq, C, w.wqk, 2 * C, w.bqk, nullptr, 0, 0, q.p, 2 * C));
                TRY(linear_t(xq.p, C, B * Nq, C, w.wv, C, w.bv, Nq, ldvt, vt.p));
            }
            qp = q.p; kp = dry() ? nullptr : q.p + C; vtp = vt.p; ldq = ldk = 2 * C;
        } else {
            Nk = kv_rows_per_batch; ldvt = (Nk + 7) / 8 * 8;
            // Round 6: the whole block - to_q with the folded LayerNorm, attention over the cached text keys, to_out + bias + residual
            // and the row statistics for the next LayerNorm - as ONE kernel per 128 rows (kernels_xattn.hip) where the shape fits it
            // (SD1.x's 64x64 level); inference path with a context cache only; tuning bit 13 of gyre_debug_gemm_ablation = off
            if (ln && !sv && ctx_cache && store && !w.bq && w.k_prescaled && residual.p == xq_in.p && !(gemm_planner_state() & 0x2000L) &&
                // (batch-invariant planning: the grid-size rule is evaluated for the canonical batch, so that every split of a request
                //  takes the same path - a sub-batch then runs the fused kernel on a small grid, slower and bit-identical)
                xattn_supports(C, w.heads, Nq, Nk, gemm_get_batch_invariant() > 0 ? gemm_get_batch_invariant() * Nq : B * Nq)) {
                if (ctx_layer >= ctx_cache->size()) GYRE_FAIL(GYRE_ERR_INVALID, "internal: context cache layer overflow");
                GemmParams pq;                         // describes to_q for the LayerNorm fold (weights W' = W gamma, colsum, bias')
                pq.A = xq_in.p; pq.lda = C; pq.mode = GEMM_LINEAR; pq.W = w.wq; pq.K = C; pq.N = C; pq.M = B * Nq; pq.bias = nullptr;
                Tn stats;
                TRY(ln_fold_into(pq, *ln, stats));
                TRY(alloc(out, B, xq.H, xq.W, C));
                if (rs_out) {
                    rs_out->nparts = 1;
                    TRY(alloc_raw(rs_out->t, (size_t)B * Nq * 2 * sizeof(float)));
                }
                if (!dry()) {
                    XattnParams xp;
                    xp.x = xq_in.p; xp.ldx = C; xp.wq = pq.W; xp.q_colsum = pq.ln_colsum; xp.q_bias = pq.bias;
                    xp.ln_parts = pq.ln_parts; xp.ln_nparts = pq.ln_nparts; xp.ln_stats = pq.ln_stats; xp.ln_eps = 1e-5f;
                    xp.k = (*ctx_cache)[ctx_layer].k; xp.vt = (*ctx_cache)[ctx_layer].vt; xp.ldvt = ldvt;
                    xp.wo = w.wo; xp.bo = w.bo; xp.out = out.p; xp.ldo = C;
                    xp.rowstat_out = rs_out ? (float*)rs_out->t.p : nullptr;
                    xp.M = B * Nq; xp.rows_per_sample = Nq; xp.Nk = Nk; xp.heads = w.heads;
                    TRY(launch_xattn(st, xp, C));
                }
                ++ctx_layer;
                free(stats);
                return 0;
            }
            TRY(alloc(q, B, xq.H, xq.W, C));
            if (ln) {
                TRY(ln_linear(xq_in, *ln, w.wq, C, w.bq, 0, q.p, C));
                ln = nullptr;
            } else {
                TRY(linear(xq.p, C, nullptr, 0, 0, B * Nq, C, w.wq, C, w.bq, nullptr, 0, 0, q.p, C));
            }
            if (ctx_cache) {  // K / V^T of this layer were projected when the context was set
                if (ctx_layer >= ctx_cache->size()) GYRE_FAIL(GYRE_ERR_INVALID, "internal: context cache layer overflow");
                kp = (*ctx_cache)[ctx_layer].k; vtp = (*ctx_cache)[ctx_layer].vt;
                ++ctx_layer;
            } else {
                TRY(alloc(k, B, Nk, 1, C));
                TRY(linear(kvsrc, kv_dim, nullptr, 0, 0, B * Nk, kv_dim, w.wk, C, w.bk, nullptr, 0, 0, k.p, C));
                TRY(alloc(vt, B, C, 1, ldvt));
                TRY(linear_t(kvsrc, kv_dim, B * Nk, kv_dim, w.wv, C, w.bv, Nk, ldvt, vt.p));
                kp = k.p; vtp = vt.p;
            }
            qp = q.p; ldq = ldk = C;
        }
        TRY(alloc(ao, B, xq.H, xq.W, C));
        if (!dry()) {
            AttnParams a;
            a.q = qp; a.ldq = ldq; a.k = kp; a.ldk = ldk; a.vt = vtp; a.ldvt = ldvt; a.o = ao.p; a.ldo = C;
            a.B = B; a.H = w.heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.k_prescaled = w.k_prescaled;
            TRY(launch_attention(st, a));
        }
        if (sv && tr > 0) { sv->qk = q; sv->km = km; q = Tn(); km = Tn(); }
        else if (sv && cross) { sv->qk = q; q = Tn(); }       // the reverse sweep's S = Q K^T recomputation starts from the kept Q
        free(q); free(k); free(vt); free(km); free(nrm);
        TRY(alloc(out, B, xq.H, xq.W, C));
        TRY(linear(ao.p, C, nullptr, 0, 0, B * Nq, C, w.wo, C, w.bo, residual.p, C, 0, out.p, C, rs_out));
        if (sv) sv->ao = ao; else free(ao);
        return 0;
    }
    int resnet(const Tn& x, const Tn* skip, const ResW& w, const float* tproj, int ld_tproj, float eps, Tn& out,
               ResSave* sv = nullptr) {
        Tn a, h1, b, sc;
        TRY(groupnorm(x, skip, w.n1g, w.n1b, eps, 1, a));
        ConvW c1{w.c1w, w.c1b, w.cin, w.cout};
        TRY(conv3(a, c1, 1, 1, 0, (tproj && w.temb_off >= 0) ? tproj + w.temb_off : nullptr, ld_tproj, nullptr, h1));
        free(a);
        TRY(groupnorm(h1, nullptr, w.n2g, w.n2b, eps, 1, b));
        if (sv) sv->h1 = h1; else free(h1);
        const Tn* res = &x;
        if (w.scw && !sv && conv3_shortcut_foldable(b, w.cout, x, skip)) {
            // Round 6: the 1x1 shortcut rides inside conv2 as extra K steps of the pipelined tile (same accumulators, one rounding):
            // no shortcut launch, no shortcut tensor, no residual read
            ConvW c2f{w.c2w, w.c2b, w.cout, w.cout};
            TRY(conv3(b, c2f, 1, 1, 0, nullptr, 0, nullptr, out, 0, 0, &x, skip, w.scw, w.scb));
            free(b);
            return 0;
        }
        if (w.scw) {
            TRY(alloc(sc, x.B, x.H, x.W, w.cout));
            TRY(linear(x.p, x.C, skip ? skip->p : nullptr, skip ? skip->C : 0, x.C, x.rows(), w.cin, w.scw, w.cout,
                       w.scb, nullptr, 0, 0, sc.p, w.cout));
            res = &sc;
        } else if (skip) {
            GYRE_FAIL(GYRE_ERR_INVALID, "resnet: concat input needs a shortcut conv");
        }
        ConvW c2{w.c2w, w.c2b, w.cout, w.cout};
        TRY(conv3(b, c2, 1, 1, 0, nullptr, 0, res, out));
        free(b); free(sc);
        return 0;
    }
    // x_full != nullptr (inference only): x holds HALF the batch - the two halves of a CFG-parallel call are identical up
    // to the first cross-attention (reference unet/cfg.py:49-57: cat[x, x] against cat[uncond, cond]) - so GroupNorm, proj_in
    // and the first block's self-attention run once per pair; their result, the block input and its row statistics are then
    // written twice along the batch axis (*x_full = the doubled x, the residual of proj_out) and the rest runs on 2B samples.
    int transformer(const Tn& x_in, const Tn& ctx, int S, int ctx_dim, const TransW& w, Tn& out, TransSave* sv = nullptr,
                    Tn* x_full = nullptr) {
        int B = x_in.B, M = x_in.rows();
        const int C = w.c;
        Tn x = x_in;
        Tn a, h;
        RowStatBuf rs_h;                 // statistics of the current block input, when its producer could leave them
        TRY(alloc(h, B, x.H, x.W, C));
        bool folded = false;
        if (!sv) TRY(gn_fold_linear(x, w.ng, w.nb, 1e-6f, w.pin, w.pinb, C, h, &rs_h, &folded));
        if (!folded) {
            TRY(groupnorm(x, nullptr, w.ng, w.nb, 1e-6f, 0, a));
            TRY(linear(a.p, C, nullptr, 0, 0, M, C, w.pin, C, w.pinb, nullptr, 0, 0, h.p, C, sv ? nullptr : &rs_h));
            free(a);
        }
        if (sv) sv->blocks.assign(w.blocks.size(), TBlockSave());
        for (size_t bi = 0; bi < w.blocks.size(); ++bi) {
            const TBlockW& bw = w.blocks[bi];
            TBlockSave* bs = sv ? &sv->blocks[bi] : nullptr;
            Tn n, h2, ff;
            if (!bs) {
                // inference: the three LayerNorms ride inside the GEMMs that consume them (Q|K|V, cross to_q, GEGLU FF1)
                // each producer of a block-internal tensor (proj_in / FF2 of the previous block, the two to_out) leaves the
                // row sums its consumer's LayerNorm needs, so no statistics pass runs in between
                RowStatBuf rs1, rs2, rs3;
                const LnFold l1{bw.ln1g, bw.ln1b, &rs_h};
                TRY(mha(h, false, nullptr, 0, 0, bw.a1, h, h2, nullptr, &l1, &rs1));
                free(rs_h.t); rs_h.nparts = 0; free(h); h = h2;
                if (x_full && bi == 0) {         // end of the part the CFG halves share: double the batch
                    Tn hd;
                    TRY(dup_batch(h, hd));
                    free(h); h = hd;
                    if (rs1.nparts > 0) {        // [nparts][M][2] floats -> [nparts][2M][2]
                        Tn rd;
                        TRY(alloc_raw(rd, 2 * rs1.t.bytes));
                        if (!dry()) TRY(launch_dup_batch(st, rs1.t.p, rd.p, rs1.nparts, (size_t)M * 2 * sizeof(float)));
                        free(rs1.t); rs1.t = rd;
                    }
                    TRY(dup_batch(x, *x_full));
                    x = *x_full;
                    batch = 2 * B; B = x.B; M = x.rows();
                }
                const LnFold l2{bw.ln2g, bw.ln2b, &rs1};
                TRY(mha(h, true, ctx.p, S, ctx_dim, bw.a2, h, h2, nullptr, &l2, &rs2));
                free(rs1.t); free(h); h = h2;
                const LnFold l3{bw.ln3g, bw.ln3b, &rs2};
                TRY(alloc(ff, B, x.H, x.W, 4 * C));
                TRY(ln_linear(h, l3, bw.ff1, 8 * C, bw.ff1b, 1, ff.p, 4 * C));
                free(rs2.t);
                TRY(alloc(h2, B, x.H, x.W, C));
                const bool more = bi + 1 < w.blocks.size();
                TRY(linear(ff.p, 4 * C, nullptr, 0, 0, M, 4 * C, bw.ff2, C, bw.ff2b, h.p, C, 0, h2.p, C, more ? &rs3 : nullptr));
                if (more) rs_h = rs3;
                free(ff); free(h);
                h = h2;
                continue;
            }
            TRY(layernorm(h, bw.ln1g, bw.ln1b, n));
            TRY(mha(n, false, nullptr, 0, 0, bw.a1, h, h2, bs ? &bs->a1 : nullptr));
            if (bs) { bs->h0 = h; bs->n1 = n; } else { free(n); free(h); }
            h = h2;
            TRY(layernorm(h, bw.ln2g, bw.ln2b, n));
            TRY(mha(n, true, ctx.p, S, ctx_dim, bw.a2, h, h2, bs ? &bs->a2 : nullptr));
            if (bs) { bs->h1 = h; bs->n2 = n; } else { free(n); free(h); }
            h = h2;
            TRY(layernorm(h, bw.ln3g, bw.ln3b, n));
            TRY(alloc(ff, B, x.H, x.W, 4 * C));
            TRY(linear(n.p, C, nullptr, 0, 0, M, C, bw.ff1, 8 * C, bw.ff1b, nullptr, 0, 1, ff.p, 4 * C));
            if (bs) bs->n3 = n; else free(n);
            TRY(alloc(h2, B, x.H, x.W, C));
            TRY(linear(ff.p, 4 * C, nullptr, 0, 0, M, 4 * C, bw.ff2, C, bw.ff2b, h.p, C, 0, h2.p, C));
            free(ff);
            if (bs) bs->h2 = h; else free(h);
            h = h2;
        }
        free(rs_h.t);
        TRY(alloc(out, B, x.H, x.W, C));
        TRY(linear(h.p, C, nullptr, 0, 0, M, C, w.pout, C, w.poutb, x.p, C, 0, out.p, C, nullptr, sv ? nullptr : &out));
        if (sv) sv->hlast = h; else free(h);
        return 0;
    }
};

// ------------------------------------------------------------------------------------------
// weight registration helpers
// ------------------------------------------------------------------------------------------
static void reg_resnet(Store& s, const std::string& p, int cin, int cout, bool temb, bf16_t* temb_w, float* temb_b,
                       int temb_dim, int& temb_cols, ResW& w) {
    w.cin = cin; w.cout = cout;
    s.vec(p + ".norm1.weight", cin, &w.n1g); s.vec(p + ".norm1.bias", cin, &w.n1b);
    w.c1w = s.conv3(p + ".conv1", cout, cin, &w.c1b);
    if (temb) {
        w.temb_off = temb_cols;
        s.add(p + ".time_emb_proj.weight", {cout, temb_dim}, PK_MAT, temb_w + (size_t)temb_cols * temb_dim, cout, temb_dim);
        s.add(p + ".time_emb_proj.bias", {cout}, PK_VEC, temb_b + temb_cols);
        temb_cols += cout;
    }
    s.vec(p + ".norm2.weight", cout, &w.n2g); s.vec(p + ".norm2.bias", cout, &w.n2b);
    w.c2w = s.conv3(p + ".conv2", cout, cout, &w.c2b);
    if (cin != cout) w.scw = s.mat(p + ".conv_shortcut", cout, cin, true, true, &w.scb);
}
static void reg_attn(Store& s, const std::string& p, int c, int heads, int kv_dim, bool self, AttnW& w) {
    w.c = c; w.heads = heads; w.kv_dim = kv_dim; w.k_prescaled = 1;
    // The softmax scale log2(e)/sqrt(head_dim) is folded into the K projection weights (fp32, before their one bf16
    // rounding): S = q.k then arrives from the matrix core already in the exp2 domain (AttnParams::k_prescaled)
    const float kscale = 1.4426950408889634f / sqrtf((float)(c / heads));
    if (self) {   // one [3C][C] matrix: rows Q | K | V, so that the three projections can run as a single GEMM
        w.wqk = (bf16_t*)s.dmalloc((size_t)3 * c * c * 2, true);
        s.add(p + ".to_q.weight", {c, c}, PK_MAT, w.wqk, c, c);
        s.add(p + ".to_k.weight", {c, c}, PK_MAT, w.wqk + (size_t)c * c, c, c)->scale = kscale;
        w.wv = w.wqk + (size_t)2 * c * c;
        s.add(p + ".to_v.weight", {c, c}, PK_MAT, w.wv, c, c);
        w.qkv_fused = true;
    } else {
        w.wq = s.mat(p + ".to_q", c, c, false, false, nullptr);
        w.wk = s.mat(p + ".to_k", c, kv_dim, false, false, nullptr);
        s.by_key[p + ".to_k.weight"]->scale = kscale;
        w.wv = s.mat(p + ".to_v", c, kv_dim, false, false, nullptr);
    }
    w.wo = s.mat(p + ".to_out.0", c, c, false, true, &w.bo);
}
static void reg_transformer(Store& s, const std::string& p, int c, int heads, int ctx_dim, int depth, bool linproj,
                            TransW& w) {
    w.c = c; w.heads = heads;
    s.vec(p + ".norm.weight", c, &w.ng); s.vec(p + ".norm.bias", c, &w.nb);
    w.pin = s.mat(p + ".proj_in", c, c, !linproj, true, &w.pinb);
    w.blocks.resize(depth);
    for (int d = 0; d < depth; ++d) {
        std::string b = p + ".transformer_blocks." + std::to_string(d);
        TBlockW& bw = w.blocks[d];
        s.vec(b + ".norm1.weight", c, &bw.ln1g); s.vec(b + ".norm1.bias", c, &bw.ln1b);
        s.vec(b + ".norm2.weight", c, &bw.ln2g); s.vec(b + ".norm2.bias", c, &bw.ln2b);
        s.vec(b + ".norm3.weight", c, &bw.ln3g); s.vec(b + ".norm3.bias", c, &bw.ln3b);
        reg_attn(s, b + ".attn1", c, heads, c, true, bw.a1);
        reg_attn(s, b + ".attn2", c, heads, ctx_dim, false, bw.a2);
        bw.ff1 = s.mat(b + ".ff.net.0.proj", 8 * c, c, false, true, &bw.ff1b, PK_MAT_GEGLU);
        bw.ff2 = s.mat(b + ".ff.net.2", c, 4 * c, false, true, &bw.ff2b);
    }
    w.pout = s.mat(p + ".proj_out", c, c, !linproj, true, &w.poutb);
}

// ------------------------------------------------------------------------------------------
// UNet
// ------------------------------------------------------------------------------------------
// Bookkeeping between the forward and the reverse half of a UNet input-gradient call (model_vjp.hip): one node per resnet
// (+ transformer) of the down / mid / up path or per resampling conv, with the activations its adjoint needs
struct UNetVjpNode {
    const ResW* rw = nullptr; const TransW* tw = nullptr; const ConvW* cw = nullptr;
    Tn x, skip, r, out; bool has_skip = false;
    ResSave rs; TransSave ts;
};
struct UNetVjpState {
    bool valid = false, dry = false;
    int B = 0, H = 0, W = 0, S = 0, xin_c = 0;
    void* ws = nullptr; size_t ws_bytes = 0;
    std::vector<UNetVjpNode> downs, ups;
    UNetVjpNode midn, mid1n;
    std::vector<Tn> skips;
    Tn h, cx, tp;
};

struct gyre_unet {
    gyre_unet_cfg cfg;
    Store store;
    Exec ex;
    UNetVjpState vjp;
    bool finalized = false;
    int temb_dim = 0, temb_cols = 0;
    std::map<std::array<long, 7>, size_t> ws_memo;   // gyre_unet_workspace_bytes: peak per (B, H, W, S, tome_r, cached context, planner state)
    int* hint_flag = nullptr;      // device word of the hint verifier (GYRE_VERIFY_HINTS=1)
    bool hint_uniform_t = false;   // gyre_unet_hint_uniform_timestep: consumed by the next forward
    bool hint_cfg_pairs = false;   // gyre_unet_hint_cfg_pairs: consumed by the next forward
    int gn_unit = 0;     // gcd of block_out_channels / groups: every GroupNorm group (skip concats included) is a whole number of units
    bf16_t *te1w, *te2w; float *te1b, *te2b;
    bf16_t* tproj_w = nullptr; float* tproj_b = nullptr;
    ConvW conv_in, conv_out;
    float *ong, *onb;
    struct Level { std::vector<ResW> res; std::vector<TransW> attn; ConvW resample; bool has_resample = false; };
    std::vector<Level> down, up;
    ResW mid0, mid1; TransW mid_attn;
    // debug taps (parity tests): name -> (device f32 NCHW buffer, capacity in bytes); consumed by the next forward
    std::map<std::string, std::pair<float*, size_t>> taps;
    // text-context cache (cross-attention K / V^T per layer, and the bf16 copy of the context)
    // GYRE_CTX_SLOTS independent entries: a hires-fix / graft tree and CFGUNet_Sequential alternate between contexts on
    // every step (reference unet/cfg.py:27-38, unet/hires_fix.py:123-235); gyre_unet_select_context switches without
    // re-projecting.  The fields below alias the CURRENT slot.
    struct CtxSlot { std::vector<CtxKV> kv; void* buf = nullptr; size_t bytes = 0; int B = 0, S = 0; bool valid = false; };
    CtxSlot ctx_slots[GYRE_CTX_SLOTS];
    int cur_slot = 0;
    CtxSlot& cur() { return ctx_slots[cur_slot]; }
    void invalidate_contexts() { for (auto& sl : ctx_slots) sl.valid = false; }
    ~gyre_unet() { for (auto& sl : ctx_slots) if (sl.buf) (void)hipFree(sl.buf); }

    template <typename F> void for_each_cross_attn(F f) {
        for (auto& lv : down) for (auto& t : lv.attn) for (auto& b : t.blocks) f(b.a2);
        for (auto& b : mid_attn.blocks) f(b.a2);
        for (auto& lv : up) for (auto& t : lv.attn) for (auto& b : t.blocks) f(b.a2);
    }
    // Project the text context through every cross-attention to_k / to_v once; the denoising loop then calls
    // forward with ctx == NULL (the context is constant over the 50+ UNet evaluations of a request).
    int set_context(hipStream_t st, const void* ctx, int cdt, int B, int S, int slot = 0) {
        if (B < 1 || S < 1) GYRE_FAIL(GYRE_ERR_INVALID, "unet: empty context");
        if (slot < 0 || slot >= GYRE_CTX_SLOTS) GYRE_FAIL(GYRE_ERR_INVALID, "unet: context slot out of range");
        cur_slot = slot;
        CtxSlot& sl = ctx_slots[slot];
        void*& kv_buf = sl.buf; size_t& kv_bytes = sl.bytes; bool& cache_valid = sl.valid; int &cache_B = sl.B, &cache_S = sl.S;
        std::vector<CtxKV>& kv_cache = sl.kv;
        const int D = cfg.cross_attention_dim, Spad = (S + 7) / 8 * 8;
        size_t need = align_up((size_t)B * S * D * 2, 256);
        for_each_cross_attn([&](AttnW& a) { need += align_up((size_t)B * S * a.c * 2, 256) + align_up((size_t)B * a.c * Spad * 2, 256); });
        cache_valid = false;
        if (need > kv_bytes) {
            if (kv_buf) { GYRE_HIP_CHECK(hipStreamSynchronize(st)); (void)hipFree(kv_buf); kv_buf = nullptr; kv_bytes = 0; }
            GYRE_HIP_CHECK(hipMalloc(&kv_buf, need));
            kv_bytes = need;
        }
        GYRE_HIP_CHECK(hipMemsetAsync(kv_buf, 0, need, st));   // V^T pad columns must be finite
        char* ptr = (char*)kv_buf;
        bf16_t* cx = (bf16_t*)ptr; ptr += align_up((size_t)B * S * D * 2, 256);
        TRY(launch_ctx_to_bf16(st, ctx, cdt, (size_t)B * S * D, cx));
        kv_cache.clear();
        int rc = 0;
        for_each_cross_attn([&](AttnW& a) {
            if (rc) return;
            CtxKV e;
            e.k = (bf16_t*)ptr; ptr += align_up((size_t)B * S * a.c * 2, 256);
            e.vt = (bf16_t*)ptr; ptr += align_up((size_t)B * a.c * Spad * 2, 256);
            GemmParams p;
            p.A = cx; p.lda = D; p.mode = GEMM_LINEAR; p.W = a.wk; p.K = D; p.N = a.c; p.M = B * S; p.bias = a.bk; p.samples = B;
            p.out = e.k; p.ldc = a.c; p.out_mode = OUT_BF16;
            rc = launch_gemm(st, p);
            if (rc) return;
            GemmParams q;
            q.A = cx; q.lda = D; q.mode = GEMM_LINEAR; q.W = a.wv; q.K = D; q.N = a.c; q.M = B * S; q.bias = a.bv; q.samples = B;
            q.out = e.vt; q.out_mode = OUT_BF16_T; q.tokens_per_batch = S; q.ldt = Spad;
            rc = launch_gemm(st, q);
            kv_cache.push_back(e);
        });
        if (rc) return rc;
        cache_B = B; cache_S = S; cache_valid = true;
        return 0;
    }

    int build() {
        const gyre_unet_cfg& c = cfg;
        const int n = c.n_levels;
        if (n < 1 || n > GYRE_MAX_LEVELS) GYRE_FAIL(GYRE_ERR_INVALID, "n_levels out of range");
        for (int i = 0; i < n; ++i) {
            if (c.block_out_channels[i] % c.norm_num_groups || c.block_out_channels[i] % 8)
                GYRE_FAIL(GYRE_ERR_INVALID, "block_out_channels must be multiples of norm_num_groups and 8");
            if (c.attn_levels[i] && (c.block_out_channels[i] % c.num_heads[i] || (c.block_out_channels[i] / c.num_heads[i]) % 8))
                GYRE_FAIL(GYRE_ERR_INVALID, "head dim must be a multiple of 8");
        }
        if (c.cross_attention_dim % 8) GYRE_FAIL(GYRE_ERR_INVALID, "cross_attention_dim must be a multiple of 8");
        ex.groups = c.norm_num_groups;
        ex.store = &store;
        gn_unit = 0;
        for (int i = 0; i < n; ++i) {
            int a = c.block_out_channels[i] / c.norm_num_groups, b = gn_unit;
            while (b) { const int t = a % b; a = b; b = t; }
            gn_unit = a;
        }
        const int c0 = c.block_out_channels[0];
        temb_dim = 4 * c0;
        // total columns of the batched time_emb_proj
        int total_cols = 0;
        for (int i = 0; i < n; ++i) total_cols += c.layers_per_block * c.block_out_channels[i];
        total_cols += 2 * c.block_out_channels[n - 1];
        for (int i = 0; i < n; ++i) total_cols += (c.layers_per_block + 1) * c.block_out_channels[n - 1 - i];
        tproj_w = (bf16_t*)store.dmalloc((size_t)total_cols * temb_dim * 2, true);
        tproj_b = (float*)store.dmalloc((size_t)total_cols * 4, true);
        if (!tproj_w || !tproj_b) GYRE_FAIL(GYRE_ERR_HIP, "hipMalloc failed");

        te1w = store.mat("time_embedding.linear_1", temb_dim, c0, false, true, &te1b);
        te2w = store.mat("time_embedding.linear_2", temb_dim, temb_dim, false, true, &te2b);
        conv_in.cin = pad8(c.in_channels); conv_in.cout = c0;
        conv_in.w = store.conv3("conv_in", c0, c.in_channels, &conv_in.b);
        std::vector<int> skip_ch{c0};
        int cin = c0;
        down.resize(n);
        for (int i = 0; i < n; ++i) {
            const int co = c.block_out_channels[i];
            for (int j = 0; j < c.layers_per_block; ++j) {
                std::string p = "down_blocks." + std::to_string(i);
                down[i].res.emplace_back();
                reg_resnet(store, p + ".resnets." + std::to_string(j), cin, co, true, tproj_w, tproj_b, temb_dim, temb_cols,
                           down[i].res.back());
                cin = co;
                if (c.attn_levels[i]) {
                    down[i].attn.emplace_back();
                    reg_transformer(store, p + ".attentions." + std::to_string(j), cin, c.num_heads[i], c.cross_attention_dim,
                                    c.transformer_depth[i], c.use_linear_projection != 0, down[i].attn.back());
                }
                skip_ch.push_back(cin);
            }
            if (i < n - 1) {
                down[i].has_resample = true;
                down[i].resample.cin = cin; down[i].resample.cout = cin;
                down[i].resample.w = store.conv3("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin,
                                                 &down[i].resample.b);
                skip_ch.push_back(cin);
            }
        }
        reg_resnet(store, "mid_block.resnets.0", cin, cin, true, tproj_w, tproj_b, temb_dim, temb_cols, mid0);
        reg_transformer(store, "mid_block.attentions.0", cin, c.num_heads[n - 1], c.cross_attention_dim,
                        c.transformer_depth[n - 1], c.use_linear_projection != 0, mid_attn);
        reg_resnet(store, "mid_block.resnets.1", cin, cin, true, tproj_w, tproj_b, temb_dim, temb_cols, mid1);
        up.resize(n);
        for (int i = 0; i < n; ++i) {
            const int lvl = n - 1 - i, co = c.block_out_channels[lvl];
            std::string p = "up_blocks." + std::to_string(i);
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                int sk = skip_ch.back(); skip_ch.pop_back();
                up[i].res.emplace_back();
                reg_resnet(store, p + ".resnets." + std::to_string(j), cin + sk, co, true, tproj_w, tproj_b, temb_dim,
                           temb_cols, up[i].res.back());
                cin = co;
                if (c.attn_levels[lvl]) {
                    up[i].attn.emplace_back();
                    reg_transformer(store, p + ".attentions." + std::to_string(j), cin, c.num_heads[lvl],
                                    c.cross_attention_dim, c.transformer_depth[lvl], c.use_linear_projection != 0,
                                    up[i].attn.back());
                }
            }
            if (i < n - 1) {
                up[i].has_resample = true;
                up[i].resample.cin = cin; up[i].resample.cout = cin;
                up[i].resample.w = store.conv3(p + ".upsamplers.0.conv", cin, cin, &up[i].resample.b);
            }
        }
        store.vec("conv_norm_out.weight", cin, &ong); store.vec("conv_norm_out.bias", cin, &onb);
        conv_out.cin = cin; conv_out.cout = c.out_channels;
        conv_out.w = store.conv3("conv_out", c.out_channels, cin, &conv_out.b);
        if (temb_cols != total_cols) GYRE_FAIL(GYRE_ERR_INVALID, "internal: temb column count mismatch");
        for (void* a : store.allocs) if (!a) GYRE_FAIL(GYRE_ERR_HIP, "hipMalloc failed");
        return 0;
    }

    // ControlNet-style residual injection (reference gyre/pipeline/controlnet/unet_patcher.py:30-95, fed by
    // unet/core.py:40-64): down_res[k] (NCHW, shape of the k-th skip connection in the order they are produced, conv_in
    // first) is added to the skip tensor the UP path consumes, mid_res to the mid block's output; the down path and the mid
    // block themselves run on the un-augmented activations.
    int run(bool dry, hipStream_t st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt, int B, int H,
            int W, int S, void* ws, size_t ws_bytes, void* out, int odt, const float* temb_add = nullptr,
            bool use_ctx_cache = false, const void* const* down_res = nullptr, int n_down_res = 0, int rdt = 0,
            const void* mid_res = nullptr, const void* const* adapter = nullptr, int n_adapter = 0, int force_pairs = -1) {
        // T2I-adapter states (reference gyre/pipeline/t2i_adapter/unet_patcher.py:21-86, fed by unet/core.py:212-216): one
        // NCHW tensor per down level, added IN PLACE to the level's last hidden state - just before its downsampler for the
        // cross-attention levels (so the skip connection taken there, the downsampler and everything after see it), after
        // the whole level (downsampler included) otherwise.
        if (adapter && n_adapter != cfg.n_levels)
            GYRE_FAIL(GYRE_ERR_INVALID, "unet: one adapter state per down level expected (" + std::to_string(cfg.n_levels) + ")");
        const gyre_unet_cfg& c = cfg;
        const int n = c.n_levels;
        if (B < 1 || H < 1 || W < 1 || S < 1) GYRE_FAIL(GYRE_ERR_INVALID, "unet: empty batch / image / context");
        ex.arena.reset((char*)ws, ws_bytes, dry);
        vjp.valid = false;                           // the arena is being reused: a pending reverse sweep has lost its activations
        ex.st = st; ex.batch = B;
        ex.cs_unit = gn_unit;                        // producers leave GroupNorm statistics where they can (attach_colstats)
        Exec& e = ex;
        const int D = c.cross_attention_dim;
        const bool cached = use_ctx_cache;
        if (cached && (!cur().valid || cur().B != B || cur().S != S))
            GYRE_FAIL(GYRE_ERR_INVALID, "unet: ctx == NULL needs a gyre_unet_set_context call with the same B and S");
        e.ctx_cache = cached ? &cur().kv : nullptr;
        e.ctx_layer = 0;
        // Every sample at the same timestep (what the samplers pass: one scalar per call) and no per-sample added
        // conditioning: the time-embedding MLP and the batched time_emb_proj run for ONE row and every resnet reads it with a
        // zero row stride (GemmParams::ld_rowbias) - same arithmetic per row, bit-identical results, 1/B of the work.
        const bool uni_t = hint_uniform_t && !temb_add;
        hint_uniform_t = false;
        const int Bt = uni_t ? 1 : B, ld_tp = uni_t ? 0 : temb_cols;
        // CFG-parallel call (reference unet/cfg.py:49-57: latents cat[x, x], timesteps cat[t, t], contexts cat[uncond, cond]):
        // samples b and b + B/2 are the SAME up to the first cross-attention, so conv_in, the first resnet and the first
        // transformer's GroupNorm / proj_in / self-attention run on B/2 samples and their results are written twice
        // (Exec::transformer x_full).  Exact - the kernels do not depend on batch position - and only on the caller's word
        // (gyre_unet_hint_cfg_pairs).  Not with ControlNet / T2I inputs (they differ per half), SDXL's per-sample added
        // conditioning, ToMe (its merge runs inside the self-attention on whatever batch it gets: fine, but keep the
        // measured configurations simple) or the debug taps.
        const bool pairs_ok = B % 2 == 0 && !down_res && !mid_res && !adapter && !temb_add && c.attn_levels[0] &&
                              c.layers_per_block >= 1 && c.transformer_depth[0] >= 1 && taps.empty();
        const bool pairs = pairs_ok && (force_pairs >= 0 ? force_pairs != 0 : hint_cfg_pairs);
        hint_cfg_pairs = false;
        const int Bp = pairs ? B / 2 : B;              // batch of the shared prefix
        // Both hints are the caller's word and a wrong one gives wrong activations, silently.  GYRE_VERIFY_HINTS=1 (debug /
        // integration runs of a foreign C-ABI caller) checks them on the device before they are used - one compare kernel and
        // one 4-byte read-back, i.e. a stream synchronisation per call - and fails the call instead.
        if (!dry && (uni_t || pairs)) {
            static const bool verify = getenv("GYRE_VERIFY_HINTS") != nullptr && getenv("GYRE_VERIFY_HINTS")[0] != '0';
            if (verify) {
                if (!hint_flag) hint_flag = (int*)store.dmalloc(16, true);
                if (!hint_flag) GYRE_FAIL(GYRE_ERR_HIP, "cannot allocate the hint-verification flag");
                if (hipMemsetAsync(hint_flag, 0, 4, st) != hipSuccess) GYRE_FAIL(GYRE_ERR_HIP, "hipMemsetAsync failed");
                const size_t elt = xdt == 0 ? 4 : 2;
                TRY(launch_verify_hints(st, t, B, uni_t ? 1 : 0, x, (size_t)(B / 2) * c.in_channels * H * W * elt, pairs ? 1 : 0, hint_flag));
                int bad = 0;
                if (hipMemcpyAsync(&bad, hint_flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
                    GYRE_FAIL(GYRE_ERR_HIP, "reading the hint-verification flag failed");
                if (bad & 1) GYRE_FAIL(GYRE_ERR_INVALID, "gyre_unet_hint_uniform_timestep was set, but the timesteps of this call differ between samples");
                if (bad & 2) GYRE_FAIL(GYRE_ERR_INVALID, "gyre_unet_hint_cfg_pairs was set, but sample b and sample b + B/2 of this call differ in their latents");
            }
        }
        Tn xin, cx, emb, t1, t2, tp;
        TRY(e.alloc(xin, Bp, H, W, pad8(c.in_channels)));
        if (!cached) TRY(e.alloc(cx, B, S, 1, D));
        TRY(e.alloc(emb, B, 1, 1, c.block_out_channels[0], 4));
        TRY(e.alloc(t1, B, 1, 1, temb_dim, 4));
        TRY(e.alloc(t2, B, 1, 1, temb_dim, 4));
        TRY(e.alloc(tp, B, 1, 1, temb_cols, 4));
        if (!dry) {
            TRY(launch_nchw_to_nhwc(st, x, xdt, Bp, c.in_channels, H * W, xin.C, xin.p));
            if (!cached) TRY(launch_ctx_to_bf16(st, ctx, cdt, (size_t)B * S * D, cx.p));
            TRY(launch_timestep_linear(st, t, Bt, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift, (float*)emb.p, te1w, te1b,
                                       temb_dim, (float*)t1.p, temb_dim));
            TRY(launch_rowvec_linear(st, (float*)t1.p, Bt, temb_dim, te2w, te2b, temb_dim, 1, (float*)t2.p, temb_dim));
            // SDXL-style added conditioning (text_time): emb = time_embedding(t) + aug_emb, aug_emb from the host
            if (temb_add) TRY(launch_add_f32(st, (float*)t2.p, temb_add, (size_t)B * temb_dim));
            // every resnet's Linear(SiLU(temb)) in one launch
            TRY(launch_rowvec_linear(st, (float*)t2.p, Bt, temb_dim, tproj_w, tproj_b, temb_cols, 1, (float*)tp.p, temb_cols));
        }
        e.free(emb); e.free(t1); e.free(t2);
        const float* tproj = (const float*)tp.p;

        auto tap = [&](const std::string& name, const Tn& t, int channels) -> int {
            if (dry) return 0;
            auto it = taps.find(name);
            if (it == taps.end()) return 0;
            const size_t need = (size_t)t.B * channels * t.H * t.W * sizeof(float);
            if (it->second.second < need) GYRE_FAIL(GYRE_ERR_INVALID, "tap buffer too small for " + name);
            return launch_nhwc_to_nchw_f32(st, t.p, t.B, channels, t.H * t.W, t.C, it->second.first);
        };
        std::vector<Tn> skips;
        Tn h;
        e.batch = Bp;
        TRY(e.conv3(xin, conv_in, 1, 1, 0, nullptr, 0, nullptr, h));
        e.free(xin);
        if (pairs) {                                   // the skip connection the up path reads holds all B samples
            Tn hd;
            TRY(e.dup_batch(h, hd));
            skips.push_back(hd);
        } else {
            skips.push_back(h);
        }
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < c.layers_per_block; ++j) {
                Tn r;
                if (pairs && i == 0 && j == 0) {
                    // shared prefix: first resnet on the half batch (h = conv_in's half-batch output), first transformer up to
                    // its first cross-attention; the transformer hands back the doubled resnet output (its residual)
                    Tn rh, a, rfull;
                    TRY(e.resnet(h, nullptr, down[0].res[0], tproj, ld_tp, 1e-5f, rh));
                    e.free(h);
                    TRY(e.transformer(rh, cx, S, D, down[0].attn[0], a, nullptr, &rfull));
                    e.free(rh); e.free(rfull);
                    e.batch = B;
                    skips.push_back(a);
                    continue;
                }
                TRY(e.resnet(skips.back(), nullptr, down[i].res[j], tproj, ld_tp, 1e-5f, r));
                if (c.attn_levels[i]) {
                    Tn a;
                    TRY(e.transformer(r, cx, S, D, down[i].attn[j], a));
                    e.free(r); r = a;
                }
                skips.push_back(r);
            }
            const bool adapt_before = adapter && (c.attn_levels[i] || !down[i].has_resample);
            auto adapt = [&]() -> int {
                Tn& s_ = skips.back();
                s_.cs_chunks = 0;                      // updated in place: its producer's statistics no longer describe it
                if (dry) return 0;
                if (!adapter[i]) GYRE_FAIL(GYRE_ERR_INVALID, "unet: null adapter state");
                return launch_add_nchw_into_nhwc(st, adapter[i], rdt, B, s_.C, s_.H * s_.W, s_.C, s_.p);
            };
            if (adapt_before) TRY(adapt());
            if (down[i].has_resample) {
                Tn d;
                TRY(e.conv3(skips.back(), down[i].resample, 2, 1, 0, nullptr, 0, nullptr, d));
                skips.push_back(d);
            }
            if (adapter && !adapt_before) TRY(adapt());
            TRY(tap("down" + std::to_string(i), skips.back(), c.block_out_channels[i]));
        }
        {
            Tn a, b2;
            TRY(e.resnet(skips.back(), nullptr, mid0, tproj, ld_tp, 1e-5f, a));
            TRY(e.transformer(a, cx, S, D, mid_attn, b2));
            e.free(a);
            TRY(e.resnet(b2, nullptr, mid1, tproj, ld_tp, 1e-5f, h));
            e.free(b2);
            TRY(tap("mid", h, c.block_out_channels[n - 1]));
        }
        if (down_res) {
            if (n_down_res != (int)skips.size())
                GYRE_FAIL(GYRE_ERR_INVALID, "unet: " + std::to_string(skips.size()) + " down-block residuals expected, got " +
                          std::to_string(n_down_res));
            for (auto& sk : skips) sk.cs_chunks = 0;   // updated in place below: the producers' statistics are stale
            if (!dry)
                for (size_t k = 0; k < skips.size(); ++k) {
                    if (!down_res[k]) GYRE_FAIL(GYRE_ERR_INVALID, "unet: null down-block residual");
                    const Tn& sk = skips[k];
                    TRY(launch_add_nchw_into_nhwc(st, down_res[k], rdt, B, sk.C, sk.H * sk.W, sk.C, sk.p));
                }
        }
        if (mid_res) h.cs_chunks = 0;
        if (mid_res && !dry) TRY(launch_add_nchw_into_nhwc(st, mid_res, rdt, B, h.C, h.H * h.W, h.C, h.p));
        for (int i = 0; i < n; ++i) {
            const int lvl = n - 1 - i;
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                Tn sk = skips.back(); skips.pop_back();
                Tn r;
                TRY(e.resnet(h, &sk, up[i].res[j], tproj, ld_tp, 1e-5f, r));
                e.free(h); e.free(sk);
                if (c.attn_levels[lvl]) {
                    Tn a;
                    TRY(e.transformer(r, cx, S, D, up[i].attn[j], a));
                    e.free(r); r = a;
                }
                h = r;
            }
            if (up[i].has_resample) {
                Tn u;   // resize to the next skip connection's size (= 2x unless the latent size is odd at this level)
                TRY(e.conv3(h, up[i].resample, 1, 1, 1, nullptr, 0, nullptr, u, skips.back().H, skips.back().W));
                e.free(h); h = u;
            }
            TRY(tap("up" + std::to_string(i), h, c.block_out_channels[lvl]));
        }
        if (!dry) taps.clear();
        Tn a;
        TRY(e.groupnorm(h, nullptr, ong, onb, 1e-5f, 1, a));
        e.free(h);
        TRY(e.conv3_nchw(a, conv_out, out, odt));
        e.free(a); e.free(tp); e.free(cx);
        return 0;
    }
};

// ------------------------------------------------------------------------------------------
// VAE
// ------------------------------------------------------------------------------------------
struct gyre_vae {
    gyre_vae_cfg cfg;
    Store store;
    Exec ex;
    struct Blk { std::vector<ResW> res; ConvW resample; bool has_resample = false; };
    // encoder
    ConvW e_in, e_out; std::vector<Blk> e_down; ResW e_mid0, e_mid1; AttnW e_attn; float *e_ag, *e_ab, *e_ng, *e_nb;
    bf16_t* quant_w; float* quant_b;
    // decoder
    ConvW d_in, d_out; std::vector<Blk> d_up; ResW d_mid0, d_mid1; AttnW d_attn; float *d_ag, *d_ab, *d_ng, *d_nb;
    bf16_t* pq_w; float* pq_b;

    void reg_vattn(const std::string& p, int c, AttnW& w, float** g, float** b) {
        w.c = c; w.heads = 1; w.kv_dim = c;
        store.vec(p + ".group_norm.weight", c, g); store.vec(p + ".group_norm.bias", c, b);
        w.wqk = (bf16_t*)store.dmalloc((size_t)2 * c * c * 2, true);
        w.bqk = (float*)store.dmalloc((size_t)2 * c * 4, true);
        store.add(p + ".query.weight", {c, c}, PK_MAT, w.wqk, c, c);
        store.add(p + ".query.bias", {c}, PK_VEC, w.bqk);
        store.add(p + ".key.weight", {c, c}, PK_MAT, w.wqk + (size_t)c * c, c, c);
        store.add(p + ".key.bias", {c}, PK_VEC, w.bqk + c);
        w.wv = store.mat(p + ".value", c, c, false, true, &w.bv);
        w.wo = store.mat(p + ".proj_attn", c, c, false, true, &w.bo);
    }
    int build() {
        const gyre_vae_cfg& c = cfg;
        const int n = c.n_levels;
        if (n < 1 || n > GYRE_MAX_LEVELS) GYRE_FAIL(GYRE_ERR_INVALID, "n_levels out of range");
        for (int i = 0; i < n; ++i)
            if (c.block_out_channels[i] % c.norm_num_groups || c.block_out_channels[i] % 8)
                GYRE_FAIL(GYRE_ERR_INVALID, "block_out_channels must be multiples of norm_num_groups and 8");
        ex.groups = c.norm_num_groups;
        ex.store = &store;
        int dummy = 0;
        const int z = c.latent_channels;
        // ---- encoder ----
        int cin = c.block_out_channels[0];
        e_in.cin = pad8(c.in_channels); e_in.cout = cin;
        e_in.w = store.conv3("encoder.conv_in", cin, c.in_channels, &e_in.b);
        e_down.resize(n);
        for (int i = 0; i < n; ++i) {
            const int co = c.block_out_channels[i];
            std::string p = "encoder.down_blocks." + std::to_string(i);
            for (int j = 0; j < c.layers_per_block; ++j) {
                e_down[i].res.emplace_back();
                reg_resnet(store, p + ".resnets." + std::to_string(j), cin, co, false, nullptr, nullptr, 0, dummy,
                           e_down[i].res.back());
                cin = co;
            }
            if (i < n - 1) {
                e_down[i].has_resample = true;
                e_down[i].resample.cin = cin; e_down[i].resample.cout = cin;
                e_down[i].resample.w = store.conv3(p + ".downsamplers.0.conv", cin, cin, &e_down[i].resample.b);
            }
        }
        reg_resnet(store, "encoder.mid_block.resnets.0", cin, cin, false, nullptr, nullptr, 0, dummy, e_mid0);
        reg_vattn("encoder.mid_block.attentions.0", cin, e_attn, &e_ag, &e_ab);
        reg_resnet(store, "encoder.mid_block.resnets.1", cin, cin, false, nullptr, nullptr, 0, dummy, e_mid1);
        store.vec("encoder.conv_norm_out.weight", cin, &e_ng); store.vec("encoder.conv_norm_out.bias", cin, &e_nb);
        e_out.cin = cin; e_out.cout = 2 * z;
        e_out.w = store.conv3("encoder.conv_out", 2 * z, cin, &e_out.b);
        quant_w = store.mat("quant_conv", 2 * z, 2 * z, true, true, &quant_b);
        // ---- decoder ----
        pq_w = store.mat("post_quant_conv", z, z, true, true, &pq_b);
        cin = c.block_out_channels[n - 1];
        d_in.cin = pad8(z); d_in.cout = cin;
        d_in.w = store.conv3("decoder.conv_in", cin, z, &d_in.b);
        reg_resnet(store, "decoder.mid_block.resnets.0", cin, cin, false, nullptr, nullptr, 0, dummy, d_mid0);
        reg_vattn("decoder.mid_block.attentions.0", cin, d_attn, &d_ag, &d_ab);
        reg_resnet(store, "decoder.mid_block.resnets.1", cin, cin, false, nullptr, nullptr, 0, dummy, d_mid1);
        d_up.resize(n);
        for (int i = 0; i < n; ++i) {
            const int co = c.block_out_channels[n - 1 - i];
            std::string p = "decoder.up_blocks." + std::to_string(i);
            for (int j = 0; j < c.layers_per_block + 1; ++j) {
                d_up[i].res.emplace_back();
                reg_resnet(store, p + ".resnets." + std::to_string(j), cin, co, false, nullptr, nullptr, 0, dummy,
                           d_up[i].res.back());
                cin = co;
            }
            if (i < n - 1) {
                d_up[i].has_resample = true;
                d_up[i].resample.cin = cin; d_up[i].resample.cout = cin;
                d_up[i].resample.w = store.conv3(p + ".upsamplers.0.conv", cin, cin, &d_up[i].resample.b);
            }
        }
        store.vec("decoder.conv_norm_out.weight", cin, &d_ng); store.vec("decoder.conv_norm_out.bias", cin, &d_nb);
        d_out.cin = cin; d_out.cout = c.out_channels;
        d_out.w = store.conv3("decoder.conv_out", c.out_channels, cin, &d_out.b);
        for (void* a : store.allocs) if (!a) GYRE_FAIL(GYRE_ERR_HIP, "hipMalloc failed");
        return 0;
    }
    // GN -> fused QK / V^T projections -> single-head attention -> proj + residual
    int vattn(Exec& e, const Tn& x, const AttnW& w, const float* g, const float* b, Tn& out) {
        Tn a;
        TRY(e.groupnorm(x, nullptr, g, b, 1e-6f, 0, a));
        TRY(e.mha(a, false, nullptr, 0, 0, w, x, out));
        e.free(a);
        return 0;
    }
    int mid(Exec& e, Tn& h, const ResW& r0, const AttnW& aw, const float* g, const float* b, const ResW& r1) {
        Tn a, b2, c2;
        TRY(e.resnet(h, nullptr, r0, nullptr, 0, 1e-6f, a)); e.free(h);
        TRY(vattn(e, a, aw, g, b, b2)); e.free(a);
        TRY(e.resnet(b2, nullptr, r1, nullptr, 0, 1e-6f, c2)); e.free(b2);
        h = c2;
        return 0;
    }
    int run_encode(bool dry, hipStream_t st, const void* img, int idt, int B, int H, int W, void* ws, size_t wsb,
                   void* out, int odt) {
        const int n = cfg.n_levels;
        if (B < 1 || H < 1 || W < 1) GYRE_FAIL(GYRE_ERR_INVALID, "vae.encode: empty input");
        if ((H % (1 << (n - 1))) || (W % (1 << (n - 1)))) GYRE_FAIL(GYRE_ERR_INVALID, "vae.encode: H, W must be multiples of 8");
        ex.arena.reset((char*)ws, wsb, dry); ex.st = st; ex.batch = B; ex.cs_unit = 0;
        Exec& e = ex;
        Tn x, h;
        TRY(e.alloc(x, B, H, W, pad8(cfg.in_channels)));
        if (!dry) TRY(launch_nchw_to_nhwc(st, img, idt, B, cfg.in_channels, H * W, x.C, x.p));
        TRY(e.conv3(x, e_in, 1, 1, 0, nullptr, 0, nullptr, h)); e.free(x);
        for (int i = 0; i < n; ++i) {
            for (auto& rw : e_down[i].res) {
                Tn r; TRY(e.resnet(h, nullptr, rw, nullptr, 0, 1e-6f, r)); e.free(h); h = r;
            }
            if (e_down[i].has_resample) {  // pad (0,1,0,1) + stride-2 conv, pad 0
                Tn d; TRY(e.conv3(h, e_down[i].resample, 2, 0, 0, nullptr, 0, nullptr, d)); e.free(h); h = d;
            }
        }
        TRY(mid(e, h, e_mid0, e_attn, e_ag, e_ab, e_mid1));
        Tn a, m;
        TRY(e.groupnorm(h, nullptr, e_ng, e_nb, 1e-6f, 1, a)); e.free(h);
        TRY(e.conv3(a, e_out, 1, 1, 0, nullptr, 0, nullptr, m)); e.free(a);
        if (!dry) {
            GemmParams p;
            p.A = m.p; p.lda = m.C; p.mode = GEMM_LINEAR; p.W = quant_w; p.K = m.C; p.N = 2 * cfg.latent_channels;
            p.M = m.rows(); p.bias = quant_b; p.rows_per_sample = m.H * m.W;
            p.out = out; p.out_mode = OUT_NCHW; p.out_dtype = odt;
            TRY(launch_gemm(st, p));
        }
        e.free(m);
        return 0;
    }
    int run_decode(bool dry, hipStream_t st, const void* z, int idt, int B, int h_, int w_, void* ws, size_t wsb,
                   void* out, int odt) {
        const int n = cfg.n_levels;
        if (B < 1 || h_ < 1 || w_ < 1) GYRE_FAIL(GYRE_ERR_INVALID, "vae.decode: empty input");
        ex.arena.reset((char*)ws, wsb, dry); ex.st = st; ex.batch = B; ex.cs_unit = 0;
        Exec& e = ex;
        Tn x, q, h;
        const int zc = pad8(cfg.latent_channels);
        TRY(e.alloc(x, B, h_, w_, zc));
        if (!dry) TRY(launch_nchw_to_nhwc(st, z, idt, B, cfg.latent_channels, h_ * w_, zc, x.p));
        TRY(e.alloc(q, B, h_, w_, zc));
        TRY(e.linear(x.p, zc, nullptr, 0, 0, x.rows(), zc, pq_w, zc, pq_b, nullptr, 0, 0, q.p, zc)); e.free(x);
        TRY(e.conv3(q, d_in, 1, 1, 0, nullptr, 0, nullptr, h)); e.free(q);
        TRY(mid(e, h, d_mid0, d_attn, d_ag, d_ab, d_mid1));
        for (int i = 0; i < n; ++i) {
            for (auto& rw : d_up[i].res) {
                Tn r; TRY(e.resnet(h, nullptr, rw, nullptr, 0, 1e-6f, r)); e.free(h); h = r;
            }
            if (d_up[i].has_resample) {
                Tn u; TRY(e.conv3(h, d_up[i].resample, 1, 1, 1, nullptr, 0, nullptr, u)); e.free(h); h = u;
            }
        }
        Tn a;
        TRY(e.groupnorm(h, nullptr, d_ng, d_nb, 1e-6f, 1, a)); e.free(h);
        TRY(e.conv3_nchw(a, d_out, out, odt)); e.free(a);
        return 0;
    }
};

// model_vjp.hip: forward + reverse sweep in one call (dry = size the workspace only)
int gyre_unet_run_vjp(gyre_unet& u, bool dry, hipStream_t st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt,
                      int B, int H, int W, int S, const void* d_eps, int ddt, void* ws, size_t ws_bytes, void* eps_out, int odt,
                      void* dx_out, int dxdt, const float* temb_add);
int gyre_unet_vjp_forward(gyre_unet& u, bool dry, hipStream_t st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt,
                          int B, int H, int W, int S, void* ws, size_t ws_bytes, void* eps_out, int odt, const float* temb_add);
int gyre_unet_vjp_reverse(gyre_unet& u, hipStream_t st, const void* d_eps, int ddt, void* dx_out, int dxdt, int b0, int nb);
int gyre_vae_run_decode_vjp(gyre_vae& v, bool dry, hipStream_t st, const void* z, int zdt, int B, int h_, int w_, const void* d_img,
                            int ddt, void* ws, size_t wsb, void* img_out, int odt, void* dz_out, int dzdt);
