// Input-gradient graphs of the UNet and the VAE decoder (vector-Jacobian products with respect to the sample).
//
// The reference's CLIP-guided mode differentiates a CLIP loss with respect to the latents through the guided UNet
// evaluation and the VAE decoder (gyre/pipeline/unet/clipguided.py:301-338 model_k_fn / model_d_fn, :340-420 cond_fn,
// `torch.autograd.grad(loss, latents)`).  Weights get no gradient, so the backward graph is the chain of transposed
// layers only.  One call = forward pass that keeps what the adjoints need (block inputs, conv1 outputs, LayerNorm
// outputs, attention outputs) + the reverse sweep; nothing survives the call, so the C ABI stays stateless:
//   gyre_unet_vjp(x, t, ctx, d_eps)        -> eps, d_x
//   gyre_vae_decode_vjp(z, d_image)        -> image, d_z
// Transposed weights (W^T for linears, rotated [Cin][3][3][Cout] for convs) are made on first use and kept next to the forward
// weights (Store::wt_cache, +1.7 GB for SD1.5); every gyre_*_set_weight bumps a version, so a per-request LoRA re-upload
// refreshes them at the next sweep.  Bare op calls (no model handle) transpose into the workspace instead.
#include "model_impl.h"

namespace {

// W^T [K][N] of a linear weight W [N][K]: from the owner's cache (transposed once per weight upload) or, for bare op calls,
// into the arena (tmp is then valid and must be freed by the caller)
int weight_t_linear(Exec& e, const bf16_t* w, int N, int K, Tn& tmp, const bf16_t** out) {
    if (e.store) {
        if (e.dry()) { *out = nullptr; return 0; }
        bool fresh = false;
        bf16_t* p = e.store->wt_lookup(w, (size_t)K * N * 2, &fresh);
        if (!p) GYRE_FAIL(GYRE_ERR_HIP, "hipMalloc failed (transposed weight cache)");
        if (!fresh) TRY(launch_transpose(e.st, w, K, N, K, p, N, 1, 0, 0));
        *out = p;
        return 0;
    }
    TRY(e.alloc(tmp, 1, 1, K, N));
    if (!e.dry()) TRY(launch_transpose(e.st, w, K, N, K, tmp.p, N, 1, 0, 0));
    *out = tmp.p;
    return 0;
}
// rotated conv weight [cin_pad][3][3][co] of W [co][3][3][cin_pad], same policy
int weight_t_conv(Exec& e, const bf16_t* w, int co, int cin_pad, Tn& tmp, const bf16_t** out) {
    if (e.store) {
        if (e.dry()) { *out = nullptr; return 0; }
        bool fresh = false;
        bf16_t* p = e.store->wt_lookup(w, (size_t)cin_pad * 9 * co * 2, &fresh);
        if (!p) GYRE_FAIL(GYRE_ERR_HIP, "hipMalloc failed (transposed weight cache)");
        if (!fresh) TRY(launch_conv_weight_t(e.st, w, co, cin_pad, p));
        *out = p;
        return 0;
    }
    TRY(e.alloc(tmp, 1, cin_pad, 9, co));
    if (!e.dry()) TRY(launch_conv_weight_t(e.st, w, co, cin_pad, tmp.p));
    *out = tmp.p;
    return 0;
}

// dx[M][K] = dy[M][N] W[N][K]  (+ addend)
int linear_bwd(Exec& e, const bf16_t* dy, int ldy, int M, int N, const bf16_t* w, int K, const bf16_t* addend, int lda,
               bf16_t* dx, int ldx) {
    Tn tmp; const bf16_t* wt;
    TRY(weight_t_linear(e, w, N, K, tmp, &wt));
    int rc = e.linear(dy, ldy, nullptr, 0, 0, M, N, wt, K, nullptr, addend, lda, 0, dx, ldx);
    e.free(tmp);
    return rc;
}

// adjoint of Exec::conv3 with respect to its input.  (Hx, Wx): spatial size of the forward input x.
int conv3_bwd(Exec& e, const Tn& dy, const ConvW& w, int cin_pad, int stride, int pad, int ups, int Hx, int Wx, const Tn* addend,
              Tn& dx) {
    if (stride == 2 && !pad) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "vjp: stride-2 conv without padding (VAE encoder) has no adjoint here");
    const int co = dy.C;                       // pad8(cout)
    Tn wt; const bf16_t* wtp;
    TRY(weight_t_conv(e, w.w, co, cin_pad, wt, &wtp));
    ConvW t{const_cast<bf16_t*>(wtp), nullptr, co, cin_pad};
    if (stride == 2) {
        Tn dz;
        TRY(e.alloc(dz, dy.B, Hx, Wx, co));
        if (!e.dry()) TRY(launch_zero_stuff2(e.st, dy.p, dy.B, dy.H, dy.W, Hx, Wx, co, dz.p));
        TRY(e.conv3(dz, t, 1, 1, 0, nullptr, 0, addend, dx));
        e.free(dz);
    } else if (ups) {
        Tn du;
        TRY(e.conv3(dy, t, 1, 1, 0, nullptr, 0, nullptr, du));
        TRY(e.alloc(dx, dy.B, Hx, Wx, cin_pad));
        if (!e.dry()) {
            TRY(launch_pool2_sum(e.st, du.p, dy.B, Hx, Wx, dy.H, dy.W, cin_pad, dx.p));
            if (addend) TRY(launch_add_bf16(e.st, dx.p, addend->p, (size_t)dx.rows() * dx.C));
        }
        e.free(du);
    } else {
        TRY(e.conv3(dy, t, 1, 1, 0, nullptr, 0, addend, dx));
    }
    e.free(wt);
    return 0;
}

// adjoint of Exec::groupnorm: x (|| x2) forward input, dy gradient of the output; addend is added to dx (first source only)
int groupnorm_bwd(Exec& e, const Tn& x, const Tn* x2, const float* g, const float* b, float eps, int silu, const Tn& dy,
                  const Tn* addend, Tn& dx, Tn* dx2) {
    const int C = x.C + (x2 ? x2->C : 0), HW = x.H * x.W;
    Tn ws;
    TRY(e.alloc_raw(ws, gn_bwd_workspace_bytes(x.B, HW, C, e.groups)));
    TRY(e.alloc(dx, x.B, x.H, x.W, x.C));
    if (x2) TRY(e.alloc(*dx2, x.B, x.H, x.W, x2->C));
    if (!e.dry()) {
        GnBwdParams p;
        p.x = x.p; p.x2 = x2 ? x2->p : nullptr; p.C1 = x.C; p.B = x.B; p.HW = HW; p.C = C; p.G = e.groups;
        p.gamma = g; p.beta = b; p.eps = eps; p.silu = silu; p.dy = dy.p; p.addend = addend ? addend->p : nullptr;
        p.dx = dx.p; p.dx2 = x2 ? dx2->p : nullptr;
        TRY(launch_groupnorm_bwd(e.st, p, ws.p));
    }
    e.free(ws);
    return 0;
}

int layernorm_bwd(Exec& e, const Tn& x, const float* g, const Tn& dy, const Tn* addend, Tn& dx) {
    TRY(e.alloc(dx, x.B, x.H, x.W, x.C));
    if (e.dry()) return 0;
    return launch_layernorm_bwd(e.st, x.p, dy.p, x.rows(), x.C, g, 1e-5f, addend ? addend->p : nullptr, dx.p);
}

// adjoint of Exec::mha with respect to xq; the residual input simply receives d_out (handled by the caller).
// Q / K / V are re-projected (three cheap GEMMs) instead of kept; the attention output comes from the forward pass.
int mha_bwd(Exec& e, const Tn& xq, bool cross, const bf16_t* kvsrc, int S, int kv_dim, const AttnW& w, const MhaSave& sv,
            const Tn& d_out, Tn& d_xq) {
    const int B = xq.B, Nq = xq.H * xq.W, C = w.c, D = C / w.heads, M = B * Nq;
    const int Nk = cross ? S : Nq;
    const int ldkt = (Nk + 31) / 32 * 32, ldqt = (Nq + 31) / 32 * 32;
    const bool need_t = attn_bwd_needs_transposes(D);    // only the register-streaming kernels (D > 160: the VAE block) read K^T / Q^T / dO^T
    Tn d_ao, qkv, kc, vc, kt, qt, dot, dqkv, dv, stats;
    TRY(e.alloc(d_ao, B, xq.H, xq.W, C));
    TRY(linear_bwd(e, d_out.p, C, M, C, w.wo, C, nullptr, 0, d_ao.p, C));
    AttnBwdParams a{};
    a.B = B; a.H = w.heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.k_prescaled = w.k_prescaled;
    a.o = sv.ao.p; a.ldo = C; a.d_o = d_ao.p; a.lddo = C;
    TRY(e.alloc_raw(stats, attn_bwd_stats_bytes(B, w.heads, Nq)));
    a.stats = stats.p;
    TRY(e.alloc(kt, B, C, 1, ldkt));
    a.kt = kt.p; a.ldkt = ldkt;
    TRY(e.alloc(d_xq, B, xq.H, xq.W, C));
    if (cross) {
        const bool kept_q = sv.qk.valid() && sv.qk.C == C;            // the forward pass kept to_q's output
        if (!kept_q) {
            TRY(e.alloc(qkv, B, xq.H, xq.W, C));
            TRY(e.linear(xq.p, C, nullptr, 0, 0, M, C, w.wq, C, w.bq, nullptr, 0, 0, qkv.p, C));
        }
        TRY(e.alloc(kc, B, Nk, 1, C));
        TRY(e.linear(kvsrc, kv_dim, nullptr, 0, 0, B * Nk, kv_dim, w.wk, C, w.bk, nullptr, 0, 0, kc.p, C));
        TRY(e.alloc(vc, B, Nk, 1, C));
        TRY(e.linear(kvsrc, kv_dim, nullptr, 0, 0, B * Nk, kv_dim, w.wv, C, w.bv, nullptr, 0, 0, vc.p, C));
        TRY(e.alloc(dqkv, B, xq.H, xq.W, C));
        a.q = kept_q ? sv.qk.p : qkv.p; a.ldq = C; a.k = kc.p; a.ldk = C; a.v = vc.p; a.ldv = C;
        a.dq = dqkv.p; a.lddq = C; a.dk = nullptr; a.dv = nullptr; a.qt = nullptr; a.d_ot = nullptr; a.ldqt = ldqt;
        if (!e.dry()) {
            if (need_t) TRY(launch_transpose(e.st, kc.p, C, Nk, C, kt.p, ldkt, B, (size_t)Nk * C, (size_t)C * ldkt));
            TRY(launch_attention_bwd(e.st, a));
        }
        TRY(linear_bwd(e, dqkv.p, C, M, C, w.wq, C, nullptr, 0, d_xq.p, C));
    } else {
        const int W3 = w.qkv_fused ? 3 * C : 2 * C;     // Q | K (| V) in one row
        const int tr = (e.tome_r > 0 && Nq % 16 == 0) ? tome_effective_r(Nq, e.tome_r) : 0;   // same rule as Exec::mha
        if (tr > 0 && !w.qkv_fused) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "vjp: token merging needs the fused Q|K|V projection");
        const bool kept = tr > 0 && sv.qk.valid() && sv.km.valid() && sv.vm.valid() && sv.idx.valid();   // the forward pass kept Q|K and the merge
        if (!kept) {
            TRY(e.alloc(qkv, B, xq.H, xq.W, W3));
            TRY(e.linear(xq.p, C, nullptr, 0, 0, M, C, w.wqk, W3, w.qkv_fused ? nullptr : w.bqk, nullptr, 0, 0, qkv.p, W3));
        }
        const bf16_t* vrows = nullptr; int ldv = W3;
        if (kept) {}
        else if (w.qkv_fused) { vrows = e.dry() ? nullptr : qkv.p + 2 * C; ldv = W3; }
        else {
            TRY(e.alloc(vc, B, xq.H, xq.W, C));
            TRY(e.linear(xq.p, C, nullptr, 0, 0, M, C, w.wv, C, w.bv, nullptr, 0, 0, vc.p, C));
            vrows = vc.p; ldv = C;
        }
        TRY(e.alloc(qt, B, C, 1, ldqt));
        TRY(e.alloc(dot, B, C, 1, ldqt));
        TRY(e.alloc(dqkv, B, xq.H, xq.W, W3));
        if (!w.qkv_fused) TRY(e.alloc(dv, B, xq.H, xq.W, C));
        a.q = kept ? sv.qk.p : qkv.p; a.ldq = kept ? 2 * C : W3; a.qt = qt.p; a.d_ot = dot.p; a.ldqt = ldqt;
        a.dq = dqkv.p; a.lddq = W3;
        if (kept) {
            const int nout = Nq - tr, half = Nq / 2, ldkm = (nout + 31) / 32 * 32;
            if (need_t) GYRE_FAIL(GYRE_ERR_UNSUPPORTED, "vjp: token merging with a head dim above 160");
            Tn dkm, dvm;
            TRY(e.alloc(dkm, B, nout, 1, C));
            TRY(e.alloc(dvm, B, nout, 1, C));
            if (!e.dry()) {
                // (index arrays laid out for the forward pass's batch; a narrowed sweep starts at its first sample)
                const size_t seg = ((size_t)(sv.idx_B ? sv.idx_B : B) * half + 63) & ~(size_t)63;
                int* order = (int*)sv.idx.p + (size_t)sv.idx_b0 * half;
                int* dstl = order + seg;
                int* inv = dstl + seg;
                a.k = sv.km.p; a.ldk = C; a.v = sv.vm.p; a.ldv = C; a.Nk = nout; a.kt = nullptr; a.ldkt = ldkm;
                a.dk = dkm.p; a.lddk = C; a.dv = dvm.p; a.lddv = C;
                TRY(launch_attention_bwd(e.st, a));
                TRY(launch_tome_unmerge(e.st, dkm.p, B, Nq, C, tr, order, dstl, inv, dqkv.p + C, W3));
                TRY(launch_tome_unmerge(e.st, dvm.p, B, Nq, C, tr, order, dstl, inv, dqkv.p + 2 * C, W3));
            }
            e.free(dkm); e.free(dvm);
        } else if (tr > 0) {
            // ToMe (nonfree/tome_unet.py:138-182): keys / values are merged to N - r rows before the attention.  The matching
            // is re-derived (deterministic), the attention adjoint runs against the merged rows, and the merge's adjoint
            // spreads d K_merged / d V_merged back over the original tokens; indices carry no gradient.
            const int nout = Nq - tr, half = Nq / 2, ldkm = (nout + 31) / 32 * 32, ldvt = (nout + 7) / 8 * 8;
            Tn km, vm, vtm, tws, idx, dkm, dvm, ktm;
            TRY(e.alloc(km, B, nout, 1, C));
            TRY(e.alloc(vm, B, nout, 1, C));
            TRY(e.alloc(vtm, B, C, 1, ldvt));
            TRY(e.alloc_raw(tws, tome_workspace_bytes(B, Nq, C)));
            TRY(e.alloc_raw(idx, (size_t)3 * B * half * 4 + 768));
            TRY(e.alloc(ktm, B, C, 1, ldkm));
            TRY(e.alloc(dkm, B, nout, 1, C));
            TRY(e.alloc(dvm, B, nout, 1, C));
            if (!e.dry()) {
                int* order = (int*)idx.p;
                int* dstl = order + (((size_t)B * half + 63) & ~(size_t)63);
                int* inv = dstl + (((size_t)B * half + 63) & ~(size_t)63);
                TomeParams tp;
                tp.k = qkv.p + C; tp.ldk = W3; tp.v = qkv.p + 2 * C; tp.ldv = W3; tp.B = B; tp.N = Nq; tp.C = C; tp.r = tr;
                tp.k_out = km.p; tp.vt_out = vtm.p; tp.ldvt = ldvt; tp.ws = tws.p; tp.ws_bytes = tws.bytes;
                tp.vrows_out = vm.p; tp.order_out = order; tp.dstlist_out = dstl;
                TRY(launch_tome_merge(e.st, tp));
                a.k = km.p; a.ldk = C; a.v = vm.p; a.ldv = C; a.Nk = nout; a.kt = ktm.p; a.ldkt = ldkm;
                a.dk = dkm.p; a.lddk = C; a.dv = dvm.p; a.lddv = C;
                if (need_t) {
                    TRY(launch_transpose(e.st, km.p, C, nout, C, ktm.p, ldkm, B, (size_t)nout * C, (size_t)C * ldkm));
                    TRY(launch_transpose(e.st, qkv.p, W3, Nq, C, qt.p, ldqt, B, (size_t)Nq * W3, (size_t)C * ldqt));
                    TRY(launch_transpose(e.st, d_ao.p, C, Nq, C, dot.p, ldqt, B, (size_t)Nq * C, (size_t)C * ldqt));
                }
                TRY(launch_attention_bwd(e.st, a));
                TRY(launch_tome_unmerge(e.st, dkm.p, B, Nq, C, tr, order, dstl, inv, dqkv.p + C, W3));
                TRY(launch_tome_unmerge(e.st, dvm.p, B, Nq, C, tr, order, dstl, inv, dqkv.p + 2 * C, W3));
            }
            e.free(km); e.free(vm); e.free(vtm); e.free(tws); e.free(idx); e.free(ktm); e.free(dkm); e.free(dvm);
        } else {
            a.k = e.dry() ? nullptr : qkv.p + C; a.ldk = W3; a.v = vrows; a.ldv = ldv;
            a.dk = e.dry() ? nullptr : dqkv.p + C; a.lddk = W3;
            a.dv = w.qkv_fused ? (e.dry() ? nullptr : dqkv.p + 2 * C) : dv.p; a.lddv = w.qkv_fused ? W3 : C;
            if (!e.dry()) {
                if (need_t) {
                    TRY(launch_transpose(e.st, qkv.p + C, W3, Nk, C, kt.p, ldkt, B, (size_t)Nk * W3, (size_t)C * ldkt));
                    TRY(launch_transpose(e.st, qkv.p, W3, Nq, C, qt.p, ldqt, B, (size_t)Nq * W3, (size_t)C * ldqt));
                    TRY(launch_transpose(e.st, d_ao.p, C, Nq, C, dot.p, ldqt, B, (size_t)Nq * C, (size_t)C * ldqt));
                }
                TRY(launch_attention_bwd(e.st, a));
            }
        }
        if (w.qkv_fused) {
            TRY(linear_bwd(e, dqkv.p, W3, M, W3, w.wqk, C, nullptr, 0, d_xq.p, C));
        } else {
            Tn tmp;
            TRY(e.alloc(tmp, B, xq.H, xq.W, C));
            TRY(linear_bwd(e, dqkv.p, W3, M, W3, w.wqk, C, nullptr, 0, tmp.p, C));
            TRY(linear_bwd(e, dv.p, C, M, C, w.wv, C, tmp.p, C, d_xq.p, C));
            e.free(tmp);
        }
    }
    e.free(d_ao); e.free(qkv); e.free(kc); e.free(vc); e.free(kt); e.free(qt); e.free(dot); e.free(dqkv); e.free(dv); e.free(stats);
    return 0;
}

// adjoint of Exec::resnet: gradients of x and (when present) of the concatenated skip source
int resnet_bwd(Exec& e, const Tn& x, const Tn* skip, const ResW& w, const float* tproj, int ld_tproj, float eps,
               const ResSave& sv, const Tn& d_out, Tn& dx, Tn* dskip) {
    (void)tproj; (void)ld_tproj;
    Tn d_b, d_h1, d_a;
    ConvW c2{w.c2w, w.c2b, w.cout, w.cout};
    TRY(conv3_bwd(e, d_out, c2, pad8(w.cout), 1, 1, 0, x.H, x.W, nullptr, d_b));
    TRY(groupnorm_bwd(e, sv.h1, nullptr, w.n2g, w.n2b, eps, 1, d_b, nullptr, d_h1, nullptr));
    e.free(d_b);
    const int cin_tot = x.C + (skip ? skip->C : 0);
    ConvW c1{w.c1w, w.c1b, w.cin, w.cout};
    TRY(conv3_bwd(e, d_h1, c1, cin_tot, 1, 1, 0, x.H, x.W, nullptr, d_a));
    e.free(d_h1);
    if (w.scw) {
        // shortcut 1x1 conv over the concatenation: its gradient is split by rows of W^T and added to the GroupNorm branch
        Tn g1, g2, wt;
        TRY(groupnorm_bwd(e, x, skip, w.n1g, w.n1b, eps, 1, d_a, nullptr, g1, skip ? &g2 : nullptr));
        e.free(d_a);
        const int N = d_out.C, M = x.rows();
        const bf16_t* wtp;
        TRY(weight_t_linear(e, w.scw, N, cin_tot, wt, &wtp));
        TRY(e.alloc(dx, x.B, x.H, x.W, x.C));
        TRY(e.linear(d_out.p, N, nullptr, 0, 0, M, N, wtp, x.C, nullptr, g1.p, x.C, 0, dx.p, x.C));
        e.free(g1);
        if (skip) {
            TRY(e.alloc(*dskip, x.B, x.H, x.W, skip->C));
            TRY(e.linear(d_out.p, N, nullptr, 0, 0, M, N, e.dry() ? nullptr : wtp + (size_t)x.C * N, skip->C, nullptr, g2.p, skip->C, 0,
                         dskip->p, skip->C));
            e.free(g2);
        }
        e.free(wt);
    } else {
        TRY(groupnorm_bwd(e, x, nullptr, w.n1g, w.n1b, eps, 1, d_a, &d_out, dx, nullptr));
        e.free(d_a);
    }
    return 0;
}

// adjoint of Exec::transformer
int transformer_bwd(Exec& e, const Tn& x, const Tn& ctx, int S, int ctx_dim, const TransW& w, const TransSave& sv,
                    const Tn& d_out, Tn& dx) {
    const int B = x.B, M = x.rows(), C = w.c;
    Tn dh;
    TRY(e.alloc(dh, B, x.H, x.W, C));
    TRY(linear_bwd(e, d_out.p, C, M, C, w.pout, C, nullptr, 0, dh.p, C));
    for (int bi = (int)w.blocks.size() - 1; bi >= 0; --bi) {
        const TBlockW& bw = w.blocks[bi];
        const TBlockSave& bs = sv.blocks[bi];
        // ---- feed-forward: h3 = h2 + ff2(geglu(ff1(n3)))
        Tn pre, dff, dpre, dn, dh2;
        TRY(e.alloc(dff, B, x.H, x.W, 4 * C));
        TRY(linear_bwd(e, dh.p, C, M, C, bw.ff2, 4 * C, nullptr, 0, dff.p, 4 * C));
        TRY(e.alloc(pre, B, x.H, x.W, 8 * C));
        TRY(e.linear(bs.n3.p, C, nullptr, 0, 0, M, C, bw.ff1, 8 * C, bw.ff1b, nullptr, 0, 0, pre.p, 8 * C));
        TRY(e.alloc(dpre, B, x.H, x.W, 8 * C));
        if (!e.dry()) TRY(launch_geglu_bwd(e.st, pre.p, dff.p, (size_t)M, 4 * C, dpre.p));
        e.free(pre); e.free(dff);
        TRY(e.alloc(dn, B, x.H, x.W, C));
        TRY(linear_bwd(e, dpre.p, 8 * C, M, 8 * C, bw.ff1, C, nullptr, 0, dn.p, C));
        e.free(dpre);
        TRY(layernorm_bwd(e, bs.h2, bw.ln3g, dn, &dh, dh2));
        e.free(dn); e.free(dh); dh = dh2;
        // ---- cross-attention: h2 = h1 + attn2(n2)
        Tn dn2, dh1;
        TRY(mha_bwd(e, bs.n2, true, ctx.p, S, ctx_dim, bw.a2, bs.a2, dh, dn2));
        TRY(layernorm_bwd(e, bs.h1, bw.ln2g, dn2, &dh, dh1));
        e.free(dn2); e.free(dh); dh = dh1;
        // ---- self-attention: h1 = h0 + attn1(n1)
        Tn dn1, dh0;
        TRY(mha_bwd(e, bs.n1, false, nullptr, 0, 0, bw.a1, bs.a1, dh, dn1));
        TRY(layernorm_bwd(e, bs.h0, bw.ln1g, dn1, &dh, dh0));
        e.free(dn1); e.free(dh); dh = dh0;
    }
    Tn da;
    TRY(e.alloc(da, B, x.H, x.W, C));
    TRY(linear_bwd(e, dh.p, C, M, C, w.pin, C, nullptr, 0, da.p, C));
    e.free(dh);
    TRY(groupnorm_bwd(e, x, nullptr, w.ng, w.nb, 1e-6f, 0, da, &d_out, dx, nullptr));
    e.free(da);
    return 0;
}

void free_trans_save(Exec& e, TransSave& s) {
    for (auto& b : s.blocks) { e.free(b.h0); e.free(b.n1); e.free(b.h1); e.free(b.n2); e.free(b.h2); e.free(b.n3); e.free(b.a1.ao); e.free(b.a2.ao);
                                 e.free(b.a1.qk); e.free(b.a1.km); e.free(b.a1.vm); e.free(b.a1.idx); e.free(b.a2.qk); }
    e.free(s.hlast);
    s.blocks.clear();
}

// final conv of both models writes NCHW straight to the caller; its adjoint starts from the caller's NCHW gradient
int conv_out_bwd(Exec& e, const void* d_out, int ddt, int B, int H, int W, int cout, const ConvW& w, int cin, Tn& d_a) {
    Tn dy;
    TRY(e.alloc(dy, B, H, W, pad8(cout)));
    if (!e.dry()) TRY(launch_nchw_to_nhwc(e.st, d_out, ddt, B, cout, H * W, dy.C, dy.p));
    TRY(conv3_bwd(e, dy, w, cin, 1, 1, 0, H, W, nullptr, d_a));
    e.free(dy);
    return 0;
}
// adjoint of the first conv: gradient of the padded NHWC input, returned to the caller as NCHW
int conv_in_bwd(Exec& e, const Tn& dh, const ConvW& w, int cin, int cin_pad, void* dx_out, int xdt) {
    Tn wt; const bf16_t* wtp;
    TRY(weight_t_conv(e, w.w, dh.C, cin_pad, wt, &wtp));
    if (e.dry()) { e.free(wt); return 0; }
    ConvW t{const_cast<bf16_t*>(wtp), nullptr, dh.C, cin};
    TRY(e.conv3_nchw(dh, t, dx_out, xdt));
    e.free(wt);
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// UNet: eps = f(x), d_x = (d eps / d x)^T d_eps
// ------------------------------------------------------------------------------------------
// Forward half: runs the UNet keeping what the reverse sweep needs and leaves that bookkeeping in u.vjp (the activations live in
// the caller's workspace, which must stay untouched until gyre_unet_vjp_reverse; any other call on the handle drops the state).
int gyre_unet_vjp_forward(gyre_unet& u, bool dry, hipStream_t st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt,
                          int B, int H, int W, int S, void* ws, size_t ws_bytes, void* eps_out, int odt, const float* temb_add) {
    UNetVjpState& V = u.vjp;
    V = UNetVjpState();
    const gyre_unet_cfg& c = u.cfg;
    const int n = c.n_levels;
    if (B < 1 || H < 1 || W < 1 || S < 1) GYRE_FAIL(GYRE_ERR_INVALID, "unet vjp: empty batch / image / context");
    if (!ctx && !dry) GYRE_FAIL(GYRE_ERR_INVALID, "unet vjp: the text context must be passed (the K/V cache holds no row-major V)");
    Exec& e = u.ex;
    e.arena.reset((char*)ws, ws_bytes, dry);
    e.st = st; e.batch = B; e.ctx_cache = nullptr; e.ctx_layer = 0; e.cs_unit = 0;   // the reverse sweep recomputes its own GroupNorm statistics
    const int D = c.cross_attention_dim;
    Tn xin, cx, emb, t1, t2, tp;
    TRY(e.alloc(xin, B, H, W, pad8(c.in_channels)));
    TRY(e.alloc(cx, B, S, 1, D));
    TRY(e.alloc(emb, B, 1, 1, c.block_out_channels[0], 4));
    TRY(e.alloc(t1, B, 1, 1, u.temb_dim, 4));
    TRY(e.alloc(t2, B, 1, 1, u.temb_dim, 4));
    TRY(e.alloc(tp, B, 1, 1, u.temb_cols, 4));
    if (!dry) {
        TRY(launch_nchw_to_nhwc(st, x, xdt, B, c.in_channels, H * W, xin.C, xin.p));
        TRY(launch_ctx_to_bf16(st, ctx, cdt, (size_t)B * S * D, cx.p));
        TRY(launch_timestep_linear(st, t, B, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift, (float*)emb.p, u.te1w, u.te1b,
                                   u.temb_dim, (float*)t1.p, u.temb_dim));
        TRY(launch_rowvec_linear(st, (float*)t1.p, B, u.temb_dim, u.te2w, u.te2b, u.temb_dim, 1, (float*)t2.p, u.temb_dim));
        if (temb_add) TRY(launch_add_f32(st, (float*)t2.p, temb_add, (size_t)B * u.temb_dim));
        TRY(launch_rowvec_linear(st, (float*)t2.p, B, u.temb_dim, u.tproj_w, u.tproj_b, u.temb_cols, 1, (float*)tp.p, u.temb_cols));
    }
    e.free(emb); e.free(t1); e.free(t2);
    const float* tproj = (const float*)tp.p;

    // ---- forward, keeping what the reverse sweep needs (forward activations are not returned to the arena) ----
    typedef UNetVjpNode Node;
    std::vector<Node>&downs = V.downs, &ups = V.ups;
    Node &midn = V.midn, &mid1n = V.mid1n;
    std::vector<Tn>& skips = V.skips;
    Tn h;
    TRY(e.conv3(xin, u.conv_in, 1, 1, 0, nullptr, 0, nullptr, h));
    skips.push_back(h);
    auto node_fwd = [&](Node& nd) -> int {
        TRY(e.resnet(nd.x, nd.has_skip ? &nd.skip : nullptr, *nd.rw, tproj, u.temb_cols, 1e-5f, nd.r, &nd.rs));
        nd.out = nd.r;
        if (nd.tw) TRY(e.transformer(nd.r, cx, S, D, *nd.tw, nd.out, &nd.ts));
        return 0;
    };
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < c.layers_per_block; ++j) {
            downs.emplace_back();
            Node& nd = downs.back();
            nd.rw = &u.down[i].res[j]; nd.x = skips.back();
            if (c.attn_levels[i]) nd.tw = &u.down[i].attn[j];
            TRY(node_fwd(nd));
            skips.push_back(nd.out);
        }
        if (u.down[i].has_resample) {
            downs.emplace_back();
            Node& nd = downs.back();
            nd.cw = &u.down[i].resample; nd.x = skips.back();
            TRY(e.conv3(nd.x, *nd.cw, 2, 1, 0, nullptr, 0, nullptr, nd.out));
            skips.push_back(nd.out);
        }
    }
    midn.rw = &u.mid0; midn.tw = &u.mid_attn; midn.x = skips.back();
    TRY(node_fwd(midn));
    mid1n.rw = &u.mid1; mid1n.x = midn.out;
    TRY(node_fwd(mid1n));
    h = mid1n.out;
    size_t sp = skips.size();
    for (int i = 0; i < n; ++i) {
        const int lvl = n - 1 - i;
        for (int j = 0; j < c.layers_per_block + 1; ++j) {
            ups.emplace_back();
            Node& nd = ups.back();
            nd.rw = &u.up[i].res[j]; nd.x = h; nd.skip = skips[--sp]; nd.has_skip = true;
            if (c.attn_levels[lvl]) nd.tw = &u.up[i].attn[j];
            TRY(node_fwd(nd));
            h = nd.out;
        }
        if (u.up[i].has_resample) {
            ups.emplace_back();
            Node& nd = ups.back();
            nd.cw = &u.up[i].resample; nd.x = h;
            TRY(e.conv3(nd.x, *nd.cw, 1, 1, 1, nullptr, 0, nullptr, nd.out, skips[sp - 1].H, skips[sp - 1].W));
            h = nd.out;
        }
    }
    if (sp != 0) GYRE_FAIL(GYRE_ERR_INVALID, "internal: skip bookkeeping");
    Tn a;
    TRY(e.groupnorm(h, nullptr, u.ong, u.onb, 1e-5f, 1, a));
    TRY(e.conv3_nchw(a, u.conv_out, eps_out, odt));
    e.free(a);
    V.h = h; V.cx = cx; V.tp = tp; V.xin_c = xin.C;
    V.B = B; V.H = H; V.W = W; V.S = S; V.ws = ws; V.ws_bytes = ws_bytes; V.dry = dry; V.valid = true;
    return 0;
}

// Reverse half: d_x = (d eps / d x)^T d_eps from the state gyre_unet_vjp_forward left behind
// Samples [b0, b0 + nb) of every kept activation (batch-major NHWC: a pointer offset; offsets / sizes of the arena bookkeeping stay)
static void narrow_tn(Tn& t, int fullB, int b0, int nb) {
    if (!t.valid() || t.B != fullB) return;
    if (t.p) t.p += (size_t)b0 * t.H * t.W * t.C;
    t.B = nb;
    t.cs = nullptr; t.cs_chunks = 0;                   // (producer statistics are not used by the sweep)
}
static void narrow_state(UNetVjpState& V, int b0, int nb) {
    const int fb = V.B;
    auto node = [&](UNetVjpNode& nd) {
        for (Tn* t : {&nd.x, &nd.skip, &nd.r, &nd.out, &nd.rs.h1, &nd.ts.hlast}) narrow_tn(*t, fb, b0, nb);
        for (auto& b : nd.ts.blocks) {
            for (Tn* t : {&b.h0, &b.n1, &b.h1, &b.n2, &b.h2, &b.n3, &b.a1.ao, &b.a2.ao, &b.a1.qk, &b.a1.km, &b.a1.vm, &b.a2.qk}) narrow_tn(*t, fb, b0, nb);
            b.a1.idx_b0 = b0;
        }
    };
    for (auto& nd : V.downs) node(nd);
    for (auto& nd : V.ups) node(nd);
    node(V.midn); node(V.mid1n);
    for (auto& t : V.skips) narrow_tn(t, fb, b0, nb);
    narrow_tn(V.h, fb, b0, nb);
    narrow_tn(V.cx, fb, b0, nb);
    V.B = nb;
}
int gyre_unet_vjp_reverse(gyre_unet& u, hipStream_t st, const void* d_eps, int ddt, void* dx_out, int dxdt, int b0, int nb) {
    UNetVjpState& V = u.vjp;
    if (!V.valid) GYRE_FAIL(GYRE_ERR_INVALID, "unet vjp: no forward state pending (another call on this handle ran in between)");
    V.valid = false;                                   // one reverse sweep per forward: gradients are freed as they are consumed
    const gyre_unet_cfg& c = u.cfg;
    Exec& e = u.ex;
    const bool dry = V.dry;
    if (nb != V.B) {
        if (V.B < 2 || b0 < 0 || nb < 1 || b0 + nb > V.B) GYRE_FAIL(GYRE_ERR_INVALID, "unet vjp: sample range outside the pending batch");
        narrow_state(V, b0, nb);
        e.batch = nb;
    }
    const int B = V.B, H = V.H, W = V.W, S = V.S, D = c.cross_attention_dim;
    e.st = st;
    std::vector<UNetVjpNode>&downs = V.downs, &ups = V.ups;
    UNetVjpNode &midn = V.midn, &mid1n = V.mid1n;
    std::vector<Tn>& skips = V.skips;
    Tn h = V.h, cx = V.cx, tp = V.tp;
    const float* tproj = (const float*)tp.p;
    typedef UNetVjpNode Node;
    // ---- reverse sweep ----
    Tn d_a, dh;
    TRY(conv_out_bwd(e, d_eps, ddt, B, H, W, c.out_channels, u.conv_out, h.C, d_a));
    TRY(groupnorm_bwd(e, h, nullptr, u.ong, u.onb, 1e-5f, 1, d_a, nullptr, dh, nullptr));
    e.free(d_a);
    auto node_bwd = [&](Node& nd, Tn& g, Tn* dskip) -> int {     // in: gradient of nd.out; out: gradient of nd.x (and of nd.skip)
        if (nd.tw) {
            Tn gr;
            TRY(transformer_bwd(e, nd.r, cx, S, D, *nd.tw, nd.ts, g, gr));
            e.free(g); g = gr;
        }
        Tn gin;
        TRY(resnet_bwd(e, nd.x, nd.has_skip ? &nd.skip : nullptr, *nd.rw, tproj, u.temb_cols, 1e-5f, nd.rs, g, gin, dskip));
        e.free(g); g = gin;
        return 0;
    };
    // the up path consumed skips[K-1] ... skips[0]; walking it backwards yields their gradients in the order 0 ... K-1
    std::vector<Tn> dskips;
    for (int k = (int)ups.size() - 1; k >= 0; --k) {
        Node& nd = ups[k];
        if (nd.cw) {
            Tn g;
            TRY(conv3_bwd(e, dh, *nd.cw, nd.x.C, 1, 1, 1, nd.x.H, nd.x.W, nullptr, g));
            e.free(dh); dh = g;
        } else {
            Tn ds;
            TRY(node_bwd(nd, dh, &ds));
            dskips.push_back(ds);
        }
    }
    if (dskips.size() != skips.size() || downs.size() + 1 != skips.size()) GYRE_FAIL(GYRE_ERR_INVALID, "internal: skip bookkeeping");
    TRY(node_bwd(mid1n, dh, nullptr));
    TRY(node_bwd(midn, dh, nullptr));
    const int K = (int)skips.size();
    if (!dry) TRY(launch_add_bf16(st, dh.p, dskips[K - 1].p, (size_t)dh.rows() * dh.C));
    e.free(dskips[K - 1]);
    // downs[k] maps skips[k] -> skips[k + 1]; dh = complete gradient of skips[k + 1]
    for (int k = K - 2; k >= 0; --k) {
        Node& nd = downs[k];
        if (nd.cw) {
            Tn g;
            TRY(conv3_bwd(e, dh, *nd.cw, nd.x.C, 2, 1, 0, nd.x.H, nd.x.W, &dskips[k], g));
            e.free(dh); dh = g;
        } else {
            TRY(node_bwd(nd, dh, nullptr));
            if (!dry) TRY(launch_add_bf16(st, dh.p, dskips[k].p, (size_t)dh.rows() * dh.C));
        }
        e.free(dskips[k]);
    }
    TRY(conv_in_bwd(e, dh, u.conv_in, c.in_channels, V.xin_c, dx_out, dxdt));
    e.free(dh);
    return 0;
}

// one-shot form: forward + reverse in one call (also the dry run that sizes the workspace)
int gyre_unet_run_vjp(gyre_unet& u, bool dry, hipStream_t st, const void* x, int xdt, const int64_t* t, const void* ctx, int cdt,
                      int B, int H, int W, int S, const void* d_eps, int ddt, void* ws, size_t ws_bytes, void* eps_out, int odt,
                      void* dx_out, int dxdt, const float* temb_add) {
    TRY(gyre_unet_vjp_forward(u, dry, st, x, xdt, t, ctx, cdt, B, H, W, S, ws, ws_bytes, eps_out, odt, temb_add));
    return gyre_unet_vjp_reverse(u, st, d_eps, ddt, dx_out, dxdt, 0, B);
}

// ------------------------------------------------------------------------------------------
// VAE decoder: image = g(z), d_z = (d image / d z)^T d_image
// ------------------------------------------------------------------------------------------
int gyre_vae_run_decode_vjp(gyre_vae& v, bool dry, hipStream_t st, const void* z, int zdt, int B, int h_, int w_, const void* d_img,
                            int ddt, void* ws, size_t wsb, void* img_out, int odt, void* dz_out, int dzdt) {
    const gyre_vae_cfg& c = v.cfg;
    const int n = c.n_levels;
    if (B < 1 || h_ < 1 || w_ < 1) GYRE_FAIL(GYRE_ERR_INVALID, "vae decode vjp: empty input");
    Exec& e = v.ex;
    e.arena.reset((char*)ws, wsb, dry); e.st = st; e.batch = B; e.cs_unit = 0;
    const int zc = pad8(c.latent_channels);
    struct RNode { const ResW* rw; Tn x, out; ResSave rs; };
    struct UNode { const ConvW* cw; Tn x, out; };
    Tn x, q, h0;
    TRY(e.alloc(x, B, h_, w_, zc));
    if (!dry) TRY(launch_nchw_to_nhwc(st, z, zdt, B, c.latent_channels, h_ * w_, zc, x.p));
    TRY(e.alloc(q, B, h_, w_, zc));
    TRY(e.linear(x.p, zc, nullptr, 0, 0, x.rows(), zc, v.pq_w, zc, v.pq_b, nullptr, 0, 0, q.p, zc));
    TRY(e.conv3(q, v.d_in, 1, 1, 0, nullptr, 0, nullptr, h0));
    RNode m0{&v.d_mid0}, m1{&v.d_mid1};
    Tn an, ao_out; MhaSave asv;
    m0.x = h0;
    TRY(e.resnet(m0.x, nullptr, *m0.rw, nullptr, 0, 1e-6f, m0.out, &m0.rs));
    TRY(e.groupnorm(m0.out, nullptr, v.d_ag, v.d_ab, 1e-6f, 0, an));
    TRY(e.mha(an, false, nullptr, 0, 0, v.d_attn, m0.out, ao_out, &asv));
    m1.x = ao_out;
    TRY(e.resnet(m1.x, nullptr, *m1.rw, nullptr, 0, 1e-6f, m1.out, &m1.rs));
    Tn h = m1.out;
    std::vector<std::pair<int, size_t>> order;   // (0 = resnet, 1 = upsample), index
    std::vector<RNode> rn; std::vector<UNode> un;
    for (int i = 0; i < n; ++i) {
        for (auto& rw : v.d_up[i].res) {
            rn.push_back(RNode{&rw});
            RNode& nd = rn.back();
            nd.x = h;
            TRY(e.resnet(nd.x, nullptr, rw, nullptr, 0, 1e-6f, nd.out, &nd.rs));
            h = nd.out;
            order.emplace_back(0, rn.size() - 1);
        }
        if (v.d_up[i].has_resample) {
            un.push_back(UNode{&v.d_up[i].resample});
            UNode& nd = un.back();
            nd.x = h;
            TRY(e.conv3(nd.x, *nd.cw, 1, 1, 1, nullptr, 0, nullptr, nd.out));
            h = nd.out;
            order.emplace_back(1, un.size() - 1);
        }
    }
    Tn a;
    TRY(e.groupnorm(h, nullptr, v.d_ng, v.d_nb, 1e-6f, 1, a));
    TRY(e.conv3_nchw(a, v.d_out, img_out, odt));
    e.free(a);

    // ---- reverse sweep ----
    Tn d_a, dh;
    TRY(conv_out_bwd(e, d_img, ddt, B, h.H, h.W, c.out_channels, v.d_out, h.C, d_a));
    TRY(groupnorm_bwd(e, h, nullptr, v.d_ng, v.d_nb, 1e-6f, 1, d_a, nullptr, dh, nullptr));
    e.free(d_a);
    auto res_bwd = [&](RNode& nd, Tn& g) -> int {
        Tn gin;
        TRY(resnet_bwd(e, nd.x, nullptr, *nd.rw, nullptr, 0, 1e-6f, nd.rs, g, gin, nullptr));
        e.free(g); g = gin;
        return 0;
    };
    for (int k = (int)order.size() - 1; k >= 0; --k) {
        if (order[k].first == 1) {
            UNode& nd = un[order[k].second];
            Tn g;
            TRY(conv3_bwd(e, dh, *nd.cw, nd.x.C, 1, 1, 1, nd.x.H, nd.x.W, nullptr, g));
            e.free(dh); dh = g;
        } else {
            TRY(res_bwd(rn[order[k].second], dh));
        }
    }
    TRY(res_bwd(m1, dh));
    {   // attention block: out = x + attn(GN(x))
        Tn dn, g;
        TRY(mha_bwd(e, an, false, nullptr, 0, 0, v.d_attn, asv, dh, dn));
        TRY(groupnorm_bwd(e, m0.out, nullptr, v.d_ag, v.d_ab, 1e-6f, 0, dn, &dh, g, nullptr));
        e.free(dn); e.free(dh); dh = g;
    }
    TRY(res_bwd(m0, dh));
    Tn dq;
    TRY(conv3_bwd(e, dh, v.d_in, zc, 1, 1, 0, h_, w_, nullptr, dq));
    e.free(dh);
    // post_quant_conv (1x1) adjoint straight to the caller's NCHW buffer
    Tn wt; const bf16_t* wtp;
    TRY(weight_t_linear(e, v.pq_w, zc, zc, wt, &wtp));
    if (!dry) {
        GemmParams p;
        p.A = dq.p; p.lda = zc; p.mode = GEMM_LINEAR; p.W = wtp; p.K = zc; p.N = c.latent_channels; p.M = dq.rows();
        p.rows_per_sample = h_ * w_; p.out = dz_out; p.out_mode = OUT_NCHW; p.out_dtype = dzdt;
        TRY(launch_gemm(st, p));
    }
    e.free(wt); e.free(dq);
    return 0;
}
