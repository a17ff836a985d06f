/*
 * gyre_hip.h - C ABI of libgyre_hip.so, the MI355X (gfx950) native UNet / VAE
 * for Gyre's diffusion generation path.
 *
 * Nothing like this exists in the reference (it is pure Python and crosses into
 * third-party diffusers modules); each entry point below replaces one of the
 * reference's Python call sites across that boundary:
 *
 *   gyre_unet_forward     <- gyre/pipeline/unet/core.py:262-274
 *                            `self.unet(latents, t, encoder_hidden_states=...).sample`
 *                            (Protocol gyre/pipeline/unet/types.py:26-39)
 *   gyre_vae_encode       <- gyre/pipeline/unified_pipeline.py:309
 *                            `self.pipeline.vae.encode(image).latent_dist`  (returns the moments;
 *                            the per-generator posterior sample :311-313 stays host PyTorch)
 *   gyre_vae_decode       <- gyre/pipeline/unified_pipeline.py:1531-1533  `self.vae.decode(latents).sample`
 *   gyre_*_create         <- gyre/manager.py:1068-1112  `Class(**config)`
 *   gyre_*_set_weight     <- gyre/manager.py:1068-1112  `load_state_dict` (keys = diffusers names,
 *                            the key space gyre/ckpt_utils.py:259-285 produces)
 *   gyre_*_destroy        <- gyre/pipeline/pipeline_wrapper.py:144-158 deactivate()
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer marked "dev" is a device pointer
 *     valid on the handle's device.  No torch / C++ types cross this boundary.
 *   - every function returns 0 on success or a negative gyre_status; on failure
 *     gyre_last_error() (thread-local) holds a message.  Nothing throws.
 *   - all work is enqueued on the hipStream_t passed as `stream` (void* here so the
 *     header needs no HIP include).  No function calls hipDeviceSynchronize; the
 *     caller owns synchronisation.  Handles hold no global mutable state: different
 *     handles may be used concurrently from different threads (one in-flight call
 *     per handle, the reference's one-thread-per-device-slot rule,
 *     gyre/manager.py:2107-2139).
 *   - activations cross the boundary in the reference's own layout: NCHW for
 *     latents/images, [B,S,D] for text embeddings; dtype per call (gyre_dtype).
 *     Internally everything is NHWC bf16 (libgyre_hip.so) or NHWC fp16 (libgyre_hip_f16.so, gyre_storage_dtype) with fp32
 *     accumulation.
 *   - `workspace` is caller-allocated device scratch (e.g. from the torch caching
 *     allocator) of at least gyre_*_workspace_bytes(...) bytes, 256-byte aligned.
 */
#ifndef GYRE_HIP_H
#define GYRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GYRE_ABI_VERSION 1

typedef enum { GYRE_F32 = 0, GYRE_BF16 = 1, GYRE_F16 = 2 } gyre_dtype;

typedef enum {
    GYRE_OK = 0,
    GYRE_ERR_INVALID = -1,     /* bad argument / shape / dtype            -> Python ValueError */
    GYRE_ERR_KEY = -2,         /* unknown or shape-mismatched weight key -> Python KeyError   */
    GYRE_ERR_INCOMPLETE = -3,  /* forward before every weight was set    -> RuntimeError      */
    GYRE_ERR_WORKSPACE = -4,   /* workspace too small                    -> RuntimeError      */
    GYRE_ERR_HIP = -5,         /* a HIP runtime call failed              -> RuntimeError      */
    GYRE_ERR_UNSUPPORTED = -6  /* configuration not implemented          -> NotImplementedError */
} gyre_status;

#define GYRE_MAX_LEVELS 8

/* Mirrors the diffusers UNet2DConditionModel config.json fields the reference
 * relies on (SD1.x values from gyre/ldm_config/v1-inference.yaml:29-47). */
typedef struct {
    int32_t in_channels;             /* 4, or 9 for the runway inpaint UNet (unified_pipeline.py:668-690) */
    int32_t out_channels;            /* 4 */
    int32_t n_levels;                /* 4 */
    int32_t block_out_channels[GYRE_MAX_LEVELS]; /* 320,640,1280,1280 */
    int32_t layers_per_block;        /* 2 */
    int32_t attn_levels[GYRE_MAX_LEVELS];        /* 1,1,1,0 */
    int32_t num_heads[GYRE_MAX_LEVELS];          /* 8,8,8,8 */
    int32_t transformer_depth[GYRE_MAX_LEVELS];  /* 1,1,1,1 */
    int32_t cross_attention_dim;     /* 768 */
    int32_t norm_num_groups;         /* 32 */
    int32_t use_linear_projection;   /* 0 for SD1.x */
    int32_t flip_sin_to_cos;         /* 1 */
    float   freq_shift;              /* 0 */
} gyre_unet_cfg;

typedef struct {
    int32_t in_channels;             /* 3 */
    int32_t out_channels;            /* 3 */
    int32_t latent_channels;         /* 4 */
    int32_t n_levels;                /* 4 */
    int32_t block_out_channels[GYRE_MAX_LEVELS]; /* 128,256,512,512 */
    int32_t layers_per_block;        /* 2 */
    int32_t norm_num_groups;         /* 32 */
} gyre_vae_cfg;

typedef struct gyre_unet gyre_unet;
typedef struct gyre_vae gyre_vae;

/* ---- library ---------------------------------------------------------- */
int gyre_abi_version(void);
/* The 16-bit type this build stores activations and weights in (accumulation and all epilogues are fp32 in both): GYRE_BF16 for
 * libgyre_hip.so, GYRE_F16 for libgyre_hip_f16.so - the same sources and the same ABI built with -DGYRE_STORE_F16, the reference's own
 * GPU arithmetic (gyre/manager.py:146-151,1199-1200 loads its pipelines in fp16).  A caller picks the library by the torch_dtype
 * it loads the model in (gyre_amd/_lib.py lib(storage)); boundary tensors keep their per-call gyre_dtype in both. */
int gyre_storage_dtype(void);
const char* gyre_last_error(void);
/* number of distinct kernel launches issued by this thread's last forward/encode/decode */
int64_t gyre_last_launch_count(void);

/* ---- UNet --------------------------------------------------------------- */
int gyre_unet_create(const gyre_unet_cfg* cfg, int device, gyre_unet** out);
void gyre_unet_destroy(gyre_unet* h);
/* Number of parameter tensors the handle expects / key of the i-th one (diffusers name). */
int gyre_unet_num_params(const gyre_unet* h);
const char* gyre_unet_param_key(const gyre_unet* h, int i);
/* Copy+repack one tensor (PyTorch layout: conv OIHW, linear [out,in], vectors [n]) from
 * `dev_ptr` (contiguous, `dtype`) into library-owned bf16/f32 buffers.  Asynchronous on `stream`;
 * the source may be released once the stream has passed this point. */
int gyre_unet_set_weight(gyre_unet* h, const char* diffusers_key, const void* dev_ptr, int dtype,
                         const int64_t* shape, int ndim, void* stream);
/* 0 when every expected key has been set; otherwise GYRE_ERR_INCOMPLETE (message lists a missing key). */
int gyre_unet_finalize(gyre_unet* h, void* stream);
size_t gyre_unet_workspace_bytes(gyre_unet* h, int B, int H, int W, int S);
/* eps[B,out_ch,H,W] = unet(x[B,in_ch,H,W], t[B] (int64, dev), ctx[B,S,cross_dim]).
 * H, W are latent sizes and must be multiples of 2^(n_levels-1). */
int gyre_unet_forward(gyre_unet* h, void* stream,
                      const void* x_nchw, int x_dtype,
                      const int64_t* t_dev,
                      const void* ctx, int ctx_dtype,
                      int B, int H, int W, int S,
                      void* workspace, size_t workspace_bytes,
                      void* eps_out_nchw, int out_dtype);

/* Hint for the NEXT gyre_unet_forward* call on this handle only: all B entries of t_dev hold the same value (the samplers
 * of the reference pass ONE timestep per call: common_scheduler.py:344 / k-diffusion's sigma_to_t).  The time-embedding MLP and
 * the batched time_emb_proj then run for one row that every resnet reads - 1/B of that work, bit-identical results.  Ignored
 * when per-sample added conditioning (temb_add) is given.  A wrong hint gives every sample the first sample's timestep. */
int gyre_unet_hint_uniform_timestep(gyre_unet* h, int on);
/* (Both hints are the caller's word.  With GYRE_VERIFY_HINTS=1 in the environment every hinted call first checks the hint on the
 * device - one compare kernel over t / over the two halves of x, one 4-byte read-back, i.e. a stream synchronisation - and returns
 * GYRE_ERR_INVALID with a message instead of computing on a wrong assumption: the switch for bringing up a foreign caller.) */
/* Hint for the NEXT gyre_unet_forward* call on this handle only: the batch is a CFG-parallel pair batch as the reference builds
 * it (unet/cfg.py:49-57: latents cat[x, x], timesteps cat[t, t], contexts cat[uncond, cond]) - sample b and sample b + B/2 have
 * IDENTICAL latents and timestep and differ in their text context only.  Everything in front of the first cross-attention
 * (conv_in, the first resnet, the first transformer's GroupNorm / proj_in / self-attention: ~4 % of the FLOPs of an SD1.x call
 * at 64x64) is then evaluated once per pair and written twice: the two halves of a pair are IDENTICAL in that prefix, and equal
 * to the unshared call up to bf16 summation order (the prefix is planned for B/2 samples, so its tile / split-K choice - and with
 * it the order of the GroupNorm sums - may differ from the full batch's) - bit for bit under gyre_set_batch_invariant.  Ignored for odd B,
 * with ControlNet / T2I inputs, per-sample added conditioning or pending debug taps.  A wrong hint gives the second half of the
 * batch the first half's activations in that prefix. */
int gyre_unet_hint_cfg_pairs(gyre_unet* h, int on);

/* Text-context cache.  The context is constant over the 50+ UNet evaluations of a request, so its cross-attention
 * K / V projections (32 small GEMMs per call for SD1.x) can be done once: set_context projects ctx[B,S,cross_dim]
 * through every attn2.to_k / to_v into handle-owned buffers (may (re)allocate: not for the per-step path); afterwards
 * gyre_unet_forward / _ex accept ctx == NULL with the same B and S and reuse them.  Any set_weight invalidates it.
 * In the reference the same tensor is re-projected on every call (unet/core.py:253-259 binds it per wrapper). */
int gyre_unet_set_context(gyre_unet* h, void* stream, const void* ctx, int ctx_dtype, int B, int S);
/* The cache holds GYRE_CTX_SLOTS entries: the leaves of a hires-fix / graft tree and CFGUNet_Sequential
 * (unet/cfg.py:27-38, unet/hires_fix.py:123-235) alternate between two to four contexts on EVERY step.  _slot projects into
 * the given entry and makes it current (gyre_unet_set_context = slot 0); gyre_unet_select_context makes an already projected
 * entry current without any device work (GYRE_ERR_INVALID when that entry is empty or was invalidated by a set_weight). */
#define GYRE_CTX_SLOTS 4
int gyre_unet_set_context_slot(gyre_unet* h, void* stream, const void* ctx, int ctx_dtype, int B, int S, int slot);
int gyre_unet_select_context(gyre_unet* h, int slot);

/* Token merging (ToMe) for the UNet's self-attentions - the reference's pipeline option "tome: <r>"
 * (gyre/pipeline/unified_pipeline.py:1580-1588 -> nonfree/tome_patcher.py:14-52, nonfree/tome_unet.py:138-182,243):
 * before each self-attention the r most redundant keys (even token positions, by cosine similarity to their best
 * odd-position match - bipartite soft matching) are averaged into that match, values follow, and the attention runs
 * against N - r keys; r is clipped to N / 2 per layer as ToMe does.  0 switches it off (default).  The matching
 * algorithm lives in the un-vendored facebookresearch/ToMe submodule; restated from the paper, parity unpinned. */
int gyre_unet_set_tome(gyre_unet* h, int r);
/* Circular ("tiling") convolutions: the reference's request option `tiling` (True / "x" / "y" / "xy") patches every Conv2d of the
 * UNet and the VAE so that the module's own padding wraps around the image instead of reading zeros
 * (gyre/pipeline/unified_pipeline.py:1671-1712) - seamless textures.  mode: 0 off, 1 along x, 2 along y, 3 both; sticky until
 * changed.  The wrapped gather exists in the 4-wave conv kernels only, so such requests run slower; the input-gradient calls
 * (CLIP guidance) refuse it (GYRE_ERR_UNSUPPORTED). */
int gyre_unet_set_tiling(gyre_unet* h, int mode);
int gyre_vae_set_tiling(gyre_vae* h, int mode);

/* gyre_unet_forward_ex plus ControlNet-style residual injection - the optional keyword arguments of the reference's UNet call,
 * unet/core.py:40-64 (`down_block_additional_residuals`, `mid_block_additional_residual`) with the semantics of the in-tree
 * patcher controlnet/unet_patcher.py:30-95: down_res[k] (NCHW, dev, res_dtype; one per skip connection in production order,
 * conv_in's output first - 12 for SD1.x) is added to the skip tensor the up path consumes, mid_res to the mid block's output.
 * n_down_res == 0 / mid_res == NULL switch either off.
 * adapter_states (T2I adapters, unet/core.py:212-216 `adapter_states=`; semantics of t2i_adapter/unet_patcher.py:21-86): one
 * NCHW tensor (res_dtype) per down level, added in place to the level's last hidden state - before its downsampler on the
 * cross-attention levels (skip connection, downsampler and everything downstream see it), after the level otherwise. */
int gyre_unet_forward_ctrl(gyre_unet* h, void* stream, const void* x_nchw, int x_dtype, const int64_t* t_dev,
                           const void* ctx, int ctx_dtype, int B, int H, int W, int S,
                           void* workspace, size_t workspace_bytes, void* eps_out_nchw, int out_dtype, const float* temb_add,
                           const void* const* down_res, int n_down_res, int res_dtype, const void* mid_res,
                           const void* const* adapter_states, int n_adapter_states);

/* Input gradient (vector-Jacobian product) of the noise prediction: one call runs the forward pass, writes
 * eps_out_nchw like gyre_unet_forward, and writes dx_out_nchw[B, in_channels, H, W] = (d eps / d x)^T d_eps.
 * Replaces what autograd does in the reference's CLIP-guided mode, gyre/pipeline/unet/clipguided.py:301-338
 * (`latents.requires_grad_()`, `unet(latents, t)`) + :420 (`torch.autograd.grad(loss, latents)`); weights and the
 * text context receive no gradient there.  ctx must be passed (the K/V cache is not used); temb_add as in
 * gyre_unet_forward_ex (may be NULL).  With token merging enabled (gyre_unet_set_tome) the sweep re-derives the matching and
 * applies the merge's adjoint to d K / d V; the matching indices themselves carry no gradient (as under autograd). */
size_t gyre_unet_vjp_workspace_bytes(gyre_unet* h, int B, int H, int W, int S);
int gyre_unet_vjp(gyre_unet* h, void* stream, const void* x_nchw, int x_dtype, const int64_t* t_dev,
                  const void* ctx, int ctx_dtype, int B, int H, int W, int S,
                  const void* d_eps_nchw, int d_eps_dtype, void* workspace, size_t workspace_bytes,
                  void* eps_out_nchw, int out_dtype, void* dx_out_nchw, int dx_dtype, const float* temb_add);

/* The same in two calls, for an autograd node: _begin runs the forward pass (eps_out as gyre_unet_forward) and keeps the
 * activations the adjoints need in `workspace` (gyre_unet_vjp_workspace_bytes; must stay untouched), _finish runs the reverse
 * sweep for a cotangent.  At most one pending pair per handle: any other call on the handle in between drops the state and
 * _finish then returns GYRE_ERR_INVALID.  gyre_unet_vjp_pending tells the caller beforehand (1 = the state of the last _begin is
 * still there, 0 = dropped: fall back to the one-shot gyre_unet_vjp), so that a real argument error of _finish is never
 * mistaken for a dropped state. */
int gyre_unet_vjp_pending(gyre_unet* h);
int gyre_unet_vjp_begin(gyre_unet* h, void* stream, const void* x_nchw, int x_dtype, const int64_t* t_dev,
                        const void* ctx, int ctx_dtype, int B, int H, int W, int S, void* workspace, size_t workspace_bytes,
                        void* eps_out_nchw, int out_dtype, const float* temb_add);
int gyre_unet_vjp_finish(gyre_unet* h, void* stream, const void* d_eps_nchw, int d_eps_dtype, void* dx_out_nchw, int dx_dtype);
/* _finish for samples [b0, b0 + nb) of the pending forward pass only: d_eps / dx_out hold nb samples.  The reference's guided mode
 * differentiates the conditional evaluation and needs the unconditional one of the same latents right after
 * (unet/clipguided.py:218-241): one activation-keeping pass over cat[uncond, cond] (batch 2B) + the reverse sweep of the
 * conditional half replaces a batch-B keeping pass and a batch-B plain pass (SD1.5 + ToMe, B = 8: 19.8 against 12.6 + 11.7 ms). */
int gyre_unet_vjp_finish_range(gyre_unet* h, void* stream, const void* d_eps_nchw, int d_eps_dtype, void* dx_out_nchw, int dx_dtype,
                               int b0, int nb);

/* Parity tests only: the next forward copies the named intermediate activation (f32, NCHW) into out.  Names follow
 * the oracle's taps: "down<i>" (end of down level i, after its downsampler), "mid", "up<i>" (end of up level i, after
 * its upsampler).  Taps are cleared by that forward. */
int gyre_unet_debug_tap(gyre_unet* h, const char* name, float* out_nchw_f32, size_t out_bytes);

/* Same, plus an optional additive term for the time embedding: temb_add[B, 4*block_out_channels[0]] (f32, dev) is
 * added to time_embedding(t) before it feeds the ResNet blocks.  This is how SDXL's `text_time` added conditioning
 * (add_embedding MLP over pooled text + size/crop ids, a few MFLOP) enters: the MLP stays host PyTorch. */
int gyre_unet_forward_ex(gyre_unet* h, void* stream, const void* x_nchw, int x_dtype, const int64_t* t_dev,
                         const void* ctx, int ctx_dtype, int B, int H, int W, int S,
                         void* workspace, size_t workspace_bytes, void* eps_out_nchw, int out_dtype,
                         const float* temb_add_dev);

/* ---- VAE ---------------------------------------------------------------- */
int gyre_vae_create(const gyre_vae_cfg* cfg, int device, gyre_vae** out);
void gyre_vae_destroy(gyre_vae* h);
int gyre_vae_num_params(const gyre_vae* h);
const char* gyre_vae_param_key(const gyre_vae* h, int i);
int gyre_vae_set_weight(gyre_vae* h, const char* diffusers_key, const void* dev_ptr, int dtype,
                        const int64_t* shape, int ndim, void* stream);
int gyre_vae_finalize(gyre_vae* h, void* stream);
size_t gyre_vae_workspace_bytes(gyre_vae* h, int B, int H, int W, int decode);
/* moments[B,2*z,H/8,W/8] (mean | logvar, f32) = quant_conv(encoder(image[B,3,H,W] in -1..1)) */
int gyre_vae_encode(gyre_vae* h, void* stream, const void* image_nchw, int in_dtype, int B, int H, int W,
                    void* workspace, size_t workspace_bytes, void* moments_out_nchw, int out_dtype);
/* image[B,3,8h,8w] = decoder(post_quant_conv(z[B,z,h,w])) */
int gyre_vae_decode(gyre_vae* h, void* stream, const void* z_nchw, int in_dtype, int B, int h_lat, int w_lat,
                    void* workspace, size_t workspace_bytes, void* image_out_nchw, int out_dtype);

/* ---- per-launch timing with HIP events on the launch stream (bench.py roofline leg) ----
 * mask: bit k enables kernel class k (names from gyre_prof_class_name; they equal the prefix of the
 * kernel's demangled name as rocprofv3 prints it).  collect(): after the caller synchronised the
 * stream, returns per class the number of timed launches, the summed event-to-event milliseconds and
 * the summed ALGORITHMIC flops / bytes (unpadded problem sizes) and clears the records. */
int gyre_prof_set_mask(unsigned long long mask);
int gyre_prof_num_classes(void);
const char* gyre_prof_class_name(int kclass);
int gyre_prof_collect(int64_t* launches, double* ms, double* flops, double* bytes);

/* Tests / tuning only: force the GEMM tile configuration of this thread's following launches
 * (0 = automatic; 1 = 4-wave 128x128, 2 = 4-wave 256x64, 3 = 4-wave 64x64, 4 = 8-wave 256x320,
 * 5 = 8-wave 128x320, 6 = 8-wave 256x256, 7 = 8-wave 128x256 (8 = 128x160, 12 = 256x128 convolutions only, 20 - 24 pipelined forms,
 * 30 / 31 A- / W-resident, 32 small-problem kernel); bits 8-15 = split-K factor for the 8-wave
 * configs, 0/1 = none).  Returns the previous value. */
int gyre_debug_force_gemm_cfg(int cfg);
/* Tests / tuning only: split-K slab space for this thread's gyre_op_* calls (the model handles carve theirs from
 * the caller's workspace).  Without it the single operators run the best single-split configuration. */
int gyre_debug_set_splitk_workspace(void* ws_dev, size_t bytes);
/* Tests / tuning only: scratch space (N * K * 2 bytes) where this thread's gyre_op_* calls pack the weights for the A-resident
 * GEMM kernel (tile config 30: K = 320 / 640 linear problems; the model handles keep a packed copy per weight).  With it the
 * planner may choose that kernel for single operators; NULL / 0 = off. */
int gyre_debug_set_ar_workspace(void* ws_dev, size_t bytes);
/* Tests / tuning only: scratch space (N * K * 2 bytes) where this thread's gyre_op_* calls make the BLOCKED weight copy the LDS-DMA tile
 * kernels read for K > 1024 (1-KiB blocks of 8 rows x 64 k; the model handles keep one per weight).  NULL / 0 = off: row-major weights. */
int gyre_debug_set_wblk_workspace(void* ws_dev, size_t bytes);
/* Tests / tuning only: 0 automatic, 1 = register-staged attention kernel, 2 / 4 = LDS-DMA kernel with 32 / 64
 * query rows per wave; with prescaled K: 3 = folded-softmax v2 kernel, 5 = software-pipelined v3 kernel (head dims
 * 16/32/40/64); 6 = automatic without the several-query-blocks-per-workgroup form of short key sequences; 7 = automatic with the
 * pipelined kernel's per-tile overflow check in every tile (the round-3 kernel); 0 / 8 = automatic: head dims <= 40 run the optimistic
 * first pass (no per-tile check; a workgroup whose row sums leave (0, 2^60) repeats its pass with the check - since round 6 that
 * bound equals the checked pass's re-centring threshold, so the default is bit-identical to variant 7 on every input).  The round-4
 * non-reproducibility of that pass under concurrent handles was a ring-slot hazard, fixed in round 5 (kernels_attn.hip header).
 * Returns the previous value. */
int gyre_debug_force_attn_variant(int v);
/* The number of attention workgroups on the CURRENT device that have repeated their pass with the per-tile overflow check so far in
 * this process (always counted since round 6: the increment sits on the redo path only; one blocking 4-byte read-back per call;
 * -1: no device / allocation failure).  bench.py reports the count of its timed region as `attn_redo_count`. */
long gyre_debug_attn_redo_count(void);
/* Tuning only: per-workgroup cycle stamps of the fused cross-attention kernel's phase boundaries (8 x uint64 per workgroup of the
 * last launch) into a caller-allocated device buffer; NULL = off (tools/r06_xattn_phases.py). */
int gyre_debug_xattn_stamps(void* dev_buf);
/* Tuning only.  Ablations (results are garbage): bit0 = skip the operand loads inside the K loop, bit1 = skip the MFMAs,
 * bit2 = no epilogue.  Planner switches for same-box A/B runs (results stay valid): bit8 = default tile order, bit9 = conv
 * zero padding from a zero page in the pipelined kernel, bit10 = no pipelined (32x32x16) tile configs, bit11 = LayerNorm as a
 * separate pass (no fold into the consuming GEMM), bit15 = folded LayerNorm takes its row statistics from a separate pass
 * instead of the producing GEMM's epilogue, bit16 = the GEGLU FF1 keeps its separate LayerNorm, bit17 = GroupNorm keeps its own
 * statistics pass (no statistics from the producing conv / GEMM), bit19 = slab-outer GEGLU epilogue of the folded-LayerNorm
 * FF1, bit20 = no statistics epilogue on the 128x160 tile, bit21 = no A-resident kernel (K = 320 / 640 linear problems go to the
 * 8-wave tiles as before), bit22 = A-resident kernel also for the C x C projections (N < 3 K),
 * bit12 = the Transformer2D GroupNorm keeps its apply pass (no fold into proj_in), bit23 = no pipelined 256x320 tile
 * for 3x3 convs whose grid of it has 128 - 159 workgroups (the 48x48 level of a 768 px request), bit5 = no small-problem kernel
 * (tile config 32, kernels_gemm_sm.hip): small linear problems go to the register-staged 4-wave tiles as before, bit6 = it does not take
 * the long-K few-row linear problems from the split-K path.  Round 5: bit24 = the two-stage K loop in the 8-wave kernels' linear mode
 * (instead of the 2 - 4 stage LDS ring with counted waits), bit25 = the deep ring at one workgroup per CU also where two 2-stage
 * workgroups would fit, bit26 = row-major weights everywhere (no blocked weight copies for the LDS-DMA kernels), bit27 = the two-stage K loop for the
 * 128x160 tile's 3x3 convolutions (instead of the same ring).  Round 6: bit13 = no fused cross-attention kernel (kernels_xattn.hip:
 * to_q -> attention -> to_out of SD1.x's 64x64 level in one launch): the three-launch chain as before.  (Bit 7 - the small kernel's
 * LayerNorm fold - went with the code it switched on; the bit is reused:) bit7 = the 1x1 shortcut of a channel-changing / concat resnet
 * stays its own launch instead of riding inside conv2 as extra K steps of the pipelined tile (round 6).
 * Epilogue ablations (garbage): bit3 = no GELU, bit4 = no stores.
 * These switches are PER CALLING THREAD (thread-local, like gyre_set_batch_invariant): they never change what another thread's
 * handle computes. */
int gyre_debug_gemm_ablation(int bits);

/* ---- batch-invariant mode ------------------------------------------------
 * The reference asserts that an image does not depend on what shares its batch (tests/batch_independance.py:15-27)
 * and splits a request into sub-batches by available memory (services/generate.py:977-990,1049-1091).  All kernels
 * here are batch-position independent and every GEMM tile configuration sums in the same order, but the split-K
 * factor of the deep (16x16 / 8x8) UNet levels is chosen from the tile count and therefore from the batch size.
 * canonical_samples > 0 plans that factor as if every call held canonical_samples batch entries (16 = the
 * 8-images-with-CFG call of the headline configuration): results become bit-identical for ANY split of a batch
 * over GPUs or sub-batches, while calls much smaller than the canonical size fill the chip less well.
 * 0 (default) = plan for the actual batch.  The setting belongs to the CALLING THREAD (one thread drives one device slot in
 * the reference, manager.py:2107-2139): it affects the forwards and workspace queries that thread issues, nothing else.
 * Returns the previous value. */
int gyre_set_batch_invariant(int canonical_samples);
int gyre_get_batch_invariant(void);

/* ---- single operators (kernel-level parity tests and profiling) --------- */
/* All tensors NHWC / row-major in the library's 16-bit storage type unless noted ("bf16" in the names and comments below reads "fp16"
 * for libgyre_hip_f16.so: gyre_storage_dtype); f32 for norm affine, bias. */
int gyre_op_groupnorm(void* stream, const void* x, const void* x2, int C1, int B, int HW, int C, int groups,
                      const float* gamma, const float* beta, float eps, int silu,
                      void* workspace, size_t workspace_bytes, void* y);
size_t gyre_op_groupnorm_workspace(int B, int HW, int C, int groups);
/* GroupNorm statistics from the PRODUCER of a tensor.  On the UNet's large feature maps the kernels that write a tensor a
 * GroupNorm will read (3x3 convs incl. their split-K reduction, the transformer's proj_out) also leave, per block of `rows`
 * consecutive output rows and per `unit` of consecutive channels, the sum and the sum of squares of the bf16 values they
 * stored: stats_out [M / rows][N / unit][2] f32, fixed summation order, no atomics.  rows is dictated by the kernel the
 * planner picks (returned through rows_out; GYRE_ERR_UNSUPPORTED when that kernel cannot: then the consumer runs its own
 * statistics pass); unit divides every GroupNorm group that will read the tensor - alone or as part of a skip concat.
 * gyre_op_groupnorm_colstats is the consumer: no statistics pass, mean / rstd are finished from the partials of x (and x2) in
 * the prologue of the one launch that normalises.  workspace of the producers: gyre_op_gemm_splitk_bytes (0 = none; for a conv
 * it assumes the square stride-1 geometry sqrt(M / B) x sqrt(M / B) - any other conv: pass at least 4 * 16 * M * N bytes, the largest
 * split factor's slabs).  Non-positive sizes, a unit / rows_per_sample of 0 or one that does not divide N / M: GYRE_ERR_INVALID. */
size_t gyre_op_gemm_splitk_bytes(int conv, int M, int N, int K, int B);
int gyre_op_conv3x3_colstats(void* stream, const void* x, int B, int Hi, int Wi, int Cin, const void* w_repacked, int Cout,
                             const float* bias, const void* residual, int stride, int ups, int unit, void* y, float* stats_out,
                             size_t stats_bytes, void* workspace, size_t workspace_bytes, int* rows_out);
int gyre_op_linear_colstats(void* stream, const void* x, int M, int K, const void* w_bf16_rowmajor, int N, const float* bias,
                            const void* residual, int rows_per_sample, int unit, void* y, float* stats_out, size_t stats_bytes,
                            void* workspace, size_t workspace_bytes, int* rows_out);
int gyre_op_groupnorm_colstats(void* stream, const void* x, const void* x2, int C1, int B, int HW, int C, int groups,
                               const float* gamma, const float* beta, float eps, int silu, const float* cs_x, int cs_x_chunks,
                               const float* cs_x2, int cs_x2_chunks, int unit, void* workspace, size_t workspace_bytes, void* y);
int gyre_op_layernorm(void* stream, const void* x, int M, int C, const float* gamma, const float* beta,
                      float eps, void* y);
/* y[M,N] = x[M,K] @ w[N,K]^T (+bias) (+residual); geglu!=0: w holds [2N,K], y = val*gelu(gate) */
int gyre_op_linear(void* stream, const void* x, int M, int K, const void* w_bf16_rowmajor, int N,
                   const float* bias, const void* residual, int geglu, void* y);
/* y = LayerNorm(x; gamma, beta, eps) @ w^T (+bias) with the normalisation folded into the GEMM: one streaming pass writes each
 * row's (rstd, rstd * mean), the GEMM multiplies the RAW rows by gamma-scaled weights and its epilogue applies
 * rstd * acc - rstd * mean * colsum[n] + bias'[n]; the normalised tensor is never written or re-read.  Folded weights,
 * column constants and row statistics live in `workspace` (gyre_op_ln_linear_workspace(rows of w, K, M) bytes).  This is how
 * the UNet's transformer blocks run norm1 -> Q|K|V, norm2 -> to_q and norm3 -> GEGLU (third-party BasicTransformerBlock,
 * reached from gyre/pipeline/unet/core.py:274).  qkv_tokens > 0: w = [3K][K] rows Q | K | V, y = Q | K ([M][2K]), vt_out = V^T
 * as in gyre_op_qkv.  GYRE_ERR_UNSUPPORTED when the planner's tile configuration for the shape has no folded form (the model
 * then runs the separate gyre_op_layernorm pass). */
size_t gyre_op_ln_linear_workspace(int w_rows, int K, int M);
/* row_parts / n_parts: instead of the streaming statistics pass, take the partial row sums the PRODUCER of x left
 * (gyre_op_linear_rowstats); NULL / 0: run the pass. */
int gyre_op_ln_linear(void* stream, const void* x, int M, int K, const float* gamma, const float* beta, float eps,
                      const void* w_bf16_rowmajor, int N, const float* bias, int geglu, int qkv_tokens, void* vt_out, int ldt,
                      const float* row_parts, int n_parts, void* workspace, size_t workspace_bytes, void* y);
/* gyre_op_linear that also leaves, per row, the (sum, sum of squares) of the bf16-rounded outputs of each of its N tiles:
 * stats_out [parts][M][2] f32, parts = gyre_op_linear_rowstats_parts(...) (0 = the planner's kernel for the shape has no such
 * epilogue -> GYRE_ERR_UNSUPPORTED).  In the UNet the producers of a transformer block's internal tensors (proj_in, the two
 * attention to_out projections, FF2 of a previous block) hand these to the LayerNorm folded into the next GEMM. */
int gyre_op_linear_rowstats_parts(int M, int K, int N, int has_residual);
int gyre_op_linear_rowstats(void* stream, const void* x, int M, int K, const void* w_bf16_rowmajor, int N, const float* bias,
                            const void* residual, void* y, float* stats_out);
/* V^T form used by attention: y[(b*N + n)*ldt + tok] = (x @ w^T + bias)[b*tokens + tok][n] */
int gyre_op_linear_t(void* stream, const void* x, int M, int K, const void* w_bf16_rowmajor, int N,
                     const float* bias, int tokens_per_batch, int ldt, void* y);
/* 3x3 conv NHWC, weights already repacked [Cout][3][3][Cin] bf16; ups!=0 fuses nearest-2x upsample
 * of the input; stride 1|2; pad 1 (pad 0 + bottom/right zero pad when asym!=0, VAE downsample). */
int gyre_op_conv3x3(void* stream, const void* x, int B, int Hi, int Wi, int Cin, const void* w_krsc, int Cout,
                    const float* bias, const void* residual, int stride, int ups, int asym, void* y);
/* A resnet's conv2 with its 1x1 shortcut folded in (round 6): y = conv3x3(x; stride 1, pad 1) + (sx | sx2) w_sc^T + bias + bias_sc in ONE
 * launch - the shortcut's channels are extra K steps of the pipelined 256x320 convolution tile, same accumulators, one rounding.
 * sx [B][H][W][C1], sx2 [B][H][W][C2] or NULL with C2 = 0 (the skip concatenation of an up block is never materialised), w_sc
 * [Cout][C1 + C2] repacked.  Cin, C1, C2 multiples of 64.  ws: Cout * (9 Cin + C1 + C2) * 2 + Cout * 4 + 512 bytes.
 * GYRE_ERR_UNSUPPORTED where the planner's kernel for the shape does not know the form (the model then runs the two launches). */
int gyre_op_conv3x3_shortcut(void* stream, const void* x, int B, int H, int W, int Cin, const void* w_krsc, int Cout, const float* bias,
                             const void* sx, int C1, const void* sx2, int C2, const void* w_sc, const float* bias_sc, void* ws,
                             size_t ws_bytes, void* y);
/* The output convolution of the UNet / VAE decoder (3x3, stride 1, zero padding 1, Cout <= 16) written as NCHW y
 * [B][Cout][H][W] of dtype y_dtype (0 f32, 1 bf16, 2 f16), as the models' last launch does: the dedicated kernel
 * (kernels_conv_out.hip) where Cin % 64 == 0, the tile kernels otherwise; force_tiles != 0 takes the tile kernels anyway. */
int gyre_op_conv3x3_nchw(void* stream, const void* x, int B, int H, int W, int Cin, const void* w_krsc, int Cout,
                         const float* bias, void* y, int y_dtype, int force_tiles);
/* Repack helpers used by the tests to build the layouts above from PyTorch-layout f32 tensors */
int gyre_op_repack_conv_weight(void* stream, const float* w_oihw, int Cout, int Cin, int KH, int KW, int Cin_pad,
                               void* w_krsc_bf16);
int gyre_op_repack_linear_weight(void* stream, const float* w_oi, int O, int I, int geglu_interleave, void* w_bf16);
int gyre_op_repack_bias(void* stream, const float* b, int n, int geglu_interleave, float* out);
/* o[B,Nq,H*D] = softmax(q k^T * D^-1/2) v ; q[B,Nq,H*D] (ldq), k[B,Nk,H*D] (ldk), vt[B,H*D,ldvt] (V transposed) */
int gyre_op_attention(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                      int B, int heads, int Nq, int Nk, int D, void* o, int ldo);
/* Fused Q|K|V projection as the UNet self-attention runs it: x[M,C] (bf16) times w_qkv[3C,C] (rows Q | K | V, repacked
 * with gyre_op_repack_linear_weight) -> qk_out[M,2C] row-major and vt_out[M/tokens][C][ldt] = V transposed per batch
 * entry.  Needs an 8-wave tile configuration whose wave tiles align with column 2C (GYRE_ERR_UNSUPPORTED otherwise;
 * the model then issues two GEMMs). */
int gyre_op_qkv(void* stream, const void* x, int M, int C, const void* w_qkv, int tokens, void* qk_out, void* vt_out, int ldt);
/* Same, k_prescaled != 0: k already holds k * log2(e)/sqrt(D) (the UNet folds that factor into its to_k weights in
 * fp32 before their bf16 rounding), which lets the kernel take exp2 of the matrix-core output directly. */
int gyre_op_attention_ex(void* stream, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                         int B, int heads, int Nq, int Nk, int D, void* o, int ldo, int k_prescaled);
/* ToMe merge of one self-attention's keys / values: k[B,N,ldk], v[B,N,ldv] (bf16 rows of C channels) ->
 * k_out[B,N-r,C], vt_out[B,C,ldvt] (values transposed, columns >= N-r zero); optional order_out / node_idx_out [B,N/2]
 * (int32, dev) receive the a-token ranking and every a token's best match.  r is clipped to N / 2. */
/* Input gradient of gyre_vae_decode (clipguided.py:366-386 decodes latent cut-outs under autograd): writes the decoded
 * image and d_z = (d image / d z)^T d_image. */
size_t gyre_vae_decode_vjp_workspace_bytes(gyre_vae* h, int B, int h_lat, int w_lat);
int gyre_vae_decode_vjp(gyre_vae* h, void* stream, const void* z_nchw, int z_dtype, int B, int h_lat, int w_lat,
                        const void* d_image_nchw, int d_dtype, void* workspace, size_t workspace_bytes,
                        void* image_out_nchw, int out_dtype, void* dz_out_nchw, int dz_dtype);

/* ---- input-gradient kernels, exposed for the parity tests (tests/test_gpu_vjp.py); activations / gradients bf16 ----
 * groupnorm_bwd: x (|| x2) is the forward INPUT, dy[B,HW,C] the gradient of the (activated) output; dx[B,HW,C1],
 *   dx2[B,HW,C-C1]; addend (optional, [B,HW,C1]) is added to dx.
 * layernorm_bwd: same for rows of [M,C].  geglu_bwd: pre[M,2F] in the packed column order of
 *   gyre_op_repack_linear_weight(geglu=1), dy[M,F], dpre[M,2F].
 * attention_bwd: q/k/v/o/d_o row-major [B,N,ld] with head h at column h*D (v is NOT transposed here); dk == NULL
 *   computes dq only (cross-attention). */
size_t gyre_op_groupnorm_bwd_workspace(int B, int HW, int C, int groups);
int gyre_op_groupnorm_bwd(void* stream, const void* x, const void* x2, int C1, int B, int HW, int C, int groups,
                          const float* gamma, const float* beta, float eps, int silu, const void* dy, const void* addend,
                          void* workspace, size_t workspace_bytes, void* dx, void* dx2);
int gyre_op_layernorm_bwd(void* stream, const void* x, const void* dy, int M, int C, const float* gamma, float eps,
                          const void* addend, void* dx);
int gyre_op_geglu_bwd(void* stream, const void* pre, const void* dy, int M, int F, void* dpre);
size_t gyre_op_attention_bwd_workspace(int B, int heads, int Nq, int Nk, int D);
int gyre_op_attention_bwd(void* stream, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                          const void* o, int ldo, const void* d_o, int lddo, int B, int heads, int Nq, int Nk, int D,
                          int k_prescaled, void* workspace, size_t workspace_bytes, void* dq, int lddq, void* dk, int lddk,
                          void* dv, int lddv);
size_t gyre_op_tome_workspace(int B, int N, int C);
int gyre_op_tome_merge(void* stream, const void* k, int ldk, const void* v, int ldv, int B, int N, int C, int r,
                       void* workspace, size_t workspace_bytes, void* k_out, void* vt_out, int ldvt,
                       int32_t* order_out, int32_t* node_idx_out);
/* The cross-attention block of a transformer block as ONE launch (round 6, kernels_xattn.hip; the model runs it for the attn2 module of
 * diffusers' BasicTransformerBlock at SD1.x's 64x64 level): out = softmax(LayerNorm(x) Wq^T . K^T) V Wo^T + bo + x and, if row_stats
 * is not NULL, per row the (sum, sum of squares) of the rounded outputs.  x [M][C] = the rows BEFORE the LayerNorm (M = B * tokens),
 * wq / wo [C][C] repacked, k [B][Nk][C] already multiplied by log2(e) / sqrt(C / heads), vt [B][C][ldvt] = V transposed (ldvt >= Nk
 * rounded up to 8).  ws: gyre_op_ln_linear_workspace(C, C, M) bytes.  GYRE_ERR_UNSUPPORTED outside the kernel's domain (C = 320,
 * 8 heads, Nk <= 80, tokens % 128 == 0, M / 128 >= 256). */
int gyre_op_cross_attention_block(void* stream, const void* x, int M, int tokens, int C, int heads, const float* gamma,
                                  const float* beta, float eps, const void* wq, const void* k_prescaled, const void* vt, int Nk,
                                  int ldvt, const void* wo, const float* bo, void* ws, size_t ws_bytes, void* out, float* row_stats);
int gyre_op_nchw_to_nhwc(void* stream, const void* x, int dtype, int B, int C, int HW, int Cpad, void* y_bf16);
/* Device-side memcpy-rate probe used by bench.py to calibrate the HBM roofline on the box. */
int gyre_op_copy_probe(void* stream, const void* src, void* dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GYRE_HIP_H */
