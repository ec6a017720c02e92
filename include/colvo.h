/* colvo.h -- C-ABI of the MI355X-native ColVO DCDP+LCC training hot path (libcolvo.so).
 *
 * The upstream reference (HNUicda/CoIVO, /root/reference) defines NO plugin, operator or FFI
 * interface -- it ships README.md and three figures only.  Each entry point
 * below therefore cites the README sentence that names the concept it implements and the function
 * of the frozen specification (oracle/colvo_spec.py) whose results it must reproduce; there is no
 * reference file:line to replace.
 *
 * Conventions
 *   - plain C types only: device pointers + sizes; `stream` is a hipStream_t passed as void*.
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates,
 *     frees or keeps a pointer after return.
 *   - every call only ENQUEUES work on `stream` and returns; no internal synchronisation, no
 *     global mutable state (colvo_run_commands keeps a ring of fork/join events) -> safe under one-process-per-GPU data parallel and under hipGraph
 *     capture.
 *   - return 0 on success, a non-zero hipError_t-style code otherwise; the message is available
 *     from colvo_last_error() (thread-local).  Nothing throws across the ABI.
 *   - images / depth at the boundary: NCHW, contiguous, fp32.  Feature maps inside the conv stack:
 *     NHWC, fp32 (COLVO_F32) or bf16 (COLVO_BF16).
 */
#ifndef COLVO_H_
#define COLVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COLVO_ABI_VERSION 10

typedef void* colvo_stream_t; /* hipStream_t */

enum { COLVO_F32 = 0, COLVO_BF16 = 1 };

int colvo_abi_version(void);
const char* colvo_last_error(void);

/* ------------------------------------------------------------------------------------------- *
 * a3..a7  fused  project -> bilinear-sample -> LCC-recalibrate -> SSIM/L1       *
 *   concept: README.md:1 "Photometric Consistency", README.md:7 "alignment of geometric       *
 *   projections between consecutive frames", README.md:5,7 LCC "recalibrating the luminosity  *
 *   values of adjacent frames".   spec: oracle/colvo_spec.py photometric_loss().               *
 * ------------------------------------------------------------------------------------------- */

/* Number of floats of scratch the fwd / bwd calls need for a [B,3,H,W] problem. */
size_t colvo_warp_loss_workspace_floats(int B, int H, int W);

/* Forward.  tgt, ref [B,3,H,W]; depth [B,1,H,W]; pose [B,6]; K [B,3,3]; lcc_a, lcc_b [B].
 * loss_state[4] (device) receives { loss, 1/max(3*n_valid,1), n_valid, masked sum = loss * max(3*n_valid,1) }.  H, W >= 2. */
int colvo_warp_loss_fwd(const float* tgt, const float* ref, const float* depth, const float* pose,
                        const float* K, const float* lcc_a, const float* lcc_b,
                        int B, int H, int W, float ssim_weight,
                        float* workspace, float* loss_state, colvo_stream_t stream);

/* Backward (recomputes the warp; nothing but loss_state is saved by the forward).
 * grad_loss: device scalar dL/dloss.  Outputs: d_depth [B,1,H,W], d_pose [B,6], d_a [B], d_b [B]. */
int colvo_warp_loss_bwd(const float* tgt, const float* ref, const float* depth, const float* pose,
                        const float* K, const float* lcc_a, const float* lcc_b,
                        int B, int H, int W, float ssim_weight,
                        const float* loss_state, const float* grad_loss,
                        float* workspace, float* d_depth, float* d_pose, float* d_a, float* d_b,
                        colvo_stream_t stream);

/* Training path: the loss and its UNNORMALISED gradients in one pass (the backward kernel evaluates everything the
 * forward does).  loss_state as in colvo_warp_loss_fwd; d_depth_raw [B,1,H,W] and grad_partials [B*14] are handed to
 * colvo_warp_loss_fused_bwd, which applies dL/dloss / max(3 n_valid, 1) and writes the four gradients.
 * grad_unit (8*B floats, may be NULL) receives the unnormalised pose / LCC gradients in PoseNet's planar output layout
 * [d_pose B x 6 | d_a B | d_b B] for a consumer that applies the two scale factors itself (colvo_pose_head_bwd). */
int colvo_warp_loss_fused(const float* tgt, const float* ref, const float* depth, const float* pose,
                          const float* K, const float* lcc_a, const float* lcc_b,
                          int B, int H, int W, float ssim_weight,
                          float* workspace, float* loss_state, float* d_depth_raw, float* grad_partials,
                          float* grad_unit, colvo_stream_t stream);
int colvo_warp_loss_fused_bwd(const float* loss_state, const float* grad_loss, const float* d_depth_raw,
                              const float* grad_partials, const float* pose, int B, int H, int W,
                              float* d_depth, float* d_pose, float* d_a, float* d_b, colvo_stream_t stream);
/* The same without the pass over d_depth: only d_pose [B,6], d_a [B], d_b [B].  For callers that hand d_depth_raw and
 * the two scale factors (grad_loss[0], loss_state[1]) to the consumer of the depth gradient instead
 * (colvo_depth_head_bwd_parts): the normalisation is then applied where d_depth is read anyway. */
int colvo_warp_loss_fused_bwd_params(const float* loss_state, const float* grad_loss, const float* grad_partials,
                                     const float* pose, int B, float* d_pose, float* d_a, float* d_b,
                                     colvo_stream_t stream);

/* Data parallel with the spec's batch normalisation (oracle/SPEC.md section 5: ONE masked mean over the whole batch).  The caller
 * adds loss_state[2..3] (valid pixels, masked sum) up over its `world` ranks -- one all-reduce of two floats -- and this call turns
 * the state into the global one: loss_state[0] = the loss of the whole batch, loss_state[1] = world / max(3 n_global, 1), so that
 * (1 / world) x the sum of the ranks' gradients is the gradient of that loss.  Every backward entry point above reads the scale
 * from loss_state[1].  world = 1 leaves the state bit for bit as the forward wrote it. */
int colvo_warp_loss_rescale(float* loss_state, int world, colvo_stream_t stream);
/* The same, for a caller that takes the exchange OFF the critical path (round 6): the backward pass has already run on the
 * UNNORMALISED gradients (its consumers were handed a scale of 1 in place of loss_state[1]) while the two floats were being
 * all-reduced; every parameter gradient is linear in the scale, so it is applied once, behind the gradient all-reduce, by the
 * optimizer: *scale_out = world / max(3 n_global, 1) is what colvo_adam_pack_step_scaled multiplies into its grad_scale.
 * loss_state[0] / [1] are rewritten as by colvo_warp_loss_rescale (the reported loss becomes the whole batch's). */
int colvo_warp_loss_rescale_to(float* loss_state, int world, float* scale_out, colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * SURVEY.md §8f-1  geometric consistency (README.md:1 "Considering Geometric and Photometric   *
 *   Consistency", :7 "alignment of geometric projections").  spec: geometric_consistency_loss. *
 * ------------------------------------------------------------------------------------------- */
/* depth_t, depth_r [B,1,H,W]; pose [B,6]; K [B,3,3].  loss_state[4] = { loss, 1/max(n_valid,1), n_valid, 0 }.
 * workspace: colvo_geo_loss_workspace_floats(B,H,W) floats.  Backward recomputes the forward; d_depth_r is zeroed by the
 * call and accumulated with float atomics (the four taps of every sample). */
size_t colvo_geo_loss_workspace_floats(int B, int H, int W);
int colvo_geo_loss_fwd(const float* depth_t, const float* depth_r, const float* pose, const float* K,
                       int B, int H, int W, float* workspace, float* loss_state, colvo_stream_t stream);
int colvo_geo_loss_bwd(const float* depth_t, const float* depth_r, const float* pose, const float* K,
                       int B, int H, int W, const float* loss_state, const float* grad_loss, float* workspace,
                       float* d_depth_t, float* d_depth_r, float* d_pose, colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * SURVEY.md §8f-2  edge-aware smoothness of the inverse depth (spec: smoothness_loss) and the  *
 *   2x2 average pooling of the multi-scale photometric term (spec: downsample2).               *
 * ------------------------------------------------------------------------------------------- */
/* depth [B,1,H,W], img [B,3,H,W]; workspace: 2 * B * ceil(H*W/256) floats; loss: one device float. */
int colvo_smooth_loss_fwd(const float* depth, const float* img, int B, int H, int W, float* workspace, float* loss,
                          colvo_stream_t stream);
int colvo_smooth_loss_bwd(const float* depth, const float* img, int B, int H, int W, const float* grad_loss,
                          float* d_depth, colvo_stream_t stream);
/* x [planes,H,W] -> y [planes,H/2,W/2] (mean of each 2x2 block), and its gradient dy -> dx (H, W: the INPUT extent). */
int colvo_avgpool2_fwd(const float* x, int planes, int H, int W, float* y, colvo_stream_t stream);
int colvo_avgpool2_bwd(const float* dy, int planes, int H, int W, float* dx, colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * SURVEY.md §8f-1 + §8f-2 in ONE call: the widened objective (spec: dcdp_full_loss)            *
 *   mean_s photometric_loss(pyramid level s) + geo_weight * geometric_consistency_loss        *
 *   + smooth_weight * smoothness_loss, value AND every gradient.                               *
 *   README.md:1 "Considering Geometric and Photometric Consistency", :7.                       *
 * ------------------------------------------------------------------------------------------- */
/* Forward: 3 launches, no host synchronisation -- (1) smoothness (value partials + raw gradient) and two levels of the 2x2-average
 * pyramid of both frames and the target depth in one launch (a fourth level takes one more), (2) the one-pass photometric kernel
 * over ALL levels in one launch (loss + unnormalised gradients; the intrinsics of level s are scaled inside the kernel), whose
 * level-0 part ALSO evaluates the geometric-consistency term on the projection and bilinear taps it has anyway (the reference
 * frame's depth depth_r is a fourth sampled plane), (3) one finalize.  The gradient w.r.t. depth_r is scattered as 64-bit fixed-point
 * atomics (2^-32 units): bit-reproducible.  tgt, ref [B,3,H,W]; depth_t, depth_r [B,1,H,W] (depth_r may be NULL when
 * geo_weight == 0); H, W divisible by 2^(num_scales-1), 1 <= num_scales <= 4.  workspace: colvo_full_objective_workspace_floats()
 * floats, 16-byte aligned, kept untouched until the backward call; loss: one device float.
 * Backward: ONE launch; grad_loss: device scalar dL/dloss; d_depth_t, d_depth_r [B,1,H,W] (d_depth_r may be NULL when
 * geo_weight == 0), d_pose [B,6], d_a, d_b [B].  colvo_full_objective_terms: *state -> 32 device floats inside the workspace:
 * [0] total, [1] geometric term, [2] smoothness term, [4+4s ..] level s: {photometric loss, (1/S)/max(3 n_valid,1), n_valid}. */
size_t colvo_full_objective_workspace_floats(int B, int H, int W, int num_scales);
int colvo_full_objective_fwd(const float* tgt, const float* ref, const float* depth_t, const float* depth_r,
                             const float* pose, const float* K, const float* lcc_a, const float* lcc_b, int B, int H, int W,
                             int num_scales, float ssim_weight, float geo_weight, float smooth_weight, float* workspace,
                             float* loss, colvo_stream_t stream);
int colvo_full_objective_bwd(const float* workspace, const float* grad_loss, const float* pose, int B, int H, int W,
                             int num_scales, float geo_weight, float smooth_weight, float* d_depth_t, float* d_depth_r,
                             float* d_pose, float* d_a, float* d_b, colvo_stream_t stream);
int colvo_full_objective_terms(const float* workspace, int B, int H, int W, int num_scales, const float** state);

/* Un-fused debugging entry (spec: inverse_warp()).  ref [B,C,H,W] -> warped [B,C,H,W], valid [B,1,H,W]. */
int colvo_inverse_warp(const float* ref, const float* depth, const float* pose, const float* K,
                       int B, int C, int H, int W, float* warped, float* valid, colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * a1, a2  conv blocks of DepthNet / PoseNet (README.md:5,7 DCDP "depth and pose estimation")  *
 *   spec: oracle/colvo_spec.py DepthNet / PoseNet (F.conv2d + F.relu + nearest up + concat).   *
 *   One implicit-GEMM convolution on NHWC feature maps, MFMA on gfx950; the decoder's nearest  *
 *   2x up-sample and the skip concat are folded into the input gather.                         *
 * ------------------------------------------------------------------------------------------- */
typedef struct ColvoConvDesc {
    int32_t dtype;      /* COLVO_F32 | COLVO_BF16: type of x0, x1, y, dy, dx and of the packed weights */
    int32_t B;          /* images */
    int32_t Ho, Wo;     /* output spatial size */
    int32_t Cout;       /* multiple of 8 */
    int32_t ksize;      /* 3 or 1 (pad = ksize/2) */
    int32_t stride;     /* 1 or 2 */
    int32_t relu;       /* 1: y = max(conv + bias, 0) */
    /* source 0 (and optional source 1, channel-concatenated after source 0) */
    int32_t C0;         /* channels of x0, multiple of 8 */
    int32_t up0;        /* 1: x0 is stored at half resolution and read through nearest 2x up-sampling */
    int32_t C1;         /* channels of x1 (0 = none), multiple of 8 */
    int32_t up1;
    int32_t Hi, Wi;     /* spatial size of the (virtual, i.e. after up-sampling) conv input */
} ColvoConvDesc;

/* y[B,Ho,Wo,Cout] = act(conv(cat(x0,x1)) + bias).  w_fwd: weights packed [Cout][ksize*ksize][C0+C1]
 * in `dtype`; bias fp32 [Cout]. */
int colvo_conv_fwd(const ColvoConvDesc* d, const void* x0, const void* x1, const void* w_fwd,
                   const float* bias, void* y, colvo_stream_t stream);

/* Input gradient of source `src` (0 or 1).  dy[B,Ho,Wo,Cout] must already be the gradient w.r.t. the
 * pre-activation (see relu_mask below).  w_bwd: weights packed [C0+C1][ksize*ksize][Cout].
 * dx has the STORED shape of the source (half resolution when up-sampled).
 * relu_mask (may be NULL): the stored output of the layer that produced this source; when given, dx is
 * multiplied by (relu_mask > 0), i.e. dx becomes that producer's pre-activation gradient.
 * accumulate: 0 = overwrite dx, 1 = dx += (fan-out of skip connections). */
int colvo_conv_dgrad(const ColvoConvDesc* d, int src, const void* dy, const void* w_bwd,
                     const void* relu_mask, void* dx, int accumulate, colvo_stream_t stream);
/* The input gradients w.r.t. BOTH sources of a two-source (concat) layer in one launch: dx0 [B][Hi][Wi][C0], dx1 [..][C1];
 * relu_mask0 / relu_mask1 as in colvo_conv_dgrad (may be NULL).  Same results as two colvo_conv_dgrad calls (which it falls
 * back to for strided / up-sampled layers or when C0 is not a multiple of 32). */
int colvo_conv_dgrad_both(const ColvoConvDesc* d, const void* dy, const void* w_bwd, const void* relu_mask0,
                          const void* relu_mask1, void* dx0, void* dx1, colvo_stream_t stream);

/* Input gradient of a 3x3 conv (stride 1 or 2, ONE directly stored source, no ReLU behind it: a network input) w.r.t. a few of its
 * input channels only, as fp32 planes: dst[c][B][1][Hi][Wi] for channels c_begin .. c_begin + c_count - 1 (c_count 1, 2 or 4; every
 * channel a contiguous [B,1,Hi,Wi] tensor of its own).  w_master: the fp32 weights [Cout][9][C0].  PoseNet's first layer: only the
 * two depth channels of its 8-channel input gradient are wanted (DCDP coupling, README.md:7); this replaces the full input gradient +
 * an unpack pass (29 us -> 5 us on the critical path between the two networks' backward passes).  accumulate: dst += . */
int colvo_conv_dgrad_planes(const ColvoConvDesc* d, const void* dy, const float* w_master, int c_begin, int c_count, float* dst,
                            int accumulate, colvo_stream_t stream);

/* The narrow full-resolution layer (bf16, 16 -> 16, stride 1, ReLU, one directly stored source: DepthNet's iconv1) AND the 3x3
 * 16 -> 1 depth head behind it in ONE pass (csrc/fwd16.hip): y = relu(conv(x, w_fwd) + bias) is written (the backward pass needs it)
 * and the head is evaluated on it from LDS -- depth = 1 / (1/max + (1/min - 1/max) sigmoid(conv(y, head_w) + head_b)), [B,1,H,W] fp32 --
 * so the 42 MB tensor is not read back (colvo_conv_fwd + colvo_depth_head_fwd: 21 + 18 us at 16 frames of 256x320).
 * colvo_conv_head_fused_ok: 1 when the layer qualifies. */
int colvo_conv_head_fused_ok(const ColvoConvDesc* d);
/* pose_in (optional; B = 2 * pairs, target frames first): PoseNet's 8-channel bf16 input [pairs][H][W][8] = [tgt rgb | ref rgb |
 * depth_t | depth_r] -- the depth of image b is ALSO written, rounded to bf16, into channel 6 (b < pairs) or 7 of pair b mod pairs;
 * colvo_pack_stem_pose fills the six rgb channels while it packs DepthNet's own input: PoseNet's packing pass disappears. */
int colvo_conv_head_fused(const ColvoConvDesc* d, const void* x, const void* w_fwd, const float* bias, const float* head_w,
                          const float* head_b, float min_depth, float max_depth, void* y, float* depth, void* pose_in,
                          colvo_stream_t stream);
/* frames [B2][3][H][W] fp32 (B2 = 2 * pairs: target frames, then reference frames) -> stem [B2][H][W][8] bf16 (rgb + 5 zero channels,
 * what colvo_pack_nchw writes) AND pose_in [pairs][H][W][8] bf16: rgb channels 0..2 / 3..5, channels 6 / 7 zeroed (the head's pass
 * writes them afterwards). */
int colvo_pack_stem_pose(const float* frames, int B2, int H, int W, void* stem, void* pose_in, colvo_stream_t stream);

/* Input gradient AND weight / bias gradient of a narrow full-resolution layer in ONE pass (csrc/bwd16.hip): bf16, 16 -> 16 channels,
 * stride 1, one directly stored source -- DepthNet's iconv1.  Both backward kernels of such a layer are HBM-bound and read the same
 * two tensors (dy with a halo; the layer's input x as ReLU mask of dx and as second operand of dw): fused, the layer's backward is 3
 * tensor passes instead of 5.  dx = (relu_mask ? x > 0 : 1) * (dy (*) w_bwd), written (not added); dw / db are ADDED to (fp32
 * atomics: use the separate calls where bitwise repeatability is wanted).  colvo_conv_bwd_fused_ok: 1 when the layer qualifies. */
int colvo_conv_bwd_fused_ok(const ColvoConvDesc* d);
/* head_dpre / head_w (both or neither): the HEAD form for the layer in front of the 3x3 16 -> 1 depth head.  `dy` is then NOT the
 * gradient but the layer's OUTPUT y (post-ReLU) and the gradient is made on the fly from the head's d(pre) plane [B][H][W] (what
 * colvo_depth_head_bwd / _bwd_parts leave in `scratch`; call them with dx = NULL) and its fp32 weights [9][16]:
 * dy[p][c] = (y[p][c] > 0) * sum_t head_w[t][c] * dpre[p + 1 - t] -- the head's input gradient is never written or read back. */
/* head_partials (HEAD form only, optional): the depth head's OWN weight / bias gradient rides along as well -- the kernel has y and
 * d(pre) staged anyway -- as colvo_conv_bwd_fused_head_rows(d) partial rows of 9 * 16 + 1 floats (one per wave, plain stores), which
 * colvo_depth_head_wgrad_reduce then adds to the head's dw [9][16] / db [1] in a fixed order: the 42 MB pass of colvo_depth_head_wgrad
 * over y disappears. */
int colvo_conv_bwd_fused(const ColvoConvDesc* d, const void* dy, const void* w_bwd, const void* x, int relu_mask, void* dx, float* dw,
                         float* db, const float* head_dpre, const float* head_w, float* head_partials, colvo_stream_t stream);
int colvo_conv_bwd_fused_head_rows(const ColvoConvDesc* d);
int colvo_depth_head_wgrad_reduce(const float* partials, int rows, float* dw, float* db, colvo_stream_t stream);
/* The 16-channel depth head's weight / bias gradient by MFMA (bf16 feature maps): partial rows ([colvo_depth_head_wgrad_mfma_rows(B, H,
 * W)][9 * 16 + 1] floats, plain stores) from y [B][H][W][16] and the d(pre) plane, then colvo_depth_head_wgrad_reduce.  d(pre) enters
 * the product rounded to bf16 (as y is); fixed order: bitwise repeatable.  colvo_depth_head_wgrad (VALU) stays the fp32 / generic form. */
int colvo_depth_head_wgrad_mfma_rows(int B, int H, int W);
int colvo_depth_head_wgrad_mfma(const void* y, const float* dpre, int B, int H, int W, float* partials, colvo_stream_t stream);

/* Weight + bias gradient, fp32, ADDED into dw[Cout][ksize*ksize][C0+C1] and db[Cout]
 * (the caller zeroes them once per step). */
int colvo_conv_wgrad(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy,
                     float* dw, float* db, colvo_stream_t stream);
/* The same with a promise: arena_is_zero != 0 says dw and db hold zeros and nothing else writes them while this call runs (the first
 * backward pass behind an optimizer step that cleared the gradients).  Layers whose grid has ONE pixel-range split per weight slab
 * then STORE their sums instead of adding them with fp32 atomics (a third of such a workgroup's life, profiles/r4_wgrad_phases.md);
 * every other layer, and arena_is_zero == 0, is colvo_conv_wgrad.  The result is the same either way. */
int colvo_conv_wgrad_clean(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy,
                           float* dw, float* db, int arena_is_zero, colvo_stream_t stream);

/* Deterministic weight gradients.  The plain calls end in fp32 atomics (one add per weight and pixel-range split): the sum
 * depends on the order the workgroups finish in, in the last bits.  The _det forms give every split a slab of its own in
 * caller-owned scratch (plain stores) and add the slabs in split order with a second small launch: bitwise repeatable run to
 * run, at the price of that launch.  scratch: >= the matching *_scratch_bytes(), no initialisation needed, private to the call
 * while it runs.  colvo_pose_head_bwd_det needs none: one thread owns each weight column and walks the images in order. */
size_t colvo_conv_wgrad_scratch_bytes(const ColvoConvDesc* d);
int colvo_conv_wgrad_det(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw, float* db,
                         void* scratch, size_t scratch_bytes, colvo_stream_t stream);
/* Grouped form (round 4; optional -- nn.*.group_wgrad -- and NOT the default: see the last sentence).  In-kernel stamps put the fp32
 * atomics that end colvo_conv_wgrad at 6-7.7 us of EVERY launch -- a third of the kernel at 16 frames, whatever the number of splits --
 * against 2.3 us for the slab stores of the _det form, whose per-layer second launch then costs ~7 us again
 * (profiles/r4_wgrad_phases.md).  colvo_conv_wgrad_slabs is the FIRST launch of colvo_conv_wgrad_det only (every pixel-range split
 * stores its sums into its slab of `scratch`; dw / db are not touched; a single split stores a slab too), and
 * colvo_wgrad_reduce_group adds the slabs of up to COLVO_WGRAD_GROUP_MAX such calls to their dw / db in ONE launch, in a fixed order:
 * four or five second launches per backward pass instead of 27, bitwise repeatable.  colvo_conv_wgrad_splits: the number of slabs the
 * call will write (what the group entry needs).  The slabs call refuses a batch it would have to slice (a tensor >= 1 GiB: the slices
 * would share the slabs); use _det there.  IN THE TRAINING STEP the form measured 5 % SLOWER than the atomics at 8 pairs and 2 % at 32
 * (the per-layer _det form: 1.6 %): 180 MB of slabs a step go out to HBM and come back beside bandwidth-bound kernels, and a group's
 * launch waits for both weight-gradient streams, while the atomics -- slow per launch -- ride in L2 under the other streams' kernels. */
#define COLVO_WGRAD_GROUP_MAX 16
typedef struct ColvoWgradSlabs {
    const void* scratch;         /* what colvo_conv_wgrad_slabs wrote: [nsplit][Cout*9*Ctot] weight slabs, then [nsplit][Cout] bias slabs */
    float* dw;                   /* [Cout][9][Ctot], added to */
    float* db;                   /* [Cout] or NULL */
    int32_t nsplit, Cout, Ctot, pad_;
} ColvoWgradSlabs;
int colvo_conv_wgrad_splits(const ColvoConvDesc* d);
int colvo_conv_wgrad_slabs(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, void* scratch,
                           size_t scratch_bytes, colvo_stream_t stream);
int colvo_wgrad_reduce_group(const ColvoWgradSlabs* sets, int n, colvo_stream_t stream);
size_t colvo_depth_head_wgrad_scratch_bytes(int B, int H, int W, int C);
int colvo_depth_head_wgrad_det(int dtype, const void* x, const float* dpre, int B, int H, int W, int C, float* dw, float* db,
                               void* scratch, size_t scratch_bytes, colvo_stream_t stream);
int colvo_pose_head_bwd_det(int dtype, const void* x, const float* w, const float* d_pose, const float* d_a, const float* d_b,
                            const float* scale_a, const float* scale_b, int B, int HW, int C, float pose_scale, float lcc_scale,
                            void* dx, float* dw, float* db, colvo_stream_t stream);

/* dy <- dy * (y > 0) in place (first consumer of a ReLU output's gradient when no dgrad produced it). */
int colvo_relu_bwd_inplace(int dtype, const void* y, void* dy, size_t n, colvo_stream_t stream);

/* Master fp32 weights [Cout][kk][Cin] -> the two packed operand layouts in `dtype`:
 * w_fwd [Cout][kk][Cin] and w_bwd [Cin][kk][Cout]. */
int colvo_pack_weights(int dtype, const float* w_master, int Cout, int kk, int Cin,
                       void* w_fwd, void* w_bwd, colvo_stream_t stream);

/* The same for every 3x3 layer of a network in ONE launch.  `table` (device memory) is an array of
 *   struct { int64 w_off, fwd_off, bwd_off; int32 Cout, kk, Cin, blk_begin; }
 * with element offsets into `master` (fp32 arena), `fwd` and `bwd` (flat operand buffers in `dtype`; fwd_off < 0
 * or fwd == NULL skips the forward copy) and the index of the layer's first workgroup; a layer takes
 * kk * ceil(Cout/32) * ceil(Cin/64) workgroups (one 32 x 64 transpose tile of one tap each) and nblocks is their sum. */
int colvo_pack_weights_multi(int dtype, const float* master, const void* table, int nlayers, int nblocks,
                             void* fwd, void* bwd, colvo_stream_t stream);

/* NCHW fp32 planes -> NHWC feature map of `Cpad` channels (zero padded), and back (gradient, fp32 NCHW).
 * src[i] points to an [B,c_i,H,W] tensor; up to 4 sources are concatenated along channels. */
int colvo_pack_nchw(int dtype, const float* const* src, const int32_t* src_channels, int nsrc,
                    int B, int H, int W, int Cpad, void* dst, colvo_stream_t stream);
/* accumulate: bit 0 = add to dst instead of overwriting it; bit 1 = dst is laid out [c_count][B][H][W] (each channel a contiguous
 * [B,1,H,W] tensor of its own) instead of [B,c_count,H,W]. */
int colvo_unpack_nhwc_grad(int dtype, const void* dsrc, int B, int H, int W, int Cpad,
                           int c_begin, int c_count, float* dst_nchw, int accumulate, colvo_stream_t stream);

/* DepthNet head: conv3x3 (C -> 1) + sigmoid + disp_to_depth, output NCHW fp32 [B,1,H,W]; and its
 * backward (d_depth -> dx NHWC [masked by x>0, the producer's ReLU], dw[9*C] += , db[1] +=).
 * scratch: B*H*W floats; it receives the gradient w.r.t. the pre-activation.  dw = db = NULL skips the weight
 * gradient, which colvo_depth_head_wgrad then computes from `scratch` (e.g. on another stream). */
int colvo_depth_head_fwd(int dtype, const void* x, const float* w, const float* bias, int B, int H, int W, int C,
                         float min_depth, float max_depth, float* depth, colvo_stream_t stream);
int colvo_depth_head_bwd(int dtype, const void* x, const float* w, const float* depth, const float* d_depth,
                         int B, int H, int W, int C, float min_depth, float max_depth, float* scratch,
                         void* dx, float* dw, float* db, colvo_stream_t stream);
int colvo_depth_head_wgrad(int dtype, const void* x, const float* dpre, int B, int H, int W, int C,
                           float* dw, float* db, colvo_stream_t stream);
/* colvo_depth_head_bwd with the incoming gradient given in parts (the DCDP step: DepthNet ran on B = 2*Bh images, target
 * frames first): with s = scale_a[0]*scale_b[0], d_depth[b] = g_first[b] + s*g_raw[b] for b < Bh,
 * g_second[b-Bh] + s*g_raw_second[b-Bh] for b >= Bh.  g_first, g_second, g_raw, g_raw_second: [Bh,1,H,W] each, any of them
 * may be NULL (= zero); scale_a, scale_b: device scalars, NULL = 1 (the fused loss hands over d_depth_raw with grad_loss and
 * loss_state + 1; the widened objective hands over finished gradients for both halves).  No concatenated / summed copy is made.
 * dx = NULL: d(pre) only (into `scratch`) -- the input gradient is then made by colvo_conv_bwd_fused's HEAD form. */
int colvo_depth_head_bwd_parts(int dtype, const void* x, const float* w, const float* depth, const float* g_first,
                               const float* g_second, const float* g_raw, const float* g_raw_second, const float* scale_a,
                               const float* scale_b, int B, int H, int W, int C, float min_depth, float max_depth,
                               float* scratch, void* dx, float* dw, float* db, colvo_stream_t stream);

/* PoseNet head: 1x1 conv (C -> 8) + spatial mean + (POSE_SCALE, LCC_SCALE) affine.
 * out (8*B floats) is PLANAR: [ pose B x 6 | lcc_a B | lcc_b B ] so the three results are contiguous views;
 * backward takes the three gradients separately (each may be NULL = zero). */
int colvo_pose_head_fwd(int dtype, const void* x, const float* w, const float* bias, int B, int HW, int C,
                        float pose_scale, float lcc_scale, float* out, colvo_stream_t stream);
/* scale_a, scale_b: device scalars multiplied into the three incoming gradients (NULL = 1): the fused loss hands its
 * gradients over unnormalised together with dL/dloss and 1/max(3 n_valid, 1). */
int colvo_pose_head_bwd(int dtype, const void* x, const float* w, const float* d_pose, const float* d_a,
                        const float* d_b, const float* scale_a, const float* scale_b, int B, int HW, int C,
                        float pose_scale, float lcc_scale, void* dx, float* dw, float* db, colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * a8  Adam over the flat parameter arena (torch.optim.Adam semantics, no weight decay)         *
 * ------------------------------------------------------------------------------------------- */
/* step_count: device int32 holding t BEFORE this step (incremented by the kernel, graph-safe).
 * grad_scale multiplies the gradient first (1/world_size for data-parallel sums).  The four arenas must be 16-byte aligned. */
int colvo_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float grad_scale,
                    int32_t* step_count, colvo_stream_t stream);
/* The same with the 1-based step number supplied by the host (no device counter, one launch).  Not for steps captured into a
 * hipGraph: the captured t would repeat at every replay -- use colvo_adam_step there. */
int colvo_adam_step_t(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                      float lr, float beta1, float beta2, float eps, float grad_scale, int t, colvo_stream_t stream);
/* fp32 <-> bf16 (round to nearest even) over n elements, 16-byte aligned buffers: the staging copies of a reduced-precision gradient
 * transport (ddp.GradBuckets).  to_bf16 != 0: src fp32 -> dst bf16; 0: src bf16 -> dst fp32 (src is declared const float* for both). */
int colvo_cast_f32_bf16(const float* src, void* dst, size_t n, int to_bf16, colvo_stream_t stream);

/* Zero `bytes` bytes of device memory on `stream` (the gradient arenas, once per step). */
int colvo_zero(void* ptr, size_t bytes, colvo_stream_t stream);
/* Both networks' arenas in ONE launch each (a dependent launch costs ~2.7 us on MI355X before it does anything and the small
 * arena alone does not fill the memory system): colvo_adam_step_t over up to COLVO_MAX_ARENAS arenas with one step number, and
 * the zeroing of up to as many buffers (16-byte aligned, multiples of 16 bytes). */
#define COLVO_MAX_ARENAS 4
typedef struct ColvoAdamArena {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    size_t n;
} ColvoAdamArena;
int colvo_adam_step_multi(const ColvoAdamArena* arenas, int count, float lr, float beta1, float beta2, float eps,
                          float grad_scale, int t, colvo_stream_t stream);
int colvo_zero_multi(void* const* ptrs, const size_t* bytes, int count, colvo_stream_t stream);
/* Adam and the operand copies of the updated weights in ONE pass over any number of arenas: colvo_adam_step[_t] followed by
 * colvo_pack_weights_multi, without the second pass over the parameters (12 bytes per parameter) and its launches.  `table`
 * (device memory) is an array of ColvoAdamPackEntry: kind 0 = the weights of one 3x3 layer ([Cout][kk][Cin] fp32 at element
 * w_off of the four arenas; forward copy at element fwd_off of `fwd` (skipped when fwd is NULL or fwd_off < 0), transposed /
 * tap-flipped copy at bwd_off of `bwd`, both in `dtype`), taking kk * ceil(Cout/32) * ceil(Cin/64) workgroups from blk_begin on;
 * kind 1 = a plain range of n elements at w_off (biases, heads, padding: update only), ceil(n / COLVO_ADAM_PLAIN_PER_WG)
 * workgroups.  nblocks = the sum.  step_count: device step counter (incremented by a second launch; for steps captured into a
 * hipGraph) or NULL, then t is the 1-based step number.  An entry with zero_grad != 0 has its range of `grad` set to ZERO once read
 * (grad is then written, its const notwithstanding): the arena-clearing launch of the next step (colvo_zero_multi, 8.5 us + a
 * dependent launch on the ColVO networks) is not needed. */
#define COLVO_ADAM_PLAIN_PER_WG 2048
typedef struct ColvoAdamPackEntry {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    void* fwd;
    void* bwd;
    int64_t w_off, fwd_off, bwd_off, n;
    int32_t Cout, kk, Cin, blk_begin;
    int32_t kind, zero_grad;
} ColvoAdamPackEntry;
int colvo_adam_pack_step(int dtype, const void* table, int nentries, int nblocks, float lr, float beta1, float beta2,
                         float eps, float grad_scale, int32_t* step_count, int t, colvo_stream_t stream);
/* ... with a second gradient factor read from the DEVICE at execution time (NULL: none): every gradient is multiplied by
 * grad_scale * *grad_scale_dev first.  For data parallel with the loss normalisation taken off the critical path
 * (colvo_warp_loss_rescale_to): grad_scale = 1 / world from the host, *grad_scale_dev = world / max(3 n_global, 1) from the
 * two-float all-reduce that ran beside the backward pass. */
int colvo_adam_pack_step_scaled(int dtype, const void* table, int nentries, int nblocks, float lr, float beta1, float beta2,
                                float eps, float grad_scale, const float* grad_scale_dev, int32_t* step_count, int t,
                                colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * SURVEY.md §8f-3  inference: dense depth maps stitched along the integrated trajectory into a  *
 *   world-frame point cloud (README.md:9, :29).  spec: backproject / stitch_point_cloud.        *
 * ------------------------------------------------------------------------------------------- */
/* depth [B,1,H,W], K [B,3,3], cam2world [B,4,4] (row-major; the trajectory integration itself is N tiny 4x4 products and
 * stays on the host) -> points [B,H*W,3] in row-major pixel order. */
int colvo_backproject(const float* depth, const float* K, const float* cam2world, int B, int H, int W,
                      float* points, colvo_stream_t stream);
/* Every `stride`-th pixel (rows and columns) of N frames whose depth is < max_depth, back-projected and compacted in
 * frame-major, row-major order (deterministic: counts + scan, no atomics).  points must hold
 * N*ceil(H/stride)*ceil(W/stride)*3 floats; n_points[0] receives the number of points written.
 * workspace: colvo_stitch_workspace_ints(N,H,W,stride) int32. */
size_t colvo_stitch_workspace_ints(int N, int H, int W, int stride);
int colvo_stitch_point_cloud(const float* depths, const float* K, const float* cam2world, int N, int H, int W,
                             int stride, float max_depth, int32_t* workspace, float* points, int32_t* n_points,
                             colvo_stream_t stream);

/* ------------------------------------------------------------------------------------------- *
 * SURVEY.md §8f-4  frames as a decoder delivers them -> the path's input format.                *
 *   (README.md:13 dataset; spec: resize_frames_u8)                                              *
 * ------------------------------------------------------------------------------------------- */
/* frames [B,h,w,3] uint8 interleaved RGB (device) -> out [B,3,H,W] fp32 in [0,1]: bilinear resize with half-pixel
 * centres (source index = (h/H)*(y+0.5)-0.5 clamped at 0), channel de-interleave and /255 in one pass. */
int colvo_frames_u8_to_f32(const uint8_t* frames, int B, int h, int w, int H, int W, float* out,
                           colvo_stream_t stream);

/* Host side of the same row (no GPU work): n raw frames -- `.npy` files holding [h,w,3] uint8 arrays in C order -- read into
 * dst[n][h][w][3] (the caller's staging buffer, normally pinned) by up to `nthreads` threads; every header is checked against
 * (h, w).  Returns non-zero with the first failing file in colvo_last_error(). */
int colvo_read_npy_u8_frames(const char* const* paths, int n, int h, int w, uint8_t* dst, int nthreads);

/* ------------------------------------------------------------------------------------------- *
 * Command lists: ONE call enqueues a recorded sequence of the entry points above (a network's  *
 * forward or backward) on a main and a side stream -- the host's per-launch cost, not the GPU, *
 * bounded the batch-8 step when every layer was a separate host call.                          *
 * ------------------------------------------------------------------------------------------- */
enum {
    COLVO_CMD_CONV_FWD = 1,      /* p: x0 x1 w_fwd bias y */
    COLVO_CMD_CONV_DGRAD,        /* i: src accumulate; p: dy w_bwd relu_mask dx */
    COLVO_CMD_CONV_WGRAD,        /* p: x0 x1 dy dw db scratch; i: scratch_bytes slabs_only arena_is_zero (scratch != NULL:
                                    colvo_conv_wgrad_det, or with slabs_only colvo_conv_wgrad_slabs; scratch == NULL and arena_is_zero:
                                    colvo_conv_wgrad_clean -- the host patches that integer before each replay) */
    COLVO_CMD_PACK_NCHW,         /* i: dtype c0 c1 c2 c3 nsrc B H W Cpad; p: src0..src3 dst */
    COLVO_CMD_UNPACK_NHWC_GRAD,  /* i: dtype B H W Cpad c_begin c_count accumulate; p: dsrc dst */
    COLVO_CMD_DEPTH_HEAD_FWD,    /* i: dtype B H W C; f: min max; p: x w bias depth */
    COLVO_CMD_DEPTH_HEAD_BWD,    /* i: dtype B H W C; f: min max; p: x w depth d_depth scratch dx dw db */
    COLVO_CMD_DEPTH_HEAD_WGRAD,  /* i: dtype B H W C scratch_bytes; p: x dpre dw db scratch (non-NULL: the _det form) */
    COLVO_CMD_POSE_HEAD_FWD,     /* i: dtype B HW C; f: pose_scale lcc_scale; p: x w bias out */
    COLVO_CMD_POSE_HEAD_BWD,     /* i: dtype B HW C det; f: pose_scale lcc_scale; p: x w d_pose d_a d_b dx dw db scale_a scale_b */
    COLVO_CMD_FORK,              /* side stream waits for the main stream's work so far */
    COLVO_CMD_JOIN,              /* main stream waits for the side stream's work so far */
    COLVO_CMD_DEPTH_HEAD_BWD_PARTS, /* i: dtype B H W C; f: min max; p: x w depth g_first g_second g_raw scale_a scale_b scratch dx g_raw_second */
    COLVO_CMD_CONV_DGRAD_BOTH,    /* p: dy w_bwd relu_mask0 relu_mask1 dx0 dx1 */
    COLVO_CMD_WGRAD_REDUCE_GROUP, /* p: sets (HOST pointer to ColvoWgradSlabs[n], alive as long as the list); i: n */
    COLVO_CMD_CONV_DGRAD_PLANES,  /* p: dy w_master dst; i: c_begin c_count accumulate */
    COLVO_CMD_CONV_BWD_FUSED,     /* p: dy w_bwd x dx dw db head_dpre head_w head_partials; i: relu_mask */
    COLVO_CMD_HEAD_WGRAD_REDUCE,  /* p: partials dw db; i: rows */
    COLVO_CMD_CONV_HEAD_FUSED,    /* p: x w_fwd bias head_w head_b y depth pose_in; f: min_depth max_depth */
    COLVO_CMD_PACK_STEM_POSE,     /* p: frames stem pose_in; i: B2 H W */
    COLVO_CMD_HEAD_WGRAD_MFMA,    /* p: y dpre partials; i: B H W */
    COLVO_CMD_SIDE_SYNC           /* (side command) the side stream in use waits for everything enqueued so far on every other side
                                    stream: what follows reads what several FORKed commands wrote */
};

typedef struct ColvoCmd {
    int32_t op;                  /* COLVO_CMD_* */
    int32_t stream;              /* 0 = main, 1 = side */
    ColvoConvDesc desc;          /* conv commands */
    const void* p[12];           /* pointer arguments in the order listed above (device memory, caller-owned) */
    int32_t i[12];               /* integer arguments in the order listed above */
    float f[4];
} ColvoCmd;

/* Enqueue cmds[0..n) in order; side_stream may be NULL when no command uses it.  FORK/JOIN use a small ring of
 * library-owned events (host objects; still no device allocation).  Stops at the first failing command. */
int colvo_run_commands(const ColvoCmd* cmds, int n, colvo_stream_t main_stream, colvo_stream_t side_stream);
/* colvo_run_commands spreads consecutive FORKs over the caller's side stream and up to n library-owned ones (default 1,
 * COLVO_SIDE_STREAMS - 1).  A process that drives MORE streams of its own beside the main and the side stream -- RCCL's
 * communicator stream in data-parallel training -- must set n = 0: with four or more hardware queues active and cross-queue
 * dependencies between them the step measured 5.2 ms instead of 1.7. */
int colvo_set_aux_side_streams(int n);
/* hipGraph form.  While main_stream is being CAPTURED (hipStreamBeginCapture) colvo_run_commands turns the list into graph nodes
 * with explicit dependencies on that one stream (hipStreamUpdateCaptureDependencies): the main-stream commands as one chain, the
 * side-stream commands as a second chain -- side_stream itself is not used and no event is recorded.  policy: 0 = one branch (side
 * commands captured in list order on the main chain), 1 = one cross-branch edge per FORK (the eager schedule node for node),
 * 2 (default) = side commands flushed `group` at a time (default 2), each segment depending on the main chain as captured at the
 * flush and on the previous segment, 3 = as 2 with the segments alternating between two side chains.  Every call ends joined.
 * Nodes are created so that the main chain stays the FIRST child of every fork point: ROCm's executor keeps the first child on the
 * parent's stream and opens a stream per further child. */
int colvo_set_capture_policy(int policy, int group);
/* Carry mode of the hipGraph form (default off: every captured colvo_run_commands call ends joined).  on: a list that ends without
 * a JOIN leaves the side chain OPEN -- the next captured call continues it, exactly as the eager schedule's deferred join leaves the
 * weight gradients of one network running beside the next network's backward pass -- and side commands still pending at the end of
 * a call are held back (as copies) until the next call has captured its first main-chain node: created earlier they would become the
 * first child of the main chain's last node and push the main chain onto a new stream (see colvo_set_capture_policy).  The caller
 * must then call colvo_capture_join(stream) wherever the eager schedule joins its side stream (before the optimizer, before a
 * collective that reads the gradients, before the capture ends). */
int colvo_set_capture_carry(int on);
/* The next node captured on `stream` depends on the main chain AND on everything the open side chain(s) hold, pending commands
 * included.  No-op when `stream` is not being captured or nothing is open. */
int colvo_capture_join(colvo_stream_t stream);
/* Structure of what has been built (tests, tools): out[0..7] describe the graph under construction on `stream` when it is being
 * captured -- nodes, edges, root nodes, leaf nodes, nodes with >= 2 children (forks), nodes with >= 2 parents (joins), largest
 * out-degree, largest in-degree -- and are -1 otherwise; out[8..15] count what colvo_run_commands has captured since
 * colvo_graph_stats_reset(): calls, main-chain commands, side-chain commands, side segments, joins, calls that started with
 * carried-over commands, the largest dependency set a call started from, commands pending right now.  n >= COLVO_GRAPH_STATS_N. */
#define COLVO_GRAPH_STATS_N 16
int colvo_graph_stats(colvo_stream_t stream, long long* out, int n);
int colvo_graph_stats_reset(void);
/* Return the library to its pre-capture state: what the last capture left in its bookkeeping (handles of graph nodes that die with
 * the graph, copies of held-back commands, the fork point), capture policy / group / carry mode back to their defaults, the calling
 * thread's launch tap disarmed.  Call it when a captured graph is destroyed or a capture failed (coivo_amd.graph.GraphedTrainStep.close);
 * fails with hipErrorStreamCaptureUnsupported-style code and changes nothing while `stream` (may be NULL) is being captured. */
int colvo_capture_reset(colvo_stream_t stream);

/* Dispatch thresholds (coivo_amd/csrc/tuning.h: ONE table, defaults measured on MI355X; production reads no environment
 * variable).  Developer / test hooks: set or read an entry by name ("quad_min_wgs", "wgrad_atomic_mb", ...); with COLVO_DEV=1 in
 * the environment at load time every entry can also be overridden by COLVO_<UPPER-CASE NAME>. */
/* Which kernel form the dispatchers chose, counted per process since the last reset (developer / test hook: the forms that are
 * selected by grid size -- k_conv_rt, the halved weight-gradient grids, the four-class / register-tiled weight gradients, the
 * clean-arena stores -- are invisible in a result; a test that means to cover one asserts that it ran).  colvo_form_counts fills
 * out[0..n) and returns the number of forms; colvo_form_name(id) names them (NULL beyond the last). */
int colvo_form_counts(long long* out, int n);
void colvo_form_counts_reset(void);
const char* colvo_form_name(int id);
int colvo_tune_set(const char* name, double value);
int colvo_tune_get(const char* name, double* value);

#ifdef __cplusplus
}
#endif
#endif /* COLVO_H_ */
