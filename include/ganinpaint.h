/*
 * libganinpaint — C-ABI of the MI355X (gfx950) backend for the GAN-inpainting hot path.
 *
 * The reference (abeytheo/gan-inpainting) is pure Python and has NO FFI of its own; the
 * interfaces this library stands behind are the Python ones cited next to each group below
 * (paths are relative to the upstream repository). INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every entry point returns 0 on success, a negative gi_status otherwise; the message of the
 *     last error on the calling thread is returned by gi_last_error(); nothing throws.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 *     buffer (it hands device memory to the handle with the *_bind calls, sizes come from the
 *     matching *_bytes / *_count queries); the handle owns no device memory.
 *   - calls are asynchronous on the hipStream_t given to gi_ctx_create (passed as void*);
 *     a handle is bound to one device and one stream and is not thread-safe.
 *   - public image tensors are fp32 (N,1,H,W) contiguous (NCHW == NHWC for one channel).
 *   - conv / transposed-conv weights are fp32 in physical order [a][ky][kx][b] where
 *     a = Conv2d out-channels (resp. ConvTranspose2d in-channels) and b the other channel
 *     axis, i.e. the torch "channels_last" image of the reference's [a,b,4,4] tensors
 *     (lib/models/networks.py:285, :293-309, :337-352), so state_dict() values are zero-copy.
 */
#ifndef GANINPAINT_H
#define GANINPAINT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GI_OK = 0,
  GI_ERR_INVALID = -1,      /* bad argument / unsupported shape */
  GI_ERR_HIP = -2,          /* a HIP runtime call or kernel launch failed */
  GI_ERR_STATE = -3,        /* call order violated (e.g. backward without forward) */
  GI_ERR_UNSUPPORTED = -4
} gi_status;

typedef enum { GI_F32 = 0, GI_F16 = 1 } gi_dtype;      /* internal compute/storage type */
typedef enum { GI_ACT_NONE = 0, GI_ACT_RELU = 1, GI_ACT_LRELU = 2 } gi_act;

typedef struct gi_ctx gi_ctx;
typedef struct gi_net gi_net;

const char* gi_last_error(void);
int gi_version(void);

/* ---- run-time options (process-wide): which kernel family serves a layer. Every shape is served under every setting
 * (the alternatives are the fallbacks for shapes the default kernels do not take, e.g. tensors beyond 2^31 bytes); results
 * agree to rounding (tests/test_options_gpu.py runs each). value < 0 restores the default, which is the environment
 * variable of the same name if set (read once), else the number in brackets. Nothing here exists in the reference.
 *   GI_IGEMM5 [7]         bit 0 / 1 / 2: halo-resident implicit GEMM for sub-pixel phases / 4x4-s2 gather / 3x3-s1 (else igemm3)
 *   GI_IGEMM6 [1]         0: first-generation halo kernels (igemm5) instead of the buffer-descriptor LDS-DMA ones (igemm6 AND igemm8)
 *   GI_IGEMM8 [1]         0: never the four-wave / two-workgroups-per-CU halo kernel (igemm8); 1: on layers with >= 512
 *                         workgroups; 2: on every eligible layer
 *   GI_IGEMM7 [1]         0: small-M layers on the generic kernel (igemm.hip) instead of the four-stage ring kernel
 *   GI_IGEMM_FIXUP [1]    0: split-K partial sums added by a finish launch instead of the last arriver inside the GEMM
 *   GI_IGEMM_VARIANT [3]  1: fp16 layers on the register-staged generic kernel only
 *   GI_BN_ACC [1]         0: BatchNorm statistics through partial rows + finalize launches (read at net creation)
 *   GI_FUSE_HEAD [1]      0: last norm + activation in front of a network's head as a separate pass (read at net creation)
 *   GI_HEAD_FAST [1]      0: PatchGAN head on the generic kernels
 *   GI_BN_BWD_FUSE [1]    0: BatchNorm-backward reduction as its own launch instead of the producing GEMM's epilogue
 *   GI_BN_BWD_SMALL [512] largest pixel count served by the one-launch BatchNorm backward (0: never)
 *   GI_WGRAD2 [1]         0: weight gradients on the register-staged kernel (wgrad.hip) only
 *   GI_WGRAD3 [1]         0: the unpipelined two-tap-row weight-gradient kernel instead of wgrad3
 *   GI_BN_FOLD [0]        1: the BatchNorm + activation (+ dropout) pass of the generator's small layers inside the GEMM that produces
 *                         the layer (igemm7's last finisher per channel column) instead of as its own launch; bit-identical results.
 *                         Off by default: measured SLOWER (+8 .. 10 us per folded layer at the headline shape)
 *   GI_C1_FUSED [1]       0: the single-channel transposed convolution (generator u1, d1's input gradient) as two launches through a
 *                         col tensor in memory instead of one launch with the col rows in LDS; bit-identical results
 *   GI_WGRAD_STREAM [1]   0: the weight-gradient GEMMs of gi_net_backward[_phase] on the context's stream, in line with the
 *                         input-gradient chain, instead of on the network's second HIP stream beside it (the entry joins the two
 *                         before it returns; same kernels, same operands, same results)
 *   GI_MASK_BITS [1]      0: the LeakyReLU backward of the first layer, fused into the second layer's input-gradient GEMM, reads the
 *                         layer's fp16 activation (128 bytes per pixel) instead of the 64-bit sign word per pixel the forward wrote
 *                         beside it, and applies the slope to the fp16-rounded gradient instead of the fp32 accumulator
 *   GI_C1W_FUSE [1]       0: the weight gradient of the first (single-channel) layer as its own launch reading the gradient at the
 *                         layer's output from memory, instead of from the tiles of the second layer's input-gradient GEMM while they
 *                         are in LDS (that gradient is then not stored at all unless the network's input gradient is asked for).
 *                         Needs GI_MASK_BITS = 1
 *   GI_IGEMM7_WAVES [8]   4: the small-M ring kernel (igemm7) as four-wave workgroups (one wave per SIMD, 64 x BN/2 wave tiles)
 *                         instead of eight-wave ones (two per SIMD); sums agree to fp32 rounding */
int gi_set_option(const char* name, int value);
int gi_get_option(const char* name, int* value);
/* name of the GEMM / weight-gradient kernel family and instantiation launched most recently by this process, e.g.
 * "igemm6<1,128,relu>", "igemm7<0,64>", "wgrad3<4>" (tests assert which kernel served a shape); "" before the first. */
const char* gi_debug_last_kernel(void);
/* number of GEMM launches of this process that normalised their own output (GI_BN_FOLD; tests count folded layers with it) */
int gi_debug_fold_count(void);

/* ---- context ------------------------------------------------------------------------------ */
int gi_ctx_create(int device_id, void* hip_stream, gi_ctx** out);
int gi_ctx_destroy(gi_ctx* ctx);
int gi_ctx_sync(gi_ctx* ctx);

/* ---- networks: lib/models/networks.py:13-28 get_network, :216-253 UnetGenerator,
 *      :331-363 PatchGANDiscriminator ------------------------------------------------------- */
/* UnetGenerator(in_c=1,out_c=1,num_downs,ngf,BatchNorm2d,use_dropout truthy): dropout_p is 0.5
 * for the reference's get_network (networks.py:18-19), 0 disables it. n_slots = number of
 * independent activation sets (forward calls that may be live before their backward). */
int gi_unet_create(gi_ctx* ctx, int num_downs, int ngf, float dropout_p, int H, int W, int max_n,
                   int dtype, int n_slots, gi_net** out);
/* UnetGenerator(1, out_c, ...) with out_c in 1..64 output channels (the frozen face-parsing network
 * UnetGenerator(1,4,7,ngf=32), train.py:171-172; Tanh head as in networks.py:293-298). out_c > 1 nets
 * are forward + input-gradient only (gi_net_backward with need_param_grads = 0, also after an eval-mode
 * forward: running-statistics BatchNorm). y / dy are (n,out_c,H,W). ngf must be a multiple of 64 for a
 * bound handle; an inventory-only handle (ctx NULL) accepts multiples of 8 (Python embeds ngf=32 in 64). */
int gi_unet_create_ex(gi_ctx* ctx, int num_downs, int ngf, int out_c, float dropout_p, int H, int W,
                      int max_n, int dtype, int n_slots, gi_net** out);
/* The same with the generator's norm layer chosen as get_norm_layer does (lib/models/networks.py:29-45):
 * norm_kind 0 = BatchNorm2d(affine, running statistics) - gi_unet_create_ex -, 1 = InstanceNorm2d(affine=False,
 * track_running_stats=False): statistics per image and channel in train AND eval mode, and every convolution carries a
 * bias (use_bias, networks.py:270-273; inventory names "<conv>.bias"), 2 = none (Identity). Kinds 1 and 2 run on the
 * unfused building blocks (no inference folding, no fused head). */
int gi_unet_create_norm(gi_ctx* ctx, int num_downs, int ngf, int out_c, int norm_kind, float dropout_p, int H,
                        int W, int max_n, int dtype, int n_slots, gi_net** out);
/* The same with level 1 (the outermost block's ngf channels) computed ch1 channels wide, ch1 a multiple of 64 >= ngf
 * (0: ngf): the inventory then reports [ch1, 1, 4, 4], [2 ngf, ch1, 4, 4], ... for the tensors that touch level 1; a
 * caller holding a narrower network (UnetGenerator(1, 4, 7, ngf=32), train.py:171-172) zero-fills the extra rows /
 * columns - zero channels stay zero through BatchNorm (beta 0) and the activations -, every other level runs at its
 * true width. */
int gi_unet_create_padded(gi_ctx* ctx, int num_downs, int ngf, int ch1, int out_c, int norm_kind, float dropout_p,
                          int H, int W, int max_n, int dtype, int n_slots, gi_net** out);
/* PatchGANDiscriminator(c=1, sigmoid): Linear(25,1) generalised to ((H/16-3)*(W/16-3),1). */
int gi_patchgan_create(gi_ctx* ctx, int H, int W, int sigmoid, int max_n, int dtype, int n_slots,
                       gi_net** out);
int gi_net_destroy(gi_net* net);

/* parameter / buffer inventory, in the reference's named_parameters() order, then buffers.
 * kind: 0 conv weight (shape [a,b,4,4], physical [a][ky][kx][b]), 1 bias / BN affine / Linear
 * (contiguous), 2 BN running_mean, 3 BN running_var. offset is in floats into the flat param
 * (kind 0,1) or flat buffer (kind 2,3) allocation. */
int gi_net_tensor_count(gi_net* net);
int gi_net_tensor_desc(gi_net* net, int index, char* name, int name_cap, int* kind, int64_t* shape4,
                       int* ndim, int64_t* offset, int64_t* numel);
int64_t gi_net_param_floats(gi_net* net);      /* flat fp32 params (and grads, optimizer states) */
int64_t gi_net_buffer_floats(gi_net* net);     /* flat fp32 BN running stats */
int64_t gi_net_workspace_bytes(gi_net* net);   /* activations, packed weights, scratch */
int gi_net_bind(gi_net* net, float* params, float* grads, float* buffers, void* workspace,
                int64_t workspace_bytes);
/* re-derive the packed (fp16 / transposed) weight copies after params were written by the caller */
int gi_net_sync_weights(gi_net* net);
int gi_net_set_train(gi_net* net, int train);
/* Inference hint (generators): with train = 0, forwards will never be differentiated. BatchNorm layers are then folded
 * into the neighbouring convolutions (scale into a second weight copy, shift as the GEMM bias, activation in the
 * epilogue); gi_net_backward on such a forward fails. What eval.py:94 / evaluate.py:127-158 run under no_grad. */
int gi_net_set_inference(gi_net* net, int inference);
int gi_net_set_loss_scale(gi_net* net, float scale);
/* discriminator only: the next forward/backward batches hold `groups` (1 or 2) consecutive, equally sized image
 * groups with INDEPENDENT BatchNorm batch statistics (running statistics updated group by group). groups = 2 runs
 * the reference's two critic calls D(ground), D(inpainted) (wgan_l1.py:134-135) as one stacked batch. */
int gi_net_set_bn_groups(gi_net* net, int groups);   /* fp16 backward scaling, default 65536 */
int gi_net_set_dropout_seed(gi_net* net, uint64_t seed);
/* keep-mask of dropout level `level` used by the last train-mode forward of `slot`,
 * as uint8 in (N,C,H,W) order (what the oracle consumes); count = N*C*H*W */
int gi_net_dropout_mask(gi_net* net, int slot, int level, uint8_t* out_nchw, int64_t count);
/* impose external keep-masks (uint8 NCHW, device) for the next forward of slot; null clears */
int gi_net_set_dropout_mask(gi_net* net, int slot, int level, const uint8_t* mask_nchw);

/* Activations the forward held in `slot` saved for its backward, as fp32 (N,C,h,w): which side of every ReLU / LeakyReLU
 * kink the forward took (parity tests hand these decisions to the oracle for units whose pre-activation is within fp32
 * rounding distance of zero). Generator, level k = 1..num_downs: kind 0 = the skip a_k = LeakyReLU(norm(conv_k(.)))
 * (networks.py:287; k = num_downs: ReLU of the innermost convolution, :299-305); kind 1 (k < num_downs) = the decoder half
 * ReLU(dropout(norm(up_{k+1}(.)))) concatenated beside it (:289, :324). Discriminator, kind 0, level i = 1..4:
 * LeakyReLU(BatchNorm(conv_i(.))) (:335-345). count = N*C*h*w. */
int gi_net_saved_activation(gi_net* net, int slot, int kind, int level, float* out_nchw, int64_t count);
/* Debug: number of split-K tile tickets of this network that are not zero between launches (must be 0: every launch leaves
 * them zero; anything else means a launch was cut short). Synchronises the stream. */
int gi_net_debug_nonzero_tickets(gi_net* net, int* count);

/* forward: x (n,1,H,W) fp32 -> y: generator (n,1,H,W) fp32, discriminator (n,1) fp32 */
int gi_net_forward(gi_net* net, int slot, const float* x, float* y, int n);
/* backward of the forward held in `slot`: dy like y; dx like x or NULL; need_wgrad=0 is the
 * frozen-net case of util.set_requires_grad(nets, False) (lib/models/util.py:19-22): only input
 * gradients are produced. Parameter gradients ACCUMULATE into the bound grads. */
int gi_net_backward(gi_net* net, int slot, const float* dy, float* dx, int need_wgrad);

/* backward in pieces so the host can overlap a gradient all-reduce with compute (phase 0 = everything).
 * Generator: phase 1 = decoder (its parameter gradients are the flat range [gi_net_phase_split(), end), complete
 * on return); phase 2 = encoder (range [0, split)), or split once more: phase 3 = innermost .. level 5 (range
 * [gi_net_phase_split2(), split): 12.6 of the encoder's 15.3 M gradients, finished first), phase 4 = levels 4 .. 1.
 * Discriminator: phase 1 = head + conv4 block (range [gi_net_phase_split(), end)), phase 2 = conv3 .. conv1. */
int gi_net_backward_phase(gi_net* net, int slot, const float* dy, float* dx, int need_wgrad, int phase);
int64_t gi_net_phase_split(gi_net* net);
int64_t gi_net_phase_split2(gi_net* net);

/* ---- data-parallel gradient exchange (NOT in the reference, which is single-device: train.py:44-46).
 * One process per GPU, replicated networks, per-rank minibatch; before every optimizer update the flat fp32 gradient
 * buffer of the network is SUM-all-reduced over RCCL (xGMI) and the optimizer divides by the world size (grad_scale).
 * The communicator is created from a 128-byte unique id (rank 0 draws it with gi_comm_unique_id, the host carries it to
 * the other ranks - e.g. torch.distributed.broadcast, MPI, a file). All calls are asynchronous on the given streams;
 * gi_net_allreduce_grads_async takes a range [begin, end) of the flat buffer (end < 0: to the end) so that ranges whose
 * gradients are final (gi_net_backward_phase / gi_net_phase_split) can be reduced while the backward still runs;
 * gi_allreduce_wait makes compute_stream wait for everything issued on comm_stream so far. librccl is loaded on
 * first use: GI_ERR_UNSUPPORTED when it is not installed. */
typedef struct gi_comm gi_comm;
int gi_comm_unique_id(char* id128_host);
int gi_comm_create(const char* id128_host, int rank, int world, int device_id, gi_comm** out);
int gi_comm_destroy(gi_comm* comm);
int gi_allreduce_sum_f32(gi_comm* comm, float* buf, int64_t count, void* hip_comm_stream);
int gi_net_allreduce_grads_async(gi_net* net, gi_comm* comm, int64_t begin, int64_t end, void* hip_comm_stream);
int gi_allreduce_wait(gi_comm* comm, void* hip_comm_stream, void* hip_compute_stream);

/* WGAN-GP EXTENSION (not in the reference, which clips weights: wgan_l1.py:151-153): accumulates the
 * parameter gradient of  lam * mean_n (||grad_x D(xhat)_n||_2 - 1)^2  into the bound grads and writes the
 * penalty to penalty_out[0] (device). xhat: (n,1,H,W) interpolates, e.g. from gi_interpolate.
 * fp32 critics without sigmoid, train mode. Uses a private activation set (no user slot is touched). */
int gi_patchgan_gradient_penalty(gi_net* net, const float* xhat, int n, float lam, float* penalty_out);
/* out[n] = eps[n]*real[n] + (1-eps[n])*fake[n] */
int gi_interpolate(gi_ctx* ctx, const float* real, const float* fake, const float* eps, int n, int64_t hw, float* out);

/* ---- mask pipeline: experiment_list/minimaxgan_l1.py:113-122 ------------------------------ */
/* mask_c = ceil(mask) if (flags & 1) else mask ; if (flags & 2) mask_c = 1 - mask_c (is_flip_mask of
 * evaluate.py:130-132) ; masked = ground * (1 - mask_c) */
int gi_mask_apply(gi_ctx* ctx, const float* ground, const float* mask, float* mask_c, float* masked,
                  int64_t count, int flags);
/* inpainted = masked + gen * mask_c */
int gi_mask_composite(gi_ctx* ctx, const float* masked, const float* gen, const float* mask_c,
                      float* inpainted, int64_t count);
/* out = a * b (the composite backward d_gen = d_inpainted * mask_c, and mask*x products) */
int gi_mul(gi_ctx* ctx, const float* a, const float* b, float* out, int64_t count);
/* out = a + alpha * b (sums the adversarial and reconstruction gradients of g_loss, minimaxgan_l1.py:168) */
int gi_add(gi_ctx* ctx, const float* a, const float* b, float* out, int64_t count, float alpha);

/* ---- losses. Each writes the scalar loss to loss_out[0] (device) and, when grad_a != NULL, the
 *      gradient w.r.t. `a` multiplied by gscale. nn.L1Loss (minimaxgan_l1.py:62,166);
 *      RMSELoss sqrt(mse+1e-16) (lib/models/loss.py:11-19); LocalLoss (loss.py:24-47,
 *      kind 0 = L1, 1 = MSE, 2 = sqrt extension); scratch: >= 4096 floats ---------------------- */
int gi_loss_l1(gi_ctx* ctx, const float* a, const float* b, int64_t count, float* loss_out,
               float* grad_a, float gscale, float* scratch);
int gi_loss_rmse(gi_ctx* ctx, const float* a, const float* b, int64_t count, float eps, float* loss_out,
                 float* grad_a, float gscale, float* scratch);
int gi_loss_mse(gi_ctx* ctx, const float* a, const float* b, int64_t count, float* loss_out,
                float* grad_a, float gscale, float* scratch);
int gi_loss_local(gi_ctx* ctx, const float* yhat, const float* y, const float* mask, int64_t count,
                  int kind, float* loss_out, float* grad_yhat, float gscale, float* scratch);
/* adversarial scalar losses on (n,) predictions: kind 0 BCE vs constant target
 * (minimaxgan_l1.py:135,141,162), 1 MSE vs constant target (experiment1_global_local_D.py:162),
 * 2 mean (wgan_l1.py:137-143,177; gscale carries the +1/-1 of backward(one/mone)) */
int gi_loss_adv(gi_ctx* ctx, const float* pred, int n, int kind, float target, float* loss_out,
                float* grad_pred, float gscale);
/* the two calls of a batch on a stacked [a | b] prediction vector (2 * n_each values: D(ground) | D(inpainted),
 * wgan_l1.py:134-143) in ONE launch: the same arithmetic per half, each with its own target, loss slot and gradient sign */
int gi_loss_adv_pair(gi_ctx* ctx, const float* pred2, int n_each, int kind, float target_a, float target_b, float* loss_a,
                     float* loss_b, float* grad_pred2, float gscale_a, float gscale_b);

/* ---- optimizers over flat fp32 buffers: optim.Adam(lr=2e-4,betas=(.5,.999))
 *      (minimaxgan_l1.py:64-65), optim.RMSprop(lr=5e-5) + p.data.clamp_(-c,c)
 *      (wgan_l1.py:64-65,151-153; clamp<=0 disables) ---------------------------------------- */
int gi_adam_step(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr,
                 float beta1, float beta2, float eps, int step, float grad_scale);
int gi_rmsprop_step(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr,
                    float alpha, float eps, float clamp, float grad_scale);
int gi_clamp(gi_ctx* ctx, float* p, int64_t count, float lo, float hi);
/* fp16 overflow guard (backend addition; the reference trains in fp32): gi_check_finite scans a flat gradient
 * buffer; flag3 is 3 device ints {running count of bad updates, verdict of this scan (0/1), scratch}. The
 * *_guarded optimizer steps do nothing when guard[1] != 0 (pass the same flag3, or NULL for the plain step). */
int gi_check_finite(gi_ctx* ctx, const float* g, int64_t count, int* flag3);
/* the same in two steps, for ONE optimizer over several flat buffers (Adam over itertools.chain(D_local, D_global),
 * experiment1_global_local_D.py:123): scan every buffer into the same flag3, then finish once - all networks of the
 * update share one verdict (all skip or none) and the running count advances by one per skipped update. */
int gi_check_finite_scan(gi_ctx* ctx, const float* g, int64_t count, int* flag3);
int gi_check_finite_finish(gi_ctx* ctx, int* flag3);
int gi_adam_step_guarded(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr,
                         float beta1, float beta2, float eps, int step, float grad_scale, const int* guard);
/* skipped_seen >= 0: `step` counts every call since the start minus the skipped updates the host has already learned of
 * (skipped_seen of them, read back from guard[0] at its own cadence); the kernel subtracts the rest, guard[0] - skipped_seen,
 * before it forms the bias corrections, so a skipped update never advances them whenever the host polls. < 0: `step` as is. */
int gi_adam_step_guarded2(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr,
                          float beta1, float beta2, float eps, int step, int skipped_seen, float grad_scale, const int* guard);
int gi_rmsprop_step_guarded(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr,
                            float alpha, float eps, float clamp, float grad_scale, const int* guard);
/* Without the finish launch: flag4 is 4 device ints {running count, verdict of the latest update, scan word, scan word}. An update
 * scans its buffers into flag4[word] (word = 2 or 3, the other one than the previous update used) and passes the same word to the
 * *_scan optimizer steps, which take their verdict from it; the FIRST step of the update (finish = 1) also records it (flag4[1],
 * flag4[0]) and clears the other word for the next update. word = 0: the *_guarded behaviour (verdict from flag4[1]). */
int gi_check_finite_scan_word(gi_ctx* ctx, const float* g, int64_t count, int* flag4, int word);
int gi_adam_step_scan(gi_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                      float eps, int step, int skipped_seen, float grad_scale, int* guard, int word, int finish);
int gi_rmsprop_step_scan(gi_ctx* ctx, float* p, const float* g, float* sq, int64_t count, float lr, float alpha, float eps,
                         float clamp, float grad_scale, int* guard, int word, int finish);
/* mean(|g|) of `nseg` segments [off[i], off[i]+len[i]) of g -> out[i]
 * (gradient-flow statistics, minimaxgan_l1.py:180-182); offsets/lengths are device int64 */
int gi_grad_absmean(gi_ctx* ctx, const float* g, const int64_t* seg_off, const int64_t* seg_len,
                    int nseg, float* out);

/* ---- SSIM metric. lib/pytorch_ssim/__init__.py:20-40 `_ssim`, :68-76 `ssim()`,
 *      called per batch at experiment1_global_local_D.py:209. img1/img2: (n,c,H,W) fp32 contiguous on
 *      the device; depthwise window_size^2 Gaussian (sigma 1.5, zero padding window_size/2; window_size
 *      odd, <= 31), C1 = 0.01^2, C2 = 0.03^2. per_sample (n floats, may be NULL) = size_average=False;
 *      mean_out (1 float, may be NULL) = size_average=True. window_host: the 1-D window as
 *      `window_size` HOST floats, or NULL for the reference's gaussian(window_size, 1.5). scratch:
 *      gi_ssim_scratch_floats(...) device floats, 8-byte aligned (returns -1 for unsupported sizes) -- */
int64_t gi_ssim_scratch_floats(int n, int c, int H, int W, int window_size);
int gi_ssim(gi_ctx* ctx, const float* img1, const float* img2, int n, int c, int H, int W, int window_size,
            const float* window_host, float* per_sample, float* mean_out, float* scratch);

/* ---- evaluation pass ------------------------------------------------------
 * lib/models/evaluate.py:127-158: per batch  out = gen*m + ground*(1-m)  (m = mask_c from
 * gi_mask_apply: ceil, optionally flipped) and
 *   rmse_global = sqrt(mean (ground-out)^2 + eps)      l1_global = mean |ground-out|
 *   rmse_local  = sqrt(sum (out*m-ground*m)^2 / count(m!=0) + eps)   l1_local = sum |out*m-ground*m| / count(m!=0)
 * acc (device, 5 floats, may be NULL): acc[0..3] += {rmse_global, l1_global, rmse_local, l1_local},
 * acc[4] += 1 (the reference divides its running sums by the number of batches, :160-165);
 * batch_out (4 floats, may be NULL) receives this batch's values; inpainted_out (count floats, may be
 * NULL) the composite. scratch: gi_eval_recon_scratch_floats(count) floats, 8-byte aligned. */
int64_t gi_eval_recon_scratch_floats(int64_t count);
int gi_eval_recon(gi_ctx* ctx, const float* ground, const float* gen, const float* mask_c, int64_t count, float eps,
                  float* inpainted_out, float* acc, float* batch_out, float* scratch);
/* lib/models/evaluate.py:179-224 calculate_segmentation_eval_metric: prediction = argmax over the
 * class axis of logits (n,num_classes,hw) fp32; for each u of unique_labels_host[nu] (HOST ints, nu<=16):
 * precision / recall / IoU per sample from the pixel counts (eps 1e-32), mean over the n samples.
 * per_class: nu*3 floats {precision, recall, iou}; across: 3 floats (mean over classes).
 * labels: (n,hw) int64. scratch_counts: n*16*3 ints. */
int gi_seg_metrics(gi_ctx* ctx, const int64_t* labels, const float* logits, int n, int num_classes, int64_t hw,
                   const int* unique_labels_host, int nu, float* per_class, float* across, int* scratch_counts);

/* ---- config-5 generator-loss extras ------------------------------------
 * tv_loss (lib/models/loss.py:138-151): tv_weight * (mean_h-diff^2 + mean_w-diff^2) over `planes` = n*c
 * planes of H x W fp32; grad (may be NULL) = d loss / d img * gscale. loss_out: 1 float.
 * scratch: >= 4096 floats, 8-byte aligned. */
int gi_loss_tv(gi_ctx* ctx, const float* img, int planes, int H, int W, float tv_weight, float* loss_out,
               float* grad, float gscale, float* scratch);
/* nn.CrossEntropyLoss(weight=w) on (n,num_classes,hw) fp32 logits and (n,hw) int64 labels
 * (wgan_perceptual_style_faceparsing.py:67-68,212-213): sum_p w[y_p] * nll_p / sum_p w[y_p]; labels equal
 * to ignore_index (torch default -100) carry no weight. class_weight_host: num_classes HOST floats or NULL
 * (ones). num_classes in {2,3,4,8,16}. loss_out: 2 floats {loss, sum of weights}; grad_logits may be NULL. */
int gi_loss_cross_entropy(gi_ctx* ctx, const float* logits, const int64_t* labels, int n, int num_classes,
                          int64_t hw, const float* class_weight_host, int ignore_index, float* loss_out,
                          float* grad_logits, float gscale, float* scratch);

/* VGG-19 features for perceptual_loss / style_loss / perceptual_and_style_loss (lib/models/loss.py:50-115)
 * and gram_matrix (:117-136): taps relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (features[1,6,11,20,29]) of
 * the grey image repeated over 3 channels (:54-55). Forward only (the reference runs it under no_grad).
 * params: fp32, torchvision layout, the 13 convolutions up to features.28 (gi_vgg19_tensor_desc gives
 * name "features.<i>.weight|bias", shape, offset); fp16 MFMA compute. max_pairs = largest n. */
typedef struct gi_vgg gi_vgg;
int gi_vgg19_create(gi_ctx* ctx, int H, int W, int max_pairs, gi_vgg** out);
void gi_vgg19_destroy(gi_vgg* v);
int64_t gi_vgg19_param_floats(const gi_vgg* v);
int64_t gi_vgg19_workspace_bytes(const gi_vgg* v);
int gi_vgg19_num_tensors(const gi_vgg* v);
int gi_vgg19_tensor_desc(const gi_vgg* v, int index, char* name, int name_cap, int* shape4, int64_t* offset);
int gi_vgg19_bind(gi_vgg* v, const float* params, void* ws, int64_t ws_bytes);   /* ws 256-byte aligned */
int gi_vgg19_sync_weights(gi_vgg* v);                                            /* after params change */
/* out2 = { weight_p * sum_taps mean((F_o-F_t)^2), weight_s * sum_taps mean((G_o-G_t)^2) }, G = F F^T/(HWC);
 * output/target: (n,1,H,W) fp32; per_tap10 (may be NULL): the 5 perceptual then the 5 style terms. */
int gi_vgg19_perceptual_style(gi_vgg* v, const float* output, const float* target, int n, float weight_p,
                              float weight_s, float* out2, float* per_tap10);
/* feature map of tap 0..4 as (n,C,h,w) fp32 (parity checks) */
int gi_vgg19_features(gi_vgg* v, const float* x, int n, int tap, float* out_nchw);

/* ---- input transform: transforms.Resize(size) + ToTensor() of train.py:69-72 on decoded
 *      8-bit grey images (lib/data/dataset.py:6-12). Bit-exact restatement of Pillow's antialiased bilinear
 *      resampler (22-bit fixed point, horizontal pass first, uint8 intermediate), then /255. ------------- */
int gi_resize_output_size(int in_h, int in_w, int size, int* out_h, int* out_w);   /* Resize(int): smaller edge */
int64_t gi_resize_table_bytes(int in_h, int in_w, int out_h, int out_w);
/* host-computed coefficient tables -> tables_dev (device, gi_resize_table_bytes); once per geometry; synchronises */
int gi_resize_build_tables(gi_ctx* ctx, int in_h, int in_w, int out_h, int out_w, void* tables_dev);
/* src: n x in_h x in_w uint8 (device). dst_f32 (n,out_h,out_w) in [0,1] and/or dst_u8 (either may be NULL).
 * tmp: n*in_h*out_w bytes (device). */
int gi_resize_to_tensor(gi_ctx* ctx, const void* tables_dev, const uint8_t* src, int n, int in_h, int in_w,
                        int out_h, int out_w, float* dst_f32, uint8_t* dst_u8, uint8_t* tmp);

/* ---- single-layer entry points (unit parity tests and kernel roofline measurements) -------- */
/* out[n,y,x,a] = act( sum_{ky,kx,b} in[n,2y-1+ky,2x-1+kx,b] * w[a][ky][kx][b] ), NHWC, dtype T.
 * in: (n,H,W,cb) ld=ldin; out: (n,H/2,W/2,ca) ld=ldout. w_packed is T [ca][16*cb].
 * partials (may be NULL) receives per-tile column sum / sum of squares; ws: split-K scratch */
int gi_conv_s2_forward(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out,
                       int n, int H, int W, int cb, int ldin, int ca, int ldout, int relu_in,
                       int act_out, float* ws, int64_t ws_bytes);
/* out[n,2y-1+ky,2x-1+kx,b] += in[n,y,x,a]*w[a][ky][kx][b] (transposed conv k4 s2 p1) as four
 * sub-pixel 2x2 convolutions. in: (n,H,W,ca); out: (n,2H,2W,cb); w_phase is T [4][cb][4*ca]
 * produced by gi_pack_weights */
int gi_convT_s2_forward(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out,
                        int n, int H, int W, int ca, int ldin, int cb, int ldout, int relu_in,
                        int act_out, float* ws, int64_t ws_bytes);
/* The same two layers with the fused epilogues the networks use (all optional, zero = absent; nothing here exists in the
 * reference: these are the seams at which its separate BatchNorm / activation modules, networks.py:285-318, are folded into the
 * GEMMs). Fields marked (returned) say whether the kernel that served the shape implemented the fusion; if not, the caller
 * runs the separate pass. */
typedef struct gi_igemm_ex {
  int relu_cend;                 /* with relu_in: only input channels [0, relu_cend) need the ReLU (0 = all). A hint for inputs whose
                                  * channels from relu_cend on are already non-negative (the decoder half of a concat buffer): kernels
                                  * may apply the ReLU to more channels. Must be a multiple of 64 and <= the input channels. */
  /* activation backward on the result: out = (out + [mask > 0] * add) * (mask > 0 ? 1 : mask_slope); mask / add laid out
   * like out with leading dimensions ldmask / ldadd (the LeakyReLU backward of a norm-free layer, networks.py:287) */
  const void* mask; int ldmask; float mask_slope;
  const void* add; int ldadd;
  /* with mask, optional: one 64-bit sign word per output pixel, bit c = [mask[p][c] > 0] for its first 64 channels. The kernel
   * that takes them (sub-pixel phases with 64 output channels) reads 8 bytes per pixel instead of 128 and applies the slope to
   * its fp32 accumulators (one rounding: within fp16 rounding of the result without them) */
  const unsigned long long* mask_bits;
  /* BatchNorm batch statistics of the result: every tile adds its column sum / sum of squares to the exact accumulator
   * block stat_acc (gi_stat_acc_words(ca) zeroed 64-bit words; tile t -> replica t mod stat_reps, a power of two <= 4;
   * stat_pg > 0: GEMM rows >= stat_pg belong to a second population). partials: per-tile rows instead, [tiles][2][c] */
  unsigned long long* stat_acc; int stat_reps; int stat_pg;
  float* partials;
  /* BatchNorm-backward reduction of the layer whose output gradient this GEMM produces: with x = bwd_x (raw convolution
   * output, ld bwd_ldx), dz = g * (fma(x, scale, shift) > 0 ? 1 : bwd_slope), xhat = (x - mean) * inv, every tile adds
   * sum dz and sum dz * xhat to bwd_acc (layout as stat_acc; populations of bwd_pg output pixels, vectors bwd_stride apart) */
  const void* bwd_x; int bwd_ldx;
  const float* bwd_scale; const float* bwd_shift; const float* bwd_mean; const float* bwd_inv; int bwd_stride;
  float bwd_slope; unsigned long long* bwd_acc; int bwd_reps; int64_t bwd_pg;
  /* bwd_c > 0: only the output columns [bwd_c0, bwd_c0 + bwd_c) carry that gradient (multiples of 128; the decoder half of a U-Net
   * concat gradient): bwd_x, the vectors and bwd_acc hold bwd_c channels, indexed by column - bwd_c0. Taken by the kernels that
   * serve layers of >= 512 workgroups; otherwise bwd_applied stays 0 */
  int bwd_c0, bwd_c;
  int mask_applied, bwd_applied, stat_used, ntiles_out;   /* (returned) */
} gi_igemm_ex;
int gi_conv_s2_forward_ex(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out,
                          int n, int H, int W, int cb, int ldin, int ca, int ldout, int relu_in,
                          int act_out, float* ws, int64_t ws_bytes, gi_igemm_ex* ex);
int gi_convT_s2_forward_ex(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out,
                           int n, int H, int W, int ca, int ldin, int cb, int ldout, int relu_in,
                           int act_out, float* ws, int64_t ws_bytes, gi_igemm_ex* ex);
/* exact accumulator blocks (csrc/stat_acc.h): words per block of c channels; totals of the two quantities of a population
 * as doubles, out_dev[2][c] (device) */
int64_t gi_stat_acc_words(int c);
int gi_stat_acc_read(gi_ctx* ctx, const unsigned long long* acc, int c, int reps, int group, double* out_dev);
/* dW[a][ky][kx][b] += scale * sum_{n,y,x} S[n,y,x,a] * L[n,2y-1+ky,2x-1+kx,b] */
int gi_wgrad_s2(gi_ctx* ctx, int dtype, const void* S, const void* L, float* dW, int n, int Hs, int Ws,
                int ca, int ldS, int cb, int ldL, int relu_S, float scale);
/* the same with a scratch buffer of gi_wgrad_s2_scratch_bytes(...) bytes: the pixel-range splits write their partial
 * tiles with plain stores and a fixed-order pass adds them to dW - deterministic and faster than the fp32 atomics
 * gi_wgrad_s2 uses (scratch NULL or too small falls back to them) */
int64_t gi_wgrad_s2_scratch_bytes(int dtype, int n, int Hs, int Ws, int ca, int cb);
int gi_wgrad_s2_ws(gi_ctx* ctx, int dtype, const void* S, const void* L, float* dW, int n, int Hs, int Ws,
                   int ca, int ldS, int cb, int ldL, int relu_S, float scale, float* scratch, int64_t scratch_bytes);
/* fp32 master [a][16][b] -> T [a][16*b] (w_packed) and T [4][b][4*a] (w_phase); either may be NULL */
int gi_pack_weights(gi_ctx* ctx, int dtype, const float* w, int ca, int cb, void* w_packed, void* w_phase);
int gi_convert(gi_ctx* ctx, int dtype, const float* src, void* dst, int64_t count);      /* fp32 -> T */
int gi_convert_back(gi_ctx* ctx, int dtype, const void* src, float* dst, int64_t count); /* T -> fp32 */

/* time the dominant kernel `iters` times with hipEvents on the context stream; returns average
 * milliseconds per launch in *ms_out (used by bench.py for the roofline object) */
int gi_time_convT_s2(gi_ctx* ctx, int dtype, const void* in, const void* w_phase, void* out, int n, int H,
                     int W, int ca, int ldin, int cb, int ldout, int iters, float* ms_out_host);
int gi_time_conv_s2(gi_ctx* ctx, int dtype, const void* in, const void* w_packed, void* out, int n, int H, int W,
                    int cb, int ldin, int ca, int ldout, int iters, float* ms_out_host);   /* the same for gi_conv_s2_forward */

#ifdef __cplusplus
}
#endif
#endif /* GANINPAINT_H */
