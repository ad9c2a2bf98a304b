/* mmvae.h -- C ABI of the MI355X-native conv-VAE train path (libmmvae_hip.so).
 *
 * The reference (praateekmahajan/moving-mnist-vae) is pure Python and has no FFI of its own: the
 * arithmetic of its hot path is reached through torch.nn / torch.optim.  This header is therefore the
 * boundary a maintainer would bind (ctypes) in place of those calls.  Each entry point
 * cites the reference lines whose arithmetic it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host"; nothing is allocated
 *     or freed by the library except the opaque mmvae_net handle (host memory only);
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), never synchronise, and are
 *     hipGraph-capturable;
 *   - return value: 0 (or a positive count where documented) on success, negative MMVAE_ERR_* on failure,
 *     with a human-readable message from mmvae_last_error();
 *   - dtype: 0 = f32 storage / exact-f32 MFMA, 1 = bf16 storage / bf16 MFMA with f32 accumulation.
 *     Parameters, gradients, BN statistics, latents and the reconstruction are always f32.
 *   - activation layout is NHWC; the network boundary (image in, reconstruction out) is NCHW like the
 *     reference (identical for the 1-channel image).
 */
#ifndef MMVAE_H_
#define MMVAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MMVAE_API __attribute__((visibility("default")))
#else
#define MMVAE_API
#endif

#define MMVAE_OK 0
#define MMVAE_ERR_ARG (-1)
#define MMVAE_ERR_HIP (-2)
#define MMVAE_ERR_WORKSPACE (-3)
#define MMVAE_ERR_UNSUPPORTED (-4)

#define MMVAE_F32 0
#define MMVAE_BF16 1
#define MMVAE_FP8 2     /* network handle only (BASELINE configs[4]): bf16 storage, but the forward convolutions of the deep layers (channel
                           counts that are multiples of 64, every one followed by a BatchNorm) run on v_mfma_f32_16x16x32_fp8_fp8: OCP e4m3
                           weights (static power-of-two scale per layer, absorbed exactly by the BatchNorm) x e4m3 activations (quantised
                           behind the fused BN+ReLU), f32 accumulation and statistics; the backward pass stays bf16 (straight-through).
                           Where the 64x64 last up-block runs on the stream kernels its two branch outputs (the largest tensors of the
                           step) are STORED as e4m3 bytes with a static scale of 16 the BatchNorms behind absorb */

/* Version of this header's signatures.  2: mmvae_conv2d_wgrad gained the caller-owned `scratch` argument (before `stream`);
 * 3: mmvae_convT_bwd_fused gained `dw2`, the weight-gradient entry points REQUIRE the scratch (partial images, fixed-order reduce: no
 * float atomics), new entries (stage_labels, encoder_fwd_staged, decoder_bwd_gauss, upblock / conv1x1_bwd_fused, set_sync_bn_comm2,
 * pixelcnn_*); in_channels 1..4.  A binding built against another version must not call this one (check mmvae_abi_version() ==
 * MMVAE_ABI_VERSION at load time). */
#define MMVAE_ABI_VERSION 3
MMVAE_API int mmvae_abi_version(void);
MMVAE_API const char* mmvae_last_error(void);          /* host string, valid until the next failing call on this thread */

/* ------------------------------------------------------------------ network handle
 * Describes VAE(in_channels, ., decoder_out_channels, ., z_dimension, pixelcnn=False, only_pixelcnn=False, ...,
 * require_rsample, ., input_image_size)  -- model.py:258-310 (encoder :88-112, decoder :153-179).  */
typedef struct mmvae_net mmvae_net;
MMVAE_API int mmvae_net_create(mmvae_net** out, int in_channels, int z_dimension, int out_channels, int image_size,
                     int need_logvar, int dtype);
/* Deeper-than-reference variant (BASELINE configs[3]): blocks_per_stage residual blocks in every encoder stage (_make_layer with
 * blocks > 1, model.py:132-146: the extra BasicBlocks keep the shape and have an identity shortcut) and in every decoder stage
 * (_make_up_block with num_layer > 1, model.py:196-209: the extra blocks come first; the reference's own extra DeconvBottleneck
 * cannot run -- 2x main path against an identity shortcut, :205-206 -- so here it is conv1x1 -> BN -> ReLU -> conv3x3 -> BN, +
 * identity, ReLU).  blocks_per_stage = 1 is the reference network. */
MMVAE_API int mmvae_net_create_ex(mmvae_net** out, int in_channels, int z_dimension, int out_channels, int image_size,
                        int need_logvar, int dtype, int blocks_per_stage);
MMVAE_API void mmvae_net_destroy(mmvae_net* net);
/* Flat storage sizes: f32 parameters, f32 BN running statistics, int64 num_batches_tracked counters; first
 * decoder parameter (for gradient bucketing); decoder output side (32 or 64, model.py:169,191). */
MMVAE_API int mmvae_net_sizes(const mmvae_net* net, int64_t* n_params, int64_t* n_bn_f32, int32_t* n_bn_i64,
                    int64_t* decoder_param_offset, int32_t* decoder_side);
/* state_dict inventory in the reference's registration order (183 entries at the default config).
 * kind: 0 parameter (offset into the flat parameter buffer), 1 BN f32 buffer, 2 BN int64 counter. */
MMVAE_API int mmvae_net_num_entries(const mmvae_net* net);
MMVAE_API int mmvae_net_entry(const mmvae_net* net, int index, char* name, int name_cap, int32_t* ndim, int32_t shape[4],
                    int32_t* kind, int64_t* offset);
/* Bytes of activation workspace needed for a batch of N frames (forward + backward). */
MMVAE_API size_t mmvae_net_workspace_bytes(mmvae_net* net, int N);

/* VAE_Encoder.forward, model.py:114-130 (BasicBlock.forward :39-55).  x: [N,1,S,S] f32.  Writes mu, logvar
 * [N,z] f32 (logvar may be NULL when need_logvar=0).  training!=0: batch statistics, running-stat update with
 * momentum 0.1 / unbiased variance and num_batches_tracked += 1 (nn.BatchNorm2d); else running statistics. */
MMVAE_API int mmvae_encoder_fwd(mmvae_net* net, int N, const float* x, const float* params, float* bn_f32, int64_t* bn_i64,
                      void* workspace, size_t workspace_bytes, float* mu, float* logvar, int training, void* stream);
/* Input staging (reference main.py:383-387, image = (label - mean) / std): labels (int64 as the reference's loader yields them,
 * label_bytes = 8, or uint8, label_bytes = 1; N * in_channels * S * S of them) -> `image` f32 [N,in_channels,S,S] (the caller's tensor:
 * network input and Gaussian-loss target) AND the storage-type copy in the workspace's input slot, in ONE pass.
 * mmvae_encoder_fwd_staged is mmvae_encoder_fwd for exactly that `image` right after: it skips the conversion pass over x (the caller
 * guarantees x has not changed since it was staged into this workspace). */
MMVAE_API int mmvae_net_stage_labels(mmvae_net* net, int N, const void* labels, int label_bytes, float mean, float stdv, float* image,
                           void* workspace, size_t workspace_bytes, void* stream);
MMVAE_API int mmvae_encoder_fwd_staged(mmvae_net* net, int N, const float* x, const float* params, float* bn_f32, int64_t* bn_i64,
                             void* workspace, size_t workspace_bytes, float* mu, float* logvar, int training, void* stream);
/* autograd of the above (loss.backward(), main.py:398): accumulates (+=) parameter gradients into `grads`
 * (same flat layout as params; caller zeroes).  No gradient is produced for the image. */
MMVAE_API int mmvae_encoder_bwd(mmvae_net* net, int N, const float* d_mu, const float* d_logvar, const float* params, float* grads,
                      void* workspace, size_t workspace_bytes, void* stream);
/* VAE_Decoder.forward, model.py:181-194 (DeconvBottleneck.forward :70-85).  encoding: [N,z] f32.
 * recon: [N,out_channels,Sd,Sd] f32 NCHW, Sd = decoder_side (crop by `adjust`, model.py:328-329, is the caller's view). */
MMVAE_API int mmvae_decoder_fwd(mmvae_net* net, int N, const float* encoding, const float* params, float* bn_f32, int64_t* bn_i64,
                      void* workspace, size_t workspace_bytes, float* recon, int training, void* stream);
MMVAE_API int mmvae_decoder_bwd(mmvae_net* net, int N, const float* d_recon, const float* params, float* grads, void* workspace,
                      size_t workspace_bytes, float* d_encoding /* may be NULL */, void* stream);
/* The same with the Gaussian reconstruction loss folded in (reference model.py:403, VAE.loss with decoder_out_channels == in_channels):
 * the gradient entering the decoder is d_recon = coef / sigma^2 * gscale[0] * (recon - target), recon being the reconstruction the last
 * mmvae_decoder_fwd on this workspace returned (full size, before any crop).  It is evaluated inside the output BatchNorm's backward
 * from the saved conv output and `target` [N,out_ch,S,S] f32 -- no d_recon tensor, two passes over the plane instead of four.
 * gscale: device scalar (the upstream gradient of the loss), nullable (= 1). */
MMVAE_API int mmvae_decoder_bwd_gauss(mmvae_net* net, int N, const float* target, float sigma, float coef, const float* gscale, const float* params,
                            float* grads, void* workspace, size_t workspace_bytes, float* d_enc, void* stream);

/* Weight gradients run on a side stream owned by the net.  By default every backward entry point orders
 * them before `stream` again when it returns.  mmvae_net_defer_join(net, 1): mmvae_decoder_bwd leaves its weight gradients in
 * flight; they are ordered before `stream` by the next mmvae_encoder_bwd on this net or by mmvae_net_join(net, stream), whichever
 * comes first -- call one of them before anything on `stream` reads the decoder's gradients (optimizer, all-reduce). */
MMVAE_API int mmvae_net_defer_join(mmvae_net* net, int enable);
MMVAE_API int mmvae_net_join(mmvae_net* net, void* stream);
/* The net's side stream, ordered behind everything enqueued on `stream` so far (`stream` itself when the net has none).  Work the caller
 * enqueues on it runs beside `stream` and is ordered before `stream` again by mmvae_net_join or the next backward entry point that joins
 * (above).  Used by the host side to take the loss scalars of reference model.py:385-406 (KL, MMD, NLL sums: logged values that no
 * gradient kernel reads) off the critical stream of a train step. */
MMVAE_API void* mmvae_net_fork(mmvae_net* net, void* stream);
/* The side stream itself (NULL when the net runs on one stream), without a new ordering edge: for work that only depends on what the
 * stream's earlier forks already ordered it behind (the host side enqueues the step's loss scalars there from the decoder's backward). */
MMVAE_API void* mmvae_net_side_stream(mmvae_net* net);
/* Backward form of the up-blocks on 16-wide maps (decoder.uplayer4; reference model.py:70-85 autograd).  1 (default): the block above hands
 * its input gradient down already masked by this block's join ReLU, the join BatchNorms' backward reduce runs unmasked and dy is evaluated
 * by the loaders of the two fused ConvTranspose2d backward passes -- no apply pass, no dy tensors.  0: reduce -> apply -> consumers, the form
 * every other block uses.  Same arithmetic per element; the parity tests compare the two. */
MMVAE_API int mmvae_net_set_join_grad(mmvae_net* net, int enable);

/* SyncBN: BatchNorm statistics over the global batch of a data-parallel job.  `fn` must SUM the `n` f32 values at
 * device pointer `buf` over all ranks in place, ordered on `stream` (the caller's stream or the net's side stream), and return 0;
 * it is called from inside the four network entry points, in the same order on every rank (two per BatchNorm and direction).
 * `world` = number of ranks; equal shards per rank are assumed (count = local count * world).  dgamma / dbeta stay per-rank
 * sums, like every other parameter gradient.  fn == NULL restores per-rank statistics. */
typedef int (*mmvae_allreduce_fn)(float* buf, int64_t n, void* stream, void* user);
MMVAE_API int mmvae_net_set_sync_bn(mmvae_net* net, mmvae_allreduce_fn fn, void* user, int world);

/* ------------------------------------------------------------------ data-parallel exchange (the reference has none,
 * main.py:433-437).  One RCCL communicator per process / GPU; RCCL is bound at run time (the copy already in the process,
 * else librccl.so / $MMVAE_RCCL_LIB).  Rank 0 creates the id, the host hands its 128 bytes to every rank (e.g. through the
 * torch.distributed store), every rank calls mmvae_comm_init (a collective).  mmvae_comm_allreduce: in-place f32 SUM over the
 * ranks, enqueued on `stream` (xGMI ring / tree chosen by RCCL); the gradient buckets and the SyncBN rows use it. */
typedef struct mmvae_comm mmvae_comm;
#define MMVAE_COMM_ID_BYTES 128
MMVAE_API int mmvae_comm_unique_id(void* id_host /* MMVAE_COMM_ID_BYTES, host */);
MMVAE_API int mmvae_comm_init(mmvae_comm** out, int world, int rank, const void* id_host);
MMVAE_API int mmvae_comm_allreduce(mmvae_comm* comm, float* buf, int64_t n, void* stream);
MMVAE_API int mmvae_comm_destroy(mmvae_comm* comm);
/* SyncBN through the communicator instead of a host callback: the row all-reduces are enqueued in-stream by the library.
 * comm == NULL restores per-rank statistics. */
MMVAE_API int mmvae_net_set_sync_bn_comm(mmvae_net* net, mmvae_comm* comm);
/* The rows are issued from TWO streams (the caller's and the net's side stream, which carries the shortcut branches), and a third
 * stream may carry the gradient buckets: RCCL orders the collectives of one communicator, so sharing one communicator would queue the
 * encoder's SyncBN rows behind the decoder's gradient bucket and relies on RCCL serialising concurrent streams.  This form gives the
 * rows of each stream a communicator of their own (both must differ from the one the gradient buckets use); on every rank each of
 * them sees its collectives in program order.  comm_side == NULL: side-stream rows use comm_main as well (the one-communicator form). */
MMVAE_API int mmvae_net_set_sync_bn_comm2(mmvae_net* net, mmvae_comm* comm_main, mmvae_comm* comm_side);

/* ------------------------------------------------------------------ latent + loss (model.py:148-150, :364-406)
 * Reparameterisation  enc = mu + eps * exp(0.5*logvar)  (VAE_Encoder.rsample, model.py:148-150). */
MMVAE_API int mmvae_rsample_fwd(const float* mu, const float* logvar, const float* eps, float* enc, int64_t n, void* stream);
MMVAE_API int mmvae_rsample_bwd(const float* d_enc, const float* logvar, const float* eps, float* d_mu, float* d_logvar, int64_t n,
                      void* stream);
/* acc[0] += -0.5*sum(logvar - exp(logvar) - mu^2 + 1)   (VAE.kl_divergence, model.py:364-365); acc is f64, caller-zeroed */
MMVAE_API int mmvae_kl_fwd(const float* mu, const float* logvar, int64_t n, double* acc, void* stream);
/* d_mu = c*mu ; d_logvar = c*0.5*(exp(logvar)-1), c = coef * (gscale ? gscale[0] : 1).  In every *_bwd below `gscale` is an
 * optional DEVICE scalar (the upstream d(loss), so loss.backward() needs no host sync). */
MMVAE_API int mmvae_kl_bwd(const float* mu, const float* logvar, float coef, const float* gscale, float* d_mu, float* d_logvar, int64_t n,
                 void* stream);
/* acc[0] += -sum log N(target; recon, sigma)   (model.py:403) */
MMVAE_API int mmvae_gauss_nll_fwd(const float* recon, const float* target, int64_t n, float sigma, double* acc, void* stream);
MMVAE_API int mmvae_gauss_nll_bwd(const float* recon, const float* target, int64_t n, float sigma, float coef, const float* gscale,
                        float* d_recon, void* stream);
/* acc[0] += sum w[t]*CE(recon[:, :, p], t)   (F.cross_entropy(reduction='none', weight).sum(), model.py:400-401);
 * recon [N,Q,HW] f32, target [N,HW] int64, weight [Q] f32 or NULL */
MMVAE_API int mmvae_ce_fwd(const float* recon, const int64_t* target, const float* weight, int N, int Q, int HW, double* acc, void* stream);
MMVAE_API int mmvae_ce_bwd(const float* recon, const int64_t* target, const float* weight, int N, int Q, int HW, float coef,
                 const float* gscale, float* d_recon, void* stream);
/* acc[0] += sum k(x,x) + sum k(y,y) - 2 sum k(x,y),  k(a,b) = exp(-|a-b|^2 / d^2)   (compute_mmd, model.py:367-383);
 * x = true_samples, y = encoding, both [n,d] f32.  Never materialises (n,n,d).  scratch: 2n floats (row norms) selects the
 * exact-f32 MFMA path |x|^2+|y|^2-2x.y; NULL the direct (x-y)^2 VALU path. */
MMVAE_API int mmvae_mmd_fwd(const float* x, const float* y, int n, int d, float* scratch, double* acc, void* stream);
/* d_y += coef * d(mmd)/dy */
MMVAE_API int mmvae_mmd_bwd(const float* x, const float* y, int n, int d, float coef, const float* gscale, float* d_y, void* stream);
/* k[i][j] = exp(-mean_d((x_i - y_j)^2) / d), out (n, m) f32   (VAE.compute_kernel, model.py:367-376) */
MMVAE_API int mmvae_rbf_kernel(const float* x, const float* y, int n, int m, int d, float* out, void* stream);
/* acc = {px, kl, mmd} (f64) -> out[4] = {(nll*px + kl_coef*kl + mmd_coef*mmd)/n, nll*px/n, kl/n, mmd/n}  (model.py:405-406) */
MMVAE_API int mmvae_loss_finish(const double* acc, float* out, float nll, float kl_coef, float mmd_coef, float n, void* stream);

/* ------------------------------------------------------------------ train-step plumbing
 * (label - mean)/std, main.py:383-387.  labels int64 [n]; image f32 [n]. */
MMVAE_API int mmvae_normalise_labels(const int64_t* labels, int64_t n, float mean, float stdv, float* image, void* stream);
/* Input pipeline on device: uint8 pixel -> x/255 -> nearest of q k-means centres (kmeans.predict, main.py:25,
 * utils.py:279-309; lowest index wins ties) -> labels int64 (may be NULL) and image = (label-mean)/std f32 (may be NULL). */
MMVAE_API int mmvae_quantise_normalise(const uint8_t* frames, int64_t n, const float* centres, int q, float mean, float stdv,
                             int64_t* labels, float* image, void* stream);
/* torch.optim.Adam defaults (main.py:468) over a flat buffer: bc1 = 1-beta1^t, bc2_sqrt = sqrt(1-beta2^t);
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
MMVAE_API int mmvae_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, float grad_scale, void* stream);

/* The same update with the step count on the DEVICE (step_dev[0], a double, incremented by the call; bias corrections are computed
 * from it inside the kernel): the form to capture in a HIP graph -- a captured launch replays its scalar arguments, so the host-side
 * bias corrections of mmvae_adam_step would freeze at the captured step. */
MMVAE_API int mmvae_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, double* step_dev, float grad_scale, void* stream);

/* ------------------------------------------------------------------ single ops (layer-level parity tests)
 * Conv2d / ConvTranspose2d with PyTorch weight layouts on NHWC activations of `dtype`:
 *   x [N,H,W,Cin], y [N,Ho,Wo,Cout];  weight f32 (Cout,Cin,k,k) for Conv2d, (Cin,Cout,k,k) for ConvTranspose2d.
 * scratch: device buffer of at least 2*numel(weight)*sizeof(dtype)+64 bytes for the packed weights.
 * mmvae_conv2d_fwd with weight == NULL reuses the packed weights an earlier call with the same geometry left in `scratch`.
 * pro_scale/pro_shift (per input channel, nullable): x := relu?(x*scale+shift) applied on load (fused BN+ReLU).
 * stats (nullable): per-channel (sum,sumsq) partials [rows][2][Cout]; the call returns `rows`. */
MMVAE_API int mmvae_conv2d_fwd(int dtype, int transposed, const void* x, const float* weight, void* y, int N, int H, int W, int Cin, int Cout,
                     int k, int stride, int pad, const float* pro_scale, const float* pro_shift, int pro_relu, float* stats,
                     void* scratch, void* stream);
/* dx [N,H,W,Cin] from dy [N,Ho,Wo,Cout] */
MMVAE_API int mmvae_conv2d_dgrad(int dtype, int transposed, const void* dy, const float* weight, void* dx, int N, int H, int W, int Cin,
                       int Cout, int k, int stride, int pad, void* scratch, void* stream);
/* dW (f32, weight layout) += ... ; pro_* as in fwd (applied to x).  scratch: MMVAE_WGRAD_SCRATCH_BYTES of device memory
 * (per-block partial images, summed by a second kernel: no atomics, bit-reproducible), owned by the caller. */
#define MMVAE_WGRAD_SCRATCH_BYTES (64u << 20)
MMVAE_API int mmvae_conv2d_wgrad(int dtype, int transposed, const void* x, const void* dy, float* dweight, int N, int H, int W, int Cin,
                       int Cout, int k, int stride, int pad, const float* pro_scale, const float* pro_shift, int pro_relu,
                       void* scratch, void* stream);
/* Weight gradients of a residual block's conv1 (Conv2d 3x3 stride 2 pad 1) AND of its 1x1 stride-2 shortcut conv from ONE pass over their
 * common input x (encoder.layer1, model.py:29,135-138): dweight [Cout][Cin][3][3] += dy^T (x) x taps, dweight_sc [Cout][Cin][1][1] += dy_sc^T (x) x.
 * bf16, 32 -> 32 channels, 32x32 -> 16x16 (MMVAE_ERR_UNSUPPORTED otherwise); pro_*: BatchNorm+ReLU applied to x on load; scratch as above. */
MMVAE_API int mmvae_conv2d_wgrad_pair(int dtype, const void* x, const void* dy, const void* dy_shortcut, float* dweight, float* dweight_sc, int N,
                            int H, int W, int Cin, int Cout, const float* pro_scale, const float* pro_shift, int pro_relu, void* scratch,
                            void* stream);
/* ---- BatchNorm2d in training mode and the 1-channel stem, op level (what a maintainer binds instead of torch.nn.BatchNorm2d,
 * model.py:14,20,30,95,..., and of encoder.conv1 + encoder.bn1, model.py:94-95,103)
 * Tensors are NHWC [npix][C] of `dtype`; statistics, parameters and their gradients are f32.
 * scratch: MMVAE_BN_SCRATCH_BYTES of device memory, caller-owned (partial sums of the two-pass reductions, coefficient rows). */
#define MMVAE_BN_SCRATCH_BYTES (16u << 20)
/* out = relu?( (y - mean) * istd * gamma + beta ) with the statistics of this batch (biased variance); running_mean / running_var
 * (nullable) are updated with `momentum` and the unbiased variance, *nbt (nullable) += 1; save_mean / save_istd (C floats each)
 * are kept for the backward pass. */
MMVAE_API int mmvae_batchnorm_fwd(int dtype, const void* y, int64_t npix, int C, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, int64_t* nbt, float momentum, float eps, int relu, void* out, float* save_mean, float* save_istd,
                        void* scratch, void* stream);
/* dy = d(loss)/dy, dgamma += , dbeta += ; dout is the gradient w.r.t. `out`; out (nullable): the forward result when a ReLU was
 * fused (its sign is the mask), NULL for a plain BatchNorm. */
MMVAE_API int mmvae_batchnorm_bwd(int dtype, const void* dout, const void* y, const void* out, int64_t npix, int C, const float* gamma,
                        const float* save_mean, const float* save_istd, void* dy, float* dgamma, float* dbeta, void* scratch, void* stream);
/* encoder.conv1: Conv2d(1 -> 32, k5 s2 p2, no bias) on x [N,S,S] of `dtype` -> y [N,S/2,S/2,32]; stats as mmvae_conv2d_fwd.
 * scratch: >= 2 KB * sizeof(dtype) for the packed weights. */
MMVAE_API int mmvae_stem_fwd(int dtype, const void* x, const float* weight, void* y, int N, int S, float* stats, void* scratch, void* stream);
/* Backward of relu(bn(conv1(x))) w.r.t. the parameters in ONE pass over g and y0 (stem_bwd.hip): g = gradient at the block input
 * (post-ReLU activation), y0 = conv1 output, (bn_scale, bn_shift) = gamma*istd, beta - mean*gamma*istd.
 * dweight (32,1,5,5) +=, dgamma +=, dbeta +=.  S in {64, 32, 16}.  scratch: MMVAE_BN_SCRATCH_BYTES. */
MMVAE_API int mmvae_stem_bwd(int dtype, const void* g, const void* y0, const void* x, const float* weight, const float* gamma,
                   const float* bn_scale, const float* bn_shift, const float* save_mean, const float* save_istd, float* dweight,
                   float* dgamma, float* dbeta, int N, int S, void* scratch, void* stream);
/* ConvTranspose2d backward in one pass over dy (bf16; Cin = 16 or 32, Cout = 16, k4 s2 p1, 32x32 -> 64x64 or 16x16 -> 32x32; else
 * MMVAE_ERR_UNSUPPORTED):
 * dweight (Cin,Cout,4,4) += , dx [N,H,W,Cin] = d(loss)/dx  (+ x2 (x) w2: x2 [N,H,W,16], w2 f32 (Cin,16,1,1), both nullable).
 * pro_*: as in mmvae_conv2d_wgrad (applied to x).  scratch: (numel(weight) + numel(w2)) * sizeof(dtype) for the packed weights;
 * wscratch: MMVAE_WGRAD_SCRATCH_BYTES.
 * dw2 (nullable, needs x2): the weight gradient of the 1x1 convolution whose data gradient x2 (x) w2 is -- x2 is that conv's output
 * gradient, pro(x) its input -- in the Conv2d layout (16, Cin, 1, 1) (the transpose of w2's (Cin, 16)): dw2 += x2^T (x) pro(x), from the
 * same pass (decoder DeconvBottleneck: conv1's gradient rides on the upsample branch's pass, reference model.py:60,70-72). */
MMVAE_API int mmvae_convT_bwd_fused(int dtype, const void* x, const void* dy, const float* weight, float* dweight, void* dx, int N, int H, int W,
                          int Cin, int Cout, int k, int stride, int pad, const float* pro_scale, const float* pro_shift, int pro_relu,
                          const void* x2, const float* w2, float* dw2, void* scratch, void* wscratch, void* stream);
/* ---- last up-block + tail conv, one output plane (decoder.uplayerN -> decoder.conv2, reference model.py:86-88,193) ----
 * y2, ys: the two branch outputs [N,H,W,16] of `dtype` (BatchNorm not yet applied); (s2,b2), (ss,bs): per-channel f32
 * scale/shift of their BatchNorms; the block output is x = relu(y2*s2+b2 + ys*ss+bs).  weight f32 (1,16,3,3), bias f32 (1).
 * H, W: W a power of two <= 128 (bf16) / 64 (f32), H*W a multiple of that bound; other shapes return MMVAE_ERR_UNSUPPORTED.
 *
 * mmvae_tail_join_fwd: r_raw [N,1,H,W] f32 = conv2d(x, weight, bias, padding=1) without storing x.
 *   stats (nullable): per-image (sum, sumsq) of r_raw, [N][2]; returns N. */
MMVAE_API int mmvae_tail_join_fwd(int dtype, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs,
                        const float* weight, const float* bias, float* r_raw, float* stats, int N, int H, int W, void* stream);
/* The same result as a per-wave MFMA stream (bf16, one output plane, 64x64 only; else MMVAE_ERR_UNSUPPORTED) -- what the network runs:
 * the joined activation and the weights pass through bf16 on their way to the MFMA (mmvae_tail_join_fwd keeps both in f32).
 * stats (nullable): partial (sum, sumsq) rows [rows][2] of r_raw; returns rows (<= N). */
MMVAE_API int mmvae_tail_join_fwd_stream(int dtype, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs,
                               const float* weight, const float* bias, float* r_raw, float* stats, int N, int H, int W, void* stream);
/* mmvae_tail_join_bwd_reduce: with g = conv_transpose2d(d_raw, weight, padding=1) masked by x > 0 (never stored), writes per-block
 *   partial sums partials[rows][3][16] = (sum g, sum g*y2, sum g*ys) and, when wpartials != NULL, the tail conv's weight-gradient
 *   partials wpartials[rows][16][9] (sum over pixels of x[ci] * d_raw[h+1-kh, w+1-kw]); returns rows (<= 1024).
 *   d_raw [N,out_planes,H,W] f32, weight (out_planes,16,3,3); wpartials needs out_planes == 1. */
MMVAE_API int mmvae_tail_join_bwd_reduce(int dtype, const float* d_raw, const float* weight, int out_planes, const void* y2, const float* s2,
                               const float* b2, const void* ys, const float* ss, const float* bs, float* partials, float* wpartials,
                               int N, int H, int W, void* stream);
/* mmvae_tail_join_bwd_apply: dy2 = A2*g + B2*y2 + C2, dys = As*g + Bs*ys + Cs with the same masked g and per-channel f32
 *   coefficient vectors (the BatchNorm-backward affine forms); dy2, dys [N,H,W,16] of `dtype`. */
MMVAE_API int mmvae_tail_join_bwd_apply(int dtype, const float* d_raw, const float* weight, int out_planes, const void* y2, const float* s2,
                              const float* b2, const void* ys, const float* ss, const float* bs, const float* A2, const float* B2,
                              const float* C2, const float* As, const float* Bs, const float* Cs, void* dy2, void* dys, int N, int H, int W,
                              void* stream);
/* ---- the last up-block's backward in ONE pass (conv_joinbwd.hip; bf16, 16 channels, 32x32 -> 64x64, one output plane) ----
 * DeconvBottleneck (reference model.py:70-85) in front of the tail conv (:193): with the join's BatchNorm-backward coefficients at hand
 * (mmvae_tail_join_bwd_reduce + finalize), the kernel produces dy2 / dys row by row in LDS and consumes them in place:
 *   dw_conv2 (16,16,4,4) += a1^T (x) dy2,  d_a1 = conv2's data gradient [N,32,32,16],  bn1_sums[2][16] = (sum g, sum g*y1), g = d_a1 [bn1(y1) > 0]
 *   dw_up    (16,16,4,4) += pro(xin)^T (x) dys,  g_in = the upsample branch's data gradient [N,32,32,16]   (conv1's share: mmvae_conv1x1_bwd_fused)
 * y1: conv2's input before bn1 (s1, b1 = bn1's scale / shift, ReLU); xin: the block input (sx, bx: optional BatchNorm+ReLU in front, nullable).
 * Nothing of size dy2 / dys is written.  scratch: MMVAE_WGRAD_SCRATCH_BYTES (packed weights, partial images). */
MMVAE_API int mmvae_upblock_bwd_fused(const float* d_raw, const float* tail_weight, const void* y2, const float* s2, const float* b2,
                            const void* ys, const float* ss, const float* bs, const float* A2, const float* B2, const float* C2,
                            const float* As, const float* Bs, const float* Cs, const void* y1, const float* s1, const float* b1,
                            const float* w_conv2, float* dw_conv2, void* d_a1, float* bn1_sums, const void* xin, const float* sx,
                            const float* bx, const float* w_up, float* dw_up, void* g_in, int N, void* scratch, void* stream);
/* conv1 (1x1, 16 -> 16) of the same block, once bn1's sums are final (A1, B1, C1 = its backward coefficients):
 *   dy1 = A1 (d_a1 [bn1(y1) > 0]) + B1 y1 + C1;  g_in += dy1 (x) W1 (in place);  dw_conv1 (16,16,1,1) += dy1^T (x) pro(xin).
 * rows = N * H rows of 32 pixels.  scratch: MMVAE_WGRAD_SCRATCH_BYTES. */
/* The same block's FORWARD tail: r_raw = decoder.conv2(relu(bn2(conv2(relu(bn1(y1)))) + upsample.1(upsample.0(x_in)))) (model.py:70-85,193) with both
 * ConvTranspose2d branch outputs recomputed row by row from their 32x32x16 inputs instead of read back (they are still written by
 * mmvae_conv2d_fwd for the statistics and the backward pass).  bf16 NHWC inputs, f32 weights in PyTorch layout (w2, wu: [16][16][4][4]; tail
 * weight [1][16][3][3] + bias), s / b pairs = forward scale / shift of the BatchNorms (sx / bx may be NULL: x_in is an activation);
 * r_raw f32 [N][64][64]; stats (nullable) receives <return value> partial rows [2] = (sum, sum of squares) of r_raw; scratch >= 16 KB. */
MMVAE_API int mmvae_upblock_tail_fwd(const void* y1, const float* s1, const float* b1, const float* w2, const void* x_in, const float* sx,
                                     const float* bx, const float* wu, const float* s2, const float* b2, const float* ss, const float* bs,
                                     const float* tail_weight, const float* tail_bias, float* r_raw, float* stats, int N, void* scratch,
                                     void* stream);
/* A block's residual join fused with the NEXT block's conv1 (reference model.py:86-88 of block i, :72 of block i + 1; conv_joinfwd.hip; bf16):
 *   out = relu(s2 y2 + b2 + ss ys + bs)  [npix][C]  (C = 16 or 32; written: the upsample branch and the backward pass read it),
 *   y1 = conv1(out), w_conv1 (16, C, 1, 1) f32, no bias, [npix][16];  stats (nullable): <return value> partial rows [2][16] = (sum, sum of
 *   squares) of y1 from the f32 accumulators.  npix a multiple of 32; scratch >= 2 KB (the packed weight). */
MMVAE_API int mmvae_join_conv1x1_fwd(const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs,
                           const float* w_conv1, int C, void* out, void* y1, float* stats, int64_t npix, void* scratch, void* stream);
MMVAE_API int mmvae_conv1x1_bwd_fused(const void* d_a1, const void* y1, const float* s1, const float* b1, const float* A1, const float* B1,
                            const float* C1, const void* xin, const float* sx, const float* bx, const float* w_conv1, float* dw_conv1,
                            void* g_in, int64_t rows, void* scratch, void* stream);
/* ------------------------------------------------------------------ PixelCNN (reference model.py:212-255)
 * PixelCNN(in_channels, intermediate_channels, out_channels, layers, "ReLu"):  InstanceNorm2d -> [MaskedConv2d('A' first, then 'B', 7x7 pad 3,
 * bias) -> InstanceNorm2d -> ReLU] x (layers - 1) -> MaskedConv2d('B').  The masks (model.py:216-220) keep the first 24 ('A') / 25 ('B') taps
 * of the 7x7 kernel in row-major order: the layers run as convolutions over exactly those taps, so a masked tap's stored weight is never
 * read and its gradient is left untouched (the reference zeroes the tap before every use, model.py:222).
 * params / grads: flat f32, per layer weight (out, in, 7, 7) then bias (out) -- the reference's parameter order.  x, d_x: [N,in,S,S] f32;
 * out, d_out: [N,out,S,S] f32 logits.  in / out channels <= 16, intermediate_channels a multiple of 16 in [16, 256], 2 <= layers <= 16;
 * dtype f32 or bf16 (storage of the activations between the layers).  mmvae_pixelcnn_bwd needs the workspace of the forward it
 * differentiates and that forward's x. */
typedef struct mmvae_pixelcnn mmvae_pixelcnn;
MMVAE_API int mmvae_pixelcnn_create(mmvae_pixelcnn** out, int in_channels, int intermediate_channels, int out_channels, int layers, int dtype);
MMVAE_API void mmvae_pixelcnn_destroy(mmvae_pixelcnn* net);
MMVAE_API int64_t mmvae_pixelcnn_num_params(mmvae_pixelcnn* net);
MMVAE_API size_t mmvae_pixelcnn_workspace_bytes(mmvae_pixelcnn* net, int N, int S);
MMVAE_API int mmvae_pixelcnn_fwd(mmvae_pixelcnn* net, int N, int S, const float* x, const float* params, void* workspace, size_t workspace_bytes,
                       float* out, void* stream);
MMVAE_API int mmvae_pixelcnn_bwd(mmvae_pixelcnn* net, int N, int S, const float* x, const float* d_out, const float* params, float* grads,
                       void* workspace, size_t workspace_bytes, float* d_x, void* stream);

/* f32 <-> dtype element conversion (n elements) */
MMVAE_API int mmvae_convert(int dtype_in, int dtype_out, const void* in, void* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_H_ */
