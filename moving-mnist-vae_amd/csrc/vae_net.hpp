// Network-level orchestration of the conv-VAE hot path (encoder / decoder forward + backward).
// Mirrors the layer structure of the reference model.py (VAE_Encoder :88-150, VAE_Decoder :153-209,
// BasicBlock :23-55, DeconvBottleneck :57-85) but owns no tensors: parameters, BN buffers, gradients and
// the activation workspace are caller-provided device pointers.
#pragma once
#include <string>
#include <vector>

#include "comm.hpp"
#include "conv_ops.hpp"
#include "kernels.hpp"

namespace mmvae {

struct NetCfg { int in_ch, z, out_ch, S, need_logvar, dtype, blocks, fp8; };   // fp8: forward convs of the deep layers on the fp8 MFMA   // blocks: residual blocks per stage (reference: 1)

enum EntryKind : int { EK_PARAM = 0, EK_BN_F32 = 1, EK_BN_I64 = 2 };
struct Entry { std::string name; int ndim; int shape[4]; int kind; long offset; };

// A conv-like weight [D0][D1][k][k] relating a "small" tensor S [N,Hs,Ws,D0] and a "large" tensor L [N,Hl,Wl,D1],
// Hs = floor((Hl + 2p - k)/s) + 1.  Conv2d: weight (out,in,k,k), forward = down(L->S).  ConvTranspose2d: weight
// (in,out,k,k), forward = up(S->L).
// fp8 (BASELINE configs[4]): the conv's FORWARD runs on the fp8 MFMA (e4m3 weights x e4m3 activations, f32 accumulate) with
// the static power-of-two weight scale wscale; its stored output is y' = wscale * y, which the BatchNorm behind it absorbs exactly
// (BnFinalizeArgs::in_scale); the backward pass uses the bf16 kernels on wscale * w and scales the weight gradient by wscale.
struct ConvW { long off; int D0, D1, k, s, p; long packD, packU; int Hl = 0; /* large-side map (square) */ bool tr = false; /* ConvTranspose2d: forward = up */ bool fp8 = false; float wscale = 1.f; };
struct Bn { long g_off, b_off, rm_off, rv_off; int nbt_idx; int C; long ws; /* float offset of this BN's 7*C scratch */ };

struct Block {   // encoder BasicBlock or decoder DeconvBottleneck (both: main c1->c2, shortcut cs, join)
  ConvW c1, c2, cs; Bn b1, b2, bs;
  int Cin, C, Hin, Win, Hmid, Wmid, Hout, Wout;
  long y1, y2, ys, out;   // byte offsets in the workspace
  // identity: a shape-preserving block of a deeper-than-reference net (blocks > 1): no shortcut conv / BatchNorm, the shortcut is
  // the block input itself (model.py:39-55 with downsample=None; decoder: conv2 is a 3x3 Conv2d instead of the ConvTranspose2d)
  bool identity = false;
};

struct Plan {
  int N = -1;
  size_t bytes = 0;
  long x_t, y0, enc_t, y0d, r_raw, d_raw, col;           // byte offsets
  long packed;                                           // packed weights (T)
  long bnws;                                             // float scratch for all BNs
  long partials;                                         // reduction partials (floats)
  long syncbuf;                                          // SyncBN all-reduce operands: 2 x 1024 floats (main / side stream)
  long edy1[2], edy2[2], edys[2];                         // private dy sets of encoder blocks 2 and 3 (index i - 2): the side
                                                          // stream may still read the shared sets for the decoder when they run
  long g[2], dy1[2], dy2[2], dys[2], da1, dh;            // backward temporaries (dy*: two sets, alternating per block, so the
                                                         // weight gradients on the side stream may lag one block behind)
  long wscratch;                                         // partial-image scratch of the weight gradients (serialised on the side stream)
  long wscratch2;                                        // partial images of the fused dgrad+wgrad passes (caller's stream)
  long stem_R, stem_gram;
  long cvec;                                             // constants: 256 ones, 256 zeros (identity shortcuts as a unit BatchNorm)
  long act0d;                                            // blocks > 1: relu(bn(decoder stem)) materialised (an identity shortcut needs it)                                // stem backward: patch gram matrix (1024 doubles) and its per-block partials
  long enc_ws_end;
};

constexpr long kPartialFloats = 4096L * 3 * 256;   // floats of one reduction-partials region

class Net {
 public:
  explicit Net(const NetCfg& c);
  ~Net();                            // destroys the side streams / events the net created (the caller's buffers are the caller's)
  Net(const Net&) = delete;
  Net& operator=(const Net&) = delete;
  NetCfg cfg;
  std::vector<Entry> entries;
  long n_params = 0, n_bnbuf = 0; int n_nbt = 0;
  long dec_param_off = 0;            // first decoder parameter (flat f32 index)
  long n_packed = 0;                 // packed weight elements
  long n_bnws = 0;                   // floats of BN scratch
  long max_w = 0;                    // elements of the largest conv weight
  int H1, W1, Hf, Wf, Sd, nup;
  // layers
  ConvW stem; Bn bn0;
  std::vector<Block> enc;            // 4 stages x cfg.blocks
  ConvW head_mu, head_lv; long head_pack_mu, head_pack_lv, head_pack_dg;
  ConvW dstem; Bn dbn0;
  std::vector<Block> dec;            // nup stages x cfg.blocks (reference order: the upsampling block is the LAST of its stage)
  ConvW tail; long tail_bias; Bn bn_out;
  long stem_pack, tail_pack_f, tail_pack_d;   // packed-weight slots of the boundary layers (channel-padded)

  size_t workspace_bytes(int N);
  const Plan& plan(int N);

  // staged: the storage-type copy of x is already in the workspace (stage_labels wrote it together with x): no conversion pass
  int encoder_fwd(int N, const float* x, const float* params, float* bnbuf, long long* nbt, void* ws, size_t ws_bytes,
                  float* mu, float* logvar, int training, hipStream_t s, bool staged = false);
  // labels (int64 or uint8, N * in_ch * S * S of them) -> image = (label - mean) / std as f32 (the caller's tensor: the network input and the
  // Gaussian loss target) AND as the storage type straight into the workspace's input slot, one pass (main.py:383-387)
  int stage_labels(int N, const void* labels, int label_bytes, float mean, float stdv, float* image, void* ws, size_t ws_bytes, hipStream_t s);
  int encoder_bwd(int N, const float* d_mu, const float* d_logvar, const float* params, float* grads, void* ws, size_t ws_bytes,
                  hipStream_t s);
  int decoder_fwd(int N, const float* enc, const float* params, float* bnbuf, long long* nbt, void* ws, size_t ws_bytes,
                  float* recon, int training, hipStream_t s);
  // gauss (optional, then d_recon == NULL): the reconstruction gradient is the Gaussian NLL's, d_recon = coef / sigma^2 * gscale[0] *
  // (recon - target) with recon the forward's output -- computed inside the output BatchNorm's backward, never stored
  struct GaussTail { const float* target; float sigma; float coef; const float* gscale; };
  int decoder_bwd(int N, const float* d_recon, const float* params, float* grads, void* ws, size_t ws_bytes, float* d_enc,
                  hipStream_t s, const GaussTail* gauss = nullptr);

 private:
  Plan plan_;
  float* wscratch_ = nullptr;
  // weight gradients run on a side stream, concurrently with the dgrad / BatchNorm-backward chain of the same block
  // (the deep-layer kernels are latency-bound and leave most CUs idle); MMVAE_SIDE_STREAM=0 disables it
  static constexpr int kForkEvents = 192;
  hipStream_t side_ = nullptr; hipEvent_t ev_[kForkEvents] = {}; int evi_ = 0; int side_state_ = 0;   // 0 unknown, 1 on, -1 off
  hipStream_t wgrad_stream(hipStream_t s);      // stream the weight gradients are enqueued on
  int side_fork(hipStream_t s);                 // side stream waits for everything enqueued on s so far
  int side_join(hipStream_t s);                 // s waits for everything enqueued on the side stream so far
  hipEvent_t gram_ev_ = nullptr;                 // the stem's patch gram matrix is ready (side stream)
  // the gram matrix depends on the staged input only: decoder_bwd launches it at ITS start, where the side stream idles beside the last
  // up-block's HBM-bound backward, when the encoder forward it belongs to ran on this workspace (encoder_bwd launches it otherwise)
  bool gram_ready_ = false; int gram_fwd_N_ = 0; const void* gram_fwd_ws_ = nullptr;
  hipEvent_t blk_ev_[16] = {};                   // side-stream progress marks, one per block of a backward pass
  int side_mark(int slot);                      // record mark `slot` on the side stream
  int side_wait_mark(int slot, hipStream_t s);  // s waits for mark `slot`
  int dt() const { return cfg.dtype; }
  size_t esz() const { return dtype_size(cfg.dtype); }
  ConvW add_conv(const std::string& name, int D0, int D1, int k, int s, int p, bool pack, bool transposed = false);
  void settle_fp8(ConvW& w) const;
  Bn add_bn(const std::string& prefix, int C);
  void add_entry(const std::string& name, std::initializer_list<int> shape, int kind, long off);

  int packs_enc_fwd(const float* params, char* base, hipStream_t s);
  int packs_enc_bwd(const float* params, char* base, hipStream_t s);
  int packs_dec_fwd(const float* params, char* base, hipStream_t s);
  int packs_dec_bwd(const float* params, char* base, bool need_denc, hipStream_t s);
  // fragment-major packing (deep2_conv_kernel) of the down / up form of a conv at its place in the net
  int frag_down(const ConvW& w) const;
  int frag_up(const ConvW& w) const;
  int pack_down(const ConvW& w, const float* params, char* base, hipStream_t s);
  int pack_up(const ConvW& w, const float* params, char* base, hipStream_t s);
  // w2 / x2 (optional): the 1x1 conv whose "up" form over x2 (a tensor on the small-side grid) is added in the same kernel
  int run_down(const ConvW& w, char* base, int N, const void* L, int Hl, int Wl, void* S, int Hs, int Ws,
               const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, int out_dt, hipStream_t s,
               const ConvW* w2 = nullptr, const void* x2 = nullptr);
  int run_up(const ConvW& w, char* base, int N, const void* S, int Hs, int Ws, void* L, int Hl, int Wl,
             const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s,
             const ConvW* w2 = nullptr, const void* x2 = nullptr);
  int run_wgrad(const ConvW& w, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b,
                const void* G, int Hl, int Wl, const float* proG_s, const float* proG_b, float* grads, hipStream_t s);
  // Last up-block forward as one kernel (join + tail conv, the joined activation is never stored); the backward then needs the
  // recomputing wgrad and the recomputing join backward.  MMVAE_TAIL_FWD_FUSED=0 restores join -> conv.
  bool tail_fwd_fused() const;
  bool stem_bwd_fused() const;       // stem backward as one pass (stem_bwd.hip)
  bool stem_dg_fused() const;        // ... with encoder.layer1's data gradient recomputed inside it (no stored gradient of the stem's output)
  long stem_dy1_ = 0, stem_dys_ = 0; // workspace offsets of layer1's dy1 / dys of the running backward pass
  // decoder_bwd leaves its weight gradients running on the side stream; encoder_bwd (or join()) orders them before the caller's stream
  bool defer_join_ = false;
  int (*ar_fn_)(float*, long long, void*, void*) = nullptr; void* ar_user_ = nullptr; int ar_world_ = 1;
  Comm* comm_ = nullptr; Comm* comm_side_ = nullptr;
  int sync_rows(char* base, const float* partials, int nparts, int width, hipStream_t s, float** out, int row_stride = 0);
 public:
  void set_defer_join(bool v) { defer_join_ = v; }
  // the join-gradient form of a 16-wide block's backward (decoder_bwd: masked gradient handed down, dy evaluated by the consumers' loaders);
  // off = the reduce -> apply -> consumers form every other block uses (kept reachable for the A/B parity test)
  bool join_grad_ = true;
  void set_join_grad(bool v) { join_grad_ = v; }
  // SyncBN: fn sums a device f32 buffer over all ranks, ordered on the given stream; NULL = per-rank statistics
  typedef int (*AllReduceFn)(float* buf, long long n, void* stream, void* user);
  void set_sync_bn(AllReduceFn fn, void* user, int world) { ar_fn_ = fn; ar_user_ = user; ar_world_ = fn ? world : 1; comm_ = comm_side_ = nullptr; }
  // SyncBN through an RCCL communicator of this library: the row all-reduce is enqueued in-stream (no host callback, capturable)
  // (side: the communicator of the rows issued on the side stream; null = the same one)
  void set_sync_bn_comm(Comm* c, Comm* side = nullptr) { comm_ = c; comm_side_ = c ? side : nullptr; ar_fn_ = nullptr; ar_user_ = nullptr; ar_world_ = c ? comm_world(c) : 1; }
  bool sync_bn_on() const { return ar_fn_ != nullptr || comm_ != nullptr; }
  int join(hipStream_t s) { return side_join(s); }
  hipStream_t side() const { return side_state_ == 1 ? side_ : nullptr; }                       // nullptr: one stream
  hipStream_t fork(hipStream_t s) { return side_fork(s) == MMVAE_OK ? wgrad_stream(s) : s; }   // the side stream, ordered behind s
 private:
  int bn_train(const Bn& bn, const float* params, float* bnbuf, long long* nbt, char* base, int nparts, double count, hipStream_t s, long part_off = 0,
               float in_scale = 1.f);
  int bn_eval(const Bn& bn, const float* params, const float* bnbuf, char* base, hipStream_t s, float in_scale = 1.f);
  // eval mode: all BatchNorms of the encoder (which = 0) / decoder (1) folded to (scale, shift) in one launch; bn_eval() is then a no-op
  int fold_bn_eval(int which, const float* params, const float* bnbuf, char* base, hipStream_t s);
  bool l1_dgrad_stream() const;
  long l1c2_flip = 0;
  bool eval_folded_ = false;
  bool store8 = false;          // fp8 mode: the last up-block's branch outputs are stored as e4m3 bytes (see the constructor)
  float* bnf(const Bn& bn, char* base, int which) const;   // 0 mean 1 istd 2 scale 3 shift 4 A 5 B 6 C
  const float* ones(char* base) const { return reinterpret_cast<const float*>(base + plan_.cvec); }
  const float* zeros(char* base) const { return reinterpret_cast<const float*>(base + plan_.cvec) + 256; }
  int fill_consts(char* base, hipStream_t s);
  int bn_backward_coefs(const Bn& bn, const float* params, float* grads, char* base, int nparts, int ny, int which, double count,
                        hipStream_t s, float* dbias_conv = nullptr);
  BnBwdFinalizeArgs bwd_finalize_args(const Bn& bn, const float* params, float* grads, char* base, const float* partials, int nparts, int ny,
                                      int which, double count) const;
  int bn_backward_coefs_join(const Bn& b2, const Bn& bs, const float* params, float* grads, char* base, int nparts, double count, hipStream_t s);
};

}  // namespace mmvae
