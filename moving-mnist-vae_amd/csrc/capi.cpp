// extern "C" surface declared in include/mmvae.h.
#include <cstring>
#include <new>

#include "../../include/mmvae.h"
#include "conv_ops.hpp"
#include "pixel_net.hpp"
#include "vae_net.hpp"

namespace mmvae { const char* last_error(); }
using namespace mmvae;

struct mmvae_net { Net* net; };
struct mmvae_comm { Comm* c; };

static inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
static inline hipStream_t S_(void* s) { return static_cast<hipStream_t>(s); }   // (where a parameter is called S)

extern "C" {

int mmvae_abi_version(void) { return MMVAE_ABI_VERSION; }
const char* mmvae_last_error(void) { return last_error(); }

int mmvae_net_create(mmvae_net** out, int in_channels, int z, int out_channels, int image_size, int need_logvar, int dtype) {
  return mmvae_net_create_ex(out, in_channels, z, out_channels, image_size, need_logvar, dtype, 1);
}
int mmvae_net_create_ex(mmvae_net** out, int in_channels, int z, int out_channels, int image_size, int need_logvar, int dtype,
                        int blocks_per_stage) {
  if (!out) return MMVAE_ERR_ARG;
  if (blocks_per_stage < 1 || blocks_per_stage > 4) { set_error("blocks_per_stage=%d unsupported (1..4)", blocks_per_stage); return MMVAE_ERR_UNSUPPORTED; }
  if (in_channels < 1 || in_channels > 4) { set_error("in_channels=%d unsupported (1..4; the benchmarked path is the 1-channel Moving-MNIST VAE)", in_channels); return MMVAE_ERR_UNSUPPORTED; }
  if (z <= 0 || z % 8) { set_error("z_dimension=%d must be a positive multiple of 8", z); return MMVAE_ERR_UNSUPPORTED; }
  if (!(out_channels == 1 || out_channels == 2 || out_channels == 3 || out_channels == 4 || out_channels == 8)) {
    set_error("decoder_out_channels=%d unsupported (1,2,3,4,8)", out_channels); return MMVAE_ERR_UNSUPPORTED; }
  if (image_size < 9 || image_size > 64) { set_error("input_image_size=%d unsupported (9..64)", image_size); return MMVAE_ERR_UNSUPPORTED; }
  if (dtype != MMVAE_F32 && dtype != MMVAE_BF16 && dtype != MMVAE_FP8) { set_error("dtype=%d unsupported", dtype); return MMVAE_ERR_ARG; }
  // MMVAE_FP8: bf16 storage everywhere, fp8 (e4m3) MFMA in the forward pass of the deep layers
  NetCfg c{in_channels, z, out_channels, image_size, need_logvar ? 1 : 0, dtype == MMVAE_FP8 ? MMVAE_BF16 : dtype, blocks_per_stage, dtype == MMVAE_FP8 ? 1 : 0};
  mmvae_net* h = new (std::nothrow) mmvae_net;
  if (!h) return MMVAE_ERR_ARG;
  h->net = new (std::nothrow) Net(c);
  if (!h->net) { delete h; return MMVAE_ERR_ARG; }
  *out = h;
  return MMVAE_OK;
}
void mmvae_net_destroy(mmvae_net* n) { if (n) { delete n->net; delete n; } }

int mmvae_net_sizes(const mmvae_net* n, int64_t* n_params, int64_t* n_bn_f32, int32_t* n_bn_i64, int64_t* dec_off, int32_t* dec_side) {
  if (!n) return MMVAE_ERR_ARG;
  if (n_params) *n_params = n->net->n_params;
  if (n_bn_f32) *n_bn_f32 = n->net->n_bnbuf;
  if (n_bn_i64) *n_bn_i64 = n->net->n_nbt;
  if (dec_off) *dec_off = n->net->dec_param_off;
  if (dec_side) *dec_side = n->net->Sd;
  return MMVAE_OK;
}
int mmvae_net_num_entries(const mmvae_net* n) { return n ? (int)n->net->entries.size() : MMVAE_ERR_ARG; }
int mmvae_net_entry(const mmvae_net* n, int i, char* name, int cap, int32_t* ndim, int32_t shape[4], int32_t* kind, int64_t* offset) {
  if (!n || i < 0 || i >= (int)n->net->entries.size()) return MMVAE_ERR_ARG;
  const Entry& e = n->net->entries[i];
  if (name && cap > 0) { std::strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (ndim) *ndim = e.ndim;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = e.shape[k];
  if (kind) *kind = e.kind;
  if (offset) *offset = e.offset;
  return MMVAE_OK;
}
size_t mmvae_net_workspace_bytes(mmvae_net* n, int N) { return n && N > 0 ? n->net->workspace_bytes(N) : 0; }

int mmvae_encoder_fwd(mmvae_net* n, int N, const float* x, const float* params, float* bn_f32, int64_t* bn_i64, void* ws, size_t wsb,
                      float* mu, float* logvar, int training, void* stream) {
  if (!n || N <= 0 || !x || !params || !ws || !mu) { set_error("encoder_fwd: bad argument"); return MMVAE_ERR_ARG; }
  if (n->net->cfg.need_logvar && !logvar) { set_error("encoder_fwd: logvar required"); return MMVAE_ERR_ARG; }
  if (!bn_f32) { set_error("encoder_fwd: BN buffers required"); return MMVAE_ERR_ARG; }
  return n->net->encoder_fwd(N, x, params, bn_f32, reinterpret_cast<long long*>(bn_i64), ws, wsb, mu, logvar, training, S(stream));
}
int mmvae_net_stage_labels(mmvae_net* n, int N, const void* labels, int label_bytes, float mean, float stdv, float* image, void* ws, size_t wsb,
                           void* stream) {
  if (!n || N <= 0 || !labels || !image || !ws) { set_error("net_stage_labels: bad argument"); return MMVAE_ERR_ARG; }
  return n->net->stage_labels(N, labels, label_bytes, mean, stdv, image, ws, wsb, S(stream));
}
int mmvae_encoder_fwd_staged(mmvae_net* n, int N, const float* x, const float* params, float* bn_f32, int64_t* bn_i64, void* ws, size_t wsb,
                             float* mu, float* logvar, int training, void* stream) {
  if (!n || N <= 0 || !x || !params || !ws || !mu) { set_error("encoder_fwd_staged: bad argument"); return MMVAE_ERR_ARG; }
  if (n->net->cfg.need_logvar && !logvar) { set_error("encoder_fwd_staged: logvar required"); return MMVAE_ERR_ARG; }
  if (!bn_f32) { set_error("encoder_fwd_staged: BN buffers required"); return MMVAE_ERR_ARG; }
  return n->net->encoder_fwd(N, x, params, bn_f32, reinterpret_cast<long long*>(bn_i64), ws, wsb, mu, logvar, training, S(stream), true);
}
int mmvae_encoder_bwd(mmvae_net* n, int N, const float* d_mu, const float* d_logvar, const float* params, float* grads, void* ws,
                      size_t wsb, void* stream) {
  if (!n || N <= 0 || !d_mu || !params || !grads || !ws) { set_error("encoder_bwd: bad argument"); return MMVAE_ERR_ARG; }
  if (n->net->cfg.need_logvar && !d_logvar) { set_error("encoder_bwd: d_logvar required"); return MMVAE_ERR_ARG; }
  return n->net->encoder_bwd(N, d_mu, d_logvar, params, grads, ws, wsb, S(stream));
}
int mmvae_decoder_fwd(mmvae_net* n, int N, const float* enc, const float* params, float* bn_f32, int64_t* bn_i64, void* ws, size_t wsb,
                      float* recon, int training, void* stream) {
  if (!n || N <= 0 || !enc || !params || !ws || !recon || !bn_f32) { set_error("decoder_fwd: bad argument"); return MMVAE_ERR_ARG; }
  return n->net->decoder_fwd(N, enc, params, bn_f32, reinterpret_cast<long long*>(bn_i64), ws, wsb, recon, training, S(stream));
}
int mmvae_decoder_bwd(mmvae_net* n, int N, const float* d_recon, const float* params, float* grads, void* ws, size_t wsb, float* d_enc,
                      void* stream) {
  if (!n || N <= 0 || !d_recon || !params || !grads || !ws) { set_error("decoder_bwd: bad argument"); return MMVAE_ERR_ARG; }
  return n->net->decoder_bwd(N, d_recon, params, grads, ws, wsb, d_enc, S(stream));
}

int mmvae_decoder_bwd_gauss(mmvae_net* n, int N, const float* target, float sigma, float coef, const float* gscale, const float* params, float* grads,
                            void* ws, size_t wsb, float* d_enc, void* stream) {
  if (!n || N <= 0 || !target || !params || !grads || !ws || !(sigma > 0.f)) { set_error("decoder_bwd_gauss: bad argument"); return MMVAE_ERR_ARG; }
  const Net::GaussTail gt{target, sigma, coef, gscale};
  return n->net->decoder_bwd(N, nullptr, params, grads, ws, wsb, d_enc, S(stream), &gt);
}

int mmvae_net_defer_join(mmvae_net* n, int enable) {
  if (!n) { set_error("net_defer_join: bad argument"); return MMVAE_ERR_ARG; }
  n->net->set_defer_join(enable != 0);
  return MMVAE_OK;
}
int mmvae_net_join(mmvae_net* n, void* stream) {
  if (!n) { set_error("net_join: bad argument"); return MMVAE_ERR_ARG; }
  return n->net->join(S(stream));
}

void* mmvae_net_fork(mmvae_net* n, void* stream) {
  if (!n) return stream;
  return reinterpret_cast<void*>(n->net->fork(S(stream)));
}

int mmvae_net_set_join_grad(mmvae_net* n, int enable) {
  if (!n) { set_error("net_set_join_grad: bad argument"); return MMVAE_ERR_ARG; }
  n->net->set_join_grad(enable != 0);
  return MMVAE_OK;
}

void* mmvae_net_side_stream(mmvae_net* n) { return n ? reinterpret_cast<void*>(n->net->side()) : nullptr; }

int mmvae_net_set_sync_bn(mmvae_net* n, mmvae_allreduce_fn fn, void* user, int world) {
  if (!n || (fn && world < 1)) { set_error("net_set_sync_bn: bad argument"); return MMVAE_ERR_ARG; }
  n->net->set_sync_bn(reinterpret_cast<mmvae::Net::AllReduceFn>(fn), user, world);
  return MMVAE_OK;
}

// ---- data-parallel exchange
int mmvae_comm_unique_id(void* id) {
  if (!id) { set_error("comm_unique_id: bad argument"); return MMVAE_ERR_ARG; }
  return comm_unique_id(id);
}
int mmvae_comm_init(mmvae_comm** out, int world, int rank, const void* id) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) { set_error("comm_init: bad argument"); return MMVAE_ERR_ARG; }
  mmvae_comm* h = new (std::nothrow) mmvae_comm;
  if (!h) return MMVAE_ERR_ARG;
  h->c = nullptr;
  const int rc = comm_init(&h->c, world, rank, id);
  if (rc < 0) { delete h; return rc; }
  *out = h;
  return MMVAE_OK;
}
int mmvae_comm_allreduce(mmvae_comm* c, float* buf, int64_t n, void* st) {
  if (!c || (!buf && n > 0)) { set_error("comm_allreduce: bad argument"); return MMVAE_ERR_ARG; }
  return comm_allreduce_sum(c->c, buf, (long long)n, S(st));
}
int mmvae_comm_destroy(mmvae_comm* c) {
  if (!c) return MMVAE_OK;
  const int rc = comm_destroy(c->c);
  delete c;
  return rc;
}
int mmvae_net_set_sync_bn_comm(mmvae_net* n, mmvae_comm* c) {
  if (!n) { set_error("net_set_sync_bn_comm: bad argument"); return MMVAE_ERR_ARG; }
  n->net->set_sync_bn_comm(c ? c->c : nullptr);
  return MMVAE_OK;
}
int mmvae_net_set_sync_bn_comm2(mmvae_net* n, mmvae_comm* cm, mmvae_comm* cs) {
  if (!n || (!cm && cs)) { set_error("net_set_sync_bn_comm2: bad argument"); return MMVAE_ERR_ARG; }
  n->net->set_sync_bn_comm(cm ? cm->c : nullptr, cs ? cs->c : nullptr);
  return MMVAE_OK;
}

// ---- PixelCNN (reference model.py:212-255)
struct mmvae_pixelcnn { PixelNet* net; };
int mmvae_pixelcnn_create(mmvae_pixelcnn** out, int in_channels, int intermediate_channels, int out_channels, int layers, int dtype) {
  if (!out) { set_error("pixelcnn_create: bad argument"); return MMVAE_ERR_ARG; }
  if (dtype != MMVAE_F32 && dtype != MMVAE_BF16) { set_error("pixelcnn_create: dtype must be f32 or bf16"); return MMVAE_ERR_UNSUPPORTED; }
  if (in_channels < 1 || in_channels > 16 || out_channels < 1 || out_channels > 16 || layers < 2 || layers > 16 || intermediate_channels < 16 ||
      intermediate_channels > 256 || intermediate_channels % 16) {
    set_error("pixelcnn_create: unsupported shape (in %d, intermediate %d, out %d, layers %d): in / out <= 16, intermediate a multiple of 16 in [16, 256], "
              "2 <= layers <= 16", in_channels, intermediate_channels, out_channels, layers);
    return MMVAE_ERR_UNSUPPORTED;
  }
  mmvae_pixelcnn* h = new (std::nothrow) mmvae_pixelcnn;
  if (!h) return MMVAE_ERR_ARG;
  h->net = new PixelNet(in_channels, intermediate_channels, out_channels, layers, dtype == MMVAE_F32 ? DT_F32 : DT_BF16);
  *out = h;
  return MMVAE_OK;
}
void mmvae_pixelcnn_destroy(mmvae_pixelcnn* h) { if (h) { delete h->net; delete h; } }
int64_t mmvae_pixelcnn_num_params(mmvae_pixelcnn* h) { return h ? (int64_t)h->net->n_params : -1; }
size_t mmvae_pixelcnn_workspace_bytes(mmvae_pixelcnn* h, int N, int S) { return (h && N > 0 && S > 0) ? h->net->workspace_bytes(N, S) : 0; }
int mmvae_pixelcnn_fwd(mmvae_pixelcnn* h, int N, int S, const float* x, const float* params, void* ws, size_t wsb, float* out, void* st) {
  if (!h || N <= 0 || S <= 0 || !x || !params || !ws || !out) { set_error("pixelcnn_fwd: bad argument"); return MMVAE_ERR_ARG; }
  return h->net->forward(N, S, x, params, ws, wsb, out, S_(st));
}
int mmvae_pixelcnn_bwd(mmvae_pixelcnn* h, int N, int S, const float* x, const float* d_out, const float* params, float* grads, void* ws, size_t wsb,
                       float* d_x, void* st) {
  if (!h || N <= 0 || S <= 0 || !x || !d_out || !params || !grads || !ws) { set_error("pixelcnn_bwd: bad argument"); return MMVAE_ERR_ARG; }
  return h->net->backward(N, S, x, d_out, params, grads, ws, wsb, d_x, S_(st));
}

// ---- latent / loss
int mmvae_rsample_fwd(const float* mu, const float* lv, const float* eps, float* enc, int64_t n, void* st) {
  return launch_rsample_fwd(DT_F32, mu, lv, eps, enc, nullptr, (long)n, S(st));
}
int mmvae_rsample_bwd(const float* d_enc, const float* lv, const float* eps, float* d_mu, float* d_lv, int64_t n, void* st) {
  return launch_rsample_bwd(d_enc, lv, eps, d_mu, d_lv, (long)n, S(st));
}
int mmvae_kl_fwd(const float* mu, const float* lv, int64_t n, double* acc, void* st) { return launch_kl_fwd(mu, lv, (long)n, acc, S(st)); }
int mmvae_kl_bwd(const float* mu, const float* lv, float coef, const float* gscale, float* d_mu, float* d_lv, int64_t n, void* st) {
  return launch_kl_bwd(mu, lv, coef, gscale, d_mu, d_lv, (long)n, S(st));
}
int mmvae_gauss_nll_fwd(const float* r, const float* t, int64_t n, float sigma, double* acc, void* st) {
  if (!(sigma > 0.f)) { set_error("gauss_nll: sigma must be > 0"); return MMVAE_ERR_ARG; }
  return launch_gauss_nll_fwd(r, t, (long)n, sigma, acc, S(st));
}
int mmvae_gauss_nll_bwd(const float* r, const float* t, int64_t n, float sigma, float coef, const float* gscale, float* d_r, void* st) {
  if (!(sigma > 0.f)) { set_error("gauss_nll: sigma must be > 0"); return MMVAE_ERR_ARG; }
  return launch_gauss_nll_bwd(r, t, (long)n, sigma, coef, gscale, d_r, S(st));
}
int mmvae_ce_fwd(const float* r, const int64_t* t, const float* w, int N, int Q, int HW, double* acc, void* st) {
  return launch_ce_fwd(r, reinterpret_cast<const long long*>(t), w, N, Q, HW, acc, S(st));
}
int mmvae_ce_bwd(const float* r, const int64_t* t, const float* w, int N, int Q, int HW, float coef, const float* gscale, float* d_r,
                 void* st) {
  return launch_ce_bwd(r, reinterpret_cast<const long long*>(t), w, N, Q, HW, coef, gscale, d_r, S(st));
}
int mmvae_mmd_fwd(const float* x, const float* y, int n, int d, float* scratch, double* acc, void* st) {
  return scratch ? launch_mmd_fwd_mfma(x, y, n, d, scratch, acc, S(st)) : launch_mmd_fwd(x, y, n, d, acc, S(st));
}
int mmvae_mmd_bwd(const float* x, const float* y, int n, int d, float coef, const float* gscale, float* d_y, void* st) {
  return launch_mmd_bwd(x, y, n, d, coef, gscale, d_y, S(st));
}
int mmvae_rbf_kernel(const float* x, const float* y, int n, int m, int d, float* out, void* st) {
  if (!x || !y || !out) { set_error("rbf_kernel: bad argument"); return MMVAE_ERR_ARG; }
  return launch_rbf_matrix(x, y, n, m, d, out, S(st));
}
int mmvae_loss_finish(const double* acc, float* out, float nll, float kl_coef, float mmd_coef, float n, void* st) {
  return launch_loss_finish(acc, out, nll, kl_coef, mmd_coef, n, S(st));
}

// ---- plumbing
int mmvae_normalise_labels(const int64_t* labels, int64_t n, float mean, float stdv, float* image, void* st) {
  return launch_normalise(DT_F32, labels, 8, (long)n, mean, stdv, nullptr, image, S(st));
}
int mmvae_quantise_normalise(const uint8_t* frames, int64_t n, const float* centres, int q, float mean, float stdv, int64_t* labels,
                             float* image, void* st) {
  return launch_quantise_normalise(frames, (long)n, centres, q, mean, stdv, reinterpret_cast<long long*>(labels), image, S(st));
}
int mmvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1,
                    float bc2_sqrt, float grad_scale, void* st) {
  AdamArgs a{p, g, m, v, (long)n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale};
  return launch_adam(a, S(st));
}

int mmvae_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                        double* step_dev, float grad_scale, void* st) {
  if (!step_dev) { set_error("adam_step_dev: step counter required"); return MMVAE_ERR_ARG; }
  AdamArgs a{p, g, m, v, (long)n, lr, b1, b2, eps, wd, 1.f, 1.f, grad_scale};
  return launch_adam_dev(a, step_dev, S(st));
}

// ---- single ops
static inline ConvGeom geom_for(int transposed, int Cin, int Cout, int k, int s, int p) {
  // Conv2d weight (Cout,Cin,k,k): D0=Cout (small side = y), D1=Cin.  ConvT weight (Cin,Cout,k,k): D0=Cin (small side = x), D1=Cout.
  return transposed ? ConvGeom{Cin, Cout, k, s, p} : ConvGeom{Cout, Cin, k, s, p};
}
static inline long numel_w(int Cin, int Cout, int k) { return (long)Cin * Cout * k * k; }
static inline int out_size(int transposed, int H, int k, int s, int p) { return transposed ? (H - 1) * s - 2 * p + k : conv_down_size(H, k, s, p); }

int mmvae_conv2d_fwd(int dt, int transposed, const void* x, const float* w, void* y, int N, int H, int W, int Cin, int Cout, int k,
                     int s, int p, const float* ps, const float* pb, int relu, float* stats, void* scratch, void* st) {
  const ConvGeom g = geom_for(transposed, Cin, Cout, k, s, p);
  const int Ho = out_size(transposed, H, k, s, p), Wo = out_size(transposed, W, k, s, p);
  // weight == NULL: `scratch` still holds the packed weights of an earlier call with the same geometry (pack once, run many)
  SecondSrc q;
  if (!transposed && conv3_stream_ok(dt, Cin, Cout, k, s, p, H, W)) {      // the first encoder block's 3x3 convs: per-wave stream
    if (w) { int rc = op_pack_down(dt, g, w, scratch, S(st)); if (rc < 0) return rc; }
    return launch_conv3_stream(dt, s, x, scratch, nullptr, y, nullptr, ps, pb, relu, stats, nullptr, N, Ho, S(st));
  }
  if (!transposed) {
    q.wfrag = op_frag_down(dt, g, H, W);
    if (w) { int rc = op_pack_down(dt, g, w, scratch, S(st), 1.f, 0, q.wfrag); if (rc < 0) return rc; }
    return op_run_down(dt, dt, g, scratch, N, x, H, W, y, Ho, Wo, ps, pb, relu, stats, 0, S(st), q);
  }
  if (convT4_stream_ok(dt, Cin, Cout, k, s, p, H, W)) {                   // the decoder's widest ConvT layers: per-wave stream
    if (w) { int rc = op_pack_up(dt, g, w, scratch, S(st)); if (rc < 0) return rc; }
    return launch_convT4_stream(dt, x, scratch, y, ps, pb, relu, stats, N, H, S(st));
  }
  q.wfrag = op_frag_up(dt, g, Ho, Wo);
  if (w) { int rc = op_pack_up(dt, g, w, scratch, S(st), 1.f, 0, q.wfrag); if (rc < 0) return rc; }
  return op_run_up(dt, g, scratch, N, x, H, W, y, Ho, Wo, ps, pb, relu, stats, 0, S(st), q);
}
int mmvae_conv2d_dgrad(int dt, int transposed, const void* dy, const float* w, void* dx, int N, int H, int W, int Cin, int Cout, int k,
                       int s, int p, void* scratch, void* st) {
  const ConvGeom g = geom_for(transposed, Cin, Cout, k, s, p);
  const int Ho = out_size(transposed, H, k, s, p), Wo = out_size(transposed, W, k, s, p);
  SecondSrc q;
  if (!transposed && s == 1 && conv3_stream_ok(dt, Cin, Cout, k, s, p, H, W)) {
    // encoder.layer1.conv2's shape: the data gradient as a forward conv over dy with the weights packed [cin][flipped tap][cout]
    PackArgs pf; std::memset(&pf, 0, sizeof(pf));
    pf.src = w; pf.dst = scratch; pf.cols = Cin; pf.K = Cout; pf.ntaps = 9; pf.s_col = 9; pf.s_k = Cin * 9; pf.scale = 1.f;
    for (int t = 0; t < 9; ++t) pf.tap_off[t] = 8 - t;
    int rc = launch_pack(dt, pf, S(st)); if (rc < 0) return rc;
    rc = launch_conv3_stream(dt, 1, dy, scratch, nullptr, dx, nullptr, nullptr, nullptr, 0, nullptr, nullptr, N, H, S(st));
    return rc < 0 ? rc : MMVAE_OK;
  }
  if (!transposed) {
    q.wfrag = op_frag_up(dt, g, H, W);
    int rc = op_pack_up(dt, g, w, scratch, S(st), 1.f, 0, q.wfrag); if (rc < 0) return rc;
    return op_run_up(dt, g, scratch, N, dy, Ho, Wo, dx, H, W, nullptr, nullptr, 0, nullptr, 0, S(st), q);
  }
  q.wfrag = op_frag_down(dt, g, Ho, Wo);
  int rc = op_pack_down(dt, g, w, scratch, S(st), 1.f, 0, q.wfrag); if (rc < 0) return rc;
  return op_run_down(dt, dt, g, scratch, N, dy, Ho, Wo, dx, H, W, nullptr, nullptr, 0, nullptr, 0, S(st), q);
}
int mmvae_conv2d_wgrad(int dt, int transposed, const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int Cout, int k,
                       int s, int p, const float* ps, const float* pb, int relu, void* scratch, void* st) {
  const ConvGeom g = geom_for(transposed, Cin, Cout, k, s, p);
  const int Ho = out_size(transposed, H, k, s, p), Wo = out_size(transposed, W, k, s, p);
  if (!scratch) { set_error("conv2d_wgrad: scratch (MMVAE_WGRAD_SCRATCH_BYTES) required"); return MMVAE_ERR_ARG; }
  static_assert(MMVAE_WGRAD_SCRATCH_BYTES == kWgradScratchBytes, "public scratch size out of sync");
  float* sc = static_cast<float*>(scratch);
  if (!transposed) return op_run_wgrad(dt, g, N, dy, Ho, Wo, nullptr, nullptr, 0, x, H, W, ps, pb, relu, dw, S(st), sc);
  return op_run_wgrad(dt, g, N, x, H, W, ps, pb, relu, dy, Ho, Wo, nullptr, nullptr, 0, dw, S(st), sc);
}
int mmvae_conv2d_wgrad_pair(int dt, const void* x, const void* dy, const void* dy_sc, float* dw, float* dw_sc, int N, int H, int W, int Cin, int Cout,
                            const float* ps, const float* pb, int relu, void* scratch, void* st) {
  if (!scratch || !x || !dy || !dy_sc || !dw || !dw_sc || N < 1) { set_error("conv2d_wgrad_pair: bad argument"); return MMVAE_ERR_ARG; }
  const ConvGeom g = geom_for(0, Cin, Cout, 3, 2, 1), gs = geom_for(0, Cin, Cout, 1, 2, 0);
  const int Ho = out_size(0, H, 3, 2, 1), Wo = out_size(0, W, 3, 2, 1);
  const int rc = op_run_wgrad_pair(dt, g, gs, N, dy, dy_sc, Ho, Wo, x, H, W, ps, pb, relu, dw, dw_sc, S(st), static_cast<float*>(scratch), 1.f, 1.f);
  if (rc == 0) { set_error("conv2d_wgrad_pair: shape not supported (bf16, 32 -> 32 channels, 32x32 -> 16x16)"); return MMVAE_ERR_UNSUPPORTED; }
  return rc < 0 ? rc : MMVAE_OK;
}
int mmvae_convT_bwd_fused(int dt, const void* x, const void* dy, const float* w, float* dw, void* dx, int N, int H, int W, int Cin, int Cout, int k,
                          int s, int p, const float* ps, const float* pb, int relu, const void* x2, const float* w2, float* dw2, void* scratch,
                          void* wscratch, void* st) {
  const ConvGeom g = geom_for(1, Cin, Cout, k, s, p);
  const int Ho = out_size(1, H, k, s, p), Wo = out_size(1, W, k, s, p);
  if (!scratch || !wscratch) { set_error("convT_bwd_fused: scratch buffers required"); return MMVAE_ERR_ARG; }
  if (!op_bwd_fusable(dt, g, N, H, W, Ho, Wo)) { set_error("convT_bwd_fused: shape not supported (bf16, 16 / 32 -> 16 channels, k4 s2 p1, 32x32 -> 64x64 or 16x16 -> 32x32)"); return MMVAE_ERR_UNSUPPORTED; }
  char* sc = static_cast<char*>(scratch);
  int rc = op_pack_down(dt, g, w, sc, S(st)); if (rc < 0) return rc;
  const void* w2p = nullptr;
  if (x2 && w2) {
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = w2; pa.dst = sc + (size_t)Cin * Cout * k * k * dtype_size(dt); pa.cols = Cin; pa.K = 16; pa.ntaps = 1; pa.s_col = 16; pa.s_k = 1; pa.scale = 1.f; pa.tap_off[0] = 0;
    rc = launch_pack(dt, pa, S(st)); if (rc < 0) return rc;
    w2p = pa.dst;
  }
  if (dw2 && !w2p) { set_error("convT_bwd_fused: dw2 needs x2 and w2"); return MMVAE_ERR_ARG; }
  rc = op_run_bwd_fused(dt, g, N, x, H, W, ps, pb, relu, dy, Ho, Wo, sc, dx, w2p ? x2 : nullptr, w2p, dw, S(st), static_cast<float*>(wscratch), 1.f,
                        nullptr, dw2, 1.f);
  if (rc == 0) { set_error("convT_bwd_fused: not taken"); return MMVAE_ERR_UNSUPPORTED; }
  return rc < 0 ? rc : MMVAE_OK;
}
// ---- op-level BatchNorm / stem (compositions of the kernels the network orchestrator launches)
static_assert(MMVAE_BN_SCRATCH_BYTES >= (1024u * 3 * 512 + 8 * 512) * 4, "BatchNorm scratch: 1024 partial rows x 3 x C<=512 + coefficient rows");
int mmvae_batchnorm_fwd(int dt, const void* y, int64_t npix, int C, const float* gamma, const float* beta, float* rm, float* rv, int64_t* nbt,
                        float momentum, float eps, int relu, void* out, float* save_mean, float* save_istd, void* scratch, void* st) {
  if (!scratch || !save_mean || !save_istd || C < 1 || C > 512 || npix < 1) { set_error("batchnorm_fwd: bad arguments (C <= 512, scratch and save_* required)"); return MMVAE_ERR_ARG; }
  float* part = static_cast<float*>(scratch);
  float* coef = part + 1024L * 3 * 512;              // scale, shift
  const int np = launch_chan_stats_nhwc(dt, y, npix, C, part, S(st));
  if (np < 0) return np;
  BnFinalizeArgs f;
  f.partials = part; f.nparts = np; f.C = C; f.count = (double)npix; f.gamma = gamma; f.beta = beta; f.running_mean = rm; f.running_var = rv;
  f.nbt = reinterpret_cast<long long*>(nbt); f.mean = save_mean; f.istd = save_istd; f.scale = coef; f.shift = coef + C; f.momentum = momentum; f.eps = eps;
  int rc = launch_bn_finalize(f, S(st));
  if (rc < 0) return rc;
  return launch_affine_act(dt, y, coef, coef + C, relu, out, npix, C, S(st));
}
int mmvae_batchnorm_bwd(int dt, const void* dout, const void* y, const void* out, int64_t npix, int C, const float* gamma, const float* save_mean,
                        const float* save_istd, void* dy, float* dgamma, float* dbeta, void* scratch, void* st) {
  if (!scratch || C < 1 || C > 512 || npix < 1) { set_error("batchnorm_bwd: bad arguments (C <= 512, scratch required)"); return MMVAE_ERR_ARG; }
  float* part = static_cast<float*>(scratch);
  float* coef = part + 1024L * 3 * 512;              // A, B, C
  const int np = launch_bn_bwd_reduce(dt, dout, out, nullptr, nullptr, y, nullptr, npix, C, part, S(st));
  if (np < 0) return np;
  BnBwdFinalizeArgs f; std::memset(&f, 0, sizeof(f));
  f.partials = part; f.nparts = np; f.C = C; f.which = 0; f.ny = 1; f.count = (double)npix; f.gamma = gamma; f.mean = save_mean; f.istd = save_istd;
  f.dgamma = dgamma; f.dbeta = dbeta; f.coefA = coef; f.coefB = coef + C; f.coefC = coef + 2 * C;
  int rc = launch_bn_bwd_finalize(f, S(st));
  if (rc < 0) return rc;
  return launch_bn_bwd_apply(dt, dout, out, nullptr, nullptr, y, coef, coef + C, coef + 2 * C, dy, nullptr, nullptr, nullptr, nullptr, nullptr, npix, C, S(st));
}
int mmvae_stem_fwd(int dt, const void* x, const float* w, void* y, int N, int Sz, float* stats, void* scratch, void* st) {
  if (!scratch || N < 1 || Sz < 9 || Sz > 64) { set_error("stem_fwd: bad arguments"); return MMVAE_ERR_ARG; }
  if (stem_fwd_stream_ok(dt, Sz)) return launch_stem_fwd_stream(dt, x, w, y, stats, N, Sz, S(st));
  const int H1 = (Sz + 4 - 5) / 2 + 1;
  const int cpad = dt == DT_F32 ? 4 : 8;
  PackArgs pa; std::memset(&pa, 0, sizeof(pa));
  pa.src = w; pa.dst = scratch;
  pa.cols = 32; pa.K = cpad; pa.K_valid = 1; pa.ntaps = 25; pa.s_col = 25; pa.s_k = 25; pa.scale = 1.f;
  for (int t = 0; t < 25; ++t) pa.tap_off[t] = t;
  int rc = launch_pack(dt, pa, S(st));
  if (rc < 0) return rc;
  GatherArgs a; std::memset(&a, 0, sizeof(a));
  a.x = x; a.w = scratch; a.y = y; a.stats = stats;
  a.x_planar = 2; a.x_planes = 1;
  a.N = N; a.Hi = Sz; a.Wi = Sz; a.Cin = cpad; a.Ho = H1; a.Wo = H1; a.Cout = 32; a.SI = 2; a.SO = 1;
  a.nphase = 1; a.phases[0] = Phase{0, 0, H1, H1, 25, 0, 0};
  for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw) a.taps[kh * 5 + kw] = Tap{kh - 2, kw - 2};
  return launch_gather_gemm(dt, dt, a, S(st));
}
int mmvae_stem_bwd(int dt, const void* g, const void* y0, const void* x, const float* w, const float* gamma, const float* bn_scale,
                   const float* bn_shift, const float* save_mean, const float* save_istd, float* dw, float* dgamma, float* dbeta, int N, int Sz,
                   void* scratch, void* st) {
  if (!scratch || N < 1) { set_error("stem_bwd: bad arguments"); return MMVAE_ERR_ARG; }
  if (!stem_bwd_fusable(Sz)) { set_error("stem_bwd: image size %d unsupported (64, 32, 16)", Sz); return MMVAE_ERR_UNSUPPORTED; }
  const int H1 = (Sz + 4 - 5) / 2 + 1;
  // carve: R (1024 doubles) | gram partials (1024 x part floats) | backward partials (the rest)
  double* R = static_cast<double*>(scratch);
  float* gram = reinterpret_cast<float*>(R + 1024);
  const long gram_floats = 1024L * stem_bwd_part_floats();
  float* part = gram + gram_floats;
  const long part_floats = (long)(MMVAE_BN_SCRATCH_BYTES - 1024 * 8) / 4 - gram_floats;
  int rc = launch_stem_gram(dt, x, gram, gram_floats, R, N, Sz, H1, H1, S(st));
  if (rc < 0) return rc;
  const int np = launch_stem_bwd(dt, g, y0, x, bn_scale, bn_shift, part, part_floats, N, Sz, H1, H1, S(st));
  if (np < 0) return np;
  return launch_stem_bwd_finalize(part, np, R, w, nullptr, (double)N * H1 * H1, gamma, save_mean, save_istd, dgamma, dbeta, dw, S(st));
}
int mmvae_tail_join_fwd(int dt, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs,
                        const float* w, const float* bias, float* r_raw, float* stats, int N, int H, int W, void* st) {
  return launch_tail_join_fwd(dt, y2, s2, b2, ys, ss, bs, w, bias, r_raw, stats, N, H, W, S(st));
}
int mmvae_tail_join_fwd_stream(int dt, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs,
                               const float* w, const float* bias, float* r_raw, float* stats, int N, int H, int W, void* st) {
  return launch_tail_fwd_stream(dt, y2, s2, b2, ys, ss, bs, w, bias, r_raw, stats, N, H, W, S(st));
}
int mmvae_tail_join_bwd_reduce(int dt, const float* d_raw, const float* w, int oc, const void* y2, const float* s2, const float* b2,
                               const void* ys, const float* ss, const float* bs, float* partials, float* wpartials, int N, int H, int W,
                               void* st) {
  return launch_tail_join_bwd_reduce(dt, d_raw, w, oc, N, H, W, s2, b2, ss, bs, y2, ys, partials, S(st), wpartials);
}
int mmvae_tail_join_bwd_apply(int dt, const float* d_raw, const float* w, int oc, const void* y2, const float* s2, const float* b2,
                              const void* ys, const float* ss, const float* bs, const float* A2, const float* B2, const float* C2,
                              const float* As, const float* Bs, const float* Cs, void* dy2, void* dys, int N, int H, int W, void* st) {
  return launch_tail_join_bwd_apply(dt, d_raw, w, oc, N, H, W, s2, b2, ss, bs, y2, A2, B2, C2, dy2, ys, As, Bs, Cs, dys, S(st));
}
int mmvae_upblock_bwd_fused(const float* d_raw, const float* tw, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss,
                            const float* bs, const float* A2, const float* B2, const float* C2, const float* As, const float* Bs, const float* Cs,
                            const void* y1, const float* s1, const float* b1, const float* w2, float* dw2, void* da1, float* bn1_sums,
                            const void* xin, const float* sx, const float* bx, const float* wu, float* dwu, void* gin, int N, void* scratch, void* st) {
  if (!scratch || !d_raw || !tw || !y2 || !ys || !y1 || !xin || !w2 || !wu || !dw2 || !dwu || !da1 || !gin || !bn1_sums || N < 1 || ((sx != nullptr) != (bx != nullptr))) {
    set_error("upblock_bwd_fused: bad argument"); return MMVAE_ERR_ARG;
  }
  if (!join_bwd_stream_ok(DT_BF16, 1, 16, 32, 64)) { set_error("upblock_bwd_fused: disabled (MMVAE_JOIN_BWD_STREAM=0)"); return MMVAE_ERR_UNSUPPORTED; }
  char* sc = static_cast<char*>(scratch);
  const ConvGeom g = geom_for(1, 16, 16, 4, 2, 1);
  int rc = op_pack_down(DT_BF16, g, w2, sc, S(st)); if (rc < 0) return rc;
  rc = op_pack_down(DT_BF16, g, wu, sc + 8192, S(st)); if (rc < 0) return rc;
  float* parts = reinterpret_cast<float*>(sc + 16384);
  JoinBwdLaunch L;
  L.d_raw = d_raw; L.w_tail = tw; L.y2 = y2; L.ys = ys; L.ms2 = s2; L.mb2 = b2; L.mss = ss; L.mbs = bs;
  L.A2 = A2; L.B2 = B2; L.C2 = C2; L.As = As; L.Bs = Bs; L.Cs = Cs;
  L.y1 = y1; L.p1s = s1; L.p1b = b1; L.wd2 = sc; L.da1 = da1; L.part2 = parts; L.bn_part = parts + 2L * 1024 * 4096;
  L.xin = xin; L.pxs = sx; L.pxb = bx; L.wds = sc + 8192; L.gin = gin; L.parts = parts + 1024L * 4096;
  L.N = N;
  const int nb = launch_join_bwd_stream(L, S(st));
  if (nb < 0) return nb;
  for (int k = 0; k < 2; ++k) {
    WgradReduceArgs u; std::memset(&u, 0, sizeof(u));
    u.part = k ? L.parts : L.part2; u.dW = k ? dwu : dw2; u.Ca = 16; u.Cb = 16; u.ntaps = 16; u.nparts = nb;
    u.Ca_valid = 16; u.Cb_valid = 16; u.sA = 16 * 16; u.sB = 16; u.scale = 1.f;
    for (int t = 0; t < 16; ++t) u.tap_off[t] = t;
    rc = launch_wgrad_reduce(u, S(st)); if (rc < 0) return rc;
  }
  return launch_partial_rowsum(L.bn_part, nb, 32, bn1_sums, S(st));
}
int mmvae_upblock_tail_fwd(const void* y1, const float* s1, const float* b1, const float* w2, const void* xin, const float* sx, const float* bx,
                           const float* wu, const float* s2, const float* b2, const float* ss, const float* bs, const float* tw, const float* bias,
                           float* r_raw, float* stats, int N, void* scratch, void* st) {
  if (!scratch || !y1 || !s1 || !b1 || !w2 || !xin || !wu || !s2 || !b2 || !ss || !bs || !tw || !r_raw || N < 1 || ((sx != nullptr) != (bx != nullptr))) {
    set_error("upblock_tail_fwd: bad argument"); return MMVAE_ERR_ARG;
  }
  if (!up5_tail_fwd_ok(DT_BF16, 1, 16, 16, 32, 64)) { set_error("upblock_tail_fwd: disabled (MMVAE_TAIL_RECOMPUTE=0)"); return MMVAE_ERR_UNSUPPORTED; }
  char* sc = static_cast<char*>(scratch);
  const ConvGeom g = geom_for(1, 16, 16, 4, 2, 1);
  int rc = op_pack_up(DT_BF16, g, w2, sc, S(st)); if (rc < 0) return rc;
  rc = op_pack_up(DT_BF16, g, wu, sc + 8192, S(st)); if (rc < 0) return rc;
  return launch_up5_tail_fwd(y1, s1, b1, sc, xin, sx, bx, sc + 8192, s2, b2, ss, bs, tw, bias, r_raw, stats, N, S(st));
}
int mmvae_conv1x1_bwd_fused(const void* da1, const void* y1, const float* s1, const float* b1, const float* A1, const float* B1, const float* C1,
                            const void* xin, const float* sx, const float* bx, const float* w1, float* dw1, void* gin, int64_t rows, void* scratch,
                            void* st) {
  if (!scratch || !da1 || !y1 || !s1 || !b1 || !A1 || !B1 || !C1 || !xin || !w1 || !dw1 || !gin || rows < 1 || ((sx != nullptr) != (bx != nullptr))) {
    set_error("conv1x1_bwd_fused: bad argument"); return MMVAE_ERR_ARG;
  }
  char* sc = static_cast<char*>(scratch);
  const ConvGeom g = geom_for(0, 16, 16, 1, 1, 0);
  int rc = op_pack_up(DT_BF16, g, w1, sc, S(st)); if (rc < 0) return rc;
  Conv1BwdLaunch C;
  C.da1 = da1; C.y1 = y1; C.ms = s1; C.mb = b1; C.A = A1; C.B = B1; C.C = C1; C.xin = xin; C.pxs = sx; C.pxb = bx; C.w1u = sc; C.gin = gin;
  C.part = reinterpret_cast<float*>(sc + 4096); C.nrows = (long)rows;
  const int nb = launch_conv1_bwd_stream(C, S(st));
  if (nb < 0) return nb;
  WgradReduceArgs u; std::memset(&u, 0, sizeof(u));
  u.part = C.part; u.dW = dw1; u.Ca = 16; u.Cb = 16; u.ntaps = 1; u.nparts = nb; u.Ca_valid = 16; u.Cb_valid = 16; u.sA = 16; u.sB = 1; u.scale = 1.f;
  return launch_wgrad_reduce(u, S(st));
}
int mmvae_join_conv1x1_fwd(const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs, const float* w1, int C,
                           void* out, void* y1, float* stats, int64_t npix, void* scratch, void* st) {
  if (!y2 || !s2 || !b2 || !ys || !ss || !bs || !w1 || !out || !y1 || !scratch) { set_error("join_conv1x1_fwd: bad argument"); return MMVAE_ERR_ARG; }
  if (!join_conv1_fwd_ok(DT_BF16, C, 16, (long)npix)) { set_error("join_conv1x1_fwd: C=%d npix=%ld unsupported (16 / 32 channels, whole 32-pixel steps)", C, (long)npix); return MMVAE_ERR_UNSUPPORTED; }
  const ConvGeom g = geom_for(0, C, 16, 1, 1, 0);
  const int rc = op_pack_down(DT_BF16, g, w1, scratch, S(st)); if (rc < 0) return rc;
  return launch_join_conv1_fwd(C, y2, s2, b2, ys, ss, bs, scratch, out, y1, stats, (long)npix, S(st));
}
int mmvae_convert(int di, int dout, const void* in, void* out, int64_t n, void* st) { return launch_convert(di, dout, in, out, (long)n, S(st)); }

}  // extern "C"
