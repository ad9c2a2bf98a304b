// Host-side launch API of the HIP kernels (internal; the public C ABI is include/mmvae.h).
#pragma once
#include "common.hpp"

namespace mmvae {

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };
inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

// ---------------------------------------------------------------- gather GEMM
// y[n, hq*SO+ph, wq*SO+pw, co] (+)= bias[co] + sum_{t<ntaps} sum_{ci}
//        pro(x[n, hq*SI+dh[t], wq*SI+dw[t], ci]) * w[co][t][ci]
// with zero for out-of-range input coordinates (zero padding is applied AFTER pro()).
// Covers Conv2d forward (SI=stride, SO=1, one launch) and every stride-phase of a
// transposed convolution / data-gradient (SI=1, SO=stride, one launch per phase).
struct Tap { int dh, dw; };
constexpr int kMaxTaps = 25;     // over all phases of one launch (5x5 stem)
constexpr int kMaxPhases = 4;
struct Phase { int ph, pw, Hq, Wq, ntaps, tap0; long w_off; /* element offset of this phase's [Cout][ntaps*Cin] matrix */ };
struct GatherArgs {
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  const float* bias;
  float* stats;            // [gridDim.z*gridDim.x][2][Cout] partial (sum, sumsq) of the f32 results, or null
  int accumulate;          // y += result
  int N, Hi, Wi, Cin, Ho, Wo, Cout;
  int SI, SO;
  int nphase; Phase phases[kMaxPhases]; Tap taps[kMaxTaps];
  int cin_vecs;
  // optional boundary layouts (patch-tile kernels only):
  int x_planar, x_planes;  // 1: x is planar f32 [N][x_planes][Hi][Wi], 2: planar T; staged as Cin=16 channels, zero padded
  int dbg;                 // developer switches (MMVAE_DBG): bit0 skip global loads, bit1 skip stores, bit2 skip MFMA
  int y_planes;            // >0: y is NCHW f32 [N][y_planes][Ho][Wo] (Cout = 16 padded GEMM rows; stats rows have y_planes channels)
  // optional SECOND source (patch-tile kernel only; launch_gather_gemm returns MMVAE_ERR_UNSUPPORTED when it cannot take it):
  // y[q-pixel of phase (x2_ph, x2_pw)] += sum_c2 x2[n, hq, wq, c2] * w2[co][c2] -- a 1x1 convolution of a tensor that lives on the
  // q grid itself ([N][Hq][Wq][Cin2] of T, no halo).  Merges the two data gradients that meet at a residual block's input
  // (3x3 / 4x4 main path + 1x1 shortcut) into one kernel: no read-modify-write of the sum.
  const void* x2; const void* w2; int Cin2, x2_ph, x2_pw;
  int fp8;                 // 1: w holds e4m3 bytes (same [cout][tap][cin] order); only the deep-layer kernel takes it (else an error)
  int wfrag, wfrag2;       // 1: w (w2) is packed fragment-major (PackArgs::frag); only deep2_conv_kernel / pos_conv_kernel read it (else an error)
  int gk, gs, gp, gup;     // the conv-like weight's (k, s, p) and the form (1: down, 2: up) when the launch comes from op_run_down / op_run_up (0: unknown)
};
// Shapes deep2_conv_kernel takes (conv_deep2.inc): the layers whose weights are packed fragment-major.  q grid Hq x Wq per phase
// and image, input map Hi x Wi, ntaps over all phases.
bool deep2_shape_ok(int dt, int Cin, int Cout, int Hq, int Wq, int Hi, int Wi, int ntaps_all, int fp8 = 0);   // fp8: e4m3 weights / LDS (bf16 storage)
// out_dt: dtype of y (may be DT_F32 while x/w are bf16).  Returns the number of stats partial rows (>0) or an error (<0).
int launch_gather_gemm(int dt, int out_dt, GatherArgs a, hipStream_t s);
constexpr int kGatherMaxGridX = 1024;

// ---------------------------------------------------------------- wgrad
// dW[a*sA + b*sB + tap_off[t]] += scale * sum_{n,hp,wp} proP(P[n,hp,wp,a]) * proG(G[n, hp*stride-pad+kh_t, wp*stride-pad+kw_t, b])
struct WgradArgs {
  const void* P; const void* G; float* dW;
  const float* proP_scale; const float* proP_shift; int proP_relu;
  const float* proG_scale; const float* proG_shift; int proG_relu;
  int N, Hp, Wp, Ca, Hg, Wg, Cb;
  int Cb_valid;            // b >= Cb_valid is computed but not written (0 -> Cb)
  int Ca_valid;            // likewise for a (0 -> Ca)
  int P_planar, P_planes;  // 1: P is planar f32 [N][P_planes][Hp][Wp] (patch-tile kernel only; Ca = 16, Ca_valid = P_planes)
  int G_planar;            // 1: G is planar T [N][1][Hg][Wg] (patch-tile kernel only; Cb = 16, Cb_valid = 1)
  float* scratch;          // optional, kWgradScratchBytes: per-block partial images [part][tap][a][b], summed by wgrad_reduce_kernel
  int stride, pad, ksz;
  int sA, sB; int ntaps; int tap_off[25];
  float scale;
  int M, pix_per_block;
  int TA16, TB16, TG;      // tile config chosen by the launcher
  struct WgradReduceArgs* defer;   // optional (stream kernels only): the reduce is NOT launched, its arguments are left here (nparts = 0 when the
                                   // launch took another kernel and reduced at once) -- the caller launches it later, off the busy phase
};
int launch_wgrad(int dt, WgradArgs a, hipStream_t s);
int try_wgrad_pos(int dt, const WgradArgs& a, hipStream_t s);     // conv_wpos.hip: both maps 1x1 .. 4x4 (1 = taken, 0 = not this kernel's shape)
bool wgrad_stream_shape(int dt, const WgradArgs& a);
int try_wgrad_stream(int dt, const WgradArgs& a, hipStream_t s);   // conv_wstream.hip: 1 = taken, 0 = not this kernel's shape, <0 error
int try_wgrad_stream_pair(int dt, const WgradArgs& a, const void* P2, float* dW2, float scale2, hipStream_t s);
// weight gradient + data gradient (w.r.t. P) of a 16 -> 16 channel k4 s2 layer in one pass over G (conv_wstream.hip)
bool dgrad_wgrad_stream_shape(int dt, const WgradArgs& a);
// dW2 (optional, with x2): the 1x1 conv's own weight gradient [16][Ca] += scale2 * x2^T (x) pro(P), from the same pass
// dy of a BatchNorm behind a residual join, evaluated by the consumer's loader: dy = A[c] * g + B[c] * y + C[c], g = the join's output gradient
// already masked by its ReLU, y = the branch's pre-BatchNorm output (conv_wstream.hip, JG)
struct JoinGrad { const void* y; const float* A; const float* B; const float* C; };
bool dgrad_wgrad_stream_jg_shape(int dt, const WgradArgs& a, bool has_x2, bool bn_sums);
int try_dgrad_wgrad_stream(int dt, const WgradArgs& a, const void* wd, void* dx, const void* x2, const void* w2, float* bn_part, hipStream_t s,
                           float* dW2 = nullptr, float scale2 = 1.f, const JoinGrad* jg = nullptr);

// ---- last up-block backward in one pass (conv_joinbwd.hip): the join's BatchNorm backward (incoming gradient recomputed from the
// one-plane reconstruction gradient), both ConvTranspose2d weight gradients and data gradients, bn1's backward sums -- dy2 / dys are
// never stored.  bf16, 16 channels, 32x32 -> 64x64, one output plane.  Coefficients: ms* / mb* forward scale / shift, A / B / C the
// BatchNorm-backward coefficients of the two branch BatchNorms (bn_bwd_finalize).
struct JoinBwdLaunch {
  const float* d_raw; const float* w_tail; const void* y2; const void* ys;
  const float* ms2; const float* mb2; const float* mss; const float* mbs;
  const float* A2; const float* B2; const float* C2; const float* As; const float* Bs; const float* Cs;
  const void* y1; const float* p1s; const float* p1b; const void* wd2; void* da1; float* part2; float* bn_part;   // main path (conv2)
  const void* xin; const float* pxs; const float* pxb; const void* wds; void* gin; float* parts;                  // shortcut (upsample)
  int N;
  int f8in = 0;            // y2 / ys are e4m3 bytes (16 per pixel): fp8 mode's storage of these two tensors
};
bool join_bwd_stream_ok(int dt, int OC, int C, int Hp, int Hg);
int launch_join_bwd_stream(const JoinBwdLaunch& L, hipStream_t s);    // returns blocks = partial images per conv = rows of bn_part
// ... and the block's 1x1 conv afterwards: dy1 from d_a1 (bn1 backward), g_in += dy1 (x) W1 in place, dW1 partial images [blocks][16][16]
struct Conv1BwdLaunch {
  const void* da1; const void* y1; const float* ms; const float* mb; const float* A; const float* B; const float* C;
  const void* xin; const float* pxs; const float* pxb; const void* w1u; void* gin; float* part; long nrows;
  int mask_out = 0;   // g_in leaves masked by the ReLU that produced xin (xin > 0): the block below evaluates its BatchNorm backward on load (JoinGrad)
};
int launch_conv1_bwd_stream(const Conv1BwdLaunch& L, hipStream_t s);  // returns blocks
// conv_joinfwd.hip: a block's residual join fused with the next block's 1x1 conv1 (16 outputs) and bn1's statistics; C = joined channels (16 / 32)
bool join_conv1_fwd_ok(int dt, int C, int Cout_next, long npix);
int launch_join_conv1_fwd(int C, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs, const void* w1_down,
                          void* out, void* y1, float* stats, long npix, hipStream_t s);   // returns the rows of `stats`

// ---------------------------------------------------------------- v2 patch-tile kernels (conv_tile.hip)
// A tile is up to TP (128, or 64/32 for wgrad on tiny feature maps) q-pixels: `qr` consecutive q-rows of one image (tiles_per_img > 0) or `segs` whole images.
// Its input patch per segment is PR x PW pixels starting at input (hq0*SI + oh, ow).
// A tile may hold `sub` (1,2,4) consecutive 128-pixel sub-tiles staged behind ONE barrier pair: sub-tile s starts `sub_j`*s
// q-rows (or `sub_seg`*s images) further, i.e. `sub_pix`*s patch pixels further.
struct TileGeom { int N, Hq, Wq, Hi, Wi, SI, oh, ow, segs, qr, PR, PW, tiles_per_img, ntiles, TP, sub, sub_j, sub_seg, sub_pix; };
bool make_tile_geom(TileGeom& g, int N, int Hq, int Wq, int Hi, int Wi, int SI, int oh, int ow, int span_h, int span_w, int TP = 128,
                    int sub = 1);
struct Wgrad2Args {
  const void* P; const void* G; float* dW;
  const float* proP_scale; const float* proP_shift; int proP_relu;
  const float* proG_scale; const float* proG_shift; int proG_relu;
  TileGeom g;
  int Ca, Cb, Cb_valid, Ca_valid, ksz, ntaps, TG;
  int sA, sB; int tap_off[25]; float scale;
  int P_planar, P_planes;  // 1: P is planar f32 [N][P_planes][Hq][Wq] staged as Ca = 16 zero-padded channels
  int G_planar;            // 1: G is planar T [N][1][Hi][Wi] (the 1-channel image) staged as Cb = 16 zero-padded channels
  int xcd_walk;            // XCD-aware tile order (tile_common.hpp)
  int dbg;                 // developer switches (MMVAE_DBG): 1 skip loads, 2 skip LDS commit, 4 skip MFMA phase
  int nw;                  // waves per block: 4, or 9 / 16 = one tap per wave (deep 3x3 / 4x4 layers)
  int big, wq_shift;       // big tiles (256 / 512 P-pixels, see conv_wgrad.inc); log2(Wq)
  int partial;             // 1: dW is the partial-image scratch [gridDim.x][ntaps][Ca][Cb] (plain stores); 0: atomics into the weight layout
};
size_t wgrad2_lds_bytes(const Wgrad2Args& a, int dt, int TA, int TB);
int wgrad2_patch_slots(const Wgrad2Args& a, int dt, int TB);
int wgrad2_taps_per_block(int ta16, int tb16, int ntaps);
int launch_wgrad2(int dt, const Wgrad2Args& a, int gx, int tiles_ab, int zg, int ta16, int tb16, hipStream_t s);
size_t gather3_lds_bytes(const GatherArgs& a, int dt, int CT);
int launch_gather3(int dt, int out_dt, const GatherArgs& a, int gx, hipStream_t s);
struct WgradReduceArgs { const float* part; float* dW; int Ca, Cb, ntaps, nparts, Ca_valid, Cb_valid, sA, sB; int tap_off[25]; float scale; int exclusive;
                         long part_stride; /* floats between two partial images (0: ntaps * Ca * Cb) */ };
int launch_wgrad_reduce(WgradReduceArgs a, hipStream_t s);
constexpr size_t kWgradScratchBytes = 64u << 20;   // capacity of the partial-image scratch every wgrad caller provides
// ---- pipelined all-phases patch kernel (conv_patch.hip)
struct PatchPhase { int ph, pw, Hq, Wq, ntaps, tap0; long w_off; int w_vec0, koff0; };
struct PatchArgs {
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  const float* bias; float* stats; int accumulate;
  int Cin, Cout, Ho, Wo, SO;
  TileGeom g;                  // one geometry for all phases: q-grid = max over phases, patch = union of the tap spans
  int nphase; PatchPhase phases[kMaxPhases]; Tap taps[kMaxTaps];
  int w_vecs, koff_total;      // LDS carve: weight vec16s of all phases, k-offset ints of all phases
  int x_planar, x_planes, y_planes;
  int npt, wq_shift;           // 16-pixel column tiles per wave (2, 4, 8); log2(Wq) when npt > 2
  int xcd_walk;                // XCD-aware tile order (tile_common.hpp)
  int uni, out_wave_bytes;     // uniform geometry: LDS-staged epilogue, bytes of one wave's staging buffer
  int phase_of[4];             // uni: phase index of output sub-position (ph, pw) = [ph*SO + pw]
  int dbg;                     // developer switches (MMVAE_DBG): 1 skip loads, 2 skip LDS commit, 4 skip MFMA+epilogue, 8 skip stores
  unsigned x_bytes;            // size of x in bytes (< 2^31): buffer-load range for the NHWC staging
  // second source (see GatherArgs): [N][Hq][Wq][Cin2] on the q grid, weights [Cout][Cin2], added into phase x2_phase
  const void* x2; const void* w2; int Cin2, x2_phase, kvp2, w2_vec0, x2_vec0, x2_slots; unsigned x2_bytes;
};
size_t patch_conv_lds_bytes(const PatchArgs& a, int dt);
void patch_conv_x2_carve(PatchArgs& a, int dt);
int patch_conv_slots(const PatchArgs& a, int dt);
int launch_patch_conv(int dt, int out_dt, const PatchArgs& a, int gx, hipStream_t s);   // returns stats rows (= gx) or <0
// ---- deep-layer implicit GEMM (conv_deep2.inc): resident unpadded patch, weights from L2 straight into MFMA fragments
struct DeepPhase { int ph, pw, Hq, Wq, ntaps, tap0; long w_off; };
struct DeepArgs {
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  const float* bias; float* stats; int accumulate;
  int N, Hi, Wi, Cin, Ho, Wo, Cout, SI, SO;
  int Hq, Wq;                  // q-grid of a tile image (max over phases)
  int nphase; DeepPhase phases[kMaxPhases]; Tap taps[kMaxTaps]; int ntaps_all;
  int ipt, ntiles;             // whole images per tile, tiles
  int ct16, npt;               // cout tile / 16 (4 or 8), 16-pixel column tiles per wave
  int fp8;                     // w is e4m3 bytes, activations are quantised in LDS: v_mfma_f32_16x16x32_fp8_fp8 (forward convs, bf16 storage)
  int nw, cpt_log2;            // deep2_conv_kernel: waves per block (32 couts each), log2(64-byte chunks per tap of a weight row)
  long long* ts;               // developer builds only (MMVAE_DEEP2_TS): per-block cycle stamps
  int wfrag;                   // weights are fragment-major (PackArgs::frag)
};
size_t deep2_conv_lds_bytes(const DeepArgs& a, int dt);
int launch_deep2_conv(int dt, int out_dt, const DeepArgs& a, int gx, hipStream_t s);  // conv_deep2.inc; returns stats rows (= gx) or <0
// ---- position-major implicit GEMM for q-grids up to 4x4 (conv_pos.inc; bf16, fragment-major weights): an MFMA column is an image, so
// (position, tap) pairs that fall into the zero padding are not computed at all
struct PosArgs {
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  float* stats; int accumulate;
  int N, Cout;
  int nw;                      // waves per block (32 couts each); grid.y = Cout / (32 nw)
  int ntiles;                  // set by the launcher
};
// up = 0: "down" form (Conv2d forward / ConvT data gradient), up = 1: "up" form; HI / HO input / output map size.  Returns stats rows (> 0),
// 0 when no instantiation takes the geometry, < 0 on error.
int launch_pos_conv(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs& a, hipStream_t s);
bool pos_conv_takes(int K, int S, int P, int up, int HI, int HO, int CIN);
bool conv_force_v1();
int conv_xcd_walk();      // MMVAE_XCD (default 1): XCD-aware tile order in the persistent patch-tile kernels   // MMVAE_CONV_V1=1 forces the generic v1 kernels (A/B and coverage)

// ---------------------------------------------------------------- weight packing
// dst[(col*ntaps + t)*K + k] = T(scale * src[col*s_col + k*s_k + tap_off[t]])
struct PackArgs {
  const float* src; void* dst; int cols, K, ntaps; int s_col, s_k; int tap_off[kMaxTaps]; float scale;
  int cols_valid, K_valid;   // >0: columns / k beyond these are written as zero (channel padding), source not read
  int fp8;                   // 1: dst receives OCP e4m3 bytes (one per element) instead of T
  // 1: fragment-major order for deep2_conv_kernel: the [cols][ntaps*K] matrix is cut into 16-column x 64-byte blocks, each stored as the
  // 1 KB one wave loads as an MFMA A fragment (lane = 16*g + r holds 16 bytes: column 16*blk + r, bytes 16*g .. of the block's chunk):
  // dst[((blk * nchunks + chunk) * 64 + 16*g + r) * VE + e]
  int frag;
};
int launch_pack(int dt, const PackArgs& a, hipStream_t s);
// Batched packing: between pack_batch_begin() and pack_batch_flush() every launch_pack() call is only recorded;
// flush packs all recorded jobs with ONE kernel (grid.y = job).  Not re-entrant (one host thread per net).
void pack_batch_begin();
int pack_batch_flush(int dt, hipStream_t s);

// ---------------------------------------------------------------- stem im2col
// im2col of the 1-channel image for the stem weight gradient: col[m][32] (25 taps, 7 zero pads), T
int launch_stem_im2col(int dt, const void* x, void* col, int N, int H, int W, int Ho, int Wo, hipStream_t s);

// stem backward in one pass (stem_bwd.hip): BatchNorm sums + the three pixel reductions dW is an affine function of
bool stem_bwd_fusable(int S);
// 32 -> 32 channel 3x3 forward convs on 16-pixel-wide output maps as a per-wave stream, optionally with the block's 1x1 stride-2 shortcut
// from the same input rows (conv_fstream.hip): returns stats rows (> 0) or an error
bool conv3_stream_ok(int dt, int Cin, int Cout, int k, int s, int p, int Hin, int Win);
int launch_conv3_stream_bwd(int dt, const void* dy, const void* w_flipped, void* dx, const void* ym, const float* ms, const float* mb, float* sums,
                            int N, int H, hipStream_t s);
int launch_conv3_stream(int dt, int stride, const void* x, const void* w, const void* wsc, void* y, void* ysc, const float* pro_scale,
                        const float* pro_shift, int pro_relu, float* stats, float* stats_sc, int N, int Ho, hipStream_t s);
// ConvTranspose2d(16 -> 16, k4 s2 p1) forward on 16x16 / 32x32 inputs as a per-wave stream (conv_fstream.hip)
bool convT4_stream_ok(int dt, int Cin, int Cout, int k, int s, int p, int Hin, int Win);
int launch_convT4_stream(int dt, const void* x, const void* w_up, void* y, const float* pro_scale, const float* pro_shift, int pro_relu, float* stats,
                         int N, int Hin, hipStream_t s, int f8out = 0);    // f8out: y leaves as e4m3 bytes (32x32 inputs only)
// last up-block join + one-plane tail conv as a per-wave MFMA stream (conv_fstream.hip; bf16, 64x64): returns stats rows or an error
bool tail_fwd_stream_ok(int dt, int OC, int H, int W);
int launch_tail_fwd_stream(int dt, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs, const float* w,
                           const float* bias, float* r_raw, float* stats, int N, int H, int W, hipStream_t s, int f8in = 0);
// the same with the two branch outputs recomputed from the ConvTranspose2d inputs instead of read (conv_fstream.hip; bf16, 32x32 -> 64x64, 16 channels)
bool up5_tail_fwd_ok(int dt, int OC, int C, int Cin, int Hin, int Hout);
int launch_up5_tail_fwd(const void* y1, const float* p1s, const float* p1b, const void* w2_up, const void* xin, const float* pxs, const float* pxb,
                        const void* wu_up, const float* s2, const float* b2, const float* ss, const float* bs, const float* w, const float* bias,
                        float* r_raw, float* stats, int N, hipStream_t s);
// stem forward as a per-wave stream (bf16; stem_bwd.hip): returns stats rows (> 0) or an error
bool stem_fwd_stream_ok(int dt, int S);
int launch_stem_fwd_stream(int dt, const void* x, const float* w, void* y, float* stats, int N, int S, hipStream_t s);
int stem_bwd_part_floats();
int launch_stem_gram(int dt, const void* x, float* scratch, long scratch_cap_floats, double* R, int N, int S, int Ho, int Wo, hipStream_t s);
int launch_stem_bwd(int dt, const void* g, const void* y0, const void* x, const float* ms, const float* mb, float* partials,
                    long partials_cap_floats, int N, int S, int Ho, int Wo, hipStream_t s);   // returns rows
bool stem_bwd_dg_ok(int dt, int S);
int launch_stem_bwd_dg(const void* dy1, const void* dys, const void* wd1, const void* wds, const void* y0, const void* x, const float* ms, const float* mb,
                       float* partials, long partials_cap_floats, int N, int S, int Ho, int Wo, hipStream_t s);   // returns rows
int launch_stem_bwd_finalize(const float* partials, int nparts, const double* R, const float* w, const float* global_sums, double count,
                             const float* gamma, const float* mean, const float* istd, float* dgamma, float* dbeta, float* dW, hipStream_t s);

// ---------------------------------------------------------------- BatchNorm pieces
// per-channel (sum, sumsq) partials over an NHWC tensor of T: out [nparts][2][C]; returns nparts
int launch_chan_stats_nhwc(int dt, const void* y, long npix, int C, float* partials, hipStream_t s);
int chan_stats_parts(long npix, int C);
// same for an NCHW f32 tensor [N][C][HW]
// training finalize: partials -> mean/istd/scale/shift; running stats update (momentum, unbiased var); nbt += 1
struct BnFinalizeArgs {
  float in_scale = 1.f;      // the statistics are of y' = in_scale * y (fp8 layers: statically scaled weights): eps and the running
                             // statistics are converted, so the BatchNorm of y' is exactly the BatchNorm of y
  const float* partials; int nparts; int C; double count;
  const float* gamma; const float* beta; float* running_mean; float* running_var; long long* nbt;
  float* mean; float* istd; float* scale; float* shift; float momentum, eps;
};
int launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);
int launch_partial_rowsum(const float* partials, int nparts, int width, float* out, hipStream_t s, int row_stride = 0);   // SyncBN: [nparts][width (stride row_stride)] -> [width]
// eval: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale
struct BnFoldEntry { int g_off, b_off, rm_off, rv_off, scale_off, shift_off, C; float in_scale; };
constexpr int kBnFoldMax = 96;                       // 96 x 32 B of kernel arguments per launch
struct BnFoldTable { BnFoldEntry e[kBnFoldMax]; };
int launch_bn_fold_eval(const BnFoldEntry* entries, int n, const float* params, const float* bnbuf, float* bnws, float eps, hipStream_t s);
// out = relu(a*sa + ba + b*sb + bb)   (NHWC, T)
int launch_join_fwd(int dt, const void* a, const float* sa, const float* ba, const void* b, const float* sb, const float* bb,
                    void* out, long npix, int C, hipStream_t s);
// out = relu?(a*sa + ba)
int launch_affine_act(int dt, const void* a, const float* sa, const float* ba, int relu, void* out, long npix, int C,
                      hipStream_t s);
// BN-backward reductions.  g = dout * mask, mask = (out > 0) if out!=null else (y0*msk_scale+msk_shift > 0) if msk_scale else 1.
// partials [nparts][1+NY][C]: sum g, sum g*y0, (sum g*y1).  Returns nparts.
int launch_bn_bwd_reduce(int dt, const void* dout, const void* out, const float* msk_scale, const float* msk_shift,
                         const void* y0, const void* y1, long npix, int C, float* partials, hipStream_t s,
                         const float* msk_scale1 = nullptr, const float* msk_shift1 = nullptr);   // both masks: join recomputed
// coefficients for dy = A*g + B*y + Cc, plus dgamma/dbeta (accumulated into grads with +=)
struct BnBwdFinalizeArgs {
  const float* partials; int nparts; int C; int which /*0: y0, 1: y1*/; int ny; double count;
  const float* gamma; const float* mean; const float* istd; float* dgamma; float* dbeta; float* coefA; float* coefB; float* coefC;
  // optional: bias gradient of the conv that feeds this BatchNorm, dbias[c] += dbias_scale * sum(dy) = A sum(g) + B sum(y) + C count
  // in closed form from the sums at hand (analytically zero: a bias in front of a BatchNorm has no gradient; what is left is the
  // rounding of the coefficients) -- no pass over dy, no atomics
  float* dbias_conv = nullptr; float dbias_scale = 1.f;
};
int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s);
int launch_bn_bwd_finalize2(const BnBwdFinalizeArgs& a0, const BnBwdFinalizeArgs& a1, hipStream_t s);   // two BNs of equal width, one launch
// dy0 = A0*g + B0*y0 + C0 ; (dy1 = A1*g + B1*y1 + C1)
int launch_bn_bwd_apply(int dt, const void* dout, const void* out, const float* msk_scale, const float* msk_shift,
                        const void* y0, const float* A0, const float* B0, const float* C0, void* dy0,
                        const void* y1, const float* A1, const float* B1, const float* C1, void* dy1,
                        long npix, int C, hipStream_t s, const float* msk_scale1 = nullptr, const float* msk_shift1 = nullptr);
// last up-block: residual-join backward with the incoming gradient recomputed from the reconstruction gradient (bn_elem.hip)
bool tail_join_fusable(int dt, int OC, int N, int H, int W);
int launch_tail_join_bwd_reduce(int dt, const float* d_raw, const float* w, int OC, int N, int H, int W, const float* ms, const float* mb,
                                const float* ms1, const float* mb1, const void* y0, const void* y1, float* partials, hipStream_t s,
                                float* wpartials = nullptr, int f8in = 0);
int launch_tail_wgrad_finalize(const float* wpartials, int nparts, float* dW, hipStream_t s);
int launch_tail_join_bwd_apply(int dt, const float* d_raw, const float* w, int OC, int N, int H, int W, const float* ms, const float* mb,
                               const float* ms1, const float* mb1, const void* y0, const float* A0, const float* B0, const float* C0, void* dy0,
                               const void* y1, const float* A1, const float* B1, const float* C1, void* dy1, hipStream_t s);
// last up-block forward: residual join + tail conv in one pass, the joined activation is not stored (bn_elem.hip)
bool tail_fwd_fusable(int dt, int OC, int N, int H, int W);
int launch_tail_join_fwd(int dt, const void* y0, const float* ms, const float* mb, const void* y1, const float* ms1, const float* mb1,
                         const float* w, const float* bias, float* r_raw, float* stats, int N, int H, int W, hipStream_t s);
// NCHW f32 variants for the output BatchNorm (decoder.bn2): recon = raw*scale+shift ; backward pieces
int launch_affine_nchw(const float* raw, const float* scale, const float* shift, float* out, int N, int C, int HW, hipStream_t s);
int launch_bn_bwd_reduce_nchw(const float* dout, const float* y, int N, int C, int HW, float* partials, hipStream_t s);
int launch_bn_bwd_apply_nchw(const float* dout, const float* y, const float* A, const float* B, const float* Cc, float* dy,
                             int N, int C, int HW, hipStream_t s);
// Gaussian NLL gradient fused into the output BatchNorm's backward: d = coef / sigma^2 * gscale[0] * (scale*raw + shift - target) is never stored
int launch_gauss_tail_reduce(const float* raw, const float* target, const float* scale, const float* shift, float sigma, float coef, const float* gscale,
                             int N, int C, int HW, float* partials, hipStream_t s);      // partial rows (sum d, sum d*raw); returns nparts
int launch_gauss_tail_apply(const float* raw, const float* target, const float* scale, const float* shift, float sigma, float coef, const float* gscale,
                            const float* A, const float* B, const float* Cc, float* dy, int N, int C, int HW, hipStream_t s);

// ---------------------------------------------------------------- PixelCNN pieces (pixelcnn.hip; reference model.py:227-255)
// InstanceNorm2d (affine = False, eps 1e-5, instance statistics always) forward / backward fused with the ReLU behind it, and the layout
// changes between NCHW f32 and NHWC storage type with 16 zero-padded channels.  stats: [N][C][2] = (mean, istd).
int launch_inorm_planar_fwd(int dt, const float* x, void* xn16, float* stats, int N, int C, int HW, hipStream_t s);
int launch_inorm_planar_bwd(int dt, const void* g16, const float* x, const float* stats, float* dx, int N, int C, int HW, hipStream_t s);
int launch_inorm_nhwc_fwd(int dt, const void* h, void* a, float* stats, int N, int C, int HW, int relu, hipStream_t s);
int launch_inorm_nhwc_bwd(int dt, const void* g, const void* h, const float* stats, void* dh, int N, int C, int HW, int relu, hipStream_t s);
int launch_nhwc16_to_planar(int dt, const void* o16, float* out, int N, int C, int HW, hipStream_t s);
int launch_planar_to_nhwc16(int dt, const float* in, void* o16, int N, int C, int HW, hipStream_t s);
int launch_planar_to_nhwc16(int dt, const void* in, int in_dt, void* o16, int N, int C, int HW, hipStream_t s);

// ---------------------------------------------------------------- latent / loss
// enc = mu + exp(0.5*logvar)*eps (f32 and T copies); kl_partial: -0.5*sum(lv - exp(lv) - mu^2 + 1) (one float, atomically added)
int launch_rsample_fwd(int dt, const float* mu, const float* logvar, const float* eps, float* enc_f32, void* enc_t, long n,
                       hipStream_t s);
int launch_rsample_bwd(const float* d_enc, const float* logvar, const float* eps, float* d_mu, float* d_logvar, long n, hipStream_t s);
int launch_kl_fwd(const float* mu, const float* logvar, long n, double* out, hipStream_t s);
// d_mu += coef*mu ; d_logvar += coef*0.5*(exp(lv)-1)   (coef = kl_weight * upstream / N)
int launch_kl_bwd(const float* mu, const float* logvar, float coef, const float* gscale, float* d_mu, float* d_logvar, long n, hipStream_t s);
// Gaussian NLL: out[0] += sum 0.5*((t-r)/sigma)^2 + log(sigma) + 0.5*log(2pi);  d_r = coef*(r-t)/sigma^2
int launch_gauss_nll_fwd(const float* r, const float* t, long n, float sigma, double* out, hipStream_t s);
int launch_gauss_nll_bwd(const float* r, const float* t, long n, float sigma, float coef, const float* gscale, float* d_r, hipStream_t s);
// weighted cross entropy over NCHW logits [N][Q][HW], int64 targets [N][HW]: out += sum w[t]*(lse - r[t])
int launch_ce_fwd(const float* r, const long long* t, const float* w, int N, int Q, int HW, double* out, hipStream_t s);
int launch_ce_bwd(const float* r, const long long* t, const float* w, int N, int Q, int HW, float coef, const float* gscale, float* d_r,
                  hipStream_t s);
// MMD (sums): out += sum_ij k(x_i,x_j) + sum_ij k(y_i,y_j) - 2 sum_ij k(x_i,y_j), k = exp(-|a-b|^2/d^2)
int launch_mmd_fwd(const float* x, const float* y, int n, int d, double* out, hipStream_t s);
int launch_mmd_fwd_mfma(const float* x, const float* y, int n, int d, float* scratch /*2n floats*/, double* out, hipStream_t s);
// k[i][j] = exp(-|x_i - y_j|^2 / d^2), (n, m) f32  (VAE.compute_kernel, model.py:367-376)
int launch_rbf_matrix(const float* x, const float* y, int n, int m, int d, float* out, hipStream_t s);
// d_y[j] += coef * d(mmd)/d(y_j)
int launch_mmd_bwd(const float* x, const float* y, int n, int d, float coef, const float* gscale, float* d_y, hipStream_t s);

// ---------------------------------------------------------------- optimiser / misc
struct AdamArgs { float* p; const float* g; float* m; float* v; long n; float lr, beta1, beta2, eps, weight_decay; float bc1, bc2; float grad_scale; };
int launch_adam(const AdamArgs& a, hipStream_t s);
int launch_adam_dev(const AdamArgs& a, double* state, hipStream_t s);   // step count on the device (graph capture): bc1 / bc2 ignored
// labels (int64) -> image T [(l - mean)/std] (+ f32 copy for the Gaussian target)
// labels: int64 (label_bytes = 8) or uint8 (1)
int launch_normalise(int dt, const void* labels, int label_bytes, long n, float mean, float stdv, void* img_t, float* img_f32, hipStream_t s);
int launch_quantise_normalise(const unsigned char* frames, long n, const float* centres, int q, float mean, float stdv,
                              long long* labels, float* image, hipStream_t s);
int launch_convert(int dt_in, int dt_out, const void* in, void* out, long n, hipStream_t s);
int launch_fill_f32(float* p, float v, long n, hipStream_t s);
int launch_concat2_to_t(int dt, const float* a, const float* b, int rows, int ca, int cb, void* out, hipStream_t s);
int launch_loss_finish(const double* acc, float* out, float nll, float klc, float mmdc, float n, hipStream_t s);
// out[c] += sum_p partials[p*row_stride + c]  (c < C)
int launch_partials_add(const float* partials, int nparts, int row_stride, int C, float* out, hipStream_t s);

}  // namespace mmvae
