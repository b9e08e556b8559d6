// Host-side launch API of the gfx950 kernels (implemented in the *.hip files of this directory).
// Everything is asynchronous on the given stream; dtype is an llie_dtype value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace llie {

// ACT_RELU6_S6 (pointwise GEMM prologue, 2-byte T): a' = clamp01(a * as + ab) with tables that ALREADY hold scale / 6 and
// shift / 6 (GnFinalizeArgs::post_scale), i.e. relu6(.) / 6 for one FMA with a free clamp; the GEMM multiplies its
// accumulators by 6 in the epilogue.  Every K-segment of such a GEMM must use it.
enum Act : int { ACT_NONE = 0, ACT_RELU6 = 1, ACT_SILU = 2, ACT_RELU6_S6 = 3 };

// Launchers of the profiled kernel classes record the name of the kernel they dispatched (template
// arguments included, as rocprofv3 prints them) so that llie_profile_report can aggregate per kernel.
void note_kernel(const char* name);
const char* last_kernel();

// Per-channel statistics slab written by producers for the next GroupNorm:
//   slab[((b * ntiles + tile) * 2 + {0: sum, 1: sum of squares}) * C + c]   (fp32)
// `tile` enumerates the producer's row tiles inside one image.  gn_finalize reduces slabs in a fixed
// order, so results are bitwise reproducible (no float atomics anywhere in the engine).
struct StatSrc {
  const float* slab;  // null = absent
  int ntiles;
  int ch;
};

// One K-segment of a pointwise GEMM's A operand: NHWC rows [M][ch] with an optional per-(image,
// channel) affine + activation applied on load:  a' = act(a * as[b][c] + ab[b][c]).
struct GemmSeg {
  const void* ptr;
  int ch;
  const float* as;   // [B][aff_ld] (pre-offset to this segment's first channel) or null (identity)
  const float* ab;   // same layout, or null (treated as 0 when as != null)
  int aff_ld;        // row stride of as/ab in floats
  int act;           // Act
};

// out[M][N] = sum_seg act(A_seg) * W[N][Ktot]^T (+bias[N]) (+res[M][N]);  optional stats slab of `out`.
struct GemmArgs {
  GemmSeg seg[3];
  int nseg;
  const void* w;      // [N][Ktot] T, K order = segments concatenated
  const float* bias;  // [N] or null
  const void* res;    // [M][N] T or null
  void* out;          // [M][N] T
  float* stats;       // slab of out or null
  int M, N, K;        // K = sum of seg ch
  int P;              // rows (pixels) per image; M = B * P
  int nostore;        // 1: compute and write only the statistics slab (out may be null)
  int dbg;            // timing ablations (set by the launcher from pw_gemm_debug; 0 in production)
  const void* dot;    // optional [M][N] T: the slab then holds (sum out*dot, sum out) instead of (sum, sum of squares)
  int nt;             // 1: `out` is stored non-temporally (set by the engine for tensors beyond the Infinity Cache, common.h: st_vec_pol)
  unsigned long long* stamps;  // diagnostic builds only (pw_gemm_stamp)
};
void pw_gemm_stamp(int v);
hipError_t pw_gemm_stamp_fetch(double* out3);
hipError_t launch_pw_gemm(int dtype, const GemmArgs& a, hipStream_t s);
int pw_gemm_tile_rows(int P);  // BM used for a given P
int pw_gemm_ntiles(int P);     // stats slab tiles per image = ceil(P / BM)
void pw_gemm_force_bk(int bk);  // tuning knob (0 = automatic)
void pw_gemm_bk128(int max_grid);  // 128-wide K chunks for launches of up to this many workgroups (0 = never)
void pw_gemm_debug(int v);      // timing ablations; results are wrong when non-zero

// Expanding pointwise GEMM in activation-stationary form (pwx.hip; 2-byte T, every segment ACT_RELU6_S6):
//   out[M][N] = sum_seg clamp01(A_seg * as + ab) . Wf^T,  Wf = the [N][K] weights times 6, pre-packed in MFMA fragment order
//   (launch_pack_expand), plus the statistics slab of `out` at 128-row granularity (== pw_gemm_tile_rows for P % 128 == 0).
struct ExpandArgs {
  GemmSeg seg[3];
  int nseg;
  const void* wf;
  void* out;
  float* stats;
  int M, N, K, P;
  int nsplit;  // set by the launcher
  int nt;      // 1: non-temporal output stores (see GemmArgs::nt)
  unsigned long long* stamps;  // diagnostic builds only (pw_expand_debug)
};
__host__ __device__ inline long long pw_expand_pack_index(int n, int k, int K) {
  const int lane = (n & 31) + 32 * ((k >> 3) & 1);
  return ((((long long)(n >> 5) * (K >> 4) + (k >> 4)) * 64 + lane) << 3) + (k & 7);
}
bool pw_expand_supported(int dtype, const GemmSeg* seg, int nseg, int M, int N, int K, int P);
bool pw_expand_serves_k(int K);  // the kernel exists for this input width (the builder packs the fragment-ordered weight copy only then)
hipError_t launch_pw_expand(int dtype, const ExpandArgs& a, hipStream_t s);
hipError_t launch_pack_expand(int dtype, const float* src, void* dst, int N, int K, float scale, hipStream_t s);
void pw_expand_enable(int v);  // knob "pwx" (1 = use where supported)
void pw_expand_debug(int ablate, int stamp, int nbw = -1);  // timing studies (results wrong when 0 < ablate < 6); -1 = leave unchanged
hipError_t pw_expand_stamp_fetch(double* out4);

// GroupNorm statistics -> per-(image, channel) affine tables.
//   mean/var over groups of cg = C/32 channels x P pixels from up to two slabs (virtual concat),
//   as[b][c] = rstd*gamma[c]*(1+fs),  ab[b][c] = (beta[c] - mean*rstd*gamma[c])*(1+fs) + fh
//   with (fs, fh) = FiLM (scale, shift) = film[b*film_stride + {c, C + c}] or (0,0) when film == null.
struct GnFinalizeArgs {
  StatSrc src[2];
  int C;             // total channels (sum of src ch)
  int groups;        // 32
  int P;             // pixels per image
  const float* gamma;
  const float* beta;
  const float* film; // null or [rows][2C] fp32
  int64_t film_stride;  // row stride in floats (0 = same row for every image)
  float eps;
  float* as;         // [B][C]
  float* ab;         // [B][C]
  int B;
  int Creal;         // 0 = C; otherwise the groups partition channels [0, Creal) and the rest (zero padding) gets a zero affine
  float* mean_out;   // optional [B][groups] (training: kept for the backward pass)
  float* rstd_out;
  float post_scale;  // 0 = 1: as and ab are multiplied by it (1/6 for consumers that carry ReLU6 as clamp01(z / 6))
};
hipError_t launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s);

// Depthwise 3x3 (pad 1, stride 1) with fused input affine + ReLU6 and SE average-pool partials.
//   in/out NHWC [B][H][W][C] T; w [9][C] fp32 (tap-major); pool slab [B][ntiles][C] fp32.
struct DwArgs {
  const void* in; void* out;
  const float* as; const float* ab;  // [B][C]
  const float* w;
  float* pool;
  unsigned long long* pool_tot;  // inference: [B][C] fixed-point (x 2^24) channel totals, added with integer atomics (no pool slab)
  int B, H, W, C;
  int no_act;        // 1: prologue is the affine alone (backward: dh2 = da3*gate + dmean/P), 0: affine + ReLU6
  int s6;            // 1 (2-byte T, forward): as / ab hold scale / 6 and shift / 6 (GnFinalizeArgs::post_scale); the
                     // prologue is clamp01(x * as + ab) = relu6(.) / 6 (one FMA, free clamp) and the weights are taken x 6
  // backward epilogue (all four set, pool null): out = conv * [0 < bx*bas + bab < 6], and
  // bslab[b][tile][0][c] = sum out, [1][c] = sum out*bx over 8-row segments (tile = dwconv_ntiles numbering)
  const void* bx; const float* bas; const float* bab; float* bslab;
  int nt;            // 1: non-temporal output stores (see GemmArgs::nt)
};
hipError_t launch_dwconv3x3(int dtype, const DwArgs& a, hipStream_t s);

// GroupNorm-2 statistics of the recompute form from the Gram matrix of the activated block input (gram.hip, 2-byte T).
struct GramArgs {
  const void* x0; const void* x1; int c0, c1;   // block input (virtual concat), K = c0 + c1 in {32, 64, 96}, c0 % 8 == 0
  const float* as1; const float* ab1;            // [B][K] GroupNorm-1 affine DIVIDED BY 6 (a' = clamp01(.) = relu6 / 6)
  float* part;                                    // [B][P / gram_rows][upper 32 x 32 blocks x 1024 + K] workgroup partials
  float* gtot;                                    // [B][K * K + K]: G = sum_px a' a'^T in full, then m = sum_px a'
  unsigned int* tickets;                          // [B], zero at launch (the kernel leaves them zero)
  int B, P, RP;                                   // RP is set by the launcher (gram_rows)
};
struct GramFinalizeArgs {
  const float* gtot;   // as written by gram_stats_kernel
  const void* w1;      // [Chid][K] T: the expand weights the main pass multiplies with
  int K, Chid, groups, P, B;
  const float* gamma; const float* beta;   // norm2
  const float* film; int64_t film_stride;  // null or [rows][2 Chid]
  float eps;
  float* as; float* ab;                    // [B][Chid]
  float post_scale;
};
int gram_rows(int K, int P);
size_t gram_part_floats(int K, int P);
bool gram_supported(int dtype, int K, int c0, int P);
hipError_t launch_gram_stats(int dtype, const GramArgs& a, hipStream_t s);
hipError_t launch_gram_finalize(int dtype, const GramFinalizeArgs& a, hipStream_t s);

// Recompute form of the block's front half (irbx.hip, 2-byte T): expand_stats writes only h1's statistics slab
// ([B][P / irbx_stats_rows(P)][2][Chid]), expand_dw produces h2 and the SE pool slab ([B][irbx_pool_tiles][Chid]).
struct IrbxArgs {
  const void* x0; const void* x1; int c0, c1;   // block input (virtual concat), Cin = c0 + c1 in {32, 64, 96, 128}
  const float* as1; const float* ab1;            // [B][Cin]   GroupNorm-1 affine DIVIDED BY 6 (post_scale): a' = clamp01(.) = relu6 / 6
  const void* w1;                                // [Chid][Cin] T
  const float* as2; const float* ab2;            // [B][Chid]  GroupNorm-2 + FiLM affine (ReLU6 follows), undivided
  const float* wd;                               // [9][Chid] fp32 depthwise weights, tap-major
  void* out; float* pool;                        // expand_dw outputs
  unsigned long long* pool_tot;                  // or: [B][Chid] fixed-point channel totals (see DwArgs::pool_tot)
  float* stats;                                  // expand_stats output
  int B, H, W, Chid;
  unsigned long long* dbg;                       // diagnostic builds only (irbx_stamp)
  int ablate;                                    // timing ablations (results wrong when non-zero; 0 in production)
  int nt;                                        // 1: h2 is stored non-temporally (see GemmArgs::nt)
};
void irbx_ablate(int v);
void irbx_grid(int ks, int v);   // knobs "irbx_grid" (ks = 0: all), "irbx_grid2/4/6": workgroups per expand_dw launch (0 = heuristics)
void irbx_var(int v);    // knob "irbx_var": expand_dw variant bits (irbx.hip: VAR)
void irbx_dwv(int v);  // depthwise phase of expand_dw: 1 = two taps per 16x16x32 MFMA (default), 0 = one tap per 32x32x16 MFMA
void irbx_stamp(int v);
hipError_t irbx_stamp_fetch(double* out10);  // 9 slots (irbx.hip: STAMP) + the number of waves averaged
bool irbx_supported(int dtype, int Cin, int c0, int Chid, int H, int W);
int irbx_pool_tiles(int H, int W);
int irbx_stats_rows(int P);
void irbx_tune(int dbuf, int tiles_per_wg);
hipError_t launch_expand_stats(int dtype, const IrbxArgs& a, hipStream_t s);
hipError_t launch_expand_dw(int dtype, const IrbxArgs& a, hipStream_t s);
int dwconv_ntiles(int H, int W);  // pool slab entries per image: (H/8 row segments) x (W / strip width)
int dw_pick_tyl(int B, int H, int W, int chunks);
void dwconv_swap(int v);   // 1: channel chunk is the fastest grid index
void dwconv_debug(int v);  // timing ablations (bit 0: no MACs, bit 1: no activation); results are wrong when set

// Squeeze-and-Excitation MLP (efficient_unet.py:96-100) in two launches.
//   fc1: mean[b][c] = sum_tiles pool / P (own launch);  hid[b][j] = relu6(b1[j] + sum_c W1[j][c] * mean[b][c])
//   fc2: gate[b][c] = sigmoid(b2[c] + sum_j W2[c][j] * hid[b][j])
struct SeArgs {
  const float* pool; int ntiles; int P;
  int pool_stride;                  // floats between consecutive tile entries of one image (0 = C)
  const void* w1; const float* b1;  // [Cs][C] T
  const void* w2; const float* b2;  // [C][Cs] T
  float* mean;                      // [B][C] scratch
  float* hid;                       // [B][Cs]
  float* gate;                      // [B][C]
  int B, C, Cs;
  const unsigned long long* tot;    // launch_se_gate: [B][C] fixed-point pool totals (kPoolFixScale)
  long long* pre;                   // launch_se_mlp_mfma: [B][Cs] 2^-32 fixed-point fc1 pre-activations, zero at launch
};
// wide blocks, 2-byte T: fc1 and fc2 as two MFMA launches with the batch as the rows of a 32 x 32 tile (small.hip)
bool se_mlp_mfma_supported(int dtype, const SeArgs& a);
hipError_t launch_se_mlp_mfma(int dtype, const SeArgs& a, hipStream_t s);
constexpr float kPoolFixScale = 16777216.f;  // 2^24: per-tile pool partials are rounded to 2^-24 before the integer add
hipError_t launch_se_gate(int dtype, const SeArgs& a, hipStream_t s);  // mean -> fc1 -> ReLU6 -> fc2 -> sigmoid, one launch
hipError_t launch_se_fc1(int dtype, const SeArgs& a, hipStream_t s);
hipError_t launch_se_fc2(int dtype, const SeArgs& a, hipStream_t s);

// Time embedding (efficient_unet.py:60-76, 412-417) and all FiLM projections in one go.
//   temb[r] = W3 * silu(W1 * sinemb(t[r]) + b1) + b3;  film[r][f] = bf[f] + sum_k Wf[f][k]*silu(temb[r][k])
//   rows = 1 (uniform t) or B.
struct TimeArgs {
  const int64_t* t; int rows;
  int dim;  // sinusoidal dim = base_channels
  const float* freqs;  // [dim/2] exp(-ln(1e4)*i/half), tabulated on the host at create time
  int T;    // time_embed_dim
  const float* w1; const float* b1; const float* w3; const float* b3;
  float* temb;        // [rows][T]
  float* silu_temb;   // [rows][T]
  float* emb_out;     // optional [rows][dim]: the sinusoidal embedding itself
};
hipError_t launch_time_embed(const TimeArgs& a, hipStream_t s);
struct FilmArgs {
  const float* silu_temb; int rows; int T;
  const float* wf; const float* bf;  // [F][T], [F]
  float* film;                       // [rows][F]
  int F;
};
hipError_t launch_film(const FilmArgs& a, hipStream_t s);
hipError_t launch_silu_rows(const float* in, float* out, int64_t n, hipStream_t s);

// Dense 3x3 convolutions.
//   init: fp32 NCHW planes (two 3-channel halves = virtual concat) -> NHWC T, + stats slab.
struct InitConvArgs {
  const float* x0; const float* x1; int c0, c1;  // c0 + c1 = Cin
  const float* w; const float* bias;             // [Cin*9][Cout] (repacked), [Cout] fp32
  const void* wp;                                // 2-byte T only: MFMA-packed [5][2][Cout][8] T, or null (VALU kernel)
  void* out; float* stats;
  int B, H, W, Cout;
};
hipError_t launch_init_conv(int dtype, const InitConvArgs& a, hipStream_t s);
int init_conv_ntiles(int H, int W, bool mfma);  // mfma: the 2-byte engines' kernel (8 x 32 tiles); else 16 x 16
//   final: NHWC T -> affine + SiLU -> 3x3 conv C->Cout(3) -> fp32 NCHW; the MFMA variant (2-byte T)
//   can apply LCMScheduler.step to its own output in the epilogue (fuse_step).
struct StepCoef { float sa, sb, sap, sbp; int is_last; int vpred; int clamp_x0; };
struct FinalConvArgs {
  const void* in; const float* as; const float* ab;
  const float* w; const float* bias;             // [9][C][4] (repacked, zero padded), [Cout]
  const void* wp;                                // 2-byte T only: MFMA-packed [C/32][18][2][4][8] T, or null (VALU kernel)
  float* out;                                    // noise prediction (may be null when fuse_step)
  int B, H, W, C, Cout;
  int fuse_step;                                 // 1: also prev = step(eps, sample, noise) (and clamp)
  StepCoef coef;
  const float* sample; const float* noise; float* prev; float* clamped;
};
hipError_t launch_final_conv(int dtype, const FinalConvArgs& a, hipStream_t s);
//   implicit-GEMM MFMA conv: mode 0 = stride-2 downsample, 1 = bilinear x2 upsample then conv (pad 1),
//   2 = plain stride-1 conv (training: conv on the stored upsampled tensor, and every input-gradient conv).
struct Conv3Args {
  const void* in;      // NHWC [B][Hi][Wi][C]
  const void* w;       // [9][Cout][Cin] T
  const float* bias;
  void* out;           // NHWC [B][Ho][Wo][Cout]
  float* stats;        // slab of out or null
  int B, Hi, Wi, Cin, Cout;
  int mode;
  int nt;              // 1: non-temporal output stores (see GemmArgs::nt)
  unsigned long long* stamps;  // diagnostic builds only (conv3x3_stamp)
};
void conv3x3_stamp(int v);
hipError_t conv3x3_stamp_fetch(double* out8);  // 7 per-wave cycle sums (conv.hip: STAMP) + waves averaged
hipError_t launch_conv3x3(int dtype, const Conv3Args& a, hipStream_t s);
int conv3x3_ntiles(int Ho, int Wo);

// Linear attention core (efficient_unet.py:288-302) on qkv NHWC [B][N][3*inner].
struct AttnArgs {
  const void* qkv; int B, N, heads;  // dim_head = 32
  float* kv;      // [nsplit][B][heads][32][33]: rows 0..31 = kv[d][e], column 32 = ksum[d]; partial sums over
                  // position ranges, added in split order by the second pass (deterministic)
  void* out;      // [B][N][inner] T
  int nsplit;     // linattn_nsplit(N)
};
int linattn_nsplit(int N);
hipError_t launch_linattn_kv(int dtype, const AttnArgs& a, hipStream_t s);
hipError_t launch_linattn_out(int dtype, const AttnArgs& a, hipStream_t s);

// y = x*as + ab (+ res), NHWC rows [M][C]; optional stats slab of y (tiles of 64 rows).
struct AffineAddArgs {
  const void* x; const float* as; const float* ab; const void* res; void* y; float* stats;
  int M, C, P;
};
hipError_t launch_affine_add(int dtype, const AffineAddArgs& a, hipStream_t s);
constexpr int kAffineTileRows = 64;

// Layout conversion at the operator boundary: fp32 NCHW <-> NHWC T (+ stats slab, tiles of 64 pixels).
//   x is [B][Csrc][P]; channels [coff, coff+C) are converted.
hipError_t launch_nchw_to_nhwc(int dtype, const float* x, void* y, float* stats, int B, int C, int P, int Csrc,
                               int coff, hipStream_t s);
hipError_t launch_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int P, hipStream_t s, int Cdst = 0,
                               int coff = 0);  // y is [B][Cdst][P]; the C channels land at [coff, coff+C)

// Weight repack at load time (fp32 reference layout -> engine layout).
hipError_t launch_cvt_rows(int dtype, const float* src, void* dst, int rows, int cols, int dst_ld, int dst_col0,
                           hipStream_t s);                       // dst[r*ld + col0 + c] = T(src[r*cols + c])
hipError_t launch_repack_conv3x3(int dtype, const float* src, void* dst, int Cout, int Cin, hipStream_t s, int Op = 0,
                                 int Ip = 0);  // OIHW -> [9][Op][Ip] (Op, Ip: padded destination dims, 0 = unpadded)
hipError_t launch_repack_dw(const float* src, float* dst, int C, hipStream_t s, int Cp = 0);                 // [C][1][3][3] -> [9][Cp]
hipError_t launch_repack_dw_flip(const float* src, float* dst, int C, hipStream_t s, int Cp = 0);            // [C][1][3][3] -> [8-tap][Cp]
hipError_t launch_repack_init(const float* src, float* dst, int O, int I, hipStream_t s, int Op = 0);        // OIHW -> [I*9][Op]
hipError_t launch_repack_final(const float* src, float* dst, int O, int I, hipStream_t s, int Ip = 0);       // OIHW -> [9][Ip][4]
hipError_t launch_repack_final_mfma(int dtype, const float* src, void* dst, int O, int I, hipStream_t s, int Ip = 0);  // OIHW -> [Ip/32][18][2][4][8] T
hipError_t launch_repack_init_mfma(int dtype, const float* src, void* dst, int O, int I, hipStream_t s, int Op = 0);   // OIHW -> [5][2][Op][8] T

// All plain / matrix / 3x3 / depthwise parameters in ONE launch (an optimiser step changes every parameter: 300+
// small repack launches would cost more than the repack itself).  Offsets are bytes into the weight blob; -1 = none.
struct LoadDesc {
  const float* src;
  int kind;            // 0 fp32 copy, 1 matrix (cvt_rows [+ transposed copy]), 2 OIHW 3x3 ([tap][O][I] [+ [8-tap][I][O]]), 3 depthwise ([tap][C] + flipped),
                       // 4 input conv (fp32 [I*9][Op] + MFMA pack at dst_t), 5 output conv (fp32 [9][Ip][4] + MFMA pack at dst_t)
  int as_t;            // matrix: destination in the compute dtype (1) or fp32 (0)
  int rows, cols, ld, col0, O, I, Op, Ip;  // Op / Ip: padded destination dims of the 3x3 / depthwise layouts
  long long numel, dst, dst_t;
  long long dst_f;     // matrix: third copy, times fscale, in MFMA fragment order (pw_expand_pack_index); -1 = none
  float fscale;
};
// state (optional, device): [0] = content hash of the last load, [1] = 1 when the parameters changed (set by launch_params_hash);
// with a state the kernel is a no-op when [1] == 0
hipError_t launch_load_all(int dtype, const LoadDesc* descs_dev, int n, char* blob, hipStream_t s, const unsigned long long* state = nullptr);
hipError_t launch_params_hash(const LoadDesc* descs_dev, int n, unsigned long long* partial, unsigned long long* state, int force,
                              hipStream_t s);

// uint8 HWC RGB <-> normalised fp32 NCHW with bilinear resize (scripts/inference.py:99-134), bit-exact with hostio.py.
hipError_t launch_preprocess_u8(const uint8_t* img, int B, int H0, int W0, float* out, int S, hipStream_t s);
hipError_t launch_postprocess_u8(const float* x, int B, int S, uint8_t* img, int H0, int W0, hipStream_t s);

// LCM scheduler elementwise ops (fp32).
hipError_t launch_lcm_step(const float* eps, const float* x, const float* noise, float* prev, float* x0,
                           float* clamped, int64_t n, StepCoef c, hipStream_t s);
hipError_t launch_add_noise(const float* x0, const float* noise, const int64_t* t, const float* acp, float* out,
                            int B, int64_t per, int velocity, int table_len, hipStream_t s);
hipError_t launch_copy_probe(const void* src, void* dst, int64_t bytes, hipStream_t s);
hipError_t launch_rw_probe(const void* src, void* dst, int64_t units, int r, int w, int nt, hipStream_t s);


// =============================================================================================
// Training: backward kernels (bwd.hip, wgrad.hip).  Gradients of activations are NHWC T like the
// activations; parameter gradients are fp32 in the reference's state_dict layout.  Every reduction
// runs in a fixed order (tile partials, then a sequential combine), so gradients are reproducible.

// (1) activation backward + per-channel partial sums for the GroupNorm backward:
//   dz = g * act'(x*as + ab)   (ReLU6 / SiLU / none);   slab[b][tile][0][c] = sum dz, [1] = sum dz * x  (tiles of 64 rows)
// x is the (virtually concatenated) input of the norm.  dz may alias g; null = do not store (act none).
struct BwdMaskArgs {
  const void* g;
  const void* x0; const void* x1; int c0, c1;
  const float* as; const float* ab;
  int act;
  void* dz;
  float* slab;
  int M, C, P;
};
hipError_t launch_bwd_mask_reduce(int dtype, const BwdMaskArgs& a, hipStream_t s);

// sum a slab over its tiles: out[b][j][c] = sum_t slab[b][t][j][c] for j < nj_out (slab rows have nj entries; pre-offset
// the pointer by j0*C to pick a single component).  Fixed order.
hipError_t launch_slab_reduce(const float* slab, float* out, int B, int ntiles, int nj, int nj_out, int C, hipStream_t s);
// out[c] = sum_b in[b*stride + c]
hipError_t launch_batch_sum(const float* in, float* out, int B, int64_t stride, int C, hipStream_t s);

// (2) GroupNorm backward coefficients from the reduced sums S[b][2][C] and the forward mean / rstd:
//   dx = dz*A + x*Bq + Cq;  dG[b][c] = sum dz*xhat,  dBc[b][c] = sum dz  (before the FiLM / batch reductions)
struct GnBwdArgs {
  const float* S; const float* mean; const float* rstd;   // [B][2][C], [B][groups], [B][groups]
  const float* slab; int ntiles;                          // or (S null): the tile partials [B][ntiles][2][C], summed inside the kernel
  const float* gamma; const float* film; int64_t film_stride;  // FiLM rows of the forward (scale at c, shift at C+c) or null
  int C, groups, P, B;
  float* A; float* Bq; float* Cq; float* dG; float* dBc;   // [B][C] each
};
hipError_t launch_gn_bwd_coef(const GnBwdArgs& a, hipStream_t s);
//   parameter gradients: dgamma[c] = sum_b dG*(1+s), dbeta[c] = sum_b dBc*(1+s);  FiLM: ds = dG*gamma + dBc*beta, df = dBc
struct GnParamGradArgs {
  const float* dG; const float* dBc; const float* gamma; const float* beta;
  const float* film; int64_t film_stride;
  float* dgamma; float* dbeta;
  float* dfilm; int64_t dfilm_stride;   // row b: [c] <- ds, [C + c] <- df  (pre-offset to this block's rows) or null
  int B, C;
};
hipError_t launch_gn_param_grad(const GnParamGradArgs& a, hipStream_t s);
// (3) dx = dz*A + x*Bq + Cq (+ add0) (+ add1): x / dx / add1 follow the virtual-concat split, add0 is dense [M][C]
struct GnApplyArgs {
  const void* dz; const void* x0; const void* x1; int c0, c1;
  const float* A; const float* Bq; const float* Cq;
  const void* add0; const void* add1_0; const void* add1_1;
  void* dx0; void* dx1;
  int M, P;
};
hipError_t launch_gn_bwd_apply(int dtype, const GnApplyArgs& a, hipStream_t s);
hipError_t launch_add_into(int dtype, void* dst, const void* src, int64_t n, hipStream_t s);   // dst += src
hipError_t launch_fill_zero(void* dst, int64_t bytes, hipStream_t s);
hipError_t launch_zero_fill(void* dst, int64_t bytes, hipStream_t s);  // kernel, not a memset node

// (4) weight gradient of a 1x1 / one tap of a 3x3 convolution (TN GEMM on MFMA, split over the rows):
//   out[n*ldn + k*ldk + off] = sum_m g[m][n] * A'[src(m)][k],   A' = act(a*as + ab) per K-segment as in GemmArgs;
//   src(m) = pixel (y*stride + dy, x*stride + dx) of the input image (zero outside), m = (b, y, x) over Ho x Wo.
struct WgradArgs {
  const void* g; int N;
  GemmSeg seg[3]; int nseg; int K;
  int B, Ho, Wo, Hi, Wi, stride, dy, dx;
  int ntap;            // 1, or 9: all taps of a 3x3 weight in one launch (dy, dx ignored; tap t lands at off + t)
  int nstore, kstore;  // only rows n < nstore / columns k < kstore are stored (0 = all): operands padded to 32 channels
  float* partial;      // [msplit][ntap][N][K] scratch
  float* out; int64_t ldn, ldk, off;
  int msplit;
};
int wgrad_msplit(int dtype, int M, int N, int K, int ntap);
void wgrad_set_target(int workgroups);  // tuning knob: workgroups per launch the row split aims for (default 1024)
hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t s);

// (5) depthwise 3x3 weight gradient: dw[c][tap] (reference layout [C][1][3][3]) =
//   sum_{b,y,x} (g*gs[b][c] + gb[b][c]) * relu6(h*as[b][c] + ab[b][c]) at (y+ky-1, x+kx-1)
struct DwWgradArgs {
  const void* g; const float* gs; const float* gb;     // incoming gradient and its affine
  const void* h; const float* as; const float* ab;     // forward input of the depthwise conv and its affine (+ReLU6)
  float* partial;      // [B*strips][9][C] scratch
  float* out;          // [C][9]
  int B, H, W, C;
};
int dw_wgrad_strips(int H, int W);
hipError_t launch_dw_wgrad(int dtype, const DwWgradArgs& a, hipStream_t s);

// (6) small dense pieces over the batch (SE MLP, FiLM / time-embedding Linears): fp32 activations
//   linear_dx:  dx[b][k] (=|+=) sum_r dy[b][r] * W[r][k]      (W is [R][Kc], T or fp32)
//   linear_dw:  dW[r][k] = sum_b dy[b][r] * x[b][k],  db[r] = sum_b dy[b][r]      (fp32 outputs)
//   dy rows have stride dy_stride (>= R) so that a slice of a wider table can be used in place
//   scratch (optional, linear_dx_chunks(R)*B*Kc floats): lets a long R be cut into chunks that run in parallel
int linear_dx_chunks(int R);
hipError_t launch_linear_dx(int wdtype, const float* dy, int64_t dy_stride, const void* W, float* dx, int B, int R, int Kc,
                            hipStream_t s, float* scratch = nullptr);
hipError_t launch_linear_dw(const float* dy, int64_t dy_stride, const float* x, float* dW, float* db, int B, int R, int Kc,
                            hipStream_t s);
//   SE gate: dpre2 = dgate * g*(1-g);  ReLU6 hidden: dpre1 = dr * [0 < r < 6];  SiLU: dx = dy * silu'(x)
hipError_t launch_sigmoid_bwd(const float* dgate, const float* gate, float* out, int64_t n, hipStream_t s);
hipError_t launch_relu6_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t s);
hipError_t launch_silu_bwd(const float* dy, const float* x, float* out, int64_t n, hipStream_t s);
hipError_t launch_scale_rows(const float* x, float* out, int64_t n, float scale, hipStream_t s);
//   sinusoidal embedding rows (SinusoidalPosEmb, efficient_unet.py:68-76) for the time-MLP weight gradient
hipError_t launch_sin_embed(const int64_t* t, const float* freqs, float* emb, int rows, int dim, hipStream_t s);

// (7) dense 3x3 pieces
hipError_t launch_upsample2x(int dtype, const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s);       // bilinear, align_corners=False
hipError_t launch_upsample2x_bwd(int dtype, const void* dout, void* din, int B, int Hi, int Wi, int C, hipStream_t s); // adjoint
hipError_t launch_dilate2x(int dtype, const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s);          // out[2y][2x] = in[y][x], zeros elsewhere
hipError_t launch_repack_conv3x3_t(int dtype, const float* src, void* dst, int Cout, int Cin, hipStream_t s);           // OIHW -> [8-tap][I][O] (input-gradient conv)
hipError_t launch_cvt_rows_t(int dtype, const float* src, void* dst, int rows, int cols, hipStream_t s);                // dst[c][r] = T(src[r][c])
//   output head: d(eps) fp32 NCHW [B][Cout][H][W] -> da NHWC [M][C] T (gradient of the SiLU output), and its weight gradient
struct FinalBwdArgs {
  const float* deps; const float* w;       // w: the forward kernel's repacked fp32 weights [9][C][4]
  void* da;                                 // [M][C] T
  int B, H, W, C, Cout;
};
hipError_t launch_final_bwd_data(int dtype, const FinalBwdArgs& a, hipStream_t s);
//   fp32 NCHW planes ([B][c0][P] and optionally [B][c1][P]) -> NHWC T [B*P][32], zero beyond the real channels: the
//   operand layout launch_wgrad wants (weight gradients of the output head and of the input conv)
hipError_t launch_pack_planes(int dtype, const float* x0, const float* x1, int c0, int c1, void* out, int B, int P, hipStream_t s);

// (8) linear attention backward.  qkv / dqkv NHWC [B][N][3*inner]; kv = forward partials [nsplit][B][heads][32][33].
struct AttnBwdArgs {
  const void* qkv; const void* dout; void* dqkv;
  const float* kv; int nsplit;
  float* dkv;          // [B][heads][N/64][32][33] partials written by pass A, read by pass B
  int B, N, heads;
};
hipError_t launch_linattn_bwd_q(int dtype, const AttnBwdArgs& a, hipStream_t s);
hipError_t launch_linattn_bwd_kv(int dtype, const AttnBwdArgs& a, hipStream_t s);

// (9) optimiser step over all parameter tensors (optim.hip): gradient norm -> clip coefficient -> AdamW (+ EMA shadow)
constexpr int kOptChunk = 4096;  // elements per workgroup
struct OptTensor { float* p; float* m; float* v; float* ema; long long goff; long long n; };  // goff: element offset of the gradient in the flat buffer
struct OptChunk { int tensor; int first; };                                                    // first element of the run inside the tensor
struct OptStepArgs {
  const OptTensor* tensors; const OptChunk* chunks; int nchunks;  // device tables
  const float* gbase;                                              // flat fp32 gradients
  double* partial;                                                 // [nchunks] scratch
  float* stats;                                                    // [3]: ||g||, factor applied to g, 1 = step skipped
  double lr, beta1, beta2, eps, weight_decay;
  double max_grad_norm;  // <= 0: no clipping
  double ema_decay;      // < 0: no EMA update
  double grad_scale;     // multiplies every gradient first (1 / loss scale, 1 / world size)
  long long step;        // 1-based step count (bias correction)
  int skip_nonfinite;    // leave everything untouched when the norm is inf / NaN (GradScaler.step)
};
hipError_t launch_optimizer_step(const OptStepArgs& a, hipStream_t s);

}  // namespace llie
