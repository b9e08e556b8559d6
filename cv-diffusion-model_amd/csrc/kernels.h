// Host-side launch API of the gfx950 kernels (implemented in the *.hip files of this directory).
// Everything is asynchronous on the given stream; dtype is an llie_dtype value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace llie {

enum Act : int { ACT_NONE = 0, ACT_RELU6 = 1, ACT_SILU = 2 };

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
};
hipError_t launch_pw_gemm(int dtype, const GemmArgs& a, hipStream_t s);
bool pw_gemm2_supported(int dtype, const GemmArgs& a);   // LDS-DMA pipelined variant (gemm2.hip)
hipError_t launch_pw_gemm2(const GemmArgs& a, hipStream_t s);
void pw_gemm_use_v2(int v);
int pw_gemm_tile_rows(int P);  // BM used for a given P (stats slab tiles = P / BM)
void pw_gemm_force_bk(int bk);  // tuning knob (0 = automatic)
void pw_gemm_debug(int v);      // timing ablations; results are wrong when non-zero

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
};
hipError_t launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s);

// Depthwise 3x3 (pad 1, stride 1) with fused input affine + ReLU6 and SE average-pool partials.
//   in/out NHWC [B][H][W][C] T; w [9][C] fp32 (tap-major); pool slab [B][ntiles][C] fp32.
struct DwArgs {
  const void* in; void* out;
  const float* as; const float* ab;  // [B][C]
  const float* w;
  float* pool;
  int B, H, W, C;
};
hipError_t launch_dwconv3x3(int dtype, const DwArgs& a, hipStream_t s);

// Fused expand + depthwise ("recompute" form, 2-byte T only): h2 = dw3x3(relu6(aff2(W1 . relu6(aff1(x))))).
// The 4x-expanded tensor h1 is never stored: every workgroup recomputes the rows it needs with MFMA
// from the narrow block input x (virtual concat of up to two NHWC tensors).
struct DwxArgs {
  const void* x0; const void* x1; int c0, c1;   // Cin = c0 + c1 in {32, 64, 96, 128}
  const float* as1; const float* ab1;            // [B][Cin]   GroupNorm-1 affine (ReLU6 follows)
  const void* w1;                                // [Chid][Cin] T
  const float* as2; const float* ab2;            // [B][Chid]  GroupNorm-2 + FiLM affine (ReLU6 follows)
  const float* wd;                               // [9][Chid] fp32 depthwise weights, tap-major
  void* out; float* pool;
  int B, H, W, Chid;
};
bool dwx_supported(int dtype, int Cin, int Chid, int H, int W);
hipError_t launch_dwx(int dtype, const DwxArgs& a, hipStream_t s);
int dwconv_ntiles(int H, int W);  // pool slab entries per image: (H/8 row segments) x (W / strip width)
int dw_pick_tyl(int B, int H, int W, int chunks);
void dwconv_debug(int v);  // timing ablations (bit 0: no MACs, bit 1: no activation); results are wrong when set

// Squeeze-and-Excitation MLP (efficient_unet.py:96-100) in two launches.
//   fc1: mean[b][c] = sum_tiles pool / P (own launch);  hid[b][j] = relu6(b1[j] + sum_c W1[j][c] * mean[b][c])
//   fc2: gate[b][c] = sigmoid(b2[c] + sum_j W2[c][j] * hid[b][j])
struct SeArgs {
  const float* pool; int ntiles; int P;
  const void* w1; const float* b1;  // [Cs][C] T
  const void* w2; const float* b2;  // [C][Cs] T
  float* mean;                      // [B][C] scratch
  float* hid;                       // [B][Cs]
  float* gate;                      // [B][C]
  int B, C, Cs;
};
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
int init_conv_ntiles(int H, int W);
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
//   implicit-GEMM MFMA conv: mode 0 = stride-2 downsample, 1 = bilinear x2 upsample then conv (pad 1).
struct Conv3Args {
  const void* in;      // NHWC [B][Hi][Wi][C]
  const void* w;       // [9][Cout][Cin] T
  const float* bias;
  void* out;           // NHWC [B][Ho][Wo][Cout]
  float* stats;        // slab of out or null
  int B, Hi, Wi, Cin, Cout;
  int mode;
};
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
hipError_t launch_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int P, hipStream_t s);

// Weight repack at load time (fp32 reference layout -> engine layout).
hipError_t launch_cvt_rows(int dtype, const float* src, void* dst, int rows, int cols, int dst_ld, int dst_col0,
                           hipStream_t s);                       // dst[r*ld + col0 + c] = T(src[r*cols + c])
hipError_t launch_repack_conv3x3(int dtype, const float* src, void* dst, int Cout, int Cin, hipStream_t s);  // OIHW -> [9][O][I]
hipError_t launch_repack_dw(const float* src, float* dst, int C, hipStream_t s);                             // [C][1][3][3] -> [9][C]
hipError_t launch_repack_init(const float* src, float* dst, int O, int I, hipStream_t s);                    // OIHW -> [I*9][O]
hipError_t launch_repack_final(const float* src, float* dst, int O, int I, hipStream_t s);                   // OIHW -> [9][I][4]
hipError_t launch_repack_final_mfma(int dtype, const float* src, void* dst, int O, int I, hipStream_t s);    // OIHW -> [I/32][18][2][4][8] T
hipError_t launch_repack_init_mfma(int dtype, const float* src, void* dst, int O, int I, hipStream_t s);     // OIHW -> [5][2][O][8] T

// uint8 HWC RGB <-> normalised fp32 NCHW with bilinear resize (scripts/inference.py:99-134), bit-exact with hostio.py.
hipError_t launch_preprocess_u8(const uint8_t* img, int B, int H0, int W0, float* out, int S, hipStream_t s);
hipError_t launch_postprocess_u8(const float* x, int B, int S, uint8_t* img, int H0, int W0, hipStream_t s);

// LCM scheduler elementwise ops (fp32).
hipError_t launch_lcm_step(const float* eps, const float* x, const float* noise, float* prev, float* x0,
                           float* clamped, int64_t n, StepCoef c, hipStream_t s);
hipError_t launch_add_noise(const float* x0, const float* noise, const int64_t* t, const float* acp, float* out,
                            int B, int64_t per, int velocity, hipStream_t s);

}  // namespace llie
