// Weight gradient of the pointwise (1x1) convolutions and, one tap per launch, of the dense 3x3
// convolutions: a TN GEMM  dW[n][k] = sum_m g[m][n] * A'[src(m)][k]  on MFMA.
//
// The contraction runs over pixels, which is the slow index of both NHWC operands.  Each 64-row chunk is
// staged row-major in LDS with 16-byte stores and read back column-wise by the hardware transpose read
// ds_read_b64_tr_b16 (gfx950), so a lane gets 16 consecutive rows of its channel per 32-row MFMA chunk (the
// forward GEMM's operand scheme with the roles of rows and channels swapped); fp32 reads columns directly.  The rows are split over blockIdx.z; the fp32 partials are summed in
// split order by a second kernel, so the result does not depend on scheduling.
#include <string>

#include "common.h"

namespace llie {

constexpr int kWgRows = 64;  // rows (pixels) per staged chunk

// 16 contraction values (rows m = base .. base+15 of one channel column) for an MFMA operand, from a row-major
// [row][channel] LDS image.  2-byte types: four ds_read_b64_tr_b16 (each hands a lane 4 consecutive rows of its
// column; lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the group's 4x16 block); fp32: 16
// scalar reads (consecutive lanes = consecutive channels, conflict-free).
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
template <typename T>
__device__ __forceinline__ void load_operand_rows(const T* img, int pitch, int row_base, int col_base, int lane, T* frag) {
  if constexpr (sizeof(T) == 2) {
    const int q = (lane & 15) >> 2, pp = lane & 3, cg = (lane >> 4) & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T* ptr = img + (row_base + 4 * j + q) * pitch + col_base + 16 * cg + 4 * pp;
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
      *reinterpret_cast<s16x4*>(frag + 4 * j) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) frag[j] = img[(row_base + j) * pitch + col_base + (lane & 31)];
  }
}

// WT = 32-wide blocks per wave and dimension: the workgroup tile is (64*WT) x (64*WT) outputs.  WT = 2 halves the
// re-reads of g (once per k-tile) and A' (once per n-tile) for the wide layers.
template <typename T, int WT>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs a, int rows_per_split) {
  constexpr int TILE = 64 * WT;
  constexpr int VEC = Elem<T>::VEC, VPR = TILE / VEC, RPP = 256 / VPR, NP = kWgRows / RPP;  // passes to load 64 rows x TILE channels
  // row pitch in bytes = 64 (mod 256): the 4-row blocks a 32-lane half reads with ds_read_b64_tr_b16 land 16 banks apart
  constexpr int PITCH = sizeof(T) == 2 ? TILE + 32 : TILE + 4;
  __shared__ __align__(16) T sG[kWgRows * PITCH];
  __shared__ __align__(16) T sA[kWgRows * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int ktiles = (a.K + TILE - 1) / TILE;
  const int tap = blockIdx.y / ktiles;  // ntap == 9: all taps of a 3x3 weight in one launch (tap-major grid.y, so the
                                        // nine workgroups that share a g chunk are scheduled together and hit L2)
  const int n0 = blockIdx.x * TILE, k0 = (blockIdx.y % ktiles) * TILE;
  const int tdy = a.ntap == 9 ? tap / 3 - 1 : a.dy, tdx = a.ntap == 9 ? tap % 3 - 1 : a.dx;
  const int cv = (tid % VPR) * VEC, rl = tid / VPR;
  const int P = a.Ho * a.Wo;
  const size_t m_begin = (size_t)blockIdx.z * rows_per_split;

  // K segment of this thread's A channels
  const int kc = k0 + cv;
  const T* aptr = nullptr;
  int ach = 0, aoff = 0, aact = ACT_NONE, ald = 0;
  const float* aas = nullptr;
  const float* aab = nullptr;
  if (kc < a.K) {
    int base = 0;
    for (int sgi = 0; sgi < a.nseg; ++sgi) {
      const GemmSeg sg = a.seg[sgi];
      if (kc < base + sg.ch) {
        aptr = reinterpret_cast<const T*>(sg.ptr); ach = sg.ch; aoff = kc - base;
        aas = sg.as; aab = sg.ab; ald = sg.aff_ld; aact = sg.act;
        break;
      }
      base += sg.ch;
    }
  }
  const bool g_ok = n0 + cv < a.N;
  const T* gp = reinterpret_cast<const T*>(a.g);

  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  typedef typename Elem<T>::vec_t vec_t;
  vec_t zero;
#pragma unroll
  for (int e = 0; e < VEC; ++e) zero[e] = (T)0.f;

  // Every 64-row chunk lies inside one image (P % 64 == 0, splits start on multiples of 64), so the image index and
  // the prologue's per-(image, channel) affine are chunk-uniform: 32-bit index math once per chunk, the affine is
  // reloaded only when the image changes.  1x1 layers (source pixel == output pixel) skip the pixel decomposition.
  const int chunks_per_image = P / kWgRows;
  const int chunk0 = (int)(m_begin / kWgRows);
  const bool plain = a.ntap == 1 && a.stride == 1 && a.dy == 0 && a.dx == 0 && a.Hi == a.Ho && a.Wi == a.Wo;
  float sc[VEC], sh[VEC];
  int cur_b = -1;
  vec_t gv[NP], av[NP];
  auto fetch = [&](int mc) {
    const int chunk = chunk0 + mc / kWgRows;
    const int b = chunk / chunks_per_image;
    const int pix0 = (chunk - b * chunks_per_image) * kWgRows;
    if (aptr && aas && b != cur_b) {
      cur_b = b;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        sc[e] = aas[(size_t)b * ald + aoff + e];
        sh[e] = aab ? aab[(size_t)b * ald + aoff + e] : 0.f;
      }
    }
    const size_t mrow0 = (size_t)chunk * kWgRows;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = ps * RPP + rl;
      gv[ps] = g_ok ? ld_vec<T>(gp + (mrow0 + r) * a.N + n0 + cv) : zero;
      av[ps] = zero;
      if (aptr) {
        bool ok = true;
        size_t src = mrow0 + r;  // plain: same pixel
        if (!plain) {
          const int pix = pix0 + r;
          const int y = (pix / a.Wo) * a.stride + tdy, x = (pix % a.Wo) * a.stride + tdx;
          ok = y >= 0 && y < a.Hi && x >= 0 && x < a.Wi;
          src = ((size_t)b * a.Hi + y) * a.Wi + x;
        }
        if (ok) {
          vec_t v = ld_vec<T>(aptr + src * ach + aoff);
          if (aas || aact != ACT_NONE) {
            float f[VEC];
            vec_to_f32<T>(v, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = apply_act(aas ? f[e] * sc[e] + sh[e] : f[e], aact);
            v = f32_to_vec<T>(f);
          }
          av[ps] = v;
        }
      }
    }
  };
  fetch(0);
  for (int mc = 0; mc < rows_per_split; mc += kWgRows) {
    wg_barrier();  // previous chunk's operand reads are done
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = ps * RPP + rl;
      st_vec<T>(sG + r * PITCH + cv, gv[ps]);
      st_vec<T>(sA + r * PITCH + cv, av[ps]);
    }
    wg_barrier();
    if (mc + kWgRows < rows_per_split) fetch(mc + kWgRows);  // next chunk's global loads fly under the MFMAs
#pragma unroll
    for (int ch = 0; ch < kWgRows / 32; ++ch) {
      T fa[WT][16], fb[WT][16];
      const int rb = ch * 32 + (lane >> 5) * 16;  // this lane half's 16 contraction rows of the 32-row MFMA chunk
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        load_operand_rows<T>(sG, PITCH, rb, (wn * WT + i) * 32, lane, fa[i]);
        load_operand_rows<T>(sA, PITCH, rb, (wk * WT + i) * 32, lane, fb[i]);
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) Mfma<T>::chunk(fa[i], fb[j], acc[i][j]);
    }
  }
  // D[row = n (A-operand row)][col = k (B-operand row)]
  float* out = a.partial + ((size_t)blockIdx.z * a.ntap + tap) * a.N * a.K;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int k = k0 + (wk * WT + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * WT + i) * 32 + mfma_row(r, lane);
        if (n < a.N && k < a.K) out[(size_t)n * a.K + k] = acc[i][j][r];
      }
    }
}

// Two-stage combine of the split partials.  Stage 1: kWgGroups blocks per 256 outputs, block g sums the splits
// g, g + kWgGroups, ... (in that order) into row g of the partial buffer (it is the only reader of those rows).
// Stage 2: one thread per output adds the kWgGroups rows in order and stores with the destination strides.
constexpr int kWgGroups = 16;
__global__ void __launch_bounds__(256) wgrad_reduce1_kernel(float* partial, int64_t nk, int msplit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= nk || g >= msplit) return;
  float s = 0.f;
  for (int sp = g; sp < msplit; sp += kWgGroups) s += partial[(size_t)sp * nk + i];
  partial[(size_t)g * nk + i] = s;
}
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* partial, float* out, int N, int K, int ntap, int msplit,
                                                           int64_t ldn, int64_t ldk, int64_t off, int nstore, int kstore) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = (int64_t)ntap * N * K;
  if (i >= tot) return;
  float s = 0.f;
  for (int sp = 0; sp < msplit; ++sp) s += partial[(size_t)sp * tot + i];
  const int tap = (int)(i / ((int64_t)N * K));
  const int64_t j = i % ((int64_t)N * K);
  const int n = (int)(j / K), k = (int)(j % K);
  if (n < nstore && k < kstore) out[(size_t)n * ldn + (size_t)k * ldk + off + (ntap == 9 ? tap : 0)] = s;
}

// number of row splits: enough workgroups to fill the GPU, each split a multiple of 64 rows
static int g_wgrad_target = 1024;  // workgroups per launch the row split aims for
void wgrad_set_target(int v) { g_wgrad_target = v > 0 ? v : 1024; }
// 128x128 tiles for the wide layers of the 2-byte engines (fp32 tiles would not fit the 64 KB static LDS)
static int wgrad_tile(int dtype, int N, int K) { return (dtype != 0 && N >= 128 && K > 64) ? 128 : 64; }
int wgrad_msplit(int dtype, int M, int N, int K, int ntap) {
  const int t = wgrad_tile(dtype, N, K);
  const int tiles = ((N + t - 1) / t) * ((K + t - 1) / t) * ntap;
  int ms = 1;
  while (tiles * ms < g_wgrad_target && M % (ms * 2 * kWgRows) == 0 && M / (ms * 2) >= 256) ms *= 2;
  return ms;
}

hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t s) {
  const int M = a.B * a.Ho * a.Wo;
  if (a.msplit < 1 || M % (a.msplit * kWgRows) || a.N % 32 || a.K % 32 || a.nseg < 1 || a.nseg > 3) return hipErrorInvalidValue;
  int k = 0;
  for (int i = 0; i < a.nseg; ++i) {
    if (a.seg[i].ch % 32) return hipErrorInvalidValue;
    k += a.seg[i].ch;
  }
  if (k != a.K) return hipErrorInvalidValue;
  if (a.ntap != 1 && a.ntap != 9) return hipErrorInvalidValue;
  const int t = wgrad_tile(dtype, a.N, a.K);
  dim3 grid((a.N + t - 1) / t, ((a.K + t - 1) / t) * a.ntap, a.msplit);
  const int rps = M / a.msplit;
  switch (dtype) {
    case 0: hipLaunchKernelGGL((wgrad_kernel<float, 1>), grid, dim3(256), 0, s, a, rps); break;
    case 1:
      if (t == 128) hipLaunchKernelGGL((wgrad_kernel<half_t, 2>), grid, dim3(256), 0, s, a, rps);
      else hipLaunchKernelGGL((wgrad_kernel<half_t, 1>), grid, dim3(256), 0, s, a, rps);
      break;
    case 2:
      if (t == 128) hipLaunchKernelGGL((wgrad_kernel<bf16_t, 2>), grid, dim3(256), 0, s, a, rps);
      else hipLaunchKernelGGL((wgrad_kernel<bf16_t, 1>), grid, dim3(256), 0, s, a, rps);
      break;
    default: return hipErrorInvalidValue;
  }
  const int64_t n = (int64_t)a.ntap * a.N * a.K;
  int rows = a.msplit;
  // two stages only where one pass would leave the chip empty: from 32 768 outputs on (128 workgroups) every thread adds its
  // msplit partials itself -- coalesced, L2-resident -- and the launch of stage 1 (60 per training step, ~13 us each) is saved
  if (a.msplit > kWgGroups && n < 32768) {
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((unsigned)((n + 255) / 256), kWgGroups), dim3(256), 0, s, a.partial, n, a.msplit);
    rows = kWgGroups;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.partial, a.out, a.N, a.K,
                     a.ntap, rows, a.ldn, a.ldk, a.off, a.nstore > 0 ? a.nstore : a.N, a.kstore > 0 ? a.kstore : a.K);
  return hipGetLastError();
}

}  // namespace llie
