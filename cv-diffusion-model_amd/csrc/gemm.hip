// Pointwise (1x1 conv) GEMM on MFMA for gfx950.
//
// Replaces the reference's 1x1 nn.Conv2d call sites (expand / project / skip efficient_unet.py:174,186,199;
// to_qkv / to_out :265-267) together with the elementwise work eager PyTorch runs around them:
//   prologue (on the A tile, while staging to LDS):  GroupNorm affine [+ReLU6]  (norm1 :207-208) or the
//                                                    SE gate multiply (:100) -- a per-(image, channel) FMA;
//   K-concatenation of up to three A segments:       virtual torch.cat([h, skip]) (:588) and the
//                                                    project+skip pair sharing one accumulator (:226,231);
//   epilogue:                                        bias, identity residual (:234), per-channel
//                                                    (sum, sumsq) slab for the next GroupNorm.
// A rows are NHWC pixels (K contiguous), W is [N][K] (K contiguous): both MFMA operands read 16
// contiguous K-elements per lane.  fp32 accumulation always; T = float uses the exact-f32 MFMA.
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace llie {

// ---- epilogue shared by the GEMM kernels: accumulators -> LDS (fp32) -> 16-byte row vectors (+bias, +residual, stats) -> HBM.
// RAGGED (64-row tiles only): the tile is the last of its image and only its first `vrows` rows exist (P % 64 != 0: image
// sizes that are not a multiple of 64) -- rows beyond are neither stored nor counted in the statistics.
template <typename T, int BM, int BN, int WM, int WN, bool RAGGED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], unsigned char* smem,
                                              int m0, int n0, int img, int vrows = BM) {
  constexpr int NT = WM * WN * 64;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int CP = BN + 4;  // fp32 C-tile pitch
  typedef typename Elem<T>::vec_t vec_t;
  float* sC = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // ---- epilogue: accumulators -> LDS (fp32) -> 16-byte row vectors (+bias, +residual, stats) -> HBM.
  // One pass per 32-row MFMA block index `pi`: the C staging tile holds only WM*32 rows, which keeps the
  // kernel's LDS footprint small enough for 4 workgroups per CU (the full-resolution layers are
  // streaming kernels: occupancy, not MFMA rate, sets their speed).
  constexpr int VR = BN / VEC;   // vectors per output row
  constexpr int RPP = NT / VR;   // rows per pass
  constexpr int SROWS = WM * 32; // rows staged per pass
  constexpr int G = BM < 128 ? BM : 128;  // statistics granularity in rows (what consumers assume: pw_gemm_tile_rows)
  constexpr int SG = BM / G;              // statistic groups per tile
  static_assert(NT % VR == 0 && VR <= 64, "epilogue mapping");
  static_assert(SG == 1 || (RPP <= 32 && 32 % RPP == 0 && MI * 32 == G), "per-group statistics need one wave row per group");
  const int cv = tid % VR, r0 = tid / VR;
  const float oscale = g.seg[0].act == ACT_RELU6_S6 ? 6.f : 1.f;  // the operand was relu6(.) / 6
  float bias[VEC], s1[SG][VEC], s2[SG][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    bias[e] = g.bias ? g.bias[n0 + cv * VEC + e] : 0.f;
#pragma unroll
    for (int q = 0; q < SG; ++q) s1[q][e] = s2[q][e] = 0.f;
  }
  T* outp = reinterpret_cast<T*>(g.out);
  const T* resp = reinterpret_cast<const T*>(g.res);
  const T* dotp = reinterpret_cast<const T*>(g.dot);
#pragma unroll
  for (int pi = 0; pi < MI; ++pi) {
    if (pi) wg_barrier();  // previous pass fully read
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + mfma_row(r, lane);
        const int col = (wn * NI + j) * 32 + (lane & 31);
        sC[row * CP + col] = acc[pi][j][r];
      }
    wg_barrier();
#pragma unroll
    for (int it = 0; it < (SROWS + RPP - 1) / RPP; ++it) {
      const int srow = r0 + it * RPP;
      if (SROWS % RPP != 0 && srow >= SROWS) break;
      const int grp = SG == 1 ? 0 : (it * RPP) >> 5;     // wave row == statistic group (static after unrolling)
      const int row = ((srow >> 5) * MI + pi) * 32 + (srow & 31);  // row inside the BM tile
      if (RAGGED && row >= vrows) continue;
      float v[VEC];
      const float* pc = sC + srow * CP + cv * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = pc[e] * oscale + bias[e];
      const size_t o = (size_t)(m0 + row) * g.N + n0 + cv * VEC;
      if (resp) {
        float rr[VEC];
        ld_f32<T>(resp + o, rr);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += rr[e];
      }
      vec_t ov = f32_to_vec<T>(v);
      if (!g.nostore) st_vec_pol<T>(outp + o, ov, g.nt != 0);  // nostore: statistics-only pass (the consumer recomputes the tensor)
      if (g.stats) {
        if (dotp) {  // backward use: column sums of out * dot (e.g. d(gate) = sum_px da3 * h2) instead of sum / sum of squares
          float dd[VEC];
          ld_f32<T>(dotp + o, dd);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float q = (float)ov[e];
            s1[grp][e] += q * dd[e];
            s2[grp][e] += q;
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float q = (float)ov[e];
            s1[grp][e] += q;
            s2[grp][e] += q * q;
          }
        }
      }
    }
  }
  if (g.stats) {
    float* red = sC + SROWS * CP;  // [waves][2][BN]
    constexpr int NW = NT / 64;
    const int ntiles = RAGGED ? (g.P + G - 1) / G : g.P / G;
#pragma unroll
    for (int q = 0; q < SG; ++q) {
      // lanes with equal cv differ in lane bits >= log2(VR)
#pragma unroll
      for (int o = VR; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          s1[q][e] += __shfl_xor(s1[q][e], o, 64);
          s2[q][e] += __shfl_xor(s2[q][e], o, 64);
        }
      if (q) wg_barrier();
      if (lane < VR) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          red[(wave * 2 + 0) * BN + cv * VEC + e] = s1[q][e];
          red[(wave * 2 + 1) * BN + cv * VEC + e] = s2[q][e];
        }
      }
      wg_barrier();
      const int tile = (m0 - img * g.P) / G + q;
      for (int i = tid; i < 2 * BN; i += NT) {
        const int which = i / BN, c = i % BN;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * BN + c];
        g.stats[((size_t)(img * ntiles + tile) * 2 + which) * g.N + n0 + c] = t;
      }
    }
  }
}

// STAMP: diagnostic build (llie_tune("gemm_stamp", 1)): s_memtime per wave at kernel start / after the K loop / at the end,
// summed into g.stamps[0..2] = {K loop, epilogue, waves}; never used in production.
// KTAIL (BK = 64 only): K segments of 64 n + 32 channels -- chunks never straddle a segment: the last chunk of such a segment has
// no upper 32 columns and stages zeros there (A and W), so the 96 -> 32 project GEMM (K = 384 + 64 + 32) and `base`'s 48 / 96 / 144-
// channel shapes run with 64-wide chunks like their even siblings instead of 32-wide ones (half the barriers per byte: 4.4 -> 4.8
// TB/s on the former).  An all-zero 32-wide k-step adds +0 to every accumulator and the others keep their order: same bits.
template <typename T, int BM, int BN, int WM, int WN, int BK, bool STAMP = false, bool RAGGED = false, bool KTAIL = false>
__global__ void __launch_bounds__(WM* WN * 64) pw_gemm_kernel(const GemmArgs g) {
  static_assert(!RAGGED || BM == 64, "ragged images use the 64-row tiles");
  static_assert(!KTAIL || (BK == 64 && sizeof(T) == 2), "the half-empty last chunk exists for 64-wide chunks only");
  unsigned long long t_start = 0, t_loop = 0;
  if constexpr (STAMP) t_start = __builtin_amdgcn_s_memtime();
  constexpr int NT = WM * WN * 64;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int VPR = BK / VEC;  // 16-byte vectors per BK-wide k-chunk row
  constexpr int PITCH = BK + VEC;  // one 16-byte pad per row: conflict-free ds_read_b128 (80/144/272-byte rows)
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int A_VECS = BM * VPR, B_VECS = BN * VPR;
  constexpr int A_PER = (A_VECS + NT - 1) / NT, B_PER = (B_VECS + NT - 1) / NT;
  constexpr int CP = BN + 4;  // fp32 C-tile pitch
  typedef typename Elem<T>::vec_t vec_t;
  static_assert(NT % VPR == 0, "thread->k-vector mapping must be loop invariant");

  extern __shared__ __align__(16) unsigned char smem[];
  T* sA = reinterpret_cast<T*>(smem);
  T* sB = sA + BM * PITCH;
  float* sC = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nb = g.N / BN;
  // n fastest: neighbours share the A tile.  Workgroups are dealt round-robin to the 8 XCDs (one L2 each), so with more than
  // one N tile the tiles of an M tile are kept on one XCD's L2 (dbg bit 4 = plain order, for A/B runs: up to 14 % faster on the 1024-pixel layers)
  int mt = blockIdx.x / nb, ntile = blockIdx.x % nb;
  const int tpi = RAGGED ? (g.P + BM - 1) / BM : g.P / BM;  // M tiles per image
  const int mtiles = (g.M / g.P) * tpi;
  if (nb > 1 && mtiles % 8 == 0 && !(g.dbg & 16)) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    mt = (slot / nb) * 8 + xcd;
    ntile = slot % nb;
  }
  const int img = mt / tpi, r0 = (mt - img * tpi) * BM;
  const int m0 = img * g.P + r0, n0 = ntile * BN;
  const int vrows = RAGGED ? (g.P - r0 < BM ? g.P - r0 : BM) : BM;  // rows of this tile that exist
  const int kv = (tid % VPR) * VEC;  // this thread's k offset inside every chunk

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  vec_t ra[A_PER], rb[B_PER];
  float rs[VEC], rbv[VEC];
  int r_aff = 0, r_act = 0;

  const int koff1 = g.seg[0].ch, koff2 = g.seg[0].ch + g.seg[1].ch;
  const T* wbase = reinterpret_cast<const T*>(g.w);

  int t_seg = 0, t_cl = 0, t_koff = 0;  // KTAIL: segment of the next chunk, its offset inside the segment, the segment's first k
  auto prefetch = [&](int k0) {
    const int s = KTAIL ? t_seg : (g.nseg > 1 && k0 >= koff1) + (g.nseg > 2 && k0 >= koff2);
    const GemmSeg sg = g.seg[s];
    const int cl = KTAIL ? t_cl : k0 - (s == 0 ? 0 : (s == 1 ? koff1 : koff2));
    if constexpr (KTAIL) k0 = t_koff + t_cl;  // W column of the chunk's first k
    const T* abase = reinterpret_cast<const T*>(sg.ptr);
    const bool kval = !KTAIL || cl + kv < sg.ch;  // this thread's 16-byte column slice exists
    if constexpr (KTAIL) {
      t_cl += BK;
      if (t_cl >= sg.ch) {
        t_koff += sg.ch;
        t_cl = 0;
        ++t_seg;
      }
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT;
      if (A_VECS % NT == 0 || idx < A_VECS) {
        int row = idx / VPR;
        if (RAGGED) row = row < vrows ? row : vrows - 1;  // rows past the image: re-read the last one (never stored or counted)
        if (kval) ra[i] = ld_vec<T>(abase + (size_t)(m0 + row) * sg.ch + cl + kv);
        else ra[i] = vec_t{};
      }
    }
    r_aff = sg.as != nullptr && kval;
    r_act = sg.act;
    if (r_aff) {
      const float* ps = sg.as + (size_t)img * sg.aff_ld + cl + kv;
#pragma unroll
      for (int e = 0; e < VEC; ++e) rs[e] = ps[e];
      if (sg.ab) {
        const float* pb = sg.ab + (size_t)img * sg.aff_ld + cl + kv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) rbv[e] = pb[e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) rbv[e] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_VECS % NT == 0 || idx < B_VECS) {
        const int n = idx / VPR;
        if (kval) rb[i] = ld_vec<T>(wbase + (size_t)(n0 + n) * g.K + k0 + kv);
        else rb[i] = vec_t{};
      }
    }
  };

  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT;
      if (A_VECS % NT == 0 || idx < A_VECS) {
        const int row = idx / VPR;
        vec_t v = ra[i];
        if (r_aff) {
          if constexpr (sizeof(T) == 2) {
            // 2-byte T: one FMA per value straight from the packed word (f16: v_fma_mixlo/hi_f16), ReLU6 as the free clamp
            // of clamp01(z / 6) when the tables come pre-divided (ACT_RELU6_S6).  This prologue is redone for every N tile,
            // and the kernel is short of VALU issue slots before it is short of anything else.
            if (r_act != ACT_RELU6) {
              u32x4 x = reinterpret_cast<const u32x4&>(v), o;
              if (r_act == ACT_RELU6_S6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = act_clamp01_pack<T, true>(x[q], rs[2 * q], rs[2 * q + 1], rbv[2 * q], rbv[2 * q + 1]);
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = act_clamp01_pack<T, false>(x[q], rs[2 * q], rs[2 * q + 1], rbv[2 * q], rbv[2 * q + 1]);
              }
              v = reinterpret_cast<const vec_t&>(o);
            } else {
              float f[VEC];
              vec_to_f32<T>(v, f);
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[e] = relu6f(f[e] * rs[e] + rbv[e]);
              v = f32_to_vec<T>(f);
            }
          } else {
            float f[VEC];
            vec_to_f32<T>(v, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float x = f[e] * rs[e] + rbv[e];
              f[e] = r_act == ACT_RELU6 ? relu6f(x) : x;
            }
            v = f32_to_vec<T>(f);
          }
        }
        st_vec<T>(sA + row * PITCH + kv, v);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_VECS % NT == 0 || idx < B_VECS) {
        const int n = idx / VPR;
        st_vec<T>(sB + n * PITCH + kv, rb[i]);
      }
    }
  };

  int nchunks = g.K / BK;
  if constexpr (KTAIL) {
    nchunks = 0;
    for (int i = 0; i < g.nseg; ++i) nchunks += (g.seg[i].ch + BK - 1) / BK;
  }
  prefetch(0);
  for (int c = 0; c < nchunks; ++c) {
    if (!(g.dbg & 2) || c == 0) stage();  // dbg bit 1: timing ablation (no re-staging)
    wg_barrier();
    if (c + 1 < nchunks && !(g.dbg & 1)) prefetch((c + 1) * BK);  // dbg bit 0: timing ablation (stale tiles)
    const int lr = lane & 31, lk = (lane >> 5) * 16;
#pragma unroll
    for (int sub = 0; sub < BK / 32; ++sub) {
      T fa[MI][16], fb[NI][16];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const T* p = sA + ((wm * MI + i) * 32 + lr) * PITCH + sub * 32 + lk;
#pragma unroll
        for (int q = 0; q < 16 / VEC; ++q) *reinterpret_cast<vec_t*>(&fa[i][q * VEC]) = *reinterpret_cast<const vec_t*>(p + q * VEC);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const T* p = sB + ((wn * NI + j) * 32 + lr) * PITCH + sub * 32 + lk;
#pragma unroll
        for (int q = 0; q < 16 / VEC; ++q) *reinterpret_cast<vec_t*>(&fb[j][q * VEC]) = *reinterpret_cast<const vec_t*>(p + q * VEC);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mfma<T>::chunk(fa[i], fb[j], acc[i][j]);
    }
    wg_barrier();
  }

  if constexpr (STAMP) t_loop = __builtin_amdgcn_s_memtime();
  gemm_epilogue<T, BM, BN, WM, WN, RAGGED>(g, acc, smem, m0, n0, img, vrows);
  if constexpr (STAMP) {
    if (g.stamps && (threadIdx.x & 63) == 0) {  // one (K loop, epilogue) pair per wave, no atomics: the stamped launch keeps its timing
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
      g.stamps[2 * wv] = t_loop - t_start;
      g.stamps[2 * wv + 1] = t_end - t_loop;
    }
  }
}

static int g_gemm_stamp = 0;
static unsigned long long* g_gemm_stamps = nullptr;
static size_t g_gemm_stamp_waves = 0;
constexpr size_t kStampWaves = 1u << 20;
void pw_gemm_stamp(int v) { g_gemm_stamp = v; }
static hipError_t stamp_buffer(size_t waves) {
  if (waves > kStampWaves) return hipErrorInvalidValue;
  if (!g_gemm_stamps && hipMalloc(reinterpret_cast<void**>(&g_gemm_stamps), kStampWaves * 2 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
  g_gemm_stamp_waves = waves;
  return hipSuccess;
}
hipError_t pw_gemm_stamp_fetch(double* out3) {  // mean s_memtime ticks per wave of the last stamped launch: {K loop, epilogue}, and the wave count
  if (!g_gemm_stamps || !g_gemm_stamp_waves) return hipErrorInvalidValue;
  std::vector<unsigned long long> h(g_gemm_stamp_waves * 2);
  hipError_t e = hipMemcpy(h.data(), g_gemm_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  double a = 0, b = 0;
  for (size_t i = 0; i < g_gemm_stamp_waves; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
  out3[0] = a / (double)g_gemm_stamp_waves;
  out3[1] = b / (double)g_gemm_stamp_waves;
  out3[2] = (double)g_gemm_stamp_waves;
  return hipSuccess;
}

template <typename T, int BM, int BN, int WM, int WN, int BK, bool RAGGED = false, bool KTAIL = false>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int NT = WM * WN * 64;
  constexpr int PITCH = BK + Elem<T>::VEC;
  constexpr size_t tiles = (size_t)(BM + BN) * PITCH * sizeof(T);
  constexpr size_t ctile = (size_t)(WM * 32) * (BN + 4) * 4 + (size_t)(NT / 64) * 2 * BN * 4;
  constexpr size_t lds = tiles > ctile ? tiles : ctile;
  static std::atomic<uint64_t> attr_done{0};
  if (lds > 48 * 1024) {
    if (hipError_t e = ensure_max_lds(reinterpret_cast<const void*>(&pw_gemm_kernel<T, BM, BN, WM, WN, BK, false, RAGGED, KTAIL>), (int)lds, attr_done);
        e != hipSuccess)
      return e;
  }
  const unsigned grid = (unsigned)((a.M / a.P) * ((a.P + BM - 1) / BM) * (a.N / BN));  // == (M / BM) * (N / BN) unless RAGGED
  if constexpr (sizeof(T) == 2 && BM == 128 && !KTAIL) {
    if (g_gemm_stamp) {
      if (hipError_t e = stamp_buffer((size_t)grid * (NT / 64)); e != hipSuccess) return e;
      GemmArgs b = a;
      b.stamps = g_gemm_stamps;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<T, BM, BN, WM, WN, BK, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((pw_gemm_kernel<T, BM, BN, WM, WN, BK, true>), dim3(grid), dim3(NT), lds, s, b);
      return hipGetLastError();
    }
  }
  static const std::string name = std::string("pw_gemm_kernel<") + TypeName<T>::value + ", " + std::to_string(BM) + ", " +
                                  std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", " +
                                  std::to_string(BK) + ">";
  note_kernel(name.c_str());
  hipLaunchKernelGGL((pw_gemm_kernel<T, BM, BN, WM, WN, BK, false, RAGGED, KTAIL>), dim3(grid), dim3(NT), lds, s, a);
  return hipGetLastError();
}

int pw_gemm_tile_rows(int P) { return (P % 128 == 0) ? 128 : 64; }
int pw_gemm_ntiles(int P) { const int bm = pw_gemm_tile_rows(P); return (P + bm - 1) / bm; }  // statistics partials per image

// tuning knobs for tools/gpu_tune.py (0 = automatic)
static int g_force_bk = 0, g_bk128 = 1024;
void pw_gemm_force_bk(int bk) { g_force_bk = bk; }
void pw_gemm_bk128(int v) { g_bk128 = v; }

template <typename T>
static hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
  const int BM = pw_gemm_tile_rows(a.P);
  const int BN = (a.N % 128 == 0) ? 128 : ((a.N % 64 == 0) ? 64 : 32);
  // BN depends on N alone, never on the grid: the epilogue's statistics partials are summed per thread over rows r0 + i * (NT / (BN / VEC)),
  // so another BN is another summation order -- narrower tiles for tiny grids (measured in round 3: -1.6 % at B = 1) made a
  // batch differ from its halves in the last bits and were removed.
  bool k64 = sizeof(T) == 2;  // BK = 64 needs every K segment to be a multiple of 64 (2-byte T only)
  for (int i = 0; i < a.nseg; ++i) k64 = k64 && (a.seg[i].ch % 64 == 0);
  if (g_force_bk == 32) k64 = false;
  if constexpr (sizeof(T) == 2) {
    // small grids (about as many workgroups as the chip holds at once): a workgroup's time is chunks x memory latency, so
    // twice the chunk: -10...-16 % on the long-K project layers of the low resolutions (and at B = 1), +10 % on large grids.
    // The sequence of 32-wide k-steps per accumulator is unchanged: same bits, whatever the batch size chooses
    if (g_bk128 && BM == 128 && BN == 128 && (long)(a.M / 128) * (a.N / 128) <= g_bk128) {
      bool k128 = true;
      for (int i = 0; i < a.nseg; ++i) k128 = k128 && (a.seg[i].ch % 128 == 0);
      if (k128) return launch_cfg<T, 128, 128, 2, 2, 128>(a, s);
    }
  }
  if (BM == 128) {
    if constexpr (sizeof(T) == 2) {
      // some segment has 64 n + 32 channels (the 96 -> 32 and 32 -> 64 blocks: K = 480, 160; `base`'s 48 / 96 / 144-channel shapes):
      // 64-wide chunks with half-empty segment tails (KTAIL) instead of 32-wide ones
      if (!k64 && g_force_bk != 32 && a.K > 64) {
        if (BN == 128) return launch_cfg<T, 128, 128, 2, 2, 64, false, true>(a, s);
        if (BN == 64) return launch_cfg<T, 128, 64, 2, 2, 64, false, true>(a, s);
        return launch_cfg<T, 128, 32, 4, 1, 64, false, true>(a, s);
      }
    }
    if (BN == 128) return k64 ? launch_cfg<T, 128, 128, 2, 2, 64>(a, s) : launch_cfg<T, 128, 128, 2, 2, 32>(a, s);
    if (BN == 64) return k64 ? launch_cfg<T, 128, 64, 2, 2, 64>(a, s) : launch_cfg<T, 128, 64, 2, 2, 32>(a, s);
    return k64 ? launch_cfg<T, 128, 32, 4, 1, 64>(a, s) : launch_cfg<T, 128, 32, 4, 1, 32>(a, s);
  }
  if (a.P % 64) {  // the last 64-row tile of every image is partly empty (image sizes that are not a multiple of 64)
    if (BN == 128) return launch_cfg<T, 64, 128, 2, 2, 32, true>(a, s);
    if (BN == 64) return launch_cfg<T, 64, 64, 2, 2, 32, true>(a, s);
    return launch_cfg<T, 64, 32, 2, 1, 32, true>(a, s);
  }
  if (BN == 128) return launch_cfg<T, 64, 128, 2, 2, 32>(a, s);
  if (BN == 64) return launch_cfg<T, 64, 64, 2, 2, 32>(a, s);
  return launch_cfg<T, 64, 32, 2, 1, 32>(a, s);
}

static int g_gemm_dbg = 0;
void pw_gemm_debug(int v) { g_gemm_dbg = v; }

hipError_t launch_pw_gemm(int dtype, const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  a.dbg = g_gemm_dbg;
  if (a.nostore && (!a.stats || a.res)) return hipErrorInvalidValue;
  // host-side shape contract of the kernel (checked before any launch: an out-of-contract shape
  // would index out of bounds on the device)
  if (a.nseg < 1 || a.nseg > 3 || a.N % 32 || a.K % 32 || a.P < 1 || a.M % a.P) return hipErrorInvalidValue;
  int k = 0;
  for (int i = 0; i < a.nseg; ++i) {
    if (a.seg[i].ch % 32 || a.seg[i].ch <= 0 || !a.seg[i].ptr) return hipErrorInvalidValue;
    k += a.seg[i].ch;
  }
  if (k != a.K) return hipErrorInvalidValue;
  for (int i = 0; i < a.nseg; ++i)  // the output scale of ACT_RELU6_S6 is per GEMM: all segments or none, 2-byte T only
    if ((a.seg[i].act == ACT_RELU6_S6) != (a.seg[0].act == ACT_RELU6_S6) || (a.seg[i].act == ACT_RELU6_S6 && (dtype == 0 || !a.seg[i].as)))
      return hipErrorInvalidValue;
  switch (dtype) {
    case 0: return launch_t<float>(a, s);
    case 1: return launch_t<half_t>(a, s);
    case 2: return launch_t<bf16_t>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace llie
