// Depthwise 3x3 convolution, NHWC, gfx950 -- row-streaming form.
//
// Replaces, for one InvertedResidualBlock (efficient_unet.py:212-223):
//   norm2 affine + FiLM (pre-folded into as/ab by gn_finalize) -> ReLU6 -> depthwise 3x3 (pad 1)
//   -> the read pass of SE's AdaptiveAvgPool2d (:97) as per-tile partial sums.
// HBM-bound: reads the 4x-expanded hidden tensor once, writes it once.
//
// A workgroup owns a column strip: TX pixels wide x TYL rows x (8 lanes x 16 B) channels, and streams
// it top to bottom one input row at a time:
//   * every thread keeps PF global loads (one 16-byte channel vector per row) in flight in registers;
//   * an arriving row is activated ONCE (affine + ReLU6; zero padding applied after the activation,
//     like the reference's conv padding) and parked in a 2-deep LDS row ring, (TX+2) x 8 vectors;
//   * after one barrier each thread reads its three horizontal neighbours (3 ds_read_b128 per output)
//     and feeds three rolling output-row accumulators, so vertical reuse lives in registers.
// LDS per workgroup is ~9 KB, so occupancy is set by registers, not LDS; halo re-reads are 2 rows per
// TYL and 2 columns per TX.
#include <string>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace llie {

// acc += w * f with w and f taken as the LOW (HI = 0) or HIGH (HI = 1) f16 half of two packed registers and
// an fp32 accumulator: one v_fma_mix_f32, no separate conversions (the compiler otherwise emits a
// v_cvt_f32_f16 per operand use -- ~100 extra VALU instructions per row of this kernel).
template <int HI>
__device__ __forceinline__ void fma_mix_f16(float& acc, uint32_t w2, uint32_t f2) {
  if (HI) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(w2), "v"(f2));
  else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(w2), "v"(f2));
}

constexpr int kDwPF = 4;  // rows of global loads kept in flight per thread

// BWD = true is the training backward's instantiation (input gradient of the depthwise conv): the epilogue
// multiplies each output by the ReLU6 derivative of the forward pre-activation z = bx*bas + bab (bx = the tensor the
// forward depthwise read) and writes per-8-row-segment partial sums (sum dz, sum dz*bx) for the GroupNorm backward,
// in place of the SE pool partials.  The forward instantiation (BWD = false) compiles none of it.
template <typename T, int TX, int PFV, bool BWD, bool S6, bool RAGGED>
__device__ __forceinline__ void dw_body_impl(const DwArgs& a, const int TYL, const int dbg, const int swap);
template <typename T, int TX, int PFV, bool BWD, bool S6 = false>
__device__ __forceinline__ void dw_body(const DwArgs& a, const int TYL, const int dbg, const int swap) {
  dw_body_impl<T, TX, PFV, BWD, S6, false>(a, TYL, dbg, swap);
}
template <typename T, int TX, int PFV, bool BWD, bool S6, bool RAGGED>
__device__ __forceinline__ void dw_body_impl(const DwArgs& a, const int TYL, const int dbg, const int swap) {
  constexpr int NT = 8 * TX;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CC = 8 * VEC;  // channels per workgroup
  constexpr int PW = TX + 2;
  constexpr int PF = PFV;
  typedef typename Elem<T>::vec_t vec_t;
  __shared__ vec_t ring[2][PW * 8];
  __shared__ float red[(BWD ? 2 : 1) * 8 * (NT / 64) * CC];  // [8-row segment of the strip][wave][channel] (x2: BWD)

  const int tid = threadIdx.x, cl = tid & 7, xl = tid >> 3;
  const int tiles_x = (a.W + TX - 1) / TX;
  // grid: x = channel chunk (fastest, so the workgroups that share a strip's DRAM pages are dispatched together when
  // g_dw_swap is set), y = strip; or x = strip, y = chunk
  const int tile_id = swap ? blockIdx.y : blockIdx.x, chunk_id = swap ? blockIdx.x : blockIdx.y;
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x;
  const int x0 = tx * TX, y0 = ty * TYL;
  const int c0 = chunk_id * CC + cl * VEC;
  const int b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.C + c0;
  T* out = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.C + c0;

  // ragged images (W % TX != 0 or H % TYL != 0: image sizes that are not a multiple of 64, forward only): columns / rows past the
  // image are read as zero (the conv's padding) and neither stored nor pooled
  const bool col_ok = RAGGED ? x0 + xl < a.W : true;
  // halo duty: threads 0..7 fetch column x0-1, threads 8..15 column x0+TX (their own channel lane)
  const bool is_halo = tid < 16;
  const int hx = tid < 8 ? x0 - 1 : x0 + TX;
  const bool hx_ok = is_halo && hx >= 0 && hx < a.W;
  const int hslot = tid < 8 ? 0 : TX + 1;

  float sc[VEC], sh[VEC];
  vec_t w[9];  // weights stay packed in T (16 B per tap); see keep_packed() below
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = a.as[(size_t)b * a.C + c0 + e];
    sh[e] = a.ab[(size_t)b * a.C + c0 + e];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) w[t][e] = (T)((S6 ? 6.f : 1.f) * a.w[(size_t)t * a.C + c0 + e]);

  const int nrows = TYL + 2;  // input rows y0-1 .. y0+TYL
  vec_t pre[PF], preh[PF];
  auto issue = [&](int r, vec_t& v, vec_t& vh) {
    const int gy = y0 - 1 + r;
    if (r < nrows && gy >= 0 && gy < a.H) {
      const T* row = in + (size_t)gy * a.W * a.C;
      if (col_ok) v = ld_vec<T>(row + (size_t)(x0 + xl) * a.C);
      if (hx_ok) vh = ld_vec<T>(row + (size_t)hx * a.C);
    }
  };
  auto activate = [&](vec_t v) {
    if constexpr (S6 && sizeof(T) == 2) {  // relu6(z) / 6 = clamp01(z / 6): one FMA per value, the clamp is free
      const u32x4 x = reinterpret_cast<const u32x4&>(v);
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = act_clamp01_pack<T, true>(x[q], sc[2 * q], sc[2 * q + 1], sh[2 * q], sh[2 * q + 1]);
      return reinterpret_cast<const vec_t&>(o);
    }
    float f[VEC];
    vec_to_f32<T>(v, f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = a.no_act ? f[e] * sc[e] + sh[e] : relu6f(f[e] * sc[e] + sh[e]);
    return f32_to_vec<T>(f);
  };
  vec_t zero;
#pragma unroll
  for (int e = 0; e < VEC; ++e) zero[e] = (T)0.f;

  // BWD: the forward input at the output pixel, fetched PF rows ahead like the input rows
  const T* bx = BWD ? reinterpret_cast<const T*>(a.bx) + (size_t)b * a.H * a.W * a.C + c0 : nullptr;
  vec_t preb[PF];
  float bsc[VEC], bsh[VEC], psum2[VEC];
  if constexpr (BWD) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      bsc[e] = a.bas[(size_t)b * a.C + c0 + e];
      bsh[e] = a.bab[(size_t)b * a.C + c0 + e];
      psum2[e] = 0.f;
    }
  }
  auto issue_b = [&](int r, vec_t& v) {  // output row r - 2 is produced at iteration r
    if constexpr (BWD) {
      if (r >= 2 && r < nrows) v = ld_vec<T>(bx + ((size_t)(y0 + r - 2) * a.W + x0 + xl) * a.C);
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    issue(j, pre[j], preh[j]);
    issue_b(j, preb[j]);
  }

  // acc[o % 3] accumulates output row o.  The row loop is unrolled 12-fold (lcm of the prefetch depth 4
  // and the 3 accumulator roles) so that accumulator indices, prefetch slots and the ring buffer parity
  // are all compile-time constants: no per-row register rotation.
  float acc[3][VEC], psum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[0][e] = acc[1][e] = acc[2][e] = psum[e] = 0.f;

  static_assert(PF == 4, "the 12-row unroll below assumes 4 prefetch slots");
  for (int r0 = 0; r0 < nrows; r0 += 12) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int r = r0 + j;
      if (r >= nrows) break;  // uniform over the workgroup
      const int gy = y0 - 1 + r;
      const bool row_ok = gy >= 0 && gy < a.H;
      if constexpr (std::is_same<T, bf16_t>::value) {
        // bf16 has no mixed-precision FMA: left alone, the compiler hoists the 72 weight conversions out of the row
        // loop (forward 216 VGPRs, backward 256 = one wave per SIMD).  Making the packed weights opaque once per row
        // keeps them packed (36 VGPRs) and re-converts at use: more VALU, one more wave per SIMD.
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          u32x4& wq = reinterpret_cast<u32x4&>(w[t]);
          asm volatile("" : "+v"(wq));
        }
      }
      vec_t* buf = ring[j & 1];  // r0 is a multiple of 12, so r & 1 == j & 1, r % 4 == j % 4, r % 3 == j % 3
      buf[(xl + 1) * 8 + cl] = (row_ok && col_ok) ? ((dbg & 2) ? pre[j % 4] : activate(pre[j % 4])) : zero;
      if (is_halo) buf[hslot * 8 + cl] = (row_ok && hx_ok) ? activate(preh[j % 4]) : zero;
      issue(r + PF, pre[j % 4], preh[j % 4]);
      wg_barrier();
      vec_t f[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) f[kx] = buf[(xl + kx) * 8 + cl];
      float* a2 = acc[(j + 1) % 3];  // ky = 2 -> output row r-2
      float* a1 = acc[(j + 2) % 3];  // ky = 1 -> output row r-1
      float* a0 = acc[j % 3];        // ky = 0 -> output row r
      if (dbg & 1) {  // ablation: no multiply-accumulate (timing experiments only)
#pragma unroll
        for (int e = 0; e < VEC; ++e) a2[e] += (float)f[1][e];
      } else if constexpr (std::is_same<T, half_t>::value) {
        // fp16: weights and data stay packed; 72 v_fma_mix_f32 per row
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const u32x4 fq = reinterpret_cast<const u32x4&>(f[kx]);
          const u32x4 w2 = reinterpret_cast<const u32x4&>(w[6 + kx]);
          const u32x4 w1 = reinterpret_cast<const u32x4&>(w[3 + kx]);
          const u32x4 w0 = reinterpret_cast<const u32x4&>(w[0 + kx]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            fma_mix_f16<0>(a2[2 * q], w2[q], fq[q]); fma_mix_f16<1>(a2[2 * q + 1], w2[q], fq[q]);
            fma_mix_f16<0>(a1[2 * q], w1[q], fq[q]); fma_mix_f16<1>(a1[2 * q + 1], w1[q], fq[q]);
            fma_mix_f16<0>(a0[2 * q], w0[q], fq[q]); fma_mix_f16<1>(a0[2 * q + 1], w0[q], fq[q]);
          }
        }
      } else {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            a2[e] += (float)w[6 + kx][e] * (float)f[kx][e];
            a1[e] += (float)w[3 + kx][e] * (float)f[kx][e];
            a0[e] += (float)w[0 + kx][e] * (float)f[kx][e];
          }
      }
      if (r >= 2) {
        if constexpr (BWD) {
          float hb[VEC], dz[VEC];
          vec_to_f32<T>(preb[j % 4], hb);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float z = hb[e] * bsc[e] + bsh[e];
            dz[e] = round_to<T>((z > 0.f && z < 6.f) ? a2[e] : 0.f);
            psum[e] += dz[e];
            psum2[e] += dz[e] * hb[e];
          }
          st_f32<T>(out + ((size_t)(y0 + r - 2) * a.W + x0 + xl) * a.C, dz);
          if (((r - 2) & 7) == 7) {  // uniform: an 8-row segment is complete
            pool_segment_flush<VEC>(psum, red + (((r - 2) >> 3) * (NT / 64) + (tid >> 6)) * CC, tid & 63);
            pool_segment_flush<VEC>(psum2, red + ((8 + ((r - 2) >> 3)) * (NT / 64) + (tid >> 6)) * CC, tid & 63);
          }
        } else {
          vec_t ov = f32_to_vec<T>(a2);
          if (!RAGGED || (col_ok && y0 + r - 2 < a.H)) {
            st_vec_pol<T>(out + ((size_t)(y0 + r - 2) * a.W + x0 + xl) * a.C, ov, a.nt != 0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) psum[e] += (float)ov[e];
          }
          if ((a.pool || a.pool_tot) && ((r - 2) & 7) == 7)  // uniform: an 8-row pool segment is complete
            pool_segment_flush<VEC>(psum, red + (((r - 2) >> 3) * (NT / 64) + (tid >> 6)) * CC, tid & 63);
        }
      }
      issue_b(r + PF, preb[j % 4]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) a2[e] = 0.f;  // slot becomes the accumulator of output row r+1
    }
  }
  // ---- SE pool partials, one per 8-row segment (layout independent of the strip height)
  if constexpr (BWD) {  // slab[b][tile][{sum dz, sum dz*bx}][C]
    wg_barrier();
    const int ntiles = tiles_x * (a.H / kPoolSegRows);
    float* base = a.bslab + (size_t)b * ntiles * 2 * a.C + chunk_id * CC;
    pool_segments_store<CC, NT>(red, TYL / kPoolSegRows, tid, base, 2 * a.C, ty * (TYL / kPoolSegRows), tiles_x, tx);
    pool_segments_store<CC, NT>(red + 8 * (NT / 64) * CC, TYL / kPoolSegRows, tid, base + a.C, 2 * a.C,
                                ty * (TYL / kPoolSegRows), tiles_x, tx);
  } else if (a.pool_tot) {
    wg_barrier();
    pool_segments_add<CC, NT>(red, TYL / kPoolSegRows, tid, a.pool_tot + (size_t)b * a.C + chunk_id * CC, kPoolFixScale);
  } else if (a.pool) {
    wg_barrier();
    const int ntiles = tiles_x * ((a.H + kPoolSegRows - 1) / kPoolSegRows);
    pool_segments_store<CC, NT>(red, TYL / kPoolSegRows, tid, a.pool + (size_t)b * ntiles * a.C + chunk_id * CC, a.C,
                                ty * (TYL / kPoolSegRows), tiles_x, tx);
  }
}

template <typename T, int TX, int PFV, bool S6 = false>
__global__ void __launch_bounds__(8 * TX, (std::is_same<T, bf16_t>::value ? 3 : 1))
dwconv3x3_kernel(const DwArgs a, const int TYL, const int dbg, const int swap) {
  dw_body<T, TX, PFV, false, S6>(a, TYL, dbg, swap);
}
template <typename T, int TX, int PFV, bool S6 = false>
__global__ void __launch_bounds__(8 * TX) dwconv3x3_ragged_kernel(const DwArgs a, const int TYL, const int dbg, const int swap) {
  dw_body_impl<T, TX, PFV, false, S6, true>(a, TYL, dbg, swap);  // partial strips at the right / bottom edge (W % 8 or H % 8 != 0)
}
template <typename T, int TX, int PFV>
__global__ void __launch_bounds__(8 * TX) dwconv3x3_bwd_kernel(const DwArgs a, const int TYL, const int swap) {
  dw_body<T, TX, PFV, true>(a, TYL, 0, swap);
}

static int g_dw_swap = 0;
void dwconv_swap(int v) { g_dw_swap = v; }
static int g_dw_dbg = 0;
void dwconv_debug(int v) { g_dw_dbg = v; }  // bits 0-1: timing ablations
static int dw_tx(int W) { return (W % 32 == 0) ? 32 : ((W % 16 == 0) ? 16 : (W % 8 == 0 ? 8 : (W > 16 ? 32 : 16))); }
// Strip height: as tall as possible (fewer halo rows) while the launch still has >= 1024 workgroups to
// fill 256 CUs; small batches get shorter strips.  `chunks` = C / channels per WG.  The pool slab does
// not depend on the choice (8-row segments), so results are bitwise independent of the batch size.
int dw_pick_tyl(int B, int H, int W, int chunks) {
  if (H % 8) return 8;  // ragged bottom edge: 8-row strips, the last one partial
  const int tiles_x = (W + dw_tx(W) - 1) / dw_tx(W);
  const int cand[4] = {64, 32, 16, 8};
  for (int i = 0; i < 4; ++i)
    if (H % cand[i] == 0 && (long)tiles_x * (H / cand[i]) * chunks * B >= 1024) return cand[i];
  return 8;
}
int dwconv_ntiles(int H, int W) { return ((H + kPoolSegRows - 1) / kPoolSegRows) * ((W + dw_tx(W) - 1) / dw_tx(W)); }

template <typename T>
static hipError_t launch_dw_t(const DwArgs& a, hipStream_t s) {
  constexpr int CC = 8 * Elem<T>::VEC;
  const bool ragged = a.H % 8 || a.W % 8;  // forward only: partial strips at the right / bottom edge
  if (a.C % CC || a.H < 1 || a.W < 1 || (ragged && a.bx)) return hipErrorInvalidValue;
  const int tx = dw_tx(a.W), tyl = dw_pick_tyl(a.B, a.H, a.W, a.C / CC);
  const int tiles = ((a.W + tx - 1) / tx) * ((a.H + tyl - 1) / tyl);
  dim3 grid(tiles, a.C / CC, a.B);
  if (g_dw_swap) grid = dim3(a.C / CC, tiles, a.B);
  static const std::string names[3] = {std::string("dwconv3x3_kernel<") + TypeName<T>::value + ", 32, 4>",
                                       std::string("dwconv3x3_kernel<") + TypeName<T>::value + ", 16, 4>",
                                       std::string("dwconv3x3_kernel<") + TypeName<T>::value + ", 8, 4>"};
  note_kernel(names[tx == 32 ? 0 : (tx == 16 ? 1 : 2)].c_str());
  if (ragged) {  // forward only (checked above); TX is 16 or 32 here
    note_kernel("dwconv3x3_ragged_kernel");
    if constexpr (sizeof(T) == 2) {
      if (a.s6) {
        if (a.no_act) return hipErrorInvalidValue;
        if (tx == 32) hipLaunchKernelGGL((dwconv3x3_ragged_kernel<T, 32, kDwPF, true>), grid, dim3(256), 0, s, a, tyl, 0, g_dw_swap);
        else hipLaunchKernelGGL((dwconv3x3_ragged_kernel<T, 16, kDwPF, true>), grid, dim3(128), 0, s, a, tyl, 0, g_dw_swap);
        return hipGetLastError();
      }
    } else if (a.s6) {
      return hipErrorInvalidValue;
    }
    if (tx == 32) hipLaunchKernelGGL((dwconv3x3_ragged_kernel<T, 32, kDwPF>), grid, dim3(256), 0, s, a, tyl, 0, g_dw_swap);
    else hipLaunchKernelGGL((dwconv3x3_ragged_kernel<T, 16, kDwPF>), grid, dim3(128), 0, s, a, tyl, 0, g_dw_swap);
    return hipGetLastError();
  }
  if (a.bx) {  // backward instantiation
    if (!a.bas || !a.bab || !a.bslab || a.pool) return hipErrorInvalidValue;
    if (tx == 32) hipLaunchKernelGGL((dwconv3x3_bwd_kernel<T, 32, kDwPF>), grid, dim3(256), 0, s, a, tyl, g_dw_swap);
    else if (tx == 16) hipLaunchKernelGGL((dwconv3x3_bwd_kernel<T, 16, kDwPF>), grid, dim3(128), 0, s, a, tyl, g_dw_swap);
    else hipLaunchKernelGGL((dwconv3x3_bwd_kernel<T, 8, kDwPF>), grid, dim3(64), 0, s, a, tyl, g_dw_swap);
    return hipGetLastError();
  }
  if constexpr (sizeof(T) == 2) {
    if (a.s6) {
      if (a.no_act) return hipErrorInvalidValue;
      if (tx == 32) hipLaunchKernelGGL((dwconv3x3_kernel<T, 32, kDwPF, true>), grid, dim3(256), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
      else if (tx == 16) hipLaunchKernelGGL((dwconv3x3_kernel<T, 16, kDwPF, true>), grid, dim3(128), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
      else hipLaunchKernelGGL((dwconv3x3_kernel<T, 8, kDwPF, true>), grid, dim3(64), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
      return hipGetLastError();
    }
  } else if (a.s6) {
    return hipErrorInvalidValue;
  }
  if (tx == 32) hipLaunchKernelGGL((dwconv3x3_kernel<T, 32, kDwPF>), grid, dim3(256), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
  else if (tx == 16) hipLaunchKernelGGL((dwconv3x3_kernel<T, 16, kDwPF>), grid, dim3(128), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
  else hipLaunchKernelGGL((dwconv3x3_kernel<T, 8, kDwPF>), grid, dim3(64), 0, s, a, tyl, g_dw_dbg & 3, g_dw_swap);
  return hipGetLastError();
}

hipError_t launch_dwconv3x3(int dtype, const DwArgs& a, hipStream_t s) {
  switch (dtype) {
    case 0: return launch_dw_t<float>(a, s);
    case 1: return launch_dw_t<half_t>(a, s);
    case 2: return launch_dw_t<bf16_t>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace llie
