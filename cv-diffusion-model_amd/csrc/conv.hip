// Dense 3x3 convolutions of the denoiser (gfx950).
//   init_conv   (efficient_unet.py:420,553)  Cin=6 -> C0, reads the two fp32 NCHW halves of
//               torch.cat([latents, low_light], 1) (low_light_diffusion.py:222) directly: the concat is virtual.
//   final head  (efficient_unet.py:528-530,600-602)  GroupNorm affine + SiLU fused into the tile load,
//               C0 -> 3, writes the fp32 NCHW noise prediction.
//   Downsample  (efficient_unet.py:367)  3x3 stride 2       } implicit GEMM on MFMA: M = 8 x TW output pixels,
//   Upsample    (efficient_unet.py:383-384) bilinear x2 + 3x3 } N = Cout tile, K = 9 taps x Cin; the upsampled
//               halo patch is interpolated on the fly into LDS, so the 4x tensor never reaches HBM.
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace llie {

// =============================================================================================
// init_conv: one thread = one output pixel x 32 output channels (K = 54 is too small for MFMA tiles
// to pay; weights are wave-uniform and come through the scalar cache).
// Weight layout here: [k = ci*9 + tap][Cout] fp32 (repacked at load).
template <typename T>
__global__ void __launch_bounds__(256) init_conv_kernel(const InitConvArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int MAXC = 8;
  __shared__ float patch[MAXC][18][19];
  __shared__ float red[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + 15) / 16;  // edge tiles may be partly empty (image sizes that are not a multiple of 16)
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = tx * 16, y0 = ty * 16;
  const int Cin = a.c0 + a.c1;
  const size_t plane = (size_t)a.H * a.W;
  for (int i = tid; i < Cin * 18 * 18; i += 256) {
    const int ci = i / 324, r = i % 324;
    const int py = r / 18, px = r % 18;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      const float* src = ci < a.c0 ? a.x0 + ((size_t)b * a.c0 + ci) * plane : a.x1 + ((size_t)b * a.c1 + (ci - a.c0)) * plane;
      v = src[(size_t)gy * a.W + gx];
    }
    patch[ci][py][px] = v;
  }
  wg_barrier();
  const int py = tid >> 4, px = tid & 15;
  const bool ok = y0 + py < a.H && x0 + px < a.W;
  const float* __restrict__ w = a.w;
  T* out = reinterpret_cast<T*>(a.out) + (((size_t)b * a.H + y0 + py) * a.W + x0 + px) * a.Cout;
  const int ntiles = tiles_x * ((a.H + 15) / 16);
  for (int oc0 = 0; oc0 < a.Cout; oc0 += 32) {
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = a.bias[oc0 + o];
    for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float v = patch[ci][py + tap / 3][px + tap % 3];
        const float* wr = w + (size_t)(ci * 9 + tap) * a.Cout + oc0;
#pragma unroll
        for (int o = 0; o < 32; ++o) acc[o] += wr[o] * v;
      }
    }
    float q[32];
#pragma unroll
    for (int o = 0; o < 32; o += VEC) {
      typename Elem<T>::vec_t ov = f32_to_vec<T>(acc + o);
      if (ok) st_vec<T>(out + oc0 + o, ov);
#pragma unroll
      for (int e = 0; e < VEC; ++e) q[o + e] = ok ? (float)ov[e] : 0.f;
    }
    if (a.stats) {
#pragma unroll
      for (int o = 0; o < 32; ++o) {
        const float s1 = wave_sum(q[o]), s2 = wave_sum(q[o] * q[o]);
        if (lane == 0) {
          red[wave][0][o] = s1;
          red[wave][1][o] = s2;
        }
      }
      wg_barrier();
      if (tid < 64) {
        const int which = tid >> 5, o = tid & 31;
        const float t = red[0][which][o] + red[1][which][o] + red[2][which][o] + red[3][which][o];
        a.stats[((size_t)(b * ntiles + blockIdx.x) * 2 + which) * a.Cout + oc0 + o] = t;
      }
      wg_barrier();
    }
  }
}
// ---------------------------------------------------------------------------------------------
// init_conv on MFMA (2-byte T): im2col without a gather.  The 18x18 halo patch is staged in LDS as
// [y][x][8 channels] (16 B per pixel; channels >= Cin are zero), and K is ordered (tap, channel8), so
// one ds_read_b128 per k-step IS the lane's A fragment: k-step s covers taps 2s (lane half 0) and 2s+1
// (lane half 1); tap 9 does not exist and is fed zeros.  5 k-steps (K = 80 >= 72).  Weights come
// pre-packed as [s][half][Cout][8] T.  One wave computes two 32-pixel x 32-channel tiles per 32 output
// channels; its fp32 tile goes through a wave-private LDS patch so stores are full 64-byte pixel rows.
template <typename T>
__global__ void __launch_bounds__(256) init_conv_mfma_kernel(const InitConvArgs a) {
  typedef typename Elem<T>::vec_t vec_t;
  constexpr int TH = 8, TW = 32, PH = TH + 2, PWD = TW + 3;  // 8 x 32 output tile: 136-byte runs of the fp32 input planes
  __shared__ vec_t patch[PH * PWD + 1];   // +1: an all-zero slot for the missing 10th tap
  __shared__ float ctile[4][32 * 33];
  __shared__ float red[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + TW - 1) / TW;  // the last tile of a row may be partly empty (W % 32 != 0)
  // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs, one L2 each): an XCD gets a run of consecutive tiles,
  // so horizontal neighbours share the lines their halo columns straddle in ONE L2.  With neighbours on different XCDs every
  // 136-byte patch row cost three 128-byte lines per XCD: 275 MB of HBM traffic per launch against 184.5 MB algorithmic.
  const int tile_id = xcd_tile_order(blockIdx.x, gridDim.x);
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x, b = blockIdx.y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int Cin = a.c0 + a.c1;
  const size_t plane = (size_t)a.H * a.W;
  // first block's weight fragments and bias up front: their (L2) latency runs under the patch staging
  vec_t wf0[5];
  {
    const T* wp0 = reinterpret_cast<const T*>(a.wp);
#pragma unroll
    for (int s = 0; s < 5; ++s) wf0[s] = ld_vec<T>(wp0 + ((size_t)(s * 2 + (lane >> 5)) * a.Cout + (lane & 31)) * 8);
  }
  const float bias0 = a.bias[lane & 31];
  for (int i = tid; i < PH * PWD + 1; i += 256) {
    const int py = i / PWD, px = i % PWD;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    vec_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
    if (i < PH * PWD && px < TW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      for (int ci = 0; ci < Cin; ++ci) {
        const float* src = ci < a.c0 ? a.x0 + ((size_t)b * a.c0 + ci) * plane : a.x1 + ((size_t)b * a.c1 + (ci - a.c0)) * plane;
        v[ci] = (T)src[(size_t)gy * a.W + gx];
      }
    }
    patch[i] = v;
  }
  wg_barrier();
  const int r = lane & 31, h = lane >> 5;
  const T* wp = reinterpret_cast<const T*>(a.wp);
  T* out = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.Cout;
  const int ntiles = tiles_x * (a.H / TH);
  float* ct = ctile[wave];
  for (int oc0 = 0; oc0 < a.Cout; oc0 += 32) {
    vec_t wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[s] = oc0 ? ld_vec<T>(wp + ((size_t)(s * 2 + h) * a.Cout + oc0 + r) * 8) : wf0[s];
    const float bias = oc0 ? a.bias[oc0 + r] : bias0;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = wave * 2 + t;                 // 32-pixel group: row g of the 8 x 32 tile
      const int py = g, px = r;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + h;
        const int idx = tap < 9 ? (py + tap / 3) * PWD + px + tap % 3 : PH * PWD;
        const vec_t av = patch[idx];
        if constexpr (std::is_same<T, half_t>::value) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wf[s], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wf[s], acc, 0, 0, 0);
      }
      // D[pixel (rows in registers)][channel = r]: + bias, round, stats, then transpose through LDS
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = (float)(T)(acc[q] + bias);
        if (x0 + mfma_row(q, lane) >= a.W) v = 0.f;  // pixel past the image (never stored either)
        s1 += v;
        s2 += v * v;
        ct[mfma_row(q, lane) * 33 + r] = v;
      }
      // wave-private tile: no block barrier needed, only LDS ordering within the wave
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      {
        const int p = lane >> 1, half = lane & 1;        // pixel in group, 16-channel half
        float f[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = ct[p * 33 + half * 16 + e];
        const int oy = y0 + g, ox = x0 + p;
        T* dst = out + ((size_t)oy * a.W + ox) * a.Cout + oc0 + half * 16;
        if (ox < a.W) {
          st_vec<T>(dst, f32_to_vec<T>(f));
          st_vec<T>(dst + 8, f32_to_vec<T>(f + 8));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (a.stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32) {
        red[wave][0][lane] = s1;
        red[wave][1][lane] = s2;
      }
      wg_barrier();
      if (tid < 64) {
        const int which = tid >> 5, o = tid & 31;
        const float tt = red[0][which][o] + red[1][which][o] + red[2][which][o] + red[3][which][o];
        a.stats[((size_t)(b * ntiles + tile_id) * 2 + which) * a.Cout + oc0 + o] = tt;
      }
      wg_barrier();
    }
  }
}
// statistics partials per image: the 2-byte engines' MFMA kernel works on 8 x 32 tiles, the fp32 kernel on 16 x 16
int init_conv_ntiles(int H, int W, bool mfma) { return mfma ? (H / 8) * ((W + 31) / 32) : ((H + 15) / 16) * ((W + 15) / 16); }
hipError_t launch_init_conv(int dtype, const InitConvArgs& a, hipStream_t s) {
  if (a.H % 8 || a.W % 8 || a.Cout % 32 || a.c0 + a.c1 > 8) return hipErrorInvalidValue;
  const bool mfma = a.wp && dtype != 0;
  dim3 grid(init_conv_ntiles(a.H, a.W, mfma), a.B);  // 256-pixel tiles: 16 x 16 (VALU kernel) or 8 x 32 (MFMA kernel)
  switch (dtype) {
    case 0: hipLaunchKernelGGL(init_conv_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1:
      if (a.wp) hipLaunchKernelGGL(init_conv_mfma_kernel<half_t>, grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(init_conv_kernel<half_t>, grid, dim3(256), 0, s, a);
      break;
    case 2:
      if (a.wp) hipLaunchKernelGGL(init_conv_mfma_kernel<bf16_t>, grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(init_conv_kernel<bf16_t>, grid, dim3(256), 0, s, a);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// =============================================================================================
// final head: affine (GroupNorm) + SiLU applied once per element while staging a 18x18 halo tile of
// 32 channels into LDS (channel-major, so a wave's pixel-consecutive reads are conflict free), then
// one thread = one pixel x Cout(<=4) outputs.  Weight layout: [tap][C][4] fp32, zero padded (repacked at load).
template <typename T>
__global__ void __launch_bounds__(256) final_conv_kernel(const FinalConvArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int VPP = 32 / VEC;  // vectors per pixel per 32-channel chunk
  __shared__ float patch[32][18][19];
  const int tid = threadIdx.x;
  const int tiles_x = (a.W + 15) / 16;  // edge tiles may be partly empty
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = tx * 16, y0 = ty * 16;
  const int py = tid >> 4, px = tid & 15;
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.C;
  const float* __restrict__ w = a.w;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int cc = 0; cc < a.C; cc += 32) {
    if (cc) wg_barrier();
    for (int i = tid; i < 18 * 18 * VPP; i += 256) {
      const int pix = i / VPP, cv = (i % VPP) * VEC;
      const int ppy = pix / 18, ppx = pix % 18;
      const int gy = y0 + ppy - 1, gx = x0 + ppx - 1;
      float f[VEC];
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        ld_f32<T>(in + ((size_t)gy * a.W + gx) * a.C + cc + cv, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int c = cc + cv + e;
          f[e] = siluf(f[e] * a.as[(size_t)b * a.C + c] + a.ab[(size_t)b * a.C + c]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) patch[cv + e][ppy][ppx] = f[e];
    }
    wg_barrier();
    for (int ci = 0; ci < 32; ++ci) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float v = patch[ci][py + tap / 3][px + tap % 3];
        const float* wr = w + ((size_t)tap * a.C + cc + ci) * 4;
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] += wr[o] * v;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
    if (o < a.Cout && y0 + py < a.H && x0 + px < a.W) a.out[(((size_t)b * a.Cout + o) * a.H + y0 + py) * a.W + x0 + px] = acc[o] + a.bias[o];
}
// ---------------------------------------------------------------------------------------------
// final head on MFMA (2-byte T) with the LCM scheduler step fused into the epilogue.
// Roles are swapped relative to a usual conv-GEMM: the (zero-padded) weights are the A operand
// (rows = output channel, only rows 0..2 are non-zero) and the activated halo patch is the B operand
// (columns = pixels), so after the K loop lane p of lane-half 0 holds the 3 outputs of ITS pixel in
// acc[0..2]: the epilogue (bias, LCMScheduler.step lcm_scheduler.py:204-242, clamp
// low_light_diffusion.py:240) is per-lane and its fp32 NCHW accesses are 64-byte row segments.
// K order per 32-channel chunk: k-step = (tap, 16-channel half); lane half h takes 8 channels.
// Weights pre-packed as [chunk][18][2][4][8] T (output channel padded to 4).
template <typename T, int TW>
__global__ void __launch_bounds__(256, 4) final_conv_mfma_kernel(const FinalConvArgs a) {
  typedef typename Elem<T>::vec_t vec_t;
  // output tile (256 / TW) x TW pixels.  TW = 16 is used: 8 x 32 tiles (128-byte instead of 64-byte runs on the fp32 planes
  // of the scheduler step) measured 210 us against 163 us at B = 32 -- this kernel is not bound by its coalescing
  constexpr int TH = 256 / TW, PH = TH + 2, PWD = TW + 3, PIX = 40;  // pixel pitch 80 B: conflict-free ds_read_b128 (cf. TilePitch)
  __shared__ __align__(16) T patch[(PH * PWD + 1) * PIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + TW - 1) / TW;  // edge tiles may be partly empty
  const int tile_id = xcd_tile_order(blockIdx.x, gridDim.x);  // horizontal neighbours on one XCD (see init_conv_mfma_kernel)
  const int tx = tile_id % tiles_x, ty = tile_id / tiles_x, b = blockIdx.y;
  const int x0 = tx * TW, y0 = ty * TH;
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.C;
  const T* wp = reinterpret_cast<const T*>(a.wp);
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  vec_t zero;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero[e] = (T)0.f;

  // The scheduler step's operands (this lane's pixel of the fp32 planes) and the first chunk's weight fragments are
  // fetched before anything else: their latency then runs under the patch staging instead of after the MFMAs.
  const size_t plane = (size_t)a.H * a.W;
  float pre_x[2][4], pre_n[2][4];
  if (a.fuse_step && !h) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = wave * 2 + t;
      const int y = y0 + g * (32 / TW) + r / TW, x = x0 + r % TW;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        pre_x[t][o] = pre_n[t][o] = 0.f;
        if (o < a.Cout && y < a.H && x < a.W) {
          const size_t idx = ((size_t)b * a.Cout + o) * plane + (size_t)y * a.W + x;
          pre_x[t][o] = a.sample[idx];
          if (!a.coef.is_last) pre_n[t][o] = a.noise[idx];
        }
      }
    }
  }
  // A operand = the weights: rows r < 4 of every fragment, zero elsewhere.  They used to live in 72 registers per lane
  // (18 k-steps x 16 bytes), which held the kernel at two waves per SIMD -- a streaming kernel that waits on its patch loads;
  // now the 2.3 KB of real rows per 32-channel chunk sit in LDS ([ks][h][4 rows][8 T], then one zero row) and each MFMA
  // reads its fragment right before use (lanes with r >= 4 all read the zero row: a broadcast).
  __shared__ __align__(16) T wsh[(18 * 2 * 4 + 1) * 8];
  auto stage_wf = [&](int chunk) {
    if (tid < 18 * 2 * 4) *reinterpret_cast<vec_t*>(wsh + tid * 8) = ld_vec<T>(wp + ((size_t)chunk * 18 * 2 * 4 + tid) * 8);
    if (tid == 18 * 2 * 4) *reinterpret_cast<vec_t*>(wsh + tid * 8) = zero;
  };
  stage_wf(0);
  const T* wrow = wsh + (r < 4 ? h * 4 + r : 18 * 2 * 4) * 8;  // + ks * 64 for rows r < 4
  const int wstep = r < 4 ? 64 : 0;

  for (int cc = 0; cc < a.C; cc += 32) {
    if (cc) {
      wg_barrier();  // everyone done with the previous chunk's patch and weights
      stage_wf(cc >> 5);
    }
    // stage the activated 10x34x32 patch (GroupNorm affine + SiLU applied once per element); a thread's
    // 8-channel slice is loop invariant (256 % 4 == 0), so its affine pairs live in registers
    const int cv = (tid & 3) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = a.as[(size_t)b * a.C + cc + cv + e];
      sh[e] = a.ab[(size_t)b * a.C + cc + cv + e];
    }
    for (int i = tid; i < (PH * PWD + 1) * 4; i += 256) {
      const int pix = i >> 2;
      const int ppy = pix / PWD, ppx = pix % PWD;
      const int gy = y0 + ppy - 1, gx = x0 + ppx - 1;
      vec_t v = zero;
      if (pix < PH * PWD && ppx < TW + 2 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        float f[8];
        ld_f32<T>(in + ((size_t)gy * a.W + gx) * a.C + cc + cv, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = siluf_fast(f[e] * sc[e] + sh[e]);
        v = f32_to_vec<T>(f);
      }
      *reinterpret_cast<vec_t*>(patch + pix * PIX + cv) = v;
    }
    wg_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = wave * 2 + t;
      const int py = g * (32 / TW) + r / TW, px = r % TW;
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        const int tap = ks >> 1, q = ks & 1;
        // one tap row (6 k-steps, 12 fragment reads) at a time: left alone, hipcc hoists all 36 reads of a block (144 registers)
        if (ks % 6 == 0) __builtin_amdgcn_sched_barrier(0);
        const vec_t bv = *reinterpret_cast<const vec_t*>(patch + ((py + tap / 3) * PWD + px + tap % 3) * PIX + q * 16 + h * 8);
        const vec_t wv = *reinterpret_cast<const vec_t*>(wrow + ks * wstep);
        if constexpr (std::is_same<T, half_t>::value) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, bv, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, bv, acc[t], 0, 0, 0);
      }
    }
  }
  if (h) return;  // rows 4..7 of D (lane half 1) are padding
  {
#pragma clang fp contract(off)  // scheduler step: keep the reference's operation order (no fused multiply-adds)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int g = wave * 2 + t;
    const int y = y0 + g * (32 / TW) + r / TW, x = x0 + r % TW;
    if (y >= a.H || x >= a.W) continue;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o >= a.Cout) break;
      const size_t idx = ((size_t)b * a.Cout + o) * plane + (size_t)y * a.W + x;
      const float e = acc[t][o] + a.bias[o];
      if (a.out) a.out[idx] = e;
      if (a.fuse_step) {
        const float xv = pre_x[t][o];
        float x0v;
        if (a.coef.vpred) x0v = a.coef.sa * xv - a.coef.sb * e;
        else x0v = (xv - a.coef.sb * e) / a.coef.sa;
        if (a.coef.clamp_x0) x0v = fminf(fmaxf(x0v, -1.f), 1.f);
        float pv = x0v;
        if (!a.coef.is_last) pv = a.coef.sap * x0v + a.coef.sbp * pre_n[t][o];
        a.prev[idx] = pv;
        if (a.clamped) a.clamped[idx] = fminf(fmaxf(pv, -1.f), 1.f);
      }
    }
  }
  }
}
hipError_t launch_final_conv(int dtype, const FinalConvArgs& a, hipStream_t s) {
  if (a.H % 8 || a.W % 8 || a.C % 32 || a.Cout > 4) return hipErrorInvalidValue;
  if (a.fuse_step && (!a.wp || dtype == 0 || !a.sample || !a.prev || (!a.coef.is_last && !a.noise))) return hipErrorInvalidValue;
  if (!a.fuse_step && !a.out) return hipErrorInvalidValue;
  dim3 grid(((a.H + 15) / 16) * ((a.W + 15) / 16), a.B);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(final_conv_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1:
      if (a.wp) hipLaunchKernelGGL((final_conv_mfma_kernel<half_t, 16>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(final_conv_kernel<half_t>, grid, dim3(256), 0, s, a);
      break;
    case 2:
      if (a.wp) hipLaunchKernelGGL((final_conv_mfma_kernel<bf16_t, 16>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(final_conv_kernel<bf16_t>, grid, dim3(256), 0, s, a);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// =============================================================================================
// Implicit-GEMM 3x3 conv on MFMA.  MODE 0: stride 2, pad 1.  MODE 1: bilinear x2 (align_corners=False:
// src = (dst+0.5)/2 - 0.5 clamped at 0, upper neighbour clamped at n-1) then stride 1, pad 1.
// STAMP: diagnostic build (llie_tune("conv_stamp", 1)): s_memtime sums per wave into a.stamps[wave][kConvStamps] = {0 patch
// commit (bounds / bilinear blend + ds_write), 1 first W tile staged + barrier, 2 operand ds_reads of a tap, 3 its MFMAs,
// 4 next W tile staged + prefetch issued, 5 the tap's barrier, 6 epilogue}; never used in production.
constexpr int kConvStamps = 7;
template <typename T, int MODE, int TW, int BN, int WM, int WN, bool RAGGED = false, bool STAMP = false>
__global__ void __launch_bounds__(WM* WN * 64) conv3x3_kernel(const Conv3Args a) {
  constexpr int NT = WM * WN * 64;
  unsigned long long tk[kConvStamps] = {}, t_prev = 0;
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      if (slot >= 0) tk[slot] += now - t_prev;
      t_prev = now;
    }
  };
  constexpr int TH = 8, BM = TH * TW;
  constexpr int VEC = Elem<T>::VEC, VPR = 32 / VEC, PITCH = TilePitch<T>::value;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int PH = MODE == 0 ? 2 * TH + 1 : TH + 2;
  constexpr int PW = MODE == 0 ? 2 * TW + 1 : TW + 2;
  constexpr int B_VECS = BN * VPR, B_PER = (B_VECS + NT - 1) / NT;
  constexpr int CP = BN + 4;
  typedef typename Elem<T>::vec_t vec_t;
  static_assert(NT % VPR == 0, "mapping");

  extern __shared__ __align__(16) unsigned char smem[];
  T* sP = reinterpret_cast<T*>(smem);
  T* sB = sP + PH * PW * PITCH;  // [2][BN][PITCH]
  float* sC = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int Ho = MODE == 0 ? a.Hi / 2 : (MODE == 1 ? a.Hi * 2 : a.Hi), Wo = MODE == 0 ? a.Wi / 2 : (MODE == 1 ? a.Wi * 2 : a.Wi);
  // ragged outputs (Ho % 8 or Wo % TW != 0: image sizes that are not a multiple of 64): edge tiles are partly empty -- pixels past
  // the image are neither stored nor counted (the patch loader already reads everything outside as zero)
  const int nb = a.Cout / BN, tiles_x = RAGGED ? (Wo + TW - 1) / TW : Wo / TW, tiles = tiles_x * (RAGGED ? (Ho + TH - 1) / TH : Ho / TH);
  int bid = blockIdx.x;
  const int ntile = bid % nb; bid /= nb;
  const int tile = bid % tiles, b = bid / tiles;
  const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
  const int n0 = ntile * BN;
  const int kv = (tid % VPR) * VEC;
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.Hi * a.Wi * a.Cin;
  const T* wbase = reinterpret_cast<const T*>(a.w);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane patch offsets of this lane's A rows (tap (0,0)); tap (dy,dx) adds (dy*PW+dx)*PITCH
  // MODE 1, TW = 16: GEMM row m of a tile is pixel (m / 16, (m % 16 + 14 (m / 16 & 1)) % 16) -- odd tile rows rotated by two
  // pixels.  A ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and + 32), and with
  // the 80-byte pixel pitch a group is conflict-free iff its 16 patch pixels differ mod 16; lanes 16..31 sit one patch row
  // (18 pixels) below lanes 0..15, so without the rotation pixels 12, 13 (4, 5) of a group collide with 22 + 6, 7: every A
  // read two-way conflicted.  (Stride 2 reads every second pixel: two-way whatever the order.)
  auto tile_px = [](int m, int& py, int& px) {
    py = m / TW;
    px = m % TW;
    if (MODE == 1 && TW == 16) px = (px + 14 * (py & 1)) & 15;
  };
  int arow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = (wm * MI + i) * 32 + (lane & 31);
    int py, px;
    tile_px(m, py, px);
    arow[i] = (MODE == 0 ? (2 * py * PW + 2 * px) : (py * PW + px)) * PITCH + (lane >> 5) * 16;
  }

  vec_t rb[B_PER];
  auto prefetch_w = [&](int tap, int c0) {
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_VECS % NT == 0 || idx < B_VECS) {
        const int n = idx / VPR;
        rb[i] = ld_vec<T>(wbase + ((size_t)tap * a.Cout + n0 + n) * a.Cin + c0 + kv);
      }
    }
  };
  auto stage_w = [&](int buf) {  // two W tiles in LDS: tap t computes from buffer t & 1 while t + 1 is being written
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int idx = tid + i * NT;
      if (B_VECS % NT == 0 || idx < B_VECS) st_vec<T>(sB + buf * (BN * PITCH) + (idx / VPR) * PITCH + kv, rb[i]);
    }
  };

  // Input patch of one 32-channel chunk: the global loads are issued one chunk ahead (issue_patch keeps the raw
  // vectors in registers while the nine taps of the current chunk run) and turned into the LDS patch afterwards
  // (commit_patch: bounds -> zeros, MODE 1 interpolates its four neighbours).
  constexpr int P_ITEMS = (PH * PW * VPR + NT - 1) / NT, P_LOADS = MODE == 1 ? 4 : 1;
  vec_t praw[P_ITEMS][P_LOADS];
  struct Src { bool ok; int o00, o01, o10, o11; float ly1, lx1; };
  auto patch_src = [&](int i) {
    Src r{false, 0, 0, 0, 0, 0.f, 0.f};
    if (i >= PH * PW * VPR) return r;
    const int pix = i / VPR;
    const int ppy = pix / PW, ppx = pix % PW;
    if (MODE == 0 || MODE == 2) {
      const int gy = (MODE == 0 ? 2 * oy0 : oy0) - 1 + ppy, gx = (MODE == 0 ? 2 * ox0 : ox0) - 1 + ppx;
      r.ok = gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi;
      r.o00 = (gy * a.Wi + gx) * a.Cin;
    } else {
      const int uy = oy0 - 1 + ppy, ux = ox0 - 1 + ppx;  // coordinates in the upsampled image
      r.ok = uy >= 0 && uy < Ho && ux >= 0 && ux < Wo;
      float sy = ((float)uy + 0.5f) * 0.5f - 0.5f, sx = ((float)ux + 0.5f) * 0.5f - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      const int iy0 = (int)sy, ix0 = (int)sx;
      const int iy1 = min(iy0 + 1, a.Hi - 1), ix1 = min(ix0 + 1, a.Wi - 1);
      r.ly1 = sy - (float)iy0; r.lx1 = sx - (float)ix0;
      r.o00 = (iy0 * a.Wi + ix0) * a.Cin; r.o01 = (iy0 * a.Wi + ix1) * a.Cin;
      r.o10 = (iy1 * a.Wi + ix0) * a.Cin; r.o11 = (iy1 * a.Wi + ix1) * a.Cin;
    }
    return r;
  };
  auto issue_item = [&](int it, int c0) {
    {
      const Src r = patch_src(tid + it * NT);
      if (r.ok) {
        const T* base = in + c0 + kv;
        praw[it][0] = ld_vec<T>(base + r.o00);
        if (MODE == 1) {
          praw[it][P_LOADS > 1 ? 1 : 0] = ld_vec<T>(base + r.o01);
          praw[it][P_LOADS > 2 ? 2 : 0] = ld_vec<T>(base + r.o10);
          praw[it][P_LOADS > 3 ? 3 : 0] = ld_vec<T>(base + r.o11);
        }
      }
    }
  };
  auto commit_item = [&](int it) {
    {
      const int i = tid + it * NT;
      if (i >= PH * PW * VPR) return;
      const Src r = patch_src(i);
      vec_t v;
      if (!r.ok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
      } else if (MODE == 1) {
        const float ly0 = 1.f - r.ly1, lx0 = 1.f - r.lx1;
        float f00[VEC], f01[VEC], f10[VEC], f11[VEC], f[VEC];
        vec_to_f32<T>(praw[it][0], f00);
        vec_to_f32<T>(praw[it][P_LOADS > 1 ? 1 : 0], f01);
        vec_to_f32<T>(praw[it][P_LOADS > 2 ? 2 : 0], f10);
        vec_to_f32<T>(praw[it][P_LOADS > 3 ? 3 : 0], f11);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          f[e] = ly0 * (lx0 * f00[e] + r.lx1 * f01[e]) + r.ly1 * (lx0 * f10[e] + r.lx1 * f11[e]);
        v = f32_to_vec<T>(f);
      } else {
        v = praw[it][0];
      }
      st_vec<T>(sP + (i / VPR) * PITCH + kv, v);
    }
  };

  // measured: the look-ahead pays everywhere except the 64-channel upsampling conv (4 x 3 raw vectors per thread
  // cost it an occupancy step: 294 -> 365 us), which keeps load-then-use
  constexpr bool PREF = !(MODE == 1 && BN == 64);
  auto issue_patch = [&](int c0) {
#pragma unroll
    for (int it = 0; it < P_ITEMS; ++it) issue_item(it, c0);
  };
  if (PREF) issue_patch(0);
  stamp(-1);
  for (int c0 = 0; c0 < a.Cin; c0 += 32) {
    prefetch_w(0, c0);
    if constexpr (PREF) {  // this chunk's patch: its loads were issued during the previous chunk's taps
#pragma unroll
      for (int it = 0; it < P_ITEMS; ++it) commit_item(it);
    } else {  // load-then-use, one item at a time (lowest register footprint); MODE 1 only
      for (int i = tid; i < PH * PW * VPR; i += NT) {
        const int pix = i / VPR;
        const int ppy = pix / PW, ppx = pix % PW;
        vec_t v;
        const int uy = oy0 - 1 + ppy, ux = ox0 - 1 + ppx;  // coordinates in the upsampled image
        if (uy >= 0 && uy < Ho && ux >= 0 && ux < Wo) {
          float sy = ((float)uy + 0.5f) * 0.5f - 0.5f, sx = ((float)ux + 0.5f) * 0.5f - 0.5f;
          sy = sy < 0.f ? 0.f : sy;
          sx = sx < 0.f ? 0.f : sx;
          const int iy0 = (int)sy, ix0 = (int)sx;
          const int iy1 = min(iy0 + 1, a.Hi - 1), ix1 = min(ix0 + 1, a.Wi - 1);
          const float ly1 = sy - (float)iy0, lx1 = sx - (float)ix0;
          const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
          float f00[VEC], f01[VEC], f10[VEC], f11[VEC], f[VEC];
          ld_f32<T>(in + ((size_t)iy0 * a.Wi + ix0) * a.Cin + c0 + kv, f00);
          ld_f32<T>(in + ((size_t)iy0 * a.Wi + ix1) * a.Cin + c0 + kv, f01);
          ld_f32<T>(in + ((size_t)iy1 * a.Wi + ix0) * a.Cin + c0 + kv, f10);
          ld_f32<T>(in + ((size_t)iy1 * a.Wi + ix1) * a.Cin + c0 + kv, f11);
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            f[e] = ly0 * (lx0 * f00[e] + lx1 * f01[e]) + ly1 * (lx0 * f10[e] + lx1 * f11[e]);
          v = f32_to_vec<T>(f);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
        }
        st_vec<T>(sP + pix * PITCH + kv, v);
      }
    }
    stamp(0);
    stage_w(0);
    prefetch_w(1, c0);
    wg_barrier();  // patch and W tile of tap 0 visible
    if (PREF && c0 + 32 < a.Cin) issue_patch(c0 + 32);
    stamp(1);
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = ((tap / 3) * PW + (tap % 3)) * PITCH;
      T fa[MI][16], fb[NI][16];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const T* p = sP + arow[i] + toff;
#pragma unroll
        for (int q = 0; q < 16 / VEC; ++q) *reinterpret_cast<vec_t*>(&fa[i][q * VEC]) = *reinterpret_cast<const vec_t*>(p + q * VEC);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const T* p = sB + (tap & 1) * (BN * PITCH) + ((wn * NI + j) * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
#pragma unroll
        for (int q = 0; q < 16 / VEC; ++q) *reinterpret_cast<vec_t*>(&fb[j][q * VEC]) = *reinterpret_cast<const vec_t*>(p + q * VEC);
      }
      if constexpr (STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(2);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) Mfma<T>::chunk(fa[i], fb[j], acc[i][j]);
      stamp(3);
      if (tap + 1 < 9) {
        stage_w((tap + 1) & 1);  // its last readers finished before the previous barrier
        if (tap + 2 < 9) prefetch_w(tap + 2, c0);
      }
      stamp(4);
      wg_barrier();  // next W tile visible; everyone done with this one (and, after tap 8, with the patch)
      stamp(5);
      // (Tried in round 4 and removed: issuing the next tap's operand ds_reads right behind this tap's MFMAs, into the same
      // registers, with a third W buffer -- MFMAs issue in order at the pipe's pace, so the reads start when the last MFMA
      // has issued, and the in-place overwrite of MFMA source registers stalls: 313 vs 244 us at C = 256.  A second
      // register set costs 32 VGPRs = one of the two waves per SIMD.  The loop is co-bound by LDS traffic: per tap and
      // workgroup 32 KB of fragment reads (128 cycles) + 8 KB of W tile writes (~104) against 256 cycles of MFMA.)
    }
  }

  // ---- epilogue (same scheme as the pointwise GEMM): one pass per 32-row MFMA block index, so the
  // fp32 staging tile holds only WM*32 rows and the kernel fits 4 workgroups per CU
  constexpr int VR = BN / VEC, RPP = NT / VR, SROWS = WM * 32;
  static_assert(NT % VR == 0 && VR <= 64, "epilogue mapping");
  const int cv = tid % VR, r0 = tid / VR;
  float bias[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    bias[e] = a.bias ? a.bias[n0 + cv * VEC + e] : 0.f;
    s1[e] = 0.f;
    s2[e] = 0.f;
  }
  T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * Ho * Wo * a.Cout;
#pragma unroll
  for (int pi = 0; pi < MI; ++pi) {
    if (pi) wg_barrier();
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + mfma_row(r, lane);
        const int col = (wn * NI + j) * 32 + (lane & 31);
        sC[row * CP + col] = acc[pi][j][r];
      }
    wg_barrier();
    for (int srow = r0; srow < SROWS; srow += RPP) {
      const int row = ((srow >> 5) * MI + pi) * 32 + (srow & 31);  // pixel index inside the BM tile
      float v[VEC];
      const float* pc = sC + srow * CP + cv * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = pc[e] + bias[e];
      int py, px;
      tile_px(row, py, px);
      const int oy = oy0 + py, ox = ox0 + px;
      if (RAGGED && (oy >= Ho || ox >= Wo)) continue;
      vec_t ov = f32_to_vec<T>(v);
      st_vec_pol<T>(outp + ((size_t)oy * Wo + ox) * a.Cout + n0 + cv * VEC, ov, a.nt != 0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float q = (float)ov[e];
        s1[e] += q;
        s2[e] += q * q;
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int o = VR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    float* red = sC + SROWS * CP;
    constexpr int NW = NT / 64;
    if (lane < VR) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[(wave * 2 + 0) * BN + cv * VEC + e] = s1[e];
        red[(wave * 2 + 1) * BN + cv * VEC + e] = s2[e];
      }
    }
    wg_barrier();
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, c = i % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * BN + c];
      a.stats[((size_t)(b * tiles + tile) * 2 + which) * a.Cout + n0 + c] = t;
    }
  }
  stamp(6);
  if constexpr (STAMP) {
    if (a.stamps && lane == 0) {
#pragma unroll
      for (int i = 0; i < kConvStamps; ++i) a.stamps[((size_t)blockIdx.x * (NT / 64) + wave) * kConvStamps + i] = tk[i];
    }
  }
}

static int g_conv_stamp = 0;
static unsigned long long* g_conv_stamps = nullptr;
static size_t g_conv_stamp_n = 0;
void conv3x3_stamp(int v) { g_conv_stamp = v; }
hipError_t conv3x3_stamp_fetch(double* out8) {  // mean cycles per wave of the last stamped launch (kConvStamps slots), then the wave count
  if (!g_conv_stamps || !g_conv_stamp_n) return hipErrorInvalidValue;
  std::vector<unsigned long long> h(g_conv_stamp_n);
  if (hipError_t e = hipMemcpy(h.data(), g_conv_stamps, h.size() * 8, hipMemcpyDeviceToHost); e != hipSuccess) return e;
  for (int i = 0; i < kConvStamps; ++i) out8[i] = 0.0;
  for (size_t i = 0; i < h.size(); ++i) out8[i % kConvStamps] += (double)h[i];
  for (int i = 0; i < kConvStamps; ++i) out8[i] /= (double)(h.size() / kConvStamps);
  out8[kConvStamps] = (double)(h.size() / kConvStamps);
  return hipSuccess;
}

static int conv_tw(int Wo) { return (Wo % 16 == 0) ? 16 : 8; }
int conv3x3_ntiles(int Ho, int Wo) { return ((Ho + 7) / 8) * ((Wo + conv_tw(Wo) - 1) / conv_tw(Wo)); }

template <typename T, int MODE, int TW, int BN, int WM, int WN, bool RAGGED = false>
static hipError_t launch_conv_cfg(const Conv3Args& a, hipStream_t s) {
  constexpr int NT = WM * WN * 64, BM = 8 * TW, PITCH = TilePitch<T>::value;
  constexpr int PH = MODE == 0 ? 17 : 10, PW = MODE == 0 ? 2 * TW + 1 : TW + 2;
  constexpr size_t tiles = (size_t)(PH * PW + 2 * BN) * PITCH * sizeof(T);
  constexpr size_t ctile = (size_t)(WM * 32) * (BN + 4) * 4 + (size_t)(NT / 64) * 2 * BN * 4;
  constexpr size_t lds = tiles > ctile ? tiles : ctile;
  static std::atomic<uint64_t> attr_done{0};
  if (lds > 48 * 1024) {
    if (hipError_t e = ensure_max_lds(reinterpret_cast<const void*>(&conv3x3_kernel<T, MODE, TW, BN, WM, WN, RAGGED>), (int)lds, attr_done);
        e != hipSuccess)
      return e;
  }
  const int Ho = MODE == 0 ? a.Hi / 2 : (MODE == 1 ? a.Hi * 2 : a.Hi), Wo = MODE == 0 ? a.Wi / 2 : (MODE == 1 ? a.Wi * 2 : a.Wi);
  const unsigned grid = (unsigned)(a.B * ((Ho + 7) / 8) * ((Wo + TW - 1) / TW) * (a.Cout / BN));
  static const std::string name = std::string("conv3x3_kernel<") + TypeName<T>::value + ", " + std::to_string(MODE) + ", " +
                                  std::to_string(TW) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " +
                                  std::to_string(WN) + ">";
  note_kernel(name.c_str());
  if constexpr (MODE == 1 && TW == 16 && !RAGGED && std::is_same<T, half_t>::value) {
    if (g_conv_stamp) {  // diagnostic build with in-kernel cycle stamps
      const size_t n = (size_t)grid * (NT / 64) * kConvStamps;
      if (n > g_conv_stamp_n || !g_conv_stamps) {
        if (g_conv_stamps) (void)hipFree(g_conv_stamps);
        if (hipError_t e = hipMalloc(reinterpret_cast<void**>(&g_conv_stamps), n * 8); e != hipSuccess) return e;
      }
      g_conv_stamp_n = n;
      Conv3Args b = a;
      b.stamps = g_conv_stamps;
      if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, MODE, TW, BN, WM, WN, RAGGED, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
      hipLaunchKernelGGL((conv3x3_kernel<T, MODE, TW, BN, WM, WN, RAGGED, true>), dim3(grid), dim3(NT), lds, s, b);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((conv3x3_kernel<T, MODE, TW, BN, WM, WN, RAGGED>), dim3(grid), dim3(NT), lds, s, a);
  return hipGetLastError();
}

template <typename T, int MODE>
static hipError_t launch_conv_t(const Conv3Args& a, hipStream_t s) {
  const int Ho = MODE == 0 ? a.Hi / 2 : (MODE == 1 ? a.Hi * 2 : a.Hi), Wo = MODE == 0 ? a.Wi / 2 : (MODE == 1 ? a.Wi * 2 : a.Wi);
  if (Ho < 1 || Wo < 1 || a.Cin % 32 || a.Cout % 32 || (MODE == 0 && (a.Hi % 2 || a.Wi % 2))) return hipErrorInvalidValue;
  const int BN = (a.Cout % 128 == 0) ? 128 : ((a.Cout % 64 == 0) ? 64 : 32);
  if (Ho % 8 || Wo % 8) {  // partly empty edge tiles: forward convs onto maps that are not a multiple of 8 (image sizes % 64 != 0)
    if constexpr (MODE == 0 || MODE == 1) {
      if (BN == 128) return launch_conv_cfg<T, MODE, 8, 128, 2, 2, true>(a, s);
      if (BN == 64) return launch_conv_cfg<T, MODE, 8, 64, 2, 2, true>(a, s);
      return launch_conv_cfg<T, MODE, 8, 32, 2, 1, true>(a, s);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (conv_tw(Wo) == 16) {
    if (BN == 128) return launch_conv_cfg<T, MODE, 16, 128, 2, 2>(a, s);
    if (BN == 64) return launch_conv_cfg<T, MODE, 16, 64, 2, 2>(a, s);
    return launch_conv_cfg<T, MODE, 16, 32, 4, 1>(a, s);
  }
  if (BN == 128) return launch_conv_cfg<T, MODE, 8, 128, 2, 2>(a, s);
  if (BN == 64) return launch_conv_cfg<T, MODE, 8, 64, 2, 2>(a, s);
  return launch_conv_cfg<T, MODE, 8, 32, 2, 1>(a, s);
}

hipError_t launch_conv3x3(int dtype, const Conv3Args& a, hipStream_t s) {
  if (a.mode < 0 || a.mode > 2) return hipErrorInvalidValue;
  switch (dtype) {
    case 0: return a.mode == 0 ? launch_conv_t<float, 0>(a, s) : (a.mode == 1 ? launch_conv_t<float, 1>(a, s) : launch_conv_t<float, 2>(a, s));
    case 1: return a.mode == 0 ? launch_conv_t<half_t, 0>(a, s) : (a.mode == 1 ? launch_conv_t<half_t, 1>(a, s) : launch_conv_t<half_t, 2>(a, s));
    case 2: return a.mode == 0 ? launch_conv_t<bf16_t, 0>(a, s) : (a.mode == 1 ? launch_conv_t<bf16_t, 1>(a, s) : launch_conv_t<bf16_t, 2>(a, s));
  }
  return hipErrorInvalidValue;
}

}  // namespace llie
