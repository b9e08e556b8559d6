// Depthwise 3x3 convolution, NHWC, gfx950.
//
// Replaces, for one InvertedResidualBlock (efficient_unet.py:212-223):
//   norm2 affine + FiLM (pre-folded into as/ab by gn_finalize) -> ReLU6 -> depthwise 3x3 (pad 1)
//   -> the read pass of SE's AdaptiveAvgPool2d (:97) as per-tile partial sums.
// HBM-bound: reads the 4x-expanded hidden tensor once, writes it once.  A workgroup owns an 8 x TX
// pixel tile x (8 lanes x 16 B) channels; the (8+2) x (TX+2) halo tile is activated ONCE while being
// staged into LDS (zero padding is applied after the activation, like the reference's conv padding),
// then every thread walks down its column with three rolling output-row accumulators, so each staged
// vector is read from LDS three times (once per horizontal tap) instead of nine.
#include "common.h"
#include "kernels.h"

namespace llie {

constexpr int kDwTY = 8;

template <typename T, int TX>
__global__ void __launch_bounds__(8 * TX) dwconv3x3_kernel(const DwArgs a) {
  constexpr int NT = 8 * TX;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int CC = 8 * VEC;  // channels per workgroup
  constexpr int PW = TX + 2, PH = kDwTY + 2;
  typedef typename Elem<T>::vec_t vec_t;
  __shared__ vec_t tile[PH * PW * 8];
  __shared__ float red[(NT / 64) * CC];

  const int tid = threadIdx.x, cl = tid & 7, xl = tid >> 3;
  const int tiles_x = a.W / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TX, y0 = ty * kDwTY;
  const int c0 = blockIdx.y * CC + cl * VEC;
  const int b = blockIdx.z;
  const T* in = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * a.C;
  T* out = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.C;

  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    sc[e] = a.as[(size_t)b * a.C + c0 + e];
    sh[e] = a.ab[(size_t)b * a.C + c0 + e];
  }
  // ---- stage the activated halo tile (thread's channel lane is loop invariant: NT % 8 == 0)
  for (int i = tid; i < PH * PW * 8; i += NT) {
    const int pix = i >> 3;
    const int py = pix / PW, px = pix % PW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    vec_t v;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      float f[VEC];
      ld_f32<T>(in + ((size_t)gy * a.W + gx) * a.C + c0, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = relu6f(f[e] * sc[e] + sh[e]);
      v = f32_to_vec<T>(f);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
    }
    tile[i] = v;
  }
  float w[9][VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) w[t][e] = a.w[(size_t)t * a.C + c0 + e];
  __syncthreads();

  float acc[3][VEC], psum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    acc[0][e] = acc[1][e] = acc[2][e] = 0.f;
    psum[e] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < PH; ++r) {
    float f[3][VEC];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) vec_to_f32<T>(tile[(r * PW + xl + kx) * 8 + cl], f[kx]);
    // input row r feeds output rows o = r - ky (ky = 0..2); accumulator slot = o mod 3
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int o = r - ky;
      if (o < 0 || o >= kDwTY) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[o % 3][e] += w[ky * 3 + kx][e] * f[kx][e];
    }
    const int done = r - 2;  // output row completed by this input row
    if (done >= 0) {
      vec_t ov = f32_to_vec<T>(acc[done % 3]);
      st_vec<T>(out + ((size_t)(y0 + done) * a.W + x0 + xl) * a.C + c0, ov);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        psum[e] += (float)ov[e];
        acc[done % 3][e] = 0.f;
      }
    }
  }
  // ---- SE pool partial: sum over the tile's pixels per channel (fixed order)
  if (a.pool) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < VEC; ++e) psum[e] += __shfl_xor(psum[e], o, 64);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[wave * CC + lane * VEC + e] = psum[e];
    }
    __syncthreads();
    if (tid < CC) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < NT / 64; ++wv) t += red[wv * CC + tid];
      const int ntiles = tiles_x * (a.H / kDwTY);
      a.pool[((size_t)b * ntiles + blockIdx.x) * a.C + blockIdx.y * CC + tid] = t;
    }
  }
}

static int dw_tx(int W) { return (W % 32 == 0) ? 32 : ((W % 16 == 0) ? 16 : 8); }
int dwconv_ntiles(int H, int W) { return (H / kDwTY) * (W / dw_tx(W)); }

template <typename T>
static hipError_t launch_dw_t(const DwArgs& a, hipStream_t s) {
  constexpr int CC = 8 * Elem<T>::VEC;
  if (a.C % CC || a.H % kDwTY || a.W % 8) return hipErrorInvalidValue;
  const int tx = dw_tx(a.W);
  dim3 grid((a.W / tx) * (a.H / kDwTY), a.C / CC, a.B);
  if (tx == 32) hipLaunchKernelGGL((dwconv3x3_kernel<T, 32>), grid, dim3(256), 0, s, a);
  else if (tx == 16) hipLaunchKernelGGL((dwconv3x3_kernel<T, 16>), grid, dim3(128), 0, s, a);
  else hipLaunchKernelGGL((dwconv3x3_kernel<T, 8>), grid, dim3(64), 0, s, a);
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
