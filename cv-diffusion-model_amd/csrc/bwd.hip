// Backward kernels of the training step: everything except the MFMA weight-gradient GEMM
// (wgrad.hip) and the input-gradient GEMMs / convolutions, which reuse the forward kernels with
// transposed weights.  Activation gradients are NHWC T, reductions are tile partials combined in a
// fixed order (bitwise reproducible, no float atomics).
#include <string>

#include "common.h"

namespace llie {

// =============================================================================================
// (1) dz = g * act'(x*as + ab); slab[b][tile][0][c] = sum dz, [1] = sum dz*x over tiles of 64 rows.
// Block = 64 rows x 64 channels.
template <typename T>
__global__ void __launch_bounds__(256) bwd_mask_reduce_kernel(const BwdMaskArgs a) {
  constexpr int VEC = Elem<T>::VEC, CB = 64, VPR = CB / VEC, RL = 256 / VPR;
  __shared__ float red[RL][2][CB];
  const int tid = threadIdx.x, cv = tid % VPR, rl = tid / VPR;
  const int c = blockIdx.y * CB + cv * VEC;
  const size_t m0 = (size_t)blockIdx.x * 64;
  const int ntiles = a.P / 64;
  const int b = (int)(blockIdx.x / (unsigned)ntiles), tile = (int)(blockIdx.x - (unsigned)b * ntiles);
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.f;
  if (c < a.C) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      sc[e] = a.as ? a.as[(size_t)b * a.C + c + e] : 1.f;
      sh[e] = a.ab ? a.ab[(size_t)b * a.C + c + e] : 0.f;
    }
    const T* xp = nullptr;
    int xc = 0, xo = 0;
    if (a.x0) {
      if (c < a.c0) { xp = reinterpret_cast<const T*>(a.x0); xc = a.c0; xo = c; }
      else { xp = reinterpret_cast<const T*>(a.x1); xc = a.c1; xo = c - a.c0; }
    }
    const T* gp = reinterpret_cast<const T*>(a.g);
    T* dzp = reinterpret_cast<T*>(a.dz);
    for (int r = rl; r < 64; r += RL) {
      const size_t m = m0 + r;
      float g[VEC], x[VEC], dz[VEC];
      ld_f32<T>(gp + m * a.C + c, g);
      if (xp) ld_f32<T>(xp + m * xc + xo, x);
      else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float z = x[e] * sc[e] + sh[e];
        float d = g[e];
        if (a.act == ACT_RELU6) d = (z > 0.f && z < 6.f) ? d : 0.f;
        else if (a.act == ACT_SILU) {
          const float sg = sigmoidf(z);
          d *= sg * (1.f + z * (1.f - sg));
        }
        dz[e] = dzp ? round_to<T>(d) : d;
        s1[e] += dz[e];
        s2[e] += dz[e] * x[e];
      }
      if (dzp) st_f32<T>(dzp + m * a.C + c, dz);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[rl][0][cv * VEC + e] = s1[e];
    red[rl][1][cv * VEC + e] = s2[e];
  }
  wg_barrier();
  if (tid < 2 * CB) {
    const int j = tid / CB, cc = tid % CB;
    if (blockIdx.y * CB + cc < a.C) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RL; ++q) t += red[q][j][cc];
      a.slab[(((size_t)b * ntiles + tile) * 2 + j) * a.C + blockIdx.y * CB + cc] = t;
    }
  }
}
hipError_t launch_bwd_mask_reduce(int dtype, const BwdMaskArgs& a, hipStream_t s) {
  if (a.P % 64 || a.M % a.P || a.C % 32 || (a.x0 && a.c0 + a.c1 != a.C)) return hipErrorInvalidValue;
  dim3 grid(a.M / 64, (a.C + 63) / 64);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(bwd_mask_reduce_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(bwd_mask_reduce_kernel<half_t>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(bwd_mask_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// out[b][j][c] = sum_t slab[b][t][j][c]; block = 64 channels of one (image, j): 16 tile groups x 16 float4 lanes
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* slab, float* out, int ntiles, int nj, int nj_out, int C) {
  __shared__ float part[16][64];
  const int tid = threadIdx.x, c4 = tid & 15, tg = tid >> 4;
  const int c = blockIdx.x * 64 + c4 * 4, b = blockIdx.y, j = blockIdx.z;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* p = slab + ((size_t)b * ntiles * nj + j) * C + c;
    for (int t = tg; t < ntiles; t += 16) s += *reinterpret_cast<const f32x4*>(p + (size_t)t * nj * C);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) part[tg][c4 * 4 + e] = s[e];
  wg_barrier();
  if (tid < 64 && blockIdx.x * 64 + tid < C) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][tid];
    out[((size_t)b * nj_out + j) * C + blockIdx.x * 64 + tid] = t;
  }
}
hipError_t launch_slab_reduce(const float* slab, float* out, int B, int ntiles, int nj, int nj_out, int C, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((C + 63) / 64, B, nj_out), dim3(256), 0, s, slab, out, ntiles, nj, nj_out, C);
  return hipGetLastError();
}
__global__ void batch_sum_kernel(const float* in, float* out, int B, int64_t stride, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += in[(size_t)b * stride + c];
  out[c] = s;
}
hipError_t launch_batch_sum(const float* in, float* out, int B, int64_t stride, int C, hipStream_t s) {
  hipLaunchKernelGGL(batch_sum_kernel, dim3((C + 127) / 128), dim3(128), 0, s, in, out, B, stride, C);
  return hipGetLastError();
}

// =============================================================================================
// (2) GroupNorm backward coefficients.  With xhat = (x-mean)*rstd, G = gamma*(1+s), z = xhat*G + Bc:
//   dG = sum dz*xhat = rstd*(S2 - mean*S1),  dBc = S1,
//   c1 = mean_group(G*dz) = sum_c G*S1/(cg*P),  c2 = mean_group(G*dz*xhat) = sum_c G*dG/(cg*P),
//   dx = rstd*(G*dz - c1 - xhat*c2) = dz*(rstd*G) + x*(-rstd^2*c2) + (-rstd*c1 + mean*rstd^2*c2).
// With a.slab set (round 4) the block first sums the producer's tile partials [B][ntiles][2][C] of its own cg channels -- what
// slab_reduce_kernel did in a launch of its own, 47 times per training step on the critical chain -- in a fixed order: thread ->
// (tile lane, channel), tile lanes combined in index order.
constexpr int kGnCoefMaxCg = 64;
__global__ void __launch_bounds__(256) gn_bwd_coef_kernel(const GnBwdArgs a) {
  __shared__ double part[2][4];
  __shared__ float ssum[2][256];
  __shared__ float sS[2][kGnCoefMaxCg];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = a.C / a.groups, c_lo = g * cg;
  const float mean = a.mean[(size_t)b * a.groups + g], rstd = a.rstd[(size_t)b * a.groups + g];
  const float* S1 = a.S + (size_t)b * 2 * a.C;
  const float* S2 = S1 + a.C;
  if (a.slab) {  // launcher: cg <= kGnCoefMaxCg
    const int tln = 256 / cg, tl = tid / cg, ci = tid - tl * cg;  // tile lanes; threads beyond tln * cg idle
    float p1 = 0.f, p2 = 0.f;
    if (tl < tln) {
      const float* ps = a.slab + (size_t)b * a.ntiles * 2 * a.C + c_lo + ci;
      for (int t = tl; t < a.ntiles; t += tln) {
        p1 += ps[(size_t)(2 * t) * a.C];
        p2 += ps[(size_t)(2 * t + 1) * a.C];
      }
    }
    ssum[0][tid] = p1;
    ssum[1][tid] = p2;
    wg_barrier();
    if (tid < cg) {
      float t1 = 0.f, t2 = 0.f;
      for (int l = 0; l < tln; ++l) {
        t1 += ssum[0][l * cg + tid];
        t2 += ssum[1][l * cg + tid];
      }
      sS[0][tid] = t1;
      sS[1][tid] = t2;
    }
    wg_barrier();
  }
  // S1 / S2 of channel c: from LDS (group-local index) when the partials were summed here, else from the [B][2][C] table
  const bool local = a.slab != nullptr;
  auto s1_of = [&](int c) { return local ? sS[0][c - c_lo] : S1[c]; };
  auto s2_of = [&](int c) { return local ? sS[1][c - c_lo] : S2[c]; };
  const float* film = a.film ? a.film + (size_t)b * a.film_stride : nullptr;
  double t1 = 0.0, t2 = 0.0;
  for (int i = tid; i < cg; i += 256) {
    const int c = c_lo + i;
    const float G = a.gamma[c] * (film ? 1.f + film[c] : 1.f);
    const float dG = rstd * (s2_of(c) - mean * s1_of(c));
    t1 += (double)G * (double)s1_of(c);
    t2 += (double)G * (double)dG;
  }
  t1 = wave_sum(t1);
  t2 = wave_sum(t2);
  if (lane == 0) { part[0][wave] = t1; part[1][wave] = t2; }
  wg_barrier();
  const double n = (double)cg * (double)a.P;
  const float c1 = (float)(((part[0][0] + part[0][1]) + (part[0][2] + part[0][3])) / n);
  const float c2 = (float)(((part[1][0] + part[1][1]) + (part[1][2] + part[1][3])) / n);
  for (int i = tid; i < cg; i += 256) {
    const int c = c_lo + i;
    const size_t o = (size_t)b * a.C + c;
    const float G = a.gamma[c] * (film ? 1.f + film[c] : 1.f);
    a.A[o] = rstd * G;
    a.Bq[o] = -rstd * rstd * c2;
    a.Cq[o] = -rstd * c1 + mean * rstd * rstd * c2;
    a.dG[o] = rstd * (s2_of(c) - mean * s1_of(c));
    a.dBc[o] = s1_of(c);
  }
}
hipError_t launch_gn_bwd_coef(const GnBwdArgs& a, hipStream_t s) {
  if (a.C % a.groups) return hipErrorInvalidValue;
  if (a.slab ? (a.ntiles <= 0 || a.C / a.groups > kGnCoefMaxCg) : !a.S) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(a.groups, a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ void gn_param_grad_kernel(const GnParamGradArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  float dg = 0.f, db = 0.f;
  const float gamma = a.gamma[c], beta = a.beta[c];
  for (int b = 0; b < a.B; ++b) {
    const float dG = a.dG[(size_t)b * a.C + c], dB = a.dBc[(size_t)b * a.C + c];
    float one_s = 1.f;
    if (a.film) one_s += a.film[(size_t)b * a.film_stride + c];
    dg += dG * one_s;
    db += dB * one_s;
    if (a.dfilm) {
      float* row = a.dfilm + (size_t)b * a.dfilm_stride;
      row[c] = dG * gamma + dB * beta;
      row[a.C + c] = dB;
    }
  }
  a.dgamma[c] = dg;
  a.dbeta[c] = db;
}
hipError_t launch_gn_param_grad(const GnParamGradArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((a.C + 127) / 128), dim3(128), 0, s, a);
  return hipGetLastError();
}

// (3) dx = dz*A + x*Bq + Cq (+ add0) (+ add1), per 16-byte channel vector
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const GnApplyArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  const int C = a.c0 + a.c1;
  const unsigned vpr = C / VEC;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;  // 32-bit index math (the launcher checks the range)
  if (idx >= (unsigned)a.M * vpr) return;
  const unsigned mi = idx / vpr;
  const size_t m = mi;
  const int c = (int)(idx - mi * vpr) * VEC;
  const int b = (int)(mi / (unsigned)a.P);
  const bool second = c >= a.c0;
  const int xc = second ? a.c1 : a.c0, xo = second ? c - a.c0 : c;
  const T* xp = reinterpret_cast<const T*>(second ? a.x1 : a.x0);
  float dz[VEC], x[VEC], r[VEC];
  ld_f32<T>(reinterpret_cast<const T*>(a.dz) + m * C + c, dz);
  ld_f32<T>(xp + m * xc + xo, x);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const size_t o = (size_t)b * C + c + e;
    r[e] = dz[e] * a.A[o] + x[e] * a.Bq[o] + a.Cq[o];
  }
  if (a.add0) {
    float t[VEC];
    ld_f32<T>(reinterpret_cast<const T*>(a.add0) + m * C + c, t);
#pragma unroll
    for (int e = 0; e < VEC; ++e) r[e] += t[e];
  }
  const T* add1 = reinterpret_cast<const T*>(second ? a.add1_1 : a.add1_0);
  if (add1) {
    float t[VEC];
    ld_f32<T>(add1 + m * xc + xo, t);
#pragma unroll
    for (int e = 0; e < VEC; ++e) r[e] += t[e];
  }
  st_f32<T>(reinterpret_cast<T*>(second ? a.dx1 : a.dx0) + m * xc + xo, r);
}
hipError_t launch_gn_bwd_apply(int dtype, const GnApplyArgs& a, hipStream_t s) {
  const int C = a.c0 + a.c1;
  if (C % 32 || a.c0 % 32 || a.M % a.P) return hipErrorInvalidValue;
  const size_t vecs = (size_t)a.M * C / (dtype == 0 ? 4 : 8);
  if (vecs >= (1ull << 31)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((vecs + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(gn_bwd_apply_kernel<half_t>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T>
__global__ void __launch_bounds__(256) add_into_kernel(T* dst, const T* src, int64_t nvec) {
  constexpr int VEC = Elem<T>::VEC;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  float a[VEC], b[VEC];
  ld_f32<T>(dst + i * VEC, a);
  ld_f32<T>(src + i * VEC, b);
#pragma unroll
  for (int e = 0; e < VEC; ++e) a[e] += b[e];
  st_f32<T>(dst + i * VEC, a);
}
hipError_t launch_add_into(int dtype, void* dst, const void* src, int64_t n, hipStream_t s) {
  const int vec = dtype == 0 ? 4 : 8;
  if (n % vec) return hipErrorInvalidValue;
  const int64_t nv = n / vec;
  dim3 grid((unsigned)((nv + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(add_into_kernel<float>, grid, dim3(256), 0, s, (float*)dst, (const float*)src, nv); break;
    case 1: hipLaunchKernelGGL(add_into_kernel<half_t>, grid, dim3(256), 0, s, (half_t*)dst, (const half_t*)src, nv); break;
    case 2: hipLaunchKernelGGL(add_into_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)dst, (const bf16_t*)src, nv); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_fill_zero(void* dst, int64_t bytes, hipStream_t s) { return hipMemsetAsync(dst, 0, (size_t)bytes, s); }

// First stage of a two-stage combine of `nparts` partial rows of `nk` floats: block (x, g) adds rows g, g + G, ...
// (in that order) into row g, of which it is the only reader; the caller then combines the first G rows.
constexpr int kPartGroups = 16;
__global__ void __launch_bounds__(256) partial_groups_kernel(float* partial, int64_t nk, int nparts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= nk || g >= nparts) return;
  float s = 0.f;
  for (int sp = g; sp < nparts; sp += kPartGroups) s += partial[(size_t)sp * nk + i];
  partial[(size_t)g * nk + i] = s;
}

// =============================================================================================
// (5) depthwise weight gradient: dw[ky][kx][c] = sum_{b,y,x} dh2[y][x] * a2[y+ky-1][x+kx-1], dh2 = g*gs + gb,
// a2 = relu6(h*as + ab).  Row-streaming like the forward depthwise kernel: a strip of TX pixels x TY rows x CC
// channels per workgroup, activated rows of a2 pass through a 2-deep LDS ring (3 reads per row), each thread keeps
// the dh2 values of its pixel for the rows R-1, R, R+1 in registers and 9 x VEC running sums, which are combined
// over the strip's pixels through LDS at the end (fixed order).
constexpr int kDwgTY = 32;
static int dwg_tx(int W) { return (W % 32 == 0) ? 32 : ((W % 16 == 0) ? 16 : 8); }
int dw_wgrad_strips(int H, int W) { return (W / dwg_tx(W)) * ((H + kDwgTY - 1) / kDwgTY); }

template <typename T, int TX>
__global__ void __launch_bounds__(8 * TX, 3) dw_wgrad_kernel(const DwWgradArgs a) {  // 3 waves per SIMD: 170 VGPRs (the unconstrained allocation lands 4 above)
  constexpr int VEC = Elem<T>::VEC, CC = 8 * VEC, PW = TX + 2;
  typedef typename Elem<T>::vec_t vec_t;
  __shared__ vec_t ring[2][PW * 8];
  __shared__ float red[TX][CC];
  const int tid = threadIdx.x, cl = tid & 7, xl = tid >> 3;
  const int tiles_x = a.W / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TX, y0 = ty * kDwgTY, TY = min(kDwgTY, a.H - y0);
  const int c0 = blockIdx.y * CC + cl * VEC, b = blockIdx.z;
  const bool is_halo = tid < 16;
  const int hx = tid < 8 ? x0 - 1 : x0 + TX;
  const bool hx_ok = is_halo && hx >= 0 && hx < a.W;
  const int hslot = tid < 8 ? 0 : TX + 1;
  float gs[VEC], gb[VEC], hs[VEC], hb[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const size_t o = (size_t)b * a.C + c0 + e;
    gs[e] = a.gs ? a.gs[o] : 1.f;
    gb[e] = a.gb ? a.gb[o] : 0.f;
    hs[e] = a.as[o];
    hb[e] = a.ab[o];
  }
  const T* gp = reinterpret_cast<const T*>(a.g) + (size_t)b * a.H * a.W * a.C + c0;
  const T* hp = reinterpret_cast<const T*>(a.h) + (size_t)b * a.H * a.W * a.C + c0;
  auto activate = [&](vec_t v) {
    float f[VEC];
    vec_to_f32<T>(v, f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = relu6f(f[e] * hs[e] + hb[e]);
    return f32_to_vec<T>(f);
  };
  vec_t zero;
#pragma unroll
  for (int e = 0; e < VEC; ++e) zero[e] = (T)0.f;
  float acc[9][VEC], gprev[VEC], gcur[VEC], gnext[VEC];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
  auto load_g = [&](int y, float* out) {  // dh2 of this thread's pixel in row y (zero outside the strip)
    if (y >= y0 && y < y0 + TY) {
      float g[VEC];
      ld_f32<T>(gp + ((size_t)y * a.W + x0 + xl) * a.C, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) out[e] = g[e] * gs[e] + gb[e];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) out[e] = 0.f;
    }
  };
#pragma unroll
  for (int e = 0; e < VEC; ++e) gprev[e] = gcur[e] = 0.f;
  load_g(y0, gnext);
  const int nrows = TY + 2;  // a2 rows y0-1 .. y0+TY
  vec_t hpre = zero, hpre_h = zero;
  auto issue = [&](int r) {
    const int gy = y0 - 1 + r;
    if (r < nrows && gy >= 0 && gy < a.H) {
      const T* row = hp + (size_t)gy * a.W * a.C;
      hpre = ld_vec<T>(row + (size_t)(x0 + xl) * a.C);
      if (hx_ok) hpre_h = ld_vec<T>(row + (size_t)hx * a.C);
    }
  };
  issue(0);
  for (int r = 0; r < nrows; ++r) {
    const int gy = y0 - 1 + r;
    const bool row_ok = gy >= 0 && gy < a.H;
    vec_t* buf = ring[r & 1];
    buf[(xl + 1) * 8 + cl] = row_ok ? activate(hpre) : zero;
    if (is_halo) buf[hslot * 8 + cl] = (row_ok && hx_ok) ? activate(hpre_h) : zero;
    issue(r + 1);
    float gload[VEC];
    load_g(gy + 2, gload);
    wg_barrier();
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float f[VEC];
      vec_to_f32<T>(buf[(xl + kx) * 8 + cl], f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc[0 + kx][e] += gnext[e] * f[e];  // ky = 0: output row gy + 1
        acc[3 + kx][e] += gcur[e] * f[e];   // ky = 1: output row gy
        acc[6 + kx][e] += gprev[e] * f[e];  // ky = 2: output row gy - 1
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gprev[e] = gcur[e];
      gcur[e] = gnext[e];
      gnext[e] = gload[e];
    }
  }
  const size_t pbase = (((size_t)b * gridDim.x + blockIdx.x) * 9) * a.C + blockIdx.y * CC;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wg_barrier();
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[xl][cl * VEC + e] = acc[t][e];
    wg_barrier();
    if (tid < CC) {
      float v = 0.f;
      for (int q = 0; q < TX; ++q) v += red[q][tid];
      a.partial[pbase + (size_t)t * a.C + tid] = v;
    }
  }
}
// out[c][tap] = sum over (b, strip) partials, sequentially
__global__ void dw_wgrad_reduce_kernel(const float* partial, float* out, int nparts, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C) return;
  const int t = i / C, c = i % C;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[((size_t)p * 9 + t) * C + c];
  out[(size_t)c * 9 + t] = s;
}
template <typename T>
static void launch_dw_wgrad_t(const DwWgradArgs& a, dim3 grid, int TX, hipStream_t s) {
  if (TX == 32) hipLaunchKernelGGL((dw_wgrad_kernel<T, 32>), grid, dim3(256), 0, s, a);
  else if (TX == 16) hipLaunchKernelGGL((dw_wgrad_kernel<T, 16>), grid, dim3(128), 0, s, a);
  else hipLaunchKernelGGL((dw_wgrad_kernel<T, 8>), grid, dim3(64), 0, s, a);
}
hipError_t launch_dw_wgrad(int dtype, const DwWgradArgs& a, hipStream_t s) {
  const int CC = dtype == 0 ? 32 : 64;
  if (a.C % CC || a.W % 8) return hipErrorInvalidValue;
  const int TX = dwg_tx(a.W);
  dim3 grid(dw_wgrad_strips(a.H, a.W), a.C / CC, a.B);
  switch (dtype) {
    case 0: launch_dw_wgrad_t<float>(a, grid, TX, s); break;
    case 1: launch_dw_wgrad_t<half_t>(a, grid, TX, s); break;
    case 2: launch_dw_wgrad_t<bf16_t>(a, grid, TX, s); break;
    default: return hipErrorInvalidValue;
  }
  int nparts = a.B * (int)grid.x;
  if (nparts > kPartGroups) {
    hipLaunchKernelGGL(partial_groups_kernel, dim3((9 * a.C + 255) / 256, kPartGroups), dim3(256), 0, s, a.partial, (int64_t)9 * a.C, nparts);
    nparts = kPartGroups;
  }
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((9 * a.C + 255) / 256), dim3(256), 0, s, a.partial, a.out, nparts, a.C);
  return hipGetLastError();
}

// =============================================================================================
// (6) small dense pieces over the batch
// dx[b][k] = sum_r dy[b][r] * W[r][k].  R is cut into gridDim.z chunks (long FiLM tables would otherwise run on a
// handful of CUs); each chunk writes its partial row, a second kernel adds the chunks in order.
template <typename WT>
__global__ void __launch_bounds__(256) linear_dx_kernel(const float* dy, int64_t dy_stride, const WT* W, float* dst, int R, int Kc,
                                                        int rchunk, int B) {
  __shared__ float part[4][64];
  const int tid = threadIdx.x, kl = tid & 63, rg = tid >> 6;
  const int k = blockIdx.x * 64 + kl, b = blockIdx.y;
  const int r0 = blockIdx.z * rchunk, r1 = min(r0 + rchunk, R);
  float acc = 0.f;
  if (k < Kc) {
    const float* dyr = dy + (size_t)b * dy_stride;
    for (int r = r0 + rg; r < r1; r += 4) acc += dyr[r] * (float)W[(size_t)r * Kc + k];
  }
  part[rg][kl] = acc;
  wg_barrier();
  if (rg == 0 && k < Kc)
    dst[((size_t)blockIdx.z * B + b) * Kc + k] = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
}
__global__ void linear_dx_combine_kernel(const float* part, float* dx, int64_t n, int chunks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)c * n + i];
  dx[i] = s;
}
int linear_dx_chunks(int R) { return R > 256 ? (R + 127) / 128 : 1; }
hipError_t launch_linear_dx(int wdtype, const float* dy, int64_t dy_stride, const void* W, float* dx, int B, int R, int Kc,
                            hipStream_t s, float* scratch) {
  const int chunks = scratch ? linear_dx_chunks(R) : 1;
  const int rchunk = (R + chunks - 1) / chunks;
  float* dst = chunks > 1 ? scratch : dx;
  dim3 grid((Kc + 63) / 64, B, chunks);
  switch (wdtype) {
    case 0: hipLaunchKernelGGL(linear_dx_kernel<float>, grid, dim3(256), 0, s, dy, dy_stride, (const float*)W, dst, R, Kc, rchunk, B); break;
    case 1: hipLaunchKernelGGL(linear_dx_kernel<half_t>, grid, dim3(256), 0, s, dy, dy_stride, (const half_t*)W, dst, R, Kc, rchunk, B); break;
    case 2: hipLaunchKernelGGL(linear_dx_kernel<bf16_t>, grid, dim3(256), 0, s, dy, dy_stride, (const bf16_t*)W, dst, R, Kc, rchunk, B); break;
    default: return hipErrorInvalidValue;
  }
  if (chunks > 1) {
    const int64_t n = (int64_t)B * Kc;
    hipLaunchKernelGGL(linear_dx_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scratch, dx, n, chunks);
  }
  return hipGetLastError();
}
__global__ void __launch_bounds__(256) linear_dw_kernel(const float* dy, int64_t dy_stride, const float* x, float* dW, float* db, int B, int R, int Kc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)R * Kc) return;
  const int r = (int)(i / Kc), k = (int)(i % Kc);
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float d = dy[(size_t)b * dy_stride + r];
    s += d * x[(size_t)b * Kc + k];
    sb += d;
  }
  dW[i] = s;
  if (db && k == 0) db[r] = sb;
}
hipError_t launch_linear_dw(const float* dy, int64_t dy_stride, const float* x, float* dW, float* db, int B, int R, int Kc,
                            hipStream_t s) {
  const int64_t n = (int64_t)R * Kc;
  hipLaunchKernelGGL(linear_dw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, dy_stride, x, dW, db, B, R, Kc);
  return hipGetLastError();
}
__global__ void sigmoid_bwd_kernel(const float* dg, const float* g, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dg[i] * g[i] * (1.f - g[i]);
}
__global__ void relu6_bwd_kernel(const float* dy, const float* y, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (y[i] > 0.f && y[i] < 6.f) ? dy[i] : 0.f;
}
__global__ void silu_bwd_kernel(const float* dy, const float* x, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float sg = sigmoidf(x[i]);
    out[i] = dy[i] * sg * (1.f + x[i] * (1.f - sg));
  }
}
__global__ void scale_rows_kernel(const float* x, float* out, int64_t n, float sc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] * sc;
}
#define LLIE_EW(name, kernel, ...)                                                                        \
  hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, __VA_ARGS__);             \
  return hipGetLastError();
hipError_t launch_sigmoid_bwd(const float* dg, const float* g, float* out, int64_t n, hipStream_t s) { LLIE_EW(a, sigmoid_bwd_kernel, dg, g, out, n) }
hipError_t launch_relu6_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t s) { LLIE_EW(a, relu6_bwd_kernel, dy, y, out, n) }
hipError_t launch_silu_bwd(const float* dy, const float* x, float* out, int64_t n, hipStream_t s) { LLIE_EW(a, silu_bwd_kernel, dy, x, out, n) }
hipError_t launch_scale_rows(const float* x, float* out, int64_t n, float sc, hipStream_t s) { LLIE_EW(a, scale_rows_kernel, x, out, n, sc) }
#undef LLIE_EW

__global__ void sin_embed_kernel(const int64_t* t, const float* freqs, float* emb, int dim) {
  const int r = blockIdx.x, half = dim / 2;
  const float tv = (float)t[r];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float arg = tv * freqs[i];
    emb[(size_t)r * dim + i] = cosf(arg);
    emb[(size_t)r * dim + half + i] = sinf(arg);
  }
}
hipError_t launch_sin_embed(const int64_t* t, const float* freqs, float* emb, int rows, int dim, hipStream_t s) {
  hipLaunchKernelGGL(sin_embed_kernel, dim3(rows), dim3(64), 0, s, t, freqs, emb, dim);
  return hipGetLastError();
}

// =============================================================================================
// (7) bilinear x2 (align_corners=False, F.interpolate semantics: source index clamped at 0) and its adjoint
struct Bil { int i0, i1; float l0, l1; };
__device__ __forceinline__ Bil bil_src(int u, int n_in) {
  float s = ((float)u + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  Bil r;
  r.i0 = (int)s;
  r.i1 = min(r.i0 + 1, n_in - 1);
  r.l1 = s - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_kernel(const T* in, T* out, int B, int Hi, int Wi, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int vpr = C / VEC, Ho = 2 * Hi, Wo = 2 * Wi;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * Ho * Wo * vpr) return;
  const int c = (int)(idx % vpr) * VEC;
  size_t pix = idx / vpr;
  const int ux = (int)(pix % Wo); pix /= Wo;
  const int uy = (int)(pix % Ho);
  const int b = (int)(pix / Ho);
  const Bil by = bil_src(uy, Hi), bx = bil_src(ux, Wi);
  const T* base = in + (size_t)b * Hi * Wi * C + c;
  float f00[VEC], f01[VEC], f10[VEC], f11[VEC], f[VEC];
  ld_f32<T>(base + ((size_t)by.i0 * Wi + bx.i0) * C, f00);
  ld_f32<T>(base + ((size_t)by.i0 * Wi + bx.i1) * C, f01);
  ld_f32<T>(base + ((size_t)by.i1 * Wi + bx.i0) * C, f10);
  ld_f32<T>(base + ((size_t)by.i1 * Wi + bx.i1) * C, f11);
#pragma unroll
  for (int e = 0; e < VEC; ++e) f[e] = by.l0 * (bx.l0 * f00[e] + bx.l1 * f01[e]) + by.l1 * (bx.l0 * f10[e] + bx.l1 * f11[e]);
  st_f32<T>(out + (((size_t)b * Ho + uy) * Wo + ux) * C + c, f);
}
// adjoint: din[y][x] = sum over the (at most 4x4) output pixels whose 2x2 footprint touches (y, x)
template <typename T>
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const T* dout, T* din, int B, int Hi, int Wi, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int vpr = C / VEC, Ho = 2 * Hi, Wo = 2 * Wi;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * Hi * Wi * vpr) return;
  const int c = (int)(idx % vpr) * VEC;
  size_t pix = idx / vpr;
  const int x = (int)(pix % Wi); pix /= Wi;
  const int y = (int)(pix % Hi);
  const int b = (int)(pix / Hi);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  const T* base = dout + (size_t)b * Ho * Wo * C + c;
  for (int uy = max(2 * y - 2, 0); uy <= min(2 * y + 2, Ho - 1); ++uy) {
    const Bil by = bil_src(uy, Hi);
    const float wy = (by.i0 == y ? by.l0 : 0.f) + (by.i1 == y ? by.l1 : 0.f);
    if (wy == 0.f) continue;
    for (int ux = max(2 * x - 2, 0); ux <= min(2 * x + 2, Wo - 1); ++ux) {
      const Bil bx = bil_src(ux, Wi);
      const float wx = (bx.i0 == x ? bx.l0 : 0.f) + (bx.i1 == x ? bx.l1 : 0.f);
      if (wx == 0.f) continue;
      float g[VEC];
      ld_f32<T>(base + ((size_t)uy * Wo + ux) * C, g);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += wy * wx * g[e];
    }
  }
  st_f32<T>(din + (((size_t)b * Hi + y) * Wi + x) * C + c, acc);
}
template <typename T>
__global__ void __launch_bounds__(256) dilate2x_kernel(const T* in, T* out, int B, int Hi, int Wi, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int vpr = C / VEC, Ho = 2 * Hi, Wo = 2 * Wi;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * Ho * Wo * vpr) return;
  const int c = (int)(idx % vpr) * VEC;
  size_t pix = idx / vpr;
  const int ux = (int)(pix % Wo); pix /= Wo;
  const int uy = (int)(pix % Ho);
  const int b = (int)(pix / Ho);
  typename Elem<T>::vec_t v;
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
  if (!(uy & 1) && !(ux & 1)) v = ld_vec<T>(in + (((size_t)b * Hi + uy / 2) * Wi + ux / 2) * C + c);
  st_vec<T>(out + (((size_t)b * Ho + uy) * Wo + ux) * C + c, v);
}
#define LLIE_PIX3(kernel, npix)                                                                                  \
  if (C % (dtype == 0 ? 4 : 8)) return hipErrorInvalidValue;                                                    \
  {                                                                                                             \
    const size_t n = (size_t)(npix) * (C / (dtype == 0 ? 4 : 8));                                                \
    dim3 grid((unsigned)((n + 255) / 256));                                                                     \
    switch (dtype) {                                                                                            \
      case 0: hipLaunchKernelGGL(kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, B, Hi, Wi, C); break;      \
      case 1: hipLaunchKernelGGL(kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)in, (half_t*)out, B, Hi, Wi, C); break;   \
      case 2: hipLaunchKernelGGL(kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, Hi, Wi, C); break;   \
      default: return hipErrorInvalidValue;                                                                     \
    }                                                                                                           \
  }                                                                                                             \
  return hipGetLastError();
hipError_t launch_upsample2x(int dtype, const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s) { LLIE_PIX3(upsample2x_kernel, (size_t)B * 4 * Hi * Wi) }
hipError_t launch_upsample2x_bwd(int dtype, const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s) { LLIE_PIX3(upsample2x_bwd_kernel, (size_t)B * Hi * Wi) }
hipError_t launch_dilate2x(int dtype, const void* in, void* out, int B, int Hi, int Wi, int C, hipStream_t s) { LLIE_PIX3(dilate2x_kernel, (size_t)B * 4 * Hi * Wi) }
#undef LLIE_PIX3

// OIHW fp32 -> [8 - tap][I][O] T: the weights of the input-gradient convolution (taps flipped, channels transposed)
template <typename T>
__global__ void repack_conv3x3_t_kernel(const float* src, T* dst, int O, int I) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)O * I * 9) return;
  const int tap = (int)(i % 9);
  const int ci = (int)((i / 9) % I), co = (int)(i / (9 * I));
  dst[((size_t)(8 - tap) * I + ci) * O + co] = (T)src[i];
}
hipError_t launch_repack_conv3x3_t(int dtype, const float* src, void* dst, int O, int I, hipStream_t s) {
  const int64_t n = (int64_t)O * I * 9;
  dim3 grid((unsigned)((n + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(repack_conv3x3_t_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, O, I); break;
    case 1: hipLaunchKernelGGL(repack_conv3x3_t_kernel<half_t>, grid, dim3(256), 0, s, src, (half_t*)dst, O, I); break;
    case 2: hipLaunchKernelGGL(repack_conv3x3_t_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, O, I); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
template <typename T>
__global__ void cvt_rows_t_kernel(const float* src, T* dst, int rows, int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(size_t)c * rows + r] = (T)src[i];
}
hipError_t launch_cvt_rows_t(int dtype, const float* src, void* dst, int rows, int cols, hipStream_t s) {
  const int64_t n = (int64_t)rows * cols;
  dim3 grid((unsigned)((n + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(cvt_rows_t_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, rows, cols); break;
    case 1: hipLaunchKernelGGL(cvt_rows_t_kernel<half_t>, grid, dim3(256), 0, s, src, (half_t*)dst, rows, cols); break;
    case 2: hipLaunchKernelGGL(cvt_rows_t_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, rows, cols); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- output head.  Forward: eps[o][p] = bias[o] + sum_{c,tap} a[p + tap - 1][c] * W[o][c][tap], a = silu(h*as + ab).
// data gradient: da[q][c] = sum_{o,tap} deps[o][q - (tap - 1)] * W[o][c][tap]
template <typename T>
__global__ void __launch_bounds__(256) final_bwd_data_kernel(const FinalBwdArgs a) {
  constexpr int VEC = Elem<T>::VEC;
  const int vpr = a.C / VEC;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.B * a.H * a.W * vpr) return;
  const int c = (int)(idx % vpr) * VEC;
  size_t pix = idx / vpr;
  const int x = (int)(pix % a.W); pix /= a.W;
  const int y = (int)(pix % a.H);
  const int b = (int)(pix / a.H);
  const size_t plane = (size_t)a.H * a.W;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int o = 0; o < a.Cout; ++o) {
    const float* dp = a.deps + ((size_t)b * a.Cout + o) * plane;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y - (ky - 1);
      if (yy < 0 || yy >= a.H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x - (kx - 1);
        if (xx < 0 || xx >= a.W) continue;
        const float d = dp[(size_t)yy * a.W + xx];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += d * a.w[((size_t)(ky * 3 + kx) * a.C + c + e) * 4 + o];  // engine layout [9][C][4]
      }
    }
  }
  st_f32<T>(reinterpret_cast<T*>(a.da) + (((size_t)b * a.H + y) * a.W + x) * a.C + c, acc);
}
hipError_t launch_final_bwd_data(int dtype, const FinalBwdArgs& a, hipStream_t s) {
  if (a.C % 8) return hipErrorInvalidValue;
  const size_t n = (size_t)a.B * a.H * a.W * (a.C / (dtype == 0 ? 4 : 8));
  dim3 grid((unsigned)((n + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(final_bwd_data_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(final_bwd_data_kernel<half_t>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(final_bwd_data_kernel<bf16_t>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// The weight gradients of the output head and of the input conv run on the MFMA weight-gradient GEMM (wgrad.hip, nine
// taps per launch) once their fp32 NCHW planes are packed into a 32-channel NHWC T tensor (channels beyond the real
// ones are zero): [B][c0 (+c1)][P] -> [B*P][32].
template <typename T>
__global__ void __launch_bounds__(256) pack_planes_kernel(const float* x0, const float* x1, int c0, int c1, T* out, int B, int P) {
  constexpr int VEC = Elem<T>::VEC;
  const size_t m = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= (size_t)B * P) return;
  const int b = (int)(m / P);
  const size_t p = m % P;
  float v[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) v[c] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < c0) v[c] = x0[((size_t)b * c0 + c) * P + p];
    else if (c < c0 + c1) v[c] = x1[((size_t)b * c1 + (c - c0)) * P + p];
  }
#pragma unroll
  for (int q = 0; q < 32 / VEC; ++q) st_f32<T>(out + m * 32 + q * VEC, v + q * VEC);
}
hipError_t launch_pack_planes(int dtype, const float* x0, const float* x1, int c0, int c1, void* out, int B, int P, hipStream_t s) {
  if (c0 + c1 > 8 || c0 < 1) return hipErrorInvalidValue;
  dim3 grid((unsigned)(((size_t)B * P + 255) / 256));
  switch (dtype) {
    case 0: hipLaunchKernelGGL(pack_planes_kernel<float>, grid, dim3(256), 0, s, x0, x1, c0, c1, (float*)out, B, P); break;
    case 1: hipLaunchKernelGGL(pack_planes_kernel<half_t>, grid, dim3(256), 0, s, x0, x1, c0, c1, (half_t*)out, B, P); break;
    case 2: hipLaunchKernelGGL(pack_planes_kernel<bf16_t>, grid, dim3(256), 0, s, x0, x1, c0, c1, (bf16_t*)out, B, P); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// =============================================================================================
// (8) linear attention backward (forward: efficient_unet.py:288-302).  With Q = phi(q), K = phi(k):
//   den = Q.ks + 1e-6, out = (Q kv)/den;  dnum = dout/den, dden = -sum_e dout*out/den
//   dQ = dnum kv^T + dden*ks;  dkv = Q^T dnum;  dks = Q^T dden;  dK = V dkv^T + dks;  dV = K dkv;  phi'(x) = x > 0 ? 1 : exp(x)
__device__ __forceinline__ float phi_f(float x) { return x > 0.f ? x + 1.f : __expf(x); }
__device__ __forceinline__ float dphi_f(float x) { return x > 0.f ? 1.f : __expf(x); }

// pass A: one block per (64 positions, head, image): dq and this tile's dkv / dks partial
template <typename T>
__global__ void __launch_bounds__(256) linattn_bwd_q_kernel(const AttnBwdArgs a) {
  __shared__ float skv[32 * 33];
  __shared__ float sq[64][33], sraw[64][33], sdn[64][33], sdd[64];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int n0 = blockIdx.x * 64, ntile = a.N / 64;
  const int inner = a.heads * 32, ld = 3 * inner;
  const float* kvp = a.kv + (size_t)(b * a.heads + h) * 32 * 33;
  const size_t sps = (size_t)a.B * a.heads * 32 * 33;
  for (int i = tid; i < 32 * 33; i += 256) {
    float v = kvp[i];
    for (int sp = 1; sp < a.nsplit; ++sp) v += kvp[sp * sps + i];
    skv[i] = v;
  }
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + ((size_t)b * a.N + n0) * ld + h * 32;
  const T* dobase = reinterpret_cast<const T*>(a.dout) + ((size_t)b * a.N + n0) * inner + h * 32;
  for (int i = tid; i < 64 * 32; i += 256) {
    const int n = i >> 5, c = i & 31;
    const float q = (float)qbase[(size_t)n * ld + c];
    sraw[n][c] = q;
    sq[n][c] = phi_f(q);
  }
  wg_barrier();
  const int n = tid >> 2, e0 = (tid & 3) * 8;
  {
    float num[8], den = 0.f, dsum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) num[q] = 0.f;
#pragma unroll 8
    for (int d = 0; d < 32; ++d) {
      const float qd = sq[n][d];
      den += qd * skv[d * 33 + 32];
#pragma unroll
      for (int q = 0; q < 8; ++q) num[q] += qd * skv[d * 33 + e0 + q];
    }
    const float inv = 1.f / (den + 1e-6f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float dout = (float)dobase[(size_t)n * inner + e0 + q];
      const float dn = dout * inv;
      sdn[n][e0 + q] = dn;
      dsum += dn * (num[q] * inv);  // dout*out/den
    }
    dsum += __shfl_xor(dsum, 1, 64);
    dsum += __shfl_xor(dsum, 2, 64);
    if ((tid & 3) == 0) sdd[n] = -dsum;
  }
  wg_barrier();
  {  // dq for d in [e0, e0+8)
    T* dq = reinterpret_cast<T*>(a.dqkv) + ((size_t)b * a.N + n0 + n) * ld + h * 32 + e0;
    const float dd = sdd[n];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int d = e0 + q;
      float v = dd * skv[d * 33 + 32];
#pragma unroll 8
      for (int e = 0; e < 32; ++e) v += sdn[n][e] * skv[d * 33 + e];
      dq[q] = (T)(v * dphi_f(sraw[n][d]));
    }
  }
  {  // dkv / dks partial of this tile
    const int d = tid >> 3, c0 = (tid & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, ks = 0.f;
#pragma unroll 8
    for (int m = 0; m < 64; ++m) {
      const float qd = sq[m][d];
      ks += qd * sdd[m];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += qd * sdn[m][c0 + q];
    }
    float* out = a.dkv + (((size_t)(b * a.heads + h)) * ntile + blockIdx.x) * (32 * 33);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[d * 33 + c0 + q] = acc[q];
    if ((tid & 7) == 0) out[d * 33 + 32] = ks;
  }
}
// pass B: dk, dv from the reduced dkv [B][heads][32][33] (a.dkv points at the reduced table here)
template <typename T>
__global__ void __launch_bounds__(256) linattn_bwd_kv_kernel(const AttnBwdArgs a) {
  __shared__ float sd[32 * 33];
  __shared__ float sk[64][33], sraw[64][33], sv[64][33];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int n0 = blockIdx.x * 64;
  const int inner = a.heads * 32, ld = 3 * inner;
  const float* dp = a.dkv + (size_t)(b * a.heads + h) * 32 * 33;
  for (int i = tid; i < 32 * 33; i += 256) sd[i] = dp[i];
  const T* base = reinterpret_cast<const T*>(a.qkv) + ((size_t)b * a.N + n0) * ld + h * 32;
  for (int i = tid; i < 64 * 32; i += 256) {
    const int n = i >> 5, c = i & 31;
    const float k = (float)base[(size_t)n * ld + inner + c];
    sraw[n][c] = k;
    sk[n][c] = phi_f(k);
    sv[n][c] = (float)base[(size_t)n * ld + 2 * inner + c];
  }
  wg_barrier();
  const int n = tid >> 2, e0 = (tid & 3) * 8;
  T* drow = reinterpret_cast<T*>(a.dqkv) + ((size_t)b * a.N + n0 + n) * ld + h * 32 + e0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int d = e0 + q;
    float dk = sd[d * 33 + 32];
#pragma unroll 8
    for (int e = 0; e < 32; ++e) dk += sv[n][e] * sd[d * 33 + e];
    drow[inner + q] = (T)(dk * dphi_f(sraw[n][d]));
    float dv = 0.f;  // dV[n][e = d] = sum_dd K[n][dd] * dkv[dd][e]
#pragma unroll 8
    for (int dd = 0; dd < 32; ++dd) dv += sk[n][dd] * sd[dd * 33 + d];
    drow[2 * inner + q] = (T)dv;
  }
}
hipError_t launch_linattn_bwd_q(int dtype, const AttnBwdArgs& a, hipStream_t s) {
  if (a.N % 64) return hipErrorInvalidValue;
  dim3 grid(a.N / 64, a.heads, a.B);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(linattn_bwd_q_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(linattn_bwd_q_kernel<half_t>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(linattn_bwd_q_kernel<bf16_t>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_linattn_bwd_kv(int dtype, const AttnBwdArgs& a, hipStream_t s) {
  if (a.N % 64) return hipErrorInvalidValue;
  dim3 grid(a.N / 64, a.heads, a.B);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(linattn_bwd_kv_kernel<float>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(linattn_bwd_kv_kernel<half_t>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(linattn_bwd_kv_kernel<bf16_t>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace llie
