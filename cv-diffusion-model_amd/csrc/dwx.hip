// Fused expand (1x1, MFMA) + depthwise 3x3 for gfx950: the "recompute" form of one
// InvertedResidualBlock's middle (efficient_unet.py:207-223) for 2-byte compute types.
//
//   h2 = dw3x3( relu6( aff2( W1 . relu6( aff1(x) ) ) ) )        aff1 = GroupNorm-1, aff2 = GroupNorm-2 + FiLM
//
// The 4x-expanded tensor h1 is the largest object of the network; the unfused path writes it (K1) and
// reads it back (depthwise).  Here K1 only produces h1's GroupNorm statistics (pw_gemm with `nostore`)
// and this kernel rebuilds h1 row by row from the narrow block input x, so per block
// 2*Chid*P - ~1.1*Cin*P elements of traffic disappear for ~2x the (small) expand FLOPs.
//
// A workgroup owns a strip of 32 output pixels x TYL rows x 64 hidden channels and streams rows:
//   (a) the prefetched x row (34 pixels incl. halo, Cin channels) is activated (aff1 + ReLU6) into LDS;
//   (b) waves 0/1 each compute one 32-channel x 32-pixel block of h1 with K/16 MFMAs -- weights are the
//       A operand (rows = channels), the activated row the B operand (columns = pixels) -- apply
//       aff2 + ReLU6 in registers and park the result in the depthwise row ring; waves 2/3 meanwhile
//       produce the two halo pixels with plain dot products;
//   (c) all four waves run the depthwise update exactly as dwconv3x3_kernel does (3 ds_read_b128 per
//       output, three rolling row accumulators).
// Zero padding is applied to the depthwise INPUT (after aff2 + ReLU6), like the reference's conv.
#include <string>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace llie {

constexpr int kDxPF = 2;  // x rows are narrow and re-read by Chid/64 workgroups (L2 hits): 2 rows in flight suffice

template <typename T, int KS>  // K = Cin = 16 * KS
__global__ void __launch_bounds__(256) dwx_kernel(const DwxArgs a, const int TYL) {
  constexpr int TX = 32, PW = TX + 2, NT = 256, CC = 64, VEC = 8, PF = kDxPF;
  constexpr int K = 16 * KS, KP = K + 8;       // padded LDS row (conflict-free ds_read_b128)
  constexpr int KV = K / 8;                    // 16-byte vectors per pixel of x
  constexpr int NV = PW * KV;                  // vectors per x row
  constexpr int XV = (NV + NT - 1) / NT;       // vectors per thread per row
  typedef typename Elem<T>::vec_t vec_t;
  __shared__ __align__(16) T arow[PW * KP];
  __shared__ __align__(16) T w1s[CC * KP];
  __shared__ vec_t ring[2][PW * 8];
  __shared__ float aff1s[2][K];
  __shared__ __align__(16) float aff2s[2][CC];
  __shared__ __align__(16) T wds[9 * CC];  // depthwise weights of this channel chunk, packed in T
  __shared__ float red[8 * (NT / 64) * CC];  // [8-row segment][wave][channel]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = tid & 7, xl = tid >> 3;
  const int tiles_x = a.W / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TX, y0 = ty * TYL;
  const int cbase = blockIdx.y * CC;
  const int b = blockIdx.z;
  const T* xs0 = reinterpret_cast<const T*>(a.x0) + (size_t)b * a.H * a.W * a.c0;
  const T* xs1 = a.x1 ? reinterpret_cast<const T*>(a.x1) + (size_t)b * a.H * a.W * a.c1 : nullptr;
  T* out = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * a.Chid + cbase + cl * VEC;
  const T* w1 = reinterpret_cast<const T*>(a.w1);

  // ---- per-strip constants
  for (int i = tid; i < K; i += NT) {
    aff1s[0][i] = a.as1[(size_t)b * K + i];
    aff1s[1][i] = a.ab1[(size_t)b * K + i];
  }
  if (tid < CC) {
    aff2s[0][tid] = a.as2[(size_t)b * a.Chid + cbase + tid];
    aff2s[1][tid] = a.ab2[(size_t)b * a.Chid + cbase + tid];
  }
  for (int i = tid; i < CC * KV; i += NT) {
    const int ch = i / KV, kv = (i % KV) * 8;
    *reinterpret_cast<vec_t*>(w1s + ch * KP + kv) = ld_vec<T>(w1 + (size_t)(cbase + ch) * K + kv);
  }
  for (int i = tid; i < 9 * CC; i += NT) wds[i] = (T)a.wd[(size_t)(i / CC) * a.Chid + cbase + i % CC];
  // MFMA waves: weight fragments (A operand)
  vec_t wfrag[KS];
  const int r32 = lane & 31, hh = lane >> 5;
  if (wave < 2) {
#pragma unroll
    for (int s = 0; s < KS; ++s) wfrag[s] = ld_vec<T>(w1 + (size_t)(cbase + 32 * wave + r32) * K + 16 * s + 8 * hh);
  }

  const int nrows = TYL + 2;  // input rows y0-1 .. y0+TYL
  vec_t pre[PF][XV];
  auto issue = [&](int r, vec_t (&v)[XV]) {
    const int gy = y0 - 1 + r;
    if (r < nrows && gy >= 0 && gy < a.H) {
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        const int i = tid + j * NT;
        if (NV % NT == 0 || i < NV) {
          const int px = i / KV, cv = (i % KV) * 8;
          const int gx = x0 - 1 + px;
          if (gx >= 0 && gx < a.W) {
            const size_t pix = (size_t)gy * a.W + gx;
            v[j] = cv < a.c0 ? ld_vec<T>(xs0 + pix * a.c0 + cv) : ld_vec<T>(xs1 + pix * a.c1 + (cv - a.c0));
          }
        }
      }
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) issue(j, pre[j]);

  float acc0[VEC], acc1[VEC], acc2[VEC], psum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc0[e] = acc1[e] = acc2[e] = psum[e] = 0.f;
  T* ringT[2] = {reinterpret_cast<T*>(ring[0]), reinterpret_cast<T*>(ring[1])};
  __syncthreads();  // per-strip constants staged

  for (int r0 = 0; r0 < nrows; r0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int r = r0 + j;
      if (r >= nrows) break;  // uniform over the workgroup
      const int gy = y0 - 1 + r;
      const bool row_ok = gy >= 0 && gy < a.H;
      // ---- (a) activate the x row into LDS
      if (row_ok) {
#pragma unroll
        for (int jj = 0; jj < XV; ++jj) {
          const int i = tid + jj * NT;
          if (NV % NT == 0 || i < NV) {
            const int px = i / KV, cv = (i % KV) * 8;
            const int gx = x0 - 1 + px;
            vec_t v;
            if (gx >= 0 && gx < a.W) {
              float f[VEC];
              vec_to_f32<T>(pre[j][jj], f);
#pragma unroll
              for (int e = 0; e < VEC; ++e) f[e] = relu6f(f[e] * aff1s[0][cv + e] + aff1s[1][cv + e]);
              v = f32_to_vec<T>(f);
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) v[e] = (T)0.f;
            }
            *reinterpret_cast<vec_t*>(arow + px * KP + cv) = v;
          }
        }
      }
      issue(r + PF, pre[j]);
      __syncthreads();
      // ---- (b) rebuild the h1 row (aff2 + ReLU6 applied) into the depthwise ring
      T* rg = ringT[j & 1];  // PF is even: r & 1 == j & 1
      if (wave < 2) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        if (row_ok) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const vec_t bv = *reinterpret_cast<const vec_t*>(arow + (1 + r32) * KP + 16 * s + 8 * hh);
            if constexpr (std::is_same<T, half_t>::value) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[s], bv, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s], bv, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          typedef T t4 __attribute__((ext_vector_type(4)));
          t4 o;
          const int chl = 32 * wave + 8 * g4 + 4 * hh;  // this lane's 4 consecutive channels of the block
          const f32x4 sc4 = *reinterpret_cast<const f32x4*>(&aff2s[0][chl]);
          const f32x4 sh4 = *reinterpret_cast<const f32x4*>(&aff2s[1][chl]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int q = g4 * 4 + e;
            o[e] = row_ok ? (T)relu6f(acc[q] * sc4[e] + sh4[e]) : (T)0.f;
          }
          // channels 32*wave + 8*g4 + 4*hh + {0..3} of pixel r32 (ring pixel slot r32 + 1)
          *reinterpret_cast<t4*>(rg + (1 + r32) * CC + 32 * wave + 8 * g4 + 4 * hh) = o;
        }
      } else {
        const int u = tid - 128, side = u >> 6, ch = u & 63;
        const int pxh = side ? PW - 1 : 0;
        const int gxh = x0 - 1 + pxh;
        float v = 0.f;
        if (row_ok && gxh >= 0 && gxh < a.W) {
          float s = 0.f;
#pragma unroll
          for (int kk = 0; kk < K; kk += 8) {
            const vec_t wv = *reinterpret_cast<const vec_t*>(w1s + ch * KP + kk);
            const vec_t av = *reinterpret_cast<const vec_t*>(arow + pxh * KP + kk);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)wv[e] * (float)av[e];
          }
          v = relu6f(s * aff2s[0][ch] + aff2s[1][ch]);
        }
        rg[pxh * CC + ch] = (T)v;
      }
      __syncthreads();
      // ---- (c) depthwise update (as dwconv3x3_kernel)
      const vec_t* buf = ring[j & 1];
      vec_t f[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) f[kx] = buf[(xl + kx) * 8 + cl];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const vec_t w2 = *reinterpret_cast<const vec_t*>(wds + (6 + kx) * CC + cl * VEC);
        const vec_t w1v = *reinterpret_cast<const vec_t*>(wds + (3 + kx) * CC + cl * VEC);
        const vec_t w0 = *reinterpret_cast<const vec_t*>(wds + (0 + kx) * CC + cl * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          acc0[e] += (float)w2[e] * (float)f[kx][e];
          acc1[e] += (float)w1v[e] * (float)f[kx][e];
          acc2[e] += (float)w0[e] * (float)f[kx][e];
        }
      }
      if (r >= 2) {
        vec_t ov = f32_to_vec<T>(acc0);
        st_vec<T>(out + ((size_t)(y0 + r - 2) * a.W + x0 + xl) * a.Chid, ov);
#pragma unroll
        for (int e = 0; e < VEC; ++e) psum[e] += (float)ov[e];
        if (a.pool && ((r - 2) & 7) == 7) pool_segment_flush<VEC>(psum, red + (((r - 2) >> 3) * (NT / 64) + wave) * CC, lane);
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        acc0[e] = acc1[e];
        acc1[e] = acc2[e];
        acc2[e] = 0.f;
      }
    }
  }
  if (a.pool) {
    __syncthreads();
    const int ntiles = tiles_x * (a.H / kPoolSegRows);
    pool_segments_store<CC, NT>(red, TYL / kPoolSegRows, tid, a.pool + (size_t)b * ntiles * a.Chid + cbase, a.Chid,
                                ty * (TYL / kPoolSegRows), tiles_x, tx);
  }
}

bool dwx_supported(int dtype, int Cin, int Chid, int H, int W) {
  return (dtype == 1 || dtype == 2) && (Cin == 32 || Cin == 64 || Cin == 96 || Cin == 128) && Chid % 64 == 0 && W % 32 == 0 && H % 8 == 0;
}

template <typename T, int KS>
static hipError_t launch_dwx_k(const DwxArgs& a, hipStream_t s) {
  const int tyl = dw_pick_tyl(a.B, a.H, a.W, a.Chid / 64);  // same strips as dwconv3x3 (pool slab layout)
  dim3 grid((a.W / 32) * (a.H / tyl), a.Chid / 64, a.B);
  static const std::string name = std::string("dwx_kernel<") + TypeName<T>::value + ", " + std::to_string(KS) + ">";
  note_kernel(name.c_str());
  hipLaunchKernelGGL((dwx_kernel<T, KS>), grid, dim3(256), 0, s, a, tyl);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_dwx_t(const DwxArgs& a, hipStream_t s) {
  switch (a.c0 + a.c1) {
    case 32: return launch_dwx_k<T, 2>(a, s);
    case 64: return launch_dwx_k<T, 4>(a, s);
    case 96: return launch_dwx_k<T, 6>(a, s);
    case 128: return launch_dwx_k<T, 8>(a, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_dwx(int dtype, const DwxArgs& a, hipStream_t s) {
  if (!dwx_supported(dtype, a.c0 + a.c1, a.Chid, a.H, a.W) || a.c0 % 8 || (a.c1 && !a.x1)) return hipErrorInvalidValue;
  return dtype == 1 ? launch_dwx_t<half_t>(a, s) : launch_dwx_t<bf16_t>(a, s);
}

}  // namespace llie
