// GroupNorm-2 statistics of the recompute form from the Gram matrix of the block input (gfx950, 2-byte compute types).
//
//   h1 = 6 W1 a',  a' = relu6(norm1(x)) / 6          (efficient_unet.py:207-212: norm1, ReLU6, expand, norm2)
//   sum_px h1[c]   = 6  w_c . m          m = sum_px a'            (K values)
//   sum_px h1[c]^2 = 36 w_c^T G w_c      G = sum_px a' a'^T       (K x K, symmetric)
//
// expand_stats_kernel (irbx.hip) gets the same two sums by running the whole expand GEMM a second time and squaring its
// accumulators: 4 K MACs per pixel and input channel plus two VALU operations per product.  G costs K / 2 MACs per pixel and
// input channel on MFMA and nothing else, so the statistics pass becomes a plain read of x:
//
//   gram_stats_kernel     grid (P / RP, 1, B): a workgroup owns RP pixels of one image, a quarter per wave.  A wave
//                         activates 32 pixels at a time into an LDS tile [pixel][K] of its own; MFMA fragments "channel x
//                         8 pixels" come out of the tile by ds_read_b64_tr_b16 (the hardware transpose read), and one
//                         fragment serves as A and as B operand (D[i][j] += sum_px a[px][i] a[px][j]: any order of the
//                         pixels inside a fragment cancels).  No barrier in the loop.
//                         The workgroup's partial ([upper 32 x 32 blocks][m]) goes to a slab; the workgroup that draws an
//                         image's last ticket adds the image's partials in index order and writes the full symmetric G.
//   gram_finalize_kernel  grid (groups, B): w_c . m and w_c^T G w_c in fp64 for the group's channels, then GroupNorm +
//                         FiLM folded into one affine exactly as gn_finalize_kernel does (small.hip).
//
// No float atomics; partials are combined by index, not by arrival: bits do not depend on the batch or the schedule.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace llie {

namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 gf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ float gdot2(uint32_t x, uint32_t y, float c) {
  if constexpr (std::is_same<T, half_t>::value)
    return __builtin_amdgcn_fdot2(*reinterpret_cast<gf16x2*>(&x), *reinterpret_cast<gf16x2*>(&y), c, false);
  else
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<gbf16x2*>(&x), *reinterpret_cast<gbf16x2*>(&y), c, false);
}

// agent-scope (device-wide) accesses for the hand-over of the workgroup partials inside one kernel: the L2 of an XCD is not
// coherent with the other seven, so the partial is stored write-through and read past the reader's L2
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ f32x4 ld_agent4(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f32x4 r;
  r[0] = __uint_as_float((uint32_t)lo); r[1] = __uint_as_float((uint32_t)(lo >> 32));
  r[2] = __uint_as_float((uint32_t)hi); r[3] = __uint_as_float((uint32_t)(hi >> 32));
  return r;
}

// a' = clamp01(x * s + b) on one 16-byte slice (8 channels); tables in LDS (already divided by 6)
template <typename T>
__device__ __forceinline__ u32x4 gram_activate8(u32x4 x, const float* sc, const float* sh) {
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
  u32x4 o;
  o[0] = act_clamp01_pack<T>(x[0], s0[0], s0[1], b0[0], b0[1]);
  o[1] = act_clamp01_pack<T>(x[1], s0[2], s0[3], b0[2], b0[3]);
  o[2] = act_clamp01_pack<T>(x[2], s1[0], s1[1], b1[0], b1[1]);
  o[3] = act_clamp01_pack<T>(x[3], s1[2], s1[3], b1[2], b1[3]);
  return o;
}
}  // namespace

// LDS row pitch of the activated tile: a 32-lane half of ds_read_b64_tr_b16 reads 4 pixel rows x 32 channels (64 bytes per
// row); conflict-free when the four rows start 16 banks apart: 64-byte rows (K = 32) or 192-byte rows (K = 64 padded, K = 96)
template <int K> struct GramPitch { static constexpr int value = K == 32 ? 64 : 192; };

template <typename T, int KS>
__global__ void __launch_bounds__(256) gram_stats_kernel(const GramArgs a) {
  constexpr int K = 16 * KS, NF = K / 32, NPAIR = NF * (NF + 1) / 2, PITCH = GramPitch<K>::value;
  constexpr int GSZ = NPAIR * 1024 + K;
  static_assert(KS == 2 || KS == 4 || KS == 6, "K in {32, 64, 96}");
  typedef typename Elem<T>::vec_t vec_t;
  constexpr int TILE = 32 * PITCH;                      // one wave's 32 pixels
  constexpr int SA_BYTES = 4 * TILE > 16384 ? 4 * TILE : 16384;  // also the cross-wave staging of the epilogue (4 x 4 KB)
  __shared__ __align__(16) unsigned char sA[SA_BYTES];
  __shared__ __align__(16) float aff1[2][K];
  __shared__ float smsum[4][K];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, wg = blockIdx.x, nwg = gridDim.x;
  const T* x0 = reinterpret_cast<const T*>(a.x0) + (size_t)b * a.P * a.c0;
  const T* x1 = a.x1 ? reinterpret_cast<const T*>(a.x1) + (size_t)b * a.P * a.c1 : nullptr;
  for (int i = tid; i < K; i += 256) {
    aff1[0][i] = a.as1[(size_t)b * K + i];
    aff1[1][i] = a.ab1[(size_t)b * K + i];
  }
  // Waves run free: wave w owns pixels [w, w + 1) * RP / 4 of the workgroup's range and walks them 32 at a time through a
  // tile of its own -- LDS operations of one wave execute in order, so the transposed reads see the wave's stores and the
  // next step's stores come after them: no barrier in the loop.  Vector v = lane + 64 j of a step -> pixel v / (2 KS), channel
  // vector v % (2 KS).  PF steps of loads in flight per lane: the pass is a plain read of x.
  const int nsteps = a.RP / 128;  // steps of 32 pixels per wave
  const size_t p_first = (size_t)wg * a.RP + (size_t)wave * (a.RP / 4);
  constexpr int PF = KS == 2 ? 8 : (KS == 4 ? 4 : 2);
  u32x4 raw[PF][KS];
  auto load = [&](int step, u32x4 (&dst)[KS]) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int v = lane + j * 64;
      const size_t pix = p_first + (size_t)step * 32 + v / (2 * KS);
      const int k = (v % (2 * KS)) * 8;
      dst[j] = k < a.c0 ? *reinterpret_cast<const u32x4*>(x0 + pix * a.c0 + k) : *reinterpret_cast<const u32x4*>(x1 + pix * a.c1 + (k - a.c0));
    }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nsteps) load(u, raw[u]);

  f32x16 acc[NPAIR];
#pragma unroll
  for (int q = 0; q < NPAIR; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float msum[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) msum[f] = 0.f;
  uint32_t ones2;
  {
    typedef T t2 __attribute__((ext_vector_type(2)));
    t2 o = {(T)1.f, (T)1.f};
    ones2 = *reinterpret_cast<uint32_t*>(&o);
  }
  // transposed-read roles: 16-lane group gq = lane >> 4 takes channels 16 (gq & 1) .. + 15 of a 32-channel fragment and
  // pixels 8 (gq >> 1) .. + 7 of a 16-pixel k-step; lane 4 q + p of the group supplies the address of pixel row q, channels 4 p .. 4 p + 3
  const int li = lane & 15, gq = lane >> 4;
  unsigned char* buf = sA + wave * TILE;
  const int tr_off = (8 * (gq >> 1) + (li >> 2)) * PITCH + (16 * (gq & 1) + 4 * (li & 3)) * 2;
  wg_barrier();  // aff1 staged
  for (int step0 = 0; step0 < nsteps; step0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int step = step0 + u;
      if (step >= nsteps) break;  // uniform (short images: fewer steps than the ring holds)
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const int v = lane + j * 64;
        const int k = (v % (2 * KS)) * 8;
        *reinterpret_cast<u32x4*>(buf + (v / (2 * KS)) * PITCH + k * 2) = gram_activate8<T>(raw[u][j], &aff1[0][k], &aff1[1][k]);
      }
      if (step + PF < nsteps) load(step + PF, raw[u]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        vec_t fr[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const unsigned char* pa = buf + tr_off + ks * 16 * PITCH + f * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * PITCH));
          u32x4 w;
          w[0] = reinterpret_cast<const uint32_t*>(&lo)[0]; w[1] = reinterpret_cast<const uint32_t*>(&lo)[1];
          w[2] = reinterpret_cast<const uint32_t*>(&hi)[0]; w[3] = reinterpret_cast<const uint32_t*>(&hi)[1];
          fr[f] = reinterpret_cast<const vec_t&>(w);
#pragma unroll
          for (int q = 0; q < 4; ++q) msum[f] = gdot2<T>(w[q], ones2, msum[f]);
        }
        int q = 0;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int j = i; j < NF; ++j) {
            acc[q] = mfma16<T>(fr[i], fr[j], acc[q]);
            ++q;
          }
      }
    }
  }
  // ---- workgroup partial: the four waves' accumulators added in wave order, one 32 x 32 block at a time
  float* red = reinterpret_cast<float*>(sA);  // [4 waves][1024]
  float* part = a.part + ((size_t)b * nwg + wg) * GSZ;
#pragma unroll
  for (int q = 0; q < NPAIR; ++q) {
    wg_barrier();  // tile reads / the previous block's sums are done
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + mfma_row(r, lane) * 32 + (lane & 31)] = acc[q][r];
    wg_barrier();
    for (int e = tid; e < 1024; e += 256) st_agent(part + q * 1024 + e, ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e]);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float t = msum[f] + __shfl_xor(msum[f], 32, 64);  // the two k-slice halves of a channel
    if (lane < 32) smsum[wave][f * 32 + lane] = t;
  }
  wg_barrier();
  if (tid < K) st_agent(part + NPAIR * 1024 + tid, ((smsum[0][tid] + smsum[1][tid]) + smsum[2][tid]) + smsum[3][tid]);

  // ---- the image's last workgroup adds the partials in index order and writes G in full ([K][K] row-major, then m).
  // The partial was stored write-through (agent scope); once those stores are acknowledged the ticket may be drawn.  No
  // __threadfence(): its release half writes back the whole L2 of the XCD -- from each of 512 workgroups, next to another
  // kernel's output stream, that made this pass three times longer than the GEMM pass it replaces.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
  if (tid == 0) s_last = __hip_atomic_fetch_add(a.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nwg - 1) ? 1u : 0u;
  wg_barrier();
  if (!s_last) return;
  const float* pimg = a.part + (size_t)b * nwg * GSZ;
  float* gt = a.gtot + (size_t)b * (K * K + K);
  // 16-byte loads, up to 16 partials in flight per thread (one dependent load per partial made this tail longer than the
  // whole pass); the sum runs in partial order whatever the grouping
  for (int e4 = tid; e4 < GSZ / 4; e4 += 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < nwg; w0 += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (w0 + u < nwg) v[u] = ld_agent4(pimg + (size_t)(w0 + u) * GSZ + 4 * e4);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (w0 + u < nwg) s += v[u];
    }
    const int e = 4 * e4;
    if (e < NPAIR * 1024) {
      const int q = e >> 10, i = (e >> 5) & 31, j = e & 31;
      int bi = 0, bj = 0, cnt = 0;
#pragma unroll
      for (int ii = 0; ii < NF; ++ii)
#pragma unroll
        for (int jj = ii; jj < NF; ++jj) {
          if (cnt == q) { bi = ii; bj = jj; }
          ++cnt;
        }
      *reinterpret_cast<f32x4*>(gt + (bi * 32 + i) * K + bj * 32 + j) = s;
      if (bi != bj) {
#pragma unroll
        for (int t = 0; t < 4; ++t) gt[(bj * 32 + j + t) * K + bi * 32 + i] = s[t];
      }
    } else {
      *reinterpret_cast<f32x4*>(gt + K * K + (e - NPAIR * 1024)) = s;
    }
  }
  if (tid == 0) a.tickets[b] = 0u;  // ready for the next launch on this buffer
}

// ---------------------------------------------------------------------------------------------
// One block per (group, image): Chid = 4 K, 32 groups -> K / 8 channels per group, K / 32 per wave; lanes own rows of G.
// Pure latency between two big launches: every global operand is requested before the first is used.
template <typename T, int K>
__global__ void __launch_bounds__(256) gram_finalize_kernel(const GramFinalizeArgs a) {
  constexpr int KP = K + 4;       // row pitch: lanes walk rows with ds_read_b128 (36 / 68 / 100 dwords: conflict-free)
  constexpr int CG = K / 8;       // channels per group
  constexpr int CPW = K / 32;     // ... per wave
  constexpr int NV = K * K / 4 / 256;  // 16-byte vectors of G per thread (1 / 4 / 9)
  __shared__ __align__(16) float sG[K * KP];
  __shared__ float sm[K];
  __shared__ __align__(16) float sw[CG * K];
  __shared__ double ssum[2 * CG];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c_lo = g * CG;
  const float* gt = a.gtot + (size_t)b * (K * K + K);
  const T* w1 = reinterpret_cast<const T*>(a.w1) + (size_t)c_lo * K;
  f32x4 gv[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) gv[u] = *reinterpret_cast<const f32x4*>(gt + 4 * (tid + 256 * u));
  const float mv = tid < K ? gt[K * K + tid] : 0.f;
  float wv[(CG * K + 255) / 256];
#pragma unroll
  for (int u = 0; u < (CG * K + 255) / 256; ++u) wv[u] = tid + 256 * u < CG * K ? (float)w1[tid + 256 * u] : 0.f;
  float p_ga = 0.f, p_be = 0.f, p_fs = 0.f, p_fh = 0.f;
  if (tid < CG) {
    p_ga = a.gamma[c_lo + tid];
    p_be = a.beta[c_lo + tid];
    if (a.film) {
      const float* f = a.film + (size_t)b * a.film_stride;
      p_fs = f[c_lo + tid];
      p_fh = f[a.Chid + c_lo + tid];
    }
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int e = 4 * (tid + 256 * u), i = e / K, j = e % K;
#pragma unroll
    for (int t = 0; t < 4; ++t) sG[i * KP + j + t] = gv[u][t];  // (one ds_write_b128: i * KP + j is a multiple of 4)
  }
  if (tid < K) sm[tid] = mv;
#pragma unroll
  for (int u = 0; u < (CG * K + 255) / 256; ++u)
    if (tid + 256 * u < CG * K) sw[tid + 256 * u] = wv[u];
  wg_barrier();
  // wave: channels wave * CPW .. + CPW - 1 of the group; lane: rows i = lane (and lane + 64 for K = 96).
  // t_i[c] = sum_j G[i][j] w_c[j] in fp32 (one read of G per CPW products), then w_c . t and w_c . m in fp64
  double s1[CPW], s2[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) s1[c] = s2[c] = 0.0;
  for (int i = lane; i < K; i += 64) {
    const float* gr = sG + i * KP;
    // in fp64: G's entries are all positive (up to P) while w has mixed signs, so the cancellation of the quadratic form
    // starts in this row product; at K <= 96 the kernel is latency-bound and the wider arithmetic costs nothing measurable
    double t[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) t[c] = 0.0;
#pragma unroll 4
    for (int j = 0; j < K; j += 4) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gr + j);
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(sw + (wave * CPW + c) * K + j);  // same address in every lane
#pragma unroll
        for (int e = 0; e < 4; ++e) t[c] = __builtin_fma((double)g4[e], (double)w4[e], t[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const double wi = (double)sw[(wave * CPW + c) * K + i];
      s2[c] += wi * t[c];
      s1[c] += wi * (double)sm[i];
    }
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const double r1 = wave_sum(s1[c]), r2 = wave_sum(s2[c]);
    if (lane == 0) {
      ssum[wave * CPW + c] = 6.0 * r1;        // h1 = 6 W1 a'
      ssum[CG + wave * CPW + c] = 36.0 * r2;
    }
  }
  wg_barrier();
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int ci = 0; ci < CG; ++ci) {
    t1 += ssum[ci];
    t2 += ssum[CG + ci];
  }
  const double n = (double)CG * (double)a.P;
  const double mean = t1 / n;
  double var = t2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float fmean = (float)mean;
  if (tid < CG) {
    const int c = c_lo + tid;
    const float ga = p_ga * rstd;
    float sc = ga, sh = p_be - fmean * ga;
    if (a.film) {
      const float fs = 1.f + p_fs, fh = p_fh;
      sc *= fs;
      sh = sh * fs + fh;
    }
    if (a.post_scale != 0.f) {
      sc *= a.post_scale;
      sh *= a.post_scale;
    }
    a.as[(size_t)b * a.Chid + c] = sc;
    a.ab[(size_t)b * a.Chid + c] = sh;
  }
}

// ---------------------------------------------------------------------------------------------
int gram_rows(int K, int P) {  // pixels per workgroup: batch-independent (it fixes the summation order)
  // 16 partials per image at 256 x 256 (32 made the pass 30-40 % longer at B = 32: twice the epilogues and write-through
  // stores; at B = 1 they were 15 % faster), at least 16 where the image allows, never below 512 pixels
  int rp = 4096;
  while (rp > 512 && (P % rp || P / rp < 16)) rp >>= 1;
  return rp;
}
size_t gram_part_floats(int K, int P) {
  const int nf = K / 32;
  return (size_t)(P / gram_rows(K, P)) * (size_t)(nf * (nf + 1) / 2 * 1024 + K);
}
bool gram_supported(int dtype, int K, int c0, int P) {
  if (dtype != 1 && dtype != 2) return false;
  if (K != 32 && K != 64 && K != 96) return false;
  return c0 % 8 == 0 && P % 512 == 0 && P % gram_rows(K, P) == 0;
}

template <typename T>
static hipError_t launch_gram_t(const GramArgs& a, int K, hipStream_t s) {
  const dim3 grid(a.P / a.RP, 1, a.B);
  switch (K) {
    case 32: hipLaunchKernelGGL((gram_stats_kernel<T, 2>), grid, dim3(256), 0, s, a); break;
    case 64: hipLaunchKernelGGL((gram_stats_kernel<T, 4>), grid, dim3(256), 0, s, a); break;
    case 96: hipLaunchKernelGGL((gram_stats_kernel<T, 6>), grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gram_stats(int dtype, const GramArgs& a0, hipStream_t s) {
  GramArgs a = a0;
  const int K = a.c0 + a.c1;
  if (!gram_supported(dtype, K, a.c0, a.P) || !a.x0 || (a.c1 && !a.x1) || !a.as1 || !a.ab1 || !a.part || !a.gtot || !a.tickets || a.B <= 0)
    return hipErrorInvalidValue;
  a.RP = gram_rows(K, a.P);
  note_kernel("gram_stats_kernel");
  return dtype == 1 ? launch_gram_t<half_t>(a, K, s) : launch_gram_t<bf16_t>(a, K, s);
}

template <typename T>
static hipError_t launch_gram_finalize_t(const GramFinalizeArgs& a, hipStream_t s) {
  const dim3 grid(a.groups, a.B);
  switch (a.K) {
    case 32: hipLaunchKernelGGL((gram_finalize_kernel<T, 32>), grid, dim3(256), 0, s, a); break;
    case 64: hipLaunchKernelGGL((gram_finalize_kernel<T, 64>), grid, dim3(256), 0, s, a); break;
    case 96: hipLaunchKernelGGL((gram_finalize_kernel<T, 96>), grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_gram_finalize(int dtype, const GramFinalizeArgs& a, hipStream_t s) {
  // the recompute blocks of this engine: Chid = 4 K, nn.GroupNorm(32, Chid)
  if ((dtype != 1 && dtype != 2) || !a.gtot || !a.w1 || !a.as || !a.ab || a.groups != 32 || a.Chid != 4 * a.K || a.B <= 0) return hipErrorInvalidValue;
  note_kernel("gram_finalize_kernel");
  return dtype == 1 ? launch_gram_finalize_t<half_t>(a, s) : launch_gram_finalize_t<bf16_t>(a, s);
}

}  // namespace llie
