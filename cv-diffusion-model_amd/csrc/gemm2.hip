// Pointwise GEMM, LDS-DMA pipelined form (fp16 engine, 128-row tiles) for gfx950.
//
// Same contract as pw_gemm_kernel (gemm.hip: K-concatenated A segments with per-(image, channel) affine +
// activation prologue, W[N][K], bias / residual / statistics epilogue) but a different main loop.
// Profiling of the register-staged kernel showed its loads in flight only a small fraction of each K
// step (~17 B/clk/CU of operand traffic, MFMA pipe 22 % busy) and deeper register prefetch costs the
// occupancy it needs.  Here operand tiles go global -> LDS by DMA (`global_load_lds_dwordx4`, no VGPRs)
// into a 4-deep ring, three K chunks always in flight, one raw barrier per chunk, counted vmcnt:
//   * a stage holds the raw A tile [128][32] and the W tile [BN][32] as unpadded 64-byte rows; bank
//     conflicts are avoided by XOR-swizzling the 16-byte slot with (row>>2)&3 -- on the DMA's per-lane
//     SOURCE address (the LDS side of a DMA is lane-linear) and again on the fragment read;
//   * the A prologue (GroupNorm affine + ReLU6, or the SE gate) is applied to the fragments after the
//     LDS read, from an fp16 (scale, bias) table of the whole K staged once per workgroup, as packed
//     f16 math.
#include <string>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace llie {

#define LLIE_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int BN, int WM, int WN>
__global__ void __launch_bounds__(WM* WN * 64) pw_gemm2_kernel(const GemmArgs g) {
  typedef half_t T;
  typedef f16x8 vec_t;
  constexpr int BM = 128, NS = 4, NT = WM * WN * 64, NW = NT / 64, VEC = 8;
  constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  constexpr int A_INSTR = BM / 16, W_INSTR = BN / 16, TI = A_INSTR + W_INSTR;  // 1 KB DMA instructions per stage
  constexpr int DPW = TI / NW;                                                 // per wave (floor; used for waits)
  constexpr int STAGE = TI * 1024;
  constexpr int CP = BN + 4;
  static_assert(DPW >= 1 && DPW <= 4, "vmcnt immediates below assume 1..4 DMA instructions per wave and stage");

  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* ring = smem;
  T* afft = reinterpret_cast<T*>(smem + NS * STAGE);  // [2][K]: scale, bias
  float* sC = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nb = g.N / BN;
  const int mt = blockIdx.x / nb, ntile = blockIdx.x % nb;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int img = m0 / g.P;
  const int koff1 = g.seg[0].ch, koff2 = g.seg[0].ch + g.seg[1].ch;
  const T* wbase = reinterpret_cast<const T*>(g.w);

  // ---- affine table of the whole K for this tile's image (identity where a segment has none)
  for (int k = tid; k < g.K; k += NT) {
    const int s = (g.nseg > 1 && k >= koff1) + (g.nseg > 2 && k >= koff2);
    const GemmSeg sg = g.seg[s];
    const int kl = k - (s == 0 ? 0 : (s == 1 ? koff1 : koff2));
    const float sc = sg.as ? sg.as[(size_t)img * sg.aff_ld + kl] : 1.f;
    const float bi = (sg.as && sg.ab) ? sg.ab[(size_t)img * sg.aff_ld + kl] : 0.f;
    afft[k] = (T)sc;
    afft[g.K + k] = (T)bi;
  }
  __syncthreads();  // table visible; every ordinary load retired before the DMA stream starts (vmcnt is counted below)

  // ---- DMA of K chunk c into ring slot sl: lane i of instruction t writes LDS bytes [t*1024 + 16 i, +16)
  auto dma = [&](int c, int sl) {
    const int k0 = c * 32;
    const int s = (g.nseg > 1 && k0 >= koff1) + (g.nseg > 2 && k0 >= koff2);
    const GemmSeg sg = g.seg[s];
    const int cl = k0 - (s == 0 ? 0 : (s == 1 ? koff1 : koff2));
    const T* abase = reinterpret_cast<const T*>(sg.ptr);
    unsigned char* dst = ring + sl * STAGE;
    for (int t = wave; t < TI; t += NW) {
      const int row = (t < A_INSTR ? t : t - A_INSTR) * 16 + (lane >> 2);
      const int lslot = (lane & 3) ^ ((row >> 2) & 3);
      const T* src = t < A_INSTR ? abase + (size_t)(m0 + row) * sg.ch + cl + lslot * 8
                                 : wbase + (size_t)(n0 + row) * g.K + k0 + lslot * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + t * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = g.K / 32;
#pragma unroll
  for (int c = 0; c < NS - 1; ++c)
    if (c < nchunks) dma(c, c);

  const int lr = lane & 31, lh = lane >> 5;
  for (int c = 0; c < nchunks; ++c) {
    // chunk c landed <=> at most `ahead` younger DMA groups of this wave remain outstanding
    const int ahead = min(NS - 2, nchunks - 1 - c);
    if (ahead >= 2) {
      if (DPW == 1) LLIE_WAIT_VMCNT(2); else if (DPW == 2) LLIE_WAIT_VMCNT(4); else if (DPW == 3) LLIE_WAIT_VMCNT(6); else LLIE_WAIT_VMCNT(8);
    } else if (ahead == 1) {
      if (DPW == 1) LLIE_WAIT_VMCNT(1); else if (DPW == 2) LLIE_WAIT_VMCNT(2); else if (DPW == 3) LLIE_WAIT_VMCNT(3); else LLIE_WAIT_VMCNT(4);
    } else {
      LLIE_WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();  // every wave's share of chunk c is in LDS; everyone is done with chunk c-1
    if (c + NS - 1 < nchunks) dma(c + NS - 1, (c + NS - 1) % NS);  // refills the slot chunk c-1 used

    const unsigned char* st = ring + (c % NS) * STAGE;
    const int k0 = c * 32;
    const int s = (g.nseg > 1 && k0 >= koff1) + (g.nseg > 2 && k0 >= koff2);
    const bool aff = g.seg[s].as != nullptr;
    const bool clamp = g.seg[s].act == ACT_RELU6;
    f16x2 sc2[8], bi2[8];
    if (aff) {
      const vec_t* ps = reinterpret_cast<const vec_t*>(afft + k0 + 16 * lh);
      const vec_t* pb = reinterpret_cast<const vec_t*>(afft + g.K + k0 + 16 * lh);
      *reinterpret_cast<vec_t*>(&sc2[0]) = ps[0];
      *reinterpret_cast<vec_t*>(&sc2[4]) = ps[1];
      *reinterpret_cast<vec_t*>(&bi2[0]) = pb[0];
      *reinterpret_cast<vec_t*>(&bi2[4]) = pb[1];
    }
    T fa[MI][16], fb[NI][16];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = (wm * MI + i) * 32 + lr;
      const int f = (row >> 2) & 3;
      const unsigned char* p = st + row * 64;
      *reinterpret_cast<vec_t*>(&fa[i][0]) = *reinterpret_cast<const vec_t*>(p + (((2 * lh) ^ f) << 4));
      *reinterpret_cast<vec_t*>(&fa[i][8]) = *reinterpret_cast<const vec_t*>(p + (((2 * lh + 1) ^ f) << 4));
      if (aff) {
        f16x2* x = reinterpret_cast<f16x2*>(&fa[i][0]);
        const f16x2 lo = {(_Float16)0.f, (_Float16)0.f}, hi = {(_Float16)6.f, (_Float16)6.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          f16x2 y = x[q] * sc2[q] + bi2[q];
          if (clamp) y = __builtin_elementwise_min(__builtin_elementwise_max(y, lo), hi);
          x[q] = y;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = (wn * NI + j) * 32 + lr;
      const int f = (row >> 2) & 3;
      const unsigned char* p = st + A_INSTR * 1024 + row * 64;
      *reinterpret_cast<vec_t*>(&fb[j][0]) = *reinterpret_cast<const vec_t*>(p + (((2 * lh) ^ f) << 4));
      *reinterpret_cast<vec_t*>(&fb[j][8]) = *reinterpret_cast<const vec_t*>(p + (((2 * lh + 1) ^ f) << 4));
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) Mfma<T>::chunk(fa[i], fb[j], acc[i][j]);
  }
  __syncthreads();  // ring no longer needed: reuse it as the C staging tile

  // ---- epilogue (as pw_gemm_kernel): accumulators -> LDS (fp32) -> 16-byte row vectors
  constexpr int VR = BN / VEC, RPP = NT / VR, SROWS = WM * 32;
  static_assert(NT % VR == 0 && VR <= 64 && MI * WM * 32 == BM, "epilogue mapping");
  const int cv = tid % VR, r0 = tid / VR;
  float bias[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    bias[e] = g.bias ? g.bias[n0 + cv * VEC + e] : 0.f;
    s1[e] = s2[e] = 0.f;
  }
  T* outp = reinterpret_cast<T*>(g.out);
  const T* resp = reinterpret_cast<const T*>(g.res);
#pragma unroll
  for (int pi = 0; pi < MI; ++pi) {
    if (pi) __syncthreads();
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + mfma_row(r, lane);
        const int col = (wn * NI + j) * 32 + (lane & 31);
        sC[row * CP + col] = acc[pi][j][r];
      }
    __syncthreads();
    for (int srow = r0; srow < SROWS; srow += RPP) {
      const int row = ((srow >> 5) * MI + pi) * 32 + (srow & 31);
      float v[VEC];
      const float* pc = sC + srow * CP + cv * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = pc[e] + bias[e];
      const size_t o = (size_t)(m0 + row) * g.N + n0 + cv * VEC;
      if (resp) {
        float rr[VEC];
        ld_f32<T>(resp + o, rr);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += rr[e];
      }
      vec_t ov = f32_to_vec<T>(v);
      if (!g.nostore) st_vec<T>(outp + o, ov);
      if (g.stats) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float q = (float)ov[e];
          s1[e] += q;
          s2[e] += q * q;
        }
      }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int o = VR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    float* red = sC + SROWS * CP;
    if (lane < VR) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        red[(wave * 2 + 0) * BN + cv * VEC + e] = s1[e];
        red[(wave * 2 + 1) * BN + cv * VEC + e] = s2[e];
      }
    }
    __syncthreads();
    const int ntiles = g.P / BM, tile = (m0 % g.P) / BM;
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, c = i % BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * BN + c];
      g.stats[((size_t)(img * ntiles + tile) * 2 + which) * g.N + n0 + c] = t;
    }
  }
}

template <int BN, int WM, int WN>
static hipError_t launch2_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int NT = WM * WN * 64, NS = 4;
  constexpr size_t ring = (size_t)NS * (128 / 16 + BN / 16) * 1024;
  constexpr size_t ctile = (size_t)(WM * 32) * (BN + 4) * 4 + (size_t)(NT / 64) * 2 * BN * 4;
  static_assert(ctile <= ring, "C staging must fit in the ring");
  const size_t lds = ring + (size_t)2 * a.K * 2;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t attr_set = 0;
  if (lds > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm2_kernel<BN, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) return e;
    attr_set = 160 * 1024;
  }
  static const std::string name = std::string("pw_gemm2_kernel<") + std::to_string(BN) + ", " + std::to_string(WM) + ", " +
                                  std::to_string(WN) + ">";
  note_kernel(name.c_str());
  const unsigned grid = (unsigned)((a.M / 128) * (a.N / BN));
  hipLaunchKernelGGL((pw_gemm2_kernel<BN, WM, WN>), dim3(grid), dim3(NT), lds, s, a);
  return hipGetLastError();
}

bool pw_gemm2_supported(int dtype, const GemmArgs& a) {
  return dtype == 1 && a.P % 128 == 0 && a.N % 32 == 0 && a.K % 32 == 0 && a.K <= 4096;
}

hipError_t launch_pw_gemm2(const GemmArgs& a, hipStream_t s) {
  if (a.N % 128 == 0) return launch2_cfg<128, 2, 2>(a, s);
  if (a.N % 64 == 0) return launch2_cfg<64, 2, 2>(a, s);
  return launch2_cfg<32, 4, 1>(a, s);
}

}  // namespace llie
