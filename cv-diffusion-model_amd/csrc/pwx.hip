// Expanding pointwise GEMM (1x1 conv, N = 4 K) in activation-stationary form for gfx950, 2-byte compute types.
//
// Replaces the reference's `expand` conv of the wide InvertedResidualBlocks (efficient_unet.py:174 with norm1 + ReLU6
// :207-208 in the operand prologue and the statistics of norm2 :212 in the epilogue; virtual torch.cat :588 as K segments).
//
// These launches write four times what they read (K <= 512 in, N = 4 K out): the tile-per-workgroup kernel of gemm.hip
// spends them waiting -- every output tile re-reads and re-activates its A rows, pays two barriers per K chunk and
// pushes its C tile through LDS in bursts that all workgroups issue at the same moment (profiles/r03/
// gemm_stamp_baseline.txt: 27 k cycles per wave and tile for 1.5 k cycles of MFMA).  Here
//
//   * a workgroup (4 waves) owns 128 pixels and ALL its output channels (or 1 / nsplit of them on small grids); each
//     wave loads its 32 pixel rows once, coalesced, applies norm1 + ReLU6 (clamp01(z / 6), kernels.h: ACT_RELU6_S6)
//     once and keeps them as MFMA A-operand fragments IN REGISTERS for the whole launch (K / 4 VGPRs);
//   * the weights stream past: they are stored pre-packed in MFMA B-fragment order (and pre-multiplied by the 6 of the
//     ReLU6 carry), so a 32-channel block is K / 16 KB of contiguous memory that the LDS-DMA engine
//     (global_load_lds_dwordx4) drops into a double-buffered LDS ring with no staging registers and no bank conflicts
//     on the ds_read_b128 side; one barrier per buffer;
//   * the epilogue never touches LDS with the tile: accumulators (lane = channel, registers = pixels) are rounded to T,
//     the GroupNorm partial sums come from v_dot2 on the packed words (a lane's 16 pixels, one half-swap, one LDS word
//     per wave and channel), the tile is transposed ON THE MATRIX PIPE (two MFMAs against a 0/1 selection matrix: exact)
//     into lane = pixel, registers = channels; after v_permlane32_swap a lane owns 16-byte runs of one pixel's channels,
//     which go through a wave-private 4.5 KB LDS tile (no barrier) and leave, every second block, as four stores of
//     8 rows x 128 bytes -- whole cache lines (32-byte pieces straight from registers cost 30 % of the launch:
//     profiles/r03/pwx_store_shape.txt), a steady trickle under the next block's MFMAs instead of a burst per tile.
//
// Statistics are fixed per-(128-pixel tile, channel) slab entries summed in a fixed order: bitwise independent of the batch.
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace llie {

namespace {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef T t2 __attribute__((ext_vector_type(2)));
  t2 o;
  o[0] = (T)a;
  o[1] = (T)b;
  return *reinterpret_cast<uint32_t*>(&o);
}
// c + x.lo * y.lo + x.hi * y.hi on packed 2-byte pairs, fp32 accumulate (v_dot2_f32_f16 / v_dot2_f32_bf16)
template <typename T> __device__ __forceinline__ float dot2(uint32_t x, uint32_t y, float c) {
  if constexpr (std::is_same<T, half_t>::value)
    return __builtin_amdgcn_fdot2(*reinterpret_cast<f16x2_t*>(&x), *reinterpret_cast<f16x2_t*>(&y), c, false);
  else
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<bf16x2_v*>(&x), *reinterpret_cast<bf16x2_v*>(&y), c, false);
}
template <typename T> __device__ __forceinline__ f32x16 mfma_u(const u32x4& a, const u32x4& b, f32x16 c) {
  typedef typename Elem<T>::vec_t vec_t;
  return mfma16<T>(reinterpret_cast<const vec_t&>(a), reinterpret_cast<const vec_t&>(b), c);
}
template <typename T> constexpr uint32_t one_bits() { return std::is_same<T, half_t>::value ? 0x3C00u : 0x3F80u; }

constexpr int kStagePitch = 144;  // bytes per staged 64-channel row: 128 + one 16-byte pad (conflict-free ds_read_b128)
constexpr bool pwx_has_tile(int KS, int NBW, int ABL) { return ABL == 0 || NBW * KS * 1024 < 4 * 32 * kStagePitch; }
constexpr int pwx_lds_bytes(int KS, int NBW, int ABL) {
  return 2 * NBW * KS * 1024 + 2 * 4 * NBW * 64 * 4 + (pwx_has_tile(KS, NBW, ABL) ? 4 * 32 * kStagePitch : 0);
}

}  // namespace

// KS = K / 16 MFMA steps; NBW = 32-channel blocks per LDS buffer.
// ABL 6 = stores straight from registers (32 rows x 32 bytes per instruction; correct results).
// Diagnostic instantiations (llie_tune "pwx_ablate" / "pwx_stamp", timing studies only): ABL 1 = every store instruction
// writes 1 KB of contiguous memory (wrong layout, same bytes), 2 = no output stores, 3 / 4 / 5 = 64 / 128 / 256 bytes per
// row and instruction (wrong layout); STAMP = s_memtime per wave after the A phase and at the end.  Template parameters, not run-time flags: a flag inside the MFMA loop changes the code it measures.
template <typename T, int KS, int NBW, int ABL = 0, bool STAMP = false>
__global__ void __launch_bounds__(256, 2) pw_expand_kernel(const ExpandArgs g) {
  unsigned long long t_start = 0, t_a = 0, t_wait = 0;
  if constexpr (STAMP) t_start = __builtin_amdgcn_s_memtime();
  static_assert(sizeof(T) == 2 && KS % 4 == 0, "");
  constexpr int BUF = NBW * KS * 1024;  // bytes per weight buffer
  constexpr bool LDSOUT = ABL == 0;  // production store path: full 128-byte lines through a wave-private LDS tile
  constexpr bool HAS_TILE = pwx_has_tile(KS, NBW, ABL);  // wave-private [32 pixels][64 channels + pad] tiles behind the statistics
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem + 2 * BUF);  // [2][4 waves][NBW][2][32]
  unsigned char* tbuf = smem + 2 * BUF + 2 * 4 * NBW * 64 * 4 + (threadIdx.x >> 6) * (32 * kStagePitch);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // tile order: the nsplit channel parts of a pixel tile, and runs of pixel tiles, stay on one XCD (round-robin dispatch)
  const int mtiles = g.M >> 7;
  int mt = blockIdx.x / g.nsplit, part = blockIdx.x % g.nsplit;
  if ((mtiles & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    mt = (slot / g.nsplit) * 8 + xcd;
    part = slot % g.nsplit;
  }
  const int tpi = g.P >> 7;
  const int img = mt / tpi, tile = mt - img * tpi;
  const size_t m0 = (size_t)mt * 128;
  const int nper = g.N / g.nsplit, nbase = part * nper, nit = nper / (32 * NBW);
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(g.wf) + (size_t)(nbase >> 5) * KS * 1024;

  auto dma = [&](int it) {  // weight buffer `it` -> LDS ring slot it & 1: NBW * KS contiguous KB, 16 bytes per lane
    const unsigned char* src = wsrc + (size_t)it * BUF + lane * 16;
    unsigned char* dst = smem + (it & 1) * BUF;
#pragma unroll
    for (int c = 0; c < NBW * KS / 4; ++c) {
      const int chunk = wave + 4 * c;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + chunk * 1024), 16, 0, 0);
    }
  };
  dma(0);

  // ---- A phase: this wave's 32 pixel rows -> activated MFMA fragments in registers --------------------------------
  u32x4 afr[KS];
  {
    // staging rows: the output tile if this variant has one, else weight buffer 1 (free until the loop starts)
    unsigned char* stg = HAS_TILE ? tbuf : smem + BUF + wave * (32 * kStagePitch);
    const int srow = lane >> 3, kv = (lane & 7) * 8;  // load mapping: 8 lanes cover one row's 128 bytes
    const int koff1 = g.seg[0].ch, koff2 = g.seg[0].ch + g.seg[1].ch;
    constexpr int NCH = KS / 4;  // 64-channel chunks
    constexpr int PF = NCH < 3 ? NCH : 3;
    u32x4 raw[PF][4];
    f32x4 sc[PF][2], bi[PF][2];
    auto issue = [&](int c, int slot) {
      const int k0 = c * 64;
      const int s = (g.nseg > 1 && k0 >= koff1) + (g.nseg > 2 && k0 >= koff2);
      const GemmSeg sg = g.seg[s];
      const int cl = k0 - (s == 0 ? 0 : (s == 1 ? koff1 : koff2)) + kv;
      const T* base = reinterpret_cast<const T*>(sg.ptr) + (m0 + wave * 32 + srow) * sg.ch + cl;
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[slot][j] = *reinterpret_cast<const u32x4*>(base + (size_t)(8 * j) * sg.ch);
      const float* ps = sg.as + (size_t)img * sg.aff_ld + cl;
      const float* pb = sg.ab + (size_t)img * sg.aff_ld + cl;
      sc[slot][0] = *reinterpret_cast<const f32x4*>(ps);
      sc[slot][1] = *reinterpret_cast<const f32x4*>(ps + 4);
      bi[slot][0] = *reinterpret_cast<const f32x4*>(pb);
      bi[slot][1] = *reinterpret_cast<const f32x4*>(pb + 4);
    };
#pragma unroll
    for (int c = 0; c < PF; ++c) issue(c, c);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int slot = c % PF;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          o[q] = act_clamp01_pack<T, true>(raw[slot][j][q], sc[slot][q >> 1][(2 * q) & 3], sc[slot][q >> 1][(2 * q + 1) & 3],
                                           bi[slot][q >> 1][(2 * q) & 3], bi[slot][q >> 1][(2 * q + 1) & 3]);
        *reinterpret_cast<u32x4*>(stg + (srow + 8 * j) * kStagePitch + kv * 2) = o;
      }
      if (c + PF < NCH) issue(c + PF, slot);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private staging: this wave's own writes, then its reads
#pragma unroll
      for (int q = 0; q < 4; ++q) afr[4 * c + q] = *reinterpret_cast<const u32x4*>(stg + lr * kStagePitch + (16 * q + 8 * lh) * 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments read before the next chunk overwrites the rows
    }
  }

  // 0/1 selection matrices of the register transpose: B[k][j] = 1 iff pixel(k) == j, with the pixel order of the packed
  // accumulator words (k = 8 * half + e  <->  accumulator register e (+ 8 for the second MFMA) of that lane half)
  u32x4 id1, id2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t w1 = 0, w2 = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = 2 * q + t;
      const int px = (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (px == lr) w1 |= one_bits<T>() << (16 * t);
      if (px + 16 == lr) w2 |= one_bits<T>() << (16 * t);
    }
    id1[q] = w1;
    id2[q] = w2;
  }
  const uint32_t ones2 = one_bits<T>() | (one_bits<T>() << 16);

  T* outp = reinterpret_cast<T*>(g.out) + (m0 + wave * 32 + lr) * g.N + nbase + lh * 8;
  const int ntiles = tpi;
  auto flush_stats = [&](int it) {  // after a barrier: combine the four waves' partial sums of buffer `it` in wave order
    if (tid < NBW * 64) {
      const float* r = red + (size_t)(it & 1) * 4 * NBW * 64 + tid;
      const float t = ((r[0] + r[NBW * 64]) + r[2 * NBW * 64]) + r[3 * NBW * 64];
      const int j = tid >> 6, which = (tid >> 5) & 1, c = tid & 31;
      g.stats[((size_t)(img * ntiles + tile) * 2 + which) * g.N + nbase + (it * NBW + j) * 32 + c] = t;
    }
  };

  if constexpr (STAMP) t_a = __builtin_amdgcn_s_memtime();
  // buffer 0 has been in flight since the start; buffer 1's slot held the staging rows until here
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (nit > 1) dma(1);

  for (int it = 0; it < nit; ++it) {
    const unsigned char* wb = smem + (it & 1) * BUF + lane * 16;
    float* redw = red + (size_t)((it & 1) * 4 + wave) * NBW * 64 + lane;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        // the weight fragments run PD reads ahead of their MFMAs (left alone, hipcc issues two reads, waits for both and
        // multiplies twice: at two waves per SIMD the matrix pipe then waits on LDS latency in every pair)
        constexpr int PD = KS >= 32 ? 4 : 6;
        u32x4 wfrag[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) wfrag[s] = *reinterpret_cast<const u32x4*>(wb + (j * KS + s) * 1024);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma_u<T>(afr[s], wfrag[s], acc);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int s = 0; s < KS - PD; ++s) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, PD, 0);
      }
      // ---- epilogue of one 32 pixel x 32 channel block (lane = channel lr, registers = 16 pixels) ----
      u32x4 h0, h1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h0[q] = pack2<T>(acc[2 * q], acc[2 * q + 1]);
        h1[q] = pack2<T>(acc[8 + 2 * q], acc[8 + 2 * q + 1]);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1 = dot2<T>(h0[q], ones2, s1);
        s2 = dot2<T>(h0[q], h0[q], s2);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1 = dot2<T>(h1[q], ones2, s1);
        s2 = dot2<T>(h1[q], h1[q], s2);
      }
      {  // lanes 0-31 end up with the channel's sum, lanes 32-63 with its sum of squares (one half swap)
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
        redw[j * 64] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
      f32x16 d2;
#pragma unroll
      for (int r = 0; r < 16; ++r) d2[r] = 0.f;
      d2 = mfma_u<T>(h0, id1, d2);
      d2 = mfma_u<T>(h1, id2, d2);  // lane = pixel lr, register r = channel (r & 3) + 8 (r >> 2) + 4 lh, values exact
      uint32_t o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = pack2<T>(d2[2 * q], d2[2 * q + 1]);
#pragma unroll
      for (int q = 0; q < 8; q += 4) {  // (o[q], o[q+1] | o[q+2], o[q+3]): channels 0-3 | 8-11 (+ 4 lh) of a 16-channel half
        const auto a = __builtin_amdgcn_permlane32_swap(o[q], o[q + 2], false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(o[q + 1], o[q + 3], false, false);
        u32x4 v = {a[0], b[0], a[1], b[1]};  // lower lanes: channels 0-7, upper lanes: channels 8-15 of that half
        if constexpr (LDSOUT) {
          *reinterpret_cast<u32x4*>(tbuf + lr * kStagePitch + (((it * NBW + j) & 1) * 32 + lh * 8 + q * 4) * 2) = v;
        } else if constexpr (ABL == 6) {  // straight from registers: 32 rows x 32 bytes per instruction
          *reinterpret_cast<u32x4*>(outp + (it * NBW + j) * 32 + q * 4) = v;
        } else if constexpr (ABL == 1) {
          T* lin = reinterpret_cast<T*>(g.out) + ((m0 + wave * 32) * g.N) + (size_t)(((it * NBW + j) * 2 + (q >> 2)) * 64 + lane) * 8;
          *reinterpret_cast<u32x4*>(lin) = v;
        } else if constexpr (ABL >= 3) {  // same footprint, PPR 16-byte pieces per row and instruction (3: 64 B, 4: 128 B, 5: 256 B)
          constexpr int PPR = ABL == 3 ? 4 : (ABL == 4 ? 8 : 16);
          constexpr int IPG = PPR / 2;  // instructions per group of PPR / 4 blocks
          const int t = ((it * NBW + j) * 2 + (q >> 2));
          const int row = (t % IPG) * (64 / PPR) + lane / PPR, col = (t / IPG) * (PPR * 8) + (lane % PPR) * 8;
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(g.out) + (m0 + wave * 32 + row) * g.N + nbase + col) = v;
        } else {
          asm volatile("" ::"v"(v));
        }
      }
      if constexpr (LDSOUT) {
        if (((it * NBW + j) & 1) == 1) {  // 64 channels of the wave's 32 pixels are in the tile: 4 stores of 8 rows x 128 bytes
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own writes (the tile is wave-private)
          u32x4 rowv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) rowv[i] = *reinterpret_cast<const u32x4*>(tbuf + (8 * i + (lane >> 3)) * kStagePitch + (lane & 7) * 16);
          // the reads must have returned before another lane's next write lands on their rows: the compiler orders
          // memory operations per thread only, so pin the order here
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          T* orow = reinterpret_cast<T*>(g.out) + (m0 + wave * 32 + (lane >> 3)) * g.N + nbase + ((it * NBW + j) >> 1) * 64 + (lane & 7) * 8;
          if (g.nt) {  // uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(rowv[i], reinterpret_cast<u32x4*>(orow + (size_t)(8 * i) * g.N));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(orow + (size_t)(8 * i) * g.N) = rowv[i];
          }
        }
      }
    }
    if (it + 1 < nit) {  // hand over to the next buffer (kept at the END of the body: a uniform loop, nothing peeled)
      unsigned long long t0 = 0;
      if constexpr (STAMP) t0 = __builtin_amdgcn_s_memtime();
      // the DMA of the next buffer was issued before this buffer's output stores: leave those in flight.
      // (LDSOUT stores 4 instructions per pair of blocks, the direct path 2 per block: 2 * NBW either way, except that
      // with one block per buffer the LDS path stores only behind every second buffer.)
      if constexpr (LDSOUT && NBW == 1) {
        if (it & 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NBW) : "memory");
      }
      if constexpr (STAMP) t_wait += __builtin_amdgcn_s_memtime() - t0;
      __builtin_amdgcn_s_barrier();
      flush_stats(it);
      if (it + 2 < nit) dma(it + 2);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  flush_stats(nit - 1);
  if constexpr (STAMP) {
    if (g.stamps && lane == 0) {
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      const size_t wv = (size_t)blockIdx.x * 4 + wave;
      g.stamps[3 * wv] = t_a - t_start;
      g.stamps[3 * wv + 1] = t_end - t_a;
      g.stamps[3 * wv + 2] = t_wait;
    }
  }
}

// Weight pack: fp32 [N][K] (OIHW of a 1x1 conv) -> T in MFMA B-fragment order, times `scale`:
//   dst[((nblk * KS + s) * 64 + lane) * 8 + e] = scale * W[32 nblk + (lane & 31)][16 s + 8 (lane >> 5) + e]
template <typename T>
__global__ void pack_expand_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int K, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * K) return;
  dst[pw_expand_pack_index((int)(i / K), (int)(i % K), K)] = (T)(src[i] * scale);
}
hipError_t launch_pack_expand(int dtype, const float* src, void* dst, int N, int K, float scale, hipStream_t s) {
  if (N % 32 || K % 16) return hipErrorInvalidValue;
  const dim3 grid((unsigned)(((long long)N * K + 255) / 256));
  if (dtype == 1) hipLaunchKernelGGL(pack_expand_kernel<half_t>, grid, dim3(256), 0, s, src, reinterpret_cast<half_t*>(dst), N, K, scale);
  else if (dtype == 2) hipLaunchKernelGGL(pack_expand_kernel<bf16_t>, grid, dim3(256), 0, s, src, reinterpret_cast<bf16_t*>(dst), N, K, scale);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

static int g_use_pwx = 1;
void pw_expand_enable(int v) { g_use_pwx = v; }

// 32-channel blocks per LDS buffer for a given K (0 = K not served); g_pwx_nbw overrides where the variant exists.
// One block per buffer everywhere: the smaller LDS footprint (three workgroups per CU up to K = 256) is worth more than
// the saved barriers (profiles/r03/pwx_nbw.txt)
static int g_pwx_nbw = 0;
static int nbw_for(int K) {
  const int f = g_pwx_nbw;
  switch (K) {
    case 128: return (f == 1 || f == 2 || f == 4) ? f : 1;
    case 192: return (f == 1 || f == 2) ? f : 1;
    case 256: return (f == 1 || f == 2) ? f : 1;
    case 384: return 1;
    case 512: return 1;
  }
  return 0;
}
static int nsplit_for(int M, int N, int nbw) {
  // small grids: split the channels of a pixel tile over workgroups until the chip's 512 slots (2 per CU) are filled
  int ns = 1;
  while ((long)(M / 128) * ns < 512 && (N / (32 * nbw)) % (2 * ns) == 0 && N / (32 * nbw * 2 * ns) >= 2 && (N / (2 * ns)) % 64 == 0) ns *= 2;
  return ns;
}
bool pw_expand_serves_k(int K) { return K == 128 || K == 192 || K == 256 || K == 384 || K == 512; }
bool pw_expand_supported(int dtype, const GemmSeg* seg, int nseg, int M, int N, int K, int P) {
  if (!g_use_pwx || (dtype != 1 && dtype != 2) || nseg < 1 || nseg > 3 || P % 128 || M % P) return false;
  const int nbw = nbw_for(K);
  if (!nbw || N % (32 * nbw) || N % 64) return false;  // pairs of 32-channel blocks leave as 128-byte lines
  int k = 0;
  for (int i = 0; i < nseg; ++i) {
    if (seg[i].ch % 64 || seg[i].act != ACT_RELU6_S6 || !seg[i].as || !seg[i].ab) return false;
    k += seg[i].ch;
  }
  return k == K;
}

static int g_pwx_ablate = 0, g_pwx_stamp = 0;
static unsigned long long* g_pwx_stamps = nullptr;
static size_t g_pwx_stamp_waves = 0;
constexpr size_t kPwxStampWaves = 1u << 18;
void pw_expand_debug(int ablate, int stamp, int nbw) {
  if (ablate >= 0) g_pwx_ablate = ablate;
  if (stamp >= 0) g_pwx_stamp = stamp;
  if (nbw >= 0) g_pwx_nbw = nbw;
}
hipError_t pw_expand_stamp_fetch(double* out4) {  // mean s_memtime ticks per wave of the last stamped launch: {A phase, channel loop, of which in the vmcnt wait at the top of a buffer}, waves
  if (!g_pwx_stamps || !g_pwx_stamp_waves) return hipErrorInvalidValue;
  std::vector<unsigned long long> h(g_pwx_stamp_waves * 3);
  if (hipError_t e = hipMemcpy(h.data(), g_pwx_stamps, h.size() * 8, hipMemcpyDeviceToHost); e != hipSuccess) return e;
  double a = 0, b = 0, w = 0;
  for (size_t i = 0; i < g_pwx_stamp_waves; ++i) { a += (double)h[3 * i]; b += (double)h[3 * i + 1]; w += (double)h[3 * i + 2]; }
  out4[0] = a / (double)g_pwx_stamp_waves; out4[1] = b / (double)g_pwx_stamp_waves; out4[2] = w / (double)g_pwx_stamp_waves;
  out4[3] = (double)g_pwx_stamp_waves;
  return hipSuccess;
}

template <typename T, int KS, int NBW, int ABL = 0, bool STAMP = false>
static hipError_t launch_one(ExpandArgs a, hipStream_t s) {
  constexpr int lds = pwx_lds_bytes(KS, NBW, ABL);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = ensure_max_lds(reinterpret_cast<const void*>(&pw_expand_kernel<T, KS, NBW, ABL, STAMP>), lds, attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL((pw_expand_kernel<T, KS, NBW, ABL, STAMP>), dim3((unsigned)((a.M / 128) * a.nsplit)), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <typename T, int KS, int NBW>
static hipError_t launch_cfg(ExpandArgs a, hipStream_t s) {
  a.nsplit = nsplit_for(a.M, a.N, NBW);
  static const std::string name = std::string("pw_expand_kernel<") + TypeName<T>::value + ", " + std::to_string(KS) + ", " + std::to_string(NBW) + ">";
  note_kernel(name.c_str());
  if constexpr (std::is_same<T, half_t>::value && (KS == 12 || KS == 24)) {  // diagnostics for two representative shapes only
    if (g_pwx_stamp) {
      const size_t waves = (size_t)(a.M / 128) * a.nsplit * 4;
      if (waves > kPwxStampWaves) return hipErrorInvalidValue;
      if (!g_pwx_stamps && hipMalloc(reinterpret_cast<void**>(&g_pwx_stamps), kPwxStampWaves * 24) != hipSuccess) return hipErrorOutOfMemory;
      g_pwx_stamp_waves = waves;
      a.stamps = g_pwx_stamps;
      if (g_pwx_ablate == 1) return launch_one<T, KS, NBW, 1, true>(a, s);
      if (g_pwx_ablate == 2) return launch_one<T, KS, NBW, 2, true>(a, s);
      return launch_one<T, KS, NBW, 0, true>(a, s);
    }
    if (g_pwx_ablate == 1) return launch_one<T, KS, NBW, 1>(a, s);
    if (g_pwx_ablate == 2) return launch_one<T, KS, NBW, 2>(a, s);
    if (g_pwx_ablate == 3) return launch_one<T, KS, NBW, 3>(a, s);
    if (g_pwx_ablate == 4) return launch_one<T, KS, NBW, 4>(a, s);
    if (g_pwx_ablate == 5) return launch_one<T, KS, NBW, 5>(a, s);
  }
  if (g_pwx_ablate == 6) return launch_one<T, KS, NBW, 6>(a, s);  // a correct variant: available for every shape
  if (g_pwx_ablate == 7) return launch_one<T, KS, NBW>(a, s);
  // K = 512: the LDS tile would leave room for one workgroup per CU only (64 KB of weight buffers): registers -> HBM there
  if constexpr (KS == 32) return launch_one<T, KS, NBW, 6>(a, s);
  return launch_one<T, KS, NBW>(a, s);
}
template <typename T>
static hipError_t launch_t(const ExpandArgs& a, hipStream_t s) {
  const int nbw = nbw_for(a.K);
  switch (a.K) {
    case 128: return nbw == 4 ? launch_cfg<T, 8, 4>(a, s) : (nbw == 2 ? launch_cfg<T, 8, 2>(a, s) : launch_cfg<T, 8, 1>(a, s));
    case 192: return nbw == 2 ? launch_cfg<T, 12, 2>(a, s) : launch_cfg<T, 12, 1>(a, s);
    case 256: return nbw == 2 ? launch_cfg<T, 16, 2>(a, s) : launch_cfg<T, 16, 1>(a, s);
    case 384: return launch_cfg<T, 24, 1>(a, s);
    case 512: return launch_cfg<T, 32, 1>(a, s);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_pw_expand(int dtype, const ExpandArgs& a, hipStream_t s) {
  // host-side shape contract (an out-of-contract shape would index out of bounds on the device)
  if (!a.wf || !a.out || !a.stats || !pw_expand_supported(dtype, a.seg, a.nseg, a.M, a.N, a.K, a.P)) return hipErrorInvalidValue;
  return dtype == 1 ? launch_t<half_t>(a, s) : launch_t<bf16_t>(a, s);
}

}  // namespace llie
