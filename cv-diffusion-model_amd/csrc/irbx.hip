// Front half of an InvertedResidualBlock in its "recompute" form for gfx950 (2-byte compute types):
//
//   h2 = dw3x3( relu6( aff2( W1 . relu6( aff1(x) ) ) ) )      aff1 = GroupNorm-1, aff2 = GroupNorm-2 + FiLM
//   (efficient_unet.py:207-220: norm1, ReLU6, expand, norm2, FiLM, ReLU6, depthwise)
//
// The 4x-expanded tensor h1 = W1 . a is the largest object of the network.  The unfused path writes it (pw_gemm)
// and reads it back (dwconv3x3); here it never reaches HBM (SURVEY.md 8d "recompute variant", 3Cin + 2Chid + Cout
// elements per pixel instead of 2Cin + 4Chid + Cout):
//
//   expand_stats_kernel  reads x once and produces only h1's per-channel (sum, sum of squares) slab for GroupNorm-2:
//                        lanes load their MFMA operand slices of x straight from HBM (16 B per lane), the 32x32
//                        accumulators are squared / summed in registers and never stored;
//   expand_dw_kernel     a workgroup owns an 8 x 16 pixel tile: it rebuilds h1 on the 10 x 18 halo tile with MFMA
//                        (weights = A operand, pixels = B operand, so a lane ends up with runs of channels of ONE
//                        pixel), applies aff2 + ReLU6 (+ zero padding) to the accumulators, parks the tile in LDS as
//                        [pixel][64 channels] and runs the depthwise 3x3 from there, 64 hidden channels at a time;
//                        the x tile is loaded once per tile (next tile prefetched) and reused by every channel chunk.
//
// The GroupNorm partial sums are fixed per-tile slab entries; the SE pool sums go into per-image 64-bit fixed-point
// totals with integer atomics (integer adds commute).  No float atomics, so results stay bitwise independent of the
// batch and of the schedule.
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace llie {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&a), *reinterpret_cast<const bf16x2_t*>(&b), c, false);
}

// ---------------------------------------------------------------------------------------------
// Both kernels run out of VALU issue slots first (a wave64 instruction holds its SIMD for 4 cycles), so the two
// activations are written for instruction count.  ReLU6 is carried as clamp01(z / 6): the clamp is the FMA's free output
// modifier, and the factor 6 is pushed through the linear operators that follow -- the expand GEMM (h1 = 6 W1 a', so
// GroupNorm-2 sees sums scaled by 6 / 36 and its affine is applied to acc' = acc / 6 with shift / 6) and the depthwise
// conv (weights staged as 6 w).  In real arithmetic nothing changes; in T the rounding points move by one operation.
// a' = clamp01(a * s + b) = relu6(norm1(a)) / 6 on one 16-byte operand slice (8 channels); sc / sh point at the slice's
// 8 scale / shift values (already divided by 6) in an LDS table (16-byte aligned), read at use
template <typename T>
__device__ __forceinline__ typename Elem<T>::vec_t activate8(typename Elem<T>::vec_t v, const float* sc, const float* sh) {
  const u32x4 x = reinterpret_cast<const u32x4&>(v);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
  u32x4 o;
  o[0] = act_clamp01_pack<T>(x[0], s0[0], s0[1], b0[0], b0[1]);
  o[1] = act_clamp01_pack<T>(x[1], s0[2], s0[3], b0[2], b0[3]);
  o[2] = act_clamp01_pack<T>(x[2], s1[0], s1[1], b1[0], b1[1]);
  o[3] = act_clamp01_pack<T>(x[3], s1[2], s1[3], b1[2], b1[3]);
  return reinterpret_cast<const typename Elem<T>::vec_t&>(o);
}

// ---------------------------------------------------------------------------------------------
// (1) statistics of h1 = W1 . relu6(aff1(x)) without storing it.
//   grid (P / RP, 1, B): a workgroup walks RP pixels of one image in steps of 128.  Each step's x rows are activated
//   ONCE, cooperatively, into a double-buffered LDS tile (one barrier per step); wave w owns the NBW = Chid / 128
//   32-channel blocks [w * NBW, (w + 1) * NBW) -- their weight slices stay in registers -- and multiplies them with all
//   four 32-pixel blocks of the tile.  MFMA roles: A = pixels (rows), B = weights (columns): a lane holds 16 pixels of
//   ONE channel, so the per-channel sums are plain register adds and no accumulator is ever stored.
template <typename T, int KS, int NBW>
__global__ void __launch_bounds__(256, 2) expand_stats_kernel(const IrbxArgs a, const int RP) {
  constexpr int K = 16 * KS, XP = (K + 8) * 2;  // LDS pixel pitch in bytes: conflict-free ds_read_b128
  typedef typename Elem<T>::vec_t vec_t;
  __shared__ __align__(16) unsigned char sA[2][128 * XP];
  __shared__ __align__(16) float aff1[2][K];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, tile = blockIdx.x;
  const int P = a.H * a.W;
  const T* x0 = reinterpret_cast<const T*>(a.x0) + (size_t)b * P * a.c0;
  const T* x1 = a.x1 ? reinterpret_cast<const T*>(a.x1) + (size_t)b * P * a.c1 : nullptr;
  const T* w1 = reinterpret_cast<const T*>(a.w1);

  vec_t wf[NBW][KS];
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[j][s] = ld_vec<T>(w1 + (size_t)((wave * NBW + j) * 32 + n) * K + 16 * s + 8 * h);
  for (int i = tid; i < K; i += 256) {
    aff1[0][i] = a.as1[(size_t)b * K + i];   // already / 6 (GnFinalizeArgs::post_scale)
    aff1[1][i] = a.ab1[(size_t)b * K + i];
  }
  float s1[NBW], s2[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) s1[j] = s2[j] = 0.f;

  // cooperative load: vector v = tid + j*256 of a step -> pixel v / (2 KS), channel vector v % (2 KS)
  const int nsteps = RP / 128;
  const size_t p_first = (size_t)tile * RP;
  vec_t raw[KS];
  auto load = [&](int step) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int v = tid + j * 256;
      const size_t pix = p_first + (size_t)step * 128 + v / (2 * KS);
      const int k = (v % (2 * KS)) * 8;
      raw[j] = k < a.c0 ? ld_vec<T>(x0 + pix * a.c0 + k) : ld_vec<T>(x1 + pix * a.c1 + (k - a.c0));
    }
  };
  load(0);
  wg_barrier();  // aff1 staged
  for (int step = 0; step < nsteps; ++step) {
    unsigned char* buf = sA[step & 1];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int v = tid + j * 256;
      const int k = (v % (2 * KS)) * 8;
      *reinterpret_cast<vec_t*>(buf + (v / (2 * KS)) * XP + k * 2) = activate8<T>(raw[j], &aff1[0][k], &aff1[1][k]);
    }
    if (step + 1 < nsteps) load(step + 1);
    wg_barrier();  // tile `step` complete; the other buffer (read during step - 1) is free for step + 1
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
      vec_t af[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const vec_t*>(buf + (pb * 32 + n) * XP + (16 * s + 8 * h) * 2);
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma16<T>(af[s], wf[j][s], acc);
        // packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two values per instruction) -- this reduction, not the MFMAs, is what
        // the wave spends its issue slots on
        f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 v = {acc[r], acc[r + 1]};
          t1 += v;
          t2 = __builtin_elementwise_fma(v, v, t2);
        }
        s1[j] += t1[0] + t1[1];
        s2[j] += t2[0] + t2[1];
      }
    }
  }
  // lane halves hold different pixel rows of the same channel; every wave owns its channels outright
  const int ntiles = P / RP;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    s1[j] += __shfl_xor(s1[j], 32, 64);
    s2[j] += __shfl_xor(s2[j], 32, 64);
    if (h == 0) {
      const int c = (wave * NBW + j) * 32 + n;
      a.stats[((size_t)(b * ntiles + tile) * 2 + 0) * a.Chid + c] = 6.f * s1[j];   // h1 = 6 W1 a'
      a.stats[((size_t)(b * ntiles + tile) * 2 + 1) * a.Chid + c] = 36.f * s2[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (2) fused expand + norm2 / FiLM / ReLU6 + depthwise 3x3 + SE pool partials.
constexpr int kXT_H = 8, kXT_W = 16;                 // output tile
constexpr int kXH_W = kXT_W + 2, kXH_H = kXT_H + 2;  // halo tile 10 x 18 = 180 pixels -> 6 blocks of 32
constexpr int kXNPX = kXH_W * kXH_H;
constexpr int kXNPB = 6;

// LDS image of the activated h1 tile: [pixel q][64 channels + 16 B pad].  Both users address it with lane = pixel and a
// fixed 16-byte channel slot (the expand epilogue's ds_write_b128, the depthwise MFMAs' ds_read_b128): the 144-byte
// pitch (36 dwords) spreads consecutive pixels over distinct bank groups for either instruction's lane grouping.
constexpr int SHP = 144;
constexpr int kXStamps = 9;

// STAMP = diagnostic build (llie_tune("irbx_stamp", 1)): s_memtime around the phases, summed per wave into a.dbg
// ([workgroup][wave][kXStamps] cycles: 0 wait for the prefetched / loaded x tile (vmcnt), 1 activate + ds_write of the x tile,
// 2 next tile's loads issued + halo validity, 3 tile-top barrier, 4 pool flush behind it, 5 chunk-top barrier + flush
// (chunks after the first), 6 expand MFMAs + epilogue + ds_write, 7 barrier behind them, 8 depthwise phase incl. the h2
// stores and the pool partial); never used in production.
// ABL (diagnostic builds only): timing ablations -- 1 no h2 stores, 2 no pool sums, 4 h2 stores confined to L2, 8 no depthwise MFMAs,
// 16 h2 stores as contiguous kilobytes (wrong layout, same bytes), 32 no workgroup barriers inside the tile loop (what a
// barrier-free structure could gain at most; results wrong).
// DWV = form of the depthwise phase: 0 = one tap per 32x32x16 MFMA (k = 16 channels of a diagonal weight matrix),
// 1 = two taps per 16x16x32 MFMA (k = 2 taps x 16 channels): the same ds_read_b128 data operand per MFMA, half the
// matrix-pipe time per MFMA -> 640 instead of 1 152 pipe cycles per 64-channel chunk and wave.
// VAR (bit mask): 1 = the 64- / 96-channel variants request the next tile's x under the last chunk's depthwise phase (sX is
// free from that chunk's expand phase on; the registers are live for one phase only), 2 = h2 leaves with non-temporal stores.
constexpr int kXDefaultVar = 0;
template <typename T, int KS, bool DBUF, bool STAMP = false, int ABL = 0, int DWV = 1, int VAR = kXDefaultVar>
__global__ void __launch_bounds__(256, (KS == 2 && !DBUF) ? 3 : 2) expand_dw_kernel(const IrbxArgs a, const int tiles_per_wg, const int chunks_per_wg) {
  constexpr int K = 16 * KS;
  constexpr int XP = (K + 8) * 2;                        // sX pixel pitch in bytes (80 / 144 / 208 / 272: conflict-free ds_read_b128)
  // x halo tile staging: thread -> (pixel xq0, 16-byte channel vector xkv); pass j covers pixel xq0 + j * QSTEP.  The channel
  // vector -- hence the K segment, its base pointer and the norm1 table slice -- is the same in every pass and every tile.
  constexpr int QSTEP = 256 / (2 * KS);                  // 64 / 32 / 21 pixels per pass
  constexpr int XTHR = QSTEP * 2 * KS;                   // threads that take part (252 of 256 at KS = 6)
  constexpr int XPT = (kXNPX + QSTEP - 1) / QSTEP;       // passes: 3 / 6 / 9
  constexpr int SH_BYTES = kXNPB * 32 * SHP;
  constexpr bool PREF = KS <= 2;        // next tile's x prefetched into registers at the top of the tile
  constexpr bool LATE = !PREF && (VAR & 1);  // ... or under the tile's last depthwise phase
  constexpr bool NTST = (VAR & 2) != 0;
  typedef typename Elem<T>::vec_t vec_t;
  extern __shared__ __align__(16) unsigned char smem[];
  // [sH: (DBUF ? 2 : 1) x 192 x 128 B][sX: 192 x XP][wds: 9 x Chid T][aff2: 2 x Chid fp32][aff1: 2 x K fp32][red: 2 x 4 x 64 fp32]
  unsigned char* sH = smem;
  unsigned char* sX = smem + (DBUF ? 2 : 1) * SH_BYTES;
  T* wds = reinterpret_cast<T*>(sX + kXNPB * 32 * XP);
  float* aff2 = reinterpret_cast<float*>(wds + 10 * a.Chid);
  float* aff1 = aff2 + 2 * a.Chid;
  float* red = aff1 + 2 * K;
  // fixed-point pool totals of this workgroup's tiles, [chunk][channel 64]: in LDS rather than in 2 KS registers of every
  // thread -- this kernel sits exactly at an occupancy step, and a spilled register costs more than its scratch access:
  // every reload is a vector-memory operation whose wait (vmcnt is in order) also waits for the h2 stores in flight
  // (KS = 6 keeps them in registers: its 79 KB of LDS are two workgroups per CU only as long as nothing is added)
  constexpr bool PACC_LDS = KS <= 4;
  long long* pacc_lds = reinterpret_cast<long long*>(red + 2 * 256);
  long long pacc_reg[PACC_LDS ? 1 : KS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int chb = wave & 1, pxg = wave >> 1;
  const int b = blockIdx.z;
  const int tiles_x = a.W / kXT_W;
  const int P = a.H * a.W;
  const T* x0 = reinterpret_cast<const T*>(a.x0) + (size_t)b * P * a.c0;
  const T* x1 = a.x1 ? reinterpret_cast<const T*>(a.x1) + (size_t)b * P * a.c1 : nullptr;
  const T* w1 = reinterpret_cast<const T*>(a.w1);
  T* out = reinterpret_cast<T*>(a.out) + (size_t)b * P * a.Chid;
  const int chunk0 = blockIdx.y * chunks_per_wg;
  const int nchunks_all = a.Chid / 64;
  const int chunk1 = chunk0 + chunks_per_wg < nchunks_all ? chunk0 + chunks_per_wg : nchunks_all;

  // ---- per-workgroup constants: depthwise weights (packed T), affine tables of this image
  if constexpr (DWV == 1) {  // [tap pair][channel][2]: one dword = this channel's weights of taps 2 p and 2 p + 1 (tap 9 = 0)
    for (int i = tid; i < 10 * a.Chid; i += 256) {
      const int t = i & 1, c = (i >> 1) % a.Chid, tap = 2 * ((i >> 1) / a.Chid) + t;
      wds[i] = tap < 9 ? (T)(6.f * a.wd[tap * a.Chid + c]) : (T)0.f;
    }
  } else {
    for (int i = tid; i < 9 * a.Chid; i += 256) wds[i] = (T)(6.f * a.wd[i]);  // the tile in LDS holds relu6(.) / 6
  }
  for (int i = tid; i < a.Chid; i += 256) {
    aff2[i] = a.as2[(size_t)b * a.Chid + i];                      // applied to acc' = acc / 6: scale unchanged,
    aff2[a.Chid + i] = a.ab2[(size_t)b * a.Chid + i] * kSixth;    // shift / 6, result clamped to [0, 1]
  }
  for (int i = tid; i < K; i += 256) {
    aff1[i] = a.as1[(size_t)b * K + i];      // already / 6 (GnFinalizeArgs::post_scale)
    aff1[K + i] = a.ab1[(size_t)b * K + i];
  }
  // pixels 180..191 of the last MFMA block do not exist: their operand rows stay zero
  for (int i = tid; i < (kXNPB * 32 - kXNPX) * (XP / 16); i += 256)
    *reinterpret_cast<u32x4*>(sX + kXNPX * XP + i * 16) = u32x4{0u, 0u, 0u, 0u};

  // depthwise phase roles: wave = (channel block chb, output rows 4 pxg .. 4 pxg + 3); lane = pixel n of a 2-row block.
  // dmask: where this lane's weight sits in the diagonal operand -- channel n of the block is element n & 7 of k-slice
  // (n >> 3) = 2 s + h, i.e. one 16-bit half of one dword for one (s, h) and nothing elsewhere
  uint32_t dmask[2][4];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int d = 0; d < 4; ++d)
      dmask[s2][d] = ((n >> 3) == 2 * s2 + h && ((n & 7) >> 1) == d) ? ((n & 1) ? 0xFFFF0000u : 0x0000FFFFu) : 0u;
  int dq[2], drow[2];  // halo pixel of tap (0, 0) and output row of this lane in its two blocks
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    drow[blk] = 2 * (2 * pxg + blk) + (n >> 4);
    dq[blk] = drow[blk] * kXH_W + (n & 15);
  }

  const int ntiles_img = tiles_x * (a.H / kXT_H);
  // tiles_per_wg > 0: fixed runs; <= 0: the image's tiles split evenly over the gridDim.x workgroups (knob "irbx_grid")
  const int tile_first = tiles_per_wg > 0 ? blockIdx.x * tiles_per_wg : (int)(blockIdx.x * (unsigned)ntiles_img / gridDim.x);
  const int tile_end = tiles_per_wg > 0 ? tile_first + tiles_per_wg : (int)((blockIdx.x + 1) * (unsigned)ntiles_img / gridDim.x);
  const int tile_last = tile_end < ntiles_img ? tile_end : ntiles_img;

  // x halo tile: per-thread constants of the staging -- nothing in the tile loop divides
  const int xkv = tid % (2 * KS), xq0 = tid / (2 * KS), xk8 = xkv * 8;
  const bool xthr = XTHR == 256 || tid < XTHR;
  const bool xseg1 = xk8 >= a.c0;
  const T* xb = xseg1 ? x1 + (xk8 - a.c0) : x0 + xk8;   // + pixel * xc
  const int xc = xseg1 ? a.c1 : a.c0;
  const bool xlast = xthr && (XPT - 1) * QSTEP + xq0 < kXNPX;   // the last pass is partial
  int xrel[XPT];                                          // pixel offset of pass j relative to the tile's first output pixel
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int q = xq0 + j * QSTEP;
    xrel[j] = (q / kXH_W - 1) * a.W + (q % kXH_W - 1);
  }
  unsigned char* sxw = sX + xq0 * XP + xk8 * 2;
  vec_t raw[XPT];
  // (ty, tx) = tile coordinates; interior tiles need no bounds checks
  auto load_tile = [&](int ty, int tx) {
    const int pix0 = ty * kXT_H * a.W + tx * kXT_W;
    const bool border = ty == 0 || tx == 0 || ty == a.H / kXT_H - 1 || tx == tiles_x - 1;
    if (!border) {
#pragma unroll
      for (int j = 0; j < XPT; ++j) {
        if (j < XPT - 1 ? xthr : xlast) raw[j] = ld_vec<T>(xb + (size_t)(pix0 + xrel[j]) * xc);
        else raw[j] = vec_t{};
      }
    } else {
#pragma unroll
      for (int j = 0; j < XPT; ++j) {
        const int q = xq0 + j * QSTEP;
        const int gy = ty * kXT_H - 1 + q / kXH_W, gx = tx * kXT_W - 1 + q % kXH_W;
        // outside the image: h1 is forced to zero there (ok[] below), the value is irrelevant
        if ((j < XPT - 1 ? xthr : xlast) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) raw[j] = ld_vec<T>(xb + (size_t)(pix0 + xrel[j]) * xc);
        else raw[j] = vec_t{};
      }
    }
  };
  int ty = tile_first / tiles_x, tx = tile_first % tiles_x;   // the only division: once per workgroup
  if ((PREF || LATE) && tile_first < tile_last) load_tile(ty, tx);
  // weight slices (A operand) of the chunk about to run.  They are always fetched one depthwise phase ahead and BEFORE
  // that phase's stores: the wait in front of the MFMAs then leaves the (younger) stores in flight instead of draining them.
  vec_t wf[KS];
  auto load_wf = [&](int chunk) {
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[s] = ld_vec<T>(w1 + (size_t)(chunk * 64 + chb * 32 + n) * K + 16 * s + 8 * h);
  };
  load_wf(chunk0);
  wg_barrier();  // constants staged

  unsigned long long tk[kXStamps] = {}, t_prev = 0;
  auto stamp = [&](int slot) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      if (slot >= 0) tk[slot] += now - t_prev;
      t_prev = now;
    }
  };
  auto tile_barrier = [&]() {  // the barriers inside the tile loop (ABL 32 removes them: timing only)
    if constexpr (!(ABL & 32)) wg_barrier();
  };
  stamp(-1);
  const bool has_pool = a.pool != nullptr || a.pool_tot != nullptr;
  int par = 0;  // sH / red buffer parity (DBUF)
  int pend_tile = -1, pend_chunk = 0, pend_par = 0;  // pool partial waiting for its cross-wave sum
  if constexpr (PACC_LDS) {
    if (tid < 64) {  // threads 0..63 own channel tid of every chunk (Chid / 64 = KS chunks at most); only they touch it
#pragma unroll
      for (int q = 0; q < KS; ++q) pacc_lds[q * 64 + tid] = 0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < KS; ++q) pacc_reg[q] = 0;
  }
  auto flush_pool = [&]() {  // after a barrier that follows the depthwise phase which wrote red[pend_par]
    if (pend_tile >= 0 && tid < 64) {
      const float* r = red + pend_par * 256;  // wave (chb, pxg) = chb + 2 pxg left its 32 channel sums at [wave * 64 + channel]
      const int cbb = tid >> 5, ci = tid & 31;
      const float t = r[cbb * 64 + ci] + r[(cbb + 2) * 64 + ci];
      if (a.pool_tot) {  // fixed-point, summed over this workgroup's tiles in registers: one global atomic per chunk at the end
        const long long v = __float2ll_rn(t * kPoolFixScale);
        if constexpr (PACC_LDS) {
          pacc_lds[(pend_chunk - chunk0) * 64 + tid] += v;
        } else {
#pragma unroll
          for (int q = 0; q < KS; ++q) pacc_reg[q] += (pend_chunk - chunk0 == q) ? v : 0ll;
        }
      } else {
        a.pool[((size_t)b * ntiles_img + pend_tile) * a.Chid + pend_chunk * 64 + tid] = t;
      }
    }
    pend_tile = -1;
  };

  for (int tile = tile_first; tile < tile_last; ++tile) {
    const int y0 = ty * kXT_H, x0p = tx * kXT_W;
    int tyn = ty, txn = tx + 1;  // the next tile of the run
    if (txn == tiles_x) { txn = 0; ++tyn; }
    // ---- activate this tile's x (norm1 + ReLU6) into sX, prefetch the next tile.  Every wave is past the last
    // MFMA phase of the previous tile here (the barrier that follows it), so sX is free.
    if (!PREF && !LATE) load_tile(ty, tx);
    // Every vector-memory operation so far has to be complete here anyway (raw[] below is older than all of them), but the
    // compiler's wait sits inside the predicated block below; said unconditionally, the chunk loop is entered with nothing
    // pending, and the wait for the prefetched weight slices at its head becomes vmcnt(4 + ...) -- the depthwise phase's four
    // h2 stores stay in flight -- instead of the vmcnt(0) that the merge with this path forced.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    stamp(0);
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      if (j < XPT - 1 ? xthr : xlast)
        *reinterpret_cast<vec_t*>(sxw + j * QSTEP * XP) = activate8<T>(raw[j], aff1 + xk8, aff1 + K + xk8);
    }
    stamp(1);
    if (PREF && tile + 1 < tile_last) load_tile(tyn, txn);
    // validity of this lane's three halo pixels (zero padding of the depthwise input): border tiles only
    const bool border = ty == 0 || tx == 0 || ty == a.H / kXT_H - 1 || tx == tiles_x - 1;
    bool ok[3] = {true, true, true};
    if (border) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = (pxg * 3 + i) * 32 + n;
        const int gy = y0 - 1 + q / kXH_W, gx = x0p - 1 + q % kXH_W;
        ok[i] = q < kXNPX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      }
    }
    stamp(2);
    tile_barrier();
    stamp(3);
    if (!DBUF) flush_pool();
    stamp(4);

    // (Unrolling this loop lets hipcc count the memory operations between a prefetch and its use -- vmcnt(5) instead of
    // vmcnt(2) for the weight slices -- but costs 11 spilled registers, and every spill reload waits vmcnt(0), i.e. for the
    // h2 stores in flight: 35.7 vs 34.7 ms per step.  The run-time loop stays.)
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
      unsigned char* buf = sH + (DBUF ? par * SH_BYTES : 0);
      if (!DBUF && chunk > chunk0) {
        tile_barrier();  // previous depthwise phase done with sH
        flush_pool();
        stamp(5);
      }
      // ---- MFMA: h1^T block (32 channels x 32 pixels) x 3 pixel blocks
      const int ch0 = chunk * 64 + chb * 32;
      // aff2 of this lane's 16 accumulator channels: ch0 + 8g + 4h + e
      f32x4 sc2[4], sh2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sc2[g] = *reinterpret_cast<const f32x4*>(aff2 + ch0 + 8 * g + 4 * h);
        sh2[g] = *reinterpret_cast<const f32x4*>(aff2 + a.Chid + ch0 + 8 * g + 4 * h);
      }
      constexpr int NACC = KS <= 2 ? 3 : 1;  // accumulators in flight: all three pixel blocks, or one at a time (registers)
      f32x16 accs[NACC];
      if constexpr (NACC == 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int q = (pxg * 3 + i) * 32 + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[i][r] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s)
            accs[i] = mfma16<T>(wf[s], *reinterpret_cast<const vec_t*>(sX + q * XP + (16 * s + 8 * h) * 2), accs[i]);
        }
        load_wf(chunk + 1 < chunk1 ? chunk + 1 : chunk0);  // next chunk (or the next tile's first): see above
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = (pxg * 3 + i) * 32 + n;
        if constexpr (NACC == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[0][r] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s)
            accs[0] = mfma16<T>(wf[s], *reinterpret_cast<const vec_t*>(sX + q * XP + (16 * s + 8 * h) * 2), accs[0]);
          if (i == 2) load_wf(chunk + 1 < chunk1 ? chunk + 1 : chunk0);
        }
        const f32x16 acc = accs[NACC == 3 ? i : 0];
        // epilogue: aff2 + ReLU6, zero outside the image (the conv's padding), pack, exchange lane halves so
        // that each lane owns 8 consecutive channels, two ds_write_b128
        uint32_t pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            pk[g][j] = affine_clamp01_pack<T>(acc[4 * g + 2 * j], acc[4 * g + 2 * j + 1], sc2[g][2 * j], sc2[g][2 * j + 1],
                                              sh2[g][2 * j], sh2[g][2 * j + 1]);
        if (border) {  // tiles on the image border: zero padding of the depthwise input (uniform branch, most tiles skip it)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 2; ++j) pk[g][j] = ok[i] ? pk[g][j] : 0u;
        }
        u32x4 lo, hi2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x2 r02 = __builtin_amdgcn_permlane32_swap(pk[0][j], pk[2][j], false, false);
          const u32x2 r13 = __builtin_amdgcn_permlane32_swap(pk[1][j], pk[3][j], false, false);
          lo[j] = r02[0]; lo[2 + j] = r02[1];    // h=0: channels 0..7 of the block; h=1: channels 16..23
          hi2[j] = r13[0]; hi2[2 + j] = r13[1];  // h=0: channels 8..15;             h=1: channels 24..31
        }
        *reinterpret_cast<u32x4*>(buf + q * SHP + (chb * 4 + 2 * h) * 16) = lo;
        *reinterpret_cast<u32x4*>(buf + q * SHP + (chb * 4 + 2 * h + 1) * 16) = hi2;
      }
      stamp(6);
      tile_barrier();
      if (DBUF) flush_pool();
      if constexpr (LATE) {  // sX has been read for the last time in this tile: its next contents are requested under the depthwise phase
        if (chunk + 1 == chunk1 && tile + 1 < tile_last) load_tile(tyn, txn);
      }
      stamp(7);
      // ---- depthwise 3x3 on the MFMA pipe.  The VALU is what this kernel runs out of (a wave64 instruction costs a SIMD
      // 4 cycles; 72 FMAs per 16 output bytes), the matrix pipe idles.  A depthwise tap is a diagonal matrix:
      //   out[ch][px] += sum_k diag(w_tap)[ch][k] * in[k][px + tap]      (k over the block's channels, 16 per MFMA)
      // so a wave owns 32 channels x two 32-pixel blocks (2 output rows each) and issues 9 taps x 2 k-steps x 2 blocks
      // = 36 MFMAs: the data operand is one ds_read_b128 per MFMA (lane = pixel, 8 channels), the weight operand this
      // lane's weight masked into its diagonal position (4 v_and per tap and k-step).  3 % of the MACs are useful, which
      // still equals the VALU's rate -- on a pipe that was idle, for a quarter of the VALU instructions.
      if constexpr (DWV == 1) {
        // ---- two taps per MFMA: D[16 ch][16 px] += A[16 ch][k] B[k][16 px], k = 16 t + c (tap slot t, channel c of the
        // 16-channel tile), in this order of k: lane (li = lane & 15, g = lane >> 4) holds k-slice g = tap slot g & 1, channels 8 (g >> 1) + j.
        //   B: lane = output pixel li of one tile row; its 8 channels of the halo pixel under tap slot g & 1: ONE ds_read_b128,
        //      address = row base + this lane's tap offset (odd and even lane quarters read different taps)
        //   A: lane = channel li of the tile; its weight of tap slot g & 1 at element li & 7 if (g >> 1) == li >> 3, else 0
        // (k = 16 t + c in MFMA order would put the channel half in g & 1: the 16-lane groups of a ds_read_b128 -- {0-3, 12-15,
        // 20-27}, ... -- would then mix 8 pixels at channel slot s with 8 at slot s + 1 and collide two-way on the 144-byte
        // pitch.  With the tap in g & 1 a group reads 16 pixels of ONE slot, 8 of them shifted by the tap distance (1 or 16
        // pixels; equal addresses broadcast): conflict-free, 4 instead of 8 LDS cycles for each of the 40 reads.)
        // A wave owns 32 channels (2 tiles) x its 4 output rows (4 pixel tiles): 8 accumulator tiles of 4 registers,
        // 5 tap pairs (the last one half empty) = 40 MFMAs of 16 cycles.
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int li = lane & 15, g = lane >> 4;
        uint32_t amask[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
          amask[d] = (((li & 7) >> 1) == d && (g >> 1) == (li >> 3)) ? ((li & 1) ? 0xFFFF0000u : 0x0000FFFFu) : 0u;
        const uint32_t* wpair = reinterpret_cast<const uint32_t*>(wds) + chunk * 64 + chb * 32 + li;  // + (pair * Chid + 16 c2)
        const uint32_t wsel = (g & 1) ? 0x03020302u : 0x01000100u;  // v_perm_b32 selector: this lane quarter's tap, duplicated
        f32x4v dacc[2][4];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int r = 0; r < 4; ++r) dacc[c2][r] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const unsigned char* bbase = buf + ((4 * pxg) * kXH_W + li) * SHP + (chb * 4 + (g >> 1)) * 16;
        const bool upper = (g & 1) != 0;
        vec_t bf[2][4];
        uint32_t wq[2];
        auto ld_step = [&](int step, vec_t (&bb)[4], uint32_t& w2) {  // step = 2 * pair + c2
          const int pr = step >> 1, c2 = step & 1;
          const int ta = 2 * pr, tb = 2 * pr + 1 < 9 ? 2 * pr + 1 : 8;
          const int offa = (ta / 3) * kXH_W + ta % 3, offb = (tb / 3) * kXH_W + tb % 3;
          const unsigned char* p0 = bbase + (upper ? offb : offa) * SHP + c2 * 32;
#pragma unroll
          for (int r = 0; r < 4; ++r) bb[r] = *reinterpret_cast<const vec_t*>(p0 + r * kXH_W * SHP);
          w2 = wpair[pr * a.Chid + 16 * c2];
        };
        ld_step(0, bf[0], wq[0]);
#pragma unroll
        for (int step = 0; step < 10; ++step) {
          if (step + 1 < 10) ld_step(step + 1, bf[(step + 1) & 1], wq[(step + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead reads above this step's MFMAs
          const int c2 = step & 1;
          const uint32_t wdup = __builtin_amdgcn_perm(wq[step & 1], wq[step & 1], wsel);
          u32x4 t;
#pragma unroll
          for (int d = 0; d < 4; ++d) t[d] = wdup & amask[d];
          const vec_t af = reinterpret_cast<const vec_t&>(t);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (!(ABL & 8)) dacc[c2][r] = mfma16x16<T>(af, bf[step & 1][r], dacc[c2][r]);
            else asm volatile("" :: "v"(bf[step & 1][r]), "v"(af));
          }
        }
        // accumulators: lane (pixel li of row r, g): channels 16 c2 + 4 g + e.  v_permlane16_swap between the two tiles gives
        // every lane 8 consecutive channels of its pixel: even g: 4 g .. 4 g + 7, odd g: 16 + 4 (g - 1) .. 16 + 4 g + 3
        const int choff = 8 * (g >> 1) + 16 * (g & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint32_t p0[2], p1[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            typedef T t2 __attribute__((ext_vector_type(2)));
            t2 o0, o1;
            o0[0] = (T)dacc[0][r][2 * j]; o0[1] = (T)dacc[0][r][2 * j + 1];
            o1[0] = (T)dacc[1][r][2 * j]; o1[1] = (T)dacc[1][r][2 * j + 1];
            p0[j] = *reinterpret_cast<uint32_t*>(&o0);
            p1[j] = *reinterpret_cast<uint32_t*>(&o1);
          }
          const u32x2 s0 = __builtin_amdgcn_permlane16_swap(p0[0], p1[0], false, false);
          const u32x2 s1 = __builtin_amdgcn_permlane16_swap(p0[1], p1[1], false, false);
          const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
          const int orow = 4 * pxg + r;
          T* op = out + ((size_t)(y0 + orow) * a.W + x0p + li) * a.Chid + chunk * 64 + chb * 32 + choff;
          if constexpr ((ABL & 4) != 0) op = out + ((size_t)orow * a.W + x0p + li) * a.Chid + chunk * 64 + chb * 32 + choff;
          if constexpr (ABL & 1) asm volatile("" :: "v"(v));
          else if constexpr (NTST) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(op));
          else *reinterpret_cast<u32x4*>(op) = v;
        }
        // SE pool partial: channel sums over the wave's 64 pixels -- the 4 rows in registers, then the 16 pixel lanes of a row by DPP
        if (has_pool && !(ABL & 2)) {
          float v[8];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * c2 + e] = (dacc[c2][0][e] + dacc[c2][1][e]) + (dacc[c2][2][e] + dacc[c2][3][e]);
          // v += rotate(v) inside each 16-lane row, as ONE v_add_f32 with a DPP source per step (through
          // __builtin_amdgcn_update_dpp hipcc emits v_mov 0 + v_mov_dpp + add: 92 instructions for this reduction instead
          // of 32).  The eight values are stepped together, so a register written by one statement is read again eight
          // statements later: the VALU-write -> DPP-read hazard (2 wait states) needs no s_nop inside the strings.
#define LLIE_DPP_STEP(ROR)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                          \
      asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:" #ROR " row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
          // the sums just written by ordinary VALU adds: 2 wait states before the first DPP read.  The eight values are operands of
          // the statement, so every add that produces them is scheduled ABOVE the nop (a bare volatile asm orders only against
          // other side effects, and hipcc's hazard recognizer does not look inside inline asm for the DPP read).
          asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
          LLIE_DPP_STEP(8)
          LLIE_DPP_STEP(4)
          LLIE_DPP_STEP(2)
          LLIE_DPP_STEP(1)
#undef LLIE_DPP_STEP
          if (li == 0) {  // channels 16 c2 + 4 g + e of the wave's block
            float* rp = red + (DBUF ? par : 0) * 256 + wave * 64 + 4 * g;
            *reinterpret_cast<f32x4*>(rp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(rp + 16) = f32x4{v[4], v[5], v[6], v[7]};
          }
          pend_tile = tile; pend_chunk = chunk; pend_par = DBUF ? par : 0;
        }
      } else
      {
        const T* wcol = wds + chunk * 64 + chb * 32 + n;
        f32x16 dacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[0][r] = dacc[1][r] = 0.f;
        // The phase is LDS-latency bound unless the operand reads run well ahead of their MFMAs (two waves per SIMD hide
        // little): all nine weights first, then the data operands of tap t + 1 are in flight while tap t multiplies.
        uint32_t wv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wv[tap] = *reinterpret_cast<const uint16_t*>(wcol + tap * a.Chid);
        vec_t bf[2][4];
        auto ld_tap = [&](int tap, vec_t (&bb)[4]) {
          const int ky = tap / 3, kx = tap % 3;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              bb[blk * 2 + s2] = *reinterpret_cast<const vec_t*>(buf + (dq[blk] + ky * kXH_W + kx) * SHP + (chb * 4 + 2 * s2 + h) * 16);
        };
        ld_tap(0, bf[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) ld_tap(tap + 1, bf[(tap + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead reads above this tap's MFMAs (the scheduler sinks them to
                                              // their use otherwise: one or two reads in flight, the phase waits on LDS latency)
          const uint32_t wdup = wv[tap] | (wv[tap] << 16);
          vec_t af[2];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 t;
#pragma unroll
            for (int d = 0; d < 4; ++d) t[d] = wdup & dmask[s2][d];
            af[s2] = reinterpret_cast<const vec_t&>(t);
          }
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const vec_t bv = bf[tap & 1][blk * 2 + s2];
              if constexpr (!(ABL & 8)) dacc[blk] = mfma16<T>(af[s2], bv, dacc[blk]);
              else asm volatile("" :: "v"(bv), "v"(af[s2]));
            }
        }
        // accumulators: lane = pixel n of the block, 16 channels (r&3) + 8(r>>2) + 4h -> T, lane halves exchanged so that
        // each lane owns 8 consecutive channels, two 16-byte stores per block
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          uint32_t pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            typedef T t2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              t2 o;
              o[0] = (T)dacc[blk][4 * g + 2 * j];
              o[1] = (T)dacc[blk][4 * g + 2 * j + 1];
              pk[g][j] = *reinterpret_cast<uint32_t*>(&o);
            }
          }
          u32x4 lo, hi2;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const u32x2 r02 = __builtin_amdgcn_permlane32_swap(pk[0][j], pk[2][j], false, false);
            const u32x2 r13 = __builtin_amdgcn_permlane32_swap(pk[1][j], pk[3][j], false, false);
            lo[j] = r02[0]; lo[2 + j] = r02[1];
            hi2[j] = r13[0]; hi2[2 + j] = r13[1];
          }
          T* op = out + ((size_t)(y0 + drow[blk]) * a.W + x0p + (n & 15)) * a.Chid + chunk * 64 + chb * 32 + 16 * h;
          if constexpr ((ABL & 4) != 0)  // timing ablation: every tile row lands in image row 0..7 -> the stores hit in L2, no HBM writes
            op = out + ((size_t)(drow[blk]) * a.W + x0p + (n & 15)) * a.Chid + chunk * 64 + chb * 32 + 16 * h;
          if constexpr ((ABL & 16) != 0) {  // timing ablation (wrong layout, same bytes): both stores of a wave write one contiguous KB each
            T* cp = out + (((((size_t)tile * nchunks_all + chunk) * 4 + wave) * 2 + blk) * 1024) + lane * 8;
            *reinterpret_cast<u32x4*>(cp) = lo;
            *reinterpret_cast<u32x4*>(cp + 512) = hi2;
          } else if constexpr (!(ABL & 1)) {
            *reinterpret_cast<u32x4*>(op) = lo;
            *reinterpret_cast<u32x4*>(op + 8) = hi2;
          } else {
            asm volatile("" :: "v"(lo), "v"(hi2));
          }
        }
        // SE pool partial of this (tile, chunk): the 16 channel values of a lane summed over the wave's 64 pixels -- the
        // two blocks in registers, then a halving butterfly over the 32 pixel lanes (lane n ends up with channel slot (n>>1)&15)
        if (has_pool && !(ABL & 2)) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = dacc[0][r] + dacc[1][r];
          // halving butterfly over pixel-lane bits 4 and 3 (lane-dependent choice between two registers as a bit blend with an
          // opaque all-ones / zero mask, v_bfi_b32: written as a ternary the compiler builds a 16-way indexed select chain);
          // bit 4 crosses the 16-lane DPP rows (ds_swizzle, the only LDS round trip), bit 3 is a row rotate by 8
          {
            uint32_t m = (n & 16) ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(m));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const uint32_t lo = __float_as_uint(v[i]), hi = __float_as_uint(v[8 + i]);
              const float keep = __uint_as_float((hi & m) | (lo & ~m));
              const int send = (int)((lo & m) | (hi & ~m));
              v[i] = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(send, 0x401F));  // lane ^ 16
            }
            m = (n & 8) ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(m));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t lo = __float_as_uint(v[i]), hi = __float_as_uint(v[4 + i]);
              const float keep = __uint_as_float((hi & m) | (lo & ~m));
              const int send = (int)((lo & m) | (hi & ~m));
              v[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, send, 0x128, 0xF, 0xF, false));  // row_ror:8 = lane ^ 8
            }
          }
          // the 4 remaining values are complete sums over the 8 lanes that share bits 4..3: quad xor 1, quad xor 2, half mirror
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
            v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
            v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x141, 0xF, 0xF, false));  // row_half_mirror
          }
          if ((n & 7) == 0) {  // r = 8 b4 + 4 b3 + {0..3} -> channel slots 8 (2 b4 + b3) + 4 h + {0..3}: 16 contiguous bytes
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(red + (DBUF ? par : 0) * 256 + wave * 64 + 8 * (n >> 3) + 4 * h) = o;
          }
          pend_tile = tile; pend_chunk = chunk; pend_par = DBUF ? par : 0;
        }
      }
      if (DBUF) par ^= 1;
      stamp(8);
    }
    ty = tyn; tx = txn;
  }
  if constexpr (STAMP) {
    if (a.dbg && lane == 0) {
      const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
      for (int i = 0; i < kXStamps; ++i) a.dbg[(wg * 4 + wave) * kXStamps + i] = tk[i];
    }
  }
  if (has_pool) {
    wg_barrier();
    flush_pool();
    if (a.pool_tot && tid < 64) {
#pragma unroll
      for (int q = 0; q < KS; ++q)
        if (chunk0 + q < chunk1) atomicAdd(a.pool_tot + (size_t)b * a.Chid + (chunk0 + q) * 64 + tid, (unsigned long long)(PACC_LDS ? pacc_lds[q * 64 + tid] : pacc_reg[PACC_LDS ? 0 : q]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
int irbx_stats_rows(int P);
bool irbx_supported(int dtype, int Cin, int c0, int Chid, int H, int W) {
  if (dtype != 1 && dtype != 2) return false;
  if (Cin != 32 && Cin != 64 && Cin != 96) return false;  // Cin = 128: 92 KB of LDS = one workgroup per CU, not worth it
  if (c0 % 16 || Chid != 4 * Cin) return false;
  return W % kXT_W == 0 && H % kXT_H == 0 && (H * W) % irbx_stats_rows(H * W) == 0;
}
int irbx_pool_tiles(int H, int W) { return (H / kXT_H) * (W / kXT_W); }
// pixels per statistics partial: fixed per image size (never a function of the batch: bitwise batch invariance)
int irbx_stats_rows(int P) {
  int rp = 128;  // a power of two in [128, 1024], about P / 64
  while (rp < 1024 && rp * 2 * 64 <= P) rp *= 2;
  return rp;
}

static int g_irbx_dbuf = 0, g_irbx_tiles = 4, g_irbx_stamp = 0, g_irbx_ablate = 0, g_irbx_dwv = 1;
static int g_irbx_grid[3] = {0, 0, 0};  // per input width (32, 64, 96 channels)
void irbx_grid(int ks, int v) {
  for (int i = 0; i < 3; ++i)
    if (ks == 0 || ks == 2 * (i + 1)) g_irbx_grid[i] = v;
}
static int g_irbx_var = kXDefaultVar;
void irbx_var(int v) { g_irbx_var = v; }
void irbx_dwv(int v) { g_irbx_dwv = v; }
void irbx_ablate(int v) { g_irbx_ablate = v; }
static unsigned long long* g_irbx_dbg = nullptr;
static size_t g_irbx_dbg_n = 0;  // entries of the last stamped launch
void irbx_stamp(int v) { g_irbx_stamp = v; }
// mean cycles per wave of the last stamped launch: out[0 .. kXStamps) = the slots listed at STAMP, out[kXStamps] = waves averaged
hipError_t irbx_stamp_fetch(double* out) {
  if (!g_irbx_dbg || !g_irbx_dbg_n) return hipErrorInvalidValue;
  std::vector<unsigned long long> h(g_irbx_dbg_n);
  hipError_t e = hipMemcpy(h.data(), g_irbx_dbg, g_irbx_dbg_n * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  for (int i = 0; i < kXStamps; ++i) out[i] = 0.0;
  for (size_t i = 0; i < g_irbx_dbg_n; ++i) out[i % kXStamps] += (double)h[i];
  for (int i = 0; i < kXStamps; ++i) out[i] /= (double)(g_irbx_dbg_n / kXStamps);
  out[kXStamps] = (double)(g_irbx_dbg_n / kXStamps);
  return hipSuccess;
}
void irbx_tune(int dbuf, int tiles_per_wg) {
  if (dbuf >= 0) g_irbx_dbuf = dbuf;
  if (tiles_per_wg > 0) g_irbx_tiles = tiles_per_wg;
}

template <typename T, int KS, int NBW>
static hipError_t launch_stats_cfg(const IrbxArgs& a, hipStream_t s) {
  const int P = a.H * a.W, RP = irbx_stats_rows(P);
  static const std::string name = std::string("expand_stats_kernel<") + TypeName<T>::value + ", " + std::to_string(KS) + ", " +
                                  std::to_string(NBW) + ">";
  note_kernel(name.c_str());
  hipLaunchKernelGGL((expand_stats_kernel<T, KS, NBW>), dim3(P / RP, 1, a.B), dim3(256), 0, s, a, RP);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_stats_t(const IrbxArgs& a, hipStream_t s) {
  const int Cin = a.c0 + a.c1;
  if (Cin == 32 && a.Chid == 128) return launch_stats_cfg<T, 2, 1>(a, s);
  if (Cin == 64 && a.Chid == 256) return launch_stats_cfg<T, 4, 2>(a, s);
  if (Cin == 96 && a.Chid == 384) return launch_stats_cfg<T, 6, 3>(a, s);
  return hipErrorInvalidValue;
}
hipError_t launch_expand_stats(int dtype, const IrbxArgs& a, hipStream_t s) {
  if (!irbx_supported(dtype, a.c0 + a.c1, a.c0, a.Chid, a.H, a.W) || (a.c1 && !a.x1) || !a.stats) return hipErrorInvalidValue;
  if (a.Chid != 4 * (a.c0 + a.c1)) return hipErrorInvalidValue;
  return dtype == 1 ? launch_stats_t<half_t>(a, s) : launch_stats_t<bf16_t>(a, s);
}

template <typename T, int KS, bool DBUF>
static hipError_t launch_dw_cfg(const IrbxArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(DBUF ? 2 : 1) * kXNPB * 32 * SHP + (size_t)kXNPB * 32 * (16 * KS + 8) * 2 + (size_t)10 * a.Chid * 2 +
                     (size_t)2 * a.Chid * 4 + (size_t)2 * 16 * KS * 4 + 2 * 256 * 4 + (KS <= 4 ? (size_t)KS * 64 * 8 : 0);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = ensure_max_lds(reinterpret_cast<const void*>(&expand_dw_kernel<T, KS, DBUF>), 128 * 1024, attr_done); e != hipSuccess)
    return e;
  const int ntiles = irbx_pool_tiles(a.H, a.W), nchunks = a.Chid / 64;
  // tiles per workgroup: a run along x (neighbouring halo columns hit L1), as long as the launch keeps >= 2048 workgroups
  int tpw = g_irbx_tiles;
  while (tpw > 1 && ((a.W / kXT_W) % tpw || (long)(ntiles / tpw) * a.B < 2048)) tpw >>= 1;
  // channel chunks per workgroup: all of them (x tile loaded once) unless the launch would be too small
  int cpw = nchunks;
  while (cpw > 1 && (long)(ntiles / tpw) * a.B * (nchunks / cpw) < 1024 && cpw % 2 == 0) cpw >>= 1;
  static const std::string name = std::string("expand_dw_kernel<") + TypeName<T>::value + ", " + std::to_string(KS) + ", " +
                                  (DBUF ? "1" : "0") + ">";
  note_kernel(name.c_str());
  dim3 grid(ntiles / tpw, nchunks / cpw, a.B);
  if (g_irbx_grid[KS / 2 - 1] > 0) {  // knobs "irbx_grid", "irbx_grid2/4/6": about this many workgroups per launch, each image's tiles split evenly over its share
    int gx = g_irbx_grid[KS / 2 - 1] / (a.B * (nchunks / cpw));
    gx = gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
    grid.x = gx;
    tpw = 0;
  }
  if constexpr (std::is_same<T, half_t>::value && !DBUF) {
    if (g_irbx_stamp || a.ablate) {  // diagnostic builds: in-kernel cycle stamps and / or timing ablations (fp16 only)
      IrbxArgs b = a;
      if (g_irbx_stamp) {
        const size_t n = (size_t)grid.x * grid.y * grid.z * 4 * kXStamps;
        if (n > g_irbx_dbg_n || !g_irbx_dbg) {
          if (g_irbx_dbg) (void)hipFree(g_irbx_dbg);
          hipError_t e = hipMalloc(reinterpret_cast<void**>(&g_irbx_dbg), n * 8);
          if (e != hipSuccess) return e;
        }
        g_irbx_dbg_n = n;
        b.dbg = g_irbx_dbg;
      }
      auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, b, tpw, cpw);
        return hipGetLastError();
      };
      if (g_irbx_stamp) {
        switch (a.ablate) {
          case 0: return go(&expand_dw_kernel<T, KS, DBUF, true, 0>);
          case 1: return go(&expand_dw_kernel<T, KS, DBUF, true, 1>);
          case 32: return go(&expand_dw_kernel<T, KS, DBUF, true, 32>);
        }
      } else {
        switch (a.ablate) {
          case 1: return go(&expand_dw_kernel<T, KS, DBUF, false, 1>);
          case 4: return go(&expand_dw_kernel<T, KS, DBUF, false, 4>);
          case 8: return go(&expand_dw_kernel<T, KS, DBUF, false, 8>);
          case 32: return go(&expand_dw_kernel<T, KS, DBUF, false, 32>);
          case 33: return go(&expand_dw_kernel<T, KS, DBUF, false, 33>);
        }
      }
      return hipErrorInvalidValue;
    }
  }
  if (g_irbx_dwv == 0) {  // one tap per 32x32x16 MFMA (round 2's form), for A/B runs
    static std::atomic<uint64_t> attr0{0};
    if (hipError_t e = ensure_max_lds(reinterpret_cast<const void*>(&expand_dw_kernel<T, KS, DBUF, false, 0, 0>), 128 * 1024, attr0); e != hipSuccess)
      return e;
    hipLaunchKernelGGL((expand_dw_kernel<T, KS, DBUF, false, 0, 0>), grid, dim3(256), lds, s, a, tpw, cpw);
    return hipGetLastError();
  }
  if constexpr (!DBUF) {
    if (g_irbx_var != kXDefaultVar || a.nt) {  // knob "irbx_var": A/B of the variants (see VAR); IrbxArgs::nt = bit 2
      auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, tpw, cpw);
        return hipGetLastError();
      };
      switch (g_irbx_var | (a.nt ? 2 : 0)) {
        case 0: return go(&expand_dw_kernel<T, KS, DBUF, false, 0, 1, 0>);
        case 1: return go(&expand_dw_kernel<T, KS, DBUF, false, 0, 1, 1>);
        case 2: return go(&expand_dw_kernel<T, KS, DBUF, false, 0, 1, 2>);
        case 3: return go(&expand_dw_kernel<T, KS, DBUF, false, 0, 1, 3>);
      }
      return hipErrorInvalidValue;
    }
  }
  hipLaunchKernelGGL((expand_dw_kernel<T, KS, DBUF>), grid, dim3(256), lds, s, a, tpw, cpw);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_dw_t(const IrbxArgs& a, hipStream_t s) {
  const int Cin = a.c0 + a.c1;
  // llie_tune("irbx_dbuf", 1): double-buffered h1 tile for the 32-channel inputs (one barrier per chunk, 76 KB of LDS = two
  // workgroups per CU).  The single-buffered kernel (49 KB, three workgroups per CU) is the default: 1 % faster end to end
  if (g_irbx_dbuf) {
    switch (Cin) {
      case 32: return launch_dw_cfg<T, 2, true>(a, s);
      case 64: return launch_dw_cfg<T, 4, false>(a, s);
      case 96: return launch_dw_cfg<T, 6, false>(a, s);
    }
  } else {
    switch (Cin) {
      case 32: return launch_dw_cfg<T, 2, false>(a, s);
      case 64: return launch_dw_cfg<T, 4, false>(a, s);
      case 96: return launch_dw_cfg<T, 6, false>(a, s);
    }
  }
  return hipErrorInvalidValue;
}
hipError_t launch_expand_dw(int dtype, const IrbxArgs& a0, hipStream_t s) {
  if (!irbx_supported(dtype, a0.c0 + a0.c1, a0.c0, a0.Chid, a0.H, a0.W) || (a0.c1 && !a0.x1) || !a0.out) return hipErrorInvalidValue;
  IrbxArgs a = a0;
  a.ablate = g_irbx_ablate;
  return dtype == 1 ? launch_dw_t<half_t>(a, s) : launch_dw_t<bf16_t>(a, s);
}

}  // namespace llie
