// Shared device helpers for the gfx950 kernels of libllie_hip.so.
// Wavefront = 64 everywhere; activations are NHWC with 16-byte channel vectors per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "kernels.h"

namespace llie {

typedef _Float16 half_t;
typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// Host: raise a kernel's dynamic-LDS limit once per *device* (the attribute lives in the per-device function object, so a
// process that drives two GPUs -- enhance_sharded, model.to("cuda:1") -- needs it set on both).  `done` is the call
// site's static bit mask of devices already handled.
inline hipError_t ensure_max_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); e != hipSuccess) return e;
  done.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// Element traits: VEC = elements per 16-byte lane vector.
template <typename T> struct TypeName;
template <> struct TypeName<float> { static constexpr const char* value = "float"; };
template <> struct TypeName<_Float16> { static constexpr const char* value = "_Float16"; };
template <> struct TypeName<__bf16> { static constexpr const char* value = "__bf16"; };

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;
  typedef f32x4 vec_t;
};
template <> struct Elem<half_t> {
  static constexpr int VEC = 8;
  typedef f16x8 vec_t;
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  typedef bf16x8 vec_t;
};

template <typename T> __device__ __forceinline__ typename Elem<T>::vec_t ld_vec(const T* p) {
  return *reinterpret_cast<const typename Elem<T>::vec_t*>(p);
}
template <typename T> __device__ __forceinline__ void st_vec(T* p, typename Elem<T>::vec_t v) {
  *reinterpret_cast<typename Elem<T>::vec_t*>(p) = v;
}
// Store with a cache policy chosen per launch (a uniform branch): nt = the tensor is larger than what the Infinity Cache
// will still hold when its consumer runs, so its lines should neither displace what IS re-read (block inputs, residuals)
// nor linger as dirty lines that are written back under the consumer's reads.  Measured on the recompute blocks: h2
// stored non-temporally made the project GEMM that reads it 17 % faster (profiles/r04).
template <typename T> __device__ __forceinline__ void st_vec_pol(T* p, typename Elem<T>::vec_t v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<typename Elem<T>::vec_t*>(p));
  else *reinterpret_cast<typename Elem<T>::vec_t*>(p) = v;
}
// 16-byte vector <-> float[VEC]
template <typename T> __device__ __forceinline__ void vec_to_f32(typename Elem<T>::vec_t v, float* f) {
#pragma unroll
  for (int i = 0; i < Elem<T>::VEC; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ typename Elem<T>::vec_t f32_to_vec(const float* f) {
  typename Elem<T>::vec_t v;
#pragma unroll
  for (int i = 0; i < Elem<T>::VEC; ++i) v[i] = (T)f[i];
  return v;
}
template <typename T> __device__ __forceinline__ void ld_f32(const T* p, float* f) { vec_to_f32<T>(ld_vec<T>(p), f); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, const float* f) { st_vec<T>(p, f32_to_vec<T>(f)); }
// value as the consumer will see it after storage in T
template <typename T> __device__ __forceinline__ float round_to(float x) { return (float)(T)x; }

__device__ __forceinline__ float relu6f(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.f), 6.f); }
__device__ __forceinline__ float siluf(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }
// SiLU for results that are rounded to a 2-byte type right away: v_rcp_f32 (1 ulp) instead of the IEEE division sequence
// (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: ~10 instructions per value; the output head's patch staging is
// bound by exactly these).  The fp32 engine keeps siluf.
__device__ __forceinline__ float siluf_fast(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == ACT_RELU6) return relu6f(x);
  if (act == ACT_SILU) return siluf(x);
  return x;
}

// acc += w * f with w and f the LOW / HIGH f16 halves of two packed registers and an fp32 accumulator: one
// v_fma_mix_f32, no separate conversions.
__device__ __forceinline__ void fma_mix_lo(float& acc, uint32_t w2, uint32_t f2) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(w2), "v"(f2));
}
__device__ __forceinline__ void fma_mix_hi(float& acc, uint32_t w2, uint32_t f2) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(w2), "v"(f2));
}

// ---------------------------------------------------------------------------------------------
// Tile index of workgroup `bid` of a 1-D run of `n` spatial tiles so that each XCD (workgroups are dealt to the eight XCDs
// round-robin, one L2 each) owns a run of n / 8 CONSECUTIVE tiles: neighbours then share halo lines in one L2.  Identity when
// n is not a multiple of 8.  A permutation of [0, n): every tile is still computed exactly once.
__device__ __forceinline__ int xcd_tile_order(int bid, int n) { return (n & 7) ? bid : (bid & 7) * (n >> 3) + (bid >> 3); }

// ---------------------------------------------------------------------------------------------
// Workgroup barrier that first drains this wave's outstanding LDS operations.  __syncthreads() alone is not enough on
// gfx950: the target has back-off barriers, so hipcc does not place an s_waitcnt in front of s_barrier by itself, and its
// memory model treats LDS as totally ordered across the waves of a workgroup -- a ds_write issued just before the
// barrier may still be in flight when a wave on another SIMD reads the location right after it (seen as a rare
// wrong pool partial in expand_dw_kernel).
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Packed affine (+ clamp to [0, 1]) for operand prologues.  The kernels of this engine run out of VALU issue slots
// before anything else (a wave64 instruction holds its SIMD for 4 cycles), so ReLU6 is carried as clamp01(z / 6) -- the
// clamp is the FMA's free output modifier -- and the factor 6 is pushed through the linear operator that follows.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kSixth = 1.f / 6.f;
__device__ __forceinline__ float clamp01f(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }

// one packed dword of T: { [clamp01](a0 * s0 + b0), [clamp01](a1 * s1 + b1) }, all inputs fp32.  Plain C on purpose: callers
// feed it MFMA results, and hipcc pads the MFMA -> VALU read hazard only for instructions it emits itself (an asm
// consumer reads stale accumulators); it folds the med3 into the FMA's clamp bit (v_fma_f32 ... clamp + v_cvt_pk).
template <typename T, bool CLAMP = true>
__device__ __forceinline__ uint32_t affine_clamp01_pack(float a0, float a1, float s0, float s1, float b0, float b1) {
  typedef T t2 __attribute__((ext_vector_type(2)));
  t2 o;
  const float v0 = __builtin_fmaf(a0, s0, b0), v1 = __builtin_fmaf(a1, s1, b1);
  o[0] = (T)(CLAMP ? clamp01f(v0) : v0);
  o[1] = (T)(CLAMP ? clamp01f(v1) : v1);
  return *reinterpret_cast<uint32_t*>(&o);
}
// the same with the two inputs taken from a packed dword of T (loaded data, never an MFMA result): for f16 one
// v_fma_mixlo/hi_f16 per value, rounded once
template <typename T, bool CLAMP = true>
__device__ __forceinline__ uint32_t act_clamp01_pack(uint32_t x2, float s0, float s1, float b0, float b1) {
  if constexpr (std::is_same<T, half_t>::value) {
    uint32_t r;
    if constexpr (CLAMP) {
      asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(x2), "v"(s0), "v"(b0));
      asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(r) : "v"(x2), "v"(s1), "v"(b1));
    } else {
      asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x2), "v"(s0), "v"(b0));
      asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(x2), "v"(s1), "v"(b1));
    }
    return r;
  } else {
    return affine_clamp01_pack<T, CLAMP>(__uint_as_float(x2 << 16), __uint_as_float(x2 & 0xFFFF0000u), s0, s1, b0, b1);
  }
}

// ---------------------------------------------------------------------------------------------
// Wavefront (64-lane) reductions by xor-shuffle.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// SE average-pool partials of the depthwise kernels.  A partial always covers one 8-row segment of a
// strip (whatever the strip height of the launch), so the pooled mean of an image is the same bits
// for every batch size.  Thread layout: lane = 8 * (pixel & 7) + cl, each thread holding VEC channel
// sums of its pixel.  A transposed butterfly over lane bits 3..5 leaves every lane with one channel's
// sum over the wave's 8 pixels (7 shuffles for VEC = 8 instead of 24).
constexpr int kPoolSegRows = 8;
template <int VEC>
__device__ __forceinline__ void pool_segment_flush(float (&psum)[VEC], float* red_wave /* [8 * VEC] */, int lane) {
  static_assert(VEC == 8 || VEC == 4, "");
  const bool h1 = lane & 8, h2 = lane & 16, h3 = lane & 32;
  const int cl = lane & 7;
  if constexpr (VEC == 8) {
    float v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) v4[i] = (h1 ? psum[4 + i] : psum[i]) + __shfl_xor(h1 ? psum[i] : psum[4 + i], 8, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) v2[i] = (h2 ? v4[2 + i] : v4[i]) + __shfl_xor(h2 ? v4[i] : v4[2 + i], 16, 64);
    const float v1 = (h3 ? v2[1] : v2[0]) + __shfl_xor(h3 ? v2[0] : v2[1], 32, 64);
    red_wave[cl * 8 + (h1 ? 4 : 0) + (h2 ? 2 : 0) + (h3 ? 1 : 0)] = v1;
  } else {
    float v2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) v2[i] = (h1 ? psum[2 + i] : psum[i]) + __shfl_xor(h1 ? psum[i] : psum[2 + i], 8, 64);
    float v1 = (h2 ? v2[1] : v2[0]) + __shfl_xor(h2 ? v2[0] : v2[1], 16, 64);
    v1 += __shfl_xor(v1, 32, 64);
    if (!h3) red_wave[cl * 4 + (h1 ? 2 : 0) + (h2 ? 1 : 0)] = v1;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) psum[e] = 0.f;
}
// After the strip: combine the waves of every segment in wave order and write the slab entries
//   pool[b][(seg_y * tiles_x + tx)][c],  seg_y = global 8-row segment index.  red = [nseg][NW][CC].
template <int CC, int NT>
__device__ __forceinline__ void pool_segments_store(const float* red, int nseg, int tid, float* pool_img /* + b*ntiles*C + cbase */,
                                                    int C /* floats between consecutive slab entries */, int seg_y0, int tiles_x, int tx) {
  constexpr int NW = NT / 64;
  for (int i = tid; i < nseg * CC; i += NT) {
    const int seg = i / CC, c = i % CC;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += red[(seg * NW + wv) * CC + c];
    pool_img[((size_t)(seg_y0 + seg) * tiles_x + tx) * C + c] = t;
  }
}

// Order-independent accumulation of fp32 partial sums: each partial is rounded to a multiple of 1 / scale and added
// as a 64-bit integer (memory-side atomic, nothing returned).  Integer adds commute, so the total is bitwise independent
// of which workgroup arrives first -- the engine's float reductions never use float atomics.
__device__ __forceinline__ void fixed_add(unsigned long long* dst, float v, float scale) {
  atomicAdd(dst, (unsigned long long)__float2ll_rn(v * scale));
}
template <int CC, int NT>
__device__ __forceinline__ void pool_segments_add(const float* red, int nseg, int tid, unsigned long long* tot /* + b*C + cbase */, float scale) {
  constexpr int NW = NT / 64;
  for (int i = tid; i < nseg * CC; i += NT) {
    const int seg = i / CC, c = i % CC;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += red[(seg * NW + wv) * CC + c];
    fixed_add(tot + c, t, scale);  // one add per 8-row segment: the total does not depend on the strip height
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA wrappers.  One k-chunk = 32 K-values; lane half h = lane>>5 owns k in [16h, 16h+16) of the
// chunk for BOTH operands (any k permutation is legal as long as A and B agree), so each lane reads
// 16 contiguous elements of its row per chunk.
template <typename T> struct Mfma;
template <> struct Mfma<float> {
  // a, b: 16 floats (this lane's k-slice of its A row / W row)
  static __device__ __forceinline__ void chunk(const float* a, const float* b, f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  }
};
template <> struct Mfma<half_t> {
  static __device__ __forceinline__ void chunk(const half_t* a, const half_t* b, f32x16& acc) {
    const f16x8* av = reinterpret_cast<const f16x8*>(a);
    const f16x8* bv = reinterpret_cast<const f16x8*>(b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[0], bv[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[1], bv[1], acc, 0, 0, 0);
  }
};
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ void chunk(const bf16_t* a, const bf16_t* b, f32x16& acc) {
    const bf16x8* av = reinterpret_cast<const bf16x8*>(a);
    const bf16x8* bv = reinterpret_cast<const bf16x8*>(b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], acc, 0, 0, 0);
  }
};
// one 32x32x16 MFMA on 8 packed 2-byte values per lane
template <typename T>
__device__ __forceinline__ f32x16 mfma16(typename Elem<T>::vec_t a, typename Elem<T>::vec_t b, f32x16 c) {
  if constexpr (std::is_same<T, half_t>::value) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// one 16x16x32 MFMA on 8 packed 2-byte values per lane: lane l holds A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col l & 15];
// C/D: col = l & 15, row = 4 (l >> 4) + register
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ f32x4acc mfma16x16(typename Elem<T>::vec_t a, typename Elem<T>::vec_t b, f32x4acc c) {
  if constexpr (std::is_same<T, half_t>::value) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// C/D layout of every 32x32 MFMA (dtype independent): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LDS row pitch (in elements) for a [rows][32] T tile: 32 elements + one 16-byte pad => conflict-free
// ds_read_b128 of 16-lane groups (80-byte rows for 2-byte T, 144-byte rows for float).
template <typename T> struct TilePitch { static constexpr int value = 32 + Elem<T>::VEC; };

}  // namespace llie
