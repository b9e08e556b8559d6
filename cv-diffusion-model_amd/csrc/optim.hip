// Optimiser step of the training loop in two launches + one single-workgroup launch between them, over every parameter
// tensor at once.
//
// Replaces, for the trainer's inner loop (reference src/training/trainer.py:300-324), the eager sequence
//   scaler.unscale_ -> torch.nn.utils.clip_grad_norm_(params, max_norm) -> optimizer.step() [torch.optim.AdamW]
//   -> EMAModel.update (trainer.py:98-104: shadow = decay * shadow + (1 - decay) * param)
// which on 321 parameter tensors is ~40 launches and several milliseconds of host time per step; here the host cost is
// three launches, whatever the number of tensors.  The arithmetic is torch's, operation for operation (torch/optim/adamw.py,
// _single_tensor_adamw: decoupled decay p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)), the clip coefficient is clip_grad_norm_'s
// min(1, max_norm / (||g|| + 1e-6)).
//
// Layout: `tensors[i]` = one parameter (fp32 master, first / second moment, optional EMA shadow, offset of its gradient
// in the flat gradient buffer the backward pass writes); `chunks[c]` = (tensor, first element) of a run of at most kOptChunk
// elements.  HBM-bound: 4 (g) bytes per element in the norm pass, 4 x 5 read + 4 x 4 written in the update.
// The gradient norm is a fixed-order sum (per-thread, per-workgroup, then one workgroup over the chunk partials, in double):
// deterministic run to run.
#include "common.h"
#include "kernels.h"

namespace llie {

constexpr int kOptThreads = 256;

__device__ __forceinline__ bool opt_aligned16(const void* a) { return (reinterpret_cast<uintptr_t>(a) & 15) == 0; }

// sum of squares of (grad * gscale) per chunk -> partial[chunk]
__global__ void __launch_bounds__(kOptThreads) opt_sumsq_kernel(const OptTensor* __restrict__ tensors, const OptChunk* __restrict__ chunks,
                                                               const float* __restrict__ gbase, double* __restrict__ partial) {
  const OptChunk ch = chunks[blockIdx.x];
  const OptTensor t = tensors[ch.tensor];
  const float* g = gbase + t.goff + ch.first;
  const long long left = t.n - ch.first;
  const int n = left < kOptChunk ? (int)left : kOptChunk;
  float acc = 0.f;
  if (opt_aligned16(g)) {
    const int nv = n >> 2;
    for (int i = threadIdx.x; i < nv; i += kOptThreads) {
      const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
      acc += v[0] * v[0];
      acc += v[1] * v[1];
      acc += v[2] * v[2];
      acc += v[3] * v[3];
    }
    for (int i = (nv << 2) + threadIdx.x; i < n; i += kOptThreads) acc += g[i] * g[i];
  } else {
    for (int i = threadIdx.x; i < n; i += kOptThreads) acc += g[i] * g[i];
  }
  __shared__ double red[kOptThreads];
  red[threadIdx.x] = (double)acc;
  wg_barrier();
  for (int o = kOptThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    wg_barrier();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// stats[0] = ||gscale * g||, stats[1] = factor applied to every gradient = gscale * clip coefficient, stats[2] = 1 if the step
// is skipped (non-finite norm and skip_nonfinite: GradScaler.step's behaviour, torch/amp/grad_scaler.py)
__global__ void __launch_bounds__(kOptThreads) opt_clip_kernel(const double* __restrict__ partial, int nchunks, float gscale, float max_norm,
                                                              int skip_nonfinite, float* __restrict__ stats) {
  __shared__ double red[kOptThreads];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += kOptThreads) acc += partial[i];
  red[threadIdx.x] = acc;
  wg_barrier();
  for (int o = kOptThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    wg_barrier();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]) * fabsf(gscale);
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;  // NaN stays NaN, as torch.clamp(max=1.0) leaves it
    }
    const bool bad = !(norm <= 3.4028234664e38f);  // inf or NaN
    stats[0] = norm;
    stats[1] = gscale * coef;
    stats[2] = (bad && skip_nonfinite) ? 1.f : 0.f;
  }
}

struct OptHyper {
  float decay_mul;   // 1 - lr * weight_decay (formed in double, as Python does for torch)
  float one_m_b1;    // 1 - beta1
  float beta2;
  float one_m_b2;    // 1 - beta2
  float step_size;   // lr / (1 - beta1^t)
  float bc2_sqrt;    // sqrt(1 - beta2^t)
  float eps;
  float ema_decay;   // < 0: no EMA
  float one_m_ema;
};

__device__ __forceinline__ void opt_update(float g, float& p, float& m, float& v, float& e, const OptHyper& h, float gf, bool has_ema) {
  g *= gf;
  p *= h.decay_mul;
  m = m + (g - m) * h.one_m_b1;
  v = v * h.beta2 + (h.one_m_b2 * g) * g;
  const float denom = __fsqrt_rn(v) / h.bc2_sqrt + h.eps;
  p = p + (m / denom) * (-h.step_size);
  if (has_ema) e = e * h.ema_decay + p * h.one_m_ema;
}

__global__ void __launch_bounds__(kOptThreads) opt_adamw_kernel(const OptTensor* __restrict__ tensors, const OptChunk* __restrict__ chunks,
                                                               const float* __restrict__ gbase, const float* __restrict__ stats, const OptHyper h) {
  if (stats[2] != 0.f) return;  // skipped step: nothing moves
  const float gf = stats[1];
  const OptChunk ch = chunks[blockIdx.x];
  const OptTensor t = tensors[ch.tensor];
  const float* g = gbase + t.goff + ch.first;
  float* p = t.p + ch.first;
  float* m = t.m + ch.first;
  float* v = t.v + ch.first;
  const bool has_ema = t.ema != nullptr && h.ema_decay >= 0.f;
  float* e = has_ema ? t.ema + ch.first : nullptr;
  const long long left = t.n - ch.first;
  const int n = left < kOptChunk ? (int)left : kOptChunk;
  int done = 0;
  if (opt_aligned16(g) && opt_aligned16(p) && opt_aligned16(m) && opt_aligned16(v) && (!has_ema || opt_aligned16(e))) {
    const int nv = n >> 2;
    for (int i = threadIdx.x; i < nv; i += kOptThreads) {
      const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
      f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], ev = {0.f, 0.f, 0.f, 0.f};
      if (has_ema) ev = reinterpret_cast<f32x4*>(e)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = pv[k], mk = mv[k], vk = vv[k], ek = ev[k];
        opt_update(gv[k], pk, mk, vk, ek, h, gf, has_ema);
        pv[k] = pk; mv[k] = mk; vv[k] = vk; ev[k] = ek;
      }
      reinterpret_cast<f32x4*>(p)[i] = pv;
      reinterpret_cast<f32x4*>(m)[i] = mv;
      reinterpret_cast<f32x4*>(v)[i] = vv;
      if (has_ema) reinterpret_cast<f32x4*>(e)[i] = ev;
    }
    done = nv << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += kOptThreads) {
    float pk = p[i], mk = m[i], vk = v[i], ek = has_ema ? e[i] : 0.f;
    opt_update(g[i], pk, mk, vk, ek, h, gf, has_ema);
    p[i] = pk; m[i] = mk; v[i] = vk;
    if (has_ema) e[i] = ek;
  }
}

hipError_t launch_optimizer_step(const OptStepArgs& a, hipStream_t s) {
  if (!a.tensors || !a.chunks || !a.gbase || !a.partial || !a.stats || a.nchunks <= 0 || a.step < 1) return hipErrorInvalidValue;
  if (!(a.lr >= 0.0) || !(a.beta1 >= 0.0 && a.beta1 < 1.0) || !(a.beta2 >= 0.0 && a.beta2 < 1.0) || !(a.eps >= 0.0) || !(a.weight_decay >= 0.0) ||
      !(a.ema_decay <= 1.0))
    return hipErrorInvalidValue;
  note_kernel("opt_sumsq_kernel");
  hipLaunchKernelGGL(opt_sumsq_kernel, dim3(a.nchunks), dim3(kOptThreads), 0, s, a.tensors, a.chunks, a.gbase, a.partial);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  note_kernel("opt_clip_kernel");
  hipLaunchKernelGGL(opt_clip_kernel, dim3(1), dim3(kOptThreads), 0, s, a.partial, a.nchunks, (float)a.grad_scale, (float)a.max_grad_norm, a.skip_nonfinite, a.stats);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  OptHyper h;
  const double bc1 = 1.0 - pow(a.beta1, (double)a.step), bc2 = 1.0 - pow(a.beta2, (double)a.step);
  h.decay_mul = (float)(1.0 - a.lr * a.weight_decay);
  h.one_m_b1 = (float)(1.0 - a.beta1);
  h.beta2 = (float)a.beta2;
  h.one_m_b2 = (float)(1.0 - a.beta2);
  h.step_size = (float)(a.lr / bc1);
  h.bc2_sqrt = (float)sqrt(bc2);
  h.eps = (float)a.eps;
  h.ema_decay = (float)a.ema_decay;
  h.one_m_ema = (float)(1.0 - a.ema_decay);
  note_kernel("opt_adamw_kernel");
  hipLaunchKernelGGL(opt_adamw_kernel, dim3(a.nchunks), dim3(kOptThreads), 0, s, a.tensors, a.chunks, a.gbase, a.stats, h);
  return hipGetLastError();
}

}  // namespace llie
