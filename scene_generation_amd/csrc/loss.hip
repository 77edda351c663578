// Scalar losses (deterministic two-stage wave-shuffle reductions, no host sync), cross-entropy and the
// fused flat-buffer Adam step.  Replaces nn.MSELoss / nn.L1Loss / bce_loss / F.cross_entropy reductions
// (losses.py:26-44,135-175; trainer.py:215,331-340; discriminators.py:35) and torch.optim.Adam
// (trainer.py:60,80,106,133).
#include "common.h"
#include <algorithm>
#include <stdint.h>

namespace {

constexpr int LOSS_BLOCKS = 512;

__device__ __forceinline__ float loss_term(int kind, float a, float b, float t) {
  switch (kind) {
    case SG_LOSS_MSE_CONST: { const float d = a - t; return d * d; }
    case SG_LOSS_MSE: { const float d = a - b; return d * d; }
    case SG_LOSS_L1: return fabsf(a - b);
    case SG_LOSS_MEAN: return a;                                            // wgan losses (losses.py:93-112)
    case SG_LOSS_MSE_SIGMOID_CONST: { const float d = 1.f / (1.f + expf(-a)) - t; return d * d; }   // losses.py:115-132
    case SG_LOSS_BCE_PROB_CONST:      // nn.BCELoss on probabilities (losses.py:147), logs clamped at -100 like torch
      return -(t * fmaxf(logf(a), -100.f) + (1.f - t) * fmaxf(logf(1.f - a), -100.f));
    default: return fmaxf(a, 0.f) - a * t + logf(1.f + expf(-fabsf(a)));   // losses.py:42-44
  }
}

__device__ __forceinline__ float loss_grad(int kind, float a, float b, float t) {
  switch (kind) {
    case SG_LOSS_MSE_CONST: return 2.f * (a - t);
    case SG_LOSS_MSE: return 2.f * (a - b);
    case SG_LOSS_L1: { const float d = a - b; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    case SG_LOSS_MEAN: return 1.f;
    case SG_LOSS_MSE_SIGMOID_CONST: { const float sg = 1.f / (1.f + expf(-a)); return 2.f * (sg - t) * sg * (1.f - sg); }
    case SG_LOSS_BCE_PROB_CONST: {
      const float g1 = logf(a) > -100.f ? 1.f / a : 0.f, g0 = logf(1.f - a) > -100.f ? 1.f / (1.f - a) : 0.f;
      return -(t * g1 - (1.f - t) * g0);
    }
    default: {
      // d/da [max(a,0) - a t + log(1+exp(-|a|))]
      const float e = expf(-fabsf(a));
      const float sgn = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
      return (a > 0.f ? 1.f : 0.f) - t - sgn * e / (1.f + e);
    }
  }
}

__global__ void __launch_bounds__(256) loss_partial_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b,
                                                          float target, size_t n, float* __restrict__ part) {
  __shared__ float red[16];
  float s = 0.f;
  // eight (a, b) pairs in flight before the first term (same terms added in the same order: a thread walks n / (256 blocks)
  // elements -- 66 dependent round trips on a 34 MB feature map when it waited for each pair in turn)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 8 * stride) {
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const size_t j = i + e * stride;
      av[e] = j < n ? a[j] : 0.f;
      bv[e] = (b && j < n) ? b[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (i + e * stride < n) s += loss_term(kind, av[e], bv[e], target);
  }
  s = sg_block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) loss_final_kernel(const float* __restrict__ part, int nb, float scale,
                                                        float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = sg_block_sum(s, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

// The two kernels above as ONE launch: the workgroup that arrives last adds the partials, in the order loss_final_kernel adds
// them (common.h: sg_arrive_last).
__global__ void __launch_bounds__(256) loss_fused_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b,
                                                        float target, size_t n, float* __restrict__ part, int* __restrict__ counter,
                                                        float scale, float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  __shared__ int last;
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 8 * stride) {
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const size_t j = i + e * stride;
      av[e] = j < n ? a[j] : 0.f;
      bv[e] = (b && j < n) ? b[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (i + e * stride < n) s += loss_term(kind, av[e], bv[e], target);
  }
  s = sg_block_sum(s, red);
  if (threadIdx.x == 0) sg_publish(&part[blockIdx.x], s);
  if (!sg_arrive_last(counter, (int)gridDim.x, &last)) return;
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += sg_consume(&part[i]);
  t = sg_block_sum(t, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + t * scale;
}

// ---- several scalar losses of one kind and their weighted sum in one launch ------------------------------------------------------
//   total = sum_t w[t] * (scale[t] * sum_i l(a_t[i], b_t[i] | target[t]))
// e.g. the 8 + 3 feature-matching L1 terms of trainer.py:331-340, the per-scale terms of GANLoss (losses.py:166-172), the five
// VGG terms (losses.py:220-224).  Same arithmetic, in the same order, as sg_loss_fwd per term followed by
// sg_weighted_sum_fwd: term t owns the workgroups [blk0[t], blk0[t+1]) (as many as sg_loss_fwd would launch for it), the last
// workgroup to arrive reduces every term's partials and adds the terms in index order.
struct MultiLossArgs {
  const float* a[SG_WSUM_MAX]; const float* b[SG_WSUM_MAX]; float* ga[SG_WSUM_MAX];
  unsigned long long n[SG_WSUM_MAX];
  float scale[SG_WSUM_MAX], w[SG_WSUM_MAX], target[SG_WSUM_MAX];
  int blk0[SG_WSUM_MAX + 1];
  int nterms, kind;
};
__device__ __forceinline__ int multi_term_of(const MultiLossArgs& A, int blk) {
  int t = 0;
  for (int q = 1; q < A.nterms; ++q) t += blk >= A.blk0[q] ? 1 : 0;
  return t;
}
__global__ void __launch_bounds__(256) multi_loss_fwd_kernel(MultiLossArgs A, float* __restrict__ part, int* __restrict__ counter,
                                                            float* __restrict__ out, float* __restrict__ terms_out) {
  __shared__ float red[16];
  __shared__ int last;
  const int t = multi_term_of(A, blockIdx.x), lb = blockIdx.x - A.blk0[t], nb = A.blk0[t + 1] - A.blk0[t];
  const float* __restrict__ a = A.a[t];
  const float* __restrict__ b = A.b[t];
  const size_t n = (size_t)A.n[t], stride = (size_t)nb * 256;
  const float target = A.target[t];
  float s = 0.f;
  for (size_t i = (size_t)lb * 256 + threadIdx.x; i < n; i += 8 * stride) {
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const size_t j = i + e * stride;
      av[e] = j < n ? a[j] : 0.f;
      bv[e] = (b && j < n) ? b[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (i + e * stride < n) s += loss_term(A.kind, av[e], bv[e], target);
  }
  s = sg_block_sum(s, red);
  if (counter == nullptr) {                      // two-kernel form: multi_loss_final_kernel follows
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    return;
  }
  if (threadIdx.x == 0) sg_publish(&part[blockIdx.x], s);
  if (!sg_arrive_last(counter, (int)gridDim.x, &last)) return;
  float total = 0.f;
  for (int q = 0; q < A.nterms; ++q) {
    float v = 0.f;
    for (int i = threadIdx.x; i < A.blk0[q + 1] - A.blk0[q]; i += 256) v += sg_consume(&part[A.blk0[q] + i]);
    v = sg_block_sum(v, red);
    const float term = 0.f + v * A.scale[q];
    if (threadIdx.x == 0 && terms_out) terms_out[q] = term;
    total += A.w[q] * term;
  }
  if (threadIdx.x == 0) out[0] = total;
}
__global__ void __launch_bounds__(256) multi_loss_final_kernel(MultiLossArgs A, const float* __restrict__ part,
                                                              float* __restrict__ out, float* __restrict__ terms_out) {
  __shared__ float red[16];
  float total = 0.f;
  for (int q = 0; q < A.nterms; ++q) {
    float v = 0.f;
    for (int i = threadIdx.x; i < A.blk0[q + 1] - A.blk0[q]; i += 256) v += part[A.blk0[q] + i];
    v = sg_block_sum(v, red);
    const float term = 0.f + v * A.scale[q];
    if (threadIdx.x == 0 && terms_out) terms_out[q] = term;
    total += A.w[q] * term;
  }
  if (threadIdx.x == 0) out[0] = total;
}
// ga_t[i] = ((w[t] * gout) * scale[t]) * dl/da: sg_weighted_sum_bwd followed by sg_loss_bwd per term, one launch
__global__ void multi_loss_bwd_kernel(MultiLossArgs A, const float* __restrict__ gout) {
  const int t = multi_term_of(A, blockIdx.x);
  float* __restrict__ ga = A.ga[t];
  if (ga == nullptr) return;
  const size_t i = (size_t)(blockIdx.x - A.blk0[t]) * 256 + threadIdx.x;
  if (i >= (size_t)A.n[t]) return;
  const float g = A.w[t] * gout[0];
  const float* __restrict__ b = A.b[t];
  ga[i] = g * A.scale[t] * loss_grad(A.kind, A.a[t][i], b ? b[i] : 0.f, A.target[t]);
}

__global__ void loss_bwd_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, float target, size_t n,
                                float scale, const float* __restrict__ gout, float* __restrict__ ga) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ga[i] = gout[0] * scale * loss_grad(kind, a[i], b ? b[i] : 0.f, target);
}

// one wave per row
__global__ void __launch_bounds__(256) ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                    int rows, int classes, float* __restrict__ row_loss) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = logits + (size_t)row * classes;
  float m = -INFINITY;
  for (int c = lane; c < classes; c += 64) m = fmaxf(m, x[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int c = lane; c < classes; c += 64) s += expf(x[c] - m);
  s = sg_wave_sum(s);
  if (lane == 0) row_loss[row] = (m + logf(s)) - x[target[row]];
}

__global__ void __launch_bounds__(256) ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                    int rows, int classes, const float* __restrict__ gout,
                                                    float* __restrict__ gl) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = logits + (size_t)row * classes;
  float m = -INFINITY;
  for (int c = lane; c < classes; c += 64) m = fmaxf(m, x[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int c = lane; c < classes; c += 64) s += expf(x[c] - m);
  s = sg_wave_sum(s);
  const float g = gout[0] / (float)rows;
  const int t = (int)target[row];
  for (int c = lane; c < classes; c += 64) gl[(size_t)row * classes + c] = g * (expf(x[c] - m) / s - (c == t ? 1.f : 0.f));
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float gscale, float step_size, float beta1,
                                         float beta2, float eps, float bc2_sqrt) {
  const float gi = g * gscale;                               // 1 / world under data parallelism (the all-reduce SUMs), else 1
  const float mi = m + (1.f - beta1) * (gi - m);             // exp_avg.lerp_(grad, 1-beta1)
  const float vi = v * beta2 + (1.f - beta2) * gi * gi;      // mul_(beta2).addcmul_(g, g, 1-beta2)
  m = mi; v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p = p - step_size * (mi / denom);                          // addcdiv_(exp_avg, denom, value=-step_size)
}

// one parameter per thread.  (A float4-per-thread form -- four parameters through 16-byte loads / stores -- was measured SLOWER on
// MI355X in round 5: 5.5 vs 6.2 TB/s over the generator's 191 M parameters; the 4-byte form keeps four times the waves in flight.)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float gscale, float step_size, float beta1, float beta2, float eps, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pi = p[i], mi = m[i], vi = v[i];
  adam_one(pi, g[i], mi, vi, gscale, step_size, beta1, beta2, eps, bc2_sqrt);
  m[i] = mi; v[i] = vi; p[i] = pi;
}

__global__ void fill_kernel(float* __restrict__ p, float value, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = value;
}

__global__ void scale_kernel(float* __restrict__ p, float a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] *= a;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}
__global__ void axpy4_kernel(float4* __restrict__ y, const float4* __restrict__ x, float a, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 u = y[i];
  const float4 v = x[i];
  u.x += a * v.x; u.y += a * v.y; u.z += a * v.z; u.w += a * v.w;
  y[i] = u;
}

// y += x ; x = 0 (float4 form: the flat gradient buffers are 256-byte aligned and 64-element padded)
__global__ void add_clear4_kernel(float4* __restrict__ y, float4* __restrict__ x, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 u = y[i];
  const float4 v = x[i];
  u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
  y[i] = u;
  x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void add_clear_kernel(float* __restrict__ y, float* __restrict__ x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { y[i] += x[i]; x[i] = 0.f; }
}

// total = sum_i w[i] * *term[i] in index order (one thread: <= 32 terms), and its dual g[i] = w[i] * gout
struct WsumArgs { const float* term[SG_WSUM_MAX]; float w[SG_WSUM_MAX]; int n; };
__global__ void wsum_fwd_kernel(WsumArgs a, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < a.n; ++i) s += a.w[i] * a.term[i][0];
    out[0] = s;
  }
}
__global__ void wsum_bwd_kernel(WsumArgs a, const float* __restrict__ gout, float* __restrict__ g) {
  const int i = threadIdx.x;
  if (i < a.n) g[i] = a.w[i] * gout[0];
}

}  // namespace

extern "C" size_t sg_loss_ws_bytes(int64_t n) { (void)n; return LOSS_BLOCKS * sizeof(float); }

extern "C" int sg_loss_fwd(int kind, const float* a, const float* b, float target, int64_t n, float scale, float* out,
                           int accumulate, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(a && out && ws && n > 0 && kind >= 0 && kind <= SG_LOSS_BCE_PROB_CONST, "sg_loss_fwd: bad arguments");
  SG_ARG_CHECK((kind != SG_LOSS_MSE && kind != SG_LOSS_L1) || b, "sg_loss_fwd: pair loss needs b");
  SG_ARG_CHECK(ws_bytes >= LOSS_BLOCKS * sizeof(float), "sg_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int nb = sg_cdiv(n, 256 * 8);
  nb = nb < 1 ? 1 : (nb > LOSS_BLOCKS ? LOSS_BLOCKS : nb);
  if (int* counter = sg_counter_alloc(s)) {
    hipLaunchKernelGGL(loss_fused_kernel, dim3(nb), dim3(256), 0, s, kind, a, b, target, (size_t)n, (float*)ws, counter, scale, out,
                       accumulate);
  } else {
    hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, s, kind, a, b, target, (size_t)n, (float*)ws);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, nb, scale, out, accumulate);
  }
  SG_LAUNCH_CHECK("sg_loss_fwd");
  return 0;
}

extern "C" int sg_loss_bwd(int kind, const float* a, const float* b, float target, int64_t n, float scale, const float* gout,
                           float* ga, sgStream stream) {
  SG_ARG_CHECK(a && gout && ga && n > 0 && kind >= 0 && kind <= SG_LOSS_BCE_PROB_CONST, "sg_loss_bwd: bad arguments");
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, kind, a, b, target, (size_t)n,
                     scale, gout, ga);
  SG_LAUNCH_CHECK("sg_loss_bwd");
  return 0;
}

namespace {
inline int loss_blocks(int64_t n) {
  int nb = sg_cdiv(n, 256 * 8);
  return nb < 1 ? 1 : (nb > LOSS_BLOCKS ? LOSS_BLOCKS : nb);
}
bool multi_args(MultiLossArgs& A, int kind, int nterms, const void* const* a_host, const void* const* b_host, const int64_t* n_host,
                const float* scale_host, const float* weight_host, const float* target_host, bool bwd) {
  A.nterms = nterms; A.kind = kind;
  long blocks = 0;
  for (int t = 0; t < SG_WSUM_MAX; ++t) {
    const bool on = t < nterms;
    A.a[t] = on ? (const float*)a_host[t] : nullptr;
    A.b[t] = (on && b_host) ? (const float*)b_host[t] : nullptr;
    A.ga[t] = nullptr;
    A.n[t] = on ? (unsigned long long)n_host[t] : 0ull;
    A.scale[t] = on ? scale_host[t] : 0.f;
    A.w[t] = on ? weight_host[t] : 0.f;
    A.target[t] = (on && target_host) ? target_host[t] : 0.f;
    A.blk0[t] = (int)blocks;
    if (on) {
      if (!A.a[t] || n_host[t] <= 0) return false;
      if ((kind == SG_LOSS_MSE || kind == SG_LOSS_L1) && !A.b[t]) return false;
      blocks += bwd ? sg_cdiv(n_host[t], 256) : loss_blocks(n_host[t]);
    }
    if (blocks > 0x3fffffff) return false;
  }
  A.blk0[SG_WSUM_MAX] = (int)blocks;
  for (int t = nterms; t <= SG_WSUM_MAX; ++t) A.blk0[t] = (int)blocks;
  return true;
}
}  // namespace

extern "C" size_t sg_multi_loss_ws_bytes(int nterms) { return (size_t)(nterms > 0 ? nterms : 1) * LOSS_BLOCKS * sizeof(float); }

extern "C" int sg_multi_loss_fwd(int kind, int nterms, const void* const* a_host, const void* const* b_host, const int64_t* n_host,
                                 const float* scale_host, const float* weight_host, const float* target_host, float* out,
                                 float* terms_out, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(a_host && n_host && scale_host && weight_host && out && ws && nterms > 0 && nterms <= SG_WSUM_MAX && kind >= 0 &&
               kind <= SG_LOSS_BCE_PROB_CONST, "sg_multi_loss_fwd: bad arguments (nterms=%d)", nterms);
  SG_ARG_CHECK(ws_bytes >= sg_multi_loss_ws_bytes(nterms), "sg_multi_loss_fwd: workspace too small");
  MultiLossArgs A;
  SG_ARG_CHECK(multi_args(A, kind, nterms, a_host, b_host, n_host, scale_host, weight_host, target_host, false),
               "sg_multi_loss_fwd: bad term (null operand, n <= 0, or a pair loss without b)");
  hipStream_t s = (hipStream_t)stream;
  const int blocks = A.blk0[nterms];
  int* counter = sg_counter_alloc(s);
  hipLaunchKernelGGL(multi_loss_fwd_kernel, dim3(blocks), dim3(256), 0, s, A, (float*)ws, counter, out, terms_out);
  if (!counter) hipLaunchKernelGGL(multi_loss_final_kernel, dim3(1), dim3(256), 0, s, A, (const float*)ws, out, terms_out);
  SG_LAUNCH_CHECK("sg_multi_loss_fwd");
  return 0;
}

extern "C" int sg_multi_loss_bwd(int kind, int nterms, const void* const* a_host, const void* const* b_host, const int64_t* n_host,
                                 const float* scale_host, const float* weight_host, const float* target_host, const float* gout,
                                 void* const* ga_host, sgStream stream) {
  SG_ARG_CHECK(a_host && n_host && scale_host && weight_host && gout && ga_host && nterms > 0 && nterms <= SG_WSUM_MAX && kind >= 0 &&
               kind <= SG_LOSS_BCE_PROB_CONST, "sg_multi_loss_bwd: bad arguments (nterms=%d)", nterms);
  MultiLossArgs A;
  SG_ARG_CHECK(multi_args(A, kind, nterms, a_host, b_host, n_host, scale_host, weight_host, target_host, true),
               "sg_multi_loss_bwd: bad term");
  for (int t = 0; t < nterms; ++t) A.ga[t] = (float*)ga_host[t];           // nullptr: that term's gradient is not wanted
  hipLaunchKernelGGL(multi_loss_bwd_kernel, dim3(A.blk0[nterms]), dim3(256), 0, (hipStream_t)stream, A, gout);
  SG_LAUNCH_CHECK("sg_multi_loss_bwd");
  return 0;
}

extern "C" int sg_cross_entropy_fwd(const float* logits, const int64_t* target, int rows, int classes, float* row_loss,
                                    float* out, sgStream stream) {
  SG_ARG_CHECK(logits && target && row_loss && out && rows > 0 && classes > 0, "sg_cross_entropy_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(sg_cdiv(rows, 4)), dim3(256), 0, s, logits, target, rows, classes, row_loss);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, rows, 1.f / (float)rows, out, 0);
  SG_LAUNCH_CHECK("sg_cross_entropy_fwd");
  return 0;
}

extern "C" int sg_cross_entropy_bwd(const float* logits, const int64_t* target, int rows, int classes, const float* gout,
                                    float* glogits, sgStream stream) {
  SG_ARG_CHECK(logits && target && gout && glogits && rows > 0 && classes > 0, "sg_cross_entropy_bwd: bad arguments");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(sg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, target, rows, classes,
                     gout, glogits);
  SG_LAUNCH_CHECK("sg_cross_entropy_bwd");
  return 0;
}

extern "C" int sg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float bias_corr1, float bias_corr2_sqrt, float grad_scale, sgStream stream) {
  SG_ARG_CHECK(p && g && m && v && n > 0 && bias_corr1 > 0.f && bias_corr2_sqrt > 0.f, "sg_adam_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_ADAM, s, 0, 28.0 * (double)n);
  hipLaunchKernelGGL(adam_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, p, g, m, v, (size_t)n, grad_scale, lr / bias_corr1, beta1,
                     beta2, eps, bias_corr2_sqrt);
  SG_LAUNCH_CHECK("sg_adam_step");
  return 0;
}

extern "C" int sg_fill(float* p, float value, int64_t n, sgStream stream) {
  SG_ARG_CHECK(p && n >= 0, "sg_fill: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, value, (size_t)n);
  SG_LAUNCH_CHECK("sg_fill");
  return 0;
}

// Host batch -> device: a kernel that READS page-locked host memory through its device mapping (hipHostMalloc memory is
// mapped into the device's address space) and writes HBM.  16 B per lane and load, four loads in flight; the grid is kept small
// (the PCIe link, not the CUs, bounds it: ~8 MB per batch).  See scene_generation_amd/pipeline.py for why the input batch does
// not go through hipMemcpyAsync on a copy stream.
typedef unsigned int stage_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stage_copy_kernel(stage_u4* __restrict__ dst, const stage_u4* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const stage_u4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

extern "C" int sg_stage_copy(void* dst, const void* src_host_mapped, int64_t nbytes, sgStream stream) {
  SG_ARG_CHECK(dst && src_host_mapped && nbytes >= 0 && nbytes % 16 == 0 && ((uintptr_t)dst % 16) == 0 &&
               ((uintptr_t)src_host_mapped % 16) == 0, "sg_stage_copy: operands must be 16-byte aligned, nbytes a multiple of 16");
  if (nbytes == 0) return 0;
  // the device-side address of the page-locked buffer (identical to the host address for hipHostMalloc memory, possibly not
  // for hipHostRegister-ed memory); fails for pageable memory, which a kernel must not be pointed at
  void* src_dev = nullptr;
  if (hipHostGetDevicePointer(&src_dev, const_cast<void*>(src_host_mapped), 0) != hipSuccess || !src_dev) {
    (void)hipGetLastError();
    sg_set_error("sg_stage_copy: the source is not page-locked, device-mapped host memory");
    return -1;
  }
  const size_t n16 = (size_t)nbytes / 16;
  const unsigned blocks = (unsigned)std::min<size_t>(sg_cdiv(n16, 256), 1024);
  hipLaunchKernelGGL(stage_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (stage_u4*)dst, (const stage_u4*)src_dev, n16);
  SG_LAUNCH_CHECK("sg_stage_copy");
  return 0;
}

extern "C" int sg_scale(float* p, float alpha, int64_t n, sgStream stream) {
  SG_ARG_CHECK(p && n >= 0, "sg_scale: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, alpha, (size_t)n);
  SG_LAUNCH_CHECK("sg_scale");
  return 0;
}

extern "C" int sg_axpy(float* y, const float* x, float alpha, int64_t n, sgStream stream) {
  SG_ARG_CHECK(y && x && n >= 0, "sg_axpy: bad arguments");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) == 0)
    hipLaunchKernelGGL(axpy4_kernel, dim3(sg_cdiv(n / 4, 256)), dim3(256), 0, s, (float4*)y, (const float4*)x, alpha, (size_t)(n / 4));
  else
    hipLaunchKernelGGL(axpy_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, y, x, alpha, (size_t)n);
  SG_LAUNCH_CHECK("sg_axpy");
  return 0;
}

extern "C" int sg_add_clear(float* y, float* x, int64_t n, sgStream stream) {
  SG_ARG_CHECK(y && x && n >= 0, "sg_add_clear: bad arguments");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) == 0)
    hipLaunchKernelGGL(add_clear4_kernel, dim3(sg_cdiv(n / 4, 256)), dim3(256), 0, s, (float4*)y, (float4*)x, (size_t)(n / 4));
  else
    hipLaunchKernelGGL(add_clear_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, y, x, (size_t)n);
  SG_LAUNCH_CHECK("sg_add_clear");
  return 0;
}

// out = a + b / out = a * b * alpha: the shortcut add of build_cnn's residual blocks (layers.py:84-118) and the mask multiply of
// nn.Dropout (layers.py:230) -- neither is on the benchmark path, both are plain HBM-bound passes
__global__ void ewise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n,
                             int mul, float alpha) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = mul ? a[i] * b[i] * alpha : a[i] + b[i];
}
extern "C" int sg_add(const float* a, const float* b, float* out, int64_t n, sgStream stream) {
  SG_ARG_CHECK(a && b && out && n >= 0, "sg_add: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ewise_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n, 0, 1.f);
  SG_LAUNCH_CHECK("sg_add");
  return 0;
}
extern "C" int sg_mul(const float* a, const float* b, float alpha, float* out, int64_t n, sgStream stream) {
  SG_ARG_CHECK(a && b && out && n >= 0, "sg_mul: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ewise_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n, 1, alpha);
  SG_LAUNCH_CHECK("sg_mul");
  return 0;
}

extern "C" int sg_weighted_sum_fwd(const void* const* terms_host, const float* weights_host, int n, float* out,
                                   sgStream stream) {
  SG_ARG_CHECK(terms_host && weights_host && out && n > 0 && n <= SG_WSUM_MAX, "sg_weighted_sum_fwd: bad arguments (n=%d)", n);
  WsumArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) { a.term[i] = (const float*)terms_host[i]; a.w[i] = weights_host[i]; }
  for (int i = n; i < SG_WSUM_MAX; ++i) { a.term[i] = nullptr; a.w[i] = 0.f; }
  hipLaunchKernelGGL(wsum_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  SG_LAUNCH_CHECK("sg_weighted_sum_fwd");
  return 0;
}

extern "C" int sg_weighted_sum_bwd(const float* weights_host, int n, const float* gout, float* gterms, sgStream stream) {
  SG_ARG_CHECK(weights_host && gout && gterms && n > 0 && n <= SG_WSUM_MAX, "sg_weighted_sum_bwd: bad arguments (n=%d)", n);
  WsumArgs a;
  a.n = n;
  for (int i = 0; i < SG_WSUM_MAX; ++i) { a.term[i] = nullptr; a.w[i] = i < n ? weights_host[i] : 0.f; }
  hipLaunchKernelGGL(wsum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, gout, gterms);
  SG_LAUNCH_CHECK("sg_weighted_sum_bwd");
  return 0;
}
