// HBM-bound normalisation / pooling / elementwise kernels (wave64 shuffle reductions, float4 streams).
// Replaces nn.InstanceNorm2d (layers.py:296), nn.BatchNorm2d (generators.py:22, layers.py:23-31),
// nn.AvgPool2d(3,2,1,count_include_pad=False) (discriminators.py:100,186), GlobalAvgPool (layers.py:82-85)
// and the activation modules fused behind them.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float act_grad_from_pre(float pre, int act, float slope) {
  switch (act) {
    case SG_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case SG_ACT_LEAKY: return pre > 0.f ? 1.f : slope;
    default: return 1.f;
  }
}

// ---------------- InstanceNorm: one wave per plane (small planes) or one block per plane -----------
template <bool WAVE>
__global__ void __launch_bounds__(256) instnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                          float* __restrict__ y, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, int NC, int HW, float eps, int act,
                                                          float slope) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63;
  const int plane = WAVE ? blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
  if (WAVE && plane >= NC) return;
  const int t0 = WAVE ? lane : threadIdx.x, nt = WAVE ? 64 : blockDim.x;
  const float* xp = x + (size_t)plane * HW;
  float s = 0.f;
  for (int i = t0; i < HW; i += nt) s += xp[i];
  s = WAVE ? sg_wave_sum(s) : sg_block_sum(s, red);
  const float mean = s / (float)HW;
  float q = 0.f;
  for (int i = t0; i < HW; i += nt) { const float d = xp[i] - mean; q += d * d; }
  q = WAVE ? sg_wave_sum(q) : sg_block_sum(q, red);
  const float rstd = 1.f / sqrtf(q / (float)HW + eps);
  if (t0 == 0) { mean_o[plane] = mean; rstd_o[plane] = rstd; }
  float* yp = y + (size_t)plane * HW;
  const float* sp = skip ? skip + (size_t)plane * HW : nullptr;
  for (int i = t0; i < HW; i += nt) {
    float v = sg_apply_act((xp[i] - mean) * rstd, act, slope);
    if (sp) v += sp[i];
    yp[i] = v;
  }
}

template <bool WAVE>
__global__ void __launch_bounds__(256) instnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          float* __restrict__ gx, int NC, int HW, int act, float slope) {
  __shared__ float red[16];
  const int lane = threadIdx.x & 63;
  const int plane = WAVE ? blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
  if (WAVE && plane >= NC) return;
  const int t0 = WAVE ? lane : threadIdx.x, nt = WAVE ? 64 : blockDim.x;
  const float* xp = x + (size_t)plane * HW;
  const float* gp = gy + (size_t)plane * HW;
  const float mean = mean_i[plane], rstd = rstd_i[plane];
  float s1 = 0.f, s2 = 0.f;
  for (int i = t0; i < HW; i += nt) {
    const float z = (xp[i] - mean) * rstd;
    const float g = gp[i] * act_grad_from_pre(z, act, slope);
    s1 += g; s2 += g * z;
  }
  if (WAVE) { s1 = sg_wave_sum(s1); s2 = sg_wave_sum(s2); }
  else { s1 = sg_block_sum(s1, red); s2 = sg_block_sum(s2, red); }
  const float inv = 1.f / (float)HW;
  const float m1 = s1 * inv, m2 = s2 * inv;
  float* op = gx + (size_t)plane * HW;
  for (int i = t0; i < HW; i += nt) {
    const float z = (xp[i] - mean) * rstd;
    const float g = gp[i] * act_grad_from_pre(z, act, slope);
    op[i] = rstd * (g - m1 - z * m2);
  }
}

// ---------------- InstanceNorm, register-resident form --------------------------------------------------------------------
// A group of G threads owns one plane and keeps ALL of it in registers (E = ceil(HW / G) <= 64 elements per thread): every load
// of the plane is issued before the first use, the mean / variance are the same two passes as above but over registers, and
// the plane is read from memory exactly once.  (The loops above read it three times, one dependent 4-byte load per lane at a
// time: 0.26 / 0.31 of the HBM peak on the algorithmic bytes.)  G = 16 (planes <= 64 elements: 16 planes per workgroup),
// 64 (one wave per plane), 256 / 1024 (one workgroup per plane).
template <int G>
__device__ __forceinline__ float group_sum(float v, float* red) {
  if constexpr (G == 16) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  } else if constexpr (G == 64) {
    return sg_wave_sum(v);
  } else {
    return sg_block_sum(v, red);
  }
}

template <int G, int E>
__global__ void __launch_bounds__(G > 256 ? G : 256) instnorm_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                                            float* __restrict__ y, float* __restrict__ mean_o,
                                                                            float* __restrict__ rstd_o, int NC, int HW, float eps,
                                                                            int act, float slope) {
  __shared__ float red[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;          // planes per workgroup
  const int tg = threadIdx.x % G;
  const int plane = blockIdx.x * PPB + threadIdx.x / G;
  const bool live = plane < NC;                        // (whole groups: no divergence inside a reduction)
  const size_t base = (size_t)(live ? plane : 0) * HW;
  float v[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tg + G * j;
    v[j] = (live && i < HW) ? x[base + i] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < E; ++j) s += v[j];
  s = group_sum<G>(s, red);
  const float mean = s / (float)HW;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const float d = v[j] - mean;
    q += (tg + G * j < HW) ? d * d : 0.f;
  }
  q = group_sum<G>(q, red);
  const float rstd = 1.f / sqrtf(q / (float)HW + eps);
  if (!live) return;
  if (tg == 0) { mean_o[plane] = mean; rstd_o[plane] = rstd; }
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tg + G * j;
    if (i < HW) {
      float o = sg_apply_act((v[j] - mean) * rstd, act, slope);
      if (skip) o += skip[base + i];
      y[base + i] = o;
    }
  }
}

template <int G, int E>
__global__ void __launch_bounds__(G > 256 ? G : 256) instnorm_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                            const float* __restrict__ mean_i,
                                                                            const float* __restrict__ rstd_i, float* __restrict__ gx,
                                                                            int NC, int HW, int act, float slope) {
  __shared__ float red[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;
  const int tg = threadIdx.x % G;
  const int plane = blockIdx.x * PPB + threadIdx.x / G;
  const bool live = plane < NC;
  const size_t base = (size_t)(live ? plane : 0) * HW;
  const float mean = mean_i[live ? plane : 0], rstd = rstd_i[live ? plane : 0];
  float z[E], g[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tg + G * j;
    const bool ok = live && i < HW;
    z[j] = ok ? x[base + i] : mean;
    g[j] = ok ? gy[base + i] : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    z[j] = (z[j] - mean) * rstd;
    g[j] *= act_grad_from_pre(z[j], act, slope);
    s1 += g[j]; s2 += g[j] * z[j];
  }
  s1 = group_sum<G>(s1, red);
  s2 = group_sum<G>(s2, red);
  if (!live) return;
  const float inv = 1.f / (float)HW;
  const float m1 = s1 * inv, m2 = s2 * inv;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int i = tg + G * j;
    if (i < HW) gx[base + i] = rstd * (g[j] - m1 - z[j] * m2);
  }
}

// ---------------- InstanceNorm, register-resident form with 16-byte accesses -------------------------------------------------------
// Same ownership as above (G threads per plane, the whole plane in registers), but every lane moves float4s: lane tg holds elements
// 4 (tg + G j) .. + 3.  The scalar form issues one 256-byte request per wave and load -- measured at 0.41 of the HBM peak on its
// algorithmic bytes (3.3 TB/s, 43 + 51 launches of ~50-60 MB per step); 16-byte accesses are what the memory pipeline is built for
// (MI355X_MICROARCH.md: 8-byte accesses run at 0.54-0.70 of the 16-byte rate, 4-byte ones below that).  Needs HW % 4 == 0 and
// 16-byte aligned operands (every plane then starts on a 16-byte boundary); E4 = float4s per thread.
typedef float in_f4 __attribute__((ext_vector_type(4)));

template <int G, int E4>
__global__ void __launch_bounds__(G > 256 ? G : 256) instnorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                                            float* __restrict__ y, float* __restrict__ mean_o,
                                                                            float* __restrict__ rstd_o, int NC, int HW, float eps,
                                                                            int act, float slope) {
  __shared__ float red[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;
  const int tg = threadIdx.x % G;
  const int plane = blockIdx.x * PPB + threadIdx.x / G;
  const bool live = plane < NC;
  const int HW4 = HW >> 2;
  const in_f4* xp = reinterpret_cast<const in_f4*>(x + (size_t)(live ? plane : 0) * HW);
  in_f4 v[E4];
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    const int i = tg + G * j;
    v[j] = (live && i < HW4) ? xp[i] : in_f4{0.f, 0.f, 0.f, 0.f};
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < E4; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  s = group_sum<G>(s, red);
  const float mean = s / (float)HW;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += (tg + G * j < HW4) ? (a * a + b * b) + (c * c + d * d) : 0.f;
  }
  q = group_sum<G>(q, red);
  const float rstd = 1.f / sqrtf(q / (float)HW + eps);
  if (!live) return;
  if (tg == 0) { mean_o[plane] = mean; rstd_o[plane] = rstd; }
  const in_f4* sp = skip ? reinterpret_cast<const in_f4*>(skip + (size_t)plane * HW) : nullptr;
  in_f4* yp = reinterpret_cast<in_f4*>(y + (size_t)plane * HW);
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    const int i = tg + G * j;
    if (i < HW4) {
      in_f4 o;
      o.x = sg_apply_act((v[j].x - mean) * rstd, act, slope);
      o.y = sg_apply_act((v[j].y - mean) * rstd, act, slope);
      o.z = sg_apply_act((v[j].z - mean) * rstd, act, slope);
      o.w = sg_apply_act((v[j].w - mean) * rstd, act, slope);
      if (sp) o += sp[i];
      yp[i] = o;
    }
  }
}

template <int G, int E4>
__global__ void __launch_bounds__(G > 256 ? G : 256) instnorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                            const float* __restrict__ mean_i,
                                                                            const float* __restrict__ rstd_i, float* __restrict__ gx,
                                                                            int NC, int HW, int act, float slope) {
  __shared__ float red[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;
  const int tg = threadIdx.x % G;
  const int plane = blockIdx.x * PPB + threadIdx.x / G;
  const bool live = plane < NC;
  const int HW4 = HW >> 2;
  const size_t base = (size_t)(live ? plane : 0) * HW;
  const in_f4* xp = reinterpret_cast<const in_f4*>(x + base);
  const in_f4* gp = reinterpret_cast<const in_f4*>(gy + base);
  const float mean = mean_i[live ? plane : 0], rstd = rstd_i[live ? plane : 0];
  in_f4 z[E4], g[E4];
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    const int i = tg + G * j;
    const bool ok = live && i < HW4;
    z[j] = ok ? xp[i] : in_f4{mean, mean, mean, mean};
    g[j] = ok ? gp[i] : in_f4{0.f, 0.f, 0.f, 0.f};
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    z[j] = (z[j] - mean) * rstd;
    g[j].x *= act_grad_from_pre(z[j].x, act, slope);
    g[j].y *= act_grad_from_pre(z[j].y, act, slope);
    g[j].z *= act_grad_from_pre(z[j].z, act, slope);
    g[j].w *= act_grad_from_pre(z[j].w, act, slope);
    s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
    s2 += (g[j].x * z[j].x + g[j].y * z[j].y) + (g[j].z * z[j].z + g[j].w * z[j].w);
  }
  s1 = group_sum<G>(s1, red);
  s2 = group_sum<G>(s2, red);
  if (!live) return;
  const float inv = 1.f / (float)HW;
  const float m1 = s1 * inv, m2 = s2 * inv;
  in_f4* op = reinterpret_cast<in_f4*>(gx + base);
#pragma unroll
  for (int j = 0; j < E4; ++j) {
    const int i = tg + G * j;
    if (i < HW4) op[i] = rstd * (g[j] - m1 - z[j] * m2);
  }
}

// ---------------- InstanceNorm, large planes (beyond the register budget of 1024 threads: 256 x 256 at configs[3]) ---------------
// One workgroup of 1024 threads per plane, float4 streams with four loads in flight per thread; the plane (256 KB at 256^2, plus
// gy in the backward) stays in the XCD's L2 between the passes, so the extra passes are L2 reads.  The register kernels above
// would need 64 (forward) / 2 x 64 (backward) values per thread there: their <1024, 64> / <1024, 32> instantiations spilled to
// scratch (round 4: 156 / 72 / 876 B), and the backward fell to the one-load-at-a-time loop (2 x 288 us per step at configs[3]).
__global__ void __launch_bounds__(1024) instnorm_fwd_big_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                               float* __restrict__ y, float* __restrict__ mean_o,
                                                               float* __restrict__ rstd_o, int HW, float eps, int act, float slope) {
  __shared__ float red[16];
  const int plane = blockIdx.x, n4 = HW >> 2, tid = threadIdx.x;
  const float4* xp = reinterpret_cast<const float4*>(x + (size_t)plane * HW);
  float s = 0.f;
  for (int i0 = tid; i0 < n4; i0 += 4096) {
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = i0 + e * 1024 < n4 ? xp[i0 + e * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
  }
  s = sg_block_sum(s, red);
  const float mean = s / (float)HW;
  float q = 0.f;
  for (int i0 = tid; i0 < n4; i0 += 4096) {
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = i0 + e * 1024 < n4 ? xp[i0 + e * 1024] : make_float4(mean, mean, mean, mean);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = v[e].x - mean, b = v[e].y - mean, c = v[e].z - mean, d = v[e].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  q = sg_block_sum(q, red);
  const float rstd = 1.f / sqrtf(q / (float)HW + eps);
  if (tid == 0) { mean_o[plane] = mean; rstd_o[plane] = rstd; }
  float4* yp = reinterpret_cast<float4*>(y + (size_t)plane * HW);
  const float4* sp = skip ? reinterpret_cast<const float4*>(skip + (size_t)plane * HW) : nullptr;
  for (int i0 = tid; i0 < n4; i0 += 4096) {
    float4 v[4], k[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = i0 + e * 1024 < n4;
      v[e] = ok ? xp[i0 + e * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
      k[e] = (ok && sp) ? sp[i0 + e * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i0 + e * 1024 >= n4) continue;
      float4 o;
      o.x = sg_apply_act((v[e].x - mean) * rstd, act, slope) + k[e].x;
      o.y = sg_apply_act((v[e].y - mean) * rstd, act, slope) + k[e].y;
      o.z = sg_apply_act((v[e].z - mean) * rstd, act, slope) + k[e].z;
      o.w = sg_apply_act((v[e].w - mean) * rstd, act, slope) + k[e].w;
      yp[i0 + e * 1024] = o;
    }
  }
}

__global__ void __launch_bounds__(1024) instnorm_bwd_big_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                               const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                               float* __restrict__ gx, int HW, int act, float slope) {
  __shared__ float red[16];
  const int plane = blockIdx.x, n4 = HW >> 2, tid = threadIdx.x;
  const float4* xp = reinterpret_cast<const float4*>(x + (size_t)plane * HW);
  const float4* gp = reinterpret_cast<const float4*>(gy + (size_t)plane * HW);
  const float mean = mean_i[plane], rstd = rstd_i[plane];
  float s1 = 0.f, s2 = 0.f;
  for (int i0 = tid; i0 < n4; i0 += 2048) {
    float4 xv[2], gv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool ok = i0 + e * 1024 < n4;
      xv[e] = ok ? xp[i0 + e * 1024] : make_float4(mean, mean, mean, mean);
      gv[e] = ok ? gp[i0 + e * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float zx = (xv[e].x - mean) * rstd, zy = (xv[e].y - mean) * rstd, zz = (xv[e].z - mean) * rstd, zw = (xv[e].w - mean) * rstd;
      const float a = gv[e].x * act_grad_from_pre(zx, act, slope), b = gv[e].y * act_grad_from_pre(zy, act, slope);
      const float c = gv[e].z * act_grad_from_pre(zz, act, slope), d = gv[e].w * act_grad_from_pre(zw, act, slope);
      s1 += (a + b) + (c + d);
      s2 += (a * zx + b * zy) + (c * zz + d * zw);
    }
  }
  s1 = sg_block_sum(s1, red);
  s2 = sg_block_sum(s2, red);
  const float inv = 1.f / (float)HW;
  const float m1 = s1 * inv, m2 = s2 * inv;
  float4* op = reinterpret_cast<float4*>(gx + (size_t)plane * HW);
  for (int i0 = tid; i0 < n4; i0 += 2048) {
    float4 xv[2], gv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool ok = i0 + e * 1024 < n4;
      xv[e] = ok ? xp[i0 + e * 1024] : make_float4(mean, mean, mean, mean);
      gv[e] = ok ? gp[i0 + e * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (i0 + e * 1024 >= n4) continue;
      const float zx = (xv[e].x - mean) * rstd, zy = (xv[e].y - mean) * rstd, zz = (xv[e].z - mean) * rstd, zw = (xv[e].w - mean) * rstd;
      float4 o;
      o.x = rstd * (gv[e].x * act_grad_from_pre(zx, act, slope) - m1 - zx * m2);
      o.y = rstd * (gv[e].y * act_grad_from_pre(zy, act, slope) - m1 - zy * m2);
      o.z = rstd * (gv[e].z * act_grad_from_pre(zz, act, slope) - m1 - zz * m2);
      o.w = rstd * (gv[e].w * act_grad_from_pre(zw, act, slope) - m1 - zw * m2);
      op[i0 + e * 1024] = o;
    }
  }
}
inline bool instnorm_big_ok(int HW, const void* a, const void* b, const void* c, const void* d) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                      reinterpret_cast<uintptr_t>(d);
  return HW % 4 == 0 && HW >= 4096 && (u & 15) == 0;
}

// (G, E) for a plane of HW elements, or G = 0 when it does not fit 64 registers per thread
inline void instnorm_shape(int HW, int maxE, int& G, int& E) {
  // (a workgroup of 1024 threads has 128 registers per thread: half the values per thread of the smaller groups, or it spills)
  G = HW <= 64 ? 16 : (HW <= 1024 ? 64 : (HW <= 256 * maxE ? 256 : (HW <= 1024 * (maxE / 2) ? 1024 : 0)));
  if (G == 0) { E = 0; return; }
  const int e = (HW + G - 1) / G;
  E = e <= 1 ? 1 : (e <= 2 ? 2 : (e <= 4 ? 4 : (e <= 8 ? 8 : (e <= 16 ? 16 : (e <= 32 ? 32 : 64)))));
}

#define SG_IN_CASE(KERNEL, Gv, Ev, ...)                                                                              \
  hipLaunchKernelGGL((KERNEL<Gv, Ev>), dim3(sg_cdiv(NC, Gv >= 256 ? 1 : 256 / Gv)), dim3(Gv > 256 ? Gv : 256), 0, s, __VA_ARGS__)
#define SG_IN_DISPATCH_E(KERNEL, Gv, ...)                                \
  switch (E) {                                                           \
    case 1: SG_IN_CASE(KERNEL, Gv, 1, __VA_ARGS__); break;               \
    case 2: SG_IN_CASE(KERNEL, Gv, 2, __VA_ARGS__); break;               \
    case 4: SG_IN_CASE(KERNEL, Gv, 4, __VA_ARGS__); break;               \
    case 8: SG_IN_CASE(KERNEL, Gv, 8, __VA_ARGS__); break;               \
    case 16: SG_IN_CASE(KERNEL, Gv, 16, __VA_ARGS__); break;             \
    case 32: SG_IN_CASE(KERNEL, Gv, 32, __VA_ARGS__); break;             \
    default: SG_IN_CASE(KERNEL, Gv, 64, __VA_ARGS__); break;             \
  }
// (1024-thread groups: only the instantiations that fit 128 registers per thread exist -- TOP = 32 values forward, 16 backward)
#define SG_IN_DISPATCH_E1024(KERNEL, TOP, ...)                           \
  switch (E) {                                                           \
    case 1: SG_IN_CASE(KERNEL, 1024, 1, __VA_ARGS__); break;             \
    case 2: SG_IN_CASE(KERNEL, 1024, 2, __VA_ARGS__); break;             \
    case 4: SG_IN_CASE(KERNEL, 1024, 4, __VA_ARGS__); break;             \
    case 8: SG_IN_CASE(KERNEL, 1024, 8, __VA_ARGS__); break;             \
    case 16: SG_IN_CASE(KERNEL, 1024, 16, __VA_ARGS__); break;           \
    default: SG_IN_CASE(KERNEL, 1024, TOP, __VA_ARGS__); break;          \
  }
#define SG_IN_DISPATCH(KERNEL, TOP, ...)                                  \
  switch (G) {                                                            \
    case 16: SG_IN_DISPATCH_E(KERNEL, 16, __VA_ARGS__) break;             \
    case 64: SG_IN_DISPATCH_E(KERNEL, 64, __VA_ARGS__) break;             \
    case 256: SG_IN_DISPATCH_E(KERNEL, 256, __VA_ARGS__) break;           \
    default: SG_IN_DISPATCH_E1024(KERNEL, TOP, __VA_ARGS__) break;        \
  }

// vector form: (G, E4) for a plane of HW = 4 HW4 elements; max4 = float4s per thread the instantiations allow
inline void instnorm_shape_vec(int HW, int max4, int& G, int& E4) {
  const int HW4 = HW / 4;
  G = HW4 <= 16 ? 16 : (HW4 <= 64 * 4 ? 64 : (HW4 <= 256 * max4 ? 256 : (HW4 <= 1024 * (max4 / 2) ? 1024 : 0)));
  if (HW % 4 != 0 || G == 0) { G = 0; E4 = 0; return; }
  const int e = (HW4 + G - 1) / G;
  E4 = e <= 1 ? 1 : (e <= 2 ? 2 : (e <= 4 ? 4 : (e <= 8 ? 8 : 16)));
}
#define SG_INV_CASE(KERNEL, Gv, Ev, ...)                                                                              \
  hipLaunchKernelGGL((KERNEL<Gv, Ev>), dim3(sg_cdiv(NC, Gv >= 256 ? 1 : 256 / Gv)), dim3(Gv > 256 ? Gv : 256), 0, s, __VA_ARGS__)
// every (G, E4) pair instnorm_shape_vec can return for the given max4 -- and no other instantiation (a 1024-thread group has 128
// registers per thread: 8 float4s of plane data forward, 2 x 4 backward)
#define SG_INV_ROW(KERNEL, Gv, ...)                                        \
  switch (E4) {                                                            \
    case 1: SG_INV_CASE(KERNEL, Gv, 1, __VA_ARGS__); break;                \
    case 2: SG_INV_CASE(KERNEL, Gv, 2, __VA_ARGS__); break;                \
    default: SG_INV_CASE(KERNEL, Gv, 4, __VA_ARGS__); break;               \
  }
inline bool instnorm_vec_ok(int HW, const void* a, const void* b, const void* c, const void* d) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                      reinterpret_cast<uintptr_t>(d);
  return HW % 4 == 0 && (u & 15) == 0;
}

// ---------------- BatchNorm2d: one block per channel ---------------------------------------------
// ---- BatchNorm2d (+ fused activation), three stages so that a 64..256-channel layer fills the chip ---------------------
//  stats : grid (C, S)  -- slice z of channel c covers a contiguous range of the (n, p) index space
//  final : one thread per channel combines the S partials in fixed order (Chan's parallel-variance update)
//  apply : elementwise over every (n, c) plane
// (the first version ran ONE workgroup per channel with three passes: ~1 TB/s at 128 channels)
struct BnIter {             // walks i -> (n, p) without a division per element
  int n, p, HW;
  __device__ __forceinline__ BnIter(long i, int HW_) : HW(HW_) { n = (int)(i / HW_); p = (int)(i - (long)n * HW_); }
  __device__ __forceinline__ void step(int d) { p += d; while (p >= HW) { p -= HW; ++n; } }
};

// slices of up to BN_REG * 256 elements stay in registers between the mean and the variance pass: x is read ONCE (the two-pass
// form read every slice twice; slices are ~1-10 K elements: bn_slices)
constexpr int BN_REG = 40;
// ``fin.counter`` != nullptr: the slice that arrives last for its channel combines the channel's S partials itself (the arithmetic
// of bn_final_kernel, same order) -- no second launch (common.h: sg_arrive_last)
struct BnFinal {
  int* counter; float* save_mean; float* save_rstd; float* rmean; float* rvar; int64_t* nbt; float eps, momentum;
};
__device__ __forceinline__ void bn_combine(const float* __restrict__ part, int c, int S, bool coherent, float& n, float& mean, float& m2) {
  n = 0.f; mean = 0.f; m2 = 0.f;
  // (the partials of 8 slices are fetched together: one thread walking S <= 64 dependent round trips took 10 us)
  for (int z0 = 0; z0 < S; z0 += 8) {
    float pn[8], pm[8], pq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* p = part + ((size_t)c * S + min(z0 + e, S - 1)) * 3;
      if (coherent) { pn[e] = sg_consume(p); pm[e] = sg_consume(p + 1); pq[e] = sg_consume(p + 2); }
      else { pn[e] = p[0]; pm[e] = p[1]; pq[e] = p[2]; }
      if (z0 + e >= S) pn[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float nb = pn[e];
      if (nb <= 0.f) continue;
      const float d = pm[e] - mean, nt = n + nb;
      mean += d * (nb / nt);
      m2 += pq[e] + d * d * (n * nb / nt);
      n = nt;
    }
  }
}
__device__ __forceinline__ void bn_write_stats(int c, float n, float mean, float m2, float* save_mean, float* save_rstd, float* rmean,
                                               float* rvar, int64_t* nbt, float eps, float momentum) {
  const float var = m2 / n;
  save_mean[c] = mean;
  save_rstd[c] = 1.f / sqrtf(var + eps);
  if (rmean) {
    const float unb = n > 1.f ? m2 / (n - 1.f) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
  if (nbt && c == 0) nbt[0] += 1;
}
__device__ __forceinline__ void bn_stats_finish(float* __restrict__ part, int c, int z, int S, float m, float mean, float q,
                                                const BnFinal& fin, int* flag) {
  float* o = part + ((size_t)c * S + z) * 3;
  if (fin.counter == nullptr) {
    if (threadIdx.x == 0) { o[0] = m; o[1] = mean; o[2] = q; }
    return;
  }
  if (threadIdx.x == 0) { sg_publish(o, m); sg_publish(o + 1, mean); sg_publish(o + 2, q); }
  if (!sg_arrive_last(fin.counter + c, S, flag)) return;
  if (threadIdx.x == 0) {
    float n, mu, m2;
    bn_combine(part, c, S, true, n, mu, m2);
    bn_write_stats(c, n, mu, m2, fin.save_mean, fin.save_rstd, fin.rmean, fin.rvar, fin.nbt, fin.eps, fin.momentum);
  }
}
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int N, int C,
                                                      int HW, int S, BnFinal fin) {
  __shared__ float red[16];
  __shared__ int lastflag;
  const int c = blockIdx.x, z = blockIdx.y;
  const long cnt = (long)N * HW;
  const long chunk = (cnt + S - 1) / S;
  const long beg = z * chunk, end = beg + chunk < cnt ? beg + chunk : cnt;
  const float m = (float)(end > beg ? end - beg : 0);
  if (end - beg <= (long)BN_REG * 256) {
    float v[BN_REG];
    float s = 0.f;
    {
      BnIter it(beg + threadIdx.x, HW);
#pragma unroll
      for (int e = 0; e < BN_REG; ++e) {
        const bool ok = beg + threadIdx.x + (long)e * 256 < end;
        v[e] = ok ? x[((size_t)it.n * C + c) * HW + it.p] : 0.f;
        s += v[e];
        if (ok) it.step(256);
      }
    }
    s = sg_block_sum(s, red);
    const float mean = m > 0.f ? s / m : 0.f;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < BN_REG; ++e) {
      const float d = v[e] - mean;
      q += (beg + threadIdx.x + (long)e * 256 < end) ? d * d : 0.f;
    }
    q = sg_block_sum(q, red);
    bn_stats_finish(part, c, z, S, m, mean, q, fin, &lastflag);
    return;
  }
  float s = 0.f;
  {
    BnIter it(beg + threadIdx.x, HW);
    for (long i = beg + threadIdx.x; i < end; i += 256, it.step(256)) s += x[((size_t)it.n * C + c) * HW + it.p];
  }
  s = sg_block_sum(s, red);
  const float mean = m > 0.f ? s / m : 0.f;
  float q = 0.f;
  {
    BnIter it(beg + threadIdx.x, HW);
    for (long i = beg + threadIdx.x; i < end; i += 256, it.step(256)) {
      const float d = x[((size_t)it.n * C + c) * HW + it.p] - mean;
      q += d * d;
    }
  }
  q = sg_block_sum(q, red);
  bn_stats_finish(part, c, z, S, m, mean, q, fin, &lastflag);
}

__global__ void bn_final_kernel(const float* __restrict__ part, float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt, int C, int S,
                                float eps, float momentum, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (!training) {
    save_mean[c] = rmean[c];
    save_rstd[c] = 1.f / sqrtf(rvar[c] + eps);
    return;
  }
  float n, mean, m2;
  bn_combine(part, c, S, false, n, mean, m2);
  bn_write_stats(c, n, mean, m2, save_mean, save_rstd, rmean, rvar, nbt, eps, momentum);
}

// y = act((x - mean[c]) * rstd[c] * gamma[c] + beta[c]); one workgroup = 1024 consecutive elements of one plane
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ y, int C, int HW,
                                                      int act, float slope) {
  const int plane = blockIdx.y, c = plane % C;
  const float mu = mean[c], a = rstd[c] * (gamma ? gamma[c] : 1.f), b = beta ? beta[c] : 0.f;
  const size_t base = (size_t)plane * HW;
  const int p0 = blockIdx.x * 1024 + threadIdx.x * 4;
  if ((HW & 3) == 0) {
    if (p0 < HW) {
      float4 v = *reinterpret_cast<const float4*>(x + base + p0);
      v.x = sg_apply_act((v.x - mu) * a + b, act, slope); v.y = sg_apply_act((v.y - mu) * a + b, act, slope);
      v.z = sg_apply_act((v.z - mu) * a + b, act, slope); v.w = sg_apply_act((v.w - mu) * a + b, act, slope);
      *reinterpret_cast<float4*>(y + base + p0) = v;
    }
  } else {
    for (int e = 0; e < 4; ++e) {
      const int p = blockIdx.x * 1024 + e * 256 + threadIdx.x;
      if (p < HW) y[base + p] = sg_apply_act((x[base + p] - mu) * a + b, act, slope);
    }
  }
}

__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ part, int N, int C, int HW, int S, int act,
                                                          float slope, int* __restrict__ counter, float* __restrict__ sums,
                                                          float* __restrict__ ggamma, float* __restrict__ gbeta) {
  __shared__ float red[16];
  __shared__ int lastflag;
  const int c = blockIdx.x, z = blockIdx.y;
  const long cnt = (long)N * HW;
  const long chunk = (cnt + S - 1) / S;
  const long beg = z * chunk, end = beg + chunk < cnt ? beg + chunk : cnt;
  const float mu = mean[c], rs = rstd[c], ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  BnIter it(beg + threadIdx.x, HW);
  // four (x, gy) pairs are fetched before the first is used -- same sums in the same order; the one-pair-per-iteration loop waited
  // for every load in turn (4..40 dependent round trips per thread)
  for (long i = beg + threadIdx.x; i < end; i += 1024) {
    float xv[4], gv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = i + (long)e * 256 < end;
      const size_t off = ((size_t)it.n * C + c) * HW + it.p;
      xv[e] = ok ? x[off] : 0.f;
      gv[e] = ok ? gy[off] : 0.f;
      if (ok) it.step(256);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (i + (long)e * 256 < end) {
        const float zv = (xv[e] - mu) * rs;
        const float g = gv[e] * act_grad_from_pre(zv * ga + be, act, slope);
        s1 += g; s2 += g * zv;
      }
    }
  }
  s1 = sg_block_sum(s1, red);
  s2 = sg_block_sum(s2, red);
  float* o = part + ((size_t)c * S + z) * 2;
  if (counter == nullptr) {                        // bn_bwd_final_kernel follows
    if (threadIdx.x == 0) { o[0] = s1; o[1] = s2; }
    return;
  }
  if (threadIdx.x == 0) { sg_publish(o, s1); sg_publish(o + 1, s2); }
  if (!sg_arrive_last(counter + c, S, &lastflag)) return;
  if (threadIdx.x == 0) {                          // the channel's last slice: bn_bwd_final_kernel's sums, same order
    float t1 = 0.f, t2 = 0.f;
    for (int z0 = 0; z0 < S; z0 += 8) {
      float a[8], b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const size_t i = ((size_t)c * S + min(z0 + e, S - 1)) * 2;
        a[e] = sg_consume(part + i); b[e] = sg_consume(part + i + 1);
        if (z0 + e >= S) { a[e] = 0.f; b[e] = 0.f; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { t1 += a[e]; t2 += b[e]; }
    }
    sums[2 * c] = t1; sums[2 * c + 1] = t2;
    if (gbeta) gbeta[c] = t1;
    if (ggamma) ggamma[c] = t2;
  }
}

__global__ void bn_bwd_final_kernel(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ ggamma,
                                    float* __restrict__ gbeta, int C, int S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int z0 = 0; z0 < S; z0 += 8) {
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const size_t i = ((size_t)c * S + min(z0 + e, S - 1)) * 2;
      a[e] = z0 + e < S ? part[i] : 0.f; b[e] = z0 + e < S ? part[i + 1] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1 += a[e]; s2 += b[e]; }
  }
  sums[2 * c] = s1; sums[2 * c + 1] = s2;
  if (gbeta) gbeta[c] = s1;
  if (ggamma) ggamma[c] = s2;
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ sums, float* __restrict__ gx, int C,
                                                          int HW, float inv_cnt, int training, int act, float slope) {
  const int plane = blockIdx.y, c = plane % C;
  const float mu = mean[c], rs = rstd[c], ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  // eval mode: the statistics are constants, so only the affine map is differentiated
  const float m1 = training ? sums[2 * c] * inv_cnt : 0.f, m2 = training ? sums[2 * c + 1] * inv_cnt : 0.f;
  const size_t base = (size_t)plane * HW;
  for (int e = 0; e < 4; ++e) {
    const int p = blockIdx.x * 1024 + e * 256 + threadIdx.x;
    if (p < HW) {
      const float zv = (x[base + p] - mu) * rs;
      const float g = gy[base + p] * act_grad_from_pre(zv * ga + be, act, slope);
      gx[base + p] = ga * rs * (g - m1 - zv * m2);
    }
  }
}

// small planes (BatchNorm1d: HW == 1): one thread per element
__global__ void bn_apply_flat_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ y,
                                     size_t total, int C, int HW, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / HW) % C);
  y[i] = sg_apply_act((x[i] - mean[c]) * rstd[c] * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f), act, slope);
}
__global__ void bn_bwd_apply_flat_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ sums, float* __restrict__ gx, size_t total, int C, int HW,
                                         float inv_cnt, int training, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / HW) % C);
  const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f, rs = rstd[c];
  const float m1 = training ? sums[2 * c] * inv_cnt : 0.f, m2 = training ? sums[2 * c + 1] * inv_cnt : 0.f;
  const float zv = (x[i] - mean[c]) * rs;
  const float g = gy[i] * act_grad_from_pre(zv * ga + be, act, slope);
  gx[i] = ga * rs * (g - m1 - zv * m2);
}

inline int bn_slices(int N, int C, int HW) {
  long cnt = (long)N * HW;
  // ~4096 workgroups per launch: the per-thread loops are chains of dependent loads, so the latency is hidden by running many
  // short slices side by side (1024 workgroups of ~136 elements per thread ran at 0.2 of the HBM peak)
  const int target = sg_opt(SG_OPT_BN_BLOCKS) > 0 ? sg_opt(SG_OPT_BN_BLOCKS) : 4096;
  int S = (target + C - 1) / C;
  const long maxS = cnt / 1024 > 0 ? cnt / 1024 : 1;      // at least 1024 elements per slice
  if (S > maxS) S = (int)maxS;
  // ... and few enough per slice for the register-resident statistics kernel (BN_REG x 256 elements: x is read ONCE).  At
  // configs[4] -- 1056 crops, 64 x 31 x 31 after the first conv -- 64 slices held 15.9 K elements each and bn_stats_kernel fell to
  // its two-pass loop of dependent loads: 85 us per launch at 3 TB/s (round 6)
  const long need = (cnt + (long)BN_REG * 256 - 1) / ((long)BN_REG * 256);
  if (S < need && need <= 512) S = (int)need;
  return S < 1 ? 1 : S;
}

// per-channel sum over (N, HW): two deterministic stages -- grid (C, S) partial sums over contiguous chunks of the
// (n, p) index space, then one block per channel adds the S partials in fixed order.
__global__ void __launch_bounds__(256) channel_sum_partial_kernel(const float* __restrict__ g, float* __restrict__ part, int N,
                                                                 int C, int HW, int S, int* __restrict__ counter,
                                                                 float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ int lastflag;
  const int c = blockIdx.x, z = blockIdx.y;
  const long cnt = (long)N * HW;
  const long chunk = (cnt + S - 1) / S;
  const long beg = z * chunk, end = beg + chunk < cnt ? beg + chunk : cnt;
  float s = 0.f;
  // four loads in flight per thread, added in index order (round 6: the one-load-per-iteration loops parked 90 % of the wave
  // cycles in s_waitcnt -- the same sum, the same order)
  if (HW >= 256) {
    BnIter it(beg + threadIdx.x, HW);          // (a 64-bit division per element made this kernel compute-bound)
    for (long i = beg + threadIdx.x; i < end; i += 1024) {
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] = i + 256 * e < end ? g[((size_t)it.n * C + c) * HW + it.p] : 0.f;
        it.step(256);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + 256 * e < end) s += t[e];
    }
  } else {
    for (long i = beg + threadIdx.x; i < end; i += 1024) {      // cnt < 2^31: 32-bit division
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long ie = i + 256 * e;
        const unsigned n = (unsigned)ie / (unsigned)HW, p = (unsigned)ie - n * (unsigned)HW;
        t[e] = ie < end ? g[((size_t)n * C + c) * HW + p] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + 256 * e < end) s += t[e];
    }
  }
  s = sg_block_sum(s, red);
  if (counter == nullptr) {                        // channel_sum_final_kernel follows
    if (threadIdx.x == 0) part[(size_t)c * S + z] = s;
    return;
  }
  if (threadIdx.x == 0) sg_publish(&part[(size_t)c * S + z], s);
  if (!sg_arrive_last(counter + c, S, &lastflag)) return;
  if (threadIdx.x == 0) {                          // the channel's last slice adds the S partials in slice order
    float t = 0.f;
    for (int z0 = 0; z0 < S; z0 += 8) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = sg_consume(&part[(size_t)c * S + min(z0 + e, S - 1)]);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (z0 + e < S) t += a[e];
    }
    out[c] = t;
  }
}

__global__ void channel_sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  out[c] = sg_sum_strided(part + (size_t)c * S, 1, S);       // (eight partials in flight, added in order)
}

__global__ void channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int C, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int cnt = N * HW;
  float s = 0.f;
  // (small tensors: HW is often 1..64, where stepping an (n, p) pair by 256 costs more than a 32-bit division -- measured.
  //  Eight loads are in flight before the first add: same sum, same order, an eighth of the dependent round trips)
  for (int i0 = threadIdx.x; i0 < cnt; i0 += 8 * blockDim.x) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = min(i0 + e * (int)blockDim.x, cnt - 1);           // (clamped: an unconditional load; the tail is not added)
      const int n = i / HW, p = i - n * HW;
      t[e] = g[((size_t)n * C + c) * HW + p];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (i0 + e * (int)blockDim.x < cnt) s += t[e];
  }
  s = sg_block_sum(s, red);
  if (threadIdx.x == 0) out[c] = s;
}

// ---------------- pooling --------------------------------------------------------------------------
__global__ void avgpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W, int OH,
                                      int OW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)NC * OH * OW;
  if (i >= total) return;
  const int ow = i % OW;
  const int oh = (i / OW) % OH;
  const size_t nc = i / ((size_t)OW * OH);
  const float* xp = x + nc * H * W;
  float s = 0.f;
  int cnt = 0;
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if (ih < 0 || ih >= H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if (iw < 0 || iw >= W) continue;
      s += xp[ih * W + iw];
      ++cnt;
    }
  }
  y[i] = s / (float)cnt;
}

__device__ __forceinline__ int pool_cnt(int o, int L) {   // valid taps of window o along one axis
  const int lo = o * 2 - 1 < 0 ? 0 : o * 2 - 1, hi = o * 2 + 1 >= L ? L - 1 : o * 2 + 1;
  return hi - lo + 1;
}

__global__ void avgpool3s2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int NC, int H, int W, int OH,
                                      int OW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)NC * H * W;
  if (i >= total) return;
  const int iw = i % W;
  const int ih = (i / W) % H;
  const size_t nc = i / ((size_t)W * H);
  const float* gp = gy + nc * OH * OW;
  float s = 0.f;
  for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
    if (oh >= OH) continue;
    for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
      if (ow >= OW) continue;
      s += gp[oh * OW + ow] / (float)(pool_cnt(oh, H) * pool_cnt(ow, W));
    }
  }
  gx[i] = s;
}

// nn.MaxPool2d(kernel_size=2, stride=2) of VGG19 (losses.py:183-198 slices torchvision's vgg19.features): floor mode, first
// maximum of the window wins (row-major scan with a strict '>' like ATen), NaN propagates.  One thread per output pixel;
// even widths read / write float2 rows.
__device__ __forceinline__ int maxpool_argmax(float a, float b, float c, float d) {
  int k = 0; float m = a;
  if (b > m || b != b) { m = b; k = 1; }
  if (c > m || c != c) { m = c; k = 2; }
  if (d > m || d != d) { m = d; k = 3; }
  return k;
}
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W, int OH, int OW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)NC * OH * OW) return;
  const int ow = i % OW;
  const int oh = (i / OW) % OH;
  const size_t nc = i / ((size_t)OW * OH);
  const float* p = x + (nc * H + 2 * oh) * W + 2 * ow;
  float a, b, c, d;
  if ((W & 1) == 0) {
    const float2 r0 = *reinterpret_cast<const float2*>(p), r1 = *reinterpret_cast<const float2*>(p + W);
    a = r0.x; b = r0.y; c = r1.x; d = r1.y;
  } else {
    a = p[0]; b = p[1]; c = p[W]; d = p[W + 1];
  }
  const int k = maxpool_argmax(a, b, c, d);
  y[i] = k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d));
}
// gx = gy routed to the window's arg-max (recomputed from x); rows / columns no window covers (odd sizes) get zero
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, int NC,
                                    int H, int W, int OH, int OW) {
  const int CW = (W + 1) / 2, CH = (H + 1) / 2;            // cells incl. the uncovered tail
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)NC * CH * CW) return;
  const int cw = i % CW;
  const int ch = (i / CW) % CH;
  const size_t nc = i / ((size_t)CW * CH);
  const size_t base = (nc * H + 2 * ch) * W + 2 * cw;
  if (ch >= OH || cw >= OW) {                              // tail cell: up to 2x2 inputs outside every window
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        if (2 * ch + a < H && 2 * cw + b < W) gx[base + a * W + b] = 0.f;
    return;
  }
  const float* p = x + base;
  const float g = gy[(nc * OH + ch) * OW + cw];
  const int k = maxpool_argmax(p[0], p[1], p[W], p[W + 1]);
  if ((W & 1) == 0) {
    *reinterpret_cast<float2*>(gx + base) = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(gx + base + W) = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
  } else {
    gx[base] = k == 0 ? g : 0.f; gx[base + 1] = k == 1 ? g : 0.f;
    gx[base + W] = k == 2 ? g : 0.f; gx[base + W + 1] = k == 3 ? g : 0.f;
  }
}

__global__ void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int HW) {
  const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (plane >= NC) return;
  const float* xp = x + (size_t)plane * HW;
  float s = 0.f;
  for (int i = lane; i < HW; i += 64) s += xp[i];
  s = sg_wave_sum(s);
  if (lane == 0) y[plane] = s / (float)HW;
}

__global__ void gap_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, size_t total, int HW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) gx[i] = gy[i / HW] / (float)HW;
}

// ---------------- elementwise / layout glue ---------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sg_apply_act(x[i], act, slope);
}

__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, size_t n,
                               int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float o = y[i];
  float d = 1.f;
  switch (act) {
    case SG_ACT_RELU: d = o > 0.f ? 1.f : 0.f; break;
    case SG_ACT_LEAKY: d = o > 0.f ? 1.f : slope; break;   // slope>0: sign(output)==sign(input)
    case SG_ACT_TANH: d = 1.f - o * o; break;
    case SG_ACT_SIGMOID: d = o * (1.f - o); break;
    default: break;
  }
  gx[i] = gy[i] * d;
}

__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int W) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ow = i % (2 * W);
  const int oh = (i / (2 * W)) % (2 * H);
  const size_t nc = i / ((size_t)4 * W * H);
  y[i] = x[nc * H * W + (size_t)(oh >> 1) * W + (ow >> 1)];
}

__global__ void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int W,
                                       int pad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int PW = W + 2 * pad, PH = H + 2 * pad;
  const int ow = i % PW;
  const int oh = (i / PW) % PH;
  const size_t nc = i / ((size_t)PW * PH);
  int ih = oh - pad, iw = ow - pad;
  ih = ih < 0 ? -ih : (ih >= H ? 2 * H - 2 - ih : ih);
  iw = iw < 0 ? -iw : (iw >= W ? 2 * W - 2 - iw : iw);
  y[i] = x[nc * H * W + (size_t)ih * W + iw];
}

// candidates (<=3) of padded-grid coordinates that reflect onto logical coordinate l (0<=l<L, pad p)
__device__ __forceinline__ int reflect_sources(int l, int L, int p, int (&a)[3]) {
  int n = 0;
  a[n++] = l + p;
  if (l >= 1 && l <= p) a[n++] = p - l;
  if (l <= L - 2 && l >= L - 1 - p) a[n++] = p + 2 * L - 2 - l;
  return n;
}

// grid (ceil(H W / 256), N C): 32-bit index arithmetic (the flat size_t form spent most of its time in the emulated 64-bit
// divisions: 147 us for the 134 MB gradient of the RGB head's reflection pad)
__global__ void pad_upsample_bwd_kernel(const float* __restrict__ gp, float* __restrict__ gx, int H, int W, int pad, int ups) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (unsigned)(H * W)) return;
  const size_t nc = blockIdx.y;
  const int h = (int)(j / (unsigned)W), w = (int)(j - (unsigned)h * (unsigned)W);
  const int LH = H * ups, LW = W * ups, PH = LH + 2 * pad, PW = LW + 2 * pad;
  const float* g = gp + nc * PH * PW;
  float s = 0.f;
  for (int dh = 0; dh < ups; ++dh) {
    int ah[3];
    const int nh = reflect_sources(h * ups + dh, LH, pad, ah);
    for (int dw = 0; dw < ups; ++dw) {
      int aw[3];
      const int nw = reflect_sources(w * ups + dw, LW, pad, aw);
      for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) s += g[ah[a] * PW + aw[b]];
    }
  }
  gx[nc * H * W + j] = s;
}

__global__ void concat_channels_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                       size_t total, int Ca, int Cb, int HW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int p = i % HW;
  const int c = (i / HW) % (Ca + Cb);
  const size_t n = i / ((size_t)HW * (Ca + Cb));
  out[i] = c < Ca ? a[(n * Ca + c) * HW + p] : b[(n * Cb + (c - Ca)) * HW + p];
}


// nn.MaxPool2d(k, k) / nn.AvgPool2d(k, k) for any window (build_cnn 'P<k>', layers.py:181-189): no padding, floor output size.
// One thread per output pixel; max follows torch (first maximum in row-major order, NaN wins).
__global__ void pool2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int W, int OH,
                                  int OW, int k, int avg) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ow = i % OW;
  const int oh = (i / OW) % OH;
  const size_t nc = i / ((size_t)OW * OH);
  const float* p = x + (nc * H + (size_t)oh * k) * W + (size_t)ow * k;
  float m = p[0], s = 0.f;
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      const float v = p[(size_t)a * W + b];
      s += v;
      if (v > m || v != v) m = v;
    }
  y[i] = avg ? s / (float)(k * k) : m;
}
// one thread per INPUT pixel: its window's gradient (avg: gy / k^2; max: gy if this pixel is the window's first maximum);
// pixels in the uncovered tail (H % k, W % k) get zero
__global__ void pool2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, size_t total,
                                  int H, int W, int OH, int OW, int k, int avg) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int w = i % W;
  const int h = (i / W) % H;
  const size_t nc = i / ((size_t)W * H);
  const int oh = h / k, ow = w / k;
  if (oh >= OH || ow >= OW) { gx[i] = 0.f; return; }
  const float g = gy[(nc * OH + oh) * OW + ow];
  if (avg) { gx[i] = g / (float)(k * k); return; }
  const float* p = x + (nc * H + (size_t)oh * k) * W + (size_t)ow * k;
  float m = p[0];
  int am = 0;
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      const float v = p[(size_t)a * W + b];
      if ((v > m || v != v) && !(m != m)) { m = v; am = a * k + b; }
    }
  gx[i] = am == (h - oh * k) * k + (w - ow * k) ? g : 0.f;
}

// nn.ReplicationPad2d(p) (ResnetBlock padding_type='replicate', layers.py:245-246,258-259) and its adjoint: every source
// pixel gathers the padded positions that were copied from it (edge rows / columns collect the pad strip, corners the pad
// square), in a fixed order -- no atomics
__global__ void replicate_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int W, int pad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int PW = W + 2 * pad, PH = H + 2 * pad;
  const int ow = i % PW;
  const int oh = (i / PW) % PH;
  const size_t nc = i / ((size_t)PW * PH);
  const int ih = min(max(oh - pad, 0), H - 1), iw = min(max(ow - pad, 0), W - 1);
  y[i] = x[nc * H * W + (size_t)ih * W + iw];
}
__global__ void replicate_pad_bwd_kernel(const float* __restrict__ gp, float* __restrict__ gx, size_t total, int H, int W, int pad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int w = i % W;
  const int h = (i / W) % H;
  const size_t nc = i / ((size_t)W * H);
  const int PW = W + 2 * pad, PH = H + 2 * pad;
  const float* g = gp + nc * PH * PW;
  // padded rows [r0, r1] and columns [c0, c1] that replicate (h, w)
  const int r0 = h == 0 ? 0 : h + pad, r1 = h == H - 1 ? PH - 1 : h + pad;
  const int c0 = w == 0 ? 0 : w + pad, c1 = w == W - 1 ? PW - 1 : w + pad;
  float s = 0.f;
  for (int r = r0; r <= r1; ++r)
    for (int c = c0; c <= c1; ++c) s += g[(size_t)r * PW + c];
  gx[i] = s;
}


// ---------------- conv over [x1 || row broadcast] with the broadcast source folded away ----------------------------------------
// MultiscaleMaskDiscriminator.singleD_forward (discriminators.py:107-110) concatenates a per-object row cond[n][C2] (the one-hot
// class), expanded over the grid, to a C1-channel feature map and runs a zero-padded conv over the C1 + C2 channels.  A channel
// that is constant over the plane contributes  sum_{taps that land inside the plane} W[m][C1 + c2][tap] * cond[n][c2]  to output
// (n, m, oh, ow): with P[n][m][tap] = sum_c2 cond[n][c2] W[m][C1 + c2][tap] (a [N x C2] x [C2 x M*R] dense layer) the conv is
//   y = conv(x1, W[:, :C1]) + sum_{tap valid at (oh, ow)} P[n][m][tap],
// i.e. C1 instead of C1 + C2 gathered channels (128 of 300 at configs[1]: 2.3x fewer MACs forward and in the weight gradient).
// Exact in real arithmetic; in fp32 the C2 products are summed per tap before they meet the C1 products.
__global__ void cond_split_w_kernel(const float* __restrict__ Wt, float* __restrict__ W1, float* __restrict__ W2r, size_t total,
                                    int C1, int C2, int R) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int C = C1 + C2;
  const int t = (int)(i % R), c = (int)((i / R) % C);
  const size_t m = i / ((size_t)R * C);
  const float v = Wt[i];
  if (c < C1) W1[(m * C1 + c) * R + t] = v;
  else W2r[(m * R + t) * C2 + (c - C1)] = v;
}
// the adjoint of the split: gW[m][c][t] = c < C1 ? gW1[m][c][t] : gW2r[m*R + t][c - C1]   (a null source reads as zeros)
__global__ void cond_merge_w_kernel(const float* __restrict__ gW1, const float* __restrict__ gW2r, float* __restrict__ gW,
                                    size_t total, int C1, int C2, int R) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int C = C1 + C2;
  const int t = (int)(i % R), c = (int)((i / R) % C);
  const size_t m = i / ((size_t)R * C);
  float v = 0.f;
  if (c < C1) { if (gW1) v = gW1[(m * C1 + c) * R + t]; }
  else if (gW2r) v = gW2r[(m * R + t) * C2 + (c - C1)];
  gW[i] = v;
}
// y[nm][oh][ow] = act(y + sum_{(kh, kw): oh*stride - pad + kh in [0, H), ow*stride - pad + kw in [0, W)} P[nm][kh*KS + kw]), in place
__global__ void cond_bias_act_kernel(float* __restrict__ y, const float* __restrict__ P, size_t total, int OH, int OW, int H, int W,
                                     int KS, int stride, int pad, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ow = (int)(i % OW), oh = (int)((i / OW) % OH);
  const size_t nm = i / ((size_t)OW * OH);
  const float* p = P + nm * (size_t)(KS * KS);
  float s = 0.f;
  for (int kh = 0; kh < KS; ++kh) {
    const int ih = oh * stride - pad + kh;
    if (ih < 0 || ih >= H) continue;
    for (int kw = 0; kw < KS; ++kw) {
      const int iw = ow * stride - pad + kw;
      if (iw >= 0 && iw < W) s += p[kh * KS + kw];
    }
  }
  y[i] = sg_apply_act(y[i] + s, act, slope);
}
// the adjoint: gP[nm][t] = sum of g[nm][oh][ow] over the outputs tap t reaches; one wave per plane, every element read once,
// fixed summation order (lane-strided partials, xor-shuffle tree)
template <int KS>
__global__ void __launch_bounds__(256) cond_window_sums_kernel(const float* __restrict__ g, float* __restrict__ gP, size_t planes,
                                                               int OH, int OW, int H, int W, int stride, int pad) {
  const size_t nm = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nm >= planes) return;
  const int lane = threadIdx.x & 63;
  const float* gp = g + nm * (size_t)OH * OW;
  float acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) acc[t] = 0.f;
  for (int p = lane; p < OH * OW; p += 64) {
    const int oh = p / OW, ow = p - oh * OW;
    const float v = gp[p];
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      const int ih = oh * stride - pad + kh;
      const bool rok = ih >= 0 && ih < H;
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const int iw = ow * stride - pad + kw;
        acc[kh * KS + kw] += (rok && iw >= 0 && iw < W) ? v : 0.f;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    const float s = sg_wave_sum(acc[t]);
    if (lane == t) gP[nm * (KS * KS) + t] = s;
  }
}

inline dim3 grid1d(size_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

}  // namespace

extern "C" int sg_instnorm_fwd(const float* x, const float* skip, float* y, float* mean, float* rstd, int NC, int HW,
                               float eps, int act, float slope, sgStream stream) {
  SG_ARG_CHECK(x && y && mean && rstd && NC > 0 && HW > 0, "sg_instnorm_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_INSTNORM, s, 0, (double)NC * HW * 4.0 * (skip ? 3 : 2));      // algorithmic: x (+ skip) in, y out
  const int reg = sg_opt(SG_OPT_INSTNORM_REG);      // 0: the three-pass kernels
  int G = 0, E = 0;
  instnorm_shape(HW, 64, G, E);
  int G4 = 0, E4 = 0;
  if (reg >= 2 && instnorm_vec_ok(HW, x, y, skip, nullptr)) instnorm_shape_vec(HW, 16, G4, E4);
  if (G4 > 0) {
    // forward: up to 16 float4s (64 values) per thread in groups of <= 256 threads, 8 in groups of 1024
#define SG_INV_FWD(Gv, Ev) SG_INV_CASE(instnorm_fwd_vec_kernel, Gv, Ev, x, skip, y, mean, rstd, NC, HW, eps, act, slope)
    if (G4 == 16) { SG_INV_FWD(16, 1); }
    else if (G4 == 64) { SG_INV_ROW(instnorm_fwd_vec_kernel, 64, x, skip, y, mean, rstd, NC, HW, eps, act, slope) }
    else if (G4 == 256 && E4 == 16) { SG_INV_FWD(256, 16); }
    else if (G4 == 256 && E4 == 8) { SG_INV_FWD(256, 8); }
    else if (G4 == 256) { SG_INV_ROW(instnorm_fwd_vec_kernel, 256, x, skip, y, mean, rstd, NC, HW, eps, act, slope) }
    else if (E4 == 8) { SG_INV_FWD(1024, 8); }
    else { SG_INV_ROW(instnorm_fwd_vec_kernel, 1024, x, skip, y, mean, rstd, NC, HW, eps, act, slope) }
#undef SG_INV_FWD
  } else if (reg && G > 0) {
    SG_IN_DISPATCH(instnorm_fwd_reg_kernel, 32, x, skip, y, mean, rstd, NC, HW, eps, act, slope)
  } else if (reg && instnorm_big_ok(HW, x, y, skip, nullptr)) {
    hipLaunchKernelGGL(instnorm_fwd_big_kernel, dim3(NC), dim3(1024), 0, s, x, skip, y, mean, rstd, HW, eps, act, slope);
  } else if (HW <= 1024) hipLaunchKernelGGL(instnorm_fwd_kernel<true>, dim3(sg_cdiv(NC, 4)), dim3(256), 0, s, x, skip, y, mean, rstd, NC, HW, eps, act, slope);
  else hipLaunchKernelGGL(instnorm_fwd_kernel<false>, dim3(NC), dim3(256), 0, s, x, skip, y, mean, rstd, NC, HW, eps, act, slope);
  SG_LAUNCH_CHECK("sg_instnorm_fwd");
  return 0;
}

extern "C" int sg_instnorm_bwd(const float* x, const float* gy, const float* mean, const float* rstd, float* gx, int NC,
                               int HW, int act, float slope, sgStream stream) {
  SG_ARG_CHECK(x && gy && mean && rstd && gx && NC > 0 && HW > 0, "sg_instnorm_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_INSTNORM_BWD, s, 0, (double)NC * HW * 4.0 * 3);      // algorithmic: x, gy in, gx out
  const int reg = sg_opt(SG_OPT_INSTNORM_REG);
  int G = 0, E = 0;
  instnorm_shape(HW, 32, G, E);                        // (two register arrays: 2 x 32 values per thread at most)
  int G4 = 0, E4 = 0;
  if (reg >= 2 && instnorm_vec_ok(HW, x, gy, gx, nullptr)) instnorm_shape_vec(HW, 8, G4, E4);
  if (G4 > 0) {
    // backward (two register arrays): up to 8 float4s per array and thread in groups of <= 256 threads, 4 in groups of 1024
#define SG_INV_BWD(Gv, Ev) SG_INV_CASE(instnorm_bwd_vec_kernel, Gv, Ev, x, gy, mean, rstd, gx, NC, HW, act, slope)
    if (G4 == 16) { SG_INV_BWD(16, 1); }
    else if (G4 == 64) { SG_INV_ROW(instnorm_bwd_vec_kernel, 64, x, gy, mean, rstd, gx, NC, HW, act, slope) }
    else if (G4 == 256 && E4 == 8) { SG_INV_BWD(256, 8); }
    else if (G4 == 256) { SG_INV_ROW(instnorm_bwd_vec_kernel, 256, x, gy, mean, rstd, gx, NC, HW, act, slope) }
    else { SG_INV_ROW(instnorm_bwd_vec_kernel, 1024, x, gy, mean, rstd, gx, NC, HW, act, slope) }
#undef SG_INV_BWD
  } else if (reg && G > 0) {
    SG_IN_DISPATCH(instnorm_bwd_reg_kernel, 16, x, gy, mean, rstd, gx, NC, HW, act, slope)
  } else if (reg && instnorm_big_ok(HW, x, gy, gx, nullptr)) {
    hipLaunchKernelGGL(instnorm_bwd_big_kernel, dim3(NC), dim3(1024), 0, s, x, gy, mean, rstd, gx, HW, act, slope);
  } else if (HW <= 1024) hipLaunchKernelGGL(instnorm_bwd_kernel<true>, dim3(sg_cdiv(NC, 4)), dim3(256), 0, s, x, gy, mean, rstd, gx, NC, HW, act, slope);
  else hipLaunchKernelGGL(instnorm_bwd_kernel<false>, dim3(NC), dim3(256), 0, s, x, gy, mean, rstd, gx, NC, HW, act, slope);
  SG_LAUNCH_CHECK("sg_instnorm_bwd");
  return 0;
}

extern "C" size_t sg_batchnorm_ws_bytes(int N, int C, int HW) {
  return ((size_t)C * bn_slices(N, C, HW) * 3 + 2 * (size_t)C) * sizeof(float) + 64;
}

extern "C" int sg_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                                float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches, int N,
                                int C, int HW, float eps, float momentum, int training, int act, float slope, void* ws,
                                size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(x && y && save_mean && save_rstd && N > 0 && C > 0 && HW > 0, "sg_batchnorm_fwd: bad arguments");
  SG_ARG_CHECK(training || (running_mean && running_var), "sg_batchnorm_fwd: eval mode needs running stats");
  SG_ARG_CHECK(!training || (ws && ws_bytes >= sg_batchnorm_ws_bytes(N, C, HW)), "sg_batchnorm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_BATCHNORM, s, 0, (double)N * C * HW * 8.0);       // algorithmic: x in, y out
  const int S = bn_slices(N, C, HW);
  float* part = reinterpret_cast<float*>(ws);
  int* counter = (training && C <= 4096) ? sg_counter_alloc(s, C, false, 2) : nullptr;
  if (training)
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, S), dim3(256), 0, s, x, part, N, C, HW, S,
                       BnFinal{counter, save_mean, save_rstd, running_mean, running_var, num_batches, eps, momentum});
  if (!counter)
    hipLaunchKernelGGL(bn_final_kernel, dim3(sg_cdiv(C, 64)), dim3(64), 0, s, (const float*)part, save_mean, save_rstd,
                       running_mean, running_var, num_batches, C, S, eps, momentum, training);
  if (HW >= 256)
    hipLaunchKernelGGL(bn_apply_kernel, dim3(sg_cdiv(HW, 1024), N * C), dim3(256), 0, s, x, gamma, beta,
                       (const float*)save_mean, (const float*)save_rstd, y, C, HW, act, slope);
  else
    hipLaunchKernelGGL(bn_apply_flat_kernel, dim3(sg_cdiv((size_t)N * C * HW, 256)), dim3(256), 0, s, x, gamma, beta,
                       (const float*)save_mean, (const float*)save_rstd, y, (size_t)N * C * HW, C, HW, act, slope);
  SG_LAUNCH_CHECK("sg_batchnorm_fwd");
  return 0;
}

extern "C" int sg_batchnorm_bwd(const float* x, const float* gy, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_rstd, float* gx, float* ggamma, float* gbeta,
                                int N, int C, int HW, int training, int act, float slope, void* ws, size_t ws_bytes,
                                sgStream stream) {
  SG_ARG_CHECK(x && gy && save_mean && save_rstd && gx && N > 0 && C > 0 && HW > 0, "sg_batchnorm_bwd: bad arguments");
  SG_ARG_CHECK(ws && ws_bytes >= sg_batchnorm_ws_bytes(N, C, HW), "sg_batchnorm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_BATCHNORM, s, 0, (double)N * C * HW * 12.0);      // algorithmic: x, gy in, gx out
  const int S = bn_slices(N, C, HW);
  float* part = reinterpret_cast<float*>(ws);
  float* sums = part + (size_t)C * S * 3;
  int* counter = C <= 4096 ? sg_counter_alloc(s, C, false, 2) : nullptr;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C, S), dim3(256), 0, s, x, gy, gamma, beta, save_mean, save_rstd, part, N, C,
                     HW, S, act, slope, counter, sums, ggamma, gbeta);
  if (!counter)
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(sg_cdiv(C, 64)), dim3(64), 0, s, (const float*)part, sums, ggamma, gbeta, C, S);
  if (HW >= 256)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(sg_cdiv(HW, 1024), N * C), dim3(256), 0, s, x, gy, gamma, beta, save_mean,
                       save_rstd, (const float*)sums, gx, C, HW, 1.f / ((float)N * HW), training, act, slope);
  else
    hipLaunchKernelGGL(bn_bwd_apply_flat_kernel, dim3(sg_cdiv((size_t)N * C * HW, 256)), dim3(256), 0, s, x, gy, gamma, beta,
                       save_mean, save_rstd, (const float*)sums, gx, (size_t)N * C * HW, C, HW, 1.f / ((float)N * HW),
                       training, act, slope);
  SG_LAUNCH_CHECK("sg_batchnorm_bwd");
  return 0;
}
extern "C" size_t sg_channel_sum_ws_bytes(int C) { return (size_t)C * 64 * sizeof(float); }

extern "C" int sg_channel_sum(const float* g, float* out, int N, int C, int HW, void* ws, size_t ws_bytes,
                              sgStream stream) {
  SG_ARG_CHECK(g && out && N > 0 && C > 0 && HW > 0, "sg_channel_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long cnt = (long)N * HW;
  // small reductions: one block per channel; large ones: split so that ~1024 workgroups stream the tensor
  int S = (int)(cnt / 4096);
  const int want = (1024 + C - 1) / C;
  if (S > want) S = want;
  if (S > 64) S = 64;
  if (S > 1 && (!ws || ws_bytes < (size_t)C * S * sizeof(float))) S = 1;      // no scratch: single-stage fallback
  // >= 256 channels already give one workgroup per CU: up to 64 elements per thread the single launch (6-8 us) beats the
  // two-stage pair (14 + 5 us measured at 512 x 8192)
  if (C >= 256 && cnt <= 16384) S = 1;
  if (S <= 1) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, s, g, out, N, C, HW);
  } else {
    float* part = reinterpret_cast<float*>(ws);
    int* counter = C <= 4096 ? sg_counter_alloc(s, C, false, 4) : nullptr;
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(C, S), dim3(256), 0, s, g, part, N, C, HW, S, counter, out);
    if (!counter)
      hipLaunchKernelGGL(channel_sum_final_kernel, dim3(sg_cdiv(C, 64)), dim3(64), 0, s, (const float*)part, out, C, S);
  }
  SG_LAUNCH_CHECK("sg_channel_sum");
  return 0;
}

extern "C" int sg_avgpool3s2_fwd(const float* x, float* y, int NC, int H, int W, int OH, int OW, sgStream stream) {
  SG_ARG_CHECK(x && y && OH == (H + 2 - 3) / 2 + 1 && OW == (W + 2 - 3) / 2 + 1, "sg_avgpool3s2_fwd: bad arguments");
  hipLaunchKernelGGL(avgpool3s2_fwd_kernel, grid1d((size_t)NC * OH * OW), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, OH, OW);
  SG_LAUNCH_CHECK("sg_avgpool3s2_fwd");
  return 0;
}

extern "C" int sg_avgpool3s2_bwd(const float* gy, float* gx, int NC, int H, int W, int OH, int OW, sgStream stream) {
  SG_ARG_CHECK(gy && gx, "sg_avgpool3s2_bwd: bad arguments");
  hipLaunchKernelGGL(avgpool3s2_bwd_kernel, grid1d((size_t)NC * H * W), dim3(256), 0, (hipStream_t)stream, gy, gx, NC, H, W, OH, OW);
  SG_LAUNCH_CHECK("sg_avgpool3s2_bwd");
  return 0;
}

extern "C" int sg_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, sgStream stream) {
  SG_ARG_CHECK(x && y && NC > 0 && H >= 2 && W >= 2, "sg_maxpool2_fwd: bad arguments");
  const int OH = H / 2, OW = W / 2;
  SG_ARG_CHECK((double)NC * H * W < 2147483647.0 * 2, "sg_maxpool2_fwd: tensor too large");
  hipLaunchKernelGGL(maxpool2_fwd_kernel, grid1d((size_t)NC * OH * OW), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, OH, OW);
  SG_LAUNCH_CHECK("sg_maxpool2_fwd");
  return 0;
}

extern "C" int sg_maxpool2_bwd(const float* x, const float* gy, float* gx, int NC, int H, int W, sgStream stream) {
  SG_ARG_CHECK(x && gy && gx && NC > 0 && H >= 2 && W >= 2, "sg_maxpool2_bwd: bad arguments");
  const int OH = H / 2, OW = W / 2;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, grid1d((size_t)NC * ((H + 1) / 2) * ((W + 1) / 2)), dim3(256), 0, (hipStream_t)stream,
                     x, gy, gx, NC, H, W, OH, OW);
  SG_LAUNCH_CHECK("sg_maxpool2_bwd");
  return 0;
}

extern "C" int sg_gap_fwd(const float* x, float* y, int NC, int HW, sgStream stream) {
  SG_ARG_CHECK(x && y && NC > 0 && HW > 0, "sg_gap_fwd: bad arguments");
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(sg_cdiv(NC, 4)), dim3(256), 0, (hipStream_t)stream, x, y, NC, HW);
  SG_LAUNCH_CHECK("sg_gap_fwd");
  return 0;
}

extern "C" int sg_gap_bwd(const float* gy, float* gx, int NC, int HW, sgStream stream) {
  SG_ARG_CHECK(gy && gx && NC > 0 && HW > 0, "sg_gap_bwd: bad arguments");
  hipLaunchKernelGGL(gap_bwd_kernel, grid1d((size_t)NC * HW), dim3(256), 0, (hipStream_t)stream, gy, gx, (size_t)NC * HW, HW);
  SG_LAUNCH_CHECK("sg_gap_bwd");
  return 0;
}

extern "C" int sg_act_fwd(const float* x, float* y, int64_t n, int act, float slope, sgStream stream) {
  SG_ARG_CHECK(x && y && n >= 0, "sg_act_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_fwd_kernel, grid1d((size_t)n), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n, act, slope);
  SG_LAUNCH_CHECK("sg_act_fwd");
  return 0;
}

extern "C" int sg_act_bwd(const float* y, const float* gy, float* gx, int64_t n, int act, float slope, sgStream stream) {
  SG_ARG_CHECK(y && gy && gx && n >= 0, "sg_act_bwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, grid1d((size_t)n), dim3(256), 0, (hipStream_t)stream, y, gy, gx, (size_t)n, act, slope);
  SG_LAUNCH_CHECK("sg_act_bwd");
  return 0;
}

extern "C" int sg_upsample2_fwd(const float* x, float* y, int NC, int H, int W, sgStream stream) {
  SG_ARG_CHECK(x && y, "sg_upsample2_fwd: bad arguments");
  const size_t total = (size_t)NC * 4 * H * W;
  hipLaunchKernelGGL(upsample2_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W);
  SG_LAUNCH_CHECK("sg_upsample2_fwd");
  return 0;
}

extern "C" int sg_reflect_pad_fwd(const float* x, float* y, int NC, int H, int W, int pad, sgStream stream) {
  SG_ARG_CHECK(x && y && pad < H && pad < W, "sg_reflect_pad_fwd: bad arguments");
  const size_t total = (size_t)NC * (H + 2 * pad) * (W + 2 * pad);
  hipLaunchKernelGGL(reflect_pad_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, pad);
  SG_LAUNCH_CHECK("sg_reflect_pad_fwd");
  return 0;
}

extern "C" int sg_pool2d_fwd(const float* x, float* y, int NC, int H, int W, int k, int avg, sgStream stream) {
  SG_ARG_CHECK(x && y && NC > 0 && k >= 1 && H >= k && W >= k, "sg_pool2d_fwd: bad arguments");
  const int OH = H / k, OW = W / k;
  const size_t total = (size_t)NC * OH * OW;
  hipLaunchKernelGGL(pool2d_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, OH, OW, k, avg);
  SG_LAUNCH_CHECK("sg_pool2d_fwd");
  return 0;
}

extern "C" int sg_pool2d_bwd(const float* x, const float* gy, float* gx, int NC, int H, int W, int k, int avg, sgStream stream) {
  SG_ARG_CHECK((x || avg) && gy && gx && NC > 0 && k >= 1 && H >= k && W >= k, "sg_pool2d_bwd: bad arguments");
  const size_t total = (size_t)NC * H * W;
  hipLaunchKernelGGL(pool2d_bwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, gy, gx, total, H, W, H / k, W / k, k, avg);
  SG_LAUNCH_CHECK("sg_pool2d_bwd");
  return 0;
}

extern "C" int sg_replicate_pad_fwd(const float* x, float* y, int NC, int H, int W, int pad, sgStream stream) {
  SG_ARG_CHECK(x && y && NC > 0 && H > 0 && W > 0 && pad >= 0, "sg_replicate_pad_fwd: bad arguments");
  const size_t total = (size_t)NC * (H + 2 * pad) * (W + 2 * pad);
  hipLaunchKernelGGL(replicate_pad_fwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, x, y, total, H, W, pad);
  SG_LAUNCH_CHECK("sg_replicate_pad_fwd");
  return 0;
}

extern "C" int sg_replicate_pad_bwd(const float* gp, float* gx, int NC, int H, int W, int pad, sgStream stream) {
  SG_ARG_CHECK(gp && gx && NC > 0 && H > 0 && W > 0 && pad >= 0, "sg_replicate_pad_bwd: bad arguments");
  const size_t total = (size_t)NC * H * W;
  hipLaunchKernelGGL(replicate_pad_bwd_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, gp, gx, total, H, W, pad);
  SG_LAUNCH_CHECK("sg_replicate_pad_bwd");
  return 0;
}

extern "C" int sg_pad_upsample_bwd(const float* gp, float* gx, int NC, int H, int W, int pad, int upsample,
                                   sgStream stream) {
  SG_ARG_CHECK(gp && gx && (upsample == 1 || upsample == 2) && pad >= 0, "sg_pad_upsample_bwd: bad arguments");
  if (NC == 0 || H * W == 0) return 0;
  for (int n0 = 0; n0 < NC; n0 += 65535) {                  // (grid.y limit)
    const int nn = NC - n0 < 65535 ? NC - n0 : 65535;
    const size_t PP = (size_t)(H * upsample + 2 * pad) * (W * upsample + 2 * pad);
    hipLaunchKernelGGL(pad_upsample_bwd_kernel, dim3(sg_cdiv(H * W, 256), nn), dim3(256), 0, (hipStream_t)stream,
                       gp + (size_t)n0 * PP, gx + (size_t)n0 * H * W, H, W, pad, upsample);
  }
  SG_LAUNCH_CHECK("sg_pad_upsample_bwd");
  return 0;
}

extern "C" int sg_concat_channels(const float* a, const float* b, float* out, int N, int Ca, int Cb, int HW,
                                  sgStream stream) {
  SG_ARG_CHECK(a && b && out, "sg_concat_channels: bad arguments");
  const size_t total = (size_t)N * (Ca + Cb) * HW;
  hipLaunchKernelGGL(concat_channels_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, a, b, out, total, Ca, Cb, HW);
  SG_LAUNCH_CHECK("sg_concat_channels");
  return 0;
}

extern "C" int sg_cond_conv_split_w(const float* w, float* w1, float* w2r, int M, int C1, int C2, int R, sgStream stream) {
  SG_ARG_CHECK(w && w1 && w2r && M > 0 && C1 > 0 && C2 > 0 && R > 0, "sg_cond_conv_split_w: bad arguments");
  const size_t total = (size_t)M * (C1 + C2) * R;
  hipLaunchKernelGGL(cond_split_w_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, w, w1, w2r, total, C1, C2, R);
  SG_LAUNCH_CHECK("sg_cond_conv_split_w");
  return 0;
}

extern "C" int sg_cond_conv_merge_w(const float* gw1, const float* gw2r, float* gw, int M, int C1, int C2, int R, sgStream stream) {
  SG_ARG_CHECK(gw && M > 0 && C1 > 0 && C2 > 0 && R > 0, "sg_cond_conv_merge_w: bad arguments");
  const size_t total = (size_t)M * (C1 + C2) * R;
  hipLaunchKernelGGL(cond_merge_w_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, gw1, gw2r, gw, total, C1, C2, R);
  SG_LAUNCH_CHECK("sg_cond_conv_merge_w");
  return 0;
}

extern "C" int sg_cond_conv_bias_act(float* y, const float* p, int NM, int OH, int OW, int H, int W, int KS, int stride, int pad,
                                     int act, float slope, sgStream stream) {
  SG_ARG_CHECK(y && p && NM >= 0 && OH > 0 && OW > 0 && H > 0 && W > 0 && KS > 0 && stride > 0 && pad >= 0,
               "sg_cond_conv_bias_act: bad arguments");
  const size_t total = (size_t)NM * OH * OW;
  if (total == 0) return 0;
  hipLaunchKernelGGL(cond_bias_act_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, y, p, total, OH, OW, H, W, KS, stride,
                     pad, act, slope);
  SG_LAUNCH_CHECK("sg_cond_conv_bias_act");
  return 0;
}

extern "C" int sg_cond_conv_window_sums(const float* g, float* gp, int NM, int OH, int OW, int H, int W, int KS, int stride, int pad,
                                        sgStream stream) {
  SG_ARG_CHECK(g && gp && NM >= 0 && OH > 0 && OW > 0 && H > 0 && W > 0 && stride > 0 && pad >= 0 && (KS == 1 || KS == 3 || KS == 4),
               "sg_cond_conv_window_sums: bad arguments (kernel sizes 1, 3, 4)");
  if (NM == 0) return 0;
  const dim3 grid((unsigned)sg_cdiv(NM, 4));
  hipStream_t s = (hipStream_t)stream;
  if (KS == 1) hipLaunchKernelGGL(cond_window_sums_kernel<1>, grid, dim3(256), 0, s, g, gp, (size_t)NM, OH, OW, H, W, stride, pad);
  else if (KS == 3) hipLaunchKernelGGL(cond_window_sums_kernel<3>, grid, dim3(256), 0, s, g, gp, (size_t)NM, OH, OW, H, W, stride, pad);
  else hipLaunchKernelGGL(cond_window_sums_kernel<4>, grid, dim3(256), 0, s, g, gp, (size_t)NM, OH, OW, H, W, stride, pad);
  SG_LAUNCH_CHECK("sg_cond_conv_window_sums");
  return 0;
}
