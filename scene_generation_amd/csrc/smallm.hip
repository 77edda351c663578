// Direct kernels for convolutions with <= 4 output channels (the generator's last layer: ReflectionPad2d(3) +
// Conv2d(64, 3, 7) + Tanh at full resolution, generators.py:88-90).  As an implicit GEMM they have M = 3 rows: a 32-row
// MFMA tile wastes 90 % of the matrix pipe (measured: forward 6.7, weight gradient 3.0 TFLOP/s).  Here the vector ALUs do
// the work: per thread a 4-pixel x MO-channel micro-tile, inputs staged through LDS with the reflection resolved at
// staging time, weights fetched as wave-uniform scalars.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int L) {
  i = i < 0 ? -i : i;
  return i >= L ? 2 * L - 2 - i : i;
}

// ---- forward: workgroup = 16 x 64 output pixels of one image, loop over input channels -------------------------------
template <int KS, int MO>
__global__ void __launch_bounds__(256) smallm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                        int W, int M, int act, float slope) {
  constexpr int PAD = (KS - 1) / 2, TH = 16, TW = 64, RH = TH + KS - 1, RW = TW + KS - 1, PITCH = (RW + 3) / 4 * 4;
  constexpr int NE = (RH * RW + 255) / 256;
  __shared__ __attribute__((aligned(16))) float tile[2][RH * PITCH];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n = blockIdx.z, oh0 = blockIdx.y * TH, ow0 = blockIdx.x * TW;
  const size_t HW = (size_t)H * W;
  const float* xn = x + (size_t)n * C * HW;
  int goff[NE], loff[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int i = tid + e * 256;
    const int r = i / RW, q = i - r * RW;
    const bool ok = i < RH * RW;
    goff[e] = ok ? reflect_idx(min(oh0 + r - PAD, 2 * H - 2), H) * W + reflect_idx(min(ow0 + q - PAD, 2 * W - 2), W) : -1;
    loff[e] = r * PITCH + q;
  }
  float acc[MO][4];
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[m][p] = 0.f;
  for (int c = 0; c < C; ++c) {
    float* T = tile[c & 1];
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (goff[e] >= 0) T[loff[e]] = xn[(size_t)c * HW + goff[e]];
    __syncthreads();
    const float* wc = w + (size_t)c * KS * KS;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      const float* row = T + (ty + kh) * PITCH + 4 * tx;
      float in[12];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const float4 t4 = *reinterpret_cast<const float4*>(row + 4 * v);
        in[4 * v] = t4.x; in[4 * v + 1] = t4.y; in[4 * v + 2] = t4.z; in[4 * v + 3] = t4.w;
      }
#pragma unroll
      for (int m = 0; m < MO; ++m) {
        if (m < M) {
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const float wv = wc[(size_t)m * C * KS * KS + kh * KS + kw];       // wave-uniform
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[m][p] = fmaf(wv, in[p + kw], acc[m][p]);
          }
        }
      }
    }
  }
  const int oh = oh0 + ty, ow = ow0 + 4 * tx;
  if (oh < H && ow < W) {
#pragma unroll
    for (int m = 0; m < MO; ++m) {
      if (m < M) {
        const float b = bias ? bias[m] : 0.f;
        float4 o;
        o.x = sg_apply_act(acc[m][0] + b, act, slope); o.y = sg_apply_act(acc[m][1] + b, act, slope);
        o.z = sg_apply_act(acc[m][2] + b, act, slope); o.w = sg_apply_act(acc[m][3] + b, act, slope);
        *reinterpret_cast<float4*>(y + ((size_t)n * M + m) * HW + (size_t)oh * W + ow) = o;
      }
    }
  }
}

// ---- weight gradient: workgroup = (image n, input channel c); thread = (kh, 4-pixel column group) -------------------
// gw[m][c][kh][kw] = sum_{n,oh,ow} gy[n][m][oh][ow] * xpad[n][c][oh+kh][ow+kw].  Partial sums per image go to
// part[n][m][c][kh][kw]; a fixed-order reduce over n finishes (deterministic).
template <int KS, int MO>
__global__ void __launch_bounds__(256) smallm_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                          float* __restrict__ part, int C, int H, int W, int M) {
  constexpr int PAD = (KS - 1) / 2, RB = 16, CB = 128, XW = CB + KS - 1, XP = (XW + 3) / 4 * 4, XR = RB + KS - 1;
  __shared__ __attribute__((aligned(16))) float xs[XR * XP];
  __shared__ __attribute__((aligned(16))) float gs[MO * RB * CB];
  const int tid = threadIdx.x, lane = tid & 31, kh = tid >> 5;      // kh == 7: staging helper only (KS == 7)
  const int c = blockIdx.x, n = blockIdx.y;
  const size_t HW = (size_t)H * W;
  const float* xc = x + ((size_t)n * C + c) * HW;
  const float* gn = gy + (size_t)n * M * HW;
  float acc[MO][KS];
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[m][k] = 0.f;
  for (int ow0 = 0; ow0 < W; ow0 += CB) {
    for (int oh0 = 0; oh0 < H; oh0 += RB) {
      __syncthreads();
      for (int i = tid; i < XR * XW; i += 256) {
        const int r = i / XW, q = i - r * XW;
        const int ih = reflect_idx(min(oh0 + r - PAD, 2 * H - 2), H), iw = reflect_idx(min(ow0 + q - PAD, 2 * W - 2), W);
        xs[r * XP + q] = xc[(size_t)ih * W + iw];
      }
      for (int i = tid; i < MO * RB * (CB / 4); i += 256) {
        const int q4 = i % (CB / 4), r = (i / (CB / 4)) % RB, m = i / (RB * (CB / 4));
        const int oh = oh0 + r, ow = ow0 + 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && oh < H && ow < W) v = *reinterpret_cast<const float4*>(gn + (size_t)m * HW + (size_t)oh * W + ow);
        *reinterpret_cast<float4*>(gs + (m * RB + r) * CB + 4 * q4) = v;
      }
      __syncthreads();
      if (kh < KS) {
#pragma unroll 2
        for (int r = 0; r < RB; ++r) {
          float in[12];
          const float* row = xs + (r + kh) * XP + 4 * lane;
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const float4 t4 = *reinterpret_cast<const float4*>(row + 4 * v);
            in[4 * v] = t4.x; in[4 * v + 1] = t4.y; in[4 * v + 2] = t4.z; in[4 * v + 3] = t4.w;
          }
#pragma unroll
          for (int m = 0; m < MO; ++m) {
            const float4 g4 = *reinterpret_cast<const float4*>(gs + (m * RB + r) * CB + 4 * lane);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
              for (int p = 0; p < 4; ++p) acc[m][kw] = fmaf(g[p], in[p + kw], acc[m][kw]);
          }
        }
      }
    }
  }
  // reduce over the 32 column groups of each kh (the two 32-lane halves of a wave hold different kh)
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      float v = acc[m][kw];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      acc[m][kw] = v;
    }
  if (lane == 0 && kh < KS) {
#pragma unroll
    for (int m = 0; m < MO; ++m)
      if (m < M)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
          part[(((size_t)n * M + m) * C + c) * KS * KS + kh * KS + kw] = acc[m][kw];
  }
}

__global__ void smallm_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += part[(size_t)z * n + i];
  out[i] = v;
}

bool smallm_ok(const sgConvDesc* d) {
  return d && d->Cout <= 4 && d->KS == 7 && d->stride == 1 && d->upsample == 1 && d->C2 == 0 && d->pad_reflect &&
         d->pad == 3 && d->W % 4 == 0 && d->H >= 4 && d->W >= 4 && d->OH == d->H && d->OW == d->W;
}

}  // namespace

extern "C" int sg_conv2d_smallm_supported(const sgConvDesc* d) { return smallm_ok(d) ? 1 : 0; }

extern "C" size_t sg_conv2d_smallm_ws_bytes(const sgConvDesc* d) {
  return smallm_ok(d) ? (size_t)d->N * d->Cout * d->C1 * 49 * sizeof(float) : 0;
}

extern "C" int sg_conv2d_smallm_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                                    float slope, sgStream stream) {
  SG_ARG_CHECK(smallm_ok(d), "sg_conv2d_smallm_fwd: unsupported desc (needs ReflectionPad(3)+7x7, stride 1, Cout <= 4, W %% 4 == 0)");
  SG_ARG_CHECK(x && w && y, "sg_conv2d_smallm_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(sg_igemm_kind(0, 7, 2), s, 2.0 * d->Cout * d->C1 * 49.0 * d->N * d->H * d->W, 0);
  const dim3 grid(sg_cdiv(d->W, 64), sg_cdiv(d->H, 16), d->N);
  hipLaunchKernelGGL((smallm_fwd_kernel<7, 4>), grid, dim3(256), 0, s, x, w, bias, y, d->C1, d->H, d->W, d->Cout, act, slope);
  SG_LAUNCH_CHECK("sg_conv2d_smallm_fwd");
  return 0;
}

extern "C" int sg_conv2d_smallm_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws,
                                      size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(smallm_ok(d), "sg_conv2d_smallm_wgrad: unsupported desc");
  SG_ARG_CHECK(gy && x && gw && ws && ws_bytes >= sg_conv2d_smallm_ws_bytes(d), "sg_conv2d_smallm_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  float* part = reinterpret_cast<float*>(ws);
  const size_t nw = (size_t)d->Cout * d->C1 * 49;
  {
    SgProfScope prof(sg_igemm_kind(2, 7, 2), s, 2.0 * d->Cout * d->C1 * 49.0 * d->N * d->H * d->W, 0);
    hipLaunchKernelGGL((smallm_wgrad_kernel<7, 4>), dim3(d->C1, d->N), dim3(256), 0, s, gy, x, part, d->C1, d->H, d->W,
                       d->Cout);
  }
  hipLaunchKernelGGL(smallm_reduce_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, (const float*)part, gw, nw, d->N);
  SG_LAUNCH_CHECK("sg_conv2d_smallm_wgrad");
  return 0;
}
