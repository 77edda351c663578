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

// ---- forward: workgroup = 16 x 64 output pixels of one image; two thread groups of 256 walk the even / odd input channels ----
// (one group: 512 workgroups of 4 waves = 2 waves per SIMD on 256 CUs and a barrier per channel -- 30 % of the vector-ALU peak;
//  two groups double the resident waves and halve the barriers per channel; the partial sums meet in LDS in a fixed order)
template <int KS, int MO>
__global__ void __launch_bounds__(512) smallm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                        int W, int M, int act, float slope) {
  constexpr int PAD = (KS - 1) / 2, TH = 16, TW = 64, RH = TH + KS - 1, RW = TW + KS - 1, PITCH = (RW + 3) / 4 * 4;
  constexpr int NE = (RH * RW + 255) / 256;
  static_assert(2 * 2 * RH * PITCH >= 256 * MO * 4, "the exchange of the partial sums re-uses the staging tiles");
  __shared__ __attribute__((aligned(16))) float tile[2][2][RH * PITCH];       // [group][buffer]
  // (readfirstlane: the group is the same for the 64 lanes of a wave, but the compiler only believes that when told -- without
  //  it the 21 weight rows of every channel came through VECTOR loads, each waited for on the spot)
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), tid = threadIdx.x & 255, tx = tid & 15, ty = tid >> 4;
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
  const int iters = (C + 1) / 2;
  // the plane of the NEXT channel is fetched into registers while this one is multiplied: the loads used to be issued and waited
  // for at the top of every iteration (~1.5 us of exposed latency per channel, 32 channels per workgroup)
  float pre[NE];
  if (grp < C) {
#pragma unroll
    for (int e = 0; e < NE; ++e) pre[e] = goff[e] >= 0 ? xn[(size_t)grp * HW + goff[e]] : 0.f;
  }
  for (int it = 0; it < iters; ++it) {
    const int c = 2 * it + grp;
    const bool live = c < C;                      // (odd C: the second group idles through the last iteration's barrier)
    float* T = tile[grp][it & 1];
    if (live) {
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (goff[e] >= 0) T[loff[e]] = pre[e];
    }
    __syncthreads();
    if (c + 2 < C) {
#pragma unroll
      for (int e = 0; e < NE; ++e) pre[e] = goff[e] >= 0 ? xn[(size_t)(c + 2) * HW + goff[e]] : 0.f;
    }
    if (live) {
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
        // rows m >= M (MO is the next instantiated size) repeat row M - 1 and are never stored: no branch in here, so the
        // scalar weight loads of a whole filter row are issued together
#pragma unroll
        for (int m = 0; m < MO; ++m) {
          const float* wm = wc + (size_t)(m < M ? m : M - 1) * C * KS * KS + kh * KS;
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const float wv = wm[kw];                                             // wave-uniform: scalar load
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[m][p] = fmaf(wv, in[p + kw], acc[m][p]);
          }
        }
      }
    }
  }
  __syncthreads();                                 // every read of the staging tiles is done: re-use them for the exchange
  float* xch = &tile[0][0][0];
  if (grp == 1) {
#pragma unroll
    for (int m = 0; m < MO; ++m)
      *reinterpret_cast<float4*>(xch + (m * 256 + tid) * 4) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int m = 0; m < MO; ++m) {
    const float4 o4 = *reinterpret_cast<const float4*>(xch + (m * 256 + tid) * 4);
    acc[m][0] += o4.x; acc[m][1] += o4.y; acc[m][2] += o4.z; acc[m][3] += o4.w;
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
      // staging: every global load of the tile is issued before the first LDS store (the plain "load, store" loops compiled to
      // a wait per element: ~12 + 6 dependent round trips per tile)
      constexpr int NX = (XR * XW + 255) / 256, NG = (MO * RB * (CB / 4) + 255) / 256;
      float xv[NX];
      float4 gv[NG];
#pragma unroll
      for (int e = 0; e < NX; ++e) {
        const int i = tid + e * 256;
        const int r = i / XW, q = i - r * XW;
        const int ih = reflect_idx(min(oh0 + r - PAD, 2 * H - 2), H), iw = reflect_idx(min(ow0 + q - PAD, 2 * W - 2), W);
        xv[e] = i < XR * XW ? xc[(size_t)ih * W + iw] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < NG; ++e) {
        const int i = tid + e * 256;
        const int q4 = i % (CB / 4), r = (i / (CB / 4)) % RB, m = i / (RB * (CB / 4));
        const int oh = oh0 + r, ow = ow0 + 4 * q4;
        gv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < MO * RB * (CB / 4) && m < M && oh < H && ow < W)
          gv[e] = *reinterpret_cast<const float4*>(gn + (size_t)m * HW + (size_t)oh * W + ow);
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < NX; ++e) {
        const int i = tid + e * 256;
        const int r = i / XW, q = i - r * XW;
        if (i < XR * XW) xs[r * XP + q] = xv[e];
      }
#pragma unroll
      for (int e = 0; e < NG; ++e) {
        const int i = tid + e * 256;
        const int q4 = i % (CB / 4), r = (i / (CB / 4)) % RB, m = i / (RB * (CB / 4));
        if (i < MO * RB * (CB / 4)) *reinterpret_cast<float4*>(gs + (m * RB + r) * CB + 4 * q4) = gv[e];
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
  out[i] = sg_sum_strided(part + i, n, S);
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
  if (d->Cout == 3)        // (the RGB head: no instructions for the fourth row)
    hipLaunchKernelGGL((smallm_fwd_kernel<7, 3>), grid, dim3(512), 0, s, x, w, bias, y, d->C1, d->H, d->W, d->Cout, act, slope);
  else
    hipLaunchKernelGGL((smallm_fwd_kernel<7, 4>), grid, dim3(512), 0, s, x, w, bias, y, d->C1, d->H, d->W, d->Cout, act, slope);
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
    // (MO = the real channel count: the <7, 4> instantiation spent a quarter of its FMAs on the zero row of the RGB head)
    if (d->Cout == 3)
      hipLaunchKernelGGL((smallm_wgrad_kernel<7, 3>), dim3(d->C1, d->N), dim3(256), 0, s, gy, x, part, d->C1, d->H, d->W, d->Cout);
    else
      hipLaunchKernelGGL((smallm_wgrad_kernel<7, 4>), dim3(d->C1, d->N), dim3(256), 0, s, gy, x, part, d->C1, d->H, d->W, d->Cout);
  }
  hipLaunchKernelGGL(smallm_reduce_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, (const float*)part, gw, nw, d->N);
  SG_LAUNCH_CHECK("sg_conv2d_smallm_wgrad");
  return 0;
}

// ================================================================================================
// Single-output-channel "head" convolutions (PatchGAN score maps: Conv2d(512, 1, 4, padding=2) of discriminators.py:232-234,
// Conv2d(256, 1, 3, padding=1) of the mask / object discriminators, the 1x1 head of mask_net, generators.py:27).
// As an implicit GEMM they have ONE row: the 32x128 MFMA tile ran at 1-2 TFLOP/s (0.27 ms per call for 0.19 GFLOP).  They are
// memory-bound reductions; here the vector ALUs do them from LDS-staged, zero-padded input planes:
//   forward : workgroup = (channel chunk, image) -> per-chunk partial maps, a second tiny launch adds them (+ bias, act)
//   dgrad   : workgroup = (channel chunk, image), gy of the image in LDS (one launch)
//   wgrad   : workgroup = (channel chunk, image), one THREAD per (channel, tap) sums over the pixels -> per-image partials,
//             combined over the images in a fixed order (deterministic)
// Global loads are issued in batches of eight before the first LDS store (a "load, store" loop waits for every load in turn:
// the first version was latency-bound at ~40 dependent HBM round trips per workgroup).
// ================================================================================================
namespace {

constexpr int HEAD_LDS_FLOATS = 12288;      // 48 KB per workgroup

struct HeadGeom {
  int C, H, W, OH, OW, KS, pad, PH, PW;
  int CH, chunks;              // channels per workgroup
  FastDiv dPW, dOW, dHW, dW, dGW, dKK, dPP;
};

HeadGeom head_geom(const sgConvDesc* d) {
  HeadGeom g;
  g.C = d->C1; g.H = d->H; g.W = d->W; g.OH = d->OH; g.OW = d->OW; g.KS = d->KS; g.pad = d->pad;
  g.PH = d->H + 2 * d->pad; g.PW = d->W + 2 * d->pad;
  const int PP = g.PH * g.PW, OP = g.OH * g.OW;
  int ch = (HEAD_LDS_FLOATS - 4 * OP) / PP;        // planes + per-wave sums (forward) / gy + pixel-offset table (wgrad)
  ch = ch > 16 ? 16 : ch;
  g.CH = ch < 1 ? 1 : ch;
  g.chunks = (g.C + g.CH - 1) / g.CH;
  g.dPW = FastDiv((unsigned)g.PW); g.dOW = FastDiv((unsigned)g.OW); g.dHW = FastDiv((unsigned)(g.H * g.W));
  g.dW = FastDiv((unsigned)g.W); g.dGW = FastDiv((unsigned)(g.OW + 2 * (g.KS - 1)));
  g.dKK = FastDiv((unsigned)(g.KS * g.KS));
  g.dPP = FastDiv((unsigned)PP);
  return g;
}

bool head_ok(const sgConvDesc* d) {
  if (!d || d->Cout != 1 || d->stride != 1 || d->upsample != 1 || d->C2 != 0 || d->pad_reflect) return false;
  if (!(d->KS == 1 || d->KS == 3 || d->KS == 4)) return false;
  if (d->OH != d->H + 2 * d->pad - d->KS + 1 || d->OW != d->W + 2 * d->pad - d->KS + 1 || d->OW > 256) return false;
  const int PP = (d->H + 2 * d->pad) * (d->W + 2 * d->pad), OP = d->OH * d->OW;
  return PP + 4 * OP <= HEAD_LDS_FLOATS && d->C1 >= 16 &&
         (d->OH + 2 * (d->KS - 1)) * (d->OW + 2 * (d->KS - 1)) + 16 * d->KS * d->KS <= HEAD_LDS_FLOATS;
}

// rows [prow0, prow0 + nrows) of the zero-padded planes of channels [c0, c0+nc) of one image -> LDS [nc][nrows*PW]
__device__ __forceinline__ void head_stage(const float* __restrict__ xn, float* __restrict__ pl, int c0, int nc, int prow0,
                                           int nrows, const FastDiv& dplane, const HeadGeom& g) {
  const int BP = nrows * g.PW, total = nc * BP;
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < total; i0 += U * 256) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      const int ii = i < total ? i : 0;
      const int c = (int)dplane.div((unsigned)ii), r = ii - c * BP, ph = (int)g.dPW.div((unsigned)r), pw = r - ph * g.PW;
      const int ih = prow0 + ph - g.pad, iw = pw - g.pad;
      const bool in = i < total && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const float t = xn[in ? ((size_t)(c0 + c) * g.H + ih) * g.W + iw : (size_t)0];      // unconditional load, clamped address
      v[u] = in ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      if (i < total) pl[i] = v[u];
    }
  }
}

// part[n][chunk][OH*OW] = sum over the chunk's channels and taps.  Wave v takes channels v, v+4, ... of the chunk for ALL pixels
// (lanes stride the pixels; weights are wave-uniform scalar loads), the four waves' sums are combined through LDS.
// (A one-launch form -- workgroup = (row band, image) looping over the channel chunks -- was 3x SLOWER: every chunk is a
// dependent stage -> barrier -> compute round trip; here the chunks run in parallel and a second tiny launch adds them up.)
template <int KS>
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ part, HeadGeom g) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int PP = g.PH * g.PW, OP = g.OH * g.OW;
  float* red = pl + g.CH * PP;                         // [4][OP]
  const int chunk = blockIdx.x, n = blockIdx.y, c0 = chunk * g.CH, nc = min(g.CH, g.C - c0);
  head_stage(x + (size_t)n * g.C * g.H * g.W, pl, c0, nc, 0, g.PH, g.dPP, g);
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: see smallm_fwd)
  for (int p = lane; p < OP; p += 64) {
    const int oh = (int)g.dOW.div((unsigned)p), ow = p - oh * g.OW;
    const float* base = pl + oh * g.PW + ow;
    float acc = 0.f;
    for (int c = wid; c < nc; c += 4) {
      const float* q = base + c * PP;
      const float* wc = w + (size_t)(c0 + c) * KS * KS;            // wave-uniform: scalar loads
#pragma unroll
      for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) acc = fmaf(wc[kh * KS + kw], q[kh * g.PW + kw], acc);
    }
    red[wid * OP + p] = acc;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < OP; p += blockDim.x)
    part[((size_t)n * g.chunks + chunk) * OP + p] = (red[p] + red[OP + p]) + (red[2 * OP + p] + red[3 * OP + p]);
}
// y[n][p] = act(bias + sum_chunk part[n][chunk][p])
__global__ void head_fwd_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                       int NB, int chunks, int OP, int act, float slope) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NB * OP) return;
  const int n = i / OP, p = i - n * OP;
  float v = bias ? bias[0] : 0.f;
  for (int k = 0; k < chunks; ++k) v += part[((size_t)n * chunks + k) * OP + p];
  y[i] = sg_apply_act(v, act, slope);
}

// gx[n][c][h][w] = sum_{kh,kw} w[c][kh][kw] * gy[n][h + pad - kh][w + pad - kw]
template <int KS>
__global__ void __launch_bounds__(256) head_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                        float* __restrict__ gx, HeadGeom g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // gy of image n, zero-extended by KS-1 on every side: [(OH + 2(KS-1))][(OW + 2(KS-1))]
  const int E = KS - 1, GW = g.OW + 2 * E, GH = g.OH + 2 * E;
  float* gl = sm;
  float* wl = sm + GH * GW;
  const int chunk = blockIdx.x, n = blockIdx.y, c0 = chunk * g.CH, nc = min(g.CH, g.C - c0);
  for (int i = threadIdx.x; i < GH * GW; i += blockDim.x) {
    const int r = (int)g.dGW.div((unsigned)i), q = i - r * GW, oh = r - E, ow = q - E;
    gl[i] = ((unsigned)oh < (unsigned)g.OH && (unsigned)ow < (unsigned)g.OW) ? gy[((size_t)n * g.OH + oh) * g.OW + ow] : 0.f;
  }
  for (int i = threadIdx.x; i < nc * KS * KS; i += blockDim.x) wl[i] = w[(size_t)c0 * KS * KS + i];
  __syncthreads();
  const int HW = g.H * g.W;
  float* out = gx + ((size_t)n * g.C + c0) * HW;
  for (int i = threadIdx.x; i < nc * HW; i += blockDim.x) {
    const int c = (int)g.dHW.div((unsigned)i), r = i - c * HW, h = (int)g.dW.div((unsigned)r), wq = r - h * g.W;
    // gy row index oh = h + pad - kh  ->  extended row (oh + E) = h + pad + E - kh
    const float* base = gl + (h + g.pad + E) * GW + (wq + g.pad + E);
    const float* wc = wl + c * KS * KS;
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) acc = fmaf(wc[kh * KS + kw], base[-kh * GW - kw], acc);
    out[i] = acc;
  }
}

// part[n][c][kh][kw] = sum_{oh,ow} gy[n][oh][ow] * xpad[n][c][oh+kh][ow+kw].  Thread j = (pixel part q, channel, tap): it walks
// the pixels p = q, q + Q, ... serially (gy[p] and the plane offset of p are LDS broadcasts) -- no cross-lane reduction per
// tap; the Q parts of a (channel, tap) are added through LDS at the end.
template <int KS>
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                        float* __restrict__ part, HeadGeom g) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  constexpr int KK = KS * KS;
  const int PP = g.PH * g.PW, OP = g.OH * g.OW;
  float* gl = pl + g.CH * PP;                          // gy of image n: [OP]
  int* offt = reinterpret_cast<int*>(gl + OP);         // plane offset of output pixel p: oh*PW + ow
  const int chunk = blockIdx.x, n = blockIdx.y, c0 = chunk * g.CH, nc = min(g.CH, g.C - c0);
  head_stage(x + (size_t)n * g.C * g.H * g.W, pl, c0, nc, 0, g.PH, g.dPP, g);
  for (int i = threadIdx.x; i < OP; i += blockDim.x) {
    gl[i] = gy[(size_t)n * OP + i];
    const int oh = (int)g.dOW.div((unsigned)i);
    offt[i] = oh * g.PW + (i - oh * g.OW);
  }
  __syncthreads();
  const int pairs = nc * KK;                           // <= 16 * 16 = 256
  const int Q = pairs >= 128 ? 1 : (pairs >= 64 ? 2 : (pairs >= 32 ? 4 : (pairs >= 16 ? 8 : 16)));   // pixel parts
  const int j = threadIdx.x % (256 / Q), q = threadIdx.x / (256 / Q);
  float acc = 0.f;
  if (j < pairs) {
    const int c = (int)g.dKK.div((unsigned)j), t = j - c * KK, kh = t / KS, kw = t - kh * KS;
    const float* b = pl + c * PP + kh * g.PW + kw;
    // (unrolled: the LDS reads of eight pixels are issued together; one-at-a-time the loop is a chain of dependent LDS
    //  round trips -- offset, then the plane element -- of ~150 cycles per pixel)
#pragma unroll 8
    for (int p = q; p < OP; p += Q) acc = fmaf(gl[p], b[offt[p]], acc);
  }
  __syncthreads();                                     // planes are dead: reuse their LDS for the parts
  pl[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < pairs) {
    float v = 0.f;
    for (int k = 0; k < Q; ++k) v += pl[k * (256 / Q) + threadIdx.x];
    part[((size_t)n * g.C + c0) * KK + threadIdx.x] = v;
  }
}

template <class K> void head_set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int sg_conv2d_head_supported(const sgConvDesc* d) { return head_ok(d) ? 1 : 0; }

extern "C" size_t sg_conv2d_head_ws_bytes(const sgConvDesc* d) {
  if (!head_ok(d)) return 0;
  const HeadGeom g = head_geom(d);
  const size_t a = (size_t)d->N * g.chunks * d->OH * d->OW, b = (size_t)d->N * d->C1 * d->KS * d->KS;
  return (a > b ? a : b) * sizeof(float);
}

// 1x1 head with pad 0 (mask_net's Conv2d(192, 1, 1), generators.py:27): y[n][p] = act(b + sum_c w[c] x[n][c][p]) is a channel
// reduction of a stream -- no taps, no halo, nothing to stage.  Each thread owns four consecutive pixels and walks the channels
// with eight float4 loads in flight (channel c of the image is HW floats further); weights are wave-uniform scalar loads.  The
// LDS-staged head kernel ran this at 1.7 TB/s (480 us for the 830 MB of configs[4]'s 1056 masks); channels are added in ascending
// order, so the result is a plain fp32 sum over c.
__global__ void __launch_bounds__(256) head1x1_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float4* __restrict__ y, int C, int HW4,
                                                         int act, float slope) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (q >= HW4) return;
  const float4* xp = x + (size_t)n * C * HW4 + q;
  const float b = bias ? bias[0] : 0.f;
  float4 acc = make_float4(b, b, b, b);
  for (int c0 = 0; c0 < C; c0 += 8) {
    float4 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = c0 + e < C ? xp[(size_t)(c0 + e) * HW4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < C) {
        const float wc = w[c0 + e];
        acc.x = fmaf(wc, t[e].x, acc.x); acc.y = fmaf(wc, t[e].y, acc.y);
        acc.z = fmaf(wc, t[e].z, acc.z); acc.w = fmaf(wc, t[e].w, acc.w);
      }
  }
  acc.x = sg_apply_act(acc.x, act, slope); acc.y = sg_apply_act(acc.y, act, slope);
  acc.z = sg_apply_act(acc.z, act, slope); acc.w = sg_apply_act(acc.w, act, slope);
  y[(size_t)n * HW4 + q] = acc;
}

#define SG_HEAD_DISPATCH(KERNEL, ...)                                                                             \
  switch (d->KS) {                                                                                                \
    case 1: head_set_lds(KERNEL<1>, lds); hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
    case 3: head_set_lds(KERNEL<3>, lds); hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
    default: head_set_lds(KERNEL<4>, lds); hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), lds, s, __VA_ARGS__); break; \
  }

static double head_bytes(const sgConvDesc* d) {
  return 4.0 * ((double)d->N * d->C1 * d->H * d->W + (double)d->N * d->OH * d->OW);
}

extern "C" int sg_conv2d_head_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                                  float slope, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(head_ok(d), "sg_conv2d_head_fwd: unsupported desc (needs Cout == 1, stride 1, zero padding, KS in {1,3,4})");
  SG_ARG_CHECK(x && w && y && ws && ws_bytes >= sg_conv2d_head_ws_bytes(d), "sg_conv2d_head_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (d->KS == 1 && d->pad == 0 && (d->H * d->W) % 4 == 0 && d->N <= 65535 &&
      (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    SgProfScope prof(SG_K_HEAD, s, 2.0 * d->C1 * (double)d->N * d->OH * d->OW, head_bytes(d));
    const int HW4 = d->H * d->W / 4;
    hipLaunchKernelGGL(head1x1_fwd_kernel, dim3(sg_cdiv(HW4, 256), d->N), dim3(256), 0, s, reinterpret_cast<const float4*>(x), w,
                       bias, reinterpret_cast<float4*>(y), d->C1, HW4, act, slope);
    SG_LAUNCH_CHECK("sg_conv2d_head_fwd");
    return 0;
  }
  const HeadGeom g = head_geom(d);
  float* part = reinterpret_cast<float*>(ws);
  const dim3 grid(g.chunks, d->N);
  const size_t lds = ((size_t)g.CH * g.PH * g.PW + 4 * (size_t)d->OH * d->OW) * sizeof(float);
  SgProfScope prof(SG_K_HEAD, s, 2.0 * d->C1 * d->KS * d->KS * (double)d->N * d->OH * d->OW, head_bytes(d));
  SG_HEAD_DISPATCH(head_fwd_kernel, x, w, part, g)
  const int tot = d->N * d->OH * d->OW;
  hipLaunchKernelGGL(head_fwd_finish_kernel, dim3(sg_cdiv(tot, 256)), dim3(256), 0, s, (const float*)part, bias, y, d->N,
                     g.chunks, d->OH * d->OW, act, slope);
  SG_LAUNCH_CHECK("sg_conv2d_head_fwd");
  return 0;
}

extern "C" int sg_conv2d_head_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, sgStream stream) {
  SG_ARG_CHECK(head_ok(d), "sg_conv2d_head_dgrad: unsupported desc");
  SG_ARG_CHECK(gy && w && gx, "sg_conv2d_head_dgrad: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const HeadGeom g = head_geom(d);
  const dim3 grid(g.chunks, d->N);
  const int E = d->KS - 1;
  const size_t lds = ((size_t)(d->OH + 2 * E) * (d->OW + 2 * E) + (size_t)g.CH * d->KS * d->KS) * sizeof(float);
  SgProfScope prof(SG_K_HEAD, s, 2.0 * d->C1 * d->KS * d->KS * (double)d->N * d->OH * d->OW, head_bytes(d));
  SG_HEAD_DISPATCH(head_dgrad_kernel, gy, w, gx, g)
  SG_LAUNCH_CHECK("sg_conv2d_head_dgrad");
  return 0;
}

extern "C" int sg_conv2d_head_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                                    sgStream stream) {
  SG_ARG_CHECK(head_ok(d), "sg_conv2d_head_wgrad: unsupported desc");
  SG_ARG_CHECK(gy && x && gw && ws && ws_bytes >= sg_conv2d_head_ws_bytes(d), "sg_conv2d_head_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const HeadGeom g = head_geom(d);
  float* part = reinterpret_cast<float*>(ws);
  const dim3 grid(g.chunks, d->N);
  // the staged planes are reused as a 256-float reduction buffer at the end of the kernel: tiny maps (a 1x1 head on a
  // < 16-pixel map, a 3x3 head on a 1x1 map) stage fewer than 256 floats -- never allocate less
  size_t lds_floats = (size_t)g.CH * g.PH * g.PW + 2 * (size_t)d->OH * d->OW;
  if (lds_floats < 256) lds_floats = 256;
  const size_t lds = lds_floats * sizeof(float);
  const size_t nw = (size_t)d->C1 * d->KS * d->KS;
  {
    SgProfScope prof(SG_K_HEAD, s, 2.0 * d->C1 * d->KS * d->KS * (double)d->N * d->OH * d->OW, head_bytes(d));
    SG_HEAD_DISPATCH(head_wgrad_kernel, gy, x, part, g)
  }
  hipLaunchKernelGGL(smallm_reduce_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, (const float*)part, gw, nw, d->N);
  SG_LAUNCH_CHECK("sg_conv2d_head_wgrad");
  return 0;
}
