// Shared helpers for the gfx950 kernels of libsg2im_hip.so (see include/sg2im_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sg2im_hip.h"

#define SG_WAVE 64

void sg_set_error(const char* fmt, ...);

#define SG_ARG_CHECK(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      sg_set_error(__VA_ARGS__);                \
      return -1;                                \
    }                                           \
  } while (0)

#define SG_LAUNCH_CHECK(name)                                               \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      sg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

// ---- profiler kinds (bench.py roofline leg) -------------------------------------------------
// ids 0..47: one per igemm instantiation family: (gather family f, kernel size index k, tile t) -> f*16 + k*4 + t
//   f: 0 = K=(c,taps) conv-style gather   (conv fwd, convT dgrad)      -> "igemm_kn0_*"
//      1 = K=(c,taps) transposed gather   (conv dgrad, convT fwd)      -> "igemm_kn1_*"
//      2 = K=(img,pix) weight-gradient    (conv / convT wgrad)         -> "igemm_nk_*"
//   k: KS in {1,3,4,7};  t: tile {128x128, 64x64, 32x128, 64x128}
enum {
  SG_K_IGEMM_BASE = 0, SG_K_IGEMM_COUNT = 48,
  SG_K_LINEAR = 48, SG_K_LAYOUT_FWD, SG_K_LAYOUT_BWD, SG_K_INSTNORM, SG_K_BATCHNORM, SG_K_ADAM, SG_K_SEGSUM,
  SG_K_CROP, SG_K_OTHER,
  // the batched dense GEMMs of the Winograd convs, one kind per template instantiation (they used to be lumped into
  // igemm_kn0_k3_t128 together with the direct 3x3 convs) and the elementwise Winograd transforms
  SG_K_WINO_GEMM_128, SG_K_WINO_GEMM_64, SG_K_WINO_XFORM,
  SG_K_HEAD,          // single-output-channel convolutions on the vector ALUs (smallm.hip), HBM-bound
  SG_K_INSTNORM_BWD,
  SG_K_WINO24_GEMM,   // the 25 (x k-chunks) batched dense GEMMs of a Winograd F(2x2,4x4) conv (same instantiation as SG_K_WINO_GEMM_128)
  SG_K_COUNT
};
static inline int sg_igemm_kind(int family, int KS, int tile) {
  const int k = KS == 1 ? 0 : (KS == 3 ? 1 : (KS == 4 ? 2 : 3));
  return SG_K_IGEMM_BASE + family * 16 + k * 4 + tile;
}
extern int g_sg_prof_on;
void sg_prof_begin(int kind, hipStream_t s);
void sg_prof_end(int kind, hipStream_t s, double flops, double bytes);

struct SgProfScope {
  int kind; hipStream_t s; double flops, bytes; bool on;
  SgProfScope(int k, hipStream_t st, double f, double b) : kind(k), s(st), flops(f), bytes(b), on(g_sg_prof_on != 0) {
    if (on) sg_prof_begin(kind, s);
  }
  ~SgProfScope() { if (on) sg_prof_end(kind, s, flops, bytes); }
};

// n / d for 0 <= n < 2^31 as one v_mul_hi + shift (a hardware-less integer division costs ~25 VALU instructions)
struct FastDiv {
  unsigned m, s, d;
  FastDiv() : m(0), s(0), d(1) {}
  explicit FastDiv(unsigned dd) : m(0), s(0), d(dd) {
    if (dd > 1) {
      unsigned sh = 0;
      while ((1u << sh) < dd) ++sh;
      m = (unsigned)((((uint64_t)1) << (31 + sh)) / dd + 1);
      s = sh - 1;
    }
  }
  __device__ __forceinline__ unsigned div(unsigned n) const { return d == 1 ? n : (__umulhi(n, m) >> s); }
};

static inline int sg_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float sg_apply_act(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_LEAKY: return v > 0.f ? v : v * slope;
    case SG_ACT_TANH: return tanhf(v);
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// wave64 reductions via cross-lane shuffles (no LDS)
__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block reduction of up to 1024 threads; result valid in every thread. `red` = >=16 floats of LDS.
__device__ __forceinline__ float sg_block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = sg_wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];   // fixed order => deterministic
  return t;
}
