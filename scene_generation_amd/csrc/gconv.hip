// Fused GraphTripleConv forward (reference graph.py:79-122): two launches per layer instead of seven.
//
//   sg_gconv_net1_fwd : rows [obj[s_t] | pred[t] | obj[o_t]] gathered straight into LDS (graph.py:79-84)
//                       -> GEMM (K1 x H) + bias + ReLU, the (32 x H) hidden block stays in LDS (graph.py:85, net1[0:2])
//                       -> GEMM (H x (2H+Dout)) + bias + ReLU -> new_t (net1[2:4]); the s / p / o split of graph.py:89-91 is
//                          a column range of new_t that the pool and the next layer read in place
//   sg_gconv_net2_fwd : deterministic segmented pool of new_s / new_o over the destination-major CSR (graph.py:94-116, the
//                       reference's CPU scatter_add order: bit-exact given new_t) straight into LDS
//                       -> GEMM (H x H) + ReLU -> LDS -> GEMM (H x Dout) + ReLU (graph.py:120)
//
// One workgroup = 32 rows (triples / nodes) x one 128-column block of the second GEMM; the first GEMM of a row block is
// recomputed by each of its column blocks (9 for net1 at H = 512): the launch is latency-bound (T = 512 triples are 16 row
// blocks), not flop-bound, and the recompute keeps 144 CUs busy instead of 16.  MFMA: v_mfma_f32_32x32x2_f32 with the same
// k pairing and k order as igemm_core.h, so the results are the ones the unfused path (sg_linear_fwd) produces.
// The intermediates autograd needs (cur_t, the hidden blocks, pooled) are written once, by column block 0.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GK = 16;          // k-tile depth
constexpr int GLDK = GK + 4;    // LDS pitch of a streamed weight tile row (conflict-free b128 fragment reads)
constexpr int GRB = 32;         // rows per workgroup

__device__ __forceinline__ float relu(float v) { return v > 0.f ? v : 0.f; }

// Streams the k-tile [k0, k0+16) of weight rows [n0, n0+NR) (W is [N][K], k contiguous) into registers / LDS.
template <int NR>
struct WTile {
  static constexpr int PASSES = NR / 64;
  float4 r[PASSES];
  __device__ __forceinline__ void load(const float* __restrict__ W, int K, int N, int n0, int k0, bool vec, int tid) {
    const int xr = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int n = n0 + xr + 64 * p, k = k0 + kq;
      const int nn = n < N ? n : N - 1;                       // rows beyond N feed accumulator columns nobody stores
      const float* src = W + (size_t)nn * K + k;
      if (vec && k + 3 < K) {
        r[p] = *reinterpret_cast<const float4*>(src);
      } else {
        r[p].x = k + 0 < K ? src[0] : 0.f; r[p].y = k + 1 < K ? src[1] : 0.f;
        r[p].z = k + 2 < K ? src[2] : 0.f; r[p].w = k + 3 < K ? src[3] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ B, int tid) const {
    const int xr = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) *reinterpret_cast<float4*>(B + (xr + 64 * p) * GLDK + kq) = r[p];
  }
};

// acc[j] (j < NT tiles of 32 columns starting at column wcol of the streamed tile) += X[32 rows][k-tile] * B^T
template <int NT>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[NT], const float* __restrict__ Xrow, const float* __restrict__ B, int wcol,
                                         int lr, int lk) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 a = *reinterpret_cast<const float4*>(Xrow + lk * 8 + h * 4);
    float4 b[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(B + (wcol + 32 * j + lr) * GLDK + lk * 8 + h * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float av = e == 0 ? a.x : (e == 1 ? a.y : (e == 2 ? a.z : a.w));
        const float bv = e == 0 ? b[j].x : (e == 1 ? b[j].y : (e == 2 ? b[j].z : b[j].w));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
      }
  }
}

// The two chained GEMMs on the 32-row block held in Xs (pitch PX floats, KA valid columns, zero padded to a multiple of 16).
//   stage a: hid = relu(Xs * Wa^T + ba)  (H columns, wave w owns columns [w*H/4, (w+1)*H/4)), written back INTO Xs
//   stage b: out[:, c0:c0+128] = relu(hid * Wb^T + bb)
template <int HT>
__device__ __forceinline__ void mlp2_block(float* __restrict__ Xs, int PX, int KA, float* __restrict__ Bs,
                                           const float* __restrict__ Wa, const float* __restrict__ ba, bool veca,
                                           const float* __restrict__ Wb, const float* __restrict__ bb, int Nb, int c0,
                                           float* __restrict__ hid_out, float* __restrict__ out, int ldo, int row0, int rows) {
  constexpr int H = HT * 128;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, lk = lane >> 5;
  // ---- stage a -------------------------------------------------------------------------------------------------
  {
    f32x16 acc[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    WTile<H> wt;
    const int ktiles = (KA + GK - 1) / GK;
    wt.load(Wa, KA, H, 0, 0, veca, tid);
    wt.store(Bs, tid);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
      float* cur = Bs + (kt & 1) * H * GLDK;
      float* nxt = Bs + ((kt + 1) & 1) * H * GLDK;
      if (kt + 1 < ktiles) wt.load(Wa, KA, H, 0, (kt + 1) * GK, veca, tid);
      mma_tile<HT>(acc, Xs + lr * PX + kt * GK, cur, wid * (H / 4), lr, lk);
      if (kt + 1 < ktiles) wt.store(nxt, tid);
      __syncthreads();
    }
    // every wave is past its last read of Xs (the barrier above): the hidden block overwrites it
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      const int col = wid * (H / 4) + 32 * j + lr;
      const float bias = ba ? ba[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const float v = relu(acc[j][r] + bias);
        Xs[row * PX + col] = v;
        if (hid_out && row < rows) hid_out[(size_t)(row0 + row) * H + col] = v;
      }
    }
  }
  __syncthreads();
  // ---- stage b -------------------------------------------------------------------------------------------------
  {
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    WTile<128> wt;
    constexpr int ktiles = H / GK;
    wt.load(Wb, H, Nb, c0, 0, true, tid);
    wt.store(Bs, tid);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
      float* cur = Bs + (kt & 1) * 128 * GLDK;
      float* nxt = Bs + ((kt + 1) & 1) * 128 * GLDK;
      if (kt + 1 < ktiles) wt.load(Wb, H, Nb, c0, (kt + 1) * GK, true, tid);
      mma_tile<1>(acc, Xs + lr * PX + kt * GK, cur, wid * 32, lr, lk);
      if (kt + 1 < ktiles) wt.store(nxt, tid);
      __syncthreads();
    }
    const int col = c0 + wid * 32 + lr;
    if (col < Nb) {
      const float bias = bb ? bb[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < rows) out[(size_t)(row0 + row) * ldo + col] = relu(acc[0][r] + bias);
      }
    }
  }
}

template <int HT>
__global__ void __launch_bounds__(256) gconv_net1_kernel(const float* __restrict__ obj, const float* __restrict__ pred, int ldp,
                                                        const int64_t* __restrict__ edges, int T, int Do, int Dp,
                                                        const float* __restrict__ W1, const float* __restrict__ b1, int vec1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2, int N2,
                                                        float* __restrict__ cur_t, float* __restrict__ h1,
                                                        float* __restrict__ new_t, int PX) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int H = HT * 128;
  float* Xs = lds;                       // [32][PX]
  float* Bs = lds + GRB * PX;            // [2][H][GLDK]
  const int tid = threadIdx.x, row0 = blockIdx.x * GRB, rows = min(GRB, T - row0), c0 = blockIdx.y * 128;
  const int K1 = 2 * Do + Dp, Kp = (K1 + GK - 1) / GK * GK;
  const bool first = blockIdx.y == 0;
  // gather the rows (a wave per row, lanes along the columns: coalesced segments)
  for (int r = tid >> 6; r < GRB; r += 4) {
    const int t = row0 + r;
    const bool live = t < T;
    const int64_t s = live ? edges[2 * t] : 0, o = live ? edges[2 * t + 1] : 0;
    const float* so = obj + (size_t)s * Do;
    const float* oo = obj + (size_t)o * Do;
    const float* pp = pred + (size_t)(live ? t : 0) * ldp;
    for (int c = tid & 63; c < Kp; c += 64) {
      float v = 0.f;
      if (live && c < K1) v = c < Do ? so[c] : (c < Do + Dp ? pp[c - Do] : oo[c - Do - Dp]);
      Xs[r * PX + c] = v;
      if (first && cur_t && live && c < K1) cur_t[(size_t)t * K1 + c] = v;
    }
  }
  __syncthreads();
  mlp2_block<HT>(Xs, PX, K1, Bs, W1, b1, vec1 != 0, W2, b2, N2, c0, first ? h1 : nullptr, new_t, N2, row0, rows);
}

constexpr int PASS_SHIFT = 30;          // csr entries: t | pass << 30 (graph.hip)

template <int HT>
__global__ void __launch_bounds__(256) gconv_net2_kernel(const float* __restrict__ new_t, int ld, int col_s, int col_o,
                                                        const int32_t* __restrict__ off, const int32_t* __restrict__ ent, int O,
                                                        int avg, const float* __restrict__ W3, const float* __restrict__ b3,
                                                        const float* __restrict__ W4, const float* __restrict__ b4, int Dout,
                                                        float* __restrict__ pooled, float* __restrict__ h2,
                                                        float* __restrict__ out, int PX) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int H = HT * 128;
  float* Xs = lds;
  float* Bs = lds + GRB * PX;
  const int tid = threadIdx.x, lane = tid & 63, row0 = blockIdx.x * GRB, rows = min(GRB, O - row0), c0 = blockIdx.y * 128;
  const bool first = blockIdx.y == 0;
  // pooled rows: sequential fp32 adds over the node's (pass, t)-ordered entries == the reference's scatter_add order
  for (int r = tid >> 6; r < GRB; r += 4) {
    const int i = row0 + r;
    const bool live = i < O;
    const int beg = live ? off[i] : 0, end = live ? off[i + 1] : 0;
    const float denom = (float)(end - beg > 1 ? end - beg : 1);
#pragma unroll
    for (int m = 0; m < H / 64; ++m) {
      const int c = lane + 64 * m;
      float acc = 0.f;
      for (int e = beg; e < end; ++e) {
        const int v = ent[e];
        const int t = v & ((1 << PASS_SHIFT) - 1);
        acc += new_t[(size_t)t * ld + ((v >> PASS_SHIFT) ? col_o : col_s) + c];
      }
      acc = avg ? acc / denom : acc;
      Xs[r * PX + c] = acc;
      if (first && pooled && live) pooled[(size_t)i * H + c] = acc;
    }
  }
  __syncthreads();
  mlp2_block<HT>(Xs, PX, H, Bs, W3, b3, true, W4, b4, Dout, c0, first ? h2 : nullptr, out, Dout, row0, rows);
}

inline int gconv_px(int K1, int H) {
  const int kp = (K1 + GK - 1) / GK * GK;
  return (kp > H ? kp : H) + 4;            // pitch = 16 m + 4: conflict-free ds_read_b128 over 16 rows
}
inline size_t gconv_lds(int K1, int H) { return ((size_t)GRB * gconv_px(K1, H) + 2 * (size_t)H * GLDK) * sizeof(float); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int sg_gconv_fused_supported(int Do, int Dp, int H, int Dout) {
  if (H != 128 && H != 256 && H != 512) return 0;
  if (Do <= 0 || Dp <= 0 || Dout <= 0 || (H % 4) != 0) return 0;
  return gconv_lds(2 * Do + Dp, H) <= 160 * 1024 ? 1 : 0;
}

#define SG_GCONV_LAUNCH(KERNEL, ...)                                                                            \
  do {                                                                                                          \
    switch (H / 128) {                                                                                          \
      case 1: hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), lds, s, __VA_ARGS__); break;                            \
      case 2: hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), lds, s, __VA_ARGS__); break;                            \
      default: hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), lds, s, __VA_ARGS__); break;                            \
    }                                                                                                           \
  } while (0)

extern "C" int sg_gconv_net1_fwd(const float* obj, const float* pred, int pred_ld, const int64_t* edges, int T, int Do, int Dp,
                                 const float* w1, const float* b1, int H, const float* w2, const float* b2, int N2,
                                 float* cur_t, float* h1, float* new_t, sgStream stream) {
  SG_ARG_CHECK(obj && pred && edges && w1 && w2 && new_t && T > 0 && N2 > 0 && pred_ld >= Dp,
               "sg_gconv_net1_fwd: bad arguments");
  SG_ARG_CHECK(sg_gconv_fused_supported(Do, Dp, H, 1), "sg_gconv_net1_fwd: unsupported dims (hidden width must be 128 / 256 / 512)");
  SG_ARG_CHECK(aligned16(w2), "sg_gconv_net1_fwd: w2 must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int K1 = 2 * Do + Dp, PX = gconv_px(K1, H);
  const size_t lds = gconv_lds(K1, H);
  const int vec1 = (K1 % 4 == 0 && aligned16(w1)) ? 1 : 0;
  const dim3 grid(sg_cdiv(T, GRB), sg_cdiv(N2, 128));
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * T * ((double)K1 * H + (double)H * N2), 0);
  SG_GCONV_LAUNCH(gconv_net1_kernel, obj, pred, pred_ld, edges, T, Do, Dp, w1, b1, vec1, w2, b2, N2, cur_t, h1, new_t, PX);
  SG_LAUNCH_CHECK("sg_gconv_net1_fwd");
  return 0;
}

extern "C" int sg_gconv_net2_fwd(const float* new_t, int ld, int col_s, int col_o, const int32_t* csr_off, const int32_t* csr_ent,
                                 int O, int avg, const float* w3, const float* b3, int H, const float* w4, const float* b4,
                                 int Dout, float* pooled, float* h2, float* out, sgStream stream) {
  SG_ARG_CHECK(new_t && csr_off && csr_ent && w3 && w4 && out && O > 0 && Dout > 0, "sg_gconv_net2_fwd: bad arguments");
  SG_ARG_CHECK(sg_gconv_fused_supported(1, 1, H, Dout), "sg_gconv_net2_fwd: unsupported dims (hidden width must be 128 / 256 / 512)");
  SG_ARG_CHECK(aligned16(w3) && aligned16(w4), "sg_gconv_net2_fwd: weights must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int PX = gconv_px(H, H);
  const size_t lds = gconv_lds(H, H);
  const dim3 grid(sg_cdiv(O, GRB), sg_cdiv(Dout, 128));
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * O * ((double)H * H + (double)H * Dout), 0);
  SG_GCONV_LAUNCH(gconv_net2_kernel, new_t, ld, col_s, col_o, csr_off, csr_ent, O, avg, w3, b3, w4, b4, Dout, pooled, h2, out, PX);
  SG_LAUNCH_CHECK("sg_gconv_net2_fwd");
  return 0;
}
