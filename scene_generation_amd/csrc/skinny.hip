// Register-streaming MFMA GEMM for the SMALL dense layers of the path: the graph-convolution MLPs (graph.py:58-122: a few
// hundred object / triple rows x 128..1152 features), box_net, repr_net, the classifier heads of the object discriminator.
//
//   C[M][N] = act( sum_k A(m,k) * B(n,k) + bias[n] )
//
// Why a second GEMM kernel: at M ~ 200 rows the LDS-tiled kernel (igemm_core.h) launches 30-60 workgroups, each walking K in a
// chain of dependent  global load -> LDS store -> barrier -> LDS read -> MFMA  rounds: 15-20 us per launch at ~6 TFLOP/s, 97
// launches per step.  These GEMMs are latency problems, not throughput problems, so this kernel maximises parallelism and
// removes every dependent round trip it can:
//   * one workgroup (4 waves) per 32x32 output tile; the four waves SPLIT K between them (16-deep chunks dealt out evenly),
//   * no LDS and no barrier in the k-loop: every lane fetches its own MFMA operands straight from global memory / L2 (the
//     operands of a 32x32x2 f32 MFMA are one value per lane: lane (r, h) supplies row r, k-index h), 8 consecutive k per
//     lane and chunk, with a whole round of 4 chunks (64 values per lane) in flight before the first MFMA and the next round
//     issued ahead of the current round's MFMAs,
//   * the four partial 32x32 accumulators are added through LDS in a FIXED order (wave 0, 1, 2, 3) -- deterministic, no
//     atomics -- and every wave finishes a quarter of the rows (bias, activation, coalesced 128-byte row stores).
// Each 32x32 tile re-reads its operand rows from L2 (8 FLOP per byte), which is why this form is for small M only: the
// caller (igemm.hip: run_dense) routes by size.
//
// Operand forms: K-contiguous rows, elem(x, k) = p[x*ld + k]  (16- / 8- / 4-byte loads by alignment), or
// X-contiguous,  elem(x, k) = p[k*ld + x]  (4-byte loads, coalesced across the 32 lanes of a half-wave).
// All loads are raw buffer loads: rows beyond X, k beyond K and the tail of the last chunk carry an offset the hardware range
// check rejects (-> 0), so there is no masking code in the loop.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
// raw buffer resource over the whole 2^31-byte window behind p (as igemm_core.h: sg_rsrc)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x80000000u, 0x00020000);
}

struct SkOperand { const float* p; int ld; int X; };
struct SkGather;

constexpr unsigned SK_INVALID = 0x80000000u;     // byte offset beyond num_records (2^31): the buffer load returns 0

// VEC: 4 / 2 / 1 = K-contiguous operand fetched as dwordx4 / dwordx2 / dword; 0 = X-contiguous operand
template <int VEC>
struct SkLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned base;        // byte offset of (row, k = 8h) [K-contiguous] or of column x [X-contiguous]; SK_INVALID: row outside
  unsigned ldb;         // X-contiguous: row pitch in bytes
  int K, h;
  __device__ __forceinline__ void init(const SkOperand& o, const SkGather&, int x0, int lane, int Ktot) {
    rsrc = sg_rsrc(o.p);
    const int r = lane & 31;
    h = lane >> 5;
    K = Ktot;
    const int x = x0 + r;
    const bool ok = x < o.X;
    if (VEC > 0) base = ok ? ((unsigned)x * (unsigned)o.ld + 8u * (unsigned)h) * 4u : SK_INVALID;
    else base = ok ? (unsigned)x * 4u : SK_INVALID;
    ldb = (unsigned)o.ld * 4u;
  }
  // the 8 values k = 16c + 8h + j, j = 0..7, of this lane's row
  __device__ __forceinline__ void load(float (&v)[8], int c) const {
    const int kb = 16 * c + 8 * h;
    if (VEC == 4) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool ok = base != SK_INVALID && kb + 4 * q < K;          // K % 4 == 0: the whole vector is inside or outside
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? base + (unsigned)(64 * c + 16 * q) : SK_INVALID), 0, 0);
        __builtin_memcpy(&v[4 * q], &t, 16);
      }
    } else if (VEC == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = base != SK_INVALID && kb + 2 * q < K;          // K % 2 == 0
        const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(ok ? base + (unsigned)(64 * c + 8 * q) : SK_INVALID), 0, 0);
        __builtin_memcpy(&v[2 * q], &t, 8);
      }
    } else if (VEC == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = base != SK_INVALID && kb + q < K;
        v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(ok ? base + (unsigned)(64 * c + 4 * q) : SK_INVALID), 0, 0));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = base != SK_INVALID && kb + q < K;
        v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(ok ? base + (unsigned)(kb + q) * ldb : SK_INVALID), 0, 0));
      }
    }
  }
};

// A operand of the first per-edge MLP layer of GraphTripleConv (graph.py:79-86): row t of the (T, 2 Do + Dp) matrix
// [obj[s_t] | pred[t] | obj[o_t]] read straight from the node / edge feature rows -- the concatenated matrix is never written.
// A lane owns one triple: its subject / object node ids are fetched once (init) and become two row offsets into ``obj``; every
// 8-value run of the k loop then comes from one of the three rows.  obj and pred are different allocations, i.e. two buffer
// descriptors: each run issues one load against each with the offset of the OTHER segment set outside the descriptor's range
// (-> 0) and ORs the bit patterns, so there is no divergent control flow and no per-element address select in 64 bits.
// VEC = 4: Do, Dp multiples of 4 and 16-byte aligned rows (the 128-dim layers: a 4-run never straddles segments); VEC = 1:
// anything (the first layer's 163-dim object rows).
struct SkGather { const float* obj; const float* pred; const int64_t* edges; int Do, Dp, T; };
template <int VEC>
struct SkLoaderGather {
  __amdgpu_buffer_rsrc_t robj, rpred;
  unsigned b0, b1, b2;      // byte offsets of the rows obj[s], pred[t], obj[o]
  int K, h, Do, Dp;
  bool ok;
  __device__ __forceinline__ void init(const SkOperand&, const SkGather& g, int x0, int lane, int Ktot) {
    robj = sg_rsrc(g.obj);
    rpred = sg_rsrc(g.pred);
    h = lane >> 5;
    K = Ktot; Do = g.Do; Dp = g.Dp;
    const int x = x0 + (lane & 31);
    ok = x < g.T;
    const long long sidx = ok ? g.edges[2 * (size_t)x] : 0, oidx = ok ? g.edges[2 * (size_t)x + 1] : 0;
    b0 = (unsigned)sidx * (unsigned)Do * 4u;
    b1 = (unsigned)x * (unsigned)Dp * 4u;
    b2 = (unsigned)oidx * (unsigned)Do * 4u;
  }
  __device__ __forceinline__ void offsets(int k, unsigned& offo, unsigned& offp) const {
    const bool in = ok && k < K;
    const bool seg0 = k < Do, seg1 = !seg0 && k < Do + Dp;
    offo = (in && !seg1) ? (seg0 ? b0 + 4u * (unsigned)k : b2 + 4u * (unsigned)(k - Do - Dp)) : SK_INVALID;
    offp = (in && seg1) ? b1 + 4u * (unsigned)(k - Do) : SK_INVALID;
  }
  __device__ __forceinline__ void load(float (&v)[8], int c) const {
    const int kb = 16 * c + 8 * h;
    if (VEC == 4) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned offo, offp;
        offsets(kb + 4 * q, offo, offp);
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(robj, (int)offo, 0, 0);
        const auto b = __builtin_amdgcn_raw_buffer_load_b128(rpred, (int)offp, 0, 0);
        unsigned ua[4], ub[4];
        __builtin_memcpy(ua, &a, 16);
        __builtin_memcpy(ub, &b, 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = __builtin_bit_cast(float, ua[e] | ub[e]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned offo, offp;
        offsets(kb + q, offo, offp);
        const unsigned a = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(robj, (int)offo, 0, 0);
        const unsigned b = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rpred, (int)offp, 0, 0);
        v[q] = __builtin_bit_cast(float, a | b);
      }
    }
  }
};
// AV: 0 / 1 / 2 / 4 = the plain forms above; 5 / 6 = gathered triple rows, 16-byte / 4-byte loads
template <int AV> struct SkALoader { using type = SkLoader<AV>; };
template <> struct SkALoader<5> { using type = SkLoaderGather<4>; };
template <> struct SkALoader<6> { using type = SkLoaderGather<1>; };

constexpr int SK_ROUND = 4;          // chunks per round: 4 x 16 k per wave in flight per operand (64 VGPRs per operand and buffer)

template <int AV, int BV>
__global__ void __launch_bounds__(256) skinny_gemm_kernel(SkOperand A, SkOperand B, float* __restrict__ C, const float* __restrict__ bias,
                                                          float* __restrict__ rowsum, int M, int N, int K, int act, float slope,
                                                          SkGather GA) {
  __shared__ float part[4][16][64];                   // partial accumulators [wave][register][lane]: 16 KB, conflict-free
  __shared__ float rpart[4][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  // 16-deep chunks dealt out evenly: wave w owns [c_beg, c_end)
  const int chunks = (K + 15) >> 4;
  const int per = chunks >> 2, extra = chunks & 3;
  const int c_beg = w * per + min(w, extra), c_end = c_beg + per + (w < extra ? 1 : 0);

  typename SkALoader<AV>::type la;
  SkLoader<BV> lb;
  la.init(A, GA, m0, lane, K);
  lb.init(B, GA, n0, lane, K);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // optional: rowsum[m] = sum_k A(m, k) (the bias gradient when A is gy of a weight-gradient GEMM), by the workgroups of the
  // first column tile, from the operand values they hold anyway
  const bool rs = rowsum != nullptr && blockIdx.x == 0;
  float racc = 0.f;

  float a0[SK_ROUND][8], b0[SK_ROUND][8], a1[SK_ROUND][8], b1[SK_ROUND][8];
  auto load_round = [&](float (&a)[SK_ROUND][8], float (&b)[SK_ROUND][8], int c) {
#pragma unroll
    for (int u = 0; u < SK_ROUND; ++u) {
      // chunks beyond c_end read k >= 16*c_end, which either belongs to the next wave (and must not be added twice) or lies
      // beyond K: force them outside
      const int cu = c + u < c_end ? c + u : (1 << 26);
      la.load(a[u], cu);
      lb.load(b[u], cu);
    }
  };
  auto mma_round = [&](const float (&a)[SK_ROUND][8], const float (&b)[SK_ROUND][8]) {
#pragma unroll
    for (int u = 0; u < SK_ROUND; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], b[u][j], acc, 0, 0, 0);
    if (rs) {
#pragma unroll
      for (int u = 0; u < SK_ROUND; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) racc += a[u][j];
    }
  };
  if (c_beg < c_end) {
    load_round(a0, b0, c_beg);
    for (int c = c_beg; c < c_end; c += 2 * SK_ROUND) {
      if (c + SK_ROUND < c_end) load_round(a1, b1, c + SK_ROUND);
      mma_round(a0, b0);
      if (c + SK_ROUND < c_end) {
        if (c + 2 * SK_ROUND < c_end) load_round(a0, b0, c + 2 * SK_ROUND);
        mma_round(a1, b1);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[w][r][lane] = acc[r];
  if (rs) {
    racc += __shfl_xor(racc, 32, 64);                 // the two k-halves of a row
    if (lane < 32) rpart[w][lane] = racc;
  }
  __syncthreads();
  if (rs && w == 0 && lane < 32 && m0 + lane < M)
    rowsum[m0 + lane] = ((rpart[0][lane] + rpart[1][lane]) + rpart[2][lane]) + rpart[3][lane];
  // wave w finishes accumulator registers 4w .. 4w+3: rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = i + 8w + 4h, column lane & 31
  const int n = n0 + (lane & 31);
  const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * w + i;
    const float v = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane];
    const int m = m0 + i + 8 * w + 4 * (lane >> 5);
    if (m < M && n < N) C[(size_t)m * N + n] = sg_apply_act(v + bv, act, slope);
  }
}

template <int AV>
void launch_b(int bv, const SkOperand& A, const SkOperand& B, float* C, const float* bias, float* rowsum, int M, int N, int K,
              int act, float slope, hipStream_t s, const SkGather& G = SkGather{nullptr, nullptr, nullptr, 0, 0, 0}) {
  const dim3 grid(sg_cdiv(N, 32), sg_cdiv(M, 32));
  switch (bv) {
    case 4: hipLaunchKernelGGL((skinny_gemm_kernel<AV, 4>), grid, dim3(256), 0, s, A, B, C, bias, rowsum, M, N, K, act, slope, G); break;
    case 2: hipLaunchKernelGGL((skinny_gemm_kernel<AV, 2>), grid, dim3(256), 0, s, A, B, C, bias, rowsum, M, N, K, act, slope, G); break;
    case 1: hipLaunchKernelGGL((skinny_gemm_kernel<AV, 1>), grid, dim3(256), 0, s, A, B, C, bias, rowsum, M, N, K, act, slope, G); break;
    default:
      if constexpr (AV < 5)      // (the gathered forms only pair with K-contiguous weights)
        hipLaunchKernelGGL((skinny_gemm_kernel<AV, 0>), grid, dim3(256), 0, s, A, B, C, bias, rowsum, M, N, K, act, slope, G);
      break;
  }
}

// widest load a K-contiguous operand allows: every row start and every 8-value run must be aligned to it
inline int vec_of(const float* p, int ld, int K) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (a % 16 == 0 && ld % 4 == 0 && K % 4 == 0) return 4;
  if (a % 8 == 0 && ld % 2 == 0 && K % 2 == 0) return 2;
  return 1;
}
}  // namespace

namespace sgk {
// a_kcontig / b_kcontig: 1 = elem(x, k) = p[x*ld + k], 0 = elem(x, k) = p[k*ld + x]
int skinny_gemm(const float* a, int lda, int a_kcontig, const float* b, int ldb, int b_kcontig, float* c, const float* bias,
                float* rowsum, int M, int N, int K, int act, float slope, hipStream_t s) {
  const SkOperand A{a, lda, M}, B{b, ldb, N};
  const int av = a_kcontig ? vec_of(a, lda, K) : 0, bv = b_kcontig ? vec_of(b, ldb, K) : 0;
  switch (av) {
    case 4: launch_b<4>(bv, A, B, c, bias, rowsum, M, N, K, act, slope, s); break;
    case 2: launch_b<2>(bv, A, B, c, bias, rowsum, M, N, K, act, slope, s); break;
    case 1: launch_b<1>(bv, A, B, c, bias, rowsum, M, N, K, act, slope, s); break;
    default: launch_b<0>(bv, A, B, c, bias, rowsum, M, N, K, act, slope, s); break;
  }
  return 0;
}
// y[T][N] = act([obj[s_t] | pred[t] | obj[o_t]] W^T + bias): the first per-edge MLP layer of GraphTripleConv with the row
// gather in the A loader (W: [N][2 Do + Dp], K-contiguous)
int skinny_gemm_gather(const float* obj, const float* pred, const int64_t* edges, int T, int Do, int Dp, const float* w, float* c,
                       const float* bias, int N, int act, float slope, hipStream_t s) {
  const int K = 2 * Do + Dp;
  const SkOperand A{nullptr, K, T}, B{w, K, N};
  const SkGather G{obj, pred, edges, Do, Dp, T};
  const bool v4 = Do % 4 == 0 && Dp % 4 == 0 && (reinterpret_cast<uintptr_t>(obj) % 16 == 0) && (reinterpret_cast<uintptr_t>(pred) % 16 == 0);
  const int bv = vec_of(w, K, K);
  if (v4) launch_b<5>(bv, A, B, c, bias, nullptr, T, N, K, act, slope, s, G);
  else launch_b<6>(bv, A, B, c, bias, nullptr, T, N, K, act, slope, s, G);
  return 0;
}
}  // namespace sgk
