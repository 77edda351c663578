// GraphTripleConv indexing kernels (graph.py:79-116) + embeddings (model.py:131-132): coalesced row
// gathers and a deterministic destination-major segmented sum (no atomics) whose accumulation order
// equals the CPU scatter_add order of the reference (s-pass then o-pass, t ascending) => bit-exact pool.
#include "common.h"

namespace {

constexpr int PASS_SHIFT = 30;

__global__ void csr_count_kernel(const int64_t* __restrict__ edges, int T, int O, int32_t* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O) return;
  int c = 0;
  for (int t = 0; t < T; ++t) {
    c += (edges[2 * t] == i);
    c += (edges[2 * t + 1] == i);
  }
  cnt[i + 1] = c;
}

// single wave exclusive scan: off[0]=0, off[i+1] = sum_{j<=i} cnt[j+1]  (in place on off[1..O])
__global__ void csr_scan_kernel(int32_t* __restrict__ off, int O) {
  const int lane = threadIdx.x;
  int carry = 0;
  if (lane == 0) off[0] = 0;
  for (int base = 0; base < O; base += 64) {
    const int i = base + lane;
    int v = i < O ? off[i + 1] : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(v, d, 64);
      if (lane >= d) v += u;
    }
    if (i < O) off[i + 1] = v + carry;
    carry += __shfl(v, 63, 64);
  }
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ edges, int T, int O, const int32_t* __restrict__ off,
                                int32_t* __restrict__ ent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O) return;
  int w = off[i];
  for (int t = 0; t < T; ++t)
    if (edges[2 * t] == i) ent[w++] = t;
  for (int t = 0; t < T; ++t)
    if (edges[2 * t + 1] == i) ent[w++] = t | (1 << PASS_SHIFT);
}

// The whole CSR in ONE launch and O(T) work: a stable counting sort of the 2T (pass, t) entries by destination node, done by
// one workgroup with the per-node counters / cursors in LDS (O <= CSR_LDS_NODES; larger graphs use the three kernels above,
// whose node threads each scan all T triples).  Entries are placed 256 at a time in (pass, t) order; inside a chunk the
// rank of an entry among the chunk's earlier entries with the same destination comes from an LDS scan, so the order inside
// every node's list is exactly (pass, t) ascending -- the order the reference's CPU scatter_add applies the updates.
constexpr int CSR_LDS_NODES = 8192;
__global__ void __launch_bounds__(256) csr_build_kernel(const int64_t* __restrict__ edges, int T, int O, int32_t* __restrict__ off,
                                                       int32_t* __restrict__ ent) {
  __shared__ int cur[CSR_LDS_NODES];
  __shared__ int dch[256];
  __shared__ int carry_s;
  const int tid = threadIdx.x;
  for (int i = tid; i < O; i += 256) cur[i] = 0;
  __syncthreads();
  for (int e = tid; e < 2 * T; e += 256) {
    const int pass = e >= T, t = e - pass * T;
    atomicAdd(&cur[(int)edges[2 * t + pass]], 1);            // integer LDS atomics: the counts do not depend on the order
  }
  __syncthreads();
  // exclusive scan of the counts: cur[i] becomes the cursor (= start) of node i, off[] the CSR offsets
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < O; base += 256) {
    const int i = base + tid;
    const int c = i < O ? cur[i] : 0;
    dch[tid] = c;
    __syncthreads();
    int incl = c;
    for (int d = 1; d < 256; d <<= 1) {                     // Hillis-Steele over the chunk
      const int u = tid >= d ? dch[tid - d] : 0;
      __syncthreads();
      incl += u;
      dch[tid] = incl;
      __syncthreads();
    }
    const int carry = carry_s;
    if (i < O) { cur[i] = carry + incl - c; off[i] = carry + incl - c; }
    __syncthreads();
    if (tid == 255) carry_s = carry + incl;
    __syncthreads();
  }
  if (tid == 0) off[O] = carry_s;
  // stable placement, 256 entries per round in (pass, t) order.  Rank of an entry among the round's earlier entries with the same
  // destination: inside its wave from ballots (the wave walks its DISTINCT destinations: one ballot + popcount each, instead of
  // every thread scanning 255 LDS slots), across the round's four waves by letting them place one after the other, each
  // advancing the nodes' LDS cursors behind it.
  const int lane = tid & 63, wv = tid >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int base = 0; base < 2 * T; base += 256) {
    const int e = base + tid;
    const bool live = e < 2 * T;
    const int pass = live && e >= T, t = live ? e - pass * T : 0;
    const int d = live ? (int)edges[2 * t + pass] : -1;
    int rank = 0, cnt = 0;
    unsigned long long todo = __ballot(live);
    while (todo) {                                           // wave-uniform loop over the distinct destinations of the wave
      const int src = __ffsll((long long)todo) - 1;
      const int d0 = __shfl(d, src, 64);
      const unsigned long long same = __ballot(live && d == d0);
      if (d == d0) { rank = __popcll(same & below); cnt = __popcll(same); }
      todo &= ~same;
    }
    for (int w = 0; w < 4; ++w) {
      if (wv == w && live) {
        const int at = cur[d] + rank;
        ent[at] = t | (pass << PASS_SHIFT);
      }
      __syncthreads();
      if (wv == w && live && rank == cnt - 1) cur[d] += cnt;  // the wave's last entry of this node advances its cursor
      __syncthreads();
    }
  }
}

__global__ void gather_concat_kernel(const float* __restrict__ obj, const float* __restrict__ pred,
                                     const int64_t* __restrict__ edges, float* __restrict__ out, int T, int Do, int Dp) {
  const int t = blockIdx.x;
  const int64_t s = edges[2 * t], o = edges[2 * t + 1];
  const int width = 2 * Do + Dp;
  float* dst = out + (size_t)t * width;
  const float* so = obj + (size_t)s * Do;
  const float* oo = obj + (size_t)o * Do;
  const float* pp = pred + (size_t)t * Dp;
  for (int c = threadIdx.x; c < width; c += blockDim.x) {
    float v;
    if (c < Do) v = so[c];
    else if (c < Do + Dp) v = pp[c - Do];
    else v = oo[c - Do - Dp];
    dst[c] = v;
  }
}

// One workgroup per node.  The node's entry list (row index | pass bit) is staged in LDS once, 1024 entries at a time, by all threads --
// the old loop made every column thread walk  ent[e] -> address -> src[...]  as a chain of dependent loads, one round trip per
// entry and level -- and the source values of eight entries are in flight before the first add.  Same adds, same order.
constexpr int SEG_CAP = 1024;
__global__ void segment_sum_kernel(const float* __restrict__ src, int src_ld, int col0, int col1, int width,
                                   const int32_t* __restrict__ off, const int32_t* __restrict__ ent,
                                   float* __restrict__ dst, int avg) {
  __shared__ unsigned rowoff[SEG_CAP];            // element offset of (row t, column block) of every staged entry
  const int i = blockIdx.x;
  const int beg = off[i], end = off[i + 1];
  const float denom = (float)(end - beg > 1 ? end - beg : 1);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};            // columns threadIdx.x + {0, 1, 2, 3} * blockDim.x (width <= 4 * blockDim.x)
  const int ncol = (width + (int)blockDim.x - 1) / (int)blockDim.x;
  for (int e0 = beg; e0 < end; e0 += SEG_CAP) {
    const int n = end - e0 < SEG_CAP ? end - e0 : SEG_CAP;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const int v = ent[e0 + j];
      rowoff[j] = (unsigned)(v & ((1 << PASS_SHIFT) - 1)) * (unsigned)src_ld + (unsigned)((v >> PASS_SHIFT) ? col1 : col0);
    }
    __syncthreads();
    for (int q = 0; q < ncol && q < 4; ++q) {
      const int c = threadIdx.x + q * (int)blockDim.x;
      if (c >= width) break;
      float a = acc[q];
      for (int j0 = 0; j0 < n; j0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)rowoff[j0 + u < n ? j0 + u : n - 1] + c];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < n) a += t[u];                // sequential fp32 adds, fixed (pass, t) order
      }
      acc[q] = a;
    }
  }
  for (int q = 0; q < ncol && q < 4; ++q) {
    const int c = threadIdx.x + q * (int)blockDim.x;
    if (c < width) dst[(size_t)i * width + c] = avg ? acc[q] / denom : acc[q];
  }
}
// (more than 4 column blocks per thread: the plain loop)
__global__ void segment_sum_wide_kernel(const float* __restrict__ src, int src_ld, int col0, int col1, int width,
                                        const int32_t* __restrict__ off, const int32_t* __restrict__ ent,
                                        float* __restrict__ dst, int avg) {
  const int i = blockIdx.x;
  const int beg = off[i], end = off[i + 1];
  const float denom = (float)(end - beg > 1 ? end - beg : 1);
  for (int c = threadIdx.x; c < width; c += blockDim.x) {
    float acc = 0.f;
    for (int e = beg; e < end; ++e) {
      const int v = ent[e];
      const int t = v & ((1 << PASS_SHIFT) - 1);
      const int col = (v >> PASS_SHIFT) ? col1 : col0;
      acc += src[(size_t)t * src_ld + col + c];      // sequential fp32 adds, fixed order
    }
    dst[(size_t)i * width + c] = avg ? acc / denom : acc;
  }
}

__global__ void pool_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ gnp,
                                const int64_t* __restrict__ edges, const int32_t* __restrict__ off,
                                float* __restrict__ out, int H, int Dout, int avg) {
  const int t = blockIdx.x;
  const int64_t s = edges[2 * t], o = edges[2 * t + 1];
  float ds = 1.f, dobj = 1.f;
  if (avg) {
    const int cs = off[s + 1] - off[s], co = off[o + 1] - off[o];
    ds = (float)(cs > 1 ? cs : 1);
    dobj = (float)(co > 1 ? co : 1);
  }
  const int width = 2 * H + Dout;
  float* dst = out + (size_t)t * width;
  for (int c = threadIdx.x; c < width; c += blockDim.x) {
    float v;
    if (c < H) v = gp[(size_t)s * H + c] / ds;
    else if (c < H + Dout) v = gnp ? gnp[(size_t)t * Dout + (c - H)] : 0.f;
    else v = gp[(size_t)o * H + (c - H - Dout)] / dobj;
    dst[c] = v;
  }
}

__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                     float* __restrict__ out, int dim) {
  const int i = blockIdx.x;
  const float* src = table + (size_t)idx[i] * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) out[(size_t)i * dim + c] = src[c];
}

// one block per table row; rows of g accumulated in ascending i (== CPU index_add order).  The matching rows are listed
// once (wave 0, ascending) so the column loop only touches them: the scan over all n indices per column made this the
// slowest small kernel of the factored layout convs (dim = Cout*KS*KS = 3136).
__global__ void embedding_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                     float* __restrict__ gt, int n, int dim) {
  constexpr int CAP = 1024;
  __shared__ int hits[CAP];
  __shared__ int nhit;
  const int row = blockIdx.x;
  // blockIdx.y: the column range [col_lo, col_hi) of this workgroup (the table has only ~180 rows: one workgroup per row left a
  // third of the chip idle and took 288 us at configs[4]'s 1056 objects; every workgroup rebuilds the row's hit list -- cheap)
  const int cper = (dim + gridDim.y - 1) / gridDim.y, col_lo = blockIdx.y * cper, col_hi = min(dim, col_lo + cper);
  for (int base = 0; base < n; base += CAP) {            // chunks of CAP indices (one chunk in practice)
    __syncthreads();
    if (threadIdx.x < 64) {                            // wave 0: ordered compaction, 64 indices per round (one thread
      int c = 0;                                       // walking the n dependent loads took ~50 us at n = 250)
      const int end = base + CAP < n ? base + CAP : n;
      for (int i0 = base; i0 < end; i0 += 64) {
        const int i = i0 + (int)threadIdx.x;
        const bool hit = i < end && idx[i] == row;
        const unsigned long long m = __ballot(hit);
        if (hit) hits[c + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = i;
        c += __popcll(m);
      }
      if (threadIdx.x == 0) nhit = c;
    }
    __syncthreads();
    const int c = nhit;
    for (int col = col_lo + threadIdx.x; col < col_hi; col += blockDim.x) {
      float acc = base == 0 ? 0.f : gt[(size_t)row * dim + col];
      for (int h0 = 0; h0 < c; h0 += 8) {              // eight rows in flight, added in ascending i (the index_add order)
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = h0 + e < c ? g[(size_t)hits[h0 + e] * dim + col] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (h0 + e < c) acc += t[e];
      }
      gt[(size_t)row * dim + col] = acc;
    }
  }
  if (n == 0)
    for (int col = col_lo + threadIdx.x; col < col_hi; col += blockDim.x) gt[(size_t)row * dim + col] = 0.f;
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int src_ld, int src_off, float* __restrict__ dst,
                                 int dst_ld, int dst_off, int rows, int width) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * width) return;
  const int r = i / width, c = i - (size_t)r * width;
  dst[(size_t)r * dst_ld + dst_off + c] = src[(size_t)r * src_ld + src_off + c];
}

__global__ void one_hot_kernel(const int64_t* __restrict__ idx, float* __restrict__ out, int n, int classes, int ld,
                               int col_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * classes) return;
  const int r = i / classes, c = i - (size_t)r * classes;
  out[(size_t)r * ld + col_off + c] = (idx[r] == c) ? 1.f : 0.f;
}

// first position whose value is outside [lo, hi): atomicMin on the position => deterministic
__global__ void index_check_kernel(const int64_t* __restrict__ idx, int64_t n, int64_t lo, int64_t hi, int* __restrict__ first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v < lo || v >= hi) atomicMin(first, (int)(i < 0x7fffffff ? i : 0x7ffffffe));
}

inline int row_threads(int width) { int t = ((width + 63) / 64) * 64; return t > 256 ? 256 : (t < 64 ? 64 : t); }

}  // namespace

extern "C" int sg_check_indices(const int64_t* idx, int64_t n, int64_t lo, int64_t hi, const char* what, sgStream stream) {
  SG_ARG_CHECK(n >= 0 && (idx || n == 0), "sg_check_indices: bad arguments");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return 0;
  int* flag = nullptr;                                  // 4 bytes, allocated per call: this is a debugging path
  if (hipMalloc((void**)&flag, sizeof(int)) != hipSuccess) { sg_set_error("sg_check_indices: hipMalloc failed"); return -1; }
  int first = 0x7fffffff;
  hipMemcpyAsync(flag, &first, sizeof(int), hipMemcpyHostToDevice, s);
  hipLaunchKernelGGL(index_check_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, idx, n, lo, hi, flag);
  hipMemcpyAsync(&first, flag, sizeof(int), hipMemcpyDeviceToHost, s);
  hipError_t e = hipStreamSynchronize(s);
  long long bad = 0;
  if (e == hipSuccess && first != 0x7fffffff) {
    int64_t v = 0;
    hipMemcpy(&v, idx + first, sizeof(int64_t), hipMemcpyDeviceToHost);
    bad = (long long)v;
  }
  hipFree(flag);
  if (e != hipSuccess) { sg_set_error("sg_check_indices: %s", hipGetErrorString(e)); return (int)e; }
  if (first != 0x7fffffff) {
    sg_set_error("index out of range: %s[%d] = %lld is not in [%lld, %lld)", what ? what : "index", first, bad, (long long)lo,
                 (long long)hi);
    return SG_ERR_INDEX;
  }
  return 0;
}

int sg_check_indices_if_enabled(const int64_t* idx, int64_t n, int64_t lo, int64_t hi, const char* what, hipStream_t s) {
  return sg_opt(SG_OPT_CHECK_INDICES) ? sg_check_indices(idx, n, lo, hi, what, (sgStream)s) : 0;
}

extern "C" int sg_build_csr(const int64_t* edges, int T, int O, int32_t* csr_off, int32_t* csr_ent, sgStream stream) {
  SG_ARG_CHECK(edges && csr_off && csr_ent && T >= 0 && O > 0, "sg_build_csr: bad arguments");
  SG_ARG_CHECK(T < (1 << PASS_SHIFT), "sg_build_csr: too many triples");
  hipStream_t s = (hipStream_t)stream;
  // the (s, o) columns index LDS counters / object rows unchecked in every kernel that follows (graph.py:79-80 raises there)
  if (const int rc = sg_check_indices_if_enabled(edges, 2 * (int64_t)T, 0, O, "edges (subject / object node ids)", s)) return rc;
  if (O <= CSR_LDS_NODES) {
    hipLaunchKernelGGL(csr_build_kernel, dim3(1), dim3(256), 0, s, edges, T, O, csr_off, csr_ent);
    SG_LAUNCH_CHECK("sg_build_csr");
    return 0;
  }
  hipLaunchKernelGGL(csr_count_kernel, dim3(sg_cdiv(O, 64)), dim3(64), 0, s, edges, T, O, csr_off);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(64), 0, s, csr_off, O);
  hipLaunchKernelGGL(csr_fill_kernel, dim3(sg_cdiv(O, 64)), dim3(64), 0, s, edges, T, O, csr_off, csr_ent);
  SG_LAUNCH_CHECK("sg_build_csr");
  return 0;
}

extern "C" int sg_gather_concat_fwd(const float* obj, const float* pred, const int64_t* edges, float* out, int T, int Do,
                                    int Dp, sgStream stream) {
  SG_ARG_CHECK(obj && pred && edges && out && T >= 0 && Do > 0 && Dp > 0, "sg_gather_concat_fwd: bad arguments");
  if (T == 0) return 0;
  hipLaunchKernelGGL(gather_concat_kernel, dim3(T), dim3(row_threads(2 * Do + Dp)), 0, (hipStream_t)stream, obj, pred,
                     edges, out, T, Do, Dp);
  SG_LAUNCH_CHECK("sg_gather_concat_fwd");
  return 0;
}

namespace sgk {
int skinny_gemm_gather(const float* obj, const float* pred, const int64_t* edges, int T, int Do, int Dp, const float* w, float* c,
                       const float* bias, int N, int act, float slope, hipStream_t s);
}
static bool gconv_fused_shape(int T, int out_f) {
  return (long)sg_cdiv(T, 32) * sg_cdiv(out_f, 32) <= (long)sg_opt(SG_OPT_LINEAR_SKINNY) && sg_opt(SG_OPT_GCONV_FUSED_GATHER) != 0;
}
extern "C" size_t sg_gconv_gather_linear_ws_bytes(int T, int Do, int Dp, int out_f) {
  return gconv_fused_shape(T, out_f) ? 0 : (size_t)(T > 0 ? T : 1) * (2 * Do + Dp) * sizeof(float);
}
extern "C" int sg_gconv_gather_linear_fwd(const float* obj, const float* pred, const int64_t* edges, const float* w, const float* b,
                                          float* y, int T, int Do, int Dp, int out_f, int act, float slope, void* ws,
                                          size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(obj && pred && edges && w && y && T >= 0 && Do > 0 && Dp > 0 && out_f > 0, "sg_gconv_gather_linear_fwd: bad arguments");
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int K = 2 * Do + Dp;
  if (gconv_fused_shape(T, out_f)) {
    SgProfScope prof(SG_K_LINEAR, s, 2.0 * T * (double)K * out_f, 0);
    sgk::skinny_gemm_gather(obj, pred, edges, T, Do, Dp, w, y, b, out_f, act, slope, s);
    SG_LAUNCH_CHECK("sg_gconv_gather_linear_fwd");
    return 0;
  }
  // large graphs (more 32x32 output tiles than the register-streaming kernel takes): materialise the rows, LDS-tiled GEMM
  SG_ARG_CHECK(ws && ws_bytes >= (size_t)T * K * sizeof(float), "sg_gconv_gather_linear_fwd: workspace too small");
  if (const int rc = sg_gather_concat_fwd(obj, pred, edges, (float*)ws, T, Do, Dp, stream)) return rc;
  return sg_linear_fwd((const float*)ws, w, b, y, T, K, out_f, act, slope, stream);
}

extern "C" int sg_segment_sum(const float* src, int src_ld, int col_off0, int col_off1, int width, const int32_t* csr_off,
                              const int32_t* csr_ent, float* dst, int O, int avg, sgStream stream) {
  SG_ARG_CHECK(src && csr_off && csr_ent && dst && O > 0 && width > 0, "sg_segment_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_SEGSUM, s, 0, 0);
  const int threads = row_threads(width);
  // (32-bit element offsets in the staged form: the operand limit of the library, 2^29 elements, holds for src)
  if (width <= 4 * threads)
    hipLaunchKernelGGL(segment_sum_kernel, dim3(O), dim3(threads), 0, s, src, src_ld, col_off0, col_off1, width, csr_off, csr_ent,
                       dst, avg);
  else
    hipLaunchKernelGGL(segment_sum_wide_kernel, dim3(O), dim3(threads), 0, s, src, src_ld, col_off0, col_off1, width, csr_off,
                       csr_ent, dst, avg);
  SG_LAUNCH_CHECK("sg_segment_sum");
  return 0;
}

extern "C" int sg_pool_bwd(const float* g_pooled, const float* g_new_p, const int64_t* edges, const int32_t* csr_off,
                           float* g_new_t, int T, int H, int Dout, int avg, sgStream stream) {
  SG_ARG_CHECK(g_pooled && edges && csr_off && g_new_t && T >= 0, "sg_pool_bwd: bad arguments");
  if (T == 0) return 0;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(T), dim3(row_threads(2 * H + Dout)), 0, (hipStream_t)stream, g_pooled, g_new_p,
                     edges, csr_off, g_new_t, H, Dout, avg);
  SG_LAUNCH_CHECK("sg_pool_bwd");
  return 0;
}

extern "C" int sg_embedding_fwd(const float* table, const int64_t* idx, float* out, int n, int dim, sgStream stream) {
  SG_ARG_CHECK(table && idx && out && n >= 0 && dim > 0, "sg_embedding_fwd: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(n), dim3(row_threads(dim)), 0, (hipStream_t)stream, table, idx, out, dim);
  SG_LAUNCH_CHECK("sg_embedding_fwd");
  return 0;
}

extern "C" int sg_embedding_bwd(const float* g, const int64_t* idx, float* g_table, int n, int num_rows, int dim,
                                sgStream stream) {
  SG_ARG_CHECK(g && idx && g_table && num_rows > 0 && dim > 0, "sg_embedding_bwd: bad arguments");
  // column chunks so that ~1024 workgroups are in flight (at least 256 columns each)
  int ysplit = (1024 + num_rows - 1) / num_rows;
  const int maxy = dim / 256 > 0 ? dim / 256 : 1;
  if (ysplit > maxy) ysplit = maxy;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(num_rows, ysplit), dim3(row_threads(dim)), 0, (hipStream_t)stream, g, idx, g_table,
                     n, dim);
  SG_LAUNCH_CHECK("sg_embedding_bwd");
  return 0;
}

extern "C" int sg_copy_cols(const float* src, int src_ld, int src_off, float* dst, int dst_ld, int dst_off, int rows,
                            int width, sgStream stream) {
  SG_ARG_CHECK(src && dst && rows >= 0 && width > 0, "sg_copy_cols: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(sg_cdiv((size_t)rows * width, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_ld, src_off, dst, dst_ld, dst_off, rows, width);
  SG_LAUNCH_CHECK("sg_copy_cols");
  return 0;
}

extern "C" int sg_one_hot(const int64_t* idx, float* out, int n, int classes, int ld, int col_off, sgStream stream) {
  SG_ARG_CHECK(idx && out && n >= 0 && classes > 0, "sg_one_hot: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(one_hot_kernel, dim3(sg_cdiv((size_t)n * classes, 256)), dim3(256), 0, (hipStream_t)stream, idx, out, n,
                     classes, ld, col_off);
  SG_LAUNCH_CHECK("sg_one_hot");
  return 0;
}
