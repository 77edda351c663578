// Implicit-GEMM family on v_mfma_f32_32x32x2_f32 (exact fp32; gfx950 f32 MFMA peak 157.3 TFLOP/s).
//
// One kernel template  C[M,N] = sum_k A[m,k] * B[k,n]  with pluggable operand loaders + epilogues:
//   conv2d fwd      : A = W [Cout][Cin*KS*KS] (k-contiguous), B = im2col gather of x (never materialised)
//   conv2d dgrad    : A = W^T [Cin][Cout*KS*KS],              B = transposed gather of gy
//   convT  fwd/dgrad: the two above with roles swapped
//   conv   wgrad    : A = gy [img][Cout][pix] (k = img*pix),  B = gather of x, N = Cin*KS*KS, split-K slabs
//   linear fwd/bwd  : plain strided operands
// Data flow per workgroup (256 threads = 4 wave64, one per SIMD): global -> registers (next tile, in
// flight during the MFMAs) -> LDS [BK][BM+4] / [BK][BN+4] (k-major, +4 pad: conflict-free b32 fragment
// reads, <=2-way on writes) -> one VGPR per operand per MFMA.  Reflection padding, nearest x2 upsampling
// and the (layout, image) channel concat are folded into the gather index, so none is materialised.
// Replaces the cuDNN/ATen conv + addmm kernels the reference dispatches (see include/sg2im_hip.h).
#include "common.h"
#include <stdlib.h>
#include <array>
#include <map>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifndef SG_NSUB
#define SG_NSUB 1    // sub-tiles per workgroup k-tile (see CfgFor below for why 1)
#endif
constexpr int BK = 16;     // sub-tile depth: the unit one loader call stages (16 k-rows)

// A workgroup k-tile is NSUB sub-tiles deep (BKT = 16*NSUB): all 2*NSUB loader calls of the next tile are issued
// before the MFMAs of the current one, so NSUB*~8 global loads per thread stay in flight across 8*NSUB*TM*TN
// MFMAs (the f32 MFMA is 64 cycles: one sub-tile of work per load round trip left the kernel latency-bound).
#ifndef SG_PIPE_DEFAULT
#define SG_PIPE_DEFAULT 1      // software-pipelined main loop for every instantiation (0: the plain loop; measured on MI355X:
#endif                         // 648 -> 660 images/s, every kernel family +1..6 %)
template <int BM_, int BN_, int WGM_, int NSUB_, int PIPE_ = SG_PIPE_DEFAULT>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = 4 / WGM_, NSUB = NSUB_, BKT = BK * NSUB_, PIPE = PIPE_;
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int LDA = BM + 4, LDB = BN + 4;
};

// LDS tile layout: [x][k] with k contiguous, row pitch LDK = 16 + 4 floats (80 B).  One ds_read_b128 then feeds FOUR
// MFMA k-steps of a 32x32 fragment row and the pitch makes both the b128 fragment reads (16-lane groups hit 16
// distinct 4-bank slots) and the b128/b32 stores conflict-free.  MFMA k-step s pairs tile column s (lanes 0-31) with
// column s+8 (lanes 32-63): any pairing is legal as long as A and B use the same one.
constexpr int LDK = BK + 4;

// ------------------------------------------------------------------------------------------------
// Operand loaders.  Each keeps its per-thread staging registers; load() issues the global reads for
// the k-tile [k0, k0+BK) (zero-filled outside [.., kend) and outside the matrix), store() writes them
// to the LDS tile laid out [BK][BX+4].
// ------------------------------------------------------------------------------------------------

// All loaders are BRANCH-FREE and DEFERRED: load() issues every global read of the k-tile from a clamped (always
// valid) address and only records a validity bit; the zero-select happens in store(), i.e. AFTER the MFMAs of
// the current tile, so the loads stay in flight across the whole compute phase.  (Exec-masked conditional
// loads, or selects placed right behind the loads, made hipcc wait vmcnt(0) before the MFMAs: 4-5x slower.)
// Offsets are 32-bit: the host rejects tensors of >= 2^31 elements.

// SG_BUFLOAD (masked variants): the gathered elements are fetched with raw BUFFER loads whose hardware range check does the
// masking -- invalid taps / k tails / pixel tails carry an offset of 2^29 elements in the LDS tap table, which lands beyond
// num_records (2^31 bytes) and makes the load return 0.  No address select, no validity bits, no zero-select before the LDS
// store: ~20 of the ~60 VALU instructions per k-tile of the 64x64 kernel (the f32 MFMA competes with VALU work for the SIMD).
#ifndef SG_BUFLOAD
#define SG_BUFLOAD 1     // measured on MI355X: 537 -> 554 images/s, conv fwd / dgrad micro-benchmarks +7..20 %
#endif
constexpr int TAP_INVALID = SG_BUFLOAD ? (1 << 29) : -1;
#if SG_BUFLOAD
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ float sg_bufload(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(elem << 2), 0, 0));
}
// (the result is moved with memcpy: naming the builtin's vector type and indexing it made hipcc 7.2 select a ONE-dword load)
__device__ __forceinline__ float4 sg_bufload4(__amdgpu_buffer_rsrc_t r, unsigned elem) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem << 2), 0, 0);
  static_assert(sizeof(v) == 16, "raw_buffer_load_b128 must return 16 bytes");
  float4 f;
  __builtin_memcpy(&f, &v, 16);
  return f;
}
#endif
constexpr unsigned ELEM_INVALID = 1u << 29;      // element offset that the range check of a buffer load rejects
// Largest tensor (in elements) an operand loader may address.  Buffer loads carry a BYTE offset in 32 bits against
// num_records = 2^31 bytes, so element indices must stay below 2^29 (an index in [2^29, 2^31) would be range-rejected and
// silently read as 0); the plain-load build addresses 2^31 elements.  Every entry point checks its operands against this.
constexpr double SG_MAX_ELEMS = SG_BUFLOAD ? 536870912.0 : 2147483647.0;


// rows of length K contiguous in memory: elem(x, k) = base[x*ld + k].  VEC: ld%4==0 and 16-B aligned base.
template <int BX, bool VEC, bool MASK = true>
struct LoadKContig {
  const float* base; int ld; int X;
  static constexpr int LDS_INTS = 0;
  static constexpr int PASSES = BX >= 64 ? BX / 64 : 1;
  struct Stage { float r[PASSES * 4]; unsigned ok; };
  int x0_, xr_, kq_;
  __device__ __forceinline__ void init(int x0, int tid, int*, int, int) { x0_ = x0; xr_ = tid >> 2; kq_ = (tid & 3) * 4; }
  __device__ __forceinline__ void set_batch(int b, int stride, int) { base += (size_t)b * (size_t)stride; }
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    st.ok = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int xl = xr_ + p * 64;
      const int x = x0_ + xl, k = k0 + kq_;
      const bool xok = (BX % 64 == 0 || xl < BX) && x < X;     // (xr_ < 64: whole passes need no row check)
      const unsigned row = (unsigned)(xok ? x : 0) * (unsigned)ld;
      if (VEC && !MASK) {                         // full tiles only (X % BX == 0, K % 16 == 0): no validity at all
        const float4 v = *reinterpret_cast<const float4*>(base + (unsigned)x * (unsigned)ld + k);
        st.r[p * 4 + 0] = v.x; st.r[p * 4 + 1] = v.y; st.r[p * 4 + 2] = v.z; st.r[p * 4 + 3] = v.w;
      } else if (VEC) {                           // kend % 4 == 0 here, so k < kend covers the whole float4
        const bool ok = xok && k < kend;
#if SG_BUFLOAD
        const float4 v = sg_bufload4(sg_rsrc(base), ok ? row + (unsigned)k : ELEM_INVALID);      // rejected => zeros
#else
        const float4 v = *reinterpret_cast<const float4*>(base + row + (ok ? k : 0));
        st.ok |= ok ? (15u << (p * 4)) : 0u;
#endif
        st.r[p * 4 + 0] = v.x; st.r[p * 4 + 1] = v.y; st.r[p * 4 + 2] = v.z; st.r[p * 4 + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = xok && k + i < kend;
#if SG_BUFLOAD
          st.r[p * 4 + i] = sg_bufload(sg_rsrc(base), ok ? row + (unsigned)(k + i) : ELEM_INVALID);
#else
          st.r[p * 4 + i] = base[row + (ok ? k + i : 0)];
          st.ok |= ok ? (1u << (p * 4 + i)) : 0u;
#endif
        }
      }
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int xl = xr_ + p * 64;
      const bool in_tile = BX % 64 == 0 || xl < BX;          // compile-time true for the 64 / 128-row tiles: no exec-mask branch
      if (in_tile && (SG_BUFLOAD || (VEC && !MASK))) {      // (buffer loads already returned zeros for the masked elements)
        *reinterpret_cast<float4*>(T + xl * LDK + kq_) = make_float4(st.r[p * 4], st.r[p * 4 + 1], st.r[p * 4 + 2], st.r[p * 4 + 3]);
      } else if (in_tile) {
        float4 v;
        v.x = ((st.ok >> (p * 4 + 0)) & 1u) ? st.r[p * 4 + 0] : 0.f;
        v.y = ((st.ok >> (p * 4 + 1)) & 1u) ? st.r[p * 4 + 1] : 0.f;
        v.z = ((st.ok >> (p * 4 + 2)) & 1u) ? st.r[p * 4 + 2] : 0.f;
        v.w = ((st.ok >> (p * 4 + 3)) & 1u) ? st.r[p * 4 + 3] : 0.f;
        *reinterpret_cast<float4*>(T + xl * LDK + kq_) = v;
      }
    }
  }
};

// store ROWS consecutive k values of tile row xl starting at column kr (kr % ROWS == 0) as 16/8-byte LDS writes
template <int ROWS, bool MASK = true>
__device__ __forceinline__ void store_krun(float* T, int xl, int kr, const float (&r)[ROWS], unsigned ok) {
  float v[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) v[i] = (!MASK || ((ok >> i) & 1u)) ? r[i] : 0.f;
  float* dst = T + xl * LDK + kr;
  if (ROWS % 4 == 0) {
#pragma unroll
    for (int i = 0; i < ROWS; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else if (ROWS % 2 == 0) {
#pragma unroll
    for (int i = 0; i < ROWS; i += 2) *reinterpret_cast<float2*>(dst + i) = make_float2(v[i], v[i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) dst[i] = v[i];
  }
}

// the M/N index contiguous in memory: elem(x, k) = base[k*ld + x]
template <int BX>
struct LoadXContig {
  const float* base; int ld; int X;
  static constexpr int LDS_INTS = 0;
  static constexpr int ROWS = BX * BK / 256;
  struct Stage { float r[ROWS]; unsigned ok; };
  int x_, xl_, kr_;
  __device__ __forceinline__ void init(int x0, int tid, int*, int, int) { xl_ = tid % BX; x_ = x0 + xl_; kr_ = (tid / BX) * ROWS; }
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    st.ok = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int k = k0 + kr_ + i;
      const bool ok = x_ < X && k < kend;
      st.r[i] = base[ok ? (unsigned)k * (unsigned)ld + (unsigned)x_ : 0u];
      st.ok |= ok ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const { store_krun<ROWS>(T, xl_, kr_, st.r, st.ok); }
};

// geometry of a gathered (im2col-style) operand
struct Gather {
  const float* src1; const float* src2;  // channel-concatenated sources (src2 may be null)
  int C1, C2;                            // channels per source
  int SH, SW;                            // stored spatial size
  int LH, LW;                            // logical size (= SH<<ushift)
  int ushift;                            // 1: nearest x2 upsample folded in
  int PH, PW;                            // pixel grid the OTHER index runs over (output grid)
  int stride, sshift, pad, reflect;
  int bcast2;                            // src2 is [img][C2], broadcast over the spatial grid
  int pstep, ph0, pw0;                   // the pixel grid is the sub-lattice (pstep*i + ph0, pstep*j + pw0) of the full one
};

// offset of tap (kh, kw) inside one stored channel plane for anchor (ah, aw); -1 = contributes zero
template <int MODE>
__device__ __forceinline__ int tap_offset(const Gather& g, int ah, int aw, int kh, int kw) {
  if (MODE == 0) {            // source position = out*stride - pad + tap   (conv-style)
    int ih = ah + kh, iw = aw + kw;
    const bool inside = (unsigned)ih < (unsigned)g.LH && (unsigned)iw < (unsigned)g.LW;
    int rh = ih < 0 ? -ih : ih; rh = rh >= g.LH ? 2 * g.LH - 2 - rh : rh;
    int rw = iw < 0 ? -iw : iw; rw = rw >= g.LW ? 2 * g.LW - 2 - rw : rw;
    ih = (g.reflect ? rh : ih) >> g.ushift;
    iw = (g.reflect ? rw : iw) >> g.ushift;
    return (g.reflect || inside) ? ih * g.SW + iw : -1;
  } else {                    // source position = (out + pad - tap)/stride if divisible (transposed conv)
    int th = ah - kh, tw = aw - kw;
    const int smask = g.stride - 1;
    bool ok = (th | tw) >= 0 && ((th | tw) & smask) == 0;
    th >>= g.sshift; tw >>= g.sshift;
    ok = ok && th < g.SH && tw < g.SW;
    return ok ? th * g.SW + tw : -1;
  }
}

// B operand of conv fwd / dgrad: k = (c, kh, kw), n = (img, ph, pw).  One pixel per thread, ROWS consecutive k.
// Everything that does not depend on the loop is precomputed so that one gathered element costs ~6 instructions:
//   * geometry (stride, zero/reflect padding, x2 upsample, transposed-conv divisibility, pixel tail) is resolved
//     ONCE per workgroup into an LDS table tap[t][pixel] of plane offsets (-1 = contributes zero; row KS2 = all -1)
//   * the k -> (channel offset, tap row, source) split is a device table ktab[k] built once per launch
//     (build_ktab_kernel) and fetched with scalar loads, one k-tile ahead
// (the first version recomputed the split with ~30 dependent SALU/VALU ops per element: the load phase of a
//  64x64 tile took ~1000 cycles per k-tile against 512 cycles of MFMA work).
struct KEntry { unsigned choff; unsigned tapsel; };     // channel-plane offset ; tap row | second-source << 8

__global__ void build_ktab_kernel(KEntry* tab, int K, int Kpad, int KS2, int C1, int C2, unsigned shw, int bcast2,
                                  int tail_valid, unsigned variant_stride) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  KEntry e;
  if (k < K) {
    const int c = k / KS2, t = k - c * KS2;
    const bool second = C2 > 0 && c >= C1;
    const unsigned cc = (unsigned)(second ? c - C1 : c);
    e.choff = ((second && bcast2) ? cc : cc * shw) + (unsigned)t * variant_stride;
    e.tapsel = (unsigned)t | (second ? 256u : 0u);
  } else {
    // k >= K: either the all-invalid tap row, or (mask-free kernels) any valid element: the A operand is zero there
    e.choff = 0u; e.tapsel = tail_valid ? 0u : (unsigned)KS2;
  }
  tab[k] = e;
}

// Shape-only index tables (k-split tables) are built once per shape and kept: the key holds everything the table
// depends on, the table lives in device memory owned by the library (the one exception to "the caller owns every buffer":
// sg_plan_cache_bytes / sg_plan_cache_clear in the header).  A table built on stream A is made visible to a later launch on
// stream B with an event wait.
struct TabEntry { void* dev; size_t bytes; hipEvent_t ready; hipStream_t stream; };
using TabKey = std::array<long long, 12>;
std::mutex g_tab_mu;
std::map<TabKey, TabEntry> g_tabs;
size_t g_tab_bytes = 0;

template <class Build>
const void* cached_table(TabKey key, size_t bytes, hipStream_t s, Build build) {
  int dev = 0;
  hipGetDevice(&dev);
  key[11] = dev;
  std::lock_guard<std::mutex> lk(g_tab_mu);
  auto it = g_tabs.find(key);
  if (it == g_tabs.end()) {
    TabEntry e{nullptr, bytes, nullptr, s};
    if (hipMalloc(&e.dev, bytes) != hipSuccess) return nullptr;
    hipEventCreateWithFlags(&e.ready, hipEventDisableTiming);
    build(e.dev);
    hipEventRecord(e.ready, s);
    g_tab_bytes += bytes;
    it = g_tabs.emplace(key, e).first;
  } else if (it->second.stream != s) {
    // (a capturing stream must not wait on an event recorded outside the capture; a graph is only ever captured after the
    //  eager warm-up iterations that built the table, so the table is long complete)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone) hipStreamWaitEvent(s, it->second.ready, 0);
  }
  return it->second.dev;
}

template <int BN, int KS, int MODE, bool TWO, bool MASK = true>
struct LoadGatherKN {
  Gather g; int Npix; const KEntry* ktab;
  static constexpr int KS2 = KS * KS;
  static constexpr int LDS_INTS = (KS2 + 1) * BN;
  static constexpr int ROWS = BN * BK / 256;
  static constexpr bool BUF = SG_BUFLOAD && MASK;
  struct Stage { float r[ROWS]; unsigned ok; KEntry e[ROWS]; };
  unsigned img1_, img2_, img2b_;
  int nl_, kr_;
  const int* tab_;
  __device__ __forceinline__ void set_batch(int b, int stride, int limit) { ktab += (size_t)b * (size_t)stride; Npix = limit; }
  __device__ __forceinline__ void init(int n0, int tid, int* tab, int, int) {
    nl_ = tid % BN;
    const int grp = tid / BN;
    kr_ = __builtin_amdgcn_readfirstlane(grp * ROWS);          // wave-uniform => ktab entries live in SGPRs
    const int n = n0 + nl_;
    const bool okn = n < Npix;
    const int nn = okn ? n : 0;
    const int phw = g.PH * g.PW;
    const int img = nn / phw;
    const int pix = nn - img * phw;
    const int pi = pix / g.PW;
    const int ph = pi * g.pstep + g.ph0, pw = (pix - pi * g.PW) * g.pstep + g.pw0;
    int ah, aw;
    if (MODE == 0) { ah = ph * g.stride - g.pad; aw = pw * g.stride - g.pad; }
    else { ah = ph + g.pad; aw = pw + g.pad; }
    const unsigned shw = (unsigned)(g.SH * g.SW);
    img1_ = (unsigned)img * (unsigned)g.C1 * shw;
    img2_ = g.bcast2 ? (unsigned)img * (unsigned)g.C2 : (unsigned)img * (unsigned)g.C2 * shw;
    constexpr int G = 256 / BN;
    for (int t = grp; t <= KS2; t += G) {
      const int kh = t / KS, kw = t - kh * KS;
      int v = (okn && t < KS2) ? tap_offset<MODE>(g, ah, aw, kh, kw) : -1;
      if (BUF && v < 0) v = TAP_INVALID;
      tab[t * BN + nl_] = v;
    }
    tab_ = tab + nl_;
  }
  // scalar fetch of the k-split entries of tile k0 (issued one tile ahead of load(), so SMEM latency is hidden)
  __device__ __forceinline__ void prefetch(Stage& st, int k0) const {
    const KEntry* e = ktab + (k0 + kr_);          // uniform address => s_load
#pragma unroll
    for (int i = 0; i < ROWS; ++i) st.e[i] = e[i];
  }
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
#if SG_BUFLOAD
    if (BUF) {
      const __amdgpu_buffer_rsrc_t r1 = sg_rsrc(g.src1);
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        const unsigned choff = st.e[i].choff, ts = st.e[i].tapsel;
        const unsigned tp = (unsigned)tab_[(ts & 255u) * BN];
        if (TWO) {
          const bool second = (ts & 256u) != 0u;                  // scalar
          const unsigned t2 = (second && g.bcast2) ? (tp >= (unsigned)TAP_INVALID ? tp : 0u) : tp;
          st.r[i] = sg_bufload(second ? sg_rsrc(g.src2) : r1, (second ? img2_ : img1_) + choff + t2);
        } else {
          st.r[i] = sg_bufload(r1, img1_ + choff + tp);
        }
      }
      return;
    }
#endif
    st.ok = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const unsigned choff = st.e[i].choff, ts = st.e[i].tapsel;
      const int tp = tab_[(ts & 255u) * BN];
      const bool ok = !MASK || tp >= 0;           // !MASK: reflection padding + full pixel tiles => every tap is valid
      if (TWO) {
        const bool second = (ts & 256u) != 0u;                  // scalar
        const float* base = second ? g.src2 : g.src1;
        const unsigned off = (second ? img2_ : img1_) + choff + ((second && g.bcast2) ? 0u : (unsigned)tp);
        st.r[i] = base[ok ? off : 0u];
      } else {
        st.r[i] = g.src1[ok ? img1_ + choff + (unsigned)tp : 0u];
      }
      st.ok |= ok ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
    if (BUF) store_krun<ROWS, false>(T, nl_, kr_, st.r, 0u);
    else store_krun<ROWS, MASK>(T, nl_, kr_, st.r, st.ok);
  }
};

// A operand of wgrad, vector form (PQ % 4 == 0, 16-byte aligned base): each thread moves one float4 of four
// consecutive pixels of one row: 1 global_load_dwordx4 + 1 ds_write_b128 per 4 elements
template <int BM>
struct LoadPixKVec {
  const float* base; int M, Mtot, PQ; FastDiv dPQ;
  static constexpr int LDS_INTS = 0;
  static constexpr int Q = (BM * 4 + 255) / 256;
  struct Stage { float4 r[Q]; unsigned ok; };
  int m0_, row_, kq_;
  __device__ __forceinline__ void init(int m0, int tid, int*, int, int) { m0_ = m0; row_ = tid >> 2; kq_ = (tid & 3) * 4; }
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    const int k = k0 + kq_;
    const bool kok = k < kend;
    const unsigned kk = kok ? (unsigned)k : 0u;
    const unsigned img = dPQ.div(kk), pix = kk - img * (unsigned)PQ;
    const unsigned p0 = img * (unsigned)Mtot * (unsigned)PQ + pix;
    st.ok = 0;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int m = m0_ + row_ + 64 * i;
      const bool ok = kok && m < M && (row_ + 64 * i < BM);
#if SG_BUFLOAD
      st.r[i] = sg_bufload4(sg_rsrc(base), ok ? p0 + (unsigned)m * (unsigned)PQ : ELEM_INVALID);       // rejected => zeros
#else
      st.r[i] = *reinterpret_cast<const float4*>(base + (ok ? p0 + (unsigned)m * (unsigned)PQ : 0u));
      st.ok |= ok ? (1u << i) : 0u;
#endif
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      if (row_ + 64 * i < BM) {
        float4 v = st.r[i];
#if !SG_BUFLOAD
        const bool ok = (st.ok >> i) & 1u;
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        *reinterpret_cast<float4*>(T + (row_ + 64 * i) * LDK + kq_) = v;
      }
    }
  }
};

// A operand of wgrad: elem(m, k) = base[(img*Mtot + m)*PQ + pix], k = img*PQ + pix.  Lanes run along k.
template <int BM, bool MASK = true>
struct LoadPixK {
  const float* base; int M, Mtot, PQ; FastDiv dPQ;
  static constexpr int LDS_INTS = 0;
  static constexpr int ROWS = BM / 16;
  struct Stage { float r[ROWS]; unsigned ok; };
  int m0_, mr_, kl_;
  __device__ __forceinline__ void init(int m0, int tid, int*, int, int) { m0_ = m0; kl_ = tid & 15; mr_ = tid >> 4; }
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    const int k = k0 + kl_;
    const bool kok = k < kend;
    const unsigned kk = kok ? (unsigned)k : 0u;
    const unsigned img = dPQ.div(kk), pix = kk - img * (unsigned)PQ;
    const unsigned p0 = img * (unsigned)Mtot * (unsigned)PQ + pix;
    st.ok = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int m = m0_ + mr_ + 16 * i;
      const bool ok = !MASK || (kok && m < M);
#if SG_BUFLOAD
      st.r[i] = sg_bufload(sg_rsrc(base), ok ? p0 + (unsigned)m * (unsigned)PQ : ELEM_INVALID);
#else
      st.r[i] = base[ok ? p0 + (unsigned)m * (unsigned)PQ : 0u];
      st.ok |= ok ? (1u << i) : 0u;
#endif
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
      T[(mr_ + 16 * i) * LDK + kl_] = (SG_BUFLOAD || !MASK || ((st.ok >> i) & 1u)) ? st.r[i] : 0.f;
  }
};

// B operand of wgrad: k = (img, ph, pw) over the gy grid, n = (c, kh, kw).  Lanes run along k (16 pixels per tile); the
// COLS columns a thread owns are fixed for the whole k-loop, so their (channel offset, tap row) split is done once.
// The geometry of the 16 pixels of a k-tile (tap offsets incl. padding / reflection / upsampling, image bases) is
// computed cooperatively ONE TILE AHEAD into a double-buffered LDS table (prefetch()), which turns a gathered
// element into: 1 ds_read + 1 add + 1 load (it was a ~30-instruction tap_offset() per element per tile).
// one axis of tap_offset<0>: source coordinate (already >> upsample) or -1
__device__ __forceinline__ int axis_offset(int a, int k, int L, int reflect, int ushift) {
  int i = a + k;
  const bool inside = (unsigned)i < (unsigned)L;
  int r = i < 0 ? -i : i; r = r >= L ? 2 * L - 2 - r : r;
  i = (reflect ? r : i) >> ushift;
  return (reflect || inside) ? i : -1;
}

template <int BN, int KS, bool TWO, bool MASK = true, int NS = SG_NSUB>
struct LoadGatherNK {
  Gather g; int Ncols;
  const int* chan_list; const int* chan_cnt; int L;     // optional per-image active-channel lists
  int zdiv;                                             // image = blockIdx.z / zdiv (k-chunks per image), 0 == 1
  static constexpr int KS2 = KS * KS;
  // per k-tile LDS table (separable): rowoff[KS][16] (= ih*SW or -1), coloff[KS][16] (= iw or -1), one all -1 row,
  // img1[16], img2[16]
  static constexpr int NEG = 2 * KS;
  static constexpr int BUF = (2 * KS + 1) * BK + 2 * BK;
  static constexpr int LDS_INTS = 2 * NS * BUF;
  static constexpr int COLS = BN / 16;
  struct Stage { float r[COLS]; unsigned ok; };
  int kl_, tid_, nr_, kbeg_, kend_;
  unsigned choff_[COLS], secmask_;
  int rrow_[COLS], crow_[COLS];
  int* lds_;
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void init(int n0, int tid, int* lds, int kbeg, int kend) {
    kl_ = tid & 15; nr_ = tid >> 4; tid_ = tid; lds_ = lds; kbeg_ = kbeg; kend_ = kend;
    const unsigned shw = (unsigned)(g.SH * g.SW);
    const int zimg = zdiv > 1 ? blockIdx.z / zdiv : blockIdx.z;
    const int* list = chan_list ? chan_list + (size_t)zimg * L : nullptr;
    const int ncols = chan_list ? chan_cnt[zimg] * KS2 : Ncols;
    secmask_ = 0;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const int n = n0 + nr_ + 16 * j;
      const bool ok = n < ncols;
      const int nn = ok ? n : 0;
      const int cj = nn / KS2;
      const int t = nn - cj * KS2;
      const int c = list ? list[cj] : cj;
      const int kh = t / KS, kw = t - kh * KS;
      const bool second = TWO && c >= g.C1;
      const unsigned cc = (unsigned)(second ? c - g.C1 : c);
      choff_[j] = (second && g.bcast2) ? cc : cc * shw;
      // column tail: the all -1 row, or (mask-free kernels) any valid tap -- the epilogue never stores n >= Ncols
      rrow_[j] = (ok ? kh : (MASK ? NEG : 0)) * BK + kl_;
      crow_[j] = (ok ? KS + kw : (MASK ? NEG : KS)) * BK + kl_;
      secmask_ |= second ? (1u << j) : 0u;
    }
  }
  __device__ __forceinline__ int* buf_of(int k0) const { return lds_ + (((k0 - kbeg_) / BK) % (2 * NS)) * BUF; }
  __device__ __forceinline__ void prefetch(Stage&, int k0) const {
    int* buf = buf_of(k0);
    const int phw = g.PH * g.PW;
    const unsigned shw = (unsigned)(g.SH * g.SW);
    for (int e = tid_; e < BUF; e += 256) {
      const int p = e & (BK - 1), row = e / BK;
      const int k = k0 + p;
      const bool kok = k < kend_;
      const int kk = kok ? k : 0;
      const int img = kk / phw, pix = kk - img * phw;
      const int ph = pix / g.PW, pw = pix - ph * g.PW;
      int val = -1;
      if (row < KS) {
        const int i = axis_offset(ph * g.stride - g.pad, row, g.LH, g.reflect, g.ushift);
        val = (kok && i >= 0) ? i * g.SW : -1;
      } else if (row < 2 * KS) {
        const int i = axis_offset(pw * g.stride - g.pad, row - KS, g.LW, g.reflect, g.ushift);
        val = kok ? i : -1;
      } else if (row == 2 * KS + 1) {
        val = (int)((unsigned)img * (unsigned)g.C1 * shw);
      } else if (row == 2 * KS + 2) {
        val = (int)(g.bcast2 ? (unsigned)img * (unsigned)g.C2 : (unsigned)img * (unsigned)g.C2 * shw);
      }
      buf[e] = val;
    }
  }
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    const int* buf = buf_of(k0);
    const unsigned img1 = (unsigned)buf[(2 * KS + 1) * BK + kl_];
    const unsigned img2 = TWO ? (unsigned)buf[(2 * KS + 2) * BK + kl_] : 0u;
    st.ok = 0;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const int ro = buf[rrow_[j]], co = buf[crow_[j]];
      const bool ok = !MASK || (ro | co) >= 0;
      const unsigned tp = (unsigned)(ro + co);
      if (TWO) {
        const bool second = (secmask_ >> j) & 1u;
        const float* base = second ? g.src2 : g.src1;
        const unsigned off = (second ? img2 : img1) + choff_[j] + ((second && g.bcast2) ? 0u : tp);
        st.r[j] = base[ok ? off : 0u];
      } else {
        st.r[j] = g.src1[ok ? img1 + choff_[j] + tp : 0u];
      }
      st.ok |= ok ? (1u << j) : 0u;
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
#pragma unroll
    for (int j = 0; j < COLS; ++j) T[(nr_ + 16 * j) * LDK + kl_] = (!MASK || ((st.ok >> j) & 1u)) ? st.r[j] : 0.f;
  }
};

// B operand of wgrad, tap-major column order: the N axis is laid out as (tap, channel) with the channel range padded
// to whole tiles, so ONE (kh, kw) is shared by a whole workgroup for its whole k-loop.  The source offset of a pixel
// is then a function of the pixel alone: 16 lanes compute it one k-tile ahead into LDS (prefetch()) and a gathered
// element costs 1 add + 1 load (the general loader above spends 2 LDS reads, 2 adds and a validity test per
// element).  EpWgrad un-permutes the columns on the way out.
template <int BN, bool TWO, bool MASK = true, int NS = SG_NSUB>
struct LoadTapNK {
  Gather g; int KS, Ccols, cpad;
  const int* chan_list; const int* chan_cnt; int L;     // optional per-image active-channel lists (image = blockIdx.z)
  FastDiv dPQ, dPW;
  static constexpr int BUF = 2 * BK;                    // off1[16], off2[16]
  static constexpr int LDS_INTS = 2 * NS * BUF;
  static constexpr int COLS = BN / 16;
  struct Stage { float r[COLS]; unsigned ok; };
  int kl_, tid_, nr_, kbeg_, kend_, kh_, kw_;
  unsigned choff_[COLS], secmask_;
  int* lds_;
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void init(int n0, int tid, int* lds, int kbeg, int kend) {
    kl_ = tid & 15; nr_ = tid >> 4; tid_ = tid; lds_ = lds; kbeg_ = kbeg; kend_ = kend;
    const int t = n0 / cpad, c0 = n0 - t * cpad;
    kh_ = t / KS; kw_ = t - kh_ * KS;
    const unsigned shw = (unsigned)(g.SH * g.SW);
    const int* list = chan_list ? chan_list + (size_t)blockIdx.z * L : nullptr;
    const int ncols = chan_list ? chan_cnt[blockIdx.z] : Ccols;
    secmask_ = 0;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const int cj = c0 + nr_ + 16 * j;
      const int cc = cj < ncols ? cj : 0;             // column tail: any valid channel, the epilogue never stores it
      const int c = list ? list[cc] : cc;
      const bool second = TWO && c >= g.C1;
      const unsigned cs = (unsigned)(second ? c - g.C1 : c);
      choff_[j] = (second && g.bcast2) ? cs : cs * shw;
      secmask_ |= second ? (1u << j) : 0u;
    }
  }
  __device__ __forceinline__ int* buf_of(int k0) const { return lds_ + (((k0 - kbeg_) / BK) % (2 * NS)) * BUF; }
  __device__ __forceinline__ void prefetch(Stage&, int k0) const {
    if (tid_ < BK) {
      int* buf = buf_of(k0);
      const int k = k0 + tid_;
      const bool kok = k < kend_;
      const unsigned kk = kok ? (unsigned)k : 0u;
      const unsigned img = dPQ.div(kk), pix = kk - img * (unsigned)(g.PH * g.PW);
      const unsigned ph = dPW.div(pix), pw = pix - ph * (unsigned)g.PW;
      const int ih = axis_offset((int)ph * g.stride - g.pad, kh_, g.LH, g.reflect, g.ushift);
      const int iw = axis_offset((int)pw * g.stride - g.pad, kw_, g.LW, g.reflect, g.ushift);
      const bool valid = kok && (ih | iw) >= 0;
      const unsigned shw = (unsigned)(g.SH * g.SW);
      const unsigned tp = (unsigned)(ih * g.SW + iw);
      buf[tid_] = valid ? (int)(img * (unsigned)g.C1 * shw + tp) : ((SG_BUFLOAD && MASK) ? (int)ELEM_INVALID : -1);
      if (TWO) buf[BK + tid_] = (SG_BUFLOAD && MASK && !valid) ? (int)ELEM_INVALID
                                    : (int)(g.bcast2 ? img * (unsigned)g.C2 : img * (unsigned)g.C2 * shw + tp);
    }
  }
  __device__ __forceinline__ void load(Stage& st, int k0, int) const {
    const int* buf = buf_of(k0);
    const int o1 = buf[kl_];
#if SG_BUFLOAD
    if (MASK) {        // invalid pixels carry ELEM_INVALID: the buffer load's range check returns zeros, nothing to select
      const unsigned u1 = (unsigned)o1, u2 = TWO ? (unsigned)buf[BK + kl_] : 0u;
      const __amdgpu_buffer_rsrc_t r1 = sg_rsrc(g.src1);
#pragma unroll
      for (int j = 0; j < COLS; ++j) {
        if (TWO) {
          const bool second = (secmask_ >> j) & 1u;
          st.r[j] = second ? sg_bufload(sg_rsrc(g.src2), u2 + choff_[j]) : sg_bufload(r1, u1 + choff_[j]);
        } else {
          st.r[j] = sg_bufload(r1, u1 + choff_[j]);
        }
      }
      st.ok = 1u;
      return;
    }
#endif
    const bool ok = !MASK || o1 >= 0;
    const unsigned b1 = ok ? (unsigned)o1 : 0u;         // invalid pixel: element 0 of the channel plane, zeroed in store()
    const unsigned b2 = (TWO && ok) ? (unsigned)buf[BK + kl_] : 0u;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      if (TWO) {
        const bool second = (secmask_ >> j) & 1u;
        const float* base = second ? g.src2 : g.src1;
        st.r[j] = base[(second ? b2 : b1) + choff_[j]];
      } else {
        st.r[j] = g.src1[b1 + choff_[j]];
      }
    }
    st.ok = ok ? 1u : 0u;
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
    const bool ok = !MASK || st.ok != 0u;
#pragma unroll
    for (int j = 0; j < COLS; ++j) T[(nr_ + 16 * j) * LDK + kl_] = ok ? st.r[j] : 0.f;
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  Accumulator register r of a 32x32 tile <-> row (r&3)+8*(r>>2)+4*(lane>>5), col lane&31.
// ------------------------------------------------------------------------------------------------
// Epilogues.  The activation switch and the bounds checks are hoisted OUT of the 16*TM*TN-element store loops: a wave whose
// 32TM x 32TN block lies inside the matrix (the common case) runs a loop specialised for its activation with unconditional
// stores; edge blocks and the rare tanh / sigmoid take the general loop.  (With the switch -- tanhf / expf inlined -- and an
// m < M test per element the epilogue was 40 KB of code per kernel and cost the dense Winograd GEMM 9 % of its run time.)
template <int ACT> __device__ __forceinline__ float act_fixed(float v, int act, float slope) {
  if constexpr (ACT == SG_ACT_NONE) return v;
  else if constexpr (ACT == SG_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == SG_ACT_LEAKY) return v > 0.f ? v : v * slope;
  else return sg_apply_act(v, act, slope);
}
template <int A> struct ActTag { static constexpr int value = A; };
// f(ActTag<A>) with A = the activation when it is one of the cheap ones and the block is full, else -1 (general loop)
template <class F> __device__ __forceinline__ void ep_dispatch(bool full, int act, F&& f) {
  if (full && act == SG_ACT_NONE) f(ActTag<SG_ACT_NONE>{});
  else if (full && act == SG_ACT_LEAKY) f(ActTag<SG_ACT_LEAKY>{});
  else if (full && act == SG_ACT_RELU) f(ActTag<SG_ACT_RELU>{});
  else f(ActTag<-1>{});
}

struct EpNCHW {     // out[z][img][m][pix], n = img*PHW + pix ; bias per row m (z = split-K slab, raw partials)
  float* out; const float* bias; int PHW, Mtot, M, Npix, act; float slope; size_t zstride;
  // optional scatter of a pixel sub-lattice into the full grid (parity-decomposed strided transposed gathers)
  int PWs, step, h0, w0, PWf, PHWf;
  __device__ __forceinline__ void set_limit(int n) { Npix = n; }
  template <int TM, int TN>
  __device__ __forceinline__ void store(f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int z) const {
    const bool full = mbase + 32 * TM <= M && nbase + 32 * TN <= Npix;
    const int mrow = mbase + 4 * (lane >> 5);
    ep_dispatch(full, act, [&](auto tag) {
      constexpr int A = decltype(tag)::value;
      constexpr bool FULL = A >= 0;
      float bv[TM][16];
      if (bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow + i * 32 + (r & 3) + 8 * (r >> 2);
            bv[i][r] = (FULL || m < M) ? bias[m] : 0.f;
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) bv[i][r] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nbase + j * 32 + (lane & 31);
        if (!FULL && n >= Npix) continue;
        const int img = n / PHW;
        int pix = n - img * PHW, plane = PHW;
        if (step > 1) {
          const int i = pix / PWs;
          pix = (i * step + h0) * PWf + (pix - i * PWs) * step + w0;
          plane = PHWf;
        }
        float* o = out + (size_t)z * zstride + ((size_t)img * Mtot + mrow) * plane + pix;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
            if (FULL || mrow + dm < M) o[(size_t)dm * plane] = act_fixed<A>(acc[i][j][r] + bv[i][r], act, slope);
          }
        }
      }
    });
  }
};

struct EpRowMajor {  // out[z][m*ldc + n] ; bias per column n
  float* out; const float* bias; int M, N, ldc, act; float slope; size_t zstride;
  __device__ __forceinline__ void set_limit(int) {}
  template <int TM, int TN>
  __device__ __forceinline__ void store(f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int z) const {
    const bool full = mbase + 32 * TM <= M && nbase + 32 * TN <= N;
    const int mrow = mbase + 4 * (lane >> 5), n0 = nbase + (lane & 31);
    float* o = out + (size_t)z * zstride + (size_t)mrow * ldc + n0;
    ep_dispatch(full, act, [&](auto tag) {
      constexpr int A = decltype(tag)::value;
      constexpr bool FULL = A >= 0;
      float b[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = (bias && (FULL || n0 + j * 32 < N)) ? bias[n0 + j * 32] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
          if (!FULL && mrow + dm >= M) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (FULL || n0 + j * 32 < N) o[(size_t)dm * ldc + j * 32] = act_fixed<A>(acc[i][j][r] + b[j], act, slope);
        }
    });
  }
};

struct EpRowMajorPlain {  // out[m*ldc + n], full tiles, no bias / activation: 64 unconditional coalesced stores per lane
  float* out; int ldc;
  __device__ __forceinline__ void set_limit(int) {}
  template <int TM, int TN>
  __device__ __forceinline__ void store(f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) const {
    float* o = out + (size_t)(mbase + 4 * (lane >> 5)) * ldc + nbase + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < TN; ++j) o[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32] = acc[i][j][r];
  }
};

struct EpWgrad {     // tap-major virtual column n = t*cpad + cj  ->  slab[z][m][t][cj] (lanes run along cj: coalesced)
  float* out; int M, cpad, KS2; size_t zstride;
  __device__ __forceinline__ void set_limit(int) {}
  template <int TM, int TN>
  __device__ __forceinline__ void store(f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int z) const {
    float* o0 = out + (size_t)z * zstride;
    const size_t ldm = (size_t)KS2 * cpad;
    const int mrow = mbase + 4 * (lane >> 5), n0 = nbase + (lane & 31), Ncols = KS2 * cpad;
    float* o = o0 + (size_t)mrow * ldm + n0;
    if (mbase + 32 * TM <= M && nbase + 32 * TN <= Ncols) {          // whole block inside: unconditional stores
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < TN; ++j) o[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldm + j * 32] = acc[i][j][r];
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (n0 + j * 32 >= Ncols) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
          if (mrow + dm < M) o[(size_t)dm * ldm + j * 32] = acc[i][j][r];
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The kernel
// ------------------------------------------------------------------------------------------------
// Batched mode (channel-sparse first layers): the N axis is split into `nbatch` images of `cols_per_batch` columns,
// tiles never straddle images, and each image has its own compact A operand, k-table and K extent.
// Parity classes of a stride-2 transposed gather run as ONE launch: class c owns the n-tiles [tile0[c], tile0[c+1]), has its
// own compact A operand (offset aoff, row length K), k-table, K extent and pixel sub-lattice.
struct ParityClasses {
  int ncls; int tile0[5]; int Npix[4]; int K[4]; int PH[4], PW[4], ph0[4], pw0[4]; unsigned aoff[4];
  const void* ktab[4];
};
struct BatchInfo {
  int cols_per_batch; int nbatch; const int* kcnt; int a_stride; int b_stride;
  // K = (image, pixel) GEMMs split so that no k-chunk straddles an image: grid.z = image * ksplit + q, chunk q of
  // image i covers pixels [i*kimg + q*kcs, min((i+1)*kimg, ... + kcs))  (ksplit == 0: plain blockIdx.z * kchunk)
  int kimg, ksplit, kcs;
  // batch_major: tiles are numbered batch-major (all tiles of batch 0, then batch 1, ...), so that with the XCD remap
  // below every XCD works on whole batches and their operands stay in ITS L2 (batched Winograd GEMMs)
  int batch_major;
  // xcd_splitk: plain split-K launch (grid.z = splits, a multiple of 8, (tiles * splits) % 8 == 0): see the kernel
  int xcd_splitk;
  ParityClasses par;
};
// per-class hooks: loaders / epilogues that can run a parity class overload these; everything else ignores the call
template <class L> __device__ __forceinline__ void set_class_a(L&, unsigned, int) {}
template <class L> __device__ __forceinline__ void set_class_b(L&, const ParityClasses&, int) {}
template <class E> __device__ __forceinline__ void set_class_ep(E&, const ParityClasses&, int) {}
template <int BX, bool VEC, bool MASK>
__device__ __forceinline__ void set_class_a(LoadKContig<BX, VEC, MASK>& l, unsigned off, int K) { l.base += off; l.ld = K; }
template <int BN, int KS, int MODE, bool TWO, bool MASK>
__device__ __forceinline__ void set_class_b(LoadGatherKN<BN, KS, MODE, TWO, MASK>& l, const ParityClasses& p, int c) {
  l.ktab = reinterpret_cast<const KEntry*>(p.ktab[c]);
  l.g.PH = p.PH[c]; l.g.PW = p.PW[c]; l.g.ph0 = p.ph0[c]; l.g.pw0 = p.pw0[c]; l.Npix = p.Npix[c];
}
__device__ __forceinline__ void set_class_ep(EpNCHW& e, const ParityClasses& p, int c) {
  e.PHW = p.PH[c] * p.PW[c]; e.Npix = p.Npix[c]; e.PWs = p.PW[c]; e.h0 = p.ph0[c]; e.w0 = p.pw0[c];
}
template <class CFG, class AL, class BL, class EP>
__global__ void __launch_bounds__(256) igemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int kchunk, BatchInfo bi) {
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN, LDA = CFG::LDA, LDB = CFG::LDB;
  constexpr int NSUB = CFG::NSUB, BKT = CFG::BKT;
  __shared__ __attribute__((aligned(16))) float As[2][NSUB * BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][NSUB * BN * LDK];
  __shared__ int tapA[AL::LDS_INTS > 0 ? AL::LDS_INTS : 1];
  __shared__ int tapB[BL::LDS_INTS > 0 ? BL::LDS_INTS : 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / CFG::WGN) * CFG::WM, wn0 = (wid % CFG::WGN) * CFG::WN;

  // XCD-aware, bijective tile remap: consecutive tiles (same weight rows, overlapping gathers) share an L2
  const int tiles_pb = bi.cols_per_batch > 0 ? (bi.cols_per_batch + BN - 1) / BN : 0;
  const int tiles_n = bi.cols_per_batch > 0 ? bi.nbatch * tiles_pb : (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  int m0 = (bid / tiles_n) * BM;
  int n0 = (bid % tiles_n) * BN;
  if (bi.par.ncls > 0) {                      // one launch for the parity classes: locate this workgroup's class
    const int tn_all = bi.par.tile0[bi.par.ncls];
    const int tn = bid % tn_all;
    m0 = (bid / tn_all) * BM;
    int c = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) c += (q < bi.par.ncls && tn >= bi.par.tile0[q]) ? 1 : 0;
    n0 = (tn - bi.par.tile0[c]) * BN;
    K = bi.par.K[c];
    set_class_a(al, bi.par.aoff[c], K);
    set_class_b(bl, bi.par, c);
    set_class_ep(ep, bi.par, c);
  }
  int zblk = blockIdx.z;
  if (bi.xcd_splitk) {
    // experiment (SG_XCD_SPLITK=1, grid.z a multiple of 8): split-K slabs pinned to XCDs -- workgroups are handed to the 8
    // XCDs round-robin in linear (z, x) order; re-numbered so that XCD i runs k-chunks i, i+8, ... of every tile.  Measured
    // neutral (532.9 vs 531.6 images/s, identical conv micro-benchmarks), so it is off by default.
    const unsigned lin = blockIdx.z * gridDim.x + blockIdx.x, xcd = lin & 7u, idx = lin >> 3;
    zblk = (int)(xcd + 8u * (idx / gridDim.x));
    const int t = (int)(idx % gridDim.x);
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
  }
  int kbeg = zblk * kchunk;
  int kend = min(K, kbeg + kchunk);
  if (bi.ksplit > 0) {
    const int img = blockIdx.z / bi.ksplit, q = blockIdx.z - img * bi.ksplit;
    kbeg = img * bi.kimg + q * bi.kcs;
    kend = min((img + 1) * bi.kimg, kbeg + bi.kcs);
  }
  if (bi.cols_per_batch > 0) {
    int tn = bid % tiles_n, batch = tn / tiles_pb;
    if (bi.batch_major) {
      const int per_batch = (int)(gridDim.x / (unsigned)bi.nbatch);       // = tiles_m * tiles_pb
      batch = bid / per_batch;
      const int r = bid - batch * per_batch;
      m0 = (r / tiles_pb) * BM;
      tn = batch * tiles_pb + (r % tiles_pb);
    }
    n0 = batch * bi.cols_per_batch + (tn - batch * tiles_pb) * BN;
    if (bi.kcnt) kend = min(kend, bi.kcnt[batch]);
    al.set_batch(batch, bi.a_stride, 0);
    bl.set_batch(batch, bi.b_stride, (batch + 1) * bi.cols_per_batch);
    ep.set_limit((batch + 1) * bi.cols_per_batch);
  }

  al.init(m0, tid, tapA, kbeg, kend);
  bl.init(n0, tid, tapB, kbeg, kend);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typename AL::Stage sa[NSUB];
  typename BL::Stage sb[NSUB];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], kbeg + u * BK); bl.prefetch(sb[u], kbeg + u * BK); }
  if (AL::LDS_INTS > 0 || BL::LDS_INTS > 0) __syncthreads();      // tap tables written by init()/prefetch()
#pragma unroll
  for (int u = 0; u < NSUB; ++u) { al.load(sa[u], kbeg + u * BK, kend); bl.load(sb[u], kbeg + u * BK, kend); }
#pragma unroll
  for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], kbeg + BKT + u * BK); bl.prefetch(sb[u], kbeg + BKT + u * BK); }
#pragma unroll
  for (int u = 0; u < NSUB; ++u) { al.store(sa[u], As[0] + u * BM * LDK); bl.store(sb[u], Bs[0] + u * BN * LDK); }
  __syncthreads();

  const int lr = lane & 31, lk = lane >> 5;
  int buf = 0;
  if constexpr (CFG::PIPE != 0) {
    // Software-pipelined form (dense Winograd GEMMs).  A k-tile is 2*NSUB PHASES of 4 MFMA k-steps (one ds_read_b128 per
    // fragment row, 4*TM*TN MFMAs); the fragments of phase p+1 are read while the MFMAs of phase p issue, the next tile
    // goes to LDS at the TOP of the iteration (its global loads were issued one iteration earlier) and the single barrier
    // sits before the LAST phase, whose MFMAs cover the first fragment reads of the next tile.  No LDS read is waited for
    // right behind a barrier (the plain loop exposes that latency once per sub-tile), the summation order is unchanged.
    constexpr int P = 2 * NSUB;
    float4 fa[2][TM], fb[2][TN];
    auto read_frag = [&](int bsel, int p, float4 (&a)[TM], float4 (&b)[TN]) {
      const int u = p >> 1, h = p & 1;
      const float* A_ = As[bsel] + u * BM * LDK + (wm0 + lr) * LDK + lk * 8 + h * 4;
      const float* B_ = Bs[bsel] + u * BN * LDK + (wn0 + lr) * LDK + lk * 8 + h * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(A_ + i * 32 * LDK);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(B_ + j * 32 * LDK);
    };
    auto mma = [&](const float4 (&a)[TM], const float4 (&b)[TN]) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? a[i].x : (e == 1 ? a[i].y : (e == 2 ? a[i].z : a[i].w));
            const float bv = e == 0 ? b[j].x : (e == 1 ? b[j].y : (e == 2 ? b[j].z : b[j].w));
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    };
    // the prologue above stored tile 0, fetched the k-split entries of tile 1 (prefetch) and passed a barrier; stage tile 1
    // in registers, fetch the entries of tile 2 and the first fragments
    if (kbeg + BKT < kend) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) { al.load(sa[u], kbeg + BKT + u * BK, kend); bl.load(sb[u], kbeg + BKT + u * BK, kend); }
#pragma unroll
      for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], kbeg + 2 * BKT + u * BK); bl.prefetch(sb[u], kbeg + 2 * BKT + u * BK); }
    }
    read_frag(0, 0, fa[0], fb[0]);
    // loaders that keep per-tile tap tables in LDS: the entries prefetch() just wrote are read by load() at the top of the
    // first iteration (later iterations have the mid-iteration barrier in between)
    if (AL::LDS_INTS > 0 || BL::LDS_INTS > 0) __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BKT) {
      const bool more1 = k0 + BKT < kend, more2 = k0 + 2 * BKT < kend;
      if (more1) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) { al.store(sa[u], As[buf ^ 1] + u * BM * LDK); bl.store(sb[u], Bs[buf ^ 1] + u * BN * LDK); }
      }
      if (more2) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) { al.load(sa[u], k0 + 2 * BKT + u * BK, kend); bl.load(sb[u], k0 + 2 * BKT + u * BK, kend); }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], k0 + 3 * BKT + u * BK); bl.prefetch(sb[u], k0 + 3 * BKT + u * BK); }
      }
#pragma unroll
      for (int p = 0; p < P - 1; ++p) {
        read_frag(buf, p + 1, fa[(p + 1) & 1], fb[(p + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);          // keep the reads IN FRONT of the MFMAs that cover their latency
        mma(fa[p & 1], fb[p & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (more1) read_frag(buf ^ 1, 0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      mma(fa[(P - 1) & 1], fb[(P - 1) & 1]);
      buf ^= 1;
    }
    ep.store(acc, m0 + wm0, n0 + wn0, lane, bi.xcd_splitk ? zblk : (int)blockIdx.z);
    return;
  }
  for (int k0 = kbeg; k0 < kend; k0 += BKT) {
    const bool more = k0 + BKT < kend;
    if (more) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) { al.load(sa[u], k0 + BKT + u * BK, kend); bl.load(sb[u], k0 + BKT + u * BK, kend); }
#pragma unroll
      for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], k0 + 2 * BKT + u * BK); bl.prefetch(sb[u], k0 + 2 * BKT + u * BK); }
    }
    // fragment reads: one ds_read_b128 = four k-steps of a 32-row fragment; all reads of a sub-tile are issued up
    // front, the second half (k-steps 4-7) lands while the first 4*TM*TN MFMAs issue
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      const float* A_ = As[buf] + u * BM * LDK + (wm0 + lr) * LDK + lk * 8;
      const float* B_ = Bs[buf] + u * BN * LDK + (wn0 + lr) * LDK + lk * 8;
      float4 a[TM][2], b[TN][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][h] = *reinterpret_cast<const float4*>(A_ + i * 32 * LDK + h * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][h] = *reinterpret_cast<const float4*>(B_ + j * 32 * LDK + h * 4);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float av = e == 0 ? a[i][h].x : (e == 1 ? a[i][h].y : (e == 2 ? a[i][h].z : a[i][h].w));
              const float bv = e == 0 ? b[j][h].x : (e == 1 ? b[j][h].y : (e == 2 ? b[j][h].z : b[j][h].w));
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) {
        al.store(sa[u], As[buf ^ 1] + u * BM * LDK);
        bl.store(sb[u], Bs[buf ^ 1] + u * BN * LDK);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  ep.store(acc, m0 + wm0, n0 + wn0, lane, bi.xcd_splitk ? zblk : (int)blockIdx.z);
}

// tile configurations.  Measured on MI355X (tools/bench_conv.py): these kernels are limited by the vector-memory
// instruction rate of the gather (one 4-byte load per lane per im2col element), not by load latency, so deeper
// k-tiles (NSUB 2/4 => 70 KB LDS => 2 workgroups/CU) LOSE 5-15 % against NSUB=1 with 6-8 resident workgroups.
template <int KS> struct CfgFor {
  using C128 = TileCfg<128, 128, 2, SG_NSUB>;
  using C64 = TileCfg<64, 64, 2, SG_NSUB>;
  using C32 = TileCfg<32, 128, 1, SG_NSUB>;
  using C64W = TileCfg<64, 128, 2, SG_NSUB>;     // 64 rows x 128 pixels: twice the MFMAs per gathered element of 64x64
};
using Cfg128 = CfgFor<3>::C128;      // dense layers use the KS-independent depths
using Cfg64 = CfgFor<3>::C64;
using Cfg32 = CfgFor<3>::C32;
// Two sub-tiles per k-tile (32 deep) for the dense Winograd GEMMs: the loads of a tile are issued a whole 32-deep MFMA block
// ahead, which hides the global-load latency two waves per SIMD cannot.  Measured on MI355X: 92.5 -> 101.5 TFLOP/s over the
// 54 launches of a step (wgrad +17 %, dgrad +7 %, fwd +4 %).  The weight-gradient GEMMs gain on some shapes (mask_net +31 %)
// and lose on others (-3..-10 % on the 128-tile ones); over the step it is a wash, so they stay at depth NSW = 1 like the
// im2col gathers (which LOSE 5-15 % at depth 2: one dword per lane per element, 80 KB of LDS = 2 workgroups per CU).
constexpr int NSW = 1;
using CfgW128 = TileCfg<128, 128, 2, NSW>;
using CfgW64 = TileCfg<64, 64, 2, NSW>;
using CfgW32 = TileCfg<32, 128, 1, NSW>;
using CfgW64W = TileCfg<64, 128, 2, NSW>;
using CfgD128 = TileCfg<128, 128, 2, 2, 0>;   // dense Winograd GEMMs (plain loop: SG_WINO_TILE=3)
using CfgDP128 = TileCfg<128, 128, 2, 2, 1>; // ... software-pipelined fragment reads, barrier before the last phase
using CfgD128x64 = TileCfg<128, 64, 2, 1>;   // experiment (SG_WINO_TILE=1): half-width tiles, 4-5 workgroups per CU

inline int pick_tile(int M, int N) {
  static int force = -2;
  if (force == -2) { const char* e = getenv("SG_TILE"); force = e ? atoi(e) : -1; }
  if (force >= 0) return force;
  if (M <= 32) return 2;
  // measured (tools/bench_conv.py): 128x128 tiles win when M is large (>= 512 rows, split-K fills the chip) or when
  // there are enough of them anyway; 64-row tiles otherwise -- 128 pixels wide (tile 3) while that still leaves >= 3
  // workgroups per CU, else 64x64
  const long t128 = (long)sg_cdiv(M, 128) * sg_cdiv(N, 128);
  const bool low_waste = sg_cdiv(M, 128) * 128 * 20 <= M * 23 && N >= 512;
  static int t128_min = -1, wide_min = -1;           // thresholds (tuning aids: SG_T128_MIN, SG_TILE3_MIN; SG_TILE3=0 disables 64x128)
  if (t128_min < 0) { const char* e = getenv("SG_T128_MIN"); t128_min = e ? atoi(e) : 384; }
  if (M >= 96 && low_waste && (M >= 512 || t128 >= t128_min)) return 0;
  static int wide = -1;
  if (wide < 0) { const char* e = getenv("SG_TILE3"); wide = e ? atoi(e) : 1; }
  if (wide_min < 0) { const char* e = getenv("SG_TILE3_MIN"); wide_min = e ? atoi(e) : 768; }
  if (wide && (long)sg_cdiv(M, 64) * sg_cdiv(N, 128) >= wide_min) return 3;
  return 1;
}

// thread-local launch modifiers (set by the sparse entry points around a regular dispatch)
thread_local BatchInfo t_batch = {};
// per-image ascending active-channel lists; wimg / gwimg (optional): per-image weights [N][M][L][KS2] in list order instead
// of one shared weight tensor (factored layout convs: the channels are the objects of the image)
struct Sparse { const int* list; const int* cnt; int L; const float* wimg; float* gwimg; };
thread_local unsigned t_variant_stride = 0;   // >0: tap t of the k-table reads from source copy t (see reflect_variants_kernel)
thread_local int t_grid_z = 0;                // >0: explicit grid.z (per-image k-chunks, see BatchInfo::ksplit)
thread_local int t_fixed_kchunk = 0;        // >0: grid.z = ceil(K / chunk) with exactly this chunk (one image per z)

template <class CFG, class AL, class BL, class EP>
int launch_cfg(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int splits, hipStream_t s) {
  int tiles = sg_cdiv(M, CFG::BM) * sg_cdiv(N, CFG::BN);
  if (t_batch.cols_per_batch > 0) tiles = sg_cdiv(M, CFG::BM) * t_batch.nbatch * sg_cdiv(t_batch.cols_per_batch, CFG::BN);
  if (t_batch.par.ncls > 0) tiles = sg_cdiv(M, CFG::BM) * t_batch.par.tile0[t_batch.par.ncls];
  int kchunk = K;
  if (splits > 1) kchunk = sg_cdiv(sg_cdiv(K, splits), CFG::BKT) * CFG::BKT;
  if (t_fixed_kchunk > 0) kchunk = t_fixed_kchunk;
  dim3 grid(tiles, 1, t_grid_z > 0 ? t_grid_z : ((splits > 1 || t_fixed_kchunk > 0) ? sg_cdiv(K, kchunk) : 1));
  BatchInfo bi = t_batch;
  static int xs = -1;
  if (xs < 0) { const char* e = getenv("SG_XCD_SPLITK"); xs = e ? atoi(e) : 0; }     // measured neutral on MI355X: off
  bi.xcd_splitk = (xs && t_grid_z == 0 && t_fixed_kchunk == 0 && bi.cols_per_batch == 0 && bi.par.ncls == 0 && bi.ksplit == 0 &&
                   grid.z >= 8 && grid.z % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((igemm_kernel<CFG, AL, BL, EP>), grid, dim3(256), 0, s, al, bl, ep, M, N, K, kchunk, bi);
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// sum split-K slabs: out[i] = sum_z ws[z*n + i]   (fixed order => deterministic)
__global__ void slab_reduce_kernel(const float* ws, float* out, size_t n, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += ws[(size_t)z * n + i];
  out[i] = v;
}

// split-K epilogue of the conv-shaped GEMMs: out[i] = act(sum_z ws[z][i] + bias[channel(i)])
__global__ void slab_reduce_nchw_kernel(const float* ws, float* out, size_t n, int S, const float* bias, int PHW, int Mtot,
                                        int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += ws[(size_t)z * n + i];
  if (bias) v += bias[((unsigned)i / (unsigned)PHW) % (unsigned)Mtot];        // (n < 2^31: 32-bit divisions)
  out[i] = sg_apply_act(v, act, slope);
}
// float4 form (n % 4 == 0, PHW % 4 == 0: the four lanes of a vector share their channel)
__global__ void slab_reduce_nchw_vec_kernel(const float4* ws, float4* out, size_t n4, int S, const float* bias, int PHW4,
                                            int Mtot, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = ws[i];
  for (int z = 1; z < S; ++z) {
    const float4 t = ws[(size_t)z * n4 + i];
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const float b = bias ? bias[((unsigned)i / (unsigned)PHW4) % (unsigned)Mtot] : 0.f;
  v.x = sg_apply_act(v.x + b, act, slope); v.y = sg_apply_act(v.y + b, act, slope);
  v.z = sg_apply_act(v.z + b, act, slope); v.w = sg_apply_act(v.w + b, act, slope);
  out[i] = v;
}

// ReflectionPad2d(1) + 3x3 conv, data gradient without the padded grid.  The gradient of the padded input folds back as
//   gx[i] = sum_k w[k] gy[i+1-k]  +  [i==1] w[0] gy[0]  +  [i==H-2] w[2] gy[H-1]        (per axis)
// i.e. tap 0 at pixel 1 sees gy[2]+gy[0], tap 2 at pixel H-2 sees gy[H-3]+gy[H-1].  Materialising one pre-folded copy
// of gy per tap turns the whole thing into a plain zero-padded transposed gather over the H x W grid (the padded
// formulation computes (H+2)(W+2)/(HW) = 56 % more pixels at 8x8).
__global__ void reflect_variants_kernel(const float* __restrict__ gy, float* __restrict__ V, size_t planes, int H, int W,
                                        size_t VS) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (idx >= planes * HW) return;
  const size_t plane = idx / HW;
  const int p = (int)(idx - plane * HW);
  const int a = p / W, b = p - a * W;
  const float* g = gy + plane * HW;
  const float v00 = g[p];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ra = (kh == 0 && a == 2) ? 0 : ((kh == 2 && a == H - 3) ? H - 1 : -1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int cb = (kw == 0 && b == 2) ? 0 : ((kw == 2 && b == W - 3) ? W - 1 : -1);
      float v = v00;
      if (ra >= 0) v += g[ra * W + b];
      if (cb >= 0) v += g[a * W + cb];
      if (ra >= 0 && cb >= 0) v += g[ra * W + cb];
      V[(size_t)(kh * 3 + kw) * VS + idx] = v;
    }
  }
}

// Wt[b][a][r] = W[a][b][r]
__global__ void permute_w_kernel(const float* W, float* Wt, int A, int B, int R) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)A * B * R;
  if (i >= n) return;
  const int r = i % R;
  const size_t ab = i / R;
  const int a = ab % A;
  const int b = ab / A;
  Wt[i] = W[((size_t)a * B + b) * R + r];
}

Gather make_gather(const float* s1, const float* s2, int C1, int C2, int SH, int SW, int ups, int PH, int PW,
                   int stride, int pad, int reflect) {
  Gather g;
  g.src1 = s1; g.src2 = s2; g.C1 = C1; g.C2 = C2; g.SH = SH; g.SW = SW;
  g.ushift = ups == 2 ? 1 : 0; g.LH = SH << g.ushift; g.LW = SW << g.ushift;
  g.PH = PH; g.PW = PW; g.stride = stride; g.sshift = stride == 2 ? 1 : 0; g.pad = pad; g.reflect = reflect;
  g.bcast2 = 0; g.pstep = 1; g.ph0 = 0; g.pw0 = 0;
  return g;
}

// ---- conv-shaped GEMM: K = (c, taps), N = pixels -------------------------------------------------
template <class CFG, int BM, int BN, int KS, int MODE>
int launch_ab(const float* A, int K, int M, bool vec, const Gather& g, int Npix, const KEntry* ktab, const EpNCHW& ep,
              int splits, bool nomask, hipStream_t s) {
  const bool two = g.C2 > 0;
  if (MODE == 0 && two) {                                 // channel-concatenated sources only exist on the forward gather
    if (vec) {
      if (nomask) return launch_cfg<CFG>(LoadKContig<BM, true, false>{A, K, M}, LoadGatherKN<BN, KS, MODE, true, false>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
      return launch_cfg<CFG>(LoadKContig<BM, true>{A, K, M}, LoadGatherKN<BN, KS, MODE, true>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
    }
    return launch_cfg<CFG>(LoadKContig<BM, false>{A, K, M}, LoadGatherKN<BN, KS, MODE, true>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
  }
  if (vec) {
    if (nomask) return launch_cfg<CFG>(LoadKContig<BM, true, false>{A, K, M}, LoadGatherKN<BN, KS, MODE, false, false>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
    return launch_cfg<CFG>(LoadKContig<BM, true>{A, K, M}, LoadGatherKN<BN, KS, MODE, false>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
  }
  return launch_cfg<CFG>(LoadKContig<BM, false>{A, K, M}, LoadGatherKN<BN, KS, MODE, false>{g, Npix, ktab}, ep, M, Npix, K, splits, s);
}

// split-K for conv-shaped GEMMs that would otherwise leave most CUs idle (few output tiles, long K): e.g. the
// Cout=1 heads of the PatchGANs (91 tiles of 32x128, K=8192)
inline int kn_tiles(int M, int Npix) {
  const int t = pick_tile(M, Npix);
  return t == 0 ? sg_cdiv(M, 128) * sg_cdiv(Npix, 128)
                : (t == 1 ? sg_cdiv(M, 64) * sg_cdiv(Npix, 64) : (t == 3 ? sg_cdiv(M, 64) * sg_cdiv(Npix, 128) : sg_cdiv(Npix, 128)));
}
inline int kn_splits(int M, int Npix, int K) {
  static int force = -2;
  if (force == -2) { const char* e = getenv("SG_SPLITS"); force = e ? atoi(e) : -1; }
  if (force > 0) return force;
  const int tiles = kn_tiles(M, Npix);
  // workgroups wanted in flight: ~6 per CU of the 64x64 / 32x128 kernels, 3 per CU (the register limit) of 128x128
  const int target = pick_tile(M, Npix) == 0 ? 1024 : 1536;
  if (tiles * 4 >= target * 3 || K < 2048) return 1;
  int sp = (target + tiles / 2) / tiles;
  if (sp > 2 && (sp & 1)) ++sp;                 // odd split counts measured 10 % slower than their even neighbours
  if (sp > K / 1024) sp = K / 1024;
  if (sp > 8) sp = 8;
  return sp < 2 ? 1 : sp;
}
inline size_t kn_slab_bytes(int M, int Npix, int K) {
  const int sp = kn_splits(M, Npix, K);
  return sp > 1 ? (size_t)sp * M * Npix * sizeof(float) : 0;
}

inline size_t ktab_bytes(int K) { return (size_t)(sg_cdiv(K, 64) * 64 + 128) * sizeof(KEntry); }

template <int KS, int MODE>
int run_kn(const float* A, int M, int K, const Gather& g, int NB, const float* bias, float* out, int Mtot, int act,
           float slope, double flops, void* ktab_ws, size_t ws_avail, hipStream_t s) {
  const int Npix = NB * g.PH * g.PW;
  const bool vec = (K % 4 == 0) && aligned16(A);
  int tile = pick_tile(M, Npix);
  if (!vec && tile == 0) tile = 1;                  // the scalar-A variant is only instantiated for the small tiles
  const int tBM = tile == 0 ? 128 : (tile == 2 ? 32 : 64), tBN = tile == 1 ? 64 : 128;
  // mask-free kernels: reflection padding (every tap valid), full pixel tiles, full M tiles; a K tail is legal because
  // the A operand... would need masking -- so also require K % 16 == 0
  const bool nomask = MODE == 0 && vec && g.reflect && (Npix % tBN == 0) && (M % tBM == 0) && (K % BK == 0);
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ktab_ws) + ktab_bytes(K));
  int splits = (Mtot == M) ? kn_splits(M, Npix, K) : 1;
  if (splits > 1 && ws_avail < ktab_bytes(K) + (size_t)splits * M * Npix * sizeof(float)) splits = 1;
  const int Kpad = sg_cdiv(K, 64) * 64 + 128;      // the k-loop prefetches entries up to two tiles past the end
  const unsigned shw_ = (unsigned)(g.SH * g.SW), vstride = t_variant_stride;
  const KEntry* ktab = reinterpret_cast<const KEntry*>(cached_table(
      TabKey{0, K, Kpad, KS * KS, g.C1, g.C2, shw_, g.bcast2, nomask ? 1 : 0, vstride, 0, 0}, (size_t)Kpad * sizeof(KEntry), s,
      [&](void* dst) {
        hipLaunchKernelGGL(build_ktab_kernel, dim3(sg_cdiv(Kpad, 256)), dim3(256), 0, s, reinterpret_cast<KEntry*>(dst), K, Kpad,
                           KS * KS, g.C1, g.C2, shw_, g.bcast2, nomask ? 1 : 0, vstride);
      }));
  SG_ARG_CHECK(ktab != nullptr, "conv: device allocation of the k-split table failed");
  const size_t nout = (size_t)M * Npix;
  EpNCHW ep{out, bias, g.PH * g.PW, Mtot, M, Npix, act, slope, 0, 0, 1, 0, 0, 0, 0};
  if (splits > 1) ep = EpNCHW{slabs, nullptr, g.PH * g.PW, Mtot, M, Npix, SG_ACT_NONE, 0.f, nout, 0, 1, 0, 0, 0, 0};
  {
    SgProfScope prof(sg_igemm_kind(MODE, KS, tile), s, flops, 0);
    switch (tile) {
      case 0: launch_ab<typename CfgFor<KS>::C128, 128, 128, KS, MODE>(A, K, M, true, g, Npix, ktab, ep, splits, nomask, s); break;
      case 1: launch_ab<typename CfgFor<KS>::C64, 64, 64, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s); break;
      case 3: launch_ab<typename CfgFor<KS>::C64W, 64, 128, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s); break;
      default: launch_ab<typename CfgFor<KS>::C32, 32, 128, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s); break;
    }
  }
  if (splits > 1)
  {
    const int PHWo = g.PH * g.PW;
    if (nout % 4 == 0 && PHWo % 4 == 0 && aligned16(slabs) && aligned16(out))
      hipLaunchKernelGGL(slab_reduce_nchw_vec_kernel, dim3(sg_cdiv(nout / 4, 256)), dim3(256), 0, s, (const float4*)slabs,
                         (float4*)out, nout / 4, splits, bias, PHWo / 4, Mtot, act, slope);
    else
      hipLaunchKernelGGL(slab_reduce_nchw_kernel, dim3(sg_cdiv(nout, 256)), dim3(256), 0, s, (const float*)slabs, out, nout,
                         splits, bias, PHWo, Mtot, act, slope);
  }
  return 0;
}

// ---- channel-sparse conv forward ---------------------------------------------------------------------
// A masks_to_layout() layout has, per image, only the one-hot channels of the classes present plus the dense
// representation block non-zero (model.py:165-168 of the reference builds it that way): ~40 of 204 channels.
// Per image b the builder below makes a compact weight matrix Wc[b][m][k'] (k' = j*KS2 + t over the image's
// active channels list[b][j], zero padded to Kc), the matching k-table, and the K extent; the regular kernel then
// runs in batched mode (tiles never straddle images).
__global__ void build_sparse_fwd_kernel(const float* W, int M, int K, int KS2, int C1, int C2, unsigned shw, int bcast2,
                                        const int* list, const int* cnt, int L, int Kc, int Kpad, float* Wc,
                                        KEntry* ktab, int* kcnt, int tail_valid, const float* Wimg) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = cnt[b];
  if (i < M * Kc) {
    const int m = i / Kc, k = i - m * Kc;
    const int j = k / KS2, t = k - j * KS2;
    float v = 0.f;
    if (j < n) v = Wimg ? Wimg[(((size_t)b * M + m) * L + j) * KS2 + t] : W[(size_t)m * K + list[b * L + j] * KS2 + t];
    Wc[((size_t)b * M + m) * Kc + k] = v;
  }
  if (i < Kpad) {
    const int j = i / KS2, t = i - j * KS2;
    KEntry e;
    if (j < n) {
      const int c = list[b * L + j];
      const bool second = C2 > 0 && c >= C1;
      const unsigned cc = (unsigned)(second ? c - C1 : c);
      e.choff = (second && bcast2) ? cc : cc * shw;
      e.tapsel = (unsigned)t | (second ? 256u : 0u);
    } else {
      e.choff = 0u; e.tapsel = tail_valid ? 0u : (unsigned)KS2;     // Wc is zero there
    }
    ktab[(size_t)b * Kpad + i] = e;
  }
  if (i == 0) kcnt[b] = ((n * KS2 + BK - 1) / BK) * BK;
}

inline int sparse_kc(int L, int KS2) { return sg_cdiv(L * KS2, BK) * BK; }
inline int sparse_kpad(int L, int KS2) { return sg_cdiv(sparse_kc(L, KS2), 64) * 64 + 128; }
inline size_t sparse_fwd_ws(int NB, int M, int L, int KS2) {
  return (size_t)NB * ((size_t)sparse_kpad(L, KS2) * sizeof(KEntry) + (size_t)M * sparse_kc(L, KS2) * sizeof(float) + 64);
}

template <int KS>
int run_kn_sparse(const float* W, int M, int K, const Gather& g, int NB, const float* bias, float* out, int act,
                  float slope, const Sparse& sp, void* ws, hipStream_t s) {
  constexpr int KS2 = KS * KS;
  const int PHW = g.PH * g.PW, Npix = NB * PHW;
  const int Kc = sparse_kc(sp.L, KS2), Kpad = sparse_kpad(sp.L, KS2);
  KEntry* ktab = reinterpret_cast<KEntry*>(ws);
  float* Wc = reinterpret_cast<float*>(ktab + (size_t)NB * Kpad);
  int* kcnt = reinterpret_cast<int*>(Wc + (size_t)NB * M * Kc);
  int tile = pick_tile(M, Npix);
  const int tBM = tile == 0 ? 128 : (tile == 2 ? 32 : 64), tBN = tile == 1 ? 64 : 128;
  const bool nomask = g.reflect && (PHW % tBN == 0) && (M % tBM == 0);
  {
    const int work = M * Kc > Kpad ? M * Kc : Kpad;
    hipLaunchKernelGGL(build_sparse_fwd_kernel, dim3(sg_cdiv(work, 256), NB), dim3(256), 0, s, W, M, K, KS2, g.C1, g.C2,
                       (unsigned)(g.SH * g.SW), g.bcast2, sp.list, sp.cnt, sp.L, Kc, Kpad, Wc, ktab, kcnt, nomask ? 1 : 0,
                       sp.wimg);
  }
  EpNCHW ep{out, bias, PHW, M, M, Npix, act, slope, 0, 0, 1, 0, 0, 0, 0};
  // flops actually issued: the padded compact K of every image (bench.py prices the dominant kernel with this)
  const double flops = 2.0 * M * (double)Kc * Npix;
  t_batch = BatchInfo{}; t_batch.cols_per_batch = PHW; t_batch.nbatch = NB; t_batch.kcnt = kcnt; t_batch.a_stride = M * Kc; t_batch.b_stride = Kpad;
  {
    SgProfScope prof(sg_igemm_kind(0, KS, tile), s, flops, 0);
    switch (tile) {
      case 0: launch_ab<typename CfgFor<KS>::C128, 128, 128, KS, 0>(Wc, Kc, M, true, g, Npix, ktab, ep, 1, nomask, s); break;
      case 1: launch_ab<typename CfgFor<KS>::C64, 64, 64, KS, 0>(Wc, Kc, M, true, g, Npix, ktab, ep, 1, nomask, s); break;
      case 3: launch_ab<typename CfgFor<KS>::C64W, 64, 128, KS, 0>(Wc, Kc, M, true, g, Npix, ktab, ep, 1, nomask, s); break;
      default: launch_ab<typename CfgFor<KS>::C32, 32, 128, KS, 0>(Wc, Kc, M, true, g, Npix, ktab, ep, 1, nomask, s); break;
    }
  }
  t_batch = BatchInfo{};
  return 0;
}

// ---- stride-2 transposed gathers (dgrad of a strided conv, forward of a transposed conv) -------------------
// An output pixel (ph, pw) only receives taps with kh == (ph + pad) and kw == (pw + pad) modulo the stride: run as
// one dense problem, 3 of 4 gathered taps are structural zeros.  Instead: one GEMM per parity class (a, b) over the
// pixels of that class, with a compact weight matrix / k-table that lists only the class's taps -- 4x fewer MACs.
struct TapList { int n; int t[16]; };

// A_c[m][r*nt_c + i] = W[(r*B + m0 + m)*R + taps_c[i]] for the (up to) four parity classes c, packed back to back
// (W = [reduction dim][B][R] in memory)
struct PermClasses { int ncls; unsigned long long off[5]; TapList tl[4]; };
__global__ void permute_sub_kernel(const float* W, float* A, int Rdim, int B, int m0, int M, int R, PermClasses pc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pc.off[pc.ncls]) return;
  int c = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q) c += (q < pc.ncls && i >= pc.off[q]) ? 1 : 0;
  const int nt = pc.tl[c].n;
  const size_t K = (size_t)Rdim * nt, j = i - pc.off[c];
  const int m = (int)(j / K);
  const int k = (int)(j - (size_t)m * K);
  const int r = k / nt, ti = k - r * nt;
  int tap = pc.tl[c].t[0];
#pragma unroll
  for (int q = 1; q < 16; ++q) tap = (q == ti) ? pc.tl[c].t[q] : tap;      // no dynamic indexing of a kernel-argument array
  A[i] = W[((size_t)r * B + m0 + m) * R + tap];
}
__global__ void build_ktab_sub_kernel(KEntry* tab, int K, int Kpad, unsigned shw, int KS2, TapList tl) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  KEntry e;
  if (k < K) {
    const int r = k / tl.n, ti = k - r * tl.n;
    int tap = tl.t[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) tap = (q == ti) ? tl.t[q] : tap;
    e.choff = (unsigned)r * shw; e.tapsel = (unsigned)tap;
  } else {
    e.choff = 0u; e.tapsel = (unsigned)KS2;
  }
  tab[k] = e;
}

template <int KS>
int run_kn_parity(const float* W, int Rdim, int B, int m0, int M, const Gather& g, int NB, const float* bias, float* out,
                  int Mtot, int act, float slope, double flops, void* ws, size_t ws_bytes, hipStream_t s) {
  constexpr int KS2 = KS * KS;
  float* wbase = reinterpret_cast<float*>(ws);
  const unsigned shw = (unsigned)(g.SH * g.SW);
  ParityClasses par = {};
  PermClasses pc = {};
  int maxNpix = 0;
  bool vec = aligned16(wbase);
  double flops_issued = 0.0;
  for (int a = 0; a < 2; ++a) {
    for (int b = 0; b < 2; ++b) {
      TapList tl; tl.n = 0;
      for (int q = 0; q < 16; ++q) tl.t[q] = 0;
      for (int kh = a; kh < KS; kh += 2)
        for (int kw = b; kw < KS; kw += 2) tl.t[tl.n++] = kh * KS + kw;
      const int ph0 = ((a - g.pad) % 2 + 2) % 2, pw0 = ((b - g.pad) % 2 + 2) % 2;
      const int PHa = g.PH > ph0 ? (g.PH - ph0 + 1) / 2 : 0, PWb = g.PW > pw0 ? (g.PW - pw0 + 1) / 2 : 0;
      if (PHa * PWb == 0 || tl.n == 0) continue;
      const int c = par.ncls++;
      const int K = Rdim * tl.n, Kpad = sg_cdiv(K, 64) * 64 + 128;
      par.K[c] = K; par.PH[c] = PHa; par.PW[c] = PWb; par.ph0[c] = ph0; par.pw0[c] = pw0;
      par.Npix[c] = NB * PHa * PWb;
      par.aoff[c] = (unsigned)pc.off[c];
      pc.tl[c] = tl;
      pc.off[c + 1] = pc.off[c] + (unsigned long long)M * K;
      par.ktab[c] = cached_table(TabKey{1, K, Kpad, (long long)shw, KS2, a, b, KS, 0, 0, 0, 0}, (size_t)Kpad * sizeof(KEntry), s,
                                 [&](void* dst) {
                                   hipLaunchKernelGGL(build_ktab_sub_kernel, dim3(sg_cdiv(Kpad, 256)), dim3(256), 0, s,
                                                      reinterpret_cast<KEntry*>(dst), K, Kpad, shw, KS2, tl);
                                 });
      SG_ARG_CHECK(par.ktab[c] != nullptr, "conv: device allocation of a k-split table failed");
      vec = vec && (K % 4 == 0);
      maxNpix = par.Npix[c] > maxNpix ? par.Npix[c] : maxNpix;
      flops_issued += flops * (4.0 * tl.n * PHa * PWb) / ((double)KS2 * g.PH * g.PW);
    }
  }
  if (par.ncls == 0) return 0;
  pc.ncls = par.ncls;
  SG_ARG_CHECK(ws_bytes >= pc.off[pc.ncls] * sizeof(float), "conv: parity workspace too small");
  hipLaunchKernelGGL(permute_sub_kernel, dim3(sg_cdiv(pc.off[pc.ncls], 256)), dim3(256), 0, s, W, wbase, Rdim, B, m0, M, KS2, pc);
  long sumNpix = 0;
  for (int c = 0; c < par.ncls; ++c) sumNpix += par.Npix[c];
  int tile = pick_tile(M, (int)sumNpix);          // all classes share the launch: the chip sees the sum of their tiles
  (void)maxNpix;
  long t128 = 0;
  for (int c = 0; c < par.ncls; ++c) t128 += (long)sg_cdiv(M, 128) * sg_cdiv(par.Npix[c], 128);
  if (tile == 0 && (!vec || t128 < 384)) tile = 1;   // no split-K here
  const int tBN = tile == 1 ? 64 : 128;
  par.tile0[0] = 0;
  for (int c = 0; c < par.ncls; ++c) par.tile0[c + 1] = par.tile0[c] + sg_cdiv(par.Npix[c], tBN);
  for (int c = par.ncls + 1; c < 5; ++c) par.tile0[c] = par.tile0[par.ncls];
  Gather gs = g;
  gs.PH = par.PH[0]; gs.PW = par.PW[0]; gs.pstep = 2; gs.ph0 = par.ph0[0]; gs.pw0 = par.pw0[0];
  const EpNCHW ep{out, bias, par.PH[0] * par.PW[0], Mtot, M, par.Npix[0], act, slope, 0, par.PW[0], 2, par.ph0[0], par.pw0[0],
                  g.PW, g.PH * g.PW};
  const KEntry* kt0 = reinterpret_cast<const KEntry*>(par.ktab[0]);
  t_batch = BatchInfo{};
  t_batch.par = par;
  {
    SgProfScope prof(sg_igemm_kind(1, KS, tile), s, flops_issued, 0);
    switch (tile) {
      case 0: launch_ab<typename CfgFor<KS>::C128, 128, 128, KS, 1>(wbase, par.K[0], M, true, gs, par.Npix[0], kt0, ep, 1, false, s); break;
      case 1: launch_ab<typename CfgFor<KS>::C64, 64, 64, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s); break;
      case 3: launch_ab<typename CfgFor<KS>::C64W, 64, 128, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s); break;
      default: launch_ab<typename CfgFor<KS>::C32, 32, 128, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s); break;
    }
  }
  t_batch = BatchInfo{};
  return 0;
}
inline size_t parity_ws(int M, int Rdim, int KS2) { return (size_t)M * Rdim * KS2 * sizeof(float) + 64; }
int run_kn_parity_ks(int KS, const float* W, int Rdim, int B, int m0, int M, const Gather& g, int NB, const float* bias,
                     float* out, int Mtot, int act, float slope, double flops, void* ws, size_t ws_bytes, hipStream_t s) {
  switch (KS) {
    case 3: return run_kn_parity<3>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
    case 4: return run_kn_parity<4>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
    case 7: return run_kn_parity<7>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
  }
  return -1;
}

template <int MODE>
int run_kn_ks(int KS, const float* A, int M, int K, const Gather& g, int NB, const float* bias, float* out, int Mtot,
              int act, float slope, double flops, void* ktab_ws, size_t ws_avail, hipStream_t s) {
  switch (KS) {
    case 1: return run_kn<1, MODE>(A, M, K, g, NB, bias, out, Mtot, act, slope, flops, ktab_ws, ws_avail, s);
    case 3: return run_kn<3, MODE>(A, M, K, g, NB, bias, out, Mtot, act, slope, flops, ktab_ws, ws_avail, s);
    case 4: return run_kn<4, MODE>(A, M, K, g, NB, bias, out, Mtot, act, slope, flops, ktab_ws, ws_avail, s);
    case 7: return run_kn<7, MODE>(A, M, K, g, NB, bias, out, Mtot, act, slope, flops, ktab_ws, ws_avail, s);
  }
  return -1;
}

// ---- wgrad-shaped GEMM: K = (img, pix), N = (c, taps) --------------------------------------------
// inverse of the per-image channel lists: inv[b][c] = position of c in list[b] or -1
__global__ void sparse_inv_kernel(const int* list, const int* cnt, int L, int C, int* inv) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int pos = -1;
  const int n = cnt[b];
  for (int j = 0; j < n; ++j) pos = list[b * L + j] == c ? j : pos;
  inv[b * C + c] = pos;
}
// gw[m][c][t] = sum_b slab[b][m][t][inv[b][c]]  (images in ascending order => deterministic)
__global__ void sparse_wgrad_reduce_kernel(const float* slabs, const int* inv, float* gw, int M, int C, int KS2, int cpad,
                                           int NB) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * C * KS2) return;
  const int t = (int)(i % KS2);
  const int c = (int)((i / KS2) % C);
  const int m = (int)(i / ((size_t)KS2 * C));
  float v = 0.f;
  for (int b = 0; b < NB; ++b) {
    const int j = inv[b * C + c];
    if (j >= 0) v += slabs[(((size_t)b * M + m) * KS2 + t) * cpad + j];
  }
  gw[i] = v;
}
// out[b][i] = sum_q ws[(b*S + q)][i]: k-chunks of one image (fixed order)
__global__ void slab_group_reduce_kernel(const float* ws, float* out, size_t n, int S, int NB) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NB) return;
  const size_t b = i / n, r = i - b * n;
  float v = 0.f;
  for (int q = 0; q < S; ++q) v += ws[(b * S + q) * n + r];
  out[i] = v;
}
// gwimg[b][m][j][t] = slab[b][m][t][j] (per-image weight gradients of a factored layout conv; zero beyond the image's list)
__global__ void sparse_wgrad_perimage_kernel(const float* slabs, const int* cnt, float* gwimg, int M, int L, int KS2, int cpad,
                                             int NB) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)NB * M * L * KS2) return;
  const int t = (int)(i % KS2);
  const int j = (int)((i / KS2) % L);
  const size_t bm = i / ((size_t)KS2 * L);
  const int b = (int)(bm / M);
  gwimg[i] = j < cnt[b] ? slabs[(bm * KS2 + t) * cpad + j] : 0.f;
}
// gw[m][c][t] = sum_z slab[z][m][t][c]: un-permutes the tap-major slabs of the weight-gradient GEMM (the GEMM epilogue
// writes them coalesced; scattering 4-byte stores at stride KS2*4 from there cost 8x write amplification in HBM)
__global__ void wgrad_unpermute_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ gw, int M, int C, int KS2,
                                              int cpad, int S) {
  // grid (ceil(C*KS2 / 256), M): one 32-bit division per thread.  (An LDS-transposed variant with fully coalesced slab reads
  // was measured SLOWER -- 21.7 vs 15.7 us per launch: the strided reads hit in L2, the extra barrier and the thinner loops
  // do not pay.)
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (unsigned)(C * KS2)) return;
  const unsigned c = j / (unsigned)KS2, t = j - c * (unsigned)KS2;
  const int m = blockIdx.y;
  const size_t zs = (size_t)M * KS2 * cpad;
  const float* p = slabs + ((size_t)m * KS2 + t) * cpad + c;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += p[(size_t)z * zs];
  gw[(size_t)m * C * KS2 + j] = v;
}
inline size_t sparse_wgrad_ws(int NB, int M, int C, int L, int KS2) {
  return (size_t)NB * M * (sg_cdiv(L, 128) * 128) * KS2 * sizeof(float) + (size_t)NB * C * sizeof(int);
}
// launch plan of a weight-gradient GEMM (shared by the workspace query and the launcher)
struct NkPlan { int tile; bool tap; int cpad; int splits; };
inline NkPlan nk_plan(int M, int C, int KS2, int Kpix, bool two) {
  NkPlan p;
  p.tap = C >= 48 || two;                                   // tap-major columns pad C to whole tiles: too wasteful for RGB inputs
  p.tile = M <= 32 ? 2 : 1;
  if (M > 64 && p.tap) {
    // 128x128 tiles issue ~0.7x the instructions per MFMA of 64x64 tiles, but pad M and C to multiples of 128
    const double w128 = (double)sg_cdiv(M, 128) * 128 * sg_cdiv(C, 128) * 128;
    const double w64 = (double)sg_cdiv(M, 64) * 64 * sg_cdiv(C, 64) * 64;
    if (w128 * 0.7 < w64) p.tile = 0;
  }
  // opt-in (SG_NK_TILE3=1): 64 rows x 128 columns where the 64x64 tile was chosen and 128-wide channel tiles pad no further
  // (each wave owns 32x64: two accumulators per fragment read).  Parity-tested; none of the benchmark's 64x64 weight-gradient
  // launches qualifies (their inputs have 64 channels), so it is not the default.
  static int nk3 = -1;
  if (nk3 < 0) { const char* e = getenv("SG_NK_TILE3"); nk3 = e ? atoi(e) : 0; }
  if (nk3 && p.tile == 1 && p.tap && sg_cdiv(C, 128) * 128 == sg_cdiv(C, 64) * 64) p.tile = 3;
  const int BMt = p.tile == 0 ? 128 : ((p.tile == 1 || p.tile == 3) ? 64 : 32), BNt = p.tile == 1 ? 64 : 128;
  p.cpad = p.tap ? sg_cdiv(C, BNt) * BNt : 0;
  const long tiles = (long)sg_cdiv(M, BMt) * (p.tap ? (long)KS2 * (p.cpad / BNt) : (long)sg_cdiv((long)C * KS2, BNt));
  const int target = p.tile == 0 ? 768 : 1024;       // resident workgroups on 256 CUs
  int s = (int)((target + tiles - 1) / tiles);
  const int maxs = Kpix / (BK * 8) > 0 ? Kpix / (BK * 8) : 1;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  p.splits = s < 1 ? 1 : s;
  return p;
}

// general (c, tap)-ordered loader: only for few-channel inputs (RGB crops / images)
template <int KS>
void launch_nk_general(int tile, const float* A, int M, int Mtot, int PQ, const Gather& g, int Ncols, const EpRowMajor& ep,
                       int Kpix, int splits, hipStream_t s, const Sparse* sp = nullptr, int zdiv = 1) {
  const FastDiv dPQ((unsigned)PQ);
  const int* sl = sp ? sp->list : nullptr; const int* sc = sp ? sp->cnt : nullptr; const int L = sp ? sp->L : 0;
  if (tile == 2)
    launch_cfg<CfgW32>(LoadPixK<32>{A, M, Mtot, PQ, dPQ}, LoadGatherNK<128, KS, false, true, NSW>{g, Ncols, sl, sc, L, zdiv}, ep,
                       M, Ncols, Kpix, splits, s);
  else
    launch_cfg<CfgW64>(LoadPixK<64>{A, M, Mtot, PQ, dPQ}, LoadGatherNK<64, KS, false, true, NSW>{g, Ncols, sl, sc, L, zdiv}, ep,
                       M, Ncols, Kpix, splits, s);
}

template <class CFG, int BMv, int BNv>
void launch_nk_tap(const float* A, int M, int Mtot, int PQ, bool vecA, const Gather& g, int KS, int Ccols, int cpad,
                   const Sparse* sp, bool nomask, const EpWgrad& ep, int Kpix, int splits, hipStream_t s) {
  const FastDiv dPQ((unsigned)PQ), dPW((unsigned)g.PW);
  const int* sl = sp ? sp->list : nullptr; const int* sc = sp ? sp->cnt : nullptr; const int L = sp ? sp->L : 0;
  const int Nv = KS * KS * cpad;
  const bool two = g.C2 > 0;
#define SG_TAP_B(TWOv, MASKv) LoadTapNK<BNv, TWOv, MASKv, CFG::NSUB>{g, KS, Ccols, cpad, sl, sc, L, dPQ, dPW}
  if (vecA) {
    const LoadPixKVec<BMv> al{A, M, Mtot, PQ, dPQ};
    if (two) launch_cfg<CFG>(al, SG_TAP_B(true, true), ep, M, Nv, Kpix, splits, s);
    else if (nomask) launch_cfg<CFG>(al, SG_TAP_B(false, false), ep, M, Nv, Kpix, splits, s);
    else launch_cfg<CFG>(al, SG_TAP_B(false, true), ep, M, Nv, Kpix, splits, s);
  } else {
    const LoadPixK<BMv> al{A, M, Mtot, PQ, dPQ};
    if (two) launch_cfg<CFG>(al, SG_TAP_B(true, true), ep, M, Nv, Kpix, splits, s);
    else if (nomask) launch_cfg<CFG>(al, SG_TAP_B(false, false), ep, M, Nv, Kpix, splits, s);
    else launch_cfg<CFG>(al, SG_TAP_B(false, true), ep, M, Nv, Kpix, splits, s);
  }
#undef SG_TAP_B
}

int run_nk_ks(int KS, const float* A, int M, int Mtot, const Gather& g, int NB, float* out, void* ws, size_t ws_bytes,
              double flops, hipStream_t s, const Sparse* sp = nullptr) {
  const int PQ = g.PH * g.PW, KS2 = KS * KS;
  const int Kpix = NB * PQ;
  const int C = g.C1 + g.C2;
  // channel-sparse input (see run_kn_sparse): one k-chunk per image, compact columns, per-image slabs that
  // sparse_wgrad_reduce_kernel scatters back to the dense gradient in a fixed order
  const int Ccols = sp ? sp->L : C;
  const int Ncols = Ccols * KS2;
  NkPlan pl = nk_plan(M, Ccols, KS2, Kpix, g.C2 > 0);
  if (sp && sp->gwimg && g.C2 == 0 && sg_cdiv(Ccols, 64) * 64 >= 3 * Ccols) {
    // a handful of channels per image (factored layout convs): the tap-major layout would pad every tap to a 64-column
    // tile; use the (channel, tap)-ordered gather instead, L*KS2 columns, k-chunks inside each image for occupancy
    const int tile = M <= 32 ? 2 : 1;
    const long tiles = (long)sg_cdiv(M, tile == 2 ? 32 : 64) * sg_cdiv(Ncols, tile == 2 ? 128 : 64) * NB;
    int S = (int)((1024 + tiles - 1) / tiles);
    if (S > 128 / Ccols) S = 128 / Ccols;           // slabs fit the workspace sized for the tap-major path
    if (S > PQ / 256) S = PQ / 256;
    if (S < 1) S = 1;
    const int kcs = sg_cdiv(sg_cdiv(PQ, S), BK) * BK;
    S = sg_cdiv(PQ, kcs);
    const size_t mnc = (size_t)M * Ncols;
    SG_ARG_CHECK(ws && ws_bytes >= mnc * sizeof(float) * (size_t)S * NB, "wgrad: workspace too small");
    float* dstp = S > 1 ? reinterpret_cast<float*>(ws) : sp->gwimg;
    const EpRowMajor ep{dstp, nullptr, M, Ncols, Ncols, SG_ACT_NONE, 0.f, mnc};
    t_batch = BatchInfo{}; t_batch.kimg = PQ; t_batch.ksplit = S; t_batch.kcs = kcs;
    t_grid_z = NB * S;
    {
      SgProfScope prof(sg_igemm_kind(2, KS, tile), s, 2.0 * M * (double)Ncols * Kpix, 0);
      switch (KS) {
        case 1: launch_nk_general<1>(tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, 1, s, sp, S); break;
        case 3: launch_nk_general<3>(tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, 1, s, sp, S); break;
        case 4: launch_nk_general<4>(tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, 1, s, sp, S); break;
        case 7: launch_nk_general<7>(tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, 1, s, sp, S); break;
      }
    }
    t_batch = BatchInfo{};
    t_grid_z = 0;
    if (S > 1)
      hipLaunchKernelGGL(slab_group_reduce_kernel, dim3(sg_cdiv(mnc * NB, 256)), dim3(256), 0, s, (const float*)ws, sp->gwimg,
                         mnc, S, NB);
    return 0;
  }
  if (sp) pl.tap = true;
  if (sp && pl.cpad == 0) pl.cpad = sg_cdiv(Ccols, pl.tile == 1 ? 64 : 128) * (pl.tile == 1 ? 64 : 128);
  int splits = sp ? NB : pl.splits;
  // slab size: the tap-major path keeps whole padded channel tiles, [m][t][cpad]
  const size_t mn = pl.tap ? (size_t)M * KS2 * pl.cpad : (size_t)M * Ncols;
  if (!sp && splits > 1 && ws_bytes < mn * sizeof(float) * (size_t)splits) splits = (int)(ws_bytes / (mn * sizeof(float)));
  if (!sp && splits < 2) splits = 1;
  SG_ARG_CHECK(!pl.tap || (ws && ws_bytes >= mn * sizeof(float) * (size_t)splits), "wgrad: workspace too small");
  const int kchunk = sp ? PQ : sg_cdiv(sg_cdiv(Kpix, splits), 64) * 64;      // multiple of every BKT
  splits = sg_cdiv(Kpix, kchunk);
  if (sp) { t_fixed_kchunk = PQ; flops = 2.0 * M * (double)Ncols * Kpix; }
  float* dst = (splits > 1 || sp || pl.tap) ? reinterpret_cast<float*>(ws) : out;
  {
    SgProfScope prof(sg_igemm_kind(2, KS, pl.tile), s, flops, 0);
    if (pl.tap) {
      const EpWgrad ep{dst, M, pl.cpad, KS2, mn};
      const bool vecA = (PQ % 4 == 0) && aligned16(A);
      // mask-free gather: reflection padding and whole 16-pixel k-tiles (split chunks are multiples of 64)
      const bool nomask = g.reflect && (Kpix % (BK * NSW) == 0) && (!sp || PQ % (BK * NSW) == 0);
      switch (pl.tile) {
        case 0: launch_nk_tap<CfgW128, 128, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s); break;
        case 1: launch_nk_tap<CfgW64, 64, 64>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s); break;
        case 3: launch_nk_tap<CfgW64W, 64, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s); break;
        default: launch_nk_tap<CfgW32, 32, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s); break;
      }
    } else {
      const EpRowMajor ep{dst, nullptr, M, Ncols, Ncols, SG_ACT_NONE, 0.f, mn};
      switch (KS) {
        case 1: launch_nk_general<1>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 3: launch_nk_general<3>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 4: launch_nk_general<4>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 7: launch_nk_general<7>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        default: t_fixed_kchunk = 0; return -1;
      }
    }
  }
  t_fixed_kchunk = 0;
  const size_t nout = (size_t)M * C * KS2;
  if (sp && sp->gwimg) {
    const size_t n = (size_t)NB * M * sp->L * KS2;
    hipLaunchKernelGGL(sparse_wgrad_perimage_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, (const float*)ws, sp->cnt,
                       sp->gwimg, M, sp->L, KS2, pl.cpad, NB);
    return 0;
  }
  if (sp) {
    int* inv = reinterpret_cast<int*>(reinterpret_cast<float*>(ws) + (size_t)NB * mn);
    hipLaunchKernelGGL(sparse_inv_kernel, dim3(sg_cdiv(C, 256), NB), dim3(256), 0, s, sp->list, sp->cnt, sp->L, C, inv);
    hipLaunchKernelGGL(sparse_wgrad_reduce_kernel, dim3(sg_cdiv(nout, 256)), dim3(256), 0, s, (const float*)ws,
                       (const int*)inv, out, M, C, KS2, pl.cpad, NB);
    return 0;
  }
  if (pl.tap)
    hipLaunchKernelGGL(wgrad_unpermute_reduce_kernel, dim3(sg_cdiv((size_t)C * KS2, 256), M), dim3(256), 0, s, (const float*)ws, out,
                       M, C, KS2, pl.cpad, splits);
  else if (splits > 1)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(sg_cdiv(mn, 256)), dim3(256), 0, s, (const float*)ws, out, mn, splits);
  return 0;
}


int check_desc(const sgConvDesc* d, const char* who) {
  SG_ARG_CHECK(d != nullptr, "%s: null desc", who);
  SG_ARG_CHECK(d->KS == 1 || d->KS == 3 || d->KS == 4 || d->KS == 7, "%s: kernel size %d unsupported", who, d->KS);
  SG_ARG_CHECK(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported", who, d->stride);
  SG_ARG_CHECK(d->upsample == 1 || d->upsample == 2, "%s: upsample %d unsupported", who, d->upsample);
  SG_ARG_CHECK(d->N > 0 && d->C1 > 0 && d->C2 >= 0 && d->Cout > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0,
               "%s: non-positive dimension", who);
  SG_ARG_CHECK(!d->pad_reflect || d->pad < d->H * d->upsample, "%s: reflect pad too large", who);
  const double lim = SG_MAX_ELEMS;     // 32-bit offsets; 2^29 elements with buffer-load masking (see SG_MAX_ELEMS)
  SG_ARG_CHECK((double)d->N * (d->C1 + d->C2) * d->H * d->W <= lim && (double)d->N * d->Cout * d->OH * d->OW <= lim &&
                   (double)d->N * (d->C1 + d->C2) * (d->H * d->upsample + 2.0 * d->pad) *
                           (d->W * d->upsample + 2.0 * d->pad) <= lim &&
                   (double)d->Cout * (d->C1 + d->C2) * d->KS * d->KS <= lim,
               "%s: a tensor has more than %.0f elements (32-bit offsets%s)", who, lim,
               SG_BUFLOAD ? ", buffer-load range masking" : "");
  return 0;
}


inline size_t wgrad_ws(int M, int C, int KS2, int Kpix, bool two) {
  const NkPlan pl = nk_plan(M, C, KS2, Kpix, two);
  return (size_t)pl.splits * (size_t)M * KS2 * (pl.tap ? pl.cpad : C) * sizeof(float);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" size_t sg_conv2d_ws_bytes(const sgConvDesc* d, int kind) {
  if (!d) return 0;
  const size_t wbytes = (size_t)d->Cout * (d->C1 + d->C2) * d->KS * d->KS * sizeof(float);
  const int Kmax = (d->Cout > d->C1 + d->C2 ? d->Cout : d->C1 + d->C2) * d->KS * d->KS;
  const size_t kt = 4 * (ktab_bytes(Kmax) + 64 * sizeof(KEntry));   // up to four parity-class tables
  const int Cin = d->C1 + d->C2, R = d->KS * d->KS;
  const size_t sl_f = kn_slab_bytes(d->Cout, d->N * d->OH * d->OW, Cin * R);                 // conv fwd / convT fwd
  const int GH = d->H * d->upsample + (d->pad_reflect ? 2 * d->pad : 0), GW = d->W * d->upsample + (d->pad_reflect ? 2 * d->pad : 0);
  const size_t sl_d = kn_slab_bytes(Cin, d->N * GH * GW, d->Cout * R);                        // conv dgrad (all channels)
  const size_t sl_t = kn_slab_bytes(d->C1, d->N * d->H * d->W, d->Cout * R);                  // convT dgrad
  const size_t sl = sl_f > sl_d ? (sl_f > sl_t ? sl_f : sl_t) : (sl_d > sl_t ? sl_d : sl_t);
  if (kind == 0) return wbytes + kt + sl;          // fwd: k-split table, split-K slabs (+ transposed weights for convT)
  if (kind == 1) return wbytes + kt + sl;          // dgrad: transposed weights + k-split table + split-K slabs
  const int M = d->Cout > (d->C1 + d->C2) ? d->Cout : (d->C1 + d->C2);
  const int Kp = d->N * (d->OH * d->OW > d->H * d->W ? d->OH * d->OW : d->H * d->W);
  size_t a = wgrad_ws(d->Cout, d->C1 + d->C2, d->KS * d->KS, d->N * d->OH * d->OW, d->C2 > 0);
  size_t b = wgrad_ws(d->C1 + d->C2, d->Cout, d->KS * d->KS, d->N * d->H * d->W, false);
  (void)M; (void)Kp;
  const size_t cs = sg_channel_sum_ws_bytes(d->Cout);
  a = a > b ? a : b;
  return a > cs ? a : cs;
}

extern "C" int sg_conv2d_fwd(const sgConvDesc* d, const float* x1, const float* x2, const float* w, const float* bias,
                             float* y, int act, float slope, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_fwd")) return -1;
  SG_ARG_CHECK(x1 && w && y && ws, "sg_conv2d_fwd: null pointer");
  SG_ARG_CHECK(ws_bytes >= ktab_bytes((d->C1 + d->C2) * d->KS * d->KS), "sg_conv2d_fwd: workspace too small");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd: C2>0 but x2 null");
  hipStream_t s = (hipStream_t)stream;
  const int Cin = d->C1 + d->C2, K = Cin * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const double flops = 2.0 * d->Cout * K * (double)d->N * d->OH * d->OW;
  int rc = run_kn_ks<0>(d->KS, w, d->Cout, K, g, d->N, bias, y, d->Cout, act, slope, flops, ws, ws_bytes, s);
  SG_LAUNCH_CHECK("sg_conv2d_fwd");
  return rc;
}

extern "C" int sg_conv2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, int c_begin, int c_end,
                               void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_dgrad")) return -1;
  const int Cin = d->C1 + d->C2, R = d->KS * d->KS;
  SG_ARG_CHECK(gy && w && gx && ws, "sg_conv2d_dgrad: null pointer");
  SG_ARG_CHECK(0 <= c_begin && c_begin < c_end && c_end <= Cin, "sg_conv2d_dgrad: bad channel range [%d,%d)", c_begin, c_end);
  SG_ARG_CHECK(ws_bytes >= (size_t)d->Cout * Cin * R * sizeof(float) + ktab_bytes(d->Cout * R),
               "sg_conv2d_dgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  // gradient w.r.t. the logical (upsampled, reflect-padded) input grid
  const int GH = d->H * d->upsample + (d->pad_reflect ? 2 * d->pad : 0);
  const int GW = d->W * d->upsample + (d->pad_reflect ? 2 * d->pad : 0);
  const int pad = d->pad_reflect ? 0 : d->pad;
  Gather g = make_gather(gy, nullptr, d->Cout, 0, d->OH, d->OW, 1, GH, GW, d->stride, pad, 0);
  const int M = c_end - c_begin, K = d->Cout * R;
  // algorithmic flops of a dgrad = those of the forward conv restricted to the requested input channels
  const double flops = 2.0 * M * (double)d->Cout * R * d->N * d->OH * d->OW;
  if (d->stride == 2 && d->KS >= 3) {          // parity classes: only the taps that can hit each output pixel
    SG_ARG_CHECK(ws_bytes >= parity_ws(M, d->Cout, R), "sg_conv2d_dgrad: workspace too small");
    int rc = run_kn_parity_ks(d->KS, w, d->Cout, Cin, c_begin, M, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops, ws,
                              ws_bytes, s);
    SG_LAUNCH_CHECK("sg_conv2d_dgrad");
    return rc;
  }
  float* wt = reinterpret_cast<float*>(ws);      // [Cin][Cout][R]
  const size_t nw = (size_t)d->Cout * Cin * R;
  hipLaunchKernelGGL(permute_w_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, w, wt, d->Cout, Cin, R);
  int rc = run_kn_ks<1>(d->KS, wt + (size_t)c_begin * K, M, K, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops,
                        wt + nw, ws_bytes - nw * sizeof(float), s);
  SG_LAUNCH_CHECK("sg_conv2d_dgrad");
  return rc;
}

// ---- dgrad w.r.t. the ACTUAL input of a reflect-padded 3x3 conv (ResnetBlock, layers.py:251-270) -----------------
static bool dgrad_folded_ok(const sgConvDesc* d) {
  return d && d->pad_reflect && d->pad == 1 && d->KS == 3 && d->stride == 1 && d->upsample == 1 && d->C2 == 0 &&
         d->H >= 3 && d->W >= 3 && d->OH == d->H && d->OW == d->W &&
         9.0 * d->N * d->Cout * d->OH * d->OW <= SG_MAX_ELEMS;     // the nine pre-folded copies of gy are ONE gathered source
}
extern "C" int sg_conv2d_dgrad_folded_supported(const sgConvDesc* d) { return dgrad_folded_ok(d) ? 1 : 0; }
extern "C" size_t sg_conv2d_dgrad_folded_ws_bytes(const sgConvDesc* d) {
  if (!dgrad_folded_ok(d)) return 0;
  return sg_conv2d_ws_bytes(d, 1) + 9 * (size_t)d->N * d->Cout * d->OH * d->OW * sizeof(float) + 256 +
         kn_slab_bytes(d->C1, d->N * d->H * d->W, d->Cout * 9);
}
extern "C" int sg_conv2d_dgrad_folded(const sgConvDesc* d, const float* gy, const float* w, float* gx, int c_begin, int c_end,
                                      void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_dgrad_folded")) return -1;
  SG_ARG_CHECK(dgrad_folded_ok(d), "sg_conv2d_dgrad_folded: unsupported desc (needs ReflectionPad(1) + 3x3, stride 1)");
  const int Cin = d->C1, R = 9;
  SG_ARG_CHECK(gy && w && gx && ws, "sg_conv2d_dgrad_folded: null pointer");
  SG_ARG_CHECK(0 <= c_begin && c_begin < c_end && c_end <= Cin, "sg_conv2d_dgrad_folded: bad channel range");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_dgrad_folded_ws_bytes(d), "sg_conv2d_dgrad_folded: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* wt = reinterpret_cast<float*>(ws);      // [Cin][Cout][R]
  const size_t nw = (size_t)d->Cout * Cin * R;
  hipLaunchKernelGGL(permute_w_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, w, wt, d->Cout, Cin, R);
  float* V = wt + ((nw + 63) / 64) * 64;         // nine pre-folded copies of gy
  const size_t VS = (size_t)d->N * d->Cout * d->OH * d->OW;
  hipLaunchKernelGGL(reflect_variants_kernel, dim3(sg_cdiv(VS, 256)), dim3(256), 0, s, gy, V, (size_t)d->N * d->Cout, d->OH,
                     d->OW, VS);
  Gather g = make_gather(V, nullptr, d->Cout, 0, d->OH, d->OW, 1, d->H, d->W, 1, 1, 0);
  const int M = c_end - c_begin, K = d->Cout * R;
  const double flops = 2.0 * M * (double)d->Cout * R * d->N * d->OH * d->OW;
  char* rest = reinterpret_cast<char*>(V + 9 * VS);
  t_variant_stride = (unsigned)VS;
  int rc = run_kn_ks<1>(3, wt + (size_t)c_begin * K, M, K, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops, rest,
                        ws_bytes - (size_t)(rest - reinterpret_cast<char*>(ws)), s);
  t_variant_stride = 0;
  SG_LAUNCH_CHECK("sg_conv2d_dgrad_folded");
  return rc;
}

extern "C" int sg_conv2d_wgrad(const sgConvDesc* d, const float* gy, const float* x1, const float* x2, float* gw,
                               float* gb, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad")) return -1;
  SG_ARG_CHECK(gy && x1 && gw, "sg_conv2d_wgrad: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad: C2>0 but x2 null");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const double flops = 2.0 * d->Cout * (double)(d->C1 + d->C2) * d->KS * d->KS * d->N * d->OH * d->OW;
  if (int rc = run_nk_ks(d->KS, gy, d->Cout, d->Cout, g, d->N, gw, ws, ws ? ws_bytes : 0, flops, s)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad");
  if (gb) return sg_channel_sum(gy, gb, d->N, d->Cout, d->OH * d->OW, ws, ws ? ws_bytes : 0, stream);
  return 0;
}

// ConvTranspose2d: y[n,co,oh,ow] = b + sum_{ci,kh,kw} w[ci,co,kh,kw] x[n,ci,(oh+p-kh)/s,(ow+p-kw)/s]

// ---- channel-sparse variants (layers fed by a masks_to_layout() layout) ---------------------------
static int check_sparse(const sgConvDesc* d, const int32_t* list, const int32_t* cnt, int L, const char* who) {
  SG_ARG_CHECK(list && cnt, "%s: null channel list", who);
  SG_ARG_CHECK(L > 0 && L <= d->C1 + d->C2, "%s: L=%d outside (0, %d]", who, L, d->C1 + d->C2);
  return 0;
}
extern "C" size_t sg_conv2d_sparse_ws_bytes(const sgConvDesc* d, int L, int kind) {
  if (!d || L <= 0) return 0;
  const int KS2 = d->KS * d->KS;
  if (kind == 0) return sparse_fwd_ws(d->N, d->Cout, L, KS2);
  const size_t a = sparse_wgrad_ws(d->N, d->Cout, d->C1 + d->C2, L, KS2), cs = sg_channel_sum_ws_bytes(d->Cout);
  return a > cs ? a : cs;
}
extern "C" int sg_conv2d_fwd_sparse(const sgConvDesc* d, const float* x1, const float* x2, const float* w,
                                    const float* bias, const int32_t* chan_list, const int32_t* chan_cnt, int L, float* y,
                                    int act, float slope, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_fwd_sparse") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_fwd_sparse")) return -1;
  SG_ARG_CHECK(x1 && w && y && ws, "sg_conv2d_fwd_sparse: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd_sparse: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 0), "sg_conv2d_fwd_sparse: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int K = (d->C1 + d->C2) * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, nullptr};
  int rc = -1;
  switch (d->KS) {
    case 1: rc = run_kn_sparse<1>(w, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 3: rc = run_kn_sparse<3>(w, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 4: rc = run_kn_sparse<4>(w, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 7: rc = run_kn_sparse<7>(w, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
  }
  SG_LAUNCH_CHECK("sg_conv2d_fwd_sparse");
  return rc;
}
extern "C" int sg_conv2d_wgrad_sparse(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                                      const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gw, float* gb,
                                      void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad_sparse") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_wgrad_sparse")) return -1;
  SG_ARG_CHECK(gy && x1 && gw && ws, "sg_conv2d_wgrad_sparse: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad_sparse: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 2), "sg_conv2d_wgrad_sparse: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, nullptr};
  if (int rc = run_nk_ks(d->KS, gy, d->Cout, d->Cout, g, d->N, gw, ws, ws_bytes, 0.0, s, &sp)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad_sparse");
  if (gb) return sg_channel_sum(gy, gb, d->N, d->Cout, d->OH * d->OW, ws, ws_bytes, stream);
  return 0;
}
// ---- per-image weights (factored layout convs) ---------------------------------------------------------------
extern "C" int sg_conv2d_fwd_perimage(const sgConvDesc* d, const float* x1, const float* x2, const float* wimg,
                                      const float* bias, const int32_t* chan_list, const int32_t* chan_cnt, int L, float* y,
                                      int act, float slope, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_fwd_perimage") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_fwd_perimage")) return -1;
  SG_ARG_CHECK(x1 && wimg && y && ws, "sg_conv2d_fwd_perimage: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd_perimage: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 0), "sg_conv2d_fwd_perimage: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int K = (d->C1 + d->C2) * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, wimg, nullptr};
  int rc = -1;
  switch (d->KS) {
    case 1: rc = run_kn_sparse<1>(nullptr, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 3: rc = run_kn_sparse<3>(nullptr, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 4: rc = run_kn_sparse<4>(nullptr, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
    case 7: rc = run_kn_sparse<7>(nullptr, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s); break;
  }
  SG_LAUNCH_CHECK("sg_conv2d_fwd_perimage");
  return rc;
}
extern "C" int sg_conv2d_wgrad_perimage(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                                        const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gwimg, void* ws,
                                        size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad_perimage") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_wgrad_perimage")) return -1;
  SG_ARG_CHECK(gy && x1 && gwimg && ws, "sg_conv2d_wgrad_perimage: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad_perimage: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 2), "sg_conv2d_wgrad_perimage: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, gwimg};
  if (int rc = run_nk_ks(d->KS, gy, d->Cout, d->Cout, g, d->N, nullptr, ws, ws_bytes, 0.0, s, &sp)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad_perimage");
  return 0;
}

extern "C" int sg_convT2d_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                              void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_convT2d_fwd")) return -1;
  SG_ARG_CHECK(x && w && y && ws, "sg_convT2d_fwd: null pointer");
  const int Cin = d->C1, R = d->KS * d->KS;
  SG_ARG_CHECK(d->C2 == 0 && d->upsample == 1 && !d->pad_reflect, "sg_convT2d_fwd: unsupported desc");
  SG_ARG_CHECK(ws_bytes >= (size_t)d->Cout * Cin * R * sizeof(float) + ktab_bytes(Cin * R),
               "sg_convT2d_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x, nullptr, Cin, 0, d->H, d->W, 1, d->OH, d->OW, d->stride, d->pad, 0);
  const double flops = 2.0 * d->Cout * Cin * R * (double)d->N * d->H * d->W;
  if (d->stride == 2 && d->KS >= 3) {
    SG_ARG_CHECK(ws_bytes >= parity_ws(d->Cout, Cin, R), "sg_convT2d_fwd: workspace too small");
    int rc = run_kn_parity_ks(d->KS, w, Cin, d->Cout, 0, d->Cout, g, d->N, bias, y, d->Cout, SG_ACT_NONE, 0.f, flops, ws,
                              ws_bytes, s);
    SG_LAUNCH_CHECK("sg_convT2d_fwd");
    return rc;
  }
  float* wt = reinterpret_cast<float*>(ws);      // [Cout][Cin][R]
  const size_t nw = (size_t)d->Cout * Cin * R;
  hipLaunchKernelGGL(permute_w_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, w, wt, Cin, d->Cout, R);
  int rc = run_kn_ks<1>(d->KS, wt, d->Cout, Cin * R, g, d->N, bias, y, d->Cout, SG_ACT_NONE, 0.f, flops, wt + nw,
                        ws_bytes - nw * sizeof(float), s);
  SG_LAUNCH_CHECK("sg_convT2d_fwd");
  return rc;
}

// gx[n,ci,ih,iw] = sum_{co,kh,kw} w[ci,co,kh,kw] gy[n,co,ih*s-p+kh,iw*s-p+kw]   (a plain strided conv over gy)
extern "C" int sg_convT2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                                sgStream stream) {
  if (check_desc(d, "sg_convT2d_dgrad")) return -1;
  SG_ARG_CHECK(gy && w && gx && ws, "sg_convT2d_dgrad: null pointer");
  SG_ARG_CHECK(ws_bytes >= ktab_bytes(d->Cout * d->KS * d->KS), "sg_convT2d_dgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int R = d->KS * d->KS;
  Gather g = make_gather(gy, nullptr, d->Cout, 0, d->OH, d->OW, 1, d->H, d->W, d->stride, d->pad, 0);
  const double flops = 2.0 * d->Cout * d->C1 * R * (double)d->N * d->H * d->W;
  int rc = run_kn_ks<0>(d->KS, w, d->C1, d->Cout * R, g, d->N, nullptr, gx, d->C1, SG_ACT_NONE, 0.f, flops, ws, ws_bytes, s);
  SG_LAUNCH_CHECK("sg_convT2d_dgrad");
  return rc;
}

// gw[ci,co,kh,kw] = sum_{n,ih,iw} x[n,ci,ih,iw] gy[n,co,ih*s-p+kh,iw*s-p+kw]
extern "C" int sg_convT2d_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, float* gb, void* ws,
                                size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_convT2d_wgrad")) return -1;
  SG_ARG_CHECK(gy && x && gw, "sg_convT2d_wgrad: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(gy, nullptr, d->Cout, 0, d->OH, d->OW, 1, d->H, d->W, d->stride, d->pad, 0);
  if (int rc = run_nk_ks(d->KS, x, d->C1, d->C1, g, d->N, gw, ws, ws ? ws_bytes : 0,
                         2.0 * d->Cout * (double)d->C1 * d->KS * d->KS * d->N * d->H * d->W, s))
    return rc;
  SG_LAUNCH_CHECK("sg_convT2d_wgrad");
  if (gb) return sg_channel_sum(gy, gb, d->N, d->Cout, d->OH * d->OW, ws, ws ? ws_bytes : 0, stream);
  return 0;
}

// ================================================================================================
// Winograd F(2x2, 3x3) for ReflectionPad2d(1) + 3x3 stride-1 convs with >= 128 channels on both sides (the ResnetBlock
// convs, layers.py:251-270): forward and weight gradient as 16 batched dense GEMMs over the transformed operands,
// 2.25x fewer MACs than the direct form.  All operands are laid out k-contiguous, so both GEMM loaders are the
// mask-free float4 ones.  (The data gradient stays on the direct kernel: its folded form has no padded-grid waste.)
// ================================================================================================
namespace {

__device__ __forceinline__ int wino_reflect(int i, int L) { i = i < 0 ? -i : i; return i >= L ? 2 * L - 2 - i : i; }

// V = B^T d B of the 4x4 input patch of tile p = (n, ti, tj) on a TH x TW tile grid; the patch starts at (2ti+off, 2tj+off)
// and is reflected (zero_pad == 0) or zero-extended (zero_pad == 1) outside the plane.  PFAST selects the layout:
// V[xi][c][p] (1) or V[xi][p][c] (0); rows p in [N*TH*TW, Pstride) are zero padding for the 128-wide GEMM tiles.
// H x W is the LOGICAL plane (the stored plane is (H >> ush) x (W >> ush): nearest x2 upsampling folded into the read).
template <int PFAST>
__global__ void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W, int TH, int TW,
                                  int off, int zero_pad, size_t Pstride, int ush) {
  const size_t P = (size_t)N * TH * TW, total = Pstride * C;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t p = PFAST ? idx % Pstride : idx / C;
  const int c = (int)(PFAST ? idx / Pstride : idx % C);
  float d[4][4];
  if (p < P) {
    const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
    const int SW = W >> ush;
    const float* xp = x + ((size_t)n * C + c) * (H >> ush) * SW;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ih0 = 2 * ti + off + a;
      const bool rok = !zero_pad || (unsigned)ih0 < (unsigned)H;
      const int ih = zero_pad ? (rok ? ih0 : 0) : wino_reflect(ih0, H);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int iw0 = 2 * tj + off + b;
        const bool ok = rok && (!zero_pad || (unsigned)iw0 < (unsigned)W);
        const int iw = zero_pad ? (ok ? iw0 : 0) : wino_reflect(iw0, W);
        const float v = xp[(ih >> ush) * SW + (iw >> ush)];
        d[a][b] = ok ? v : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) d[a][b] = 0.f;
  }
  float t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v0 = t[i][0] - t[i][2], v1 = t[i][1] + t[i][2], v2 = t[i][2] - t[i][1], v3 = t[i][1] - t[i][3];
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t xi = (size_t)(i * 4 + j);
      V[PFAST ? (xi * C + c) * Pstride + p : (xi * Pstride + p) * C + c] = v[j];
    }
  }
}

// Same transform (layout V[xi][p][c]) for small planes (H*W <= 256, the 8x8 / 16x16 maps of the residual trunk): the kernel
// above lets neighbouring lanes read neighbouring CHANNELS, i.e. addresses H*W floats apart -- 16 uncoalesced loads per
// lane (measured 34 us for an 8 MB input, ~1.2 TB/s).  Here one workgroup stages the planes of 64 channels of one image
// (one contiguous block of memory) in LDS with coalesced float4 reads and the lanes then run along the channel for both the
// LDS reads (pitch H*W + 1: conflict-free) and the global writes (256 contiguous bytes per wave).
__global__ void __launch_bounds__(256) wino_input_small_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C,
                                                              int H, int W, int TH, int TW, int off, int zero_pad, size_t Pstride) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int HW = H * W, pitch = HW + 1;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  const float* src = x + ((size_t)n * C + c0) * HW;
  for (int i = tid * 4; i < 64 * HW; i += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    const int ch = i / HW, px = i - ch * HW;
    float* d = pl + ch * pitch + px;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int c = tid & 63, g = tid >> 6;
  const float* pc = pl + c * pitch;
  const size_t P = (size_t)N * TH * TW;
  for (int t = g; t < TH * TW; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    float d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int ih0 = 2 * ti + off + a;
      const bool rok = !zero_pad || (unsigned)ih0 < (unsigned)H;
      const int ih = zero_pad ? (rok ? ih0 : 0) : wino_reflect(ih0, H);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int iw0 = 2 * tj + off + b;
        const bool ok = rok && (!zero_pad || (unsigned)iw0 < (unsigned)W);
        const int iw = zero_pad ? (ok ? iw0 : 0) : wino_reflect(iw0, W);
        const float v = pc[ih * W + iw];
        d[a][b] = ok ? v : 0.f;
      }
    }
    float tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tt[0][j] = d[0][j] - d[2][j]; tt[1][j] = d[1][j] + d[2][j]; tt[2][j] = d[2][j] - d[1][j]; tt[3][j] = d[1][j] - d[3][j];
    }
    const size_t p = (size_t)n * TH * TW + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v[4] = {tt[i][0] - tt[i][2], tt[i][1] + tt[i][2], tt[i][2] - tt[i][1], tt[i][1] - tt[i][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) V[((size_t)(i * 4 + j) * Pstride + p) * C + c0 + c] = v[j];
    }
  }
  if (n == 0)                                   // rows [P, Pstride): zero padding for the 128-wide GEMM tiles
    for (size_t p = P + g; p < Pstride; p += 4)
      for (int xi = 0; xi < 16; ++xi) V[((size_t)xi * Pstride + p) * C + c0 + c] = 0.f;
}

// U[xi][r][c] = (G g G^T)[xi] with g = w[r][c] (flip == 0) or the 180-degree rotated w[c][r] (flip == 1: the data gradient
// is a correlation with the flipped, transposed filter)
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int R, int Cc, int flip) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t RC = (size_t)R * Cc;
  if (i >= RC) return;
  const int c = (int)(i % Cc);
  const size_t r = i / Cc;
  // flip: 0 = w[r][c]; 1 = transposed and rotated by 180 degrees (w[c][r], taps reversed); 2 = transposed only (the adjoint
  // of the forward transform: same taps, channel roles swapped)
  const float* g = w + (flip ? ((size_t)c * R + r) * 9 : i * 9);
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const bool rot = flip == 1;
    const float g0 = rot ? g[8 - j] : g[j], g1 = rot ? g[5 - j] : g[3 + j], g2 = rot ? g[2 - j] : g[6 + j];
    t[0][j] = g0; t[1][j] = 0.5f * (g0 + g1 + g2); t[2][j] = 0.5f * (g0 - g1 + g2); t[3][j] = g2;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
    for (int b = 0; b < 4; ++b) U[(size_t)(a * 4 + b) * RC + i] = u[b];
  }
}

// LDS-staged form for R, Cc multiples of 32: a workgroup owns a 32x32 block of (r, c), reads its 32 source rows of 32*9
// contiguous floats coalesced (for the transposed modes those rows are w[c][r0..r0+31]: the per-thread form above reads them
// with a lane stride of R*36 bytes) and writes U with c contiguous.  Same arithmetic, same results.
constexpr int WW_PITCH = 32 * 9 + 1;
__global__ void __launch_bounds__(256) wino_weight_lds_kernel(const float* __restrict__ w, float* __restrict__ U, int R, int Cc,
                                                              int flip) {
  __shared__ float S[32 * WW_PITCH];
  const int cb = blockIdx.x % (Cc / 32), rb = blockIdx.x / (Cc / 32);
  const int r0 = rb * 32, c0 = cb * 32, t = threadIdx.x;
  // source row j: flip == 0 -> (r0 + j, c0 ..), else (c0 + j, r0 ..) of the [Cc][R] tensor
  // (rows start at multiples of 288 floats: 16-byte aligned whenever w is)
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int idx = i * 256 + t, j = idx / 72, o = (idx - j * 72) * 4;
    const size_t src = flip ? ((size_t)(c0 + j) * R + r0) * 9 + o : ((size_t)(r0 + j) * Cc + c0) * 9 + o;
    const float4 v = *reinterpret_cast<const float4*>(w + src);
    float* d = S + j * WW_PITCH + o;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const size_t RC = (size_t)R * Cc;
  const int cl = t & 31;
  const bool rot = flip == 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = (t >> 5) + 8 * q;
    const float* g = flip ? S + cl * WW_PITCH + rl * 9 : S + rl * WW_PITCH + cl * 9;
    float tt[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float g0 = rot ? g[8 - j] : g[j], g1 = rot ? g[5 - j] : g[3 + j], g2 = rot ? g[2 - j] : g[6 + j];
      tt[0][j] = g0; tt[1][j] = 0.5f * (g0 + g1 + g2); tt[2][j] = 0.5f * (g0 - g1 + g2); tt[3][j] = g2;
    }
    const size_t i = (size_t)(r0 + rl) * Cc + c0 + cl;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u[4] = {tt[a][0], 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]), 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]), tt[a][2]};
#pragma unroll
      for (int b = 0; b < 4; ++b) U[(size_t)(a * 4 + b) * RC + i] = u[b];
    }
  }
}

// one entry for the three call sites (SG_WINO_WT=0 keeps the per-thread kernel)
inline void wino_weight(const float* w, float* U, int R, int Cc, int flip, hipStream_t s) {
  static int lds = -1;
  if (lds < 0) { const char* e = getenv("SG_WINO_WT"); lds = e ? atoi(e) : 1; }
  if (lds && R % 32 == 0 && Cc % 32 == 0 && aligned16(w))
    hipLaunchKernelGGL(wino_weight_lds_kernel, dim3((R / 32) * (Cc / 32)), dim3(256), 0, s, w, U, R, Cc, flip);
  else
    hipLaunchKernelGGL(wino_weight_kernel, dim3(sg_cdiv((size_t)R * Cc, 256)), dim3(256), 0, s, w, U, R, Cc, flip);
}

// y[n][m][2ti+a][2tj+b] = act((A^T Mx A)[a][b] + bias[m]),  Mx[m][xi*Pstride + p]
__global__ void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias, float* __restrict__ y, int N,
                                   int M, int H, int W, size_t Pstride, int act, float slope) {
  const int TH = H / 2, TW = W / 2;
  const size_t P = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * M) return;
  const size_t p = idx % P;
  const int m = (int)(idx / P);
  const float* src = Mx + (size_t)m * 16 * Pstride + p;
  float q[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i >> 2][i & 3] = src[(size_t)i * Pstride];
  float s[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[0][j] = q[0][j] + q[1][j] + q[2][j]; s[1][j] = q[1][j] - q[2][j] - q[3][j]; }
  const float b = bias ? bias[m] : 0.f;
  const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
  float* o = y + (((size_t)n * M + m) * H + 2 * ti) * W + 2 * tj;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float y0 = s[a][0] + s[a][1] + s[a][2] + b, y1 = s[a][1] - s[a][2] - s[a][3] + b;
    *reinterpret_cast<float2*>(o + a * W) = make_float2(sg_apply_act(y0, act, slope), sg_apply_act(y1, act, slope));
  }
}

// Yt[xi][m][p] = (A dY A^T)[xi] of the 2x2 gradient tile p
__global__ void wino_gy_kernel(const float* __restrict__ gy, float* __restrict__ Yt, int N, int M, int H, int W) {
  const int TH = H / 2, TW = W / 2;
  const size_t P = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * M) return;
  const size_t p = idx % P;
  const int m = (int)(idx / P);
  const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
  const float* g = gy + (((size_t)n * M + m) * H + 2 * ti) * W + 2 * tj;
  const float2 r0 = *reinterpret_cast<const float2*>(g), r1 = *reinterpret_cast<const float2*>(g + W);
  const float t[4][2] = {{r0.x, r0.y}, {r0.x + r1.x, r0.y + r1.y}, {r0.x - r1.x, r0.y - r1.y}, {-r1.x, -r1.y}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v[4] = {t[i][0], t[i][0] + t[i][1], t[i][0] - t[i][1], -t[i][1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) Yt[((size_t)(i * 4 + j) * M + m) * P + p] = v[j];
  }
}

// ---- adjoint ("transposed") Winograd for the data gradient of the reflection-padded convs -----------------------------------
// y_tile = A^T [sum_c U[c] (.) (B^T d B)] A  ==>  g_d = B [sum_co U[co][ci] (.) (A gy_tile A^T)] B^T, overlap-added into the
// padded gradient grid and folded by the reflection.  The GEMM runs over the N*(H/2)*(W/2) OUTPUT tiles (512 at 8x8, batch 32)
// instead of the 800 (-> 896 padded) tiles of the (H+2)x(W+2) grid the correlation form needs: 1.75x fewer MACs.
//
// Ytp[xi][p][m] = (A gy A^T)[xi] of the 2x2 gradient tile p, small planes, LDS-staged (lanes along the channel m)
__global__ void __launch_bounds__(256) wino_gy_small_kernel(const float* __restrict__ gy, float* __restrict__ Ytp, int N, int M, int H,
                                                           int W) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int HW = H * W, pitch = HW + 1, TH = H / 2, TW = W / 2;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  const float* src = gy + ((size_t)n * M + c0) * HW;
  for (int i = tid * 4; i < 64 * HW; i += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    const int ch = i / HW, px = i - ch * HW;
    float* d = pl + ch * pitch + px;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int c = tid & 63, g = tid >> 6;
  const float* pc = pl + c * pitch;
  const size_t P = (size_t)N * TH * TW;
  for (int t = g; t < TH * TW; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    const float a0 = pc[(2 * ti) * W + 2 * tj], a1 = pc[(2 * ti) * W + 2 * tj + 1];
    const float b0 = pc[(2 * ti + 1) * W + 2 * tj], b1 = pc[(2 * ti + 1) * W + 2 * tj + 1];
    const float tt[4][2] = {{a0, a1}, {a0 + b0, a1 + b1}, {a0 - b0, a1 - b1}, {-b0, -b1}};
    const size_t p = (size_t)n * TH * TW + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v[4] = {tt[i][0], tt[i][0] + tt[i][1], tt[i][0] - tt[i][1], -tt[i][1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) Ytp[((size_t)(i * 4 + j) * P + p) * M + c0 + c] = v[j];
    }
  }
}

// gx[n][c][H][W] from G[p][xi][c] (row length 16*C): per channel, the 4x4 patches B G B^T of the tiles are overlap-added in
// tile order into a padded (H+2)x(W+2) plane (private to the thread: no synchronisation, fixed order), the reflection is
// folded, and the 64 channel planes of the workgroup -- one contiguous block of memory -- are written with coalesced float4s.
__global__ void __launch_bounds__(64) wino_patch_fold_kernel(const float* __restrict__ G, float* __restrict__ gx, int N, int C, int H,
                                                            int W) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W, PW = W + 2, PP = (H + 2) * PW, TH = H / 2, TW = W / 2;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, c = threadIdx.x;
  float* acc = lds + c * (PP + 1);
  float* stage = lds + 64 * (PP + 1);              // [64][HW + 1]
  for (int i = 0; i < PP; ++i) acc[i] = 0.f;
  const size_t ld = (size_t)16 * C;
  for (int t = 0; t < TH * TW; ++t) {
    const int ti = t / TW, tj = t - ti * TW;
    const float* src = G + ((size_t)n * TH * TW + t) * ld + c0 + c;
    float q[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i >> 2][i & 3] = src[(size_t)i * C];
    float r[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      r[0][v] = q[0][v]; r[1][v] = q[1][v] - q[2][v] + q[3][v]; r[2][v] = -q[0][v] + q[1][v] + q[2][v]; r[3][v] = -q[3][v];
    }
    float* dst = acc + (2 * ti) * PW + 2 * tj;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[a * PW + 0] += r[a][0];
      dst[a * PW + 1] += r[a][1] - r[a][2] + r[a][3];
      dst[a * PW + 2] += -r[a][0] + r[a][1] + r[a][2];
      dst[a * PW + 3] += -r[a][3];
    }
  }
  float* so = stage + c * (HW + 1);
  for (int a = 0; a < H; ++a) {
    for (int b = 0; b < W; ++b) {
      float v = acc[(a + 1) * PW + b + 1];
      const int ra = a == 1 ? 0 : -1, rb = a == H - 2 ? H + 1 : -1;        // padded rows folded onto row a
      const int ca = b == 1 ? 0 : -1, cb = b == W - 2 ? W + 1 : -1;        // padded columns folded onto column b
      if (ra >= 0) v += acc[ra * PW + b + 1];
      if (rb >= 0) v += acc[rb * PW + b + 1];
      if (ca >= 0) v += acc[(a + 1) * PW + ca];
      if (cb >= 0) v += acc[(a + 1) * PW + cb];
      if (ra >= 0 && ca >= 0) v += acc[ra * PW + ca];
      if (ra >= 0 && cb >= 0) v += acc[ra * PW + cb];
      if (rb >= 0 && ca >= 0) v += acc[rb * PW + ca];
      if (rb >= 0 && cb >= 0) v += acc[rb * PW + cb];
      so[a * W + b] = v;
    }
  }
  __syncthreads();
  float* dstg = gx + ((size_t)n * C + c0) * HW;
  for (int i = c * 4; i < 64 * HW; i += 256) {
    const int ch = i / HW, px = i - ch * HW;
    const float* sp = stage + ch * (HW + 1) + px;
    *reinterpret_cast<float4*>(dstg + i) = make_float4(sp[0], sp[1], sp[2], sp[3]);
  }
}

// gw[m][c][3][3] = G^T T G,  T[m][xi*C + c]
__global__ void wino_wgrad_output_kernel(const float* __restrict__ T, float* __restrict__ gw, int M, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * C) return;
  const int c = (int)(idx % C);
  const size_t m = idx / C;
  const float* src = T + m * 16 * C + c;
  float q[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i >> 2][i & 3] = src[(size_t)i * C];
  float s[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = q[0][j] + 0.5f * (q[1][j] + q[2][j]);
    s[1][j] = 0.5f * (q[1][j] - q[2][j]);
    s[2][j] = 0.5f * (q[1][j] + q[2][j]) + q[3][j];
  }
  float* o = gw + idx * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[a * 3 + 0] = s[a][0] + 0.5f * (s[a][1] + s[a][2]);
    o[a * 3 + 1] = 0.5f * (s[a][1] - s[a][2]);
    o[a * 3 + 2] = 0.5f * (s[a][1] + s[a][2]) + s[a][3];
  }
}

// V[xi][p][c] of x (logical H x W plane, ush = folded upsample shift): LDS-staged kernel for small planes, general otherwise
void wino_input_pc(const float* x, float* V, int N, int C, int H, int W, int TH, int TW, int off, int zero_pad, size_t Pstride,
                   int ush, hipStream_t s) {
  const int HW = H * W;
  SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0);
  if (ush == 0 && HW <= 256 && HW % 4 == 0 && C % 64 == 0 && aligned16(x)) {
    const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_input_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds);
    hipLaunchKernelGGL(wino_input_small_kernel, dim3(C / 64, N), dim3(256), lds, s, x, V, N, C, H, W, TH, TW, off, zero_pad, Pstride);
    return;
  }
  hipLaunchKernelGGL(wino_input_kernel<0>, dim3(sg_cdiv(Pstride * C, 256)), dim3(256), 0, s, x, V, N, C, H, W, TH, TW, off, zero_pad,
                     Pstride, ush);
}

// Winograd applies to 3x3 / stride 1 / pad 1 convs (reflection or zero padding, optionally behind a folded nearest x2
// upsample) with >= 128 channels on both sides whose channel counts and tile count fill whole 128-wide GEMM tiles (the
// ResnetBlock and VGG19 convs).  Below 128 channels the elementwise transforms (16 x the activation bytes) cost more than the
// GEMM saves; 64-wide tiles (mask_net's 192 channels) were measured SLOWER than the direct kernel there (60 TFLOP/s in the
// 64x64 dense GEMM + the transforms vs 120 TFLOP/s direct) and are not used.
int wino_tile(const sgConvDesc* d) {
  if (!d || d->pad != 1 || d->KS != 3 || d->stride != 1 || d->C2 != 0) return 0;
  if (d->upsample != 1 && (d->upsample != 2 || d->pad_reflect)) return 0;
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  if (LH < 4 || LW < 4 || (LH & 1) || (LW & 1) || d->OH != LH || d->OW != LW) return 0;
  if (d->C1 < 128 || d->Cout < 128) return 0;
  const long P = (long)d->N * (LH / 2) * (LW / 2);
  // 32-bit element offsets into the transformed operands (the reflect dgrad works on the (LH+2) x (LW+2) grid, 128-padded)
  const double Pmax = (double)d->N * (LH / 2 + 1) * (LW / 2 + 1) + 128.0;
  if (!(16.0 * Pmax * (d->C1 > d->Cout ? d->C1 : d->Cout) < 2147483647.0 && 16.0 * d->C1 * d->Cout < 2147483647.0)) return 0;
  if (d->C1 % 128 == 0 && d->Cout % 128 == 0 && P % 128 == 0) return 128;
  return 0;
}
bool wino_ok(const sgConvDesc* d) { return wino_tile(d) != 0; }

// C[m][b*cols + j] = sum_k A[b][m][k] * B[b*cols + j][k]   (16 batches, everything a multiple of the tile)
void wino_bgemm(const float* A, const float* B, float* Cout, int M, int cols, int K, double flops, hipStream_t s) {
  EpRowMajor ep{Cout, nullptr, M, 16 * cols, 16 * cols, SG_ACT_NONE, 0.f, 0};
  t_batch = BatchInfo{}; t_batch.cols_per_batch = cols; t_batch.nbatch = 16; t_batch.a_stride = M * K; t_batch.batch_major = 1;
  {
    SgProfScope prof(SG_K_WINO_GEMM_128, s, flops, 0);
    static int wt = -1;
    if (wt < 0) { const char* e = getenv("SG_WINO_TILE"); wt = e ? atoi(e) : 0; }
    if (wt == 1)
      launch_cfg<CfgD128x64>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<64, true, false>{B, K, 16 * cols}, ep, M,
                             16 * cols, K, 1, s);
    else if (wt == 2)
      launch_cfg<Cfg128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols}, ep, M,
                         16 * cols, K, 1, s);
    else if (wt == 3)       // the plain loop (before the software-pipelined form)
      launch_cfg<CfgD128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols}, ep, M,
                          16 * cols, K, 1, s);
    else if (wt == 6)       // pipelined, 16-deep k-tiles (40 KB of LDS: three workgroups per CU): measured 0.763 vs 0.775 of peak
      launch_cfg<TileCfg<128, 128, 2, 1, 1>>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols},
                                             EpRowMajorPlain{Cout, 16 * cols}, M, 16 * cols, K, 1, s);
    else if (wt == 5)       // the plain loop with the unconditional epilogue
      launch_cfg<CfgD128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols},
                          EpRowMajorPlain{Cout, 16 * cols}, M, 16 * cols, K, 1, s);
    else if (wt == 4)       // pipelined loop, general epilogue
      launch_cfg<CfgDP128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols}, ep, M,
                           16 * cols, K, 1, s);
    else
      launch_cfg<CfgDP128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, 16 * cols},
                           EpRowMajorPlain{Cout, 16 * cols}, M, 16 * cols, K, 1, s);
  }
  t_batch = BatchInfo{};
}

}  // namespace

extern "C" int sg_conv2d_wino_supported(const sgConvDesc* d) { return wino_ok(d) ? 1 : 0; }

static size_t wino_dgrad_tiles(const sgConvDesc* d) {      // reflect: tiles of the (H+2) x (W+2) padded gradient grid
  const size_t T = (size_t)wino_tile(d);
  const size_t LH = (size_t)d->H * d->upsample, LW = (size_t)d->W * d->upsample;
  const size_t P = d->pad_reflect ? (size_t)d->N * (LH / 2 + 1) * (LW / 2 + 1) : (size_t)d->N * (LH / 2) * (LW / 2);
  return (P + T - 1) / T * T;
}
extern "C" size_t sg_conv2d_wino_ws_bytes(const sgConvDesc* d) {
  if (!wino_ok(d)) return 0;
  const size_t LH = (size_t)d->H * d->upsample, LW = (size_t)d->W * d->upsample;
  const size_t P = (size_t)d->N * (LH / 2) * (LW / 2), M = d->Cout, C = d->C1, Pd = wino_dgrad_tiles(d);
  const size_t a = 16 * (M * C + P * C + M * P);
  const size_t b = 16 * (M * C + Pd * M + C * Pd) + (size_t)d->N * C * (LH + 2) * (LW + 2);
  return (a > b ? a : b) * sizeof(float) + 1024;
}

// gx [N, C1, H, W].  Reflection padding: Winograd over the (H+2) x (W+2) gradient of the reflect-padded input (correlation of
// the zero-extended gy with the rotated filter), then the reflection fold (sg_pad_upsample_bwd); 1.44x fewer MACs than the
// direct folded form.  Zero padding: the same correlation straight on the H x W grid (2.25x fewer MACs); behind a folded x2
// upsample the result is on the upsampled grid and is summed back 2x2.
extern "C" int sg_conv2d_wino_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                                    sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_dgrad: unsupported desc");
  SG_ARG_CHECK(gy && w && gx && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_dgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->C1, K = d->Cout;                 // rows = input channels, reduction over output channels
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  const int refl = d->pad_reflect;
  static int adj = -1;
  if (adj < 0) { const char* e = getenv("SG_WINO_ADJOINT"); adj = e ? atoi(e) : 1; }
  if (adj && refl && d->upsample == 1 && d->H * d->W <= 256 && (d->H * d->W) % 4 == 0 && M % 64 == 0 && K % 64 == 0 &&
      aligned16(gy) && aligned16(gx)) {
    // adjoint Winograd over the output tiles (see wino_gy_small_kernel / wino_patch_fold_kernel)
    const int HW = d->H * d->W;
    const size_t P = (size_t)d->N * (d->H / 2) * (d->W / 2);
    float* UT = reinterpret_cast<float*>(ws);       // [16][C1][Cout]
    float* Ytp = UT + 16 * (size_t)M * K;           // [16][P][Cout]
    float* G = Ytp + 16 * P * K;                    // [P][16][C1]
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0);
      wino_weight(w, UT, M, K, 2, s); }
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0);
      const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gy_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wino_gy_small_kernel, dim3(K / 64, d->N), dim3(256), lds, s, gy, Ytp, d->N, K, d->H, d->W); }
    // G[p][xi*C1 + ci] = sum_co Ytp[xi][p][co] * UT[xi][ci][co]
    wino_bgemm(Ytp, UT, G, (int)P, M, K, 2.0 * M * (double)K * 16.0 * P, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0);
      const size_t lds = (size_t)64 * ((d->H + 2) * (d->W + 2) + 1 + HW + 1) * sizeof(float);
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_patch_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wino_patch_fold_kernel, dim3(M / 64, d->N), dim3(64), lds, s, (const float*)G, gx, d->N, M, d->H, d->W); }
    SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad");
    return 0;
  }
  const int TH = LH / 2 + (refl ? 1 : 0), TW = LW / 2 + (refl ? 1 : 0);
  const size_t Pd = wino_dgrad_tiles(d);
  float* U = reinterpret_cast<float*>(ws);          // [16][C1][Cout]
  float* V = U + 16 * (size_t)M * K;                // [16][Pd][Cout]
  float* Mx = V + 16 * Pd * K;                      // [C1][16][Pd]
  float* gpad = Mx + 16 * Pd * M;                   // [N][C1][LH+2][LW+2] (reflect) / [N][C1][LH][LW] (upsample)
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); wino_weight(w, U, M, K, 1, s); }
  wino_input_pc(gy, V, d->N, K, LH, LW, TH, TW, refl ? -2 : -1, 1, Pd, 0, s);
  wino_bgemm(U, V, Mx, M, (int)Pd, K, 2.0 * M * (double)K * 16.0 * ((double)d->N * TH * TW), s);   // flops of the real tiles
  const bool direct = !refl && d->upsample == 1;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); hipLaunchKernelGGL(wino_output_kernel, dim3(sg_cdiv((size_t)d->N * TH * TW * M, 256)), dim3(256), 0, s, (const float*)Mx,
                     (const float*)nullptr, direct ? gx : gpad, d->N, M, 2 * TH, 2 * TW, Pd, SG_ACT_NONE, 0.f); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad");
  if (direct) return 0;
  return sg_pad_upsample_bwd(gpad, gx, d->N * M, d->H, d->W, refl ? 1 : 0, d->upsample, stream);
}

extern "C" int sg_conv2d_wino_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                                  float slope, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_fwd: unsupported desc");
  SG_ARG_CHECK(x && w && y && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  const size_t P = (size_t)d->N * (LH / 2) * (LW / 2);
  float* U = reinterpret_cast<float*>(ws);
  float* V = U + 16 * (size_t)M * C;
  float* Mx = V + 16 * P * C;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); wino_weight(w, U, M, C, 0, s); }
  wino_input_pc(x, V, d->N, C, LH, LW, LH / 2, LW / 2, -1, d->pad_reflect ? 0 : 1, P, d->upsample == 2 ? 1 : 0, s);
  wino_bgemm(U, V, Mx, M, (int)P, C, 2.0 * M * (double)C * 16.0 * P, s);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); hipLaunchKernelGGL(wino_output_kernel, dim3(sg_cdiv(P * M, 256)), dim3(256), 0, s, (const float*)Mx, bias, y, d->N, M, LH, LW, P,
                     act, slope); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_fwd");
  return 0;
}

extern "C" int sg_conv2d_wino_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                                    sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_wgrad: unsupported desc");
  SG_ARG_CHECK(gy && x && gw && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  const size_t P = (size_t)d->N * (LH / 2) * (LW / 2);
  float* T = reinterpret_cast<float*>(ws);          // [M][16][C]
  float* Vp = T + 16 * (size_t)M * C;               // [16][C][P]
  float* Yt = Vp + 16 * P * C;                      // [16][M][P]
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); hipLaunchKernelGGL(wino_input_kernel<1>, dim3(sg_cdiv(P * C, 256)), dim3(256), 0, s, x, Vp, d->N, C, LH, LW, LH / 2, LW / 2, -1,
                     d->pad_reflect ? 0 : 1, P, d->upsample == 2 ? 1 : 0); }
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); hipLaunchKernelGGL(wino_gy_kernel, dim3(sg_cdiv(P * M, 256)), dim3(256), 0, s, gy, Yt, d->N, M, LH, LW); }
  wino_bgemm(Yt, Vp, T, M, C, (int)P, 2.0 * M * (double)C * 16.0 * P, s);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 0); hipLaunchKernelGGL(wino_wgrad_output_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, (const float*)T, gw, M, C); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_wgrad");
  return 0;
}

// ---- Interpolate(x2 nearest) + conv3x3(pad 1) as a sub-pixel transposed conv (see the header) ---------------------------
namespace {
// taps of the 3x3 filter that land on transposed-conv tap k (per axis): [first, last]
__device__ __forceinline__ void upconv_range(int k, int& lo, int& hi) {
  lo = k == 0 ? 2 : (k == 1 ? 1 : 0);
  hi = k == 0 ? 2 : (k == 1 ? 2 : (k == 2 ? 1 : 0));
}
__global__ void upconv3_fold_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // over wt [Cin][Cout][4][4]
  if (idx >= (size_t)Cin * Cout * 16) return;
  const int kw = idx & 3, kh = (idx >> 2) & 3;
  const size_t cc = idx >> 4;
  const int co = (int)(cc % Cout), ci = (int)(cc / Cout);
  const float* g = w + ((size_t)co * Cin + ci) * 9;
  int i0, i1, j0, j1;
  upconv_range(kh, i0, i1);
  upconv_range(kw, j0, j1);
  float v = 0.f;
  for (int i = i0; i <= i1; ++i)
    for (int j = j0; j <= j1; ++j) v += g[i * 3 + j];
  wt[idx] = v;
}
__global__ void upconv3_unfold_kernel(const float* __restrict__ gwt, float* __restrict__ gw, int Cout, int Cin) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;        // over gw [Cout][Cin][3][3]
  if (idx >= (size_t)Cout * Cin * 9) return;
  const int t = (int)(idx % 9), i = t / 3, j = t - i * 3;
  const size_t cc = idx / 9;
  const int ci = (int)(cc % Cin), co = (int)(cc / Cin);
  const float* g = gwt + ((size_t)ci * Cout + co) * 16;
  // tap i of the 3x3 filter contributes to transposed-conv taps {2,3} (i=0), {1,2} (i=1), {0,1} (i=2)
  const int kh0 = 2 - i, kw0 = 2 - j;
  gw[idx] = (g[kh0 * 4 + kw0] + g[kh0 * 4 + kw0 + 1]) + (g[(kh0 + 1) * 4 + kw0] + g[(kh0 + 1) * 4 + kw0 + 1]);
}
}  // namespace

extern "C" int sg_upconv3_fold_weights(const float* w, float* wt, int Cout, int Cin, sgStream stream) {
  SG_ARG_CHECK(w && wt && Cout > 0 && Cin > 0, "sg_upconv3_fold_weights: bad arguments");
  const size_t n = (size_t)Cin * Cout * 16;
  hipLaunchKernelGGL(upconv3_fold_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin);
  SG_LAUNCH_CHECK("sg_upconv3_fold_weights");
  return 0;
}
extern "C" int sg_upconv3_unfold_wgrad(const float* gwt, float* gw, int Cout, int Cin, sgStream stream) {
  SG_ARG_CHECK(gwt && gw && Cout > 0 && Cin > 0, "sg_upconv3_unfold_wgrad: bad arguments");
  const size_t n = (size_t)Cout * Cin * 9;
  hipLaunchKernelGGL(upconv3_unfold_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gwt, gw, Cout, Cin);
  SG_LAUNCH_CHECK("sg_upconv3_unfold_wgrad");
  return 0;
}

extern "C" size_t sg_plan_cache_bytes(void) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  return g_tab_bytes;
}
extern "C" int sg_plan_cache_clear(void) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  for (auto& kv : g_tabs) { hipEventSynchronize(kv.second.ready); hipEventDestroy(kv.second.ready); hipFree(kv.second.dev); }
  g_tabs.clear();
  g_tab_bytes = 0;
  return 0;
}

// ---- dense layers --------------------------------------------------------------------------------
namespace {
inline bool dense_sizes_ok(int rows, int in_f, int out_f) {
  return (double)rows * in_f <= SG_MAX_ELEMS && (double)rows * out_f <= SG_MAX_ELEMS && (double)in_f * out_f <= SG_MAX_ELEMS;
}
template <class A64, class B64, class A32, class B128>
int run_dense(const A64& a64, const B64& b64, const A32& a32, const B128& b128, const EpRowMajor& ep, int M, int N,
              int K, hipStream_t s) {
  // these GEMMs are a chain of K/16 dependent load -> LDS -> MFMA rounds on a grid that does not even fill the chip
  // (graph-conv MLPs: ~150 workgroups): 32-deep k-tiles halve the number of rounds.  SG_LINEAR_NSUB=1 restores depth 16.
  static int deep = -1;
  if (deep < 0) { const char* e = getenv("SG_LINEAR_NSUB"); deep = e ? atoi(e) : 2; }
  if (deep == 2 && K >= 64) {
    if (M <= 32) return launch_cfg<TileCfg<32, 128, 1, 2>>(a32, b128, ep, M, N, K, 1, s);
    return launch_cfg<TileCfg<64, 64, 2, 2>>(a64, b64, ep, M, N, K, 1, s);
  }
  if (M <= 32) return launch_cfg<Cfg32>(a32, b128, ep, M, N, K, 1, s);
  return launch_cfg<Cfg64>(a64, b64, ep, M, N, K, 1, s);
}
}  // namespace

extern "C" int sg_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in_f, int out_f,
                             int act, float slope, sgStream stream) {
  SG_ARG_CHECK(x && w && y && rows > 0 && in_f > 0 && out_f > 0, "sg_linear_fwd: bad arguments");
  SG_ARG_CHECK(dense_sizes_ok(rows, in_f, out_f), "sg_linear_fwd: operand exceeds %.0f elements", SG_MAX_ELEMS);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{y, b, rows, out_f, out_f, act, slope, 0};
  const bool vec = (in_f % 4 == 0) && aligned16(x) && aligned16(w);
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
  if (vec)
    run_dense(LoadKContig<64, true>{x, in_f, rows}, LoadKContig<64, true>{w, in_f, out_f},
              LoadKContig<32, true>{x, in_f, rows}, LoadKContig<128, true>{w, in_f, out_f}, ep, rows, out_f, in_f, s);
  else
    run_dense(LoadKContig<64, false>{x, in_f, rows}, LoadKContig<64, false>{w, in_f, out_f},
              LoadKContig<32, false>{x, in_f, rows}, LoadKContig<128, false>{w, in_f, out_f}, ep, rows, out_f, in_f, s);
  SG_LAUNCH_CHECK("sg_linear_fwd");
  return 0;
}

extern "C" int sg_linear_bwd_data(const float* gy, const float* w, float* gx, int rows, int in_f, int out_f,
                                  sgStream stream) {
  SG_ARG_CHECK(gy && w && gx && rows > 0 && in_f > 0 && out_f > 0, "sg_linear_bwd_data: bad arguments");
  SG_ARG_CHECK(dense_sizes_ok(rows, in_f, out_f), "sg_linear_bwd_data: operand exceeds %.0f elements", SG_MAX_ELEMS);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{gx, nullptr, rows, in_f, in_f, SG_ACT_NONE, 0.f, 0};
  const bool vec = (out_f % 4 == 0) && aligned16(gy);
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
  if (vec)
    run_dense(LoadKContig<64, true>{gy, out_f, rows}, LoadXContig<64>{w, in_f, in_f}, LoadKContig<32, true>{gy, out_f, rows},
              LoadXContig<128>{w, in_f, in_f}, ep, rows, in_f, out_f, s);
  else
    run_dense(LoadKContig<64, false>{gy, out_f, rows}, LoadXContig<64>{w, in_f, in_f},
              LoadKContig<32, false>{gy, out_f, rows}, LoadXContig<128>{w, in_f, in_f}, ep, rows, in_f, out_f, s);
  SG_LAUNCH_CHECK("sg_linear_bwd_data");
  return 0;
}

extern "C" int sg_linear_bwd_weight(const float* gy, const float* x, float* gw, float* gb, int rows, int in_f,
                                    int out_f, sgStream stream) {
  SG_ARG_CHECK(gy && x && gw && rows > 0 && in_f > 0 && out_f > 0, "sg_linear_bwd_weight: bad arguments");
  SG_ARG_CHECK(dense_sizes_ok(rows, in_f, out_f), "sg_linear_bwd_weight: operand exceeds %.0f elements", SG_MAX_ELEMS);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{gw, nullptr, out_f, in_f, in_f, SG_ACT_NONE, 0.f, 0};
  {
    SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
    run_dense(LoadXContig<64>{gy, out_f, out_f}, LoadXContig<64>{x, in_f, in_f}, LoadXContig<32>{gy, out_f, out_f},
              LoadXContig<128>{x, in_f, in_f}, ep, out_f, in_f, rows, s);
  }
  SG_LAUNCH_CHECK("sg_linear_bwd_weight");
  if (gb) return sg_channel_sum(gy, gb, rows, out_f, 1, nullptr, 0, stream);   // column sums of gy[rows][out_f]
  return 0;
}
