// Core of the implicit-GEMM family (see igemm.hip): operand loaders, epilogues, the kernel template and its launcher.
// Included by the four translation units igemm.hip (C ABI, Winograd, dense layers), igemm_kn0.hip (conv-style gathers),
// igemm_kn1.hip (transposed gathers, parity classes) and igemm_nk.hip (weight gradients): they compile in parallel (the single
// file took 3.5 minutes).  Templates and kernels have internal linkage (one copy per unit); the plain structs and the
// shape-table cache shared between units live in namespace sgk.
#pragma once
#include <type_traits>
#include "common.h"
#include <stdlib.h>
#include <array>
#include <map>
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));


namespace sgk {
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

struct KEntry { unsigned choff; unsigned tapsel; };     // channel-plane offset ; tap row | second-source << 8
// per-image ascending active-channel lists; wimg / gwimg (optional): per-image weights [N][M][L][KS2] in list order instead
// of one shared weight tensor (factored layout convs: the channels are the objects of the image)
struct Sparse { const int* list; const int* cnt; int L; const float* wimg; float* gwimg; };
// Shape-only index tables (k-split tables) are built once per shape and kept: the key holds everything the table
// depends on, the table lives in device memory owned by the library (the one exception to "the caller owns every buffer":
// sg_plan_cache_bytes / sg_plan_cache_clear in the header).  A table built on stream A is made visible to a later launch on
// stream B with an event wait.
struct TabEntry { void* dev; size_t bytes; hipEvent_t ready; hipStream_t stream; };
using TabKey = std::array<long long, 12>;
// algorithmic bytes (operands read once + result written once) of the GEMM launches the current entry point issues: set by
// the C-ABI entry points, written next to each launch when SG_LAUNCH_LOG=<file> (tools/pmc_db_summary.py joins that log with
// the PMC database in dispatch order to print traffic / algorithmic-bytes ratios)
extern thread_local double t_alg_bytes;
extern std::mutex g_tab_mu;
extern std::map<TabKey, TabEntry> g_tabs;
extern size_t g_tab_bytes;

// entry points of the gather families (one translation unit each); variant_stride > 0: tap t of the k-table reads from source
// copy t (see reflect_variants_kernel)
int kn0_run(int KS, const float* A, int M, int K, const Gather& g, int NB, const float* bias, float* out, int Mtot, int act,
            float slope, double flops, void* ktab_ws, size_t ws_avail, hipStream_t s);
int kn1_run(int KS, const float* A, int M, int K, const Gather& g, int NB, const float* bias, float* out, int Mtot, int act,
            float slope, double flops, void* ktab_ws, size_t ws_avail, unsigned variant_stride, hipStream_t s);
int kn_sparse_run(int KS, const float* W, int M, int K, const Gather& g, int NB, const float* bias, float* out, int act,
                  float slope, const Sparse& sp, void* ws, hipStream_t s);
int kn_parity_run(int KS, const float* W, int Rdim, int B, int m0, int M, const Gather& g, int NB, const float* bias,
                  float* out, int Mtot, int act, float slope, double flops, void* ws, size_t ws_bytes, hipStream_t s);
// gb (optional): row sums of A over all pixels = the bias gradient, produced by the GEMM's own loaders when the launch plan
// allows it; *gb_done says whether it was (else the caller runs sg_channel_sum)
int nk_run(int KS, const float* A, int M, int Mtot, const Gather& g, int NB, float* out, void* ws, size_t ws_bytes,
           double flops, hipStream_t s, const Sparse* sp, float* gb = nullptr, bool* gb_done = nullptr);
// register-streaming GEMM for small dense layers (skinny.hip): C[M][N] = act(sum_k A(m,k) B(n,k) + bias[n]);
// *_kcontig: 1 = elem(x, k) = p[x*ld + k], 0 = elem(x, k) = p[k*ld + x]
// rowsum (optional): [M] row sums of A over k, written by the workgroups of the first column tile
int skinny_gemm(const float* a, int lda, int a_kcontig, const float* b, int ldb, int b_kcontig, float* c, const float* bias,
                float* rowsum, int M, int N, int K, int act, float slope, hipStream_t s);
int skinny_gemm_gather(const float* obj, const float* pred, const int64_t* edges, int T, int Do, int Dp, const float* w, float* c,
                       const float* bias, int N, int act, float slope, hipStream_t s);
inline bool skinny_shape(int M, int N) {
  return (long)sg_cdiv(M, 32) * sg_cdiv(N, 32) <= (long)sg_opt(SG_OPT_LINEAR_SKINNY);
}
}  // namespace sgk

namespace {
using namespace sgk;
thread_local unsigned t_variant_stride = 0;   // >0: tap t of the k-table reads from source copy t (set by kn1_run)

// -DSG_TIMELINE (tools/probe/build_timeline.sh: a SEPARATE library, never the product build): wave 0 of every workgroup of an
// igemm_kernel launch stamps s_memtime at five points -- entry, loaders initialised, first k-tile staged in LDS, main loop done,
// epilogue stored -- plus its hardware id into a buffer the host hands over (sg_debug_timeline_set_<unit>): where a workgroup's
// life goes, per launch (tools/probe/timeline_probe.py).  Each translation unit has its own copy of the pointer.
#ifdef SG_TIMELINE
__device__ unsigned long long* g_sg_tl = nullptr;
__device__ unsigned g_sg_tl_cap = 0;
#define SG_TL_STAMP(slot)                                                                       \
  do {                                                                                          \
    if (tl_on) tl_rec[slot] = __builtin_amdgcn_s_memtime();                                     \
  } while (0)
#else
#define SG_TL_STAMP(slot) do { } while (0)
#endif

#ifndef SG_NSUB
#define SG_NSUB 1    // sub-tiles per workgroup k-tile (see CfgFor below for why 1)
#endif
constexpr int BK = 16;     // sub-tile depth: the unit one loader call stages (16 k-rows)

// A workgroup k-tile is NSUB sub-tiles deep (BKT = 16*NSUB): all 2*NSUB loader calls of the next tile are issued
// before the MFMAs of the current one, so NSUB*~8 global loads per thread stay in flight across 8*NSUB*TM*TN
// MFMAs (the f32 MFMA is 64 cycles: one sub-tile of work per load round trip left the kernel latency-bound).
#ifndef SG_PIPE_DEFAULT
#define SG_PIPE_DEFAULT 2      // main loop of every instantiation: 0 = plain; 1 = software-pipelined fragment reads (measured on
#endif                         // MI355X: 648 -> 660 images/s, every family +1..6 %); 2 = 1 + the LDS stores of the next tile
                               // interleaved with the MFMAs of phase 0 (855 -> 859 images/s, profiles/r04_ab_sessions.md)
// KFOLD > 0 (pipelined loop only): the k-sum is accumulated in CHUNKS of KFOLD elements -- every KFOLD k the MFMA accumulator is
// added into a second register set and cleared, the chunks meet in ascending order.  The fp32 rounding error of a length-K fma
// chain grows like K; in the transformed domain of Winograd F(4x4,3x3) the running sums are large against the output the
// inverse transform extracts from them, and that accumulation error is what dominated the form's conv-level error (round 6
// study with tools/winograd_f43.py: all-fp32 3.9e-6 of max|y| at K = 1024, 2.5e-6 with 256-chunks, 1.4e-6 with 128-chunks,
// 0.6e-6 with an fp64 accumulator; transforms in fp64 instead: no change).  Cost: 2 x 16 vector-ALU instructions per wave and
// chunk boundary and 16 more registers per 32x32 accumulator.  Deterministic; the order is a function of K alone.
// TS_ = 1 (every pipelined instantiation): the kernel carries the TAIL-SPLIT schedule (BatchInfo::tail_*): see igemm_kernel.
template <int BM_, int BN_, int WGM_, int NSUB_, int PIPE_ = SG_PIPE_DEFAULT, int KFOLD_ = 0, int TS_ = (PIPE_ != 0 ? 1 : 0)>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = 4 / WGM_, NSUB = NSUB_, BKT = BK * NSUB_, PIPE = PIPE_;
  static constexpr int KFOLD = KFOLD_, TAILSPLIT = TS_;
  static_assert(TS_ == 0 || PIPE_ != 0, "tail split: pipelined loop only");
  static_assert(KFOLD_ == 0 || (PIPE_ != 0 && KFOLD_ % (BK * NSUB_) == 0), "KFOLD: whole k-tiles of the pipelined loop");
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int LDA = BM + 4, LDB = BN + 4;
};

// LDS tile layout: [x][k] with k contiguous, row pitch LDK = 16 + 4 floats (80 B).  One ds_read_b128 then feeds FOUR
// MFMA k-steps of a 32x32 fragment row and the pitch makes the b128 fragment READS conflict-free (the 16 lanes of a group
// hit 16 distinct 4-bank slots).  The b128 staging STORES of the K-contiguous loaders are 2-way conflicted on 4 of the 32
// banks (8-lane groups cover two rows: floats 20r .. 20r+15 and 20r+20 .. 20r+35, the last four wrap onto the first four):
// one extra LDS-array cycle per lane group, which is what SQ_LDS_BANK_CONFLICT counts on these kernels (33 % of
// SQ_LDS_IDX_ACTIVE) and which costs <= 0.6 % because a b128 store is bound by its 13-cycle data transfer, not by the array
// (profiles/r04_lds_bank_conflict.md).  MFMA k-step s pairs tile column s (lanes 0-31) with column s+8 (lanes 32-63): any
// pairing is legal as long as A and B use the same one.
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
typedef unsigned int sg_u32x4 __attribute__((ext_vector_type(4)));
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
  // vector form: byte offset of (row, kq_) per pass, fixed for the whole k-loop (rows outside the matrix / the tile: 2^31, which
  // the range check of the buffer load rejects => zeros).  The k advance is wave-uniform and travels in the SGPR offset operand
  // of the buffer load, so a full k-tile costs NO vector-ALU instruction per load (the f32 MFMA does not overlap with VALU
  // work; the 64-bit address arithmetic of the plain loads was 0.5 VALU per MFMA in the dense Winograd GEMM)
  unsigned voff_[PASSES];
  __device__ __forceinline__ void init(int x0, int tid, int*, int, int) {
    x0_ = x0; xr_ = tid >> 2; kq_ = (tid & 3) * 4;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int xl = xr_ + p * 64;
      const int x = x0_ + xl;
      const bool xok = (BX % 64 == 0 || xl < BX) && x < X;
      voff_[p] = xok ? ((unsigned)x * (unsigned)ld + (unsigned)kq_) * 4u : 0x80000000u;
    }
  }
  __device__ __forceinline__ void set_batch(int b, int stride, int) { base += (size_t)b * (size_t)stride; }
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int kend) const {
    st.ok = 0;
#if SG_BUFLOAD
    if (VEC && (!MASK || k0 + BK <= kend)) {      // whole k-tile inside [.., kend): wave-uniform test (always true without MASK)
      const __amdgpu_buffer_rsrc_t rs = sg_rsrc(base);
      const int so = __builtin_amdgcn_readfirstlane(k0 * 4);
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff_[p], so, 0);
        __builtin_memcpy(&st.r[p * 4], &v, 16);
      }
      return;
    }
#endif
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

// The same operand form (x contiguous: elem(x, k) = base[k*ld + x]) for FULL tiles (X % BX == 0, K % 16 == 0), all addressing
// out of the vector ALU: the lane's byte offset x*4 is fixed for the whole k-loop and the k row -- wave-uniform, because the 64
// lanes of a wave own 64 consecutive x of the same ROWS k rows -- travels in the SGPR offset operand of the buffer load.  Batched
// launches (Winograd weight gradient from the forward's V[xi][p][c] and the data gradient's Ytp[xi][p][m]: both "pixel-major",
// i.e. x-contiguous for a GEMM whose k axis is the tile index p): batch b starts b*stride floats further and ``cols`` columns
// later on the x axis.
template <int BX>
struct LoadXContigS {
  const float* base; int ld; int cols;          // cols: columns per batch when the x axis is the batched (B operand) one, else 0
  static constexpr int LDS_INTS = 0;
  static constexpr int ROWS = BX * BK / 256;
  struct Stage { float r[ROWS]; };
  int xl_, kr_, xsub_ = 0;
  unsigned voff_, ld4_;
  __device__ __forceinline__ void set_batch(int b, int stride, int) { base += (size_t)b * (size_t)stride; xsub_ = b * cols; }
  __device__ __forceinline__ void init(int x0, int tid, int*, int, int) {
    xl_ = tid % BX;
    kr_ = __builtin_amdgcn_readfirstlane((tid / BX) * ROWS);
    voff_ = (unsigned)(x0 - xsub_ + xl_) * 4u;
    ld4_ = (unsigned)ld * 4u;
  }
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int) const {
    const __amdgpu_buffer_rsrc_t rs = sg_rsrc(base);
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(k0 + kr_) * ld4_));
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
      st.r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff_, (int)(sb + (unsigned)i * ld4_), 0));
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const { store_krun<ROWS, false>(T, xl_, kr_, st.r, 0u); }
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

// B operand of conv fwd / dgrad when every 16-deep k-tile holds WHOLE channels: taps per channel NT = 1 << lg divides 16
// (4x4 kernels: 16 taps; 1x1: 1; the parity classes of stride-2 3x3 / 4x4 transposed gathers: 1, 2 or 4) and K % 16 == 0.
// The ROWS consecutive k a thread owns then always name the SAME taps, only the channel advances with the k-loop:
//   * the (pixel, tap) byte offsets are computed once in init() and stay in ROWS VGPRs (invalid taps / pixel tail: 2^31,
//     which the range check of the buffer load rejects => 0),
//   * the channel offset is wave-uniform and goes into the SGPR offset operand of the buffer load.
// A gathered element costs one VMEM instruction and NO vector-ALU work, no LDS tap table, no k-split table (LoadGatherKN:
// 1 LDS read + 3 VALU per element and a scalar table fetch per tile whose lgkmcnt wait sat in front of the MFMAs; the f32 MFMA
// does not overlap with VALU work).
struct FixedTaps { int lg; unsigned tapcode; };      // lg < 4: tap ids of the class in the nibbles of tapcode; lg == 4: identity
template <int BN, int MODE>
struct LoadFixedKN {
  Gather g; int Npix; int KS; int lg; unsigned tapcode;
  int dedup;                            // host_prepare(): 1 = compute the distinct tap offsets of a parity class once (init())
  FastDiv dphw, dpw;                    // n -> (image, pixel row): set by host_prepare() / set_class_b() (no division in init())
  static constexpr int LDS_INTS = 0;
  static constexpr int ROWS = BN * BK / 256;
  struct Stage { float r[ROWS]; };
  int nl_, kr_;
  unsigned voff_[ROWS];
  unsigned shw4_;
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void init(int n0, int tid, int*, int, int) {
    nl_ = tid % BN;
    kr_ = __builtin_amdgcn_readfirstlane((tid / BN) * ROWS);   // wave-uniform (BN >= 64)
    const int n = n0 + nl_;
    const bool okn = n < Npix;
    const int nn = okn ? n : 0;
    const int phw = g.PH * g.PW;
    const int img = (int)dphw.div((unsigned)nn);
    const int pix = nn - img * phw;
    const int pi = (int)dpw.div((unsigned)pix);
    const int ph = pi * g.pstep + g.ph0, pw = (pix - pi * g.PW) * g.pstep + g.pw0;
    int ah, aw;
    if (MODE == 0) { ah = ph * g.stride - g.pad; aw = pw * g.stride - g.pad; }
    else { ah = ph + g.pad; aw = pw + g.pad; }
    const unsigned shw = (unsigned)(g.SH * g.SW);
    shw4_ = shw * 4u;
    const unsigned img1 = (unsigned)img * (unsigned)g.C1 * shw;
    const int nt = 1 << lg;
    auto tap_voff = [&](int ti) -> unsigned {
      const int t = lg == 4 ? ti : (int)((tapcode >> (4 * ti)) & 15u);
      // t < KS*KS <= 16: the tap row without an integer division (~25 VALU per tap)
      const int kh = KS == 4 ? (t >> 2) : (KS == 3 ? ((t * 11) >> 5) : (KS == 1 ? 0 : t / KS)), kw = t - kh * KS;
      const int v = okn ? tap_offset<MODE>(g, ah, aw, kh, kw) : -1;
      return v < 0 ? 0x80000000u : (img1 + (unsigned)v) * 4u;
    };
    if (lg <= 2 && dedup) {
      // parity classes (1, 2 or 4 taps): the ROWS = 4 / 8 rows of a thread repeat the class's nt taps -- kr_ is a multiple of ROWS and
      // nt divides ROWS, so row i reads tap i & (nt - 1).  Compute nt offsets instead of ROWS (wave-uniform branch): a new wave's
      // vector instructions each wait behind the other waves' 64-cycle MFMAs, and the 1- and 2-tap classes of a stride-2 3x3
      // transposed gather spent 22 k cycles in this function against a 31-60 k cycle main loop (profiles/r06_timeline_*.txt)
      unsigned o[4];
      o[0] = tap_voff(0);
      o[1] = nt > 1 ? tap_voff(1) : o[0];
      if (nt > 2) { o[2] = tap_voff(2); o[3] = tap_voff(3); } else { o[2] = o[0]; o[3] = o[1]; }
#pragma unroll
      for (int i = 0; i < ROWS; ++i) voff_[i] = o[i & 3];
    } else {
#pragma unroll
      for (int i = 0; i < ROWS; ++i) voff_[i] = tap_voff((kr_ + i) & (nt - 1));
    }
  }
  __device__ __forceinline__ void prefetch(Stage&, int) const {}
  __device__ __forceinline__ void load(Stage& st, int k0, int) const {
    const __amdgpu_buffer_rsrc_t r1 = sg_rsrc(g.src1);
    // channel of element i = (k0 + kr_ + i) >> lg; kr_ is a multiple of ROWS and k0 of 16, so it splits into a per-tile
    // scalar and a loop-invariant per-element scalar
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(k0 + kr_) >> lg) * shw4_));
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const unsigned so = sb + (unsigned)(i >> lg) * shw4_;
      st.r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)voff_[i], (int)so, 0));
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const { store_krun<ROWS, false>(T, nl_, kr_, st.r, 0u); }
};

// k -> (image, pixel) tables of the weight-gradient loaders.  k = img*PQ + pix walks every pixel of every image; splitting it
// costs a division, and the gather side needs a 2-D decode plus padding / reflection on top (~50 VALU instructions).  Done
// per 16-pixel k-tile by a few lanes it sat in EVERY wave's instruction stream (61 VALU per 8 MFMA in the 64x64 kernel: the
// f32 MFMA does not overlap with VALU work, measured with tools/probe/mfma_valu_coexec.hip).  Instead all 256 threads decode
// a BLOCK of 256 pixels at once into a double-buffered LDS table when the k-loop crosses a block boundary (prefetch() runs
// >= 1 barrier ahead of the loads that read it): 1/16 of the instructions per k-tile, and a gathered element costs one
// LDS read + one add.
constexpr int KBLK = 256;

// A operand of wgrad, vector form (PQ % 4 == 0, 16-byte aligned base): each thread moves one float4 of four
// consecutive pixels of one row: 1 global_load_dwordx4 + 1 ds_write_b128 per 4 elements
template <int BM>
struct LoadPixKVec {
  const float* base; int M, Mtot, PQ; FastDiv dPQ;
  float* rowsum;                 // optional [z][M] slab: row sums of the operand over the workgroup's k-range (= bias gradient)
  static constexpr int LDS_INTS = 2 * KBLK;
  static constexpr int Q = (BM * 4 + 255) / 256;
  struct Stage { float4 r[Q]; unsigned ok; };
  int row_, kq_, tid_, kbeg_, kend_, m0_;
  unsigned rowoff_[Q];
  int* lds_;
  mutable float racc_[Q];
  bool rs_on_ = false;
  __device__ __forceinline__ void init(int m0, int tid, int* lds, int kbeg, int kend) {
    row_ = tid >> 2; kq_ = (tid & 3) * 4; tid_ = tid; lds_ = lds; kbeg_ = kbeg; kend_ = kend; m0_ = m0;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      // rows beyond M read the last valid row: their products land in accumulator rows the epilogue never stores
      const int m = min(m0 + row_ + 64 * i, M - 1);
      rowoff_[i] = (unsigned)m * (unsigned)PQ;
      racc_[i] = 0.f;
    }
  }
  // row sums (see rowsum): only the workgroups of the first column tile accumulate (uniform branch in store())
  __device__ __forceinline__ void rowsum_begin(bool first_col_tile) { rs_on_ = rowsum != nullptr && first_col_tile; }
  __device__ __forceinline__ void rowsum_finish(int z) const {
    if (!rs_on_) return;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      float v = racc_[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const int m = m0_ + row_ + 64 * i;
      if ((tid_ & 3) == 0 && row_ + 64 * i < BM && m < M) rowsum[(size_t)z * M + m] = v;
    }
  }
  __device__ __forceinline__ void set_batch(int, int, int) {}
  __device__ __forceinline__ void prefetch(Stage&, int k0) const {
    const int rel = k0 - kbeg_;
    if ((rel & (KBLK - 1)) != 0) return;
    const int k = k0 + tid_;
    const unsigned kk = k < kend_ ? (unsigned)k : 0u;
    const unsigned img = dPQ.div(kk), pix = kk - img * (unsigned)PQ;
    lds_[((rel / KBLK) & 1) * KBLK + tid_] = k < kend_ ? (int)(img * (unsigned)Mtot * (unsigned)PQ + pix) : (int)ELEM_INVALID;
  }
  __device__ __forceinline__ void load(Stage& st, int k0, int) const {
    const int rel = k0 - kbeg_;
    const unsigned p0 = (unsigned)lds_[((rel / KBLK) & 1) * KBLK + (rel & (KBLK - 1)) + kq_];     // ELEM_INVALID beyond kend
    st.ok = p0 < ELEM_INVALID ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
#if SG_BUFLOAD
      st.r[i] = sg_bufload4(sg_rsrc(base), p0 + rowoff_[i]);                  // rejected by the range check => zeros
#else
      st.r[i] = *reinterpret_cast<const float4*>(base + (st.ok ? p0 + rowoff_[i] : 0u));
#endif
    }
  }
  __device__ __forceinline__ void store(const Stage& st, float* T) const {
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      if (row_ + 64 * i < BM) {
        float4 v = st.r[i];
#if !SG_BUFLOAD
        if (!st.ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        *reinterpret_cast<float4*>(T + (row_ + 64 * i) * LDK + kq_) = v;
        if (rs_on_) racc_[i] += (v.x + v.y) + (v.z + v.w);
      }
    }
  }
};

// A operand of wgrad: elem(m, k) = base[(img*Mtot + m)*PQ + pix], k = img*PQ + pix.  Lanes run along k.
template <int BM, bool MASK = true>
struct LoadPixK {
  const float* base; int M, Mtot, PQ; FastDiv dPQ;
  float* rowsum;                 // see LoadPixKVec
  static constexpr int LDS_INTS = 0;
  static constexpr int ROWS = BM / 16;
  struct Stage { float r[ROWS]; unsigned ok; };
  int m0_, mr_, kl_;
  mutable float racc_[ROWS];
  bool rs_on_ = false;
  __device__ __forceinline__ void init(int m0, int tid, int*, int, int) {
    m0_ = m0; kl_ = tid & 15; mr_ = tid >> 4;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) racc_[i] = 0.f;
  }
  __device__ __forceinline__ void rowsum_begin(bool first_col_tile) { rs_on_ = rowsum != nullptr && first_col_tile; }
  __device__ __forceinline__ void rowsum_finish(int z) const {
    if (!rs_on_) return;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      float v = racc_[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      const int m = m0_ + mr_ + 16 * i;
      if (kl_ == 0 && m < M) rowsum[(size_t)z * M + m] = v;
    }
  }
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
    for (int i = 0; i < ROWS; ++i) {
      const float v = (SG_BUFLOAD || !MASK || ((st.ok >> i) & 1u)) ? st.r[i] : 0.f;
      T[(mr_ + 16 * i) * LDK + kl_] = v;
      if (rs_on_) racc_[i] += v;
    }
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
  int zdiv;                                             // k-chunks per image (informational: the image comes from kbeg)
  FastDiv dphw, dpw;                                    // k -> (image, pixel row) without integer divisions: host_prepare()
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
    // the image of a per-image k-chunk launch (BatchInfo::ksplit: chunk (img, q) starts at pixel img * PH*PW + q * kcs); taken
    // from kbeg, not from blockIdx.z: the XCD pinning of igemm_kernel re-numbers the z slices
    const int zimg = chan_list ? kbeg / (g.PH * g.PW) : 0;
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
      // (two emulated 32-bit divisions per table entry, ~272 entries per 16-pixel k-tile, sat in front of every tile's MFMAs:
      //  round 6 -- multiply-high + shift with host-prepared constants, as in the other loaders)
      const int img = (int)dphw.div((unsigned)kk), pix = kk - img * phw;
      const int ph = (int)dpw.div((unsigned)pix), pw = pix - ph * g.PW;
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
  static constexpr int LDS_INTS = 2 * KBLK * (TWO ? 2 : 1);      // off1[2][256] (, off2[2][256]): see KBLK above
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
  __device__ __forceinline__ void prefetch(Stage&, int k0) const {
    const int rel = k0 - kbeg_;
    if ((rel & (KBLK - 1)) != 0) return;
    int* buf = lds_ + ((rel / KBLK) & 1) * KBLK;
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
    if (TWO) buf[2 * KBLK + tid_] = (SG_BUFLOAD && MASK && !valid) ? (int)ELEM_INVALID
                                        : (int)(g.bcast2 ? img * (unsigned)g.C2 : img * (unsigned)g.C2 * shw + tp);
  }
  __device__ __forceinline__ void load(Stage& st, int k0, int) const {
    const int rel = k0 - kbeg_;
    const int* buf = lds_ + ((rel / KBLK) & 1) * KBLK + (rel & (KBLK - 1));
    const int o1 = buf[kl_];
#if SG_BUFLOAD
    if (MASK) {        // invalid pixels carry ELEM_INVALID: the buffer load's range check returns zeros, nothing to select
      const unsigned u1 = (unsigned)o1, u2 = TWO ? (unsigned)buf[2 * KBLK + kl_] : 0u;
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
    const unsigned b2 = (TWO && ok) ? (unsigned)buf[2 * KBLK + kl_] : 0u;
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
  FastDiv dPHW, dPWs;                   // n -> image, pixel -> row of the sub-lattice: set by host_prepare() / set_class_ep()
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
        const int img = (int)dPHW.div((unsigned)n);
        int pix = n - img * PHW, plane = PHW;
        if (step > 1) {
          const int i = (int)dPWs.div((unsigned)pix);
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
  int lg[4]; unsigned tapcode[4];        // LoadFixedKN: taps per channel (log2) and tap ids of each class
  unsigned dphw_m[4], dphw_s[4], dpw_m[4], dpw_s[4];      // FastDiv(PH * PW), FastDiv(PW) of each class (launch_cfg fills them)
  // split[c] > 1 (launch_cfg): the tiles of class c run as split[c] workgroups of 1 / split[c] of its k range each (tile0 then
  // counts WORKGROUPS, the pieces of a tile adjacent); slot0[c]: first ticket slot of the class within a tile row, nslots: per row
  int split[4], slot0[4], nslots;
};
struct BatchInfo {
  int cols_per_batch; int nbatch; const int* kcnt; int a_stride; int b_stride;
  // K = (image, pixel) GEMMs split so that no k-chunk straddles an image: grid.z = image * ksplit + q, chunk q of
  // image i covers pixels [i*kimg + q*kcs, min((i+1)*kimg, ... + kcs))  (ksplit == 0: plain blockIdx.z * kchunk)
  int kimg, ksplit, kcs;
  // batch_major: tiles are numbered batch-major (all tiles of batch 0, then batch 1, ...), so that with the XCD remap
  // below every XCD works on whole batches and their operands stay in ITS L2 (batched Winograd GEMMs)
  int batch_major;
  // xcd_z: plain split-K launch whose grid.z is a multiple of 8: k-chunk z runs on XCD z % 8 (see the kernel)
  int xcd_z;
  // par_chunk (parity-class launches): tiles go to the XCDs in chunks of this many (power of two) instead of one contiguous
  // eighth of the tile range per XCD; prio: raise the wave priority outside the main loop (both: see the kernel; set by launch_cfg)
  int par_chunk, prio;
  // tail split (TileCfg::TAILSPLIT kernels, batch_major launches whose tile count leaves HALF a round of workgroups per CU:
  // 36 x 32 = 1152 tiles of the F(4x4,3x3) GEMMs on 256 CUs = 4.5 per CU): tail_sx > 0 = tiles per XCD that run as TWO workgroups
  // of half the k range each; grid.x = tiles + 8 * tail_sx.  tail_slab: 2 x 16 KB x TM x TN per split tile of raw accumulators,
  // tail_cnt: one arrival counter per split tile (zero, reset by the last arriver).
  int tail_sx; float* tail_slab; int* tail_cnt;
  // general form (plain launches: grid.z == 1, no batches, no parity classes): the LAST tail_n tiles run as tail_s workgroups of
  // 1 / tail_s of the k range each, numbered before (tail_first) or after the whole tiles; grid.x = tiles + tail_n (tail_s - 1)
  int tail_n, tail_s, tail_first;
  ParityClasses par;
};
// per-class hooks: loaders / epilogues that can run a parity class overload these; everything else ignores the call
template <class L> __device__ __forceinline__ void rowsum_begin(L&, bool) {}
template <class L> __device__ __forceinline__ void rowsum_finish(const L&, int) {}
template <int BM> __device__ __forceinline__ void rowsum_begin(LoadPixKVec<BM>& l, bool f) { l.rowsum_begin(f); }
template <int BM> __device__ __forceinline__ void rowsum_finish(const LoadPixKVec<BM>& l, int z) { l.rowsum_finish(z); }
template <int BM, bool MASK> __device__ __forceinline__ void rowsum_begin(LoadPixK<BM, MASK>& l, bool f) { l.rowsum_begin(f); }
template <int BM, bool MASK> __device__ __forceinline__ void rowsum_finish(const LoadPixK<BM, MASK>& l, int z) { l.rowsum_finish(z); }
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
template <int BN, int MODE>
__device__ __forceinline__ void set_class_b(LoadFixedKN<BN, MODE>& l, const ParityClasses& p, int c) {
  l.g.PH = p.PH[c]; l.g.PW = p.PW[c]; l.g.ph0 = p.ph0[c]; l.g.pw0 = p.pw0[c]; l.Npix = p.Npix[c];
  l.lg = p.lg[c]; l.tapcode = p.tapcode[c];
  l.dphw.m = p.dphw_m[c]; l.dphw.s = p.dphw_s[c]; l.dphw.d = (unsigned)(p.PH[c] * p.PW[c]);
  l.dpw.m = p.dpw_m[c]; l.dpw.s = p.dpw_s[c]; l.dpw.d = (unsigned)p.PW[c];
}
__device__ __forceinline__ void set_class_ep(EpNCHW& e, const ParityClasses& p, int c) {
  e.PHW = p.PH[c] * p.PW[c]; e.Npix = p.Npix[c]; e.PWs = p.PW[c]; e.h0 = p.ph0[c]; e.w0 = p.pw0[c];
  e.dPHW.m = p.dphw_m[c]; e.dPHW.s = p.dphw_s[c]; e.dPHW.d = (unsigned)e.PHW;
  e.dPWs.m = p.dpw_m[c]; e.dPWs.s = p.dpw_s[c]; e.dPWs.d = (unsigned)e.PWs;
}
// host side, just before a launch: operands that carry FastDiv members get them from their own fields
template <class T> inline void host_prepare(T&) {}
inline void host_prepare(EpNCHW& e) {
  e.dPHW = FastDiv((unsigned)(e.PHW > 0 ? e.PHW : 1));
  e.dPWs = FastDiv((unsigned)(e.PWs > 0 ? e.PWs : 1));
}
template <int BN, int KS, bool TWO, bool MASK, int NS> inline void host_prepare(LoadGatherNK<BN, KS, TWO, MASK, NS>& l) {
  const int phw = l.g.PH * l.g.PW;
  l.dphw = FastDiv((unsigned)(phw > 0 ? phw : 1));
  l.dpw = FastDiv((unsigned)(l.g.PW > 0 ? l.g.PW : 1));
}
template <int BN, int MODE> inline void host_prepare(LoadFixedKN<BN, MODE>& l) {
  l.dedup = sg_opt(SG_OPT_FIXEDTAP) != 2 ? 1 : 0;
  const int phw = l.g.PH * l.g.PW;
  l.dphw = FastDiv((unsigned)(phw > 0 ? phw : 1));
  l.dpw = FastDiv((unsigned)(l.g.PW > 0 ? l.g.PW : 1));
}
template <class CFG, class AL, class BL, class EP>
__global__ void __launch_bounds__(256) igemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int kchunk, BatchInfo bi) {
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN, LDA = CFG::LDA, LDB = CFG::LDB;
  constexpr int NSUB = CFG::NSUB, BKT = CFG::BKT;
  // tail split compiled in (weight-gradient launches are split-K launches with row sums: never eligible, and the extra code cost
  // the 128x128 tap-major instantiation its second wave per SIMD)
  constexpr bool TSPLIT = CFG::TAILSPLIT != 0 && !std::is_same<EP, EpWgrad>::value;
  __shared__ __attribute__((aligned(16))) float As[2][NSUB * BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][NSUB * BN * LDK];
  __shared__ int tapA[AL::LDS_INTS > 0 ? AL::LDS_INTS : 1];
  __shared__ int tapB[BL::LDS_INTS > 0 ? BL::LDS_INTS : 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / CFG::WGN) * CFG::WM, wn0 = (wid % CFG::WGN) * CFG::WN;
#ifdef SG_TIMELINE
  const unsigned tl_wg = blockIdx.x + gridDim.x * blockIdx.z;
  const bool tl_on = g_sg_tl != nullptr && tid == 0 && tl_wg < g_sg_tl_cap;
  unsigned long long* tl_rec = g_sg_tl + (size_t)tl_wg * 8;
  SG_TL_STAMP(0);
  if (tl_on) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tl_rec[5] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif

  // XCD-aware, bijective tile remap: consecutive tiles (same weight rows, overlapping gathers) share an L2
  const int tiles_pb = bi.cols_per_batch > 0 ? (bi.cols_per_batch + BN - 1) / BN : 0;
  const int tiles_n = bi.cols_per_batch > 0 ? bi.nbatch * tiles_pb : (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  int ts_half = -1, ts_slot = 0, ts_s = 2;     // tail split: which piece of the k range this workgroup runs (-1: all of it), of how many
  // Prologue and epilogue of a workgroup are a few hundred vector-ALU / scalar instructions (index arithmetic, store addresses)
  // on a SIMD whose other resident waves issue 64-cycle f32 MFMAs back to back; the f32 MFMA does not co-execute with VALU
  // work and the arbiter serves the oldest wave first, so the NEW wave's instructions each waited for a whole MFMA slot:
  // s_memtime stamps (tools/probe/timeline_probe.py, profiles/r06_timeline_*.txt) showed 12-25 k cycles of "init" and 7-8 k of
  // epilogue in workgroups whose main loop is 28-130 k.  Raised priority outside the main loop (option wave_prio) gets those
  // phases out of the way -- and was measured a net LOSS: what the new wave wins, the waves in their main loops lose (most classes
  // +-1 %, the 3x3 transposed gather at 128x128 +11 %, the x-contiguous F(4x4,3x3) GEMMs +7 %, the step -0.3 %:
  // profiles/r06_gemm_prio.md).  The switch stays for the record, default off.
  if (bi.prio) __builtin_amdgcn_s_setprio(3);
  if (bi.par.ncls > 0 && bi.par_chunk > 0) {
    // Parity classes of a 3x3 stride-2 transposed gather have 4 / 2 / 2 / 1 taps, i.e. K extents 4 : 2 : 2 : 1, and their tiles
    // are numbered class by class (heaviest first).  With one contiguous eighth of the tile range per XCD (below), XCDs 0-1 ran
    // ALL tiles of the 4-tap class and XCDs 6-7 only 1-tap tiles whenever M fits one tile row: the launch took as long as XCD 0
    // (timeline probe: 300 us where the balanced schedule needs ~180).  Chunks of par_chunk consecutive tiles (same weights,
    // neighbouring pixels: they still share an L2) are dealt to the XCDs round-robin instead: every XCD gets an eighth of every
    // class, the heavy class is still dispatched first.
    const int G = bi.par_chunk, full = nwg / (8 * G) * (8 * G);
    if (bid < full) {
      const int xcd = bid & 7, idx = bid >> 3;
      bid = ((idx / G) * 8 + xcd) * G + (idx % G);
    }
  } else if (TSPLIT && bi.tail_n > 0) {
    // Tail split, general form.  A launch of T tiles runs as rounds of (CUs x resident workgroups) and its last round is rarely
    // full: the workgroups of that round run with the CU nearly to themselves -- a lone wave per SIMD drives the matrix pipe at
    // 48 % (timeline probe) -- while the rest of the chip idles; with 1.3-4.5 rounds per launch (every conv GEMM of the step) that
    // is 10-30 % of the launch.  The last tail_n tiles therefore run as tail_s workgroups of 1 / tail_s of the k range each:
    // more, shorter workgroups in the ragged round (dispatched last), or -- when everything is resident at once -- short extra
    // workgroups next to the whole tiles of every CU (dispatched first).  The pieces of a tile meet in the epilogue.
    const int s_ = bi.tail_s, P = bi.tail_n * s_, nfull = nwg - P;
    const int b = bid;
    const bool piece = bi.tail_first ? b < P : b >= nfull;
    if (piece) {
      const int pb = bi.tail_first ? b : b - nfull;
      ts_slot = pb / s_;
      ts_half = pb - ts_slot * s_;
      ts_s = s_;
      bid = nfull + ts_slot;
    } else {
      const int f = bi.tail_first ? b - P : b;      // (host: P % 8 == 0 when the pieces come first, so f & 7 is still the XCD)
      const int q = nfull >> 3, rem = nfull & 7, xcd = f & 7, idx = f >> 3;
      bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
  } else if (TSPLIT && bi.tail_sx > 0) {
    // Tail split.  T tiles on 256 CUs with T % 256 == 128: every CU runs 4 workgroups and half of the CUs a 5th -- five resident
    // workgroups share the matrix pipe of those CUs (2560 cycles per k-tile instead of 2170: tools/probe/timeline_probe.py) and
    // the launch ends when THEY end, the other half of the chip idle for the last ~10 % (profiles/r06_timeline_after_Gres_fwd.txt).
    // Instead sx = T / 16 / 8 tiles per XCD run as TWO workgroups of half the k range each, numbered FIRST so that every CU gets
    // one of them next to its four whole tiles: 4.5 tiles of work on every CU.  The halves meet in the epilogue (below).
    const int per_x = nwg >> 3, xcd = bid & 7, idx = bid >> 3, sx = bi.tail_sx, tx = per_x - sx;      // tx tiles per XCD
    int lt;
    if (idx < 2 * sx) {
      ts_half = idx & 1;
      const int q = idx >> 1;
      ts_slot = xcd * sx + q;
      lt = (xcd & 1) ? q : tx - sx + q;                  // the half batch at the seam between XCD 2i and 2i + 1
    } else {
      const int f = idx - 2 * sx;
      lt = (xcd & 1) ? sx + f : f;
    }
    bid = xcd * tx + lt;
  } else {
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
    int tq = tn - bi.par.tile0[c];
    if (TSPLIT && bi.par.split[c] > 1) {
      // The 4-tap class of a 3x3 stride-2 transposed gather runs 4x the k range of the 1-tap class.  With every workgroup of the
      // launch resident at once (few tiles: the generator's 8x8 -> 16x16 up-conv and its mirror) the launch ends with the 4-tap
      // workgroups ALONE on their CUs -- one wave per SIMD drives the matrix pipe at 48 % (timeline probe).  Its tiles therefore run
      // as two half-k workgroups that meet through the tail-split ticket below: the longest workgroup is 2 units instead of 4.
      const int sc = bi.par.split[c];
      ts_half = tq % sc;
      tq /= sc;
      ts_s = sc;
      ts_slot = (bid / tn_all) * bi.par.nslots + bi.par.slot0[c] + tq;
    }
    n0 = tq * BN;
    K = bi.par.K[c];
    set_class_a(al, bi.par.aoff[c], K);
    set_class_b(bl, bi.par, c);
    set_class_ep(ep, bi.par, c);
  }
  int zblk = blockIdx.z;
  if (bi.xcd_z) {
    // Weight-gradient GEMMs are split-K launches of FEW output tiles over a LONG k axis (pixels): the tiles of one k-chunk all
    // read the same gy / x pixels.  Workgroups are handed to the 8 XCDs round-robin in linear (z, x) order, which spreads the
    // tiles of a chunk over all eight L2s -- each L2 fetched the chunk on its own (5-9x the algorithmic bytes at the fabric,
    // profiles/r03_pmc_traffic.md).  Re-numbered so that XCD i runs ALL tiles of the chunks i, i+8, ...: the operands of a
    // chunk are fetched into one L2 once.  (grid.z % 8 == 0, so every XCD gets whole chunks.)
    const unsigned lin = blockIdx.z * gridDim.x + blockIdx.x, xcd = lin & 7u, idx = lin >> 3;
    zblk = (int)(xcd + 8u * (idx / gridDim.x));
    const int t = (int)(idx % gridDim.x);
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
  }
  int kbeg = zblk * kchunk;
  int kend = min(K, kbeg + kchunk);
  if (bi.ksplit > 0) {
    const int img = zblk / bi.ksplit, q = zblk - img * bi.ksplit;
    kbeg = img * bi.kimg + q * bi.kcs;
    kend = min((img + 1) * bi.kimg, kbeg + bi.kcs);
  }
  if (TSPLIT && ts_half >= 0) {
    // piece p of ts_s: whole k-tiles, ceil(tiles / ts_s) each (host: boundaries are multiples of KFOLD where the kernel has one);
    // a piece beyond the end of the range is empty (it still takes its ticket)
    const int kt = (kend - kbeg + BKT - 1) / BKT, per = (kt + ts_s - 1) / ts_s;
    const int kb = min(kend, kbeg + ts_half * per * BKT);
    kend = min(kend, kb + per * BKT);
    kbeg = kb;
  }
  if (bi.cols_per_batch > 0) {
    int tn = bid % tiles_n, batch = tn / tiles_pb;
    if (bi.batch_major) {
      const int per_batch = (int)((gridDim.x - 8u * (unsigned)(TSPLIT ? bi.tail_sx : 0)) / (unsigned)bi.nbatch);   // = tiles_m * tiles_pb
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
  rowsum_begin(al, n0 == 0);
#ifdef SG_TIMELINE
  if (tl_on) { tl_rec[6] = ((unsigned long long)(unsigned)(kend - kbeg) << 32) | (unsigned)n0; tl_rec[7] = (unsigned)m0; }
#endif
  SG_TL_STAMP(1);

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
  SG_TL_STAMP(2);
  if (bi.prio) __builtin_amdgcn_s_setprio(0);

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
    constexpr bool FOLD = CFG::KFOLD > 0;
    f32x16 acc2[FOLD ? TM : 1][FOLD ? TN : 1];
    if constexpr (FOLD) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    for (int k0 = kbeg; k0 < kend; k0 += BKT) {
      const bool more1 = k0 + BKT < kend, more2 = k0 + 2 * BKT < kend;
      if constexpr (FOLD) {
        // chunk boundary (wave-uniform): at the top of an iteration ``acc`` holds exactly the products of [chunk start, k0)
        if (k0 > kbeg && ((k0 - kbeg) % CFG::KFOLD) == 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) { acc2[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
      }
      // PIPE == 2 (dense Winograd GEMMs): the LDS stores of the next tile do not sit in front of the iteration's first MFMA
      // (8 ds_write_b128 = ~100 cycles during which this wave feeds nothing to the matrix pipe) but BETWEEN the MFMAs of phase 0,
      // one store call after each group of four: a 64-cycle MFMA covers the 13 cycles a store takes to issue.  The global loads
      // of the tile after next re-use the staging registers, so they follow the stores (one phase later than with PIPE == 1).
      constexpr bool INTER = CFG::PIPE == 2 && 2 * NSUB <= 4;
      auto stage_next = [&]() {
        if (more2) {
#pragma unroll
          for (int u = 0; u < NSUB; ++u) { al.load(sa[u], k0 + 2 * BKT + u * BK, kend); bl.load(sb[u], k0 + 2 * BKT + u * BK, kend); }
#pragma unroll
          for (int u = 0; u < NSUB; ++u) { al.prefetch(sa[u], k0 + 3 * BKT + u * BK); bl.prefetch(sb[u], k0 + 3 * BKT + u * BK); }
        }
      };
      if (!INTER) {
        if (more1) {
#pragma unroll
          for (int u = 0; u < NSUB; ++u) { al.store(sa[u], As[buf ^ 1] + u * BM * LDK); bl.store(sb[u], Bs[buf ^ 1] + u * BN * LDK); }
        }
        stage_next();
      }
#pragma unroll
      for (int p = 0; p < P - 1; ++p) {
        read_frag(buf, p + 1, fa[(p + 1) & 1], fb[(p + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);          // keep the reads IN FRONT of the MFMAs that cover their latency
        if (INTER && p == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                const float av = e == 0 ? fa[0][i].x : (e == 1 ? fa[0][i].y : (e == 2 ? fa[0][i].z : fa[0][i].w));
                const float bv = e == 0 ? fb[0][j].x : (e == 1 ? fb[0][j].y : (e == 2 ? fb[0][j].z : fb[0][j].w));
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
              }
            __builtin_amdgcn_sched_barrier(0);
            if (more1 && e < 2 * NSUB) {
              const int u = e >> 1;
              if ((e & 1) == 0) al.store(sa[u], As[buf ^ 1] + u * BM * LDK);
              else bl.store(sb[u], Bs[buf ^ 1] + u * BN * LDK);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          stage_next();
        } else {
          mma(fa[p & 1], fb[p & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (more1) read_frag(buf ^ 1, 0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      mma(fa[(P - 1) & 1], fb[(P - 1) & 1]);
      buf ^= 1;
    }
    if constexpr (FOLD) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = acc2[i][j][r] + acc[i][j][r];
    }
    SG_TL_STAMP(3);
    if (bi.prio) __builtin_amdgcn_s_setprio(3);
    if constexpr (TSPLIT) {
      if (ts_half >= 0) {
        // The pieces of a split tile: each dumps its raw accumulators (write-through 16-byte stores, lane-contiguous: 1 KB per
        // wave instruction) and takes a ticket; the one that arrives LAST combines and stores the tile.  No waiting, no pre-zeroed
        // output.  Two pieces: the last arriver adds the other's dump to its registers (x + y is commutative: the result does not
        // depend on who arrives last); more: it re-reads ALL dumps, its own included, and adds them in piece order.
        // (Visibility: sc1 stores drained before the ticket, sc1 loads after it -- the recipe of sg_arrive_last, common.h.)
        // (the ticket flag lives in the first word of the A tile: the main loop is over, and sg_arrive_last passes a barrier before
        //  it writes -- a separate __shared__ int pushed the 128x128 instantiations from 81920 to 81924 bytes of LDS, i.e. from two
        //  workgroups per CU to one)
        int* ts_flag = reinterpret_cast<int*>(&As[0][0]);
        constexpr int PERW = TM * TN * 16 * 64;                                  // floats one wave dumps
        float* mine = bi.tail_slab + ((size_t)(ts_slot * ts_s + ts_half) * 4 + wid) * PERW;
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(mine, 0, PERW * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              sg_u32x4 d;
              __builtin_memcpy(&d, reinterpret_cast<const char*>(&acc[i][j]) + 16 * v, 16);
              __builtin_amdgcn_raw_buffer_store_b128(d, rm, (((i * TN + j) * 4 + v) * 64 + lane) * 16, 0, 16 /* sc1 */);
            }
        if (!sg_arrive_last(bi.tail_cnt + ts_slot, ts_s, ts_flag)) return;
        const bool two = ts_s == 2;
        if (!two) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        for (int pc = 0; pc < ts_s; ++pc) {
          if (two && pc == ts_half) continue;
          const float* theirs = bi.tail_slab + ((size_t)(ts_slot * ts_s + pc) * 4 + wid) * PERW;
          const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(theirs), 0, PERW * 4, 0x00020000);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const auto d = __builtin_amdgcn_raw_buffer_load_b128(rt, (((i * TN + j) * 4 + v) * 64 + lane) * 16, 0, 16 /* sc1 */);
                float4 f;
                __builtin_memcpy(&f, &d, 16);
                acc[i][j][4 * v + 0] += f.x; acc[i][j][4 * v + 1] += f.y; acc[i][j][4 * v + 2] += f.z; acc[i][j][4 * v + 3] += f.w;
              }
        }
      }
    }
    rowsum_finish(al, zblk);
    ep.store(acc, m0 + wm0, n0 + wn0, lane, zblk);
#ifdef SG_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);                 // (vmcnt(0): the stamp is taken when this wave's stores have been accepted)
#endif
    SG_TL_STAMP(4);
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
  SG_TL_STAMP(3);
  if (bi.prio) __builtin_amdgcn_s_setprio(3);
  rowsum_finish(al, zblk);
  ep.store(acc, m0 + wm0, n0 + wn0, lane, zblk);
#ifdef SG_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SG_TL_STAMP(4);
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
#ifndef SG_NSW
#define SG_NSW 1
#endif
constexpr int NSW = SG_NSW;
using CfgW128 = TileCfg<128, 128, 2, NSW>;
using CfgW64 = TileCfg<64, 64, 2, NSW>;
using CfgW32 = TileCfg<32, 128, 1, NSW>;
using CfgW64W = TileCfg<64, 128, 2, NSW>;
using CfgDP128 = TileCfg<128, 128, 2, 2, 1>; // ... software-pipelined fragment reads, barrier before the last phase
using CfgDI128 = TileCfg<128, 128, 2, 2, 2>; // ... and the LDS stores of the next tile interleaved with the MFMAs of phase 0

inline int pick_tile(int M, int N) {
  const int force = sg_opt(SG_OPT_TILE);
  if (force >= 0 && force <= 3) return force;
  if (M <= 32) return 2;
  // measured (tools/bench_conv.py): 128x128 tiles win when M is large (>= 512 rows, split-K fills the chip) or when
  // there are enough of them anyway; 64-row tiles otherwise -- 128 pixels wide (tile 3) while that still leaves >= 3
  // workgroups per CU, else 64x64
  const long t128 = (long)sg_cdiv(M, 128) * sg_cdiv(N, 128);
  const bool low_waste = sg_cdiv(M, 128) * 128 * 20 <= M * 23 && N >= 512;
  const int t128_min = sg_opt(SG_OPT_T128_MIN), wide_min = sg_opt(SG_OPT_TILE3_MIN), wide = sg_opt(SG_OPT_TILE3);     // tuning aids
  // (two 128x128 workgroups are resident per CU: a launch of a few more than 512 / 1024 of them runs a nearly empty extra round --
  //  545 tiles, the first PatchGAN layer on the 2N batch, took 207 us against 189 us on 64x64 tiles: profiles/r06_gemm_tile_sweep.md)
  const long over = t128 % 512;
  const bool ragged128 = t128 > 512 && t128 < 1536 && over > 0 && over < 128;
  if (M >= 96 && low_waste && !ragged128 && (M >= 512 || t128 >= t128_min)) return 0;
  if (wide && (long)sg_cdiv(M, 64) * sg_cdiv(N, 128) >= wide_min) return 3;
  return 1;
}

// thread-local launch modifiers (set by the sparse entry points around a regular dispatch)
thread_local BatchInfo t_batch = {};
thread_local int t_grid_z = 0;                // >0: explicit grid.z (per-image k-chunks, see BatchInfo::ksplit)
thread_local int t_fixed_kchunk = 0;        // >0: grid.z = ceil(K / chunk) with exactly this chunk (one image per z)
thread_local int t_xcd_z = 0;               // 1: pin the k-chunks of a plain split-K launch to XCDs (weight gradients)
thread_local int t_min_z = 0;               // >0: at least this many k-chunks (trailing ones may be empty: zero slabs)

template <class CFG, class AL, class BL, class EP>
int launch_cfg(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int splits, hipStream_t s) {
  int tiles = sg_cdiv(M, CFG::BM) * sg_cdiv(N, CFG::BN);
  if (t_batch.cols_per_batch > 0) tiles = sg_cdiv(M, CFG::BM) * t_batch.nbatch * sg_cdiv(t_batch.cols_per_batch, CFG::BN);
  if (t_batch.par.ncls > 0) tiles = sg_cdiv(M, CFG::BM) * t_batch.par.tile0[t_batch.par.ncls];
  int kchunk = K;
  if (splits > 1) kchunk = sg_cdiv(sg_cdiv(K, splits), CFG::BKT) * CFG::BKT;
  if (t_fixed_kchunk > 0) kchunk = t_fixed_kchunk;
  dim3 grid(tiles, 1, t_grid_z > 0 ? t_grid_z : ((splits > 1 || t_fixed_kchunk > 0) ? sg_cdiv(K, kchunk) : 1));
  if (t_min_z > 0 && (int)grid.z < t_min_z && t_grid_z == 0) grid.z = t_min_z;
  BatchInfo bi = t_batch;
  bi.prio = sg_opt(SG_OPT_WAVE_PRIO) ? 1 : 0;
  bi.par_chunk = 0;
  {
    const int g = sg_opt(SG_OPT_PAR_XCD_CHUNK);
    if (bi.par.ncls > 0 && g > 0 && (g & (g - 1)) == 0) bi.par_chunk = g;
  }
  // k-chunks pinned to XCDs: plain split-K launches, and the per-image k-chunks (ksplit) of the factored stem's weight gradient
  // (grid.z = images x chunks: its column tiles re-read the same gy chunk from seven L2s -- 6.5x the algorithmic bytes in round 4)
  const bool plain_z = t_grid_z == 0 && t_fixed_kchunk == 0 && bi.ksplit == 0;
  const bool image_z = t_grid_z > 0 && bi.ksplit > 0 && t_xcd_z == 2;
  bi.xcd_z = (t_xcd_z && (plain_z || image_z) && bi.cols_per_batch == 0 && bi.par.ncls == 0 && grid.z >= 8 && grid.z % 8 == 0) ? 1 : 0;
  bi.tail_sx = 0; bi.tail_slab = nullptr; bi.tail_cnt = nullptr; bi.tail_n = 0; bi.tail_s = 0; bi.tail_first = 0;
  for (int c = 0; c < 4; ++c) { bi.par.split[c] = 1; bi.par.slot0[c] = 0; }
  bi.par.nslots = 0;
  if constexpr (CFG::TAILSPLIT != 0 && !std::is_same<EP, EpWgrad>::value) {
    static const int n_cu = [] { int d = 0, v = 0; hipGetDevice(&d); hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v; }();
    // workgroups of THIS instantiation a CU holds at once (registers / LDS); asked for by the general form only
    auto resident_wgs = [] {
      static const int c = [] {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, reinterpret_cast<const void*>(&igemm_kernel<CFG, AL, BL, EP>), 256, 0) != hipSuccess) v = 0;
        return v;
      }();
      return c;
    };
    const int mode = sg_opt(SG_OPT_W43_TAIL_SPLIT);     // 0: off, 1: the F(4x4,3x3) half-round case only, 2: + the general form
    constexpr size_t PIECE_BYTES = (size_t)4 * CFG::TM * CFG::TN * 16 * 64 * sizeof(float);
    const bool plain = grid.z == 1 && splits <= 1 && bi.kcnt == nullptr && bi.ksplit == 0 && bi.par.ncls == 0 && !bi.xcd_z;
    const int kh = K / 2;
    if (mode >= 1 && n_cu == 256 && plain && bi.batch_major && bi.cols_per_batch > 0 &&
        tiles % 256 == 128 && tiles >= 384 && K % 2 == 0 && kh % CFG::BKT == 0 && kh >= 4 * CFG::BKT &&
        (CFG::KFOLD == 0 || kh % CFG::KFOLD == 0) && M % CFG::BM == 0 && bi.cols_per_batch % CFG::BN == 0) {
      // half a round of workgroups per CU left over (the 36 x 32 tiles of the F(4x4,3x3) GEMMs at the benchmark shape): the tiles
      // of the half round as two half-k workgroups each, one next to the four whole tiles of every CU (see the kernel)
      const int sx = 16, nsplit = 8 * sx;
      float* slab = sg_tail_scratch(s, (size_t)nsplit * 2 * PIECE_BYTES);
      int* cnt = slab ? sg_counter_alloc(s, nsplit, true) : nullptr;
      if (slab && cnt) { bi.tail_sx = sx; bi.tail_slab = slab; bi.tail_cnt = cnt; grid.x = tiles + nsplit; }
    } else if (sg_opt(SG_OPT_PAR_SPLIT) && n_cu > 0 && bi.par.ncls > 1 && grid.z == 1 && splits <= 1 && CFG::KFOLD == 0 && !bi.xcd_z &&
               M % CFG::BM == 0 && (CFG::BM * CFG::BN <= 64 * 64 || sg_opt(SG_OPT_PAR_SPLIT) >= 2)) {
      // (64x64 tiles only: on 128x128 tiles -- two resident workgroups per CU, 2.5 rounds -- the same split measured +10 %:
      //  187 -> 206 us, profiles/r06_gemm_par_split.md)
      // parity classes: split the heaviest class in two when it runs >= 4x the k range of the lightest and the launch is small
      // enough for its tail to matter (at most ~2 rounds of resident workgroups)
      int kmax = 0, kmin = 1 << 30;
      for (int c = 0; c < bi.par.ncls; ++c) { kmax = bi.par.K[c] > kmax ? bi.par.K[c] : kmax; kmin = bi.par.K[c] < kmin ? bi.par.K[c] : kmin; }
      const int rows = sg_cdiv(M, CFG::BM);
      int nsl = 0, old0[5];
      for (int c = 0; c <= bi.par.ncls; ++c) old0[c] = bi.par.tile0[c];
      bool any = false;
      for (int c = 0; c < bi.par.ncls; ++c) {
        const int tc = old0[c + 1] - old0[c];
        const bool sp2 = bi.par.K[c] == kmax && kmax >= 4 * kmin && (kmax / 2) % CFG::BKT == 0 && kmax / 2 >= sg_opt(SG_OPT_TAIL_KTMIN) * CFG::BKT;
        bi.par.split[c] = sp2 ? 2 : 1;
        bi.par.slot0[c] = nsl;
        if (sp2) { nsl += tc; any = true; }
      }
      const long wgs = (long)rows * (old0[bi.par.ncls] + nsl);
      if (any && wgs <= 2L * n_cu * 8 && (long)rows * nsl <= 4096) {
        float* slab = sg_tail_scratch(s, (size_t)rows * nsl * 2 * PIECE_BYTES);
        int* cnt = slab ? sg_counter_alloc(s, rows * nsl, true) : nullptr;
        if (slab && cnt) {
          bi.par.nslots = nsl;
          bi.par.tile0[0] = 0;
          for (int c = 0; c < bi.par.ncls; ++c) bi.par.tile0[c + 1] = bi.par.tile0[c] + (old0[c + 1] - old0[c]) * bi.par.split[c];
          for (int c = bi.par.ncls + 1; c < 5; ++c) bi.par.tile0[c] = bi.par.tile0[bi.par.ncls];
          bi.tail_slab = slab; bi.tail_cnt = cnt;
          grid.x = rows * bi.par.tile0[bi.par.ncls];
        } else any = false;
      } else any = false;
      if (!any) for (int c = 0; c < 4; ++c) { bi.par.split[c] = 1; bi.par.slot0[c] = 0; }
    } else if (mode >= 2 && n_cu > 0 && plain && bi.cols_per_batch == 0 && CFG::KFOLD == 0 && resident_wgs() > 0) {
      const int resident = resident_wgs();
      const int smax = sg_opt(SG_OPT_TAIL_SMAX) < 2 ? 2 : (sg_opt(SG_OPT_TAIL_SMAX) > 8 ? 8 : sg_opt(SG_OPT_TAIL_SMAX));
      const int kt = sg_cdiv(K, CFG::BKT), slots = n_cu * resident;
      int n = 0, sp = 0, first = 0;
      if (tiles <= slots) {                      // everything resident at once: r CUs carry one workgroup more than the others
        const int r = tiles % n_cu;
        if (r > 0 && tiles > n_cu) { n = r; sp = (n_cu + r / 2) / r; first = 1; }
      } else {                                   // the last round holds R of `slots` workgroups
        const int R = tiles % slots;
        if (R > 0 && 4 * R < 3 * slots) { n = R; sp = slots / R; first = 0; }
      }
      if (sp > smax) sp = smax;
      while (sp >= 2 && kt / sp < sg_opt(SG_OPT_TAIL_KTMIN)) --sp;      // pieces of at least this many k-tiles
      if (first && sp >= 2) {                    // pieces first: n * sp must be a multiple of 8 (the whole tiles keep their XCDs)
        int g = sp, h = 8;
        while (h) { const int t = g % h; g = h; h = t; }
        n -= n % (8 / g);
      }
      if (n > 0 && sp >= 2 && n <= 4096) {
        float* slab = sg_tail_scratch(s, (size_t)n * sp * PIECE_BYTES);
        int* cnt = slab ? sg_counter_alloc(s, n, true) : nullptr;
        if (slab && cnt) {
          bi.tail_n = n; bi.tail_s = sp; bi.tail_first = first; bi.tail_slab = slab; bi.tail_cnt = cnt;
          grid.x = tiles + n * (sp - 1);
        }
      }
    }
  }
  AL al2 = al; BL bl2 = bl; EP ep2 = ep;
  host_prepare(al2); host_prepare(bl2); host_prepare(ep2);
  for (int c = 0; c < bi.par.ncls && c < 4; ++c) {
    const FastDiv a((unsigned)(bi.par.PH[c] * bi.par.PW[c] > 0 ? bi.par.PH[c] * bi.par.PW[c] : 1)), b((unsigned)(bi.par.PW[c] > 0 ? bi.par.PW[c] : 1));
    bi.par.dphw_m[c] = a.m; bi.par.dphw_s[c] = a.s; bi.par.dpw_m[c] = b.m; bi.par.dpw_s[c] = b.s;
  }
  hipLaunchKernelGGL((igemm_kernel<CFG, AL, BL, EP>), grid, dim3(256), 0, s, al2, bl2, ep2, M, N, K, kchunk, bi);
  FILE* lf = g_sg_launch_log;
  if (lf) { fprintf(lf, "%u %u %d %d %d %.0f\n", grid.x * 256u, grid.z, M, N, K, sgk::t_alg_bytes); fflush(lf); }
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// sum split-K slabs: out[i] = sum_z ws[z*n + i]   (fixed order => deterministic)
__global__ void slab_reduce_kernel(const float* ws, float* out, size_t n, int S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = sg_sum_strided(ws + i, n, S);
}

// the same for MANY slabs of a SMALL result (weight gradient of a few-channel conv: 64 x 48 outputs, up to 256 k-chunks): a
// workgroup = 16 columns x 16 slab groups, every thread sums S / 16 consecutive slabs (loads issued eight at a time), the groups
// meet in LDS in ascending order.  One thread per column walking S dependent round trips took 16 us at S = 64.
__global__ void __launch_bounds__(256) slab_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n, int S) {
  __shared__ float red[16][17];
  const int j = threadIdx.x & 15, q = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + j;
  const int per = (S + 15) / 16, zb = q * per, ze = zb + per < S ? zb + per : S;
  float acc = 0.f;
  if (i < n) {
    for (int z0 = zb; z0 < ze; z0 += 8) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = z0 + e < ze ? ws[(size_t)(z0 + e) * n + i] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += t[e];
    }
  }
  red[q][j] = acc;
  __syncthreads();
  if (q == 0 && i < n) {
    float v = red[0][j];
#pragma unroll
    for (int g = 1; g < 16; ++g) v += red[g][j];
    out[i] = v;
  }
}

// split-K epilogue of the conv-shaped GEMMs: out[i] = act(sum_z ws[z][i] + bias[channel(i)])
__global__ void slab_reduce_nchw_kernel(const float* ws, float* out, size_t n, int S, const float* bias, int PHW, int Mtot,
                                        int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = sg_sum_strided(ws + i, n, S);           // (eight slab loads in flight, added in slab order)
  if (bias) v += bias[((unsigned)i / (unsigned)PHW) % (unsigned)Mtot];        // (n < 2^31: 32-bit divisions)
  out[i] = sg_apply_act(v, act, slope);
}
// float4 form (n % 4 == 0, PHW % 4 == 0: the four lanes of a vector share their channel)
__global__ void slab_reduce_nchw_vec_kernel(const float4* ws, float4* out, size_t n4, int S, const float* bias, int PHW4,
                                            int Mtot, int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = ws[i];
  for (int z0 = 1; z0 < S; z0 += 4) {               // four slab loads in flight, added in slab order (S <= 8)
    float4 t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = z0 + e < S ? ws[(size_t)(z0 + e) * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (z0 + e < S) { v.x += t[e].x; v.y += t[e].y; v.z += t[e].z; v.w += t[e].w; }
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

inline bool fixed_taps_enabled() {
  return sg_opt(SG_OPT_FIXEDTAP) != 0;
}

// ---- conv-shaped GEMM: K = (c, taps), N = pixels -------------------------------------------------
template <class CFG, int BM, int BN, int KS, int MODE>
int launch_ab(const float* A, int K, int M, bool vec, const Gather& g, int Npix, const KEntry* ktab, const EpNCHW& ep,
              int splits, bool nomask, hipStream_t s, const FixedTaps* fx = nullptr) {
  const bool two = g.C2 > 0;
  if (fx && vec && !two)          // whole channels per k-tile: taps fixed per thread, channel offset in an SGPR (LoadFixedKN)
    return launch_cfg<CFG>(LoadKContig<BM, true>{A, K, M}, LoadFixedKN<BN, MODE>{g, Npix, KS, fx->lg, fx->tapcode}, ep, M, Npix, K, splits, s);
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
  const int force = sg_opt(SG_OPT_SPLITS);
  if (force > 0) return force;
  const int tiles = kn_tiles(M, Npix);
  // workgroups wanted in flight: ~6 per CU of the 64x64 / 32x128 kernels, 3 per CU (the register limit) of 128x128
  const int target = pick_tile(M, Npix) == 0 ? 1024 : sg_opt(SG_OPT_SPLIT_TARGET);
  const int kmin = sg_opt(SG_OPT_SPLIT_KMIN);
  if (tiles * 4 >= target * 3 || K < kmin) return 1;
  int sp = (target + tiles / 2) / tiles;
  if (sp > 2 && (sp & 1)) ++sp;                 // odd split counts measured 10 % slower than their even neighbours
  if (sp > K / (kmin / 2)) sp = K / (kmin / 2);
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
  // 4x4 and 1x1 kernels on one source with whole channels per k-tile: fixed taps per thread (LoadFixedKN)
  const FixedTaps fixed{KS == 4 ? 4 : 0, 0u};
  const bool use_fixed = fixed_taps_enabled() && (KS == 4 || KS == 1) && vec && g.C2 == 0 && K % BK == 0 && vstride == 0;
  const FixedTaps* fx = use_fixed ? &fixed : nullptr;
  {
    SgProfScope prof(sg_igemm_kind(MODE, KS, tile), s, flops, 0);
    switch (tile) {
      case 0: launch_ab<typename CfgFor<KS>::C128, 128, 128, KS, MODE>(A, K, M, true, g, Npix, ktab, ep, splits, nomask, s, fx); break;
      case 1: launch_ab<typename CfgFor<KS>::C64, 64, 64, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s, fx); break;
      case 3: launch_ab<typename CfgFor<KS>::C64W, 64, 128, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s, fx); break;
      default: launch_ab<typename CfgFor<KS>::C32, 32, 128, KS, MODE>(A, K, M, vec, g, Npix, ktab, ep, splits, nomask, s, fx); break;
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

inline int sparse_kc(int L, int KS2) { return sg_cdiv(L * KS2, BK) * BK; }
inline int sparse_kpad(int L, int KS2) { return sg_cdiv(sparse_kc(L, KS2), 64) * 64 + 128; }
inline size_t sparse_fwd_ws(int NB, int M, int L, int KS2) {
  return (size_t)NB * ((size_t)sparse_kpad(L, KS2) * sizeof(KEntry) + (size_t)M * sparse_kc(L, KS2) * sizeof(float) + 64);
}
inline size_t parity_ws(int M, int Rdim, int KS2) { return (size_t)M * Rdim * KS2 * sizeof(float) + 64; }
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
  const int BMt = p.tile == 0 ? 128 : ((p.tile == 1 || p.tile == 3) ? 64 : 32), BNt = p.tile == 1 ? 64 : 128;
  p.cpad = p.tap ? sg_cdiv(C, BNt) * BNt : 0;
  const long tiles = (long)sg_cdiv(M, BMt) * (p.tap ? (long)KS2 * (p.cpad / BNt) : (long)sg_cdiv((long)C * KS2, BNt));
  const int target = p.tile == 0 ? 768 : 1024;       // resident workgroups on 256 CUs
  int s = (int)((target + tiles - 1) / tiles);
  const int maxs = Kpix / (BK * 8) > 0 ? Kpix / (BK * 8) : 1;
  if (s > maxs) s = maxs;
  // (tap-major launches have many tiles; the few-channel (c, tap) form has 1..4 of them and only its k-chunks to fill the chip
  // with: 64 chunks left the first conv of the crop discriminator on 64 workgroups for 93 us)
  const int cap = p.tap ? 64 : 256;
  if (s > cap) s = cap;
  // XCD-pinned k-chunks (see the kernel) need a multiple of 8 of them
  if (sg_opt(SG_OPT_WGRAD_XCD) && s >= 6) {
    int s8 = (s + 7) / 8 * 8;
    if (s8 > maxs) s8 = maxs / 8 * 8;
    if (s8 >= 8) s = s8;
  }
  p.splits = s < 1 ? 1 : s;
  return p;
}

}  // namespace
