// Shared helpers for the gfx950 kernels of libsg2im_hip.so (see include/sg2im_hip.h for the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/sg2im_hip.h"

#define SG_WAVE 64

void sg_set_error(const char* fmt, ...);
// range check of an index operand when the option check_indices is on (graph.hip); 0 or SG_ERR_INDEX
int sg_check_indices_if_enabled(const int64_t* idx, int64_t n, int64_t lo, int64_t hi, const char* what, hipStream_t s);

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

// ---- tuning / debugging switches (include/sg2im_hip.h: sg_set_option) -------------------------------------------
// One table (runtime.hip), initialised ONCE when the library is loaded: compiled-in default, overridden by the environment
// variable SG_<NAME> if set.  After that the library never reads the environment; sg_set_option() stores atomically.  (Until
// round 4 every switch was an environment lookup cached in an unsynchronised function-local static: a data race on first use from two
// threads, and state the header did not admit to.)
enum SgOpt {
  SG_OPT_TILE,            // force the tile of the gather GEMMs: 0 128x128, 1 64x64, 2 32x128, 3 64x128; -1 = chosen per shape
  SG_OPT_T128_MIN,        // 128x128 tiles from this many tiles on
  SG_OPT_TILE3,           // allow 64x128 tiles
  SG_OPT_TILE3_MIN,       // ... from this many tiles on
  SG_OPT_SPLIT_TARGET,    // conv-shaped GEMMs on 64-row tiles: split K until about this many workgroups are in flight
  SG_OPT_SPLIT_KMIN,      // ... but only reductions at least this long (chunks stay >= half of it)
  SG_OPT_SPLITS,          // force the split-K count of the conv-shaped GEMMs; -1 = chosen per shape
  SG_OPT_FIXEDTAP,        // LoadFixedKN loaders (taps in VGPRs, channel offset in the SGPR operand)
  SG_OPT_WINO_WT,         // LDS-staged Winograd filter transform (0: one thread per filter)
  SG_OPT_W24_SMALL,       // LDS-staged F(2x2,4x4) input transform
  SG_OPT_W24_S,           // force the k-chunk count of the F(2x2,4x4) weight gradient; -1 = chosen per shape
  SG_OPT_W24_PMIN,        // fewest tiles for which F(2x2,4x4) is used
  SG_OPT_WINO_ADJOINT,    // adjoint-form data gradient of the reflection-padded Winograd convs
  SG_OPT_WINO24,          // F(2x2,4x4) for the stride-1 4x4 convs
  SG_OPT_LINEAR_NSUB,     // k-tile depth (x16) of the LDS-tiled dense layers
  SG_OPT_LINEAR_SKINNY,   // dense layers with at most this many 32x32 output tiles run the register-streaming kernel (skinny.hip)
  SG_OPT_WGRAD_ROWSUM,    // bias gradient from the weight-gradient GEMM's A loader
  SG_OPT_LAYOUT_REG,      // register-resident masks_to_layout forward
  SG_OPT_LAYOUT_DSPLIT,   // channel chunks per masks_to_layout tile
  SG_OPT_BN_BLOCKS,       // workgroups per BatchNorm statistics launch
  SG_OPT_INSTNORM_REG,    // register-resident InstanceNorm
  SG_OPT_WGRAD_XCD,       // weight-gradient GEMMs: k-chunks pinned to XCDs (their tiles share the chunk's pixels in ONE L2)
  SG_OPT_WINO_REUSE,      // Winograd weight gradient from the forward's V and the data gradient's Ytp (no second transforms)
  SG_OPT_WINO_FOLD_CELLS, // four-wave cell-gather form of the adjoint Winograd output fold (0: one thread per channel walks the tiles)
  SG_OPT_WINO_PIPE,       // main loop of the dense Winograd GEMMs: 1 = stores at the top of the iteration, 2 = interleaved with phase 0
  SG_OPT_CHECK_INDICES,   // debugging: range-check index operands on the device (one stream synchronisation per check); sg_check_indices
  SG_OPT_LAST_BLOCK,      // two-stage reductions finished by the last workgroup to arrive (one launch) instead of a second kernel;
                          // OFF by default: measured on MI355X (profiles/r05_ab_sessions.md) the ticket + write-through hand-off in
                          // thousands of short workgroups costs more than the 4-6 us final kernels it removes (35.04 vs 34.94 ms)
  SG_OPT_WINO_GEMM_TILE,  // tile of the K-contiguous batched Winograd GEMMs: 0 = 128x128, 1 = 64x128, 2 = 64x64
  SG_OPT_WINO43,          // Winograd F(4x4,3x3) for the small-plane reflection-padded ResnetBlock convs (0: F(2x2,3x3))
  SG_OPT_GCONV_FUSED_GATHER, // GraphTripleConv: the (s, p, o) row gather inside the first MLP layer's A loader (0: materialise cur_t)
  SG_OPT_W24_GEMM_TILE,   // tile of the F(2x2,4x4) GEMMs (K = 256 / 512): 2 = 64x64 (measured 2.03 vs 2.14 ms/step on 128x128), 0, 1
  SG_OPT_WINO_IN_FUSE,    // F(4x4,3x3) conv + InstanceNorm: output transform + norm in one launch, norm backward + gradient transform in one
  SG_OPT_W43_NSUB,        // k-tile depth (x16) of the F(4x4,3x3) GEMMs: 1 = 16-deep (default: 20 KB of LDS, 7 workgroups per CU; measured +0.5 % on the step), 2 = 32-deep
  SG_OPT_W43_KFOLD,       // F(4x4,3x3) forward / data-gradient GEMMs: accumulate the channel sum in chunks of this many k (256 / 128; 0 = one fma chain): see TileCfg::KFOLD
  SG_OPT_WAVE_PRIO,       // GEMM kernels: s_setprio 3 outside the main loop (prologue / epilogue VALU work does not queue behind other waves' MFMAs).
                          // OFF: measured on MI355X, profiles/r06_gemm_prio.md -- most classes +-1 %, Gup4 fwd +11 %, F(4x4,3x3) dgrad / wgrad +7 %, step -0.3 %
  SG_OPT_PAR_XCD_CHUNK,   // parity-class launches: tiles dealt to the XCDs in chunks of this many (power of two; 0 = one contiguous eighth per XCD)
  SG_OPT_W43_TAIL_SPLIT,  // tail split (igemm_kernel): 0 = off; 1 = F(4x4,3x3) forward / data-gradient GEMMs whose tile count leaves half a round per CU (36 x 32 tiles on 256 CUs); 2 = also the general form (plain conv GEMM launches with a ragged last round):
                          // the tiles of the half round run as two workgroups of half the k range each (igemm_kernel, TileCfg::TAILSPLIT)
  SG_OPT_TAIL_SMAX,       // tail split, general form: at most this many pieces per tile (2..8)
  SG_OPT_TAIL_KTMIN,      // ... and no piece shorter than this many k-tiles
  SG_OPT_W43_WGRAD_TILE,  // tile of the F(4x4,3x3) weight-gradient GEMMs (K = the 128 Winograd tiles): 0 = 64x64 (9216 workgroups), 1 = 128x128, 2 = 64x128
  SG_OPT_PAR_SPLIT,       // parity-class launches (3x3 stride-2 transposed gathers: classes of 4 / 2 / 2 / 1 taps): the tiles of the 4-tap class run as two half-k workgroups that meet through the tail-split ticket (0: off)
  SG_OPT_TAIL_CAPTURE,    // tail-split / parity-split schedules inside a hipGraph capture (shared per-device scratch, runtime.hip): 1 on, 0 = captured launches run the unsplit form (rounds 1-5 behaviour)
  SG_OPT_COUNT
};
extern std::atomic<int> g_sg_opt[SG_OPT_COUNT];
static inline int sg_opt(int id) { return g_sg_opt[id].load(std::memory_order_relaxed); }
extern FILE* g_sg_launch_log;      // SG_LAUNCH_LOG=<file> at load time: one line per GEMM launch (tools/pmc_db_summary.py)

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
  SG_K_WINO43_GEMM,   // the 36 batched dense GEMMs of a Winograd F(4x4,3x3) conv (64x64 tiles)
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

// sum_{z < S} p[z * stride] in ascending z -- the value (and rounding) of the plain loop, with the loads issued eight at a time:
// a lone thread adding S partials one dependent round trip after the other is ~1.5 us per partial when few workgroups run
__device__ __forceinline__ float sg_sum_strided(const float* __restrict__ p, size_t stride, int S) {
  float v = 0.f;
  for (int z0 = 0; z0 < S; z0 += 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = z0 + e < S ? p[(size_t)(z0 + e) * stride] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (z0 + e < S) v += t[e];
  }
  return v;
}

// ---- in-launch finalisation of two-stage reductions ("last block") ----------------------------------------------------------
// Stage 1 workgroups publish their partials with WRITE-THROUGH (sc1) stores, every wave drains its stores, one lane takes a
// ticket from an agent-scope counter; the workgroup that draws the last ticket reads all partials with sc1 loads (L1 is never
// refreshed by other CUs' stores, per-XCD L2s are not coherent: cdna_hip_programming.md 6 G16, form R1 with sc1 loads) and
// finishes the reduction in the SAME fixed order the separate final kernel used -- results are bit-identical, a ~4-6 us launch
// and its ~1.7 us boundary are gone.  The counter comes from sg_counter_alloc() (zero-initialised library memory) and is reset
// by the last arriver, so it is clean for the next launch that is handed it, also under hipGraph replay.
__device__ __forceinline__ void sg_publish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sg_consume(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Called by ALL threads of the workgroup after its sg_publish() calls.  True (in every thread) in the workgroup that arrived last
// of ``narrive``.  ``flag``: one int of LDS.
__device__ __forceinline__ bool sg_arrive_last(int* counter, int narrive, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its sc1 stores
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == narrive - 1;
    if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
// a zeroed 4-byte counter in library-owned device memory for ONE launch on stream ``s`` (runtime.hip); nullptr: none available
// (the caller then runs its two-kernel form).  ``n`` consecutive counters.
int* sg_counter_alloc(hipStream_t s, int n = 1, bool always = false, int family = 1);     // always: also when the option last_block is off;
// family: bit of the option last_block that enables the caller (1 = losses, 2 = BatchNorm statistics, 4 = channel sums)
// library-owned device scratch of the tail-split GEMM launches (raw accumulator dumps), one buffer per (device, stream), grown
// outside stream captures only; nullptr: not available (the caller launches the plain schedule)
float* sg_tail_scratch(hipStream_t s, size_t bytes);

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
