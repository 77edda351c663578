// weight-gradient GEMMs: K = (image, pixel), N = (channel, tap) (see igemm_core.h / igemm.hip)
#include "igemm_core.h"

namespace {
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
                                              int cpad, int S, const float* __restrict__ rowsum, float* __restrict__ gb) {
  // grid (ceil(C*KS2 / 256), M): one 32-bit division per thread.  (An LDS-transposed variant with fully coalesced slab reads
  // was measured SLOWER -- 21.7 vs 15.7 us per launch: the strided reads hit in L2, the extra barrier and the thinner loops
  // do not pay.)
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (gb && j == 0) {                     // bias gradient: the k-chunks' row sums of gy (LoadPixK*::rowsum), fixed order
    const float b = sg_sum_strided(rowsum + m, (size_t)M, S);
    gb[m] = b;
  }
  if (j >= (unsigned)(C * KS2)) return;
  const unsigned c = j / (unsigned)KS2, t = j - c * (unsigned)KS2;
  const size_t zs = (size_t)M * KS2 * cpad;
  const float* p = slabs + ((size_t)m * KS2 + t) * cpad + c;
  // eight slab loads in flight per thread, added in slab order (sg_sum_strided; the plain loop compiled to one dependent load /
  // wait / add per slab: 95 % of the wave cycles parked in s_waitcnt on 8..64 L2 round trips, profiles/r06_pmc_classes_cycles.md)
  const float v = sg_sum_strided(p, zs, S);
  gw[(size_t)m * C * KS2 + j] = v;
}
// general (c, tap)-ordered loader: only for few-channel inputs (RGB crops / images)
template <int KS>
void launch_nk_general(int tile, const float* A, int M, int Mtot, int PQ, const Gather& g, int Ncols, const EpRowMajor& ep,
                       int Kpix, int splits, hipStream_t s, const Sparse* sp = nullptr, int zdiv = 1) {
  float* const rowsum = nullptr;
  const FastDiv dPQ((unsigned)PQ);
  const int* sl = sp ? sp->list : nullptr; const int* sc = sp ? sp->cnt : nullptr; const int L = sp ? sp->L : 0;
  if (tile == 2)
    launch_cfg<CfgW32>(LoadPixK<32>{A, M, Mtot, PQ, dPQ, rowsum}, LoadGatherNK<128, KS, false, true, NSW>{g, Ncols, sl, sc, L, zdiv}, ep,
                       M, Ncols, Kpix, splits, s);
  else if (KS == 7 && sp && (PQ % 4 == 0) && aligned16(A))
    // the per-image weight gradient of the factored 7x7 stem (64 x 9*49 x 16 384 per image: the slowest GEMM launch of the step,
    // 45 TFLOP/s): the output gradient is read with 16-byte loads -- one vector-memory instruction per thread and k-tile instead of
    // four next to the four gathered elements of B (same-box A/B, two pairs: 31.24 vs 31.31 ms/step, +0.2 %)
    launch_cfg<CfgW64>(LoadPixKVec<64>{A, M, Mtot, PQ, dPQ, rowsum}, LoadGatherNK<64, KS, false, true, NSW>{g, Ncols, sl, sc, L, zdiv}, ep,
                       M, Ncols, Kpix, splits, s);
  else
    launch_cfg<CfgW64>(LoadPixK<64>{A, M, Mtot, PQ, dPQ, rowsum}, LoadGatherNK<64, KS, false, true, NSW>{g, Ncols, sl, sc, L, zdiv}, ep,
                       M, Ncols, Kpix, splits, s);
}

template <class CFG, int BMv, int BNv>
void launch_nk_tap(const float* A, int M, int Mtot, int PQ, bool vecA, const Gather& g, int KS, int Ccols, int cpad,
                   const Sparse* sp, bool nomask, const EpWgrad& ep, int Kpix, int splits, hipStream_t s, float* rowsum) {
  const FastDiv dPQ((unsigned)PQ), dPW((unsigned)g.PW);
  const int* sl = sp ? sp->list : nullptr; const int* sc = sp ? sp->cnt : nullptr; const int L = sp ? sp->L : 0;
  const int Nv = KS * KS * cpad;
  const bool two = g.C2 > 0;
#define SG_TAP_B(TWOv, MASKv) LoadTapNK<BNv, TWOv, MASKv, CFG::NSUB>{g, KS, Ccols, cpad, sl, sc, L, dPQ, dPW}
  if (vecA) {
    const LoadPixKVec<BMv> al{A, M, Mtot, PQ, dPQ, rowsum};
    if (two) launch_cfg<CFG>(al, SG_TAP_B(true, true), ep, M, Nv, Kpix, splits, s);
    else if (nomask) launch_cfg<CFG>(al, SG_TAP_B(false, false), ep, M, Nv, Kpix, splits, s);
    else launch_cfg<CFG>(al, SG_TAP_B(false, true), ep, M, Nv, Kpix, splits, s);
  } else {
    const LoadPixK<BMv> al{A, M, Mtot, PQ, dPQ, rowsum};
    if (two) launch_cfg<CFG>(al, SG_TAP_B(true, true), ep, M, Nv, Kpix, splits, s);
    else if (nomask) launch_cfg<CFG>(al, SG_TAP_B(false, false), ep, M, Nv, Kpix, splits, s);
    else launch_cfg<CFG>(al, SG_TAP_B(false, true), ep, M, Nv, Kpix, splits, s);
  }
#undef SG_TAP_B
}

int run_nk_ks(int KS, const float* A, int M, int Mtot, const Gather& g, int NB, float* out, void* ws, size_t ws_bytes,
              double flops, hipStream_t s, const Sparse* sp, float* gb, bool* gb_done) {
  if (gb_done) *gb_done = false;
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
    t_xcd_z = sg_opt(SG_OPT_WGRAD_XCD) ? 2 : 0;      // (image, chunk) slices pinned to XCDs when grid.z % 8 == 0 (launch_cfg)
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
    t_xcd_z = 0;
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
  // XCD-pinned k-chunks (igemm_kernel: xcd_z) need grid.z % 8 == 0: pad with empty chunks (kbeg >= Kpix: they store zero slabs)
  // when the workspace has room for them
  int zpad = 0;
  if (sg_opt(SG_OPT_WGRAD_XCD) && !sp && splits >= 6) {
    const int z8 = (splits + 7) / 8 * 8;
    if (ws && ws_bytes >= (mn + (size_t)M) * sizeof(float) * (size_t)z8) { zpad = z8; splits = z8; }
  }
  // bias gradient from the A loader's row sums: tap-major dense launches whose workspace has room for [splits][M] behind the slabs
  const int fuse_gb = sg_opt(SG_OPT_WGRAD_ROWSUM);
  float* rowsum = nullptr;
  if (fuse_gb && gb && pl.tap && !sp && M == Mtot && ws_bytes >= (mn + (size_t)M) * sizeof(float) * (size_t)splits)
    rowsum = reinterpret_cast<float*>(ws) + mn * (size_t)splits;
  if (sp) { t_fixed_kchunk = PQ; flops = 2.0 * M * (double)Ncols * Kpix; }
  t_xcd_z = zpad > 0 ? 1 : 0;
  t_min_z = zpad;
  float* dst = (splits > 1 || sp || pl.tap) ? reinterpret_cast<float*>(ws) : out;
  {
    SgProfScope prof(sg_igemm_kind(2, KS, pl.tile), s, flops, 0);
    if (pl.tap) {
      const EpWgrad ep{dst, M, pl.cpad, KS2, mn};
      const bool vecA = (PQ % 4 == 0) && aligned16(A);
      // mask-free gather: reflection padding and whole 16-pixel k-tiles (split chunks are multiples of 64)
      const bool nomask = g.reflect && (Kpix % (BK * NSW) == 0) && (!sp || PQ % (BK * NSW) == 0);
      switch (pl.tile) {
        case 0: launch_nk_tap<CfgW128, 128, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s, rowsum); break;
        case 1: launch_nk_tap<CfgW64, 64, 64>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s, rowsum); break;
        case 3: launch_nk_tap<CfgW64W, 64, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s, rowsum); break;
        default: launch_nk_tap<CfgW32, 32, 128>(A, M, Mtot, PQ, vecA, g, KS, Ccols, pl.cpad, sp, nomask, ep, Kpix, splits, s, rowsum); break;
      }
    } else {
      const EpRowMajor ep{dst, nullptr, M, Ncols, Ncols, SG_ACT_NONE, 0.f, mn};
      switch (KS) {
        case 1: launch_nk_general<1>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 3: launch_nk_general<3>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 4: launch_nk_general<4>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        case 7: launch_nk_general<7>(pl.tile, A, M, Mtot, PQ, g, Ncols, ep, Kpix, splits, s); break;
        default: t_fixed_kchunk = 0; t_xcd_z = 0; t_min_z = 0; return -1;
      }
    }
  }
  t_fixed_kchunk = 0;
  t_xcd_z = 0;
  t_min_z = 0;
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
  {
    hipLaunchKernelGGL(wgrad_unpermute_reduce_kernel, dim3(sg_cdiv((size_t)C * KS2, 256), M), dim3(256), 0, s, (const float*)ws, out,
                       M, C, KS2, pl.cpad, splits, (const float*)rowsum, rowsum ? gb : nullptr);
    if (rowsum && gb_done) *gb_done = true;
  }
  else if (splits >= 32)
    hipLaunchKernelGGL(slab_reduce_wide_kernel, dim3(sg_cdiv(mn, 16)), dim3(256), 0, s, (const float*)ws, out, mn, splits);
  else if (splits > 1)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(sg_cdiv(mn, 256)), dim3(256), 0, s, (const float*)ws, out, mn, splits);
  return 0;
}


}  // namespace

int sgk::nk_run(int KS, const float* A, int M, int Mtot, const Gather& g, int NB, float* out, void* ws, size_t ws_bytes,
                double flops, hipStream_t s, const Sparse* sp, float* gb, bool* gb_done) {
  return run_nk_ks(KS, A, M, Mtot, g, NB, out, ws, ws_bytes, flops, s, sp, gb, gb_done);
}

#ifdef SG_TIMELINE
// debugging build only (tools/probe/build_timeline.sh): hand this translation unit's igemm_kernel instantiations a stamp buffer of
// ``cap`` workgroups x 8 x u64 (nullptr: off)
extern "C" int sg_debug_timeline_set_igemm_nk(void* buf, unsigned cap) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl), &p, sizeof(p)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
  return 0;
}
#endif
