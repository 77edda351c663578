// transposed gathers: conv data gradient, transposed-conv forward; stride-2 parity classes (see igemm_core.h / igemm.hip)
#include "igemm_core.h"

namespace {
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
  // (32-bit division whenever the packed matrices hold fewer than 2^32 elements -- always, in practice: the emulated 64-bit one
  //  was most of this kernel's 6.5 us)
  const int m = pc.off[pc.ncls] <= 0xffffffffull ? (int)((unsigned)j / (unsigned)K) : (int)(j / K);
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
  bool fixed_ok = fixed_taps_enabled() && KS <= 4;      // every class has 1, 2 or 4 taps and whole channels per k-tile
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
      par.lg[c] = tl.n == 1 ? 0 : (tl.n == 2 ? 1 : (tl.n == 4 ? 2 : -1));
      par.tapcode[c] = 0u;
      for (int q = 0; q < tl.n && q < 4; ++q) par.tapcode[c] |= (unsigned)tl.t[q] << (4 * q);
      fixed_ok = fixed_ok && par.lg[c] >= 0 && K % BK == 0;
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
  // no split-K here.  768: measured with the fixed-tap loaders (tools/conv_sweep.py) -- the dgrad of Conv2d(256, 512, 3, s2) at
  // 32x32 (512 tiles of 128x128) runs 0.213 ms on 64x64 tiles against 0.275 ms; from 1024 tiles on the two are level
  if (tile == 0 && (!vec || t128 < 768)) tile = 1;
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
  const FixedTaps fixed{par.lg[0], par.tapcode[0]};
  const FixedTaps* fx = (fixed_ok && vec) ? &fixed : nullptr;
  {
    SgProfScope prof(sg_igemm_kind(1, KS, tile), s, flops_issued, 0);
    switch (tile) {
      case 0: launch_ab<typename CfgFor<KS>::C128, 128, 128, KS, 1>(wbase, par.K[0], M, true, gs, par.Npix[0], kt0, ep, 1, false, s, fx); break;
      case 1: launch_ab<typename CfgFor<KS>::C64, 64, 64, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s, fx); break;
      case 3: launch_ab<typename CfgFor<KS>::C64W, 64, 128, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s, fx); break;
      default: launch_ab<typename CfgFor<KS>::C32, 32, 128, KS, 1>(wbase, par.K[0], M, vec, gs, par.Npix[0], kt0, ep, 1, false, s, fx); break;
    }
  }
  t_batch = BatchInfo{};
  return 0;
}
int run_kn_parity_ks(int KS, const float* W, int Rdim, int B, int m0, int M, const Gather& g, int NB, const float* bias,
                     float* out, int Mtot, int act, float slope, double flops, void* ws, size_t ws_bytes, hipStream_t s) {
  switch (KS) {
    case 3: return run_kn_parity<3>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
    case 4: return run_kn_parity<4>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
    case 7: return run_kn_parity<7>(W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
  }
  return -1;
}

}  // namespace

int sgk::kn1_run(int KS, const float* A, int M, int K, const Gather& g, int NB, const float* bias, float* out, int Mtot, int act,
                 float slope, double flops, void* ktab_ws, size_t ws_avail, unsigned variant_stride, hipStream_t s) {
  t_variant_stride = variant_stride;
  const int rc = run_kn_ks<1>(KS, A, M, K, g, NB, bias, out, Mtot, act, slope, flops, ktab_ws, ws_avail, s);
  t_variant_stride = 0;
  return rc;
}
int sgk::kn_parity_run(int KS, const float* W, int Rdim, int B, int m0, int M, const Gather& g, int NB, const float* bias,
                       float* out, int Mtot, int act, float slope, double flops, void* ws, size_t ws_bytes, hipStream_t s) {
  return run_kn_parity_ks(KS, W, Rdim, B, m0, M, g, NB, bias, out, Mtot, act, slope, flops, ws, ws_bytes, s);
}

#ifdef SG_TIMELINE
// debugging build only (tools/probe/build_timeline.sh): hand this translation unit's igemm_kernel instantiations a stamp buffer of
// ``cap`` workgroups x 8 x u64 (nullptr: off)
extern "C" int sg_debug_timeline_set_igemm_kn1(void* buf, unsigned cap) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl), &p, sizeof(p)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
  return 0;
}
#endif
