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
#include "igemm_core.h"

namespace sgk {
thread_local double t_alg_bytes = 0.0;
std::mutex g_tab_mu;
std::map<TabKey, TabEntry> g_tabs;
size_t g_tab_bytes = 0;
}  // namespace sgk

namespace {
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
  // (+ M per k-chunk: the row sums that become the bias gradient, see nk_run)
  return (size_t)pl.splits * ((size_t)M * KS2 * (pl.tap ? pl.cpad : C) + (size_t)M) * sizeof(float);
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
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1 + d->C2) * d->H * d->W + (double)d->Cout * (d->C1 + d->C2) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(x1 && w && y && ws, "sg_conv2d_fwd: null pointer");
  SG_ARG_CHECK(ws_bytes >= ktab_bytes((d->C1 + d->C2) * d->KS * d->KS), "sg_conv2d_fwd: workspace too small");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd: C2>0 but x2 null");
  hipStream_t s = (hipStream_t)stream;
  const int Cin = d->C1 + d->C2, K = Cin * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const double flops = 2.0 * d->Cout * K * (double)d->N * d->OH * d->OW;
  int rc = sgk::kn0_run(d->KS, w, d->Cout, K, g, d->N, bias, y, d->Cout, act, slope, flops, ws, ws_bytes, s);
  SG_LAUNCH_CHECK("sg_conv2d_fwd");
  return rc;
}

extern "C" int sg_conv2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, int c_begin, int c_end,
                               void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_dgrad")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1 + d->C2) * d->H * d->W + (double)d->Cout * (d->C1 + d->C2) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
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
    int rc = sgk::kn_parity_run(d->KS, w, d->Cout, Cin, c_begin, M, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops, ws,
                              ws_bytes, s);
    SG_LAUNCH_CHECK("sg_conv2d_dgrad");
    return rc;
  }
  float* wt = reinterpret_cast<float*>(ws);      // [Cin][Cout][R]
  const size_t nw = (size_t)d->Cout * Cin * R;
  hipLaunchKernelGGL(permute_w_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, w, wt, d->Cout, Cin, R);
  int rc = sgk::kn1_run(d->KS, wt + (size_t)c_begin * K, M, K, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops,
                        wt + nw, ws_bytes - nw * sizeof(float), 0u, s);
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
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1 + d->C2) * d->H * d->W + (double)d->Cout * (d->C1 + d->C2) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
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
  int rc = sgk::kn1_run(3, wt + (size_t)c_begin * K, M, K, g, d->N, nullptr, gx, M, SG_ACT_NONE, 0.f, flops, rest,
                        ws_bytes - (size_t)(rest - reinterpret_cast<char*>(ws)), (unsigned)VS, s);
  SG_LAUNCH_CHECK("sg_conv2d_dgrad_folded");
  return rc;
}

extern "C" int sg_conv2d_wgrad(const sgConvDesc* d, const float* gy, const float* x1, const float* x2, float* gw,
                               float* gb, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1 + d->C2) * d->H * d->W + (double)d->Cout * (d->C1 + d->C2) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(gy && x1 && gw, "sg_conv2d_wgrad: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad: C2>0 but x2 null");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const double flops = 2.0 * d->Cout * (double)(d->C1 + d->C2) * d->KS * d->KS * d->N * d->OH * d->OW;
  bool gb_done = false;      // the weight-gradient GEMM reads all of gy anyway: its loaders also produce the bias gradient
  if (int rc = sgk::nk_run(d->KS, gy, d->Cout, d->Cout, g, d->N, gw, ws, ws ? ws_bytes : 0, flops, s, nullptr, gb, &gb_done)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad");
  if (gb && !gb_done) return sg_channel_sum(gy, gb, d->N, d->Cout, d->OH * d->OW, ws, ws ? ws_bytes : 0, stream);
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
  sgk::t_alg_bytes = 4.0 * ((double)d->N * L * d->H * d->W + (double)d->N * d->Cout * L * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(x1 && w && y && ws, "sg_conv2d_fwd_sparse: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd_sparse: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 0), "sg_conv2d_fwd_sparse: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int K = (d->C1 + d->C2) * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, nullptr};
  int rc = sgk::kn_sparse_run(d->KS, w, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s);
  SG_LAUNCH_CHECK("sg_conv2d_fwd_sparse");
  return rc;
}
extern "C" int sg_conv2d_wgrad_sparse(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                                      const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gw, float* gb,
                                      void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad_sparse") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_wgrad_sparse")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * L * d->H * d->W + (double)d->N * d->Cout * L * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(gy && x1 && gw && ws, "sg_conv2d_wgrad_sparse: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad_sparse: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 2), "sg_conv2d_wgrad_sparse: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, nullptr};
  if (int rc = sgk::nk_run(d->KS, gy, d->Cout, d->Cout, g, d->N, gw, ws, ws_bytes, 0.0, s, &sp)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad_sparse");
  if (gb) return sg_channel_sum(gy, gb, d->N, d->Cout, d->OH * d->OW, ws, ws_bytes, stream);
  return 0;
}
// ---- per-image weights (factored layout convs) ---------------------------------------------------------------
extern "C" int sg_conv2d_fwd_perimage(const sgConvDesc* d, const float* x1, const float* x2, const float* wimg,
                                      const float* bias, const int32_t* chan_list, const int32_t* chan_cnt, int L, float* y,
                                      int act, float slope, void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_fwd_perimage") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_fwd_perimage")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * L * d->H * d->W + (double)d->N * d->Cout * L * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(x1 && wimg && y && ws, "sg_conv2d_fwd_perimage: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_fwd_perimage: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 0), "sg_conv2d_fwd_perimage: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int K = (d->C1 + d->C2) * d->KS * d->KS;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, wimg, nullptr};
  int rc = sgk::kn_sparse_run(d->KS, nullptr, d->Cout, K, g, d->N, bias, y, act, slope, sp, ws, s);
  SG_LAUNCH_CHECK("sg_conv2d_fwd_perimage");
  return rc;
}
extern "C" int sg_conv2d_wgrad_perimage(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                                        const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gwimg, void* ws,
                                        size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_conv2d_wgrad_perimage") || check_sparse(d, chan_list, chan_cnt, L, "sg_conv2d_wgrad_perimage")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * L * d->H * d->W + (double)d->N * d->Cout * L * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(gy && x1 && gwimg && ws, "sg_conv2d_wgrad_perimage: null pointer");
  SG_ARG_CHECK(d->C2 == 0 || x2, "sg_conv2d_wgrad_perimage: C2>0 but x2 null");
  SG_ARG_CHECK(ws_bytes >= sg_conv2d_sparse_ws_bytes(d, L, 2), "sg_conv2d_wgrad_perimage: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(x1, x2, d->C1, d->C2, d->H, d->W, d->upsample, d->OH, d->OW, d->stride, d->pad, d->pad_reflect);
  g.bcast2 = d->x2_broadcast;
  const Sparse sp{chan_list, chan_cnt, L, nullptr, gwimg};
  if (int rc = sgk::nk_run(d->KS, gy, d->Cout, d->Cout, g, d->N, nullptr, ws, ws_bytes, 0.0, s, &sp)) return rc;
  SG_LAUNCH_CHECK("sg_conv2d_wgrad_perimage");
  return 0;
}

extern "C" int sg_convT2d_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                              void* ws, size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_convT2d_fwd")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1) * d->H * d->W + (double)d->Cout * (d->C1) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
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
    int rc = sgk::kn_parity_run(d->KS, w, Cin, d->Cout, 0, d->Cout, g, d->N, bias, y, d->Cout, SG_ACT_NONE, 0.f, flops, ws,
                              ws_bytes, s);
    SG_LAUNCH_CHECK("sg_convT2d_fwd");
    return rc;
  }
  float* wt = reinterpret_cast<float*>(ws);      // [Cout][Cin][R]
  const size_t nw = (size_t)d->Cout * Cin * R;
  hipLaunchKernelGGL(permute_w_kernel, dim3(sg_cdiv(nw, 256)), dim3(256), 0, s, w, wt, Cin, d->Cout, R);
  int rc = sgk::kn1_run(d->KS, wt, d->Cout, Cin * R, g, d->N, bias, y, d->Cout, SG_ACT_NONE, 0.f, flops, wt + nw,
                        ws_bytes - nw * sizeof(float), 0u, s);
  SG_LAUNCH_CHECK("sg_convT2d_fwd");
  return rc;
}

// gx[n,ci,ih,iw] = sum_{co,kh,kw} w[ci,co,kh,kw] gy[n,co,ih*s-p+kh,iw*s-p+kw]   (a plain strided conv over gy)
extern "C" int sg_convT2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                                sgStream stream) {
  if (check_desc(d, "sg_convT2d_dgrad")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1) * d->H * d->W + (double)d->Cout * (d->C1) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(gy && w && gx && ws, "sg_convT2d_dgrad: null pointer");
  SG_ARG_CHECK(ws_bytes >= ktab_bytes(d->Cout * d->KS * d->KS), "sg_convT2d_dgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int R = d->KS * d->KS;
  Gather g = make_gather(gy, nullptr, d->Cout, 0, d->OH, d->OW, 1, d->H, d->W, d->stride, d->pad, 0);
  const double flops = 2.0 * d->Cout * d->C1 * R * (double)d->N * d->H * d->W;
  int rc = sgk::kn0_run(d->KS, w, d->C1, d->Cout * R, g, d->N, nullptr, gx, d->C1, SG_ACT_NONE, 0.f, flops, ws, ws_bytes, s);
  SG_LAUNCH_CHECK("sg_convT2d_dgrad");
  return rc;
}

// gw[ci,co,kh,kw] = sum_{n,ih,iw} x[n,ci,ih,iw] gy[n,co,ih*s-p+kh,iw*s-p+kw]
extern "C" int sg_convT2d_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, float* gb, void* ws,
                                size_t ws_bytes, sgStream stream) {
  if (check_desc(d, "sg_convT2d_wgrad")) return -1;
  sgk::t_alg_bytes = 4.0 * ((double)d->N * (d->C1) * d->H * d->W + (double)d->Cout * (d->C1) * d->KS * d->KS + (double)d->N * d->Cout * d->OH * d->OW);
  SG_ARG_CHECK(gy && x && gw, "sg_convT2d_wgrad: null pointer");
  hipStream_t s = (hipStream_t)stream;
  Gather g = make_gather(gy, nullptr, d->Cout, 0, d->OH, d->OW, 1, d->H, d->W, d->stride, d->pad, 0);
  if (int rc = sgk::nk_run(d->KS, x, d->C1, d->C1, g, d->N, gw, ws, ws ? ws_bytes : 0,
                         2.0 * d->Cout * (double)d->C1 * d->KS * d->KS * d->N * d->H * d->W, s, nullptr))
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
  for (int i0 = tid * 4; i0 < 64 * HW; i0 += 4096) {          // four float4 loads in flight before the first LDS store
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e * 1024 < 64 * HW) v[e] = *reinterpret_cast<const float4*>(src + i0 + e * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e * 1024;
      if (i < 64 * HW) {
        const int ch = i / HW, px = i - ch * HW;
        float* d = pl + ch * pitch + px;
        d[0] = v[e].x; d[1] = v[e].y; d[2] = v[e].z; d[3] = v[e].w;
      }
    }
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
// UT (optional, flip == 0 only): the same transformed filters with the channel roles swapped, UT[xi][c][r] = U[xi][r][c] -- what
// the adjoint data gradient of the SAME conv multiplies with later in the step (flip == 2 of a second call used to rebuild it
// from the weights: one more pass over 38 MB per ResnetBlock conv).  Written from the same LDS tile with the lanes along r.
__global__ void __launch_bounds__(256) wino_weight_lds_kernel(const float* __restrict__ w, float* __restrict__ U, int R, int Cc,
                                                              int flip, float* __restrict__ UT) {
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
  if (UT == nullptr) return;
  const int rl2 = t & 31;                      // lanes along r this time: UT rows are contiguous in r
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cl2 = (t >> 5) + 8 * q;
    const float* g = S + rl2 * WW_PITCH + cl2 * 9;
    float tt[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
      tt[0][j] = g0; tt[1][j] = 0.5f * (g0 + g1 + g2); tt[2][j] = 0.5f * (g0 - g1 + g2); tt[3][j] = g2;
    }
    const size_t i = (size_t)(c0 + cl2) * R + r0 + rl2;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u[4] = {tt[a][0], 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]), 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]), tt[a][2]};
#pragma unroll
      for (int b = 0; b < 4; ++b) UT[(size_t)(a * 4 + b) * RC + i] = u[b];
    }
  }
}

// one entry for the three call sites (SG_WINO_WT=0 keeps the per-thread kernel).  ``UT``: see wino_weight_lds_kernel; returns
// whether it was written
inline bool wino_weight(const float* w, float* U, int R, int Cc, int flip, hipStream_t s, float* UT = nullptr) {
  if (sg_opt(SG_OPT_WINO_WT) && R % 32 == 0 && Cc % 32 == 0 && aligned16(w)) {
    hipLaunchKernelGGL(wino_weight_lds_kernel, dim3((R / 32) * (Cc / 32)), dim3(256), 0, s, w, U, R, Cc, flip,
                       flip == 0 ? UT : nullptr);
    return flip == 0 && UT != nullptr;
  }
  hipLaunchKernelGGL(wino_weight_kernel, dim3(sg_cdiv((size_t)R * Cc, 256)), dim3(256), 0, s, w, U, R, Cc, flip);
  return false;
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
  for (int i0 = tid * 4; i0 < 64 * HW; i0 += 4096) {          // four float4 loads in flight before the first LDS store
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e * 1024 < 64 * HW) v[e] = *reinterpret_cast<const float4*>(src + i0 + e * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e * 1024;
      if (i < 64 * HW) {
        const int ch = i / HW, px = i - ch * HW;
        float* d = pl + ch * pitch + px;
        d[0] = v[e].x; d[1] = v[e].y; d[2] = v[e].z; d[3] = v[e].w;
      }
    }
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

// The same result from the receiving side, four waves per workgroup: thread (channel c, group g) owns the 2x2 CELLS g, g+4, ...
// of the padded plane (cell (ci, cj) = padded rows 2ci..2ci+1, columns 2cj..2cj+1) and gathers each cell from the <= 4 tiles
// whose 4x4 patch covers it -- tile (ti, tj) contributes the sub-block (2(ci-ti), 2(cj-tj)) of B G B^T -- in ascending tile
// order, i.e. with exactly the additions, in exactly the order, of the overlap-add above (bit-identical).  No two threads
// touch the same accumulator, so the tiles need not be walked serially by ONE thread per channel: 4x the waves per workgroup
// and 3 workgroups per CU instead of 2 single-wave ones (the kernel is latency-bound: 22 -> ~8 us at 8x8 planes).  Every
// tile is read by its four cells (L1 / L2 hits).
__global__ void __launch_bounds__(256) wino_patch_fold_cells_kernel(const float* __restrict__ G, float* __restrict__ gx, int N, int C,
                                                                   int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W, PW = W + 2, PP = (H + 2) * PW, TH = H / 2, TW = W / 2, CW = TW + 1, NCELL = (TH + 1) * CW;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float* acc = lds + c * (PP + 1);
  float* stage = lds + 64 * (PP + 1);              // [64][HW + 1]
  const size_t ld = (size_t)16 * C;
  const float* base = G + (size_t)n * TH * TW * ld + c0 + c;
  for (int q = g; q < NCELL; q += 4) {
    const int ci = q / CW, cj = q - ci * CW;
    float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int dti = 1; dti >= 0; --dti) {
#pragma unroll
      for (int dtj = 1; dtj >= 0; --dtj) {
        const int ti = ci - dti, tj = cj - dtj;
        if (ti < 0 || ti >= TH || tj < 0 || tj >= TW) continue;          // (wave-uniform: q depends on the wave only)
        const float* src = base + (size_t)(ti * TW + tj) * ld;
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i >> 2][i & 3] = src[(size_t)i * C];
        float r[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (dti == 0) { r[0][u] = v[0][u]; r[1][u] = v[1][u] - v[2][u] + v[3][u]; }           // patch rows 0, 1
          else { r[0][u] = -v[0][u] + v[1][u] + v[2][u]; r[1][u] = -v[3][u]; }                  // patch rows 2, 3
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (dtj == 0) { o[a][0] += r[a][0]; o[a][1] += r[a][1] - r[a][2] + r[a][3]; }         // patch columns 0, 1
          else { o[a][0] += -r[a][0] + r[a][1] + r[a][2]; o[a][1] += -r[a][3]; }                // patch columns 2, 3
        }
      }
    }
    float* dst = acc + (2 * ci) * PW + 2 * cj;
    dst[0] = o[0][0]; dst[1] = o[0][1]; dst[PW] = o[1][0]; dst[PW + 1] = o[1][1];
  }
  __syncthreads();
  float* so = stage + c * (HW + 1);
  for (int a = g; a < H; a += 4) {
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
  for (int i = threadIdx.x * 4; i < 64 * HW; i += 1024) {
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
  SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)N * C * (H >> ush) * (W >> ush) + 16.0 * (double)Pstride * C));
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

// The batched GEMM of wino_bgemm() below on a choice of tiles: sg_batched_gemm_nt (tools/bench_wino_gemm.py), the F(4x4,3x3)
// convs (36 x [1024 x 1024] x [1024 x 128] at the trunk: 128x128 tiles would give 288 workgroups for 256 CUs) and the A/B
// switch wino_gemm_tile.  tile: 0 = 128x128, 1 = 64x128, 2 = 64x64; all 32-deep, software-pipelined, plain epilogue.
using CfgDI64W = TileCfg<64, 128, 2, 2, 2>;
using CfgDI64 = TileCfg<64, 64, 2, 2, 2>;
// 16-deep k-tiles: 20 KB of LDS per workgroup instead of 40 -- the 1152 workgroups of an F(4x4,3x3) GEMM (4.5 per CU) are then
// all resident at once (32-deep: 4 per CU by LDS = 1024 slots, the other 128 run as a second, nearly empty round)
using CfgDI64S = TileCfg<64, 64, 2, 1, 2>;
// ... with the channel sum accumulated in chunks (TileCfg::KFOLD): the F(4x4,3x3) forward / data-gradient GEMMs (w43_kfold)
using CfgDI64F256 = TileCfg<64, 64, 2, 2, 2, 256, 1>;
using CfgDI64SF256 = TileCfg<64, 64, 2, 1, 2, 256, 1>;
using CfgDI64F128 = TileCfg<64, 64, 2, 2, 2, 128, 1>;
using CfgDI64SF128 = TileCfg<64, 64, 2, 1, 2, 128, 1>;
// kfold (0 / 128 / 256) x k-tile depth -> one of the six 64x64 instantiations
template <class AL, class BL>
int launch_w43(const AL& al, const BL& bl, const EpRowMajorPlain& ep, int M, int N, int K, hipStream_t s) {
  const int kf = (K > 256 || (K > 128 && sg_opt(SG_OPT_W43_KFOLD) == 128)) ? sg_opt(SG_OPT_W43_KFOLD) : 0;
  const bool deep = sg_opt(SG_OPT_W43_NSUB) != 1;
  if (kf == 256) return deep ? launch_cfg<CfgDI64F256>(al, bl, ep, M, N, K, 1, s) : launch_cfg<CfgDI64SF256>(al, bl, ep, M, N, K, 1, s);
  if (kf == 128) return deep ? launch_cfg<CfgDI64F128>(al, bl, ep, M, N, K, 1, s) : launch_cfg<CfgDI64SF128>(al, bl, ep, M, N, K, 1, s);
  return deep ? launch_cfg<CfgDI64>(al, bl, ep, M, N, K, 1, s) : launch_cfg<CfgDI64S>(al, bl, ep, M, N, K, 1, s);
}
void wino_bgemm_tile(int tile, const float* A, const float* B, float* Cout, int M, int cols, int K, double flops, hipStream_t s, int NB,
                     int kind = SG_K_OTHER) {
  sgk::t_alg_bytes = 4.0 * NB * ((double)M * K + (double)cols * K + (double)M * cols);
  t_batch = BatchInfo{}; t_batch.cols_per_batch = cols; t_batch.nbatch = NB; t_batch.a_stride = M * K; t_batch.batch_major = 1;
  {
    SgProfScope prof(kind, s, flops, 0);
    if (tile == 1)
      launch_cfg<CfgDI64W>(LoadKContig<64, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, NB * cols},
                           EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
    else if ((tile == 2 || tile == 3) && kind == SG_K_WINO43_GEMM)        // F(4x4,3x3) forward: chunked channel sum (w43_kfold)
      launch_w43(LoadKContig<64, true, false>{A, K, M}, LoadKContig<64, true, false>{B, K, NB * cols},
                 EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, s);
    else if (tile == 2)
      launch_cfg<CfgDI64>(LoadKContig<64, true, false>{A, K, M}, LoadKContig<64, true, false>{B, K, NB * cols},
                          EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
    else if (tile == 3)
      launch_cfg<CfgDI64S>(LoadKContig<64, true, false>{A, K, M}, LoadKContig<64, true, false>{B, K, NB * cols},
                           EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
    else
      launch_cfg<CfgDI128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, NB * cols},
                           EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
  }
  t_batch = BatchInfo{};
}

// C[m][b*cols + j] = sum_k A[b][m][k] * B[b*cols + j][k]   (NB batches -- 16 for F(2x2,3x3), 25 (x k-chunks) for F(2x2,4x4) --,
// everything a multiple of the tile)
void wino_bgemm(const float* A, const float* B, float* Cout, int M, int cols, int K, double flops, hipStream_t s, int NB = 16) {
  const int tsel = NB == 16 ? sg_opt(SG_OPT_WINO_GEMM_TILE) : sg_opt(SG_OPT_W24_GEMM_TILE);
  if ((tsel == 1 || tsel == 2) && M % 64 == 0 && cols % 128 == 0 && K % 32 == 0) {
    wino_bgemm_tile(tsel, A, B, Cout, M, cols, K, flops, s, NB, NB == 16 ? SG_K_WINO_GEMM_128 : SG_K_WINO24_GEMM);
    return;
  }
  sgk::t_alg_bytes = 4.0 * NB * ((double)M * K + (double)cols * K + (double)M * cols);
  t_batch = BatchInfo{}; t_batch.cols_per_batch = cols; t_batch.nbatch = NB; t_batch.a_stride = M * K; t_batch.batch_major = 1;
  {
    SgProfScope prof(NB == 16 ? SG_K_WINO_GEMM_128 : SG_K_WINO24_GEMM, s, flops, 0);
    // 128x128 tiles, 32-deep k-tiles, software-pipelined fragment reads, unconditional epilogue (the variants this replaced --
    // 128x64 tiles, 16-deep tiles, the plain loop, the general epilogue -- measured 1.5..9 % slower: profiles/r03_sweep_tiles.txt)
    if (sg_opt(SG_OPT_WINO_PIPE) == 2)
      launch_cfg<CfgDI128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, NB * cols},
                           EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
    else
      launch_cfg<CfgDP128>(LoadKContig<128, true, false>{A, K, M}, LoadKContig<128, true, false>{B, K, NB * cols},
                           EpRowMajorPlain{Cout, NB * cols}, M, NB * cols, K, 1, s);
  }
  t_batch = BatchInfo{};
}


// T[m][xi*Cc + c] = sum_p Ytp[xi][p][m] * V[xi][p][c]: the Winograd weight gradient straight from the operands the forward (V) and
// the adjoint data gradient (Ytp) of the same conv already built -- both tile-major, i.e. x-contiguous for a GEMM over p
// (LoadXContigS): no second input / gradient transform (2 launches and 2 x 41 MB per ResnetBlock conv saved)
void wino_bgemm_x(const float* Ytp, const float* V, float* T, int M, int Cc, int P, double flops, hipStream_t s) {
  sgk::t_alg_bytes = 4.0 * 16 * ((double)M * P + (double)Cc * P + (double)M * Cc);
  t_batch = BatchInfo{}; t_batch.cols_per_batch = Cc; t_batch.nbatch = 16; t_batch.a_stride = P * M; t_batch.b_stride = P * Cc;
  t_batch.batch_major = 1;
  {
    SgProfScope prof(SG_K_WINO_GEMM_128, s, flops, 0);
    launch_cfg<CfgDI128>(LoadXContigS<128>{Ytp, M, 0}, LoadXContigS<128>{V, Cc, Cc}, EpRowMajorPlain{T, 16 * Cc}, M, 16 * Cc, P, 1, s);
  }
  t_batch = BatchInfo{};
}


// ================================================================================================
// Winograd F(4x4, 3x3) for the small-plane reflection-padded ResnetBlock convs (generators.py:62-91 through layers.py:234-273: the
// nine 1024-channel blocks at 8x8 -- 16x16 at 256^2 -- are 54 of the step's GEMM launches): 36 multiplies per 4x4 output tile
// and channel pair instead of 4 x 16 = 64 with F(2x2,3x3), i.e. 1.78x fewer MACs, 2.25x (not 4x) the activation bytes in the
// transformed operands.  Interpolation points 0, 1, -1, 1/2, -2, inf -- the set with the smallest fp32 error of the ones
// tried (tools/winograd_f43.py: 4.1e-6 of max|y| over a 1024-channel reduction against 1.3e-6 for F(2x2,3x3) and 0.8e-6 direct;
// the classic 0, +-1, +-2 set: 8.1e-6):
//   A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
//   G   = [1 0 0; 1/3 1/3 1/3; -1/3 1/3 -1/3; -16/15 -8/15 -4/15; 1/15 -2/15 4/15; 0 0 1]
//   B^T = [1 -3/2 -2 3/2 1 0; 0 -1 1/2 5/2 1 0; 0 1 -5/2 1/2 1 0; 0 -2 -1 2 1 0; 0 1/2 -1 -1/2 1 0; 0 1 -3/2 -2 3/2 1]
//   y = A^T [sum_c (G g G^T) (.) (B^T d B)] A            (forward)
//   g_d = B [sum_k (G g G^T) (.) (A g_y A^T)] B^T        (data gradient: the adjoint over the OUTPUT tiles, overlap-added)
//   g_w = G^T [sum_p (A g_y A^T) (.) (B^T d B)] G        (weight gradient)
// Same structure as the F(2x2,3x3) path above: LDS-staged transforms with the lanes along the channel, 36 batched dense GEMMs
// on 64x64 tiles (P = N H W / 16 tiles per launch: 128 at the benchmark shape, too few columns for 128-wide tiles), the
// filter transform U kept from the forward for the data gradient (read x-contiguous there: no transposed twin), V and Ytp
// kept for the weight gradient.
// ================================================================================================
namespace {
__device__ __forceinline__ constexpr float w43_bt(int i, int j) {
  constexpr float m[6][6] = {{1.f, -1.5f, -2.f, 1.5f, 1.f, 0.f}, {0.f, -1.f, 0.5f, 2.5f, 1.f, 0.f}, {0.f, 1.f, -2.5f, 0.5f, 1.f, 0.f},
                             {0.f, -2.f, -1.f, 2.f, 1.f, 0.f},   {0.f, 0.5f, -1.f, -0.5f, 1.f, 0.f}, {0.f, 1.f, -1.5f, -2.f, 1.5f, 1.f}};
  return m[i][j];
}
__device__ __forceinline__ constexpr float w43_g(int i, int j) {
  constexpr float m[6][3] = {{1.f, 0.f, 0.f}, {1.f / 3.f, 1.f / 3.f, 1.f / 3.f}, {-1.f / 3.f, 1.f / 3.f, -1.f / 3.f},
                             {-16.f / 15.f, -8.f / 15.f, -4.f / 15.f}, {1.f / 15.f, -2.f / 15.f, 4.f / 15.f}, {0.f, 0.f, 1.f}};
  return m[i][j];
}
__device__ __forceinline__ constexpr float w43_at(int i, int j) {
  constexpr float m[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 0.5f, -2.f, 0.f}, {0.f, 1.f, 1.f, 0.25f, 4.f, 0.f},
                             {0.f, 1.f, -1.f, 0.125f, -8.f, 1.f}};
  return m[i][j];
}
// d act(z) / dz from the pre-activation z (the InstanceNorm kernels' convention, norm.hip: act_grad_from_pre)
__device__ __forceinline__ float act_grad_from_pre_w(float z, int act, float slope) {
  switch (act) {
    case SG_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case SG_ACT_LEAKY: return z > 0.f ? 1.f : slope;
    default: return 1.f;                 // (only ReLU / LeakyReLU fuse into a normalisation launch: layers._peek_act)
  }
}
// acc += c * x for a compile-time coefficient: nothing for 0, an add / subtract for +-1
#define W43_ACC(acc, c, x) do { if ((c) == 1.f) (acc) += (x); else if ((c) == -1.f) (acc) -= (x); else if ((c) != 0.f) (acc) += (c) * (x); } while (0)

// v = B^T d B of a 6x6 patch
__device__ __forceinline__ void w43_input_xform(const float (&d)[6][6], float (&v)[6][6]) {
  float t[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_bt(i, k), d[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_bt(j, k), t[i][k]);
      v[i][j] = a;
    }
}
// u = G g G^T of a 3x3 filter
__device__ __forceinline__ void w43_weight_xform(const float* __restrict__ g, float (&u)[6][6]) {
  float t[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) W43_ACC(a, w43_g(i, k), g[k * 3 + j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) W43_ACC(a, w43_g(j, k), t[i][k]);
      u[i][j] = a;
    }
}
// y = A^T q A (4x4 from 6x6)
__device__ __forceinline__ void w43_output_xform(const float (&q)[6][6], float (&y)[4][4]) {
  float t[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_at(i, k), q[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_at(j, k), t[i][k]);
      y[i][j] = a;
    }
}
// yt = A gy A^T (6x6 from a 4x4 gradient tile), A = (A^T)^T
__device__ __forceinline__ void w43_gy_xform(const float (&g)[4][4], float (&yt)[6][6]) {
  float t[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) W43_ACC(a, w43_at(k, i), g[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) W43_ACC(a, w43_at(k, j), t[i][k]);
      yt[i][j] = a;
    }
}
// r = B q B^T (the adjoint of the input transform), B = (B^T)^T
__device__ __forceinline__ void w43_patch_xform(const float (&q)[6][6], float (&r)[6][6]) {
  float t[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_bt(k, i), q[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_bt(k, j), t[i][k]);
      r[i][j] = a;
    }
}
// gw = G^T q G (3x3 from 6x6)
__device__ __forceinline__ void w43_wgrad_xform(const float (&q)[6][6], float (&o)[3][3]) {
  float t[3][6];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_g(k, i), q[k][j]);
      t[i][j] = a;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) W43_ACC(a, w43_g(k, j), t[i][k]);
      o[i][j] = a;
    }
}

// stage the H x W planes of 64 consecutive channels of image n (one contiguous block) in LDS, pitch HW + 1
__device__ __forceinline__ void w43_stage_planes(const float* __restrict__ src, float* __restrict__ pl, int HW, int tid) {
  const int pitch = HW + 1;
  for (int i0 = tid * 4; i0 < 64 * HW; i0 += 4096) {          // four float4 loads in flight before the first LDS store
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e * 1024 < 64 * HW) v[e] = *reinterpret_cast<const float4*>(src + i0 + e * 1024);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e * 1024;
      if (i < 64 * HW) {
        const int ch = i / HW, px = i - ch * HW;
        float* d = pl + ch * pitch + px;
        d[0] = v[e].x; d[1] = v[e].y; d[2] = v[e].z; d[3] = v[e].w;
      }
    }
  }
}

// V[xi][p][c] = (B^T d B)[xi] of the reflection-padded 6x6 patch of tile p = (n, ti, tj): rows 4ti-1 .. 4ti+4
__global__ void __launch_bounds__(256) w43_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int HW = H * W, pitch = HW + 1, TH = H / 4, TW = W / 4;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  w43_stage_planes(x + ((size_t)n * C + c0) * HW, pl, HW, tid);
  __syncthreads();
  const int c = tid & 63, g = tid >> 6;
  const float* pc = pl + c * pitch;
  const size_t P = (size_t)N * TH * TW;
  for (int t = g; t < TH * TW; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    float d[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int ih = wino_reflect(4 * ti - 1 + a, H);
#pragma unroll
      for (int b = 0; b < 6; ++b) d[a][b] = pc[ih * W + wino_reflect(4 * tj - 1 + b, W)];
    }
    float v[6][6];
    w43_input_xform(d, v);
    const size_t p = (size_t)n * TH * TW + t;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) V[((size_t)(i * 6 + j) * P + p) * C + c0 + c] = v[i][j];
  }
}

// U[xi][r][c] = (G g G^T)[xi], g = w[r][c]: a workgroup owns a 32x32 block of (r, c) (wino_weight_lds_kernel's staging)
__global__ void __launch_bounds__(256) w43_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int R, int Cc) {
  __shared__ float S[32 * WW_PITCH];
  const int cb = blockIdx.x % (Cc / 32), rb = blockIdx.x / (Cc / 32);
  const int r0 = rb * 32, c0 = cb * 32, t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int idx = i * 256 + t, j = idx / 72, o = (idx - j * 72) * 4;
    const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)(r0 + j) * Cc + c0) * 9 + o);
    float* d = S + j * WW_PITCH + o;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const size_t RC = (size_t)R * Cc;
  const int cl = t & 31;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = (t >> 5) + 8 * q;
    float u[6][6];
    w43_weight_xform(S + rl * WW_PITCH + cl * 9, u);
    const size_t i = (size_t)(r0 + rl) * Cc + c0 + cl;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) U[(size_t)(a * 6 + b) * RC + i] = u[a][b];
  }
}

// y[n][m][4ti+a][4tj+b] = act((A^T Mx A)[a][b] + bias[m]),  Mx[m][xi*P + p]
__global__ void w43_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias, float* __restrict__ y, int N, int M,
                                  int H, int W, int act, float slope) {
  const int TH = H / 4, TW = W / 4;
  const size_t P = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * M) return;
  const size_t p = idx % P;
  const int m = (int)(idx / P);
  const float* src = Mx + (size_t)m * 36 * P + p;
  float q[6][6];
#pragma unroll
  for (int i = 0; i < 36; ++i) q[i / 6][i % 6] = src[(size_t)i * P];
  float o[4][4];
  w43_output_xform(q, o);
  const float b = bias ? bias[m] : 0.f;
  const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
  float* dst = y + (((size_t)n * M + m) * H + 4 * ti) * W + 4 * tj;
#pragma unroll
  for (int a = 0; a < 4; ++a)
    *reinterpret_cast<float4*>(dst + a * W) = make_float4(sg_apply_act(o[a][0] + b, act, slope), sg_apply_act(o[a][1] + b, act, slope),
                                                          sg_apply_act(o[a][2] + b, act, slope), sg_apply_act(o[a][3] + b, act, slope));
}

// Ytp[xi][p][m] = (A gy A^T)[xi] of the 4x4 gradient tile p (LDS-staged, lanes along the channel m)
__global__ void __launch_bounds__(256) w43_gy_kernel(const float* __restrict__ gy, float* __restrict__ Ytp, int N, int M, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float pl[];
  const int HW = H * W, pitch = HW + 1, TH = H / 4, TW = W / 4;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  w43_stage_planes(gy + ((size_t)n * M + c0) * HW, pl, HW, tid);
  __syncthreads();
  const int c = tid & 63, g = tid >> 6;
  const float* pc = pl + c * pitch;
  const size_t P = (size_t)N * TH * TW;
  for (int t = g; t < TH * TW; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    float gt[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) gt[a][b] = pc[(4 * ti + a) * W + 4 * tj + b];
    float yt[6][6];
    w43_gy_xform(gt, yt);
    const size_t p = (size_t)n * TH * TW + t;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) Ytp[((size_t)(i * 6 + j) * P + p) * M + c0 + c] = yt[i][j];
  }
}

// gx[n][c][H][W] from G[p][xi*C + c]: the 6x6 patches B G B^T of the tiles are overlap-added into the padded (H+2) x (W+2) plane
// of their channel and the reflection is folded.  Thread (c, g) transforms the tiles g, g+4, ... of channel c (all loads and
// the arithmetic of the four groups run side by side); the patches are then added into the channel's LDS plane ROUND by round
// in tile order -- in round t only the owner of tile t adds -- so every cell receives its contributions in ascending tile
// order whatever the wave timing: deterministic, no atomics.
__global__ void __launch_bounds__(256) w43_fold_kernel(const float* __restrict__ G, float* __restrict__ gx, int N, int C, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W, PW = W + 2, PP = (H + 2) * PW, TH = H / 4, TW = W / 4, NT = TH * TW;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, c = threadIdx.x & 63, g = threadIdx.x >> 6;
  float* acc = lds + c * (PP + 1);
  float* stage = lds + 64 * (PP + 1);              // [64][HW + 1]
  for (int i = g; i < PP; i += 4) acc[i] = 0.f;
  const size_t ld = (size_t)36 * C;
  const float* base = G + (size_t)n * NT * ld + c0 + c;
  __syncthreads();
  for (int t0 = 0; t0 < NT; t0 += 4) {
    const int t = t0 + g;                          // this thread's tile of the round (wave-uniform)
    float r[6][6];
    if (t < NT) {
      float q[6][6];
      const float* src = base + (size_t)t * ld;
#pragma unroll
      for (int i = 0; i < 36; ++i) q[i / 6][i % 6] = src[(size_t)i * C];
      w43_patch_xform(q, r);
    }
    for (int k = 0; k < 4; ++k) {                  // tiles t0 .. t0+3 in order
      if (g == k && t < NT) {
        const int ti = t / TW, tj = t - ti * TW;
        float* dst = acc + (4 * ti) * PW + 4 * tj;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b) dst[a * PW + b] += r[a][b];
      }
      __syncthreads();
    }
  }
  float* so = stage + c * (HW + 1);
  for (int a = g; a < H; a += 4) {
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
  for (int i = threadIdx.x * 4; i < 64 * HW; i += 1024) {
    const int ch = i / HW, px = i - ch * HW;
    const float* sp = stage + ch * (HW + 1) + px;
    *reinterpret_cast<float4*>(dstg + i) = make_float4(sp[0], sp[1], sp[2], sp[3]);
  }
}

// ---- F(4x4,3x3) conv + InstanceNorm fused at both ends of the GEMMs (ResnetBlock: conv -> InstanceNorm -> ReLU -> conv ->
// InstanceNorm -> + x, layers.py:251-270) --------------------------------------------------------------------------------------
// Forward: the output transform of tile-major GEMM results Mx[p][xi*M + m] (lanes along the channel, like the fold kernel), the
// bias, and the InstanceNorm of the plane the workgroup's threads hold between them: the conv result is written once (the
// backward needs it) next to the normalised (+ activation)(+ residual) output -- the separate InstanceNorm launch and its read
// of the conv result are gone.  TPT tiles per thread (1 at 8x8 planes, 4 at 16x16).
template <int TPT>
__global__ void __launch_bounds__(256) w43_output_in_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                           const float* __restrict__ skip, float* __restrict__ ypre,
                                                           float* __restrict__ out, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int N, int M, int H, int W, float eps, int act,
                                                           float slope) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W, TH = H / 4, TW = W / 4, NT = TH * TW, pitch = HW + 1;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, c = threadIdx.x & 63, g = threadIdx.x >> 6, m = c0 + c;
  float* sa = lds;                                  // [64][HW + 1]: conv result
  float* sb = lds + 64 * pitch;                     // [64][HW + 1]: normalised result
  float* red = lds + 128 * pitch;                   // [4][64]
  const float b = bias ? bias[m] : 0.f;
  const size_t ld = (size_t)36 * M;
  float o[TPT][4][4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int t = g + 4 * k;
    if (t < NT) {
      const float* src = Mx + ((size_t)n * NT + t) * ld + m;
      float q[6][6];
#pragma unroll
      for (int i = 0; i < 36; ++i) q[i / 6][i % 6] = src[(size_t)i * M];
      w43_output_xform(q, o[k]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[k][a][e] += b; s += o[k][a][e]; }
    }
  }
  red[g * 64 + c] = s;
  __syncthreads();
  const float mean = (((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c]) / (float)HW;
  __syncthreads();
  float qv = 0.f;
#pragma unroll
  for (int k = 0; k < TPT; ++k)
    if (g + 4 * k < NT) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dlt = o[k][a][e] - mean; qv += dlt * dlt; }
    }
  red[g * 64 + c] = qv;
  __syncthreads();
  const float rstd = 1.f / sqrtf((((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c]) / (float)HW + eps);
  if (g == 0) { mean_o[(size_t)n * M + m] = mean; rstd_o[(size_t)n * M + m] = rstd; }
#pragma unroll
  for (int k = 0; k < TPT; ++k) {
    const int t = g + 4 * k;
    if (t < NT) {
      const int ti = t / TW, tj = t - ti * TW;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int px = (4 * ti + a) * W + 4 * tj + e;
          sa[c * pitch + px] = o[k][a][e];
          sb[c * pitch + px] = sg_apply_act((o[k][a][e] - mean) * rstd, act, slope);
        }
    }
  }
  __syncthreads();
  // the 64 planes of the workgroup are one contiguous block of each tensor: coalesced float4 rows
  const size_t base = ((size_t)n * M + c0) * HW;
  for (int i = threadIdx.x * 4; i < 64 * HW; i += 1024) {
    const int ch = i / HW, px = i - ch * HW;
    const float* pa = sa + ch * pitch + px;
    const float* pb = sb + ch * pitch + px;
    *reinterpret_cast<float4*>(ypre + base + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    float4 v = make_float4(pb[0], pb[1], pb[2], pb[3]);
    if (skip) {
      const float4 k4 = *reinterpret_cast<const float4*>(skip + base + i);
      v.x += k4.x; v.y += k4.y; v.z += k4.z; v.w += k4.w;
    }
    *reinterpret_cast<float4*>(out + base + i) = v;
  }
}

// Backward: InstanceNorm's backward (gout -> the conv's gy, layers.py:296 affine=False) and the gradient transform A gy A^T of
// that conv in one launch: the planes of gout and of the conv result are staged in LDS, thread (c, g) owns the tiles g, g+4, ...
// of channel c.  gconv is written too (the bias gradient and a weight gradient without saved operands read it).
__global__ void __launch_bounds__(256) w43_gy_in_kernel(const float* __restrict__ gout, const float* __restrict__ ypre,
                                                       const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                       float* __restrict__ gconv, float* __restrict__ Ytp, float* __restrict__ gb_part,
                                                       int N, int M, int H, int W, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = H * W, pitch = HW + 1, TH = H / 4, TW = W / 4, NT = TH * TW;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x, c = tid & 63, g = tid >> 6;
  float* pg = lds;                                  // [64][HW + 1]: gout, then the conv's gy
  float* px = lds + 64 * pitch;                     // [64][HW + 1]: conv result
  float* red = lds + 128 * pitch;                   // [2][4][64]
  const size_t base = ((size_t)n * M + c0) * HW;
  w43_stage_planes(gout + base, pg, HW, tid);
  w43_stage_planes(ypre + base, px, HW, tid);
  __syncthreads();
  const float mean = mean_i[(size_t)n * M + c0 + c], rstd = rstd_i[(size_t)n * M + c0 + c];
  float* gc = pg + c * pitch;
  const float* xc = px + c * pitch;
  float s1 = 0.f, s2 = 0.f;
  for (int t = g; t < NT; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = (4 * ti + a) * W + 4 * tj + e;
        const float z = (xc[p] - mean) * rstd;
        const float gp = gc[p] * act_grad_from_pre_w(z, act, slope);
        s1 += gp; s2 += gp * z;
      }
  }
  red[g * 64 + c] = s1;
  red[256 + g * 64 + c] = s2;
  __syncthreads();
  const float inv = 1.f / (float)HW;
  const float m1 = (((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c]) * inv;
  const float m2 = (((red[256 + c] + red[320 + c]) + red[384 + c]) + red[448 + c]) * inv;
  const size_t P = (size_t)N * NT;
  float sb = 0.f;                                   // this thread's share of the plane sum of the conv's gy (= bias gradient)
  for (int t = g; t < NT; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    float gt[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = (4 * ti + a) * W + 4 * tj + e;
        const float z = (xc[p] - mean) * rstd;
        const float gp = gc[p] * act_grad_from_pre_w(z, act, slope);
        gt[a][e] = rstd * (gp - m1 - z * m2);
        gc[p] = gt[a][e];                           // (this thread's own cells: no other thread reads them before the barrier)
        sb += gt[a][e];
      }
    float yt[6][6];
    w43_gy_xform(gt, yt);
    const size_t p = (size_t)n * NT + t;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) Ytp[((size_t)(i * 6 + j) * P + p) * M + c0 + c] = yt[i][j];
  }
  red[512 + g * 64 + c] = sb;                       // (own region: slower threads may still be reading red[0..511] for m1 / m2)
  __syncthreads();
  // per-(image, channel) sums of the conv's gy: the bias gradient is their sum over the images (w43_bias_sum_kernel) -- one
  // 3 us launch instead of sg_channel_sum's two passes over gconv (round 6; 18 convs per step)
  if (gb_part != nullptr && g == 0)
    gb_part[(size_t)n * M + c0 + c] = ((red[512 + c] + red[576 + c]) + red[640 + c]) + red[704 + c];
  for (int i = tid * 4; i < 64 * HW; i += 1024) {
    const int ch = i / HW, q = i - ch * HW;
    const float* sp = pg + ch * pitch + q;
    *reinterpret_cast<float4*>(gconv + base + i) = make_float4(sp[0], sp[1], sp[2], sp[3]);
  }
}

// gb[m] = sum_n part[n][m], images in ascending order (deterministic)
__global__ void w43_bias_sum_kernel(const float* __restrict__ part, float* __restrict__ gb, int N, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float v = 0.f;
  for (int n0 = 0; n0 < N; n0 += 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = n0 + e < N ? part[(size_t)(n0 + e) * M + m] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v += t[e];
  }
  gb[m] = v;
}

// gw[m][c][3][3] = G^T T G,  T[m][xi*C + c]
__global__ void w43_wgrad_output_kernel(const float* __restrict__ T, float* __restrict__ gw, int M, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * C) return;
  const int c = (int)(idx % C);
  const size_t m = idx / C;
  const float* src = T + m * 36 * C + c;
  float q[6][6];
#pragma unroll
  for (int i = 0; i < 36; ++i) q[i / 6][i % 6] = src[(size_t)i * C];
  float o[3][3];
  w43_wgrad_xform(q, o);
  float* dst = gw + idx * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) dst[a * 3 + b] = o[a][b];
}

// the three batched GEMMs (36 batches, 64x64 tiles, 32-deep software-pipelined k-tiles)
//   forward:          Mx[m][xi*P + p]  = sum_c U[xi][m][c] * V[xi][p][c]        (both K-contiguous)
//   data gradient:    G[p][xi*C + c]   = sum_k Ytp[xi][p][k] * U[xi][k][c]      (A K-contiguous, B x-contiguous)
//   weight gradient:  T[m][xi*C + c]   = sum_p Ytp[xi][p][m] * V[xi][p][c]      (both x-contiguous, K = P)
inline int w43_tile() { return sg_opt(SG_OPT_W43_NSUB) == 1 ? 3 : 2; }
void w43_gemm_fwd(const float* U, const float* V, float* Mx, int M, int P, int C, hipStream_t s) {
  wino_bgemm_tile(w43_tile(), U, V, Mx, M, P, C, 2.0 * 36.0 * M * (double)P * C, s, 36, SG_K_WINO43_GEMM);
}
void w43_gemm_dgrad(const float* Ytp, const float* U, float* G, int P, int C, int K, hipStream_t s) {
  sgk::t_alg_bytes = 4.0 * 36 * ((double)P * K + (double)C * K + (double)P * C);
  t_batch = BatchInfo{}; t_batch.cols_per_batch = C; t_batch.nbatch = 36; t_batch.a_stride = P * K; t_batch.b_stride = K * C;
  t_batch.batch_major = 1;
  {
    SgProfScope prof(SG_K_WINO43_GEMM, s, 2.0 * 36.0 * P * (double)C * K, 0);
    launch_w43(LoadKContig<64, true, false>{Ytp, K, P}, LoadXContigS<64>{U, C, C}, EpRowMajorPlain{G, 36 * C}, P, 36 * C, K, s);
  }
  t_batch = BatchInfo{};
}
void w43_gemm_wgrad(const float* Ytp, const float* V, float* T, int M, int C, int P, hipStream_t s) {
  sgk::t_alg_bytes = 4.0 * 36 * ((double)M * P + (double)C * P + (double)M * C);
  t_batch = BatchInfo{}; t_batch.cols_per_batch = C; t_batch.nbatch = 36; t_batch.a_stride = P * M; t_batch.b_stride = P * C;
  t_batch.batch_major = 1;
  {
    SgProfScope prof(SG_K_WINO43_GEMM, s, 2.0 * 36.0 * M * (double)C * P, 0);
    const int wt = sg_opt(SG_OPT_W43_WGRAD_TILE);
    if (wt == 1 && M % 128 == 0 && C % 128 == 0 && P % 32 == 0)
      launch_cfg<CfgDI128>(LoadXContigS<128>{Ytp, M, 0}, LoadXContigS<128>{V, C, C}, EpRowMajorPlain{T, 36 * C}, M, 36 * C, P, 1, s);
    else if (wt == 2 && C % 128 == 0 && P % 32 == 0)
      launch_cfg<CfgDI64W>(LoadXContigS<64>{Ytp, M, 0}, LoadXContigS<128>{V, C, C}, EpRowMajorPlain{T, 36 * C}, M, 36 * C, P, 1, s);
    else if (sg_opt(SG_OPT_W43_NSUB) == 1)
      launch_cfg<CfgDI64S>(LoadXContigS<64>{Ytp, M, 0}, LoadXContigS<64>{V, C, C}, EpRowMajorPlain{T, 36 * C}, M, 36 * C, P, 1, s);
    else
      launch_cfg<CfgDI64>(LoadXContigS<64>{Ytp, M, 0}, LoadXContigS<64>{V, C, C}, EpRowMajorPlain{T, 36 * C}, M, 36 * C, P, 1, s);
  }
  t_batch = BatchInfo{};
}
template <class K> inline void w43_lds_attr(K kernel, size_t lds) {
  if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
}  // namespace

// ================================================================================================
// Winograd F(2x2, 4x4) for the stride-1 4x4 convs of the PatchGANs (discriminators.py:221-228: Conv2d(256, 512, 4, 1, 2), the
// largest single layer of the discriminator steps): 25 multiplies per 2x2 output tile and channel pair instead of 64.
// Interpolation points 0, 1, -1, -2, inf (the set with the smallest fp32 error of the ones tried: ~3x the direct kernel's):
//   A^T = [1 1 1 1 0; 0 1 -1 -2 1]
//   G   = [-1/2 0 0 0; 1/6 1/6 1/6 1/6; 1/2 -1/2 1/2 -1/2; -1/6 1/3 -2/3 4/3; 0 0 0 1]
//   B^T = [-2 -1 2 1 0; 0 2 3 1 0; 0 -2 1 1 0; 0 -1 0 1 0; 0 -2 -1 2 1]
//   y = A^T [(G g G^T) (.) (B^T d B)] A,   dL/dg = G^T [sum_tiles (A dy A^T) (.) (B^T d B)] G,   data gradient = the forward
//   form on gy with the 180-degree-rotated, transposed filter and padding 3 - pad.
// Forward / data gradient: 25 batched GEMMs [M x K] x [K x tiles]; weight gradient: 25 x S batched GEMMs over k-chunks of the
// tiles (M x C is only a handful of 128-tiles: the chunks fill the chip), summed in a fixed order by the output transform.
// ================================================================================================
__device__ __forceinline__ void w24_bt(const float (&d)[5], float (&o)[5]) {
  o[0] = -2.f * d[0] - d[1] + 2.f * d[2] + d[3];
  o[1] = 2.f * d[1] + 3.f * d[2] + d[3];
  o[2] = -2.f * d[1] + d[2] + d[3];
  o[3] = d[3] - d[1];
  o[4] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
}
__device__ __forceinline__ void w24_g(const float (&g)[4], float (&o)[5]) {
  o[0] = -0.5f * g[0];
  o[1] = (g[0] + g[1] + g[2] + g[3]) * (1.f / 6.f);
  o[2] = (g[0] - g[1] + g[2] - g[3]) * 0.5f;
  o[3] = (-g[0] + 2.f * g[1] - 4.f * g[2] + 8.f * g[3]) * (1.f / 6.f);
  o[4] = g[3];
}
__device__ __forceinline__ void w24_at(const float (&m)[5], float (&o)[2]) {
  o[0] = m[0] + m[1] + m[2] + m[3];
  o[1] = m[1] - m[2] - 2.f * m[3] + m[4];
}
__device__ __forceinline__ void w24_a(const float (&y)[2], float (&o)[5]) {      // A y
  o[0] = y[0]; o[1] = y[0] + y[1]; o[2] = y[0] - y[1]; o[3] = y[0] - 2.f * y[1]; o[4] = y[1];
}
__device__ __forceinline__ void w24_gt(const float (&t)[5], float (&o)[4]) {     // G^T t
  o[0] = -0.5f * t[0] + (t[1] - t[3]) * (1.f / 6.f) + 0.5f * t[2];
  o[1] = (t[1] + 2.f * t[3]) * (1.f / 6.f) - 0.5f * t[2];
  o[2] = (t[1] - 4.f * t[3]) * (1.f / 6.f) + 0.5f * t[2];
  o[3] = (t[1] + 8.f * t[3]) * (1.f / 6.f) - 0.5f * t[2] + t[4];
}

// V = B^T d B of the 5x5 patch of tile p = (n, ti, tj) (patch origin (2ti + off, 2tj + off), zero outside the plane).
// KMAJ = 0: V[xi][p][c] (GEMM operand with the channel as k; lanes along c: coalesced stores);
// KMAJ = 1: V[xi*S + s][c][pc], p = s*Pc + pc (weight gradient: the tile index is k; lanes along p).  Rows p >= P are zeros.
template <int KMAJ>
__global__ void w24_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H, int W, int TH, int TW,
                                 int off, size_t Pstride, int Pc, int S) {
  const size_t P = (size_t)N * TH * TW, total = Pstride * C;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t p = KMAJ ? idx % Pstride : idx / C;
  const int c = (int)(KMAJ ? idx / Pstride : idx % C);
  float d[5][5];
  if (p < P) {
    const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
    const float* xp = x + ((size_t)n * C + c) * H * W;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int ih = 2 * ti + off + a;
      const bool rok = (unsigned)ih < (unsigned)H;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int iw = 2 * tj + off + b;
        const bool ok = rok && (unsigned)iw < (unsigned)W;
        d[a][b] = ok ? xp[ih * W + iw] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) d[a][b] = 0.f;
  }
  float t[5][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {               // columns: t = B^T d
    const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]};
    float o[5];
    w24_bt(col, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {               // rows: V = t B
    float o[5];
    w24_bt(t[i], o);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const size_t xi = (size_t)(i * 5 + j);
      if (KMAJ) {
        const size_t sidx = p / (size_t)Pc, pc = p - sidx * (size_t)Pc;
        V[((xi * S + sidx) * C + c) * (size_t)Pc + pc] = o[j];
      } else {
        V[(xi * Pstride + p) * C + c] = o[j];
      }
    }
  }
}

// The same transform (layout V[xi][p][c]) for the small planes of the PatchGANs (H*W <= 639: the 17x17 / 18x18 / 9x9 / 10x10 maps):
// the kernel above lets neighbouring lanes read neighbouring CHANNELS, H*W floats apart (measured 52-100 us per call, ~1 TB/s).
// Here a workgroup stages the rows its tiles need of 64 channel planes of one image in LDS (lanes along the pixels: coalesced),
// then the lanes run along the channel for the LDS reads (pitch H*W + 1: conflict-free) and the global stores (256 contiguous
// bytes per wave).  grid = (C/64, N, row splits): the tile rows of an image are shared out over blockIdx.z to fill the chip.
__global__ void __launch_bounds__(256) w24_input_small_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int H,
                                                             int W, int TH, int TW, int off, size_t Pstride, int rows_per,
                                                             int pitch) {
  extern __shared__ __attribute__((aligned(16))) float pl[];          // [64 channels][pitch]: input rows [rlo, rhi) of each plane
  const int HW = H * W;
  const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  const int tr0 = (int)blockIdx.z * rows_per, tr1 = min(TH, tr0 + rows_per);
  const int lane = tid & 63, g = tid >> 6;
  const int rlo = max(0, 2 * tr0 + off), rhi = min(H, 2 * (tr1 - 1) + off + 5);
  {
    const int cnt = (rhi - rlo) * W;
    const float* src = x + ((size_t)n * C + c0) * HW + rlo * W;
    for (int ch = g; ch < 64; ch += 4)
      for (int e = lane; e < cnt; e += 64) pl[ch * pitch + e] = src[(size_t)ch * HW + e];
  }
  __syncthreads();
  const float* pc = pl + lane * pitch - rlo * W;
  const size_t P = (size_t)N * TH * TW;
  for (int t = tr0 * TW + g; t < tr1 * TW; t += 4) {
    const int ti = t / TW, tj = t - ti * TW;
    float d[5][5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int ih = 2 * ti + off + a;
      const bool rok = (unsigned)ih < (unsigned)H;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int iw = 2 * tj + off + b;
        const bool ok = rok && (unsigned)iw < (unsigned)W;
        d[a][b] = ok ? pc[ih * W + iw] : 0.f;
      }
    }
    float tt[5][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]};
      float o[5];
      w24_bt(col, o);
#pragma unroll
      for (int i = 0; i < 5; ++i) tt[i][j] = o[i];
    }
    const size_t p = (size_t)n * TH * TW + t;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float o[5];
      w24_bt(tt[i], o);
#pragma unroll
      for (int j = 0; j < 5; ++j) V[((size_t)(i * 5 + j) * Pstride + p) * C + c0 + lane] = o[j];
    }
  }
  if (n == 0 && blockIdx.z == 0)                 // rows [P, Pstride): zero padding for the 128-wide GEMM tiles
    for (size_t p = P + g; p < Pstride; p += 4)
      for (int xi = 0; xi < 25; ++xi) V[((size_t)xi * Pstride + p) * C + c0 + lane] = 0.f;
}
void w24_input_pc(const float* x, float* V, int N, int C, int H, int W, int TH, int TW, int off, size_t Pstride, hipStream_t s) {
  const int HW = H * W;
  SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)N * C * HW + 25.0 * (double)Pstride * C));
  if (sg_opt(SG_OPT_W24_SMALL) && HW >= 144 && W <= 64 && C % 64 == 0) {
    // tile rows per workgroup: as few as keep ~2048 workgroups busy (each stages 2*rows + 3 input rows of 64 planes: <= ~24 KB)
    int rows_per = (int)(((long)(C / 64) * N * TH + 2047) / 2048);
    if (rows_per < 1) rows_per = 1;
    const int nrows = std::min(H, 2 * rows_per + 3);
    const int pitch = (nrows * W) | 1;
    const size_t lds = (size_t)64 * pitch * sizeof(float);
    if (lds <= 64 * 1024) {
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&w24_input_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(w24_input_small_kernel, dim3(C / 64, N, sg_cdiv(TH, rows_per)), dim3(256), lds, s, x, V, N, C, H, W, TH, TW, off,
                         Pstride, rows_per, pitch);
      return;
    }
  }
  hipLaunchKernelGGL(w24_input_kernel<0>, dim3(sg_cdiv(Pstride * C, 256)), dim3(256), 0, s, x, V, N, C, H, W, TH, TW, off, Pstride, 0, 1);
}

// U[xi][r][c] = (G g G^T)[xi]; mode 0: r = co, c = ci, g = w[co][ci]; mode 1 (data gradient): r = ci, c = co, g = w[co][ci] rotated
__global__ void w24_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int R, int Cc, int mode) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)R * Cc) return;
  const int c = (int)(idx % Cc), r = (int)(idx / Cc);
  const float* g = w + (mode ? ((size_t)c * R + r) : ((size_t)r * Cc + c)) * 16;
  float k[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i >> 2][i & 3] = mode ? g[15 - i] : g[i];
  float t[5][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float col[4] = {k[0][j], k[1][j], k[2][j], k[3][j]};
    float o[5];
    w24_g(col, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float o[5];
    w24_g(t[i], o);
#pragma unroll
    for (int j = 0; j < 5; ++j) U[((size_t)(i * 5 + j) * R + r) * Cc + c] = o[j];
  }
}

// y[n][m][2ti+a][2tj+b] = act((A^T Mx A)[a][b] + bias[m]),  Mx[m][xi*Pstride + p]; OH / OW may be odd (edge tiles are clipped)
__global__ void w24_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias, float* __restrict__ y, int N,
                                  int M, int OH, int OW, int TH, int TW, size_t Pstride, int act, float slope) {
  const size_t P = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * M) return;
  const size_t p = idx % P;
  const int m = (int)(idx / P);
  const float* src = Mx + (size_t)m * 25 * Pstride + p;
  float q[5][5];
#pragma unroll
  for (int i = 0; i < 25; ++i) q[i / 5][i % 5] = src[(size_t)i * Pstride];
  float sres[2][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float col[5] = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j]};
    float o[2];
    w24_at(col, o);
    sres[0][j] = o[0]; sres[1][j] = o[1];
  }
  const float b = bias ? bias[m] : 0.f;
  const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
  float* o = y + (((size_t)n * M + m) * OH + 2 * ti) * OW + 2 * tj;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    float v[2];
    w24_at(sres[a], v);
    if (2 * ti + a < OH) {
      o[a * OW] = sg_apply_act(v[0] + b, act, slope);
      if (2 * tj + 1 < OW) o[a * OW + 1] = sg_apply_act(v[1] + b, act, slope);
    }
  }
}

// Yt[xi*S + s][m][pc] = (A dy A^T)[xi] of the 2x2 gradient tile p = s*Pc + pc (zeros for p >= P and outside the plane)
__global__ void w24_gy_kernel(const float* __restrict__ gy, float* __restrict__ Yt, int N, int M, int OH, int OW, int TH, int TW,
                              size_t Pstride, int Pc, int S) {
  const size_t P = (size_t)N * TH * TW;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Pstride * M) return;
  const size_t p = idx % Pstride;
  const int m = (int)(idx / Pstride);
  float y2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  if (p < P) {
    const int n = (int)(p / (TH * TW)), r = (int)(p - (size_t)n * TH * TW), ti = r / TW, tj = r - ti * TW;
    const float* g = gy + (((size_t)n * M + m) * OH + 2 * ti) * OW + 2 * tj;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (2 * ti + a < OH && 2 * tj + b < OW) y2[a][b] = g[a * OW + b];
  }
  float t[5][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float col[2] = {y2[0][j], y2[1][j]};
    float o[5];
    w24_a(col, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i][j] = o[i];
  }
  const size_t sidx = p / (size_t)Pc, pc = p - sidx * (size_t)Pc;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float o[5];
    w24_a(t[i], o);
#pragma unroll
    for (int j = 0; j < 5; ++j) Yt[(((size_t)(i * 5 + j) * S + sidx) * M + m) * (size_t)Pc + pc] = o[j];
  }
}

// gw[m][c][4][4] = G^T (sum_s T[m][(xi*S + s)*C + c]) G
__global__ void w24_wgrad_output_kernel(const float* __restrict__ T, float* __restrict__ gw, int M, int C, int S) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)M * C) return;
  const int c = (int)(idx % C), m = (int)(idx / C);
  const float* src = T + (size_t)m * 25 * S * C + c;
  float q[5][5];
#pragma unroll
  for (int i = 0; i < 25; ++i) q[i / 5][i % 5] = 0.f;
  for (int z = 0; z < S; ++z) {                 // the 25 values of k-chunk z are fetched together; every sum still adds in ascending z
    float t[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) t[i] = src[((size_t)i * S + z) * C];
#pragma unroll
    for (int i = 0; i < 25; ++i) q[i / 5][i % 5] += t[i];
  }
  float t[4][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float col[5] = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j]};
    float o[4];
    w24_gt(col, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i][j] = o[i];
  }
  float* dst = gw + ((size_t)m * C + c) * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float o[4];
    w24_gt(t[i], o);
    *reinterpret_cast<float4*>(dst + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

struct W24Plan { int TH, TW, THd, TWd; size_t P, Ps, Pd, Pds; int S, Pc; };
bool w24_plan(const sgConvDesc* d, W24Plan* pl) {
  if (!d || d->KS != 4 || d->stride != 1 || d->C2 != 0 || d->upsample != 1 || d->pad_reflect || d->pad < 0 || d->pad > 3) return false;
  if (d->C1 % 128 != 0 || d->Cout % 128 != 0 || d->OH < 2 || d->OW < 2) return false;
  if (d->OH != d->H + 2 * d->pad - 3 || d->OW != d->W + 2 * d->pad - 3) return false;
  W24Plan p;
  p.TH = (d->OH + 1) / 2; p.TW = (d->OW + 1) / 2;         // forward / weight gradient: tiles of the output grid
  p.THd = (d->H + 1) / 2; p.TWd = (d->W + 1) / 2;         // data gradient: tiles of the input grid
  p.P = (size_t)d->N * p.TH * p.TW; p.Pd = (size_t)d->N * p.THd * p.TWd;
  p.Ps = (p.P + 127) / 128 * 128; p.Pds = (p.Pd + 127) / 128 * 128;
  // weight gradient: k-chunks of the tiles so that 25 * S * (M/128) * (C/128) workgroups fill the chip (measured at 512 x 256
  // channels, 2592 tiles: S = 2 -> 0.251 ms, 3 -> 0.261, 4 -> 0.276, 6 -> 0.277: longer k-loops beat more workgroups)
  const long t = 25L * (d->Cout / 128) * (d->C1 / 128);
  int S = (int)((400 + t - 1) / t);
  if (sg_opt(SG_OPT_W24_S) > 0) S = sg_opt(SG_OPT_W24_S);   // tuning aid
  const int maxS = (int)(p.P / 256) > 0 ? (int)(p.P / 256) : 1;
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  p.Pc = (int)(((p.P + S - 1) / S + 31) / 32 * 32);
  p.S = (int)((p.P + p.Pc - 1) / p.Pc);
  const double cm = d->C1 > d->Cout ? d->C1 : d->Cout;
  const double pm = (double)(p.Ps > p.Pds ? p.Ps : p.Pds) + 32.0 * p.S;
  if (!(25.0 * pm * cm < 2147483647.0 && 25.0 * (double)d->C1 * d->Cout < 2147483647.0)) return false;
  // the transforms move 25/4 x the activation bytes and the GEMMs need >= 2 column tiles: measured at 256 -> 512 channels, the
  // 6x6 maps of the third PatchGAN scale (288 tiles) still gain (forward 0.094 -> 0.064 ms, weight gradient 0.084 -> 0.062)
  if (p.P < (size_t)sg_opt(SG_OPT_W24_PMIN)) return false;
  if (pl) *pl = p;
  return true;
}
}  // namespace

extern "C" int sg_batched_gemm_nt(const float* a, const float* b, float* c, int nbatch, int M, int cols, int K, int tile,
                                  sgStream stream) {
  SG_ARG_CHECK(a && b && c && nbatch > 0 && M > 0 && cols > 0 && K > 0 && tile >= 0 && tile <= 3, "sg_batched_gemm_nt: bad arguments");
  const int bm = tile == 0 ? 128 : 64, bn = tile >= 2 ? 64 : 128;
  SG_ARG_CHECK(M % bm == 0 && cols % bn == 0 && K % 32 == 0, "sg_batched_gemm_nt: M, cols, K must be multiples of the tile (%d, %d, 32)", bm, bn);
  SG_ARG_CHECK(aligned16(a) && aligned16(b) && aligned16(c), "sg_batched_gemm_nt: operands must be 16-byte aligned");
  SG_ARG_CHECK((double)nbatch * M * K < SG_MAX_ELEMS && (double)nbatch * cols * K < SG_MAX_ELEMS, "sg_batched_gemm_nt: operand too large");
  // (the 64x64 tiles are the F(4x4,3x3) GEMMs: same launch path, incl. the chunked channel sum selected by w43_kfold)
  wino_bgemm_tile(tile, a, b, c, M, cols, K, 2.0 * nbatch * (double)M * cols * K, (hipStream_t)stream, nbatch,
                  tile >= 2 ? SG_K_WINO43_GEMM : SG_K_OTHER);
  SG_LAUNCH_CHECK("sg_batched_gemm_nt");
  return 0;
}

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
  // F(4x4,3x3) (wino43_shape; the switch can change between the query and the call: always room for both forms):
  // U[36][M][C] + max(V + Mx (forward), Ytp + G (data gradient), T + V + Ytp (weight gradient without saved operands))
  const size_t P4 = (size_t)d->N * (LH / 4) * (LW / 4), mx = M > C ? M : C;
  const size_t c = 36 * (M * C + 2 * P4 * mx + M * C);
  const size_t m = a > b ? (a > c ? a : c) : (b > c ? b : c);
  return m * sizeof(float) + 1024;
}

// gx [N, C1, H, W].  Reflection padding: Winograd over the (H+2) x (W+2) gradient of the reflect-padded input (correlation of
// the zero-extended gy with the rotated filter), then the reflection fold (sg_pad_upsample_bwd); 1.44x fewer MACs than the
// direct folded form.  Zero padding: the same correlation straight on the H x W grid (2.25x fewer MACs); behind a folded x2
// upsample the result is on the upsampled grid and is summed back 2x2.
static bool wino_adjoint_shape(const sgConvDesc* d) {
  return sg_opt(SG_OPT_WINO_ADJOINT) && wino_ok(d) && d->pad_reflect && d->upsample == 1 && d->H * d->W <= 256 && (d->H * d->W) % 4 == 0 &&
         d->C1 % 64 == 0 && d->Cout % 64 == 0;
}
// F(4x4,3x3) instead of F(2x2,3x3): the adjoint-form shapes whose planes split into 4x4 output tiles and whose tile count fills
// whole 64-column GEMM tiles
static bool wino43_shape(const sgConvDesc* d) {
  if (!sg_opt(SG_OPT_WINO43) || !wino_adjoint_shape(d) || d->H % 4 != 0 || d->W % 4 != 0) return false;
  const double P = (double)d->N * (d->H / 4) * (d->W / 4), Cmax = d->C1 > d->Cout ? d->C1 : d->Cout;
  return (long)P % 64 == 0 && 36.0 * P * Cmax < SG_MAX_ELEMS && 36.0 * d->C1 * d->Cout < SG_MAX_ELEMS;
}
// floats of the filter transform sg_conv2d_wino_fwd can hand to sg_conv2d_wino_dgrad (0: that conv's data gradient does not use
// it): F(2x2,3x3) -- the transposed twin UT[16][C1][Cout]; F(4x4,3x3) -- U[36][Cout][C1] itself (read x-contiguous)
extern "C" size_t sg_conv2d_wino_ut_floats(const sgConvDesc* d) {
  if (wino43_shape(d)) return (size_t)36 * d->C1 * d->Cout;
  return (wino_adjoint_shape(d) && d->C1 % 32 == 0 && d->Cout % 32 == 0) ? (size_t)16 * d->C1 * d->Cout : 0;
}

// floats of the input transform V / the gradient transform Ytp a conv can hand from its forward / data gradient to its weight
// gradient (0: that conv's weight gradient rebuilds its operands)
extern "C" size_t sg_conv2d_wino_v_floats(const sgConvDesc* d) {
  if (!sg_opt(SG_OPT_WINO_REUSE) || !wino_adjoint_shape(d)) return 0;
  if (wino43_shape(d)) return (size_t)36 * d->N * (d->H / 4) * (d->W / 4) * d->C1;
  return (size_t)16 * d->N * (d->H / 2) * (d->W / 2) * d->C1;
}
extern "C" size_t sg_conv2d_wino_ytp_floats(const sgConvDesc* d) {
  if (!sg_opt(SG_OPT_WINO_REUSE) || !wino_adjoint_shape(d)) return 0;
  if (wino43_shape(d)) return (size_t)36 * d->N * (d->H / 4) * (d->W / 4) * d->Cout;
  return (size_t)16 * d->N * (d->H / 2) * (d->W / 2) * d->Cout;
}

extern "C" int sg_conv2d_wino_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, const float* ut_saved,
                                    float* ytp_save, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_dgrad: unsupported desc");
  SG_ARG_CHECK(gy && w && gx && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_dgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->C1, K = d->Cout;                 // rows = input channels, reduction over output channels
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  const int refl = d->pad_reflect;
  // the form is a function of the desc ALONE (the saved-operand sizes sg_conv2d_wino_{ut,v,ytp}_floats are): an unaligned operand
  // on an F(4x4,3x3) shape is an argument error, not a silent switch to the F(2x2,3x3) layouts
  SG_ARG_CHECK(!wino43_shape(d) || (aligned16(gy) && aligned16(gx) && aligned16(w)),
               "sg_conv2d_wino_dgrad: F(4x4,3x3) shapes need 16-byte aligned gy / gx / w");
  if (wino43_shape(d)) {
    // F(4x4,3x3), adjoint form: Ytp = A gy A^T, G = Ytp x U (U from the forward, x-contiguous), overlap-add of B G B^T + fold
    const int HW = d->H * d->W;
    const size_t P = (size_t)d->N * (d->H / 4) * (d->W / 4);
    float* Uw = reinterpret_cast<float*>(ws);       // [36][Cout][C1]
    float* Ytp_ws = Uw + 36 * (size_t)M * K;        // [36][P][Cout]
    float* G = Ytp_ws + 36 * P * (size_t)(K > M ? K : M);      // [P][36][C1]
    float* Ytp = ytp_save ? ytp_save : Ytp_ws;
    const float* U = ut_saved;
    if (U == nullptr) {
      SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 45.0 * (double)M * K);
      hipLaunchKernelGGL(w43_weight_kernel, dim3((K / 32) * (M / 32)), dim3(256), 0, s, w, Uw, K, M);
      U = Uw;
    }
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * K * HW + 36.0 * (double)P * K));
      const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
      w43_lds_attr(&w43_gy_kernel, lds);
      hipLaunchKernelGGL(w43_gy_kernel, dim3(K / 64, d->N), dim3(256), lds, s, gy, Ytp, d->N, K, d->H, d->W); }
    w43_gemm_dgrad(Ytp, U, G, (int)P, M, K, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (36.0 * (double)P * M + (double)d->N * M * HW));
      const size_t lds = (size_t)64 * ((d->H + 2) * (d->W + 2) + 1 + HW + 1) * sizeof(float);
      w43_lds_attr(&w43_fold_kernel, lds);
      hipLaunchKernelGGL(w43_fold_kernel, dim3(M / 64, d->N), dim3(256), lds, s, (const float*)G, gx, d->N, M, d->H, d->W); }
    SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad");
    return 0;
  }
  if (wino_adjoint_shape(d) && aligned16(gy) && aligned16(gx)) {
    // adjoint Winograd over the output tiles (see wino_gy_small_kernel / wino_patch_fold_kernel)
    const int HW = d->H * d->W;
    const size_t P = (size_t)d->N * (d->H / 2) * (d->W / 2);
    float* UTw = reinterpret_cast<float*>(ws);      // [16][C1][Cout]
    float* Ytp_ws = UTw + 16 * (size_t)M * K;       // [16][P][Cout]
    float* G = Ytp_ws + 16 * P * K;                 // [P][16][C1]
    float* Ytp = ytp_save ? ytp_save : Ytp_ws;      // kept for the weight gradient of the same conv when the caller wants it
    const float* UT = ut_saved;                     // built by the forward pass of this step (same weights) when given
    if (UT == nullptr) {
      SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 25.0 * (double)M * K);
      wino_weight(w, UTw, M, K, 2, s);
      UT = UTw;
    }
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * K * HW + 16.0 * (double)P * K));
      const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gy_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wino_gy_small_kernel, dim3(K / 64, d->N), dim3(256), lds, s, gy, Ytp, d->N, K, d->H, d->W); }
    // G[p][xi*C1 + ci] = sum_co Ytp[xi][p][co] * UT[xi][ci][co]
    wino_bgemm(Ytp, UT, G, (int)P, M, K, 2.0 * M * (double)K * 16.0 * P, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (16.0 * (double)P * M + (double)d->N * M * HW));
      const size_t lds = (size_t)64 * ((d->H + 2) * (d->W + 2) + 1 + HW + 1) * sizeof(float);
      if (lds > 48 * 1024)
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_patch_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (sg_opt(SG_OPT_WINO_FOLD_CELLS)) {
        if (lds > 48 * 1024)
          hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_patch_fold_cells_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(wino_patch_fold_cells_kernel, dim3(M / 64, d->N), dim3(256), lds, s, (const float*)G, gx, d->N, M, d->H, d->W);
      } else {
        hipLaunchKernelGGL(wino_patch_fold_kernel, dim3(M / 64, d->N), dim3(64), lds, s, (const float*)G, gx, d->N, M, d->H, d->W);
      } }
    SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad");
    return 0;
  }
  SG_ARG_CHECK(ytp_save == nullptr, "sg_conv2d_wino_dgrad: ytp_save given but this desc does not run the adjoint form");
  const int TH = LH / 2 + (refl ? 1 : 0), TW = LW / 2 + (refl ? 1 : 0);
  const size_t Pd = wino_dgrad_tiles(d);
  float* U = reinterpret_cast<float*>(ws);          // [16][C1][Cout]
  float* V = U + 16 * (size_t)M * K;                // [16][Pd][Cout]
  float* Mx = V + 16 * Pd * K;                      // [C1][16][Pd]
  float* gpad = Mx + 16 * Pd * M;                   // [N][C1][LH+2][LW+2] (reflect) / [N][C1][LH][LW] (upsample)
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 25.0 * (double)M * K); wino_weight(w, U, M, K, 1, s); }
  wino_input_pc(gy, V, d->N, K, LH, LW, TH, TW, refl ? -2 : -1, 1, Pd, 0, s);
  wino_bgemm(U, V, Mx, M, (int)Pd, K, 2.0 * M * (double)K * 16.0 * ((double)d->N * TH * TW), s);   // flops of the real tiles
  const bool direct = !refl && d->upsample == 1;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (16.0 * (double)Pd * M + (double)d->N * M * 4.0 * TH * TW)); hipLaunchKernelGGL(wino_output_kernel, dim3(sg_cdiv((size_t)d->N * TH * TW * M, 256)), dim3(256), 0, s, (const float*)Mx,
                     (const float*)nullptr, direct ? gx : gpad, d->N, M, 2 * TH, 2 * TW, Pd, SG_ACT_NONE, 0.f); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad");
  if (direct) return 0;
  return sg_pad_upsample_bwd(gpad, gx, d->N * M, d->H, d->W, refl ? 1 : 0, d->upsample, stream);
}

extern "C" int sg_conv2d_wino_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                                  float slope, float* ut_save, float* v_save, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_fwd: unsupported desc");
  SG_ARG_CHECK(x && w && y && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  SG_ARG_CHECK(!wino43_shape(d) || (aligned16(x) && aligned16(y) && aligned16(w)),
               "sg_conv2d_wino_fwd: F(4x4,3x3) shapes need 16-byte aligned x / y / w");
  if (wino43_shape(d)) {
    // F(4x4,3x3): U = G g G^T (into ut_save when the data gradient follows), V = B^T d B, 36 GEMMs, y = A^T Mx A + bias
    const int HW = d->H * d->W;
    const size_t P = (size_t)d->N * (d->H / 4) * (d->W / 4);
    float* U_ws = reinterpret_cast<float*>(ws);
    float* V_ws = U_ws + 36 * (size_t)M * C;
    float* Mx = V_ws + 36 * P * (size_t)(C > M ? C : M);
    float* U = ut_save ? ut_save : U_ws;
    float* V = v_save ? v_save : V_ws;
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 45.0 * (double)M * C);
      hipLaunchKernelGGL(w43_weight_kernel, dim3((M / 32) * (C / 32)), dim3(256), 0, s, w, U, M, C); }
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * C * HW + 36.0 * (double)P * C));
      const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
      w43_lds_attr(&w43_input_kernel, lds);
      hipLaunchKernelGGL(w43_input_kernel, dim3(C / 64, d->N), dim3(256), lds, s, x, V, d->N, C, d->H, d->W); }
    w43_gemm_fwd(U, V, Mx, M, (int)P, C, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (36.0 * (double)P * M + (double)d->N * M * HW));
      hipLaunchKernelGGL(w43_output_kernel, dim3(sg_cdiv(P * M, 256)), dim3(256), 0, s, (const float*)Mx, bias, y, d->N, M, d->H, d->W,
                         act, slope); }
    SG_LAUNCH_CHECK("sg_conv2d_wino_fwd");
    return 0;
  }
  const size_t P = (size_t)d->N * (LH / 2) * (LW / 2);
  float* U = reinterpret_cast<float*>(ws);
  float* V_ws = U + 16 * (size_t)M * C;
  float* Mx = V_ws + 16 * P * C;
  SG_ARG_CHECK(v_save == nullptr || sg_conv2d_wino_v_floats(d) > 0, "sg_conv2d_wino_fwd: v_save given but unused by this desc");
  float* V = v_save ? v_save : V_ws;               // [16][P][C1]: kept for the weight gradient when the caller wants it
  SG_ARG_CHECK(ut_save == nullptr || sg_conv2d_wino_ut_floats(d) > 0, "sg_conv2d_wino_fwd: ut_save given but unused by this desc");
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (25.0 + (ut_save ? 16.0 : 0.0)) * (double)M * C);
    const bool wrote = wino_weight(w, U, M, C, 0, s, ut_save);
    SG_ARG_CHECK(ut_save == nullptr || wrote, "sg_conv2d_wino_fwd: the transposed filter transform needs the LDS weight kernel"); }
  wino_input_pc(x, V, d->N, C, LH, LW, LH / 2, LW / 2, -1, d->pad_reflect ? 0 : 1, P, d->upsample == 2 ? 1 : 0, s);
  wino_bgemm(U, V, Mx, M, (int)P, C, 2.0 * M * (double)C * 16.0 * P, s);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (16.0 * (double)P * M + (double)d->N * M * LH * LW)); hipLaunchKernelGGL(wino_output_kernel, dim3(sg_cdiv(P * M, 256)), dim3(256), 0, s, (const float*)Mx, bias, y, d->N, M, LH, LW, P,
                     act, slope); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_fwd");
  return 0;
}

extern "C" int sg_conv2d_wino_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, const float* v_saved,
                                    const float* ytp_saved, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(wino_ok(d), "sg_conv2d_wino_wgrad: unsupported desc");
  SG_ARG_CHECK(gy && x && gw && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d), "sg_conv2d_wino_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  const int LH = d->H * d->upsample, LW = d->W * d->upsample;
  const size_t P = (size_t)d->N * (LH / 2) * (LW / 2);
  SG_ARG_CHECK(!wino43_shape(d) || (aligned16(x) && aligned16(gy)),
               "sg_conv2d_wino_wgrad: F(4x4,3x3) shapes need 16-byte aligned x / gy");
  if (wino43_shape(d)) {
    // F(4x4,3x3): T = Ytp^T x V over the tiles, gw = G^T T G; the operands come from this conv's forward / data gradient when
    // the caller kept them, else they are rebuilt here
    const int HW = d->H * d->W;
    const size_t P4 = (size_t)d->N * (d->H / 4) * (d->W / 4);
    float* T4 = reinterpret_cast<float*>(ws);       // [M][36][C]
    const float* V = v_saved;
    const float* Ytp = ytp_saved;
    if (!(V && Ytp)) {
      float* Vw = T4 + 36 * (size_t)M * C;
      float* Yw = Vw + 36 * P4 * (size_t)(C > M ? C : M);
      const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
      { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * C * HW + 36.0 * (double)P4 * C));
        w43_lds_attr(&w43_input_kernel, lds);
        hipLaunchKernelGGL(w43_input_kernel, dim3(C / 64, d->N), dim3(256), lds, s, x, Vw, d->N, C, d->H, d->W); }
      { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * M * HW + 36.0 * (double)P4 * M));
        w43_lds_attr(&w43_gy_kernel, lds);
        hipLaunchKernelGGL(w43_gy_kernel, dim3(M / 64, d->N), dim3(256), lds, s, gy, Yw, d->N, M, d->H, d->W); }
      V = Vw; Ytp = Yw;
    }
    w43_gemm_wgrad(Ytp, V, T4, M, C, (int)P4, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 45.0 * (double)M * C);
      hipLaunchKernelGGL(w43_wgrad_output_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, (const float*)T4, gw, M, C); }
    SG_LAUNCH_CHECK("sg_conv2d_wino_wgrad");
    return 0;
  }
  float* T = reinterpret_cast<float*>(ws);          // [M][16][C]
  if (v_saved && ytp_saved) {
    // operands already built by the forward (V) and the adjoint data gradient (Ytp) of this conv in this step
    SG_ARG_CHECK(sg_conv2d_wino_v_floats(d) > 0 && P % 32 == 0, "sg_conv2d_wino_wgrad: saved operands given but unused by this desc");
    wino_bgemm_x(ytp_saved, v_saved, T, M, C, (int)P, 2.0 * M * (double)C * 16.0 * P, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 25.0 * (double)M * C); hipLaunchKernelGGL(wino_wgrad_output_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, (const float*)T, gw, M, C); }
    SG_LAUNCH_CHECK("sg_conv2d_wino_wgrad");
    return 0;
  }
  float* Vp = T + 16 * (size_t)M * C;               // [16][C][P]
  float* Yt = Vp + 16 * P * C;                      // [16][M][P]
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * C * d->H * d->W + 16.0 * (double)P * C)); hipLaunchKernelGGL(wino_input_kernel<1>, dim3(sg_cdiv(P * C, 256)), dim3(256), 0, s, x, Vp, d->N, C, LH, LW, LH / 2, LW / 2, -1,
                     d->pad_reflect ? 0 : 1, P, d->upsample == 2 ? 1 : 0); }
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * M * LH * LW + 16.0 * (double)P * M)); hipLaunchKernelGGL(wino_gy_kernel, dim3(sg_cdiv(P * M, 256)), dim3(256), 0, s, gy, Yt, d->N, M, LH, LW); }
  wino_bgemm(Yt, Vp, T, M, C, (int)P, 2.0 * M * (double)C * 16.0 * P, s);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 25.0 * (double)M * C); hipLaunchKernelGGL(wino_wgrad_output_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, (const float*)T, gw, M, C); }
  SG_LAUNCH_CHECK("sg_conv2d_wino_wgrad");
  return 0;
}

// ---- F(4x4,3x3) conv + InstanceNorm (see w43_output_in_kernel / w43_gy_in_kernel) ------------------------------------------------
extern "C" int sg_conv2d_wino_in_supported(const sgConvDesc* d) {
  if (!sg_opt(SG_OPT_WINO_IN_FUSE) || !wino43_shape(d)) return 0;
  const int NT = (d->H / 4) * (d->W / 4);
  return (NT <= 16 && sg_opt(SG_OPT_WINO_REUSE)) ? 1 : 0;      // <= 4 tiles per thread: planes up to 16x16
}

extern "C" int sg_conv2d_wino_fwd_instnorm(const sgConvDesc* d, const float* x, const float* w, const float* bias, const float* skip,
                                           float* ypre, float* out, float* mean, float* rstd, float eps, int act, float slope,
                                           float* ut_save, float* v_save, void* ws, size_t ws_bytes, sgStream stream) {
  SG_ARG_CHECK(sg_conv2d_wino_in_supported(d), "sg_conv2d_wino_fwd_instnorm: unsupported desc");
  SG_ARG_CHECK(x && w && ypre && out && mean && rstd && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d),
               "sg_conv2d_wino_fwd_instnorm: bad arguments");
  SG_ARG_CHECK(aligned16(x) && aligned16(w) && aligned16(ypre) && aligned16(out) && (!skip || aligned16(skip)),
               "sg_conv2d_wino_fwd_instnorm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1, HW = d->H * d->W, NT = (d->H / 4) * (d->W / 4);
  const size_t P = (size_t)d->N * NT;
  float* U_ws = reinterpret_cast<float*>(ws);
  float* V_ws = U_ws + 36 * (size_t)M * C;
  float* Mx = V_ws + 36 * P * (size_t)(C > M ? C : M);        // [P][36][M]
  float* U = ut_save ? ut_save : U_ws;
  float* V = v_save ? v_save : V_ws;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 45.0 * (double)M * C);
    hipLaunchKernelGGL(w43_weight_kernel, dim3((M / 32) * (C / 32)), dim3(256), 0, s, w, U, M, C); }
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * C * HW + 36.0 * (double)P * C));
    const size_t lds = (size_t)64 * (HW + 1) * sizeof(float);
    w43_lds_attr(&w43_input_kernel, lds);
    hipLaunchKernelGGL(w43_input_kernel, dim3(C / 64, d->N), dim3(256), lds, s, x, V, d->N, C, d->H, d->W); }
  // tile-major result Mx[p][xi*M + m] = sum_c V[xi][p][c] U[xi][m][c]: the same GEMM with the operand roles swapped
  wino_bgemm_tile(w43_tile(), V, U, Mx, (int)P, M, C, 2.0 * 36.0 * M * (double)P * C, s, 36, SG_K_WINO43_GEMM);
  { SgProfScope xf(SG_K_INSTNORM, s, 0, 4.0 * (36.0 * (double)P * M + (double)d->N * M * HW * (skip ? 3.0 : 2.0)));
    const size_t lds = (size_t)(128 * (HW + 1) + 256) * sizeof(float);
    if (NT <= 4) {
      w43_lds_attr(&w43_output_in_kernel<1>, lds);
      hipLaunchKernelGGL(w43_output_in_kernel<1>, dim3(M / 64, d->N), dim3(256), lds, s, (const float*)Mx, bias, skip, ypre, out, mean,
                         rstd, d->N, M, d->H, d->W, eps, act, slope);
    } else {
      w43_lds_attr(&w43_output_in_kernel<4>, lds);
      hipLaunchKernelGGL(w43_output_in_kernel<4>, dim3(M / 64, d->N), dim3(256), lds, s, (const float*)Mx, bias, skip, ypre, out, mean,
                         rstd, d->N, M, d->H, d->W, eps, act, slope);
    } }
  SG_LAUNCH_CHECK("sg_conv2d_wino_fwd_instnorm");
  return 0;
}

extern "C" int sg_conv2d_wino_dgrad_instnorm(const sgConvDesc* d, const float* gout, const float* ypre, const float* mean,
                                             const float* rstd, int act, float slope, const float* w, float* gconv, float* gx,
                                             float* gb, const float* ut_saved, float* ytp_save, void* ws, size_t ws_bytes,
                                             sgStream stream) {
  SG_ARG_CHECK(sg_conv2d_wino_in_supported(d), "sg_conv2d_wino_dgrad_instnorm: unsupported desc");
  SG_ARG_CHECK(gout && ypre && mean && rstd && w && gconv && ws && ws_bytes >= sg_conv2d_wino_ws_bytes(d),
               "sg_conv2d_wino_dgrad_instnorm: bad arguments");
  SG_ARG_CHECK(aligned16(gout) && aligned16(ypre) && aligned16(gconv) && aligned16(w) && (!gx || aligned16(gx)),
               "sg_conv2d_wino_dgrad_instnorm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->C1, K = d->Cout, HW = d->H * d->W;
  const size_t P = (size_t)d->N * (d->H / 4) * (d->W / 4);
  float* Uw = reinterpret_cast<float*>(ws);
  float* Ytp_ws = Uw + 36 * (size_t)M * K;
  float* G = Ytp_ws + 36 * P * (size_t)(K > M ? K : M);
  float* Ytp = ytp_save ? ytp_save : Ytp_ws;
  // [N][Cout] partial bias gradients: behind G in the workspace (the weight gradient's T region: free during this call)
  float* gb_part = gb ? G + 36 * P * (size_t)M : nullptr;
  { SgProfScope xf(SG_K_INSTNORM_BWD, s, 0, 4.0 * ((double)d->N * K * HW * 3.0 + 36.0 * (double)P * K));
    const size_t lds = (size_t)(128 * (HW + 1) + 768) * sizeof(float);
    w43_lds_attr(&w43_gy_in_kernel, lds);
    hipLaunchKernelGGL(w43_gy_in_kernel, dim3(K / 64, d->N), dim3(256), lds, s, gout, ypre, mean, rstd, gconv, Ytp, gb_part, d->N, K,
                       d->H, d->W, act, slope);
    if (gb) hipLaunchKernelGGL(w43_bias_sum_kernel, dim3(sg_cdiv(K, 256)), dim3(256), 0, s, (const float*)gb_part, gb, d->N, K); }
  if (gx) {
    const float* U = ut_saved;
    if (U == nullptr) {
      SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 45.0 * (double)M * K);
      hipLaunchKernelGGL(w43_weight_kernel, dim3((K / 32) * (M / 32)), dim3(256), 0, s, w, Uw, K, M);
      U = Uw;
    }
    w43_gemm_dgrad(Ytp, U, G, (int)P, M, K, s);
    { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (36.0 * (double)P * M + (double)d->N * M * HW));
      const size_t lds = (size_t)64 * ((d->H + 2) * (d->W + 2) + 1 + HW + 1) * sizeof(float);
      w43_lds_attr(&w43_fold_kernel, lds);
      hipLaunchKernelGGL(w43_fold_kernel, dim3(M / 64, d->N), dim3(256), lds, s, (const float*)G, gx, d->N, M, d->H, d->W); }
  }
  SG_LAUNCH_CHECK("sg_conv2d_wino_dgrad_instnorm");
  return 0;
}

// ---- Winograd F(2x2, 4x4): see w24_* above ----------------------------------------------------------------------------------
extern "C" int sg_conv2d_wino24_supported(const sgConvDesc* d) {
  return (sg_opt(SG_OPT_WINO24) && w24_plan(d, nullptr)) ? 1 : 0;
}
extern "C" size_t sg_conv2d_wino24_ws_bytes(const sgConvDesc* d) {
  W24Plan p;
  if (!w24_plan(d, &p)) return 0;
  const size_t M = d->Cout, C = d->C1;
  const size_t f = 25 * (M * C + p.Ps * C + p.Ps * M);
  const size_t g = 25 * (M * C + p.Pds * M + p.Pds * C);
  const size_t w = 25 * (size_t)p.S * ((size_t)p.Pc * (M + C) + M * C);
  const size_t mx = f > g ? (f > w ? f : w) : (g > w ? g : w);
  return mx * sizeof(float) + 256;
}
extern "C" int sg_conv2d_wino24_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                                    float slope, void* ws, size_t ws_bytes, sgStream stream) {
  W24Plan p;
  SG_ARG_CHECK(w24_plan(d, &p), "sg_conv2d_wino24_fwd: unsupported desc");
  SG_ARG_CHECK(x && w && y && ws && ws_bytes >= sg_conv2d_wino24_ws_bytes(d), "sg_conv2d_wino24_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  float* U = reinterpret_cast<float*>(ws);
  float* V = U + (size_t)25 * M * C;
  float* Mx = V + 25 * p.Ps * C;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 26.0 * (double)M * C);
    hipLaunchKernelGGL(w24_weight_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, w, U, M, C, 0); }
  w24_input_pc(x, V, d->N, C, d->H, d->W, p.TH, p.TW, -d->pad, p.Ps, s);
  wino_bgemm(U, V, Mx, M, (int)p.Ps, C, 2.0 * M * (double)C * 25.0 * p.P, s, 25);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (25.0 * (double)p.P * M + (double)d->N * M * d->OH * d->OW));
    hipLaunchKernelGGL(w24_output_kernel, dim3(sg_cdiv(p.P * M, 256)), dim3(256), 0, s, (const float*)Mx, bias, y, d->N, M, d->OH,
                       d->OW, p.TH, p.TW, p.Ps, act, slope); }
  SG_LAUNCH_CHECK("sg_conv2d_wino24_fwd");
  return 0;
}
extern "C" int sg_conv2d_wino24_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                                      sgStream stream) {
  W24Plan p;
  SG_ARG_CHECK(w24_plan(d, &p), "sg_conv2d_wino24_dgrad: unsupported desc");
  SG_ARG_CHECK(gy && w && gx && ws && ws_bytes >= sg_conv2d_wino24_ws_bytes(d), "sg_conv2d_wino24_dgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->C1, K = d->Cout;             // gx[ci] = sum_co rot(w[co][ci]) * gy[co], padding 3 - pad
  float* U = reinterpret_cast<float*>(ws);
  float* V = U + (size_t)25 * M * K;
  float* Mx = V + 25 * p.Pds * K;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * 26.0 * (double)M * K);
    hipLaunchKernelGGL(w24_weight_kernel, dim3(sg_cdiv((size_t)M * K, 256)), dim3(256), 0, s, w, U, M, K, 1); }
  w24_input_pc(gy, V, d->N, K, d->OH, d->OW, p.THd, p.TWd, -(3 - d->pad), p.Pds, s);
  wino_bgemm(U, V, Mx, M, (int)p.Pds, K, 2.0 * M * (double)K * 25.0 * p.Pd, s, 25);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (25.0 * (double)p.Pd * M + (double)d->N * M * d->H * d->W));
    hipLaunchKernelGGL(w24_output_kernel, dim3(sg_cdiv(p.Pd * M, 256)), dim3(256), 0, s, (const float*)Mx, (const float*)nullptr, gx,
                       d->N, M, d->H, d->W, p.THd, p.TWd, p.Pds, SG_ACT_NONE, 0.f); }
  SG_LAUNCH_CHECK("sg_conv2d_wino24_dgrad");
  return 0;
}
extern "C" int sg_conv2d_wino24_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                                      sgStream stream) {
  W24Plan p;
  SG_ARG_CHECK(w24_plan(d, &p), "sg_conv2d_wino24_wgrad: unsupported desc");
  SG_ARG_CHECK(gy && x && gw && ws && ws_bytes >= sg_conv2d_wino24_ws_bytes(d), "sg_conv2d_wino24_wgrad: bad arguments");
  SG_ARG_CHECK(aligned16(gw), "sg_conv2d_wino24_wgrad: gw must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int M = d->Cout, C = d->C1;
  const size_t Pall = (size_t)p.S * p.Pc;       // tiles incl. the zero padding of the last k-chunk
  float* Yt = reinterpret_cast<float*>(ws);
  float* V = Yt + 25 * Pall * M;
  float* T = V + 25 * Pall * C;
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * M * d->OH * d->OW + 25.0 * (double)Pall * M));
    hipLaunchKernelGGL(w24_gy_kernel, dim3(sg_cdiv(Pall * M, 256)), dim3(256), 0, s, gy, Yt, d->N, M, d->OH, d->OW, p.TH, p.TW, Pall,
                       p.Pc, p.S); }
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * ((double)d->N * C * d->H * d->W + 25.0 * (double)Pall * C));
    hipLaunchKernelGGL(w24_input_kernel<1>, dim3(sg_cdiv(Pall * C, 256)), dim3(256), 0, s, x, V, d->N, C, d->H, d->W, p.TH, p.TW,
                       -d->pad, Pall, p.Pc, p.S); }
  wino_bgemm(Yt, V, T, M, C, p.Pc, 2.0 * M * (double)C * 25.0 * p.P, s, 25 * p.S);
  { SgProfScope xf(SG_K_WINO_XFORM, s, 0, 4.0 * (25.0 * p.S + 16.0) * (double)M * C);
    hipLaunchKernelGGL(w24_wgrad_output_kernel, dim3(sg_cdiv((size_t)M * C, 256)), dim3(256), 0, s, (const float*)T, gw, M, C, p.S); }
  SG_LAUNCH_CHECK("sg_conv2d_wino24_wgrad");
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
  const int deep = sg_opt(SG_OPT_LINEAR_NSUB);
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
  sgk::t_alg_bytes = 4.0 * ((double)rows * in_f + (double)in_f * out_f + (double)rows * out_f);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{y, b, rows, out_f, out_f, act, slope, 0};
  const bool vec = (in_f % 4 == 0) && aligned16(x) && aligned16(w);
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
  if (sgk::skinny_shape(rows, out_f))          // small layer: latency-bound, register-streaming kernel (skinny.hip)
    sgk::skinny_gemm(x, in_f, 1, w, in_f, 1, y, b, nullptr, rows, out_f, in_f, act, slope, s);
  else if (vec)
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
  sgk::t_alg_bytes = 4.0 * ((double)rows * in_f + (double)in_f * out_f + (double)rows * out_f);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{gx, nullptr, rows, in_f, in_f, SG_ACT_NONE, 0.f, 0};
  const bool vec = (out_f % 4 == 0) && aligned16(gy);
  SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
  if (sgk::skinny_shape(rows, in_f))           // gx[rows][in_f] = sum_o gy[row][o] w[o][in_f]: B(n = i, k = o) = w[o*in_f + i]
    sgk::skinny_gemm(gy, out_f, 1, w, in_f, 0, gx, nullptr, nullptr, rows, in_f, out_f, SG_ACT_NONE, 0.f, s);
  else if (vec)
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
  sgk::t_alg_bytes = 4.0 * ((double)rows * in_f + (double)in_f * out_f + (double)rows * out_f);
  hipStream_t s = (hipStream_t)stream;
  EpRowMajor ep{gw, nullptr, out_f, in_f, in_f, SG_ACT_NONE, 0.f, 0};
  {
    SgProfScope prof(SG_K_LINEAR, s, 2.0 * rows * (double)in_f * out_f, 0);
    if (sgk::skinny_shape(out_f, in_f)) {      // gw[o][i] = sum_row gy[row][o] x[row][i]: both operands row-index-major;
      // the bias gradient (column sums of gy = row sums of the A operand) comes out of the same launch
      sgk::skinny_gemm(gy, out_f, 0, x, in_f, 0, gw, nullptr, gb, out_f, in_f, rows, SG_ACT_NONE, 0.f, s);
      gb = nullptr;
    } else
      run_dense(LoadXContig<64>{gy, out_f, out_f}, LoadXContig<64>{x, in_f, in_f}, LoadXContig<32>{gy, out_f, out_f},
                LoadXContig<128>{x, in_f, in_f}, ep, out_f, in_f, rows, s);
  }
  SG_LAUNCH_CHECK("sg_linear_bwd_weight");
  if (gb) return sg_channel_sum(gy, gb, rows, out_f, 1, nullptr, 0, stream);   // column sums of gy[rows][out_f]
  return 0;
}


#ifdef SG_TIMELINE
// debugging build only (tools/probe/build_timeline.sh): hand this translation unit's igemm_kernel instantiations a stamp buffer of
// ``cap`` workgroups x 8 x u64 (nullptr: off)
extern "C" int sg_debug_timeline_set_igemm(void* buf, unsigned cap) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl), &p, sizeof(p)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_sg_tl_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
  return 0;
}
#endif
