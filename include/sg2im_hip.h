/*
 * sg2im_hip.h -- C ABI of libsg2im_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the
 * scene-graph -> image G+D training step.
 *
 * The reference (ashual/scene_generation) has no FFI: its hot path dispatches PyTorch ops.  Each entry
 * point below replaces the op group a reference call site dispatches (cited per function, paths
 * relative to /root/reference/scene_generation/).  The only caller is the Python host layer
 * (scene_generation_amd/ops/, ctypes); INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - caller owns every buffer (incl. workspace); no allocation, no host sync, no stream creation inside
 *   - all pointers are DEVICE pointers unless the name ends in _host; tensors are dense row-major,
 *     images NCHW fp32, indices int64 (graph/crop) or int32 (CSR / segment offsets / plans)
 *   - `stream` is a hipStream_t passed as void*; every launch goes to that stream, in order
 *   - return 0 = OK; <0 = argument error detected before launch; >0 = hipError_t from the launch
 *     sg_last_error_string() describes the last non-zero return on the calling thread
 *   - thread-safe for distinct streams.  Global state, all of it: the opt-in profiler (sg_prof_*), the shape-table cache
 *     (sg_plan_cache_*: mutex-guarded) and the option table (sg_set_option: atomic ints, initialised once at load time)
 */
#ifndef SG2IM_HIP_H
#define SG2IM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sgStream;

/* activation codes fused into epilogues */
enum { SG_ACT_NONE = 0, SG_ACT_RELU = 1, SG_ACT_LEAKY = 2, SG_ACT_TANH = 3, SG_ACT_SIGMOID = 4 };
/* scalar loss kinds (sg_loss_fwd / sg_loss_bwd) */
enum { SG_LOSS_MSE_CONST = 0, SG_LOSS_MSE = 1, SG_LOSS_L1 = 2, SG_LOSS_BCE_LOGITS_CONST = 3,
       SG_LOSS_MEAN = 4,               /* sum a_i                      (wgan_*_loss, losses.py:93-112) */
       SG_LOSS_MSE_SIGMOID_CONST = 5,  /* (sigmoid(a_i) - target)^2    (lsgan_*_loss, losses.py:115-132) */
       SG_LOSS_BCE_PROB_CONST = 6 };   /* nn.BCELoss vs a constant     (GANLoss(use_lsgan=False), losses.py:147) */
#define SG_WSUM_MAX 32
/* return code of an index operand outside its range (the reference raises IndexError: graph.py:79-80, model.py:131) */
#define SG_ERR_INDEX (-2)

int sg_version(void);
const char* sg_last_error_string(void);
/* Tuning / debugging switches (which kernel variant or threshold a launch plan uses; results stay within the tolerances of
 * the parity suite for every setting).  Each switch has a compiled-in default and is initialised ONCE, when the library is
 * loaded, from the environment variable SG_<NAME> (upper case) if that is set; afterwards the library never reads the
 * environment.  sg_set_option stores atomically: launches planned after the call see the new value.  Names:
 * sg_option_name(0 .. sg_num_options()-1).  Returns -1 for an unknown name. */
int sg_num_options(void);
const char* sg_option_name(int index);
int sg_option_default(int index);
int sg_get_option(const char* name, int* value);
int sg_set_option(const char* name, int value);
/* device bytes held by the shape-table cache; drop it (synchronises the tables' build events) */
size_t sg_plan_cache_bytes(void);
int sg_plan_cache_clear(void);

/* ---------------------------------------------------------------------------------------------
 * Convolution family = implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
 * Replaces nn.Conv2d / nn.ConvTranspose2d (+ the nn.ReflectionPad2d, Interpolate(x2, nearest) and
 * torch.cat((layout, img), 1) feeding them) at generators.py:20-27,68-89, layers.py:160-180,251-270,
 * discriminators.py:137-158,215-234 and trainer.py:246,250,328.
 * ------------------------------------------------------------------------------------------- */
typedef struct sgConvDesc {
  int32_t N;            /* images */
  int32_t C1, C2;       /* input channels taken from x1 / x2 (channel concat folded into the gather; C2=0 if none) */
  int32_t H, W;         /* stored input spatial size */
  int32_t Cout;
  int32_t KS;           /* square kernel: 1, 3, 4 or 7 */
  int32_t stride;       /* 1 or 2 */
  int32_t pad;
  int32_t pad_reflect;  /* 0 zero padding, 1 reflection padding (nn.ReflectionPad2d folded in) */
  int32_t upsample;     /* 1, or 2 = nearest x2 upsample of the input folded in (layers.py:304-314) */
  int32_t OH, OW;       /* output spatial size */
  int32_t out_pad;      /* conv-transpose only: output_padding */
  int32_t x2_broadcast; /* 1: x2 is [N, C2] and is broadcast over H x W (the one-hot class map of discriminators.py:107-110) */
} sgConvDesc;

/* scratch the caller must provide (kind 0: conv / convT forward, 1: conv / convT dgrad, 2: conv / convT wgrad) */
size_t sg_conv2d_ws_bytes(const sgConvDesc* d, int kind);
/* y[N,Cout,OH,OW] = act(conv(x) + bias) ; w [Cout, C1+C2, KS, KS] */
int sg_conv2d_fwd(const sgConvDesc* d, const float* x1, const float* x2, const float* w, const float* bias,
                  float* y, int act, float slope, void* ws, size_t ws_bytes, sgStream stream);
/* gx[N, c_end-c_begin, Hg, Wg]: gradient w.r.t. input channels [c_begin, c_end) of the *padded/upsampled*
 * logical input: Hg = H*upsample + (pad_reflect ? 2*pad : 0).  Fold with sg_pad_upsample_bwd. */
int sg_conv2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, int c_begin, int c_end,
                    void* ws, size_t ws_bytes, sgStream stream);
/* Same gathers with PER-IMAGE weights wimg / gwimg [N, Cout, L, KS, KS] (position j of image n's list <-> channel
   chan_list[n][j]): the factored form of a conv over a masks_to_layout() layout, whose channels are linear combinations of
   the per-object sampled-mask planes (layout = sum_o vecs[o] (x) S_o, reference layout.py:85-86), so
   conv(layout) = sum_o (sum_c vecs[o][c] W[:, c]) * S_o: the "channels" become the <= 9 objects of the image. */
int sg_conv2d_fwd_perimage(const sgConvDesc* d, const float* x1, const float* x2, const float* wimg, const float* bias,
                           const int32_t* chan_list, const int32_t* chan_cnt, int L, float* y, int act, float slope,
                           void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_wgrad_perimage(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                             const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gwimg, void* ws,
                             size_t ws_bytes, sgStream stream);
/* Data gradient w.r.t. the ACTUAL [N, c_end-c_begin, H, W] input of ReflectionPad2d(1) + 3x3 stride-1 conv (the
   ResnetBlock convs, reference layers.py:251-270): the reflection fold is applied to gy (one pre-folded copy per tap)
   instead of computing the gradient on the padded (H+2)x(W+2) grid and folding it with sg_pad_upsample_bwd. */
int sg_conv2d_dgrad_folded_supported(const sgConvDesc* d);
size_t sg_conv2d_dgrad_folded_ws_bytes(const sgConvDesc* d);
int sg_conv2d_dgrad_folded(const sgConvDesc* d, const float* gy, const float* w, float* gx, int c_begin, int c_end,
                           void* ws, size_t ws_bytes, sgStream stream);
/* gw[Cout, C1+C2, KS, KS] (+ gb[Cout] if non-null) */
int sg_conv2d_wgrad(const sgConvDesc* d, const float* gy, const float* x1, const float* x2, float* gw, float* gb,
                    void* ws, size_t ws_bytes, sgStream stream);
/* Channel-sparse variants for a layer whose input is a masks_to_layout() layout (reference model.py:165-168 and
   layout.py:64-93: per image only the one-hot planes of the classes present and the dense representation block are
   non-zero).  chan_list [N, L] int32: ascending concat-channel ids (over x1|x2) that may be non-zero in image n, the
   first chan_cnt[n] (<= L) entries are used; every other channel of that image MUST be all-zero or the result
   differs from sg_conv2d_fwd / sg_conv2d_wgrad.  Same outputs as the dense calls (different summation order). */
size_t sg_conv2d_sparse_ws_bytes(const sgConvDesc* d, int L, int kind /* 0 fwd, 2 wgrad */);
int sg_conv2d_fwd_sparse(const sgConvDesc* d, const float* x1, const float* x2, const float* w, const float* bias,
                         const int32_t* chan_list, const int32_t* chan_cnt, int L, float* y, int act, float slope,
                         void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_wgrad_sparse(const sgConvDesc* d, const float* gy, const float* x1, const float* x2,
                           const int32_t* chan_list, const int32_t* chan_cnt, int L, float* gw, float* gb,
                           void* ws, size_t ws_bytes, sgStream stream);
/* Winograd F(2x2, 3x3) forward / data gradient / weight gradient for 3x3 stride-1 pad-1 convs (reflection padding: the
   ResnetBlock convs, layers.py:251-270; zero padding: the VGG19 convs of VGGLoss, losses.py:183-198, and -- behind the folded
   nearest x2 upsample -- mask_net, generators.py:20-22) with >= 128 channels on both sides, whose channel counts and tile count
   N*(OH/2)*(OW/2) are all multiples of 128 or all multiples of 64: 2.25x fewer MACs; fp32 throughout, results agree with
   sg_conv2d_fwd / _dgrad / _wgrad to fp32 rounding (gb via sg_channel_sum). */
int sg_conv2d_wino_supported(const sgConvDesc* d);
size_t sg_conv2d_wino_ws_bytes(const sgConvDesc* d);
/* ut_save (optional, sg_conv2d_wino_ut_floats(d) floats): the forward also writes the filter transform with the channel
   roles swapped -- what sg_conv2d_wino_dgrad of the SAME conv multiplies with (ut_saved): the weights are transformed once
   per step instead of once per direction. */
size_t sg_conv2d_wino_ut_floats(const sgConvDesc* d);
/* v_save / ytp_save (optional, sg_conv2d_wino_v_floats(d) / sg_conv2d_wino_ytp_floats(d) floats; 0 = not applicable to this
   desc): the forward keeps its input transform V[16][P][C1], the data gradient its gradient transform Ytp[16][P][Cout], and
   sg_conv2d_wino_wgrad of the SAME conv in the same step, given both (v_saved, ytp_saved), runs its 16 GEMMs straight on them
   instead of transforming x and gy a second time. */
size_t sg_conv2d_wino_v_floats(const sgConvDesc* d);
size_t sg_conv2d_wino_ytp_floats(const sgConvDesc* d);
int sg_conv2d_wino_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                       float slope, float* ut_save, float* v_save, void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_wino_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, const float* v_saved,
                         const float* ytp_saved, void* ws, size_t ws_bytes, sgStream stream);
/* gx [N, C1, H, W] (all input channels): Winograd on the padded gradient grid + reflection fold */
int sg_conv2d_wino_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, const float* ut_saved,
                         float* ytp_save, void* ws, size_t ws_bytes, sgStream stream);
/* ReflectionPad(1) + Conv3x3 + InstanceNorm2d(affine=False) [+ ReLU / LeakyReLU] [+ residual] of a ResnetBlock (layers.py:251-270,
 * 296) on the Winograd F(4x4,3x3) path, fused at both ends of the batched GEMMs (sg_conv2d_wino_in_supported: 8x8 .. 16x16 planes,
 * channels multiples of 128, tile count a multiple of 64):
 *   forward : output transform + bias + InstanceNorm + activation + skip in one launch; ypre = the conv result (kept for the
 *             backward), out = act((ypre - mean) * rstd) + skip, mean / rstd [N * Cout]
 *   backward: InstanceNorm's backward and the gradient transform in one launch (gconv = the conv's gy: its weight gradient
 *             is sg_conv2d_wino_wgrad(d, gconv, x, gw, v_saved, ytp_saved)); gx is computed when non-NULL; gb [Cout], when
 *             non-NULL, receives the conv's bias gradient = sum of gconv over images and pixels (per-image plane sums from the
 *             same launch, added over the images in ascending order by one small launch; NULL: the caller runs
 *             sg_channel_sum(gconv)).  ut_save / v_save / ytp_save as in sg_conv2d_wino_fwd / _dgrad.
 * Same arithmetic as sg_conv2d_wino_fwd + sg_instnorm_fwd / sg_instnorm_bwd + sg_conv2d_wino_dgrad up to the order of the
 * per-plane sums. */
int sg_conv2d_wino_in_supported(const sgConvDesc* d);
int sg_conv2d_wino_fwd_instnorm(const sgConvDesc* d, const float* x, const float* w, const float* bias, const float* skip,
                                float* ypre, float* out, float* mean, float* rstd, float eps, int act, float slope,
                                float* ut_save, float* v_save, void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_wino_dgrad_instnorm(const sgConvDesc* d, const float* gout, const float* ypre, const float* mean, const float* rstd,
                                  int act, float slope, const float* w, float* gconv, float* gx, float* gb,
                                  const float* ut_saved, float* ytp_save, void* ws, size_t ws_bytes, sgStream stream);
/* The GEMM stage of the Winograd convs on its own (the transforms of layers.py:251-270's convs aside):
 *   c[m][z*cols + j] = sum_k a[z][m][k] * b[z*cols + j][k],   z < nbatch   (both operands K-contiguous, fp32 MFMA)
 * tile: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 64x64 with 16-deep k-tiles; M, cols multiples of the tile, K of 32.  Exposed for
 * micro-benchmarks of candidate Winograd forms (tools/bench_wino_gemm.py: the F(4x4,3x3) study) and for tests. */
int sg_batched_gemm_nt(const float* a, const float* b, float* c, int nbatch, int M, int cols, int K, int tile,
                       sgStream stream);
/* Winograd F(2x2, 4x4) for the stride-1 4x4 convs of the PatchGANs (reference discriminators.py:221-228:
   nn.Conv2d(nf_prev, nf, kernel_size=4, stride=1, padding=2), 256 -> 512 channels: the largest layer of the discriminator steps):
   KS 4, stride 1, zero padding 0..3, one source, C1 and Cout multiples of 128, >= 256 output tiles.  25 multiplies per 2x2
   output tile and channel pair instead of 64; fp32 throughout, results agree with sg_conv2d_fwd / _dgrad / _wgrad to fp32
   rounding (asserted at the same tolerances; gb via sg_channel_sum).  ws: sg_conv2d_wino24_ws_bytes. */
int sg_conv2d_wino24_supported(const sgConvDesc* d);
size_t sg_conv2d_wino24_ws_bytes(const sgConvDesc* d);
int sg_conv2d_wino24_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                         float slope, void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_wino24_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                           sgStream stream);
int sg_conv2d_wino24_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                           sgStream stream);
/* Direct (vector-ALU) kernels for ReflectionPad2d(3) + Conv2d(C, Cout <= 4, 7) [+ act]: the generator's RGB head
   (reference generators.py:88-90).  Same results as sg_conv2d_fwd / sg_conv2d_wgrad (gb via sg_channel_sum). */
int sg_conv2d_smallm_supported(const sgConvDesc* d);
size_t sg_conv2d_smallm_ws_bytes(const sgConvDesc* d);
int sg_conv2d_smallm_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                         float slope, sgStream stream);
int sg_conv2d_smallm_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                           sgStream stream);
/* Direct (vector-ALU) kernels for SINGLE-output-channel convs with zero padding, stride 1, KS in {1,3,4}: the PatchGAN score
   heads (reference discriminators.py:152-158,232-234) and the 1x1 head of mask_net (generators.py:27).  Memory-bound
   reductions; channel chunks / images are combined in a fixed order.  ws: sg_conv2d_head_ws_bytes.  gb via sg_channel_sum. */
int sg_conv2d_head_supported(const sgConvDesc* d);
size_t sg_conv2d_head_ws_bytes(const sgConvDesc* d);
int sg_conv2d_head_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y, int act,
                       float slope, void* ws, size_t ws_bytes, sgStream stream);
int sg_conv2d_head_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, sgStream stream);
int sg_conv2d_head_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, void* ws, size_t ws_bytes,
                         sgStream stream);
/* nn.ConvTranspose2d(k3,s2,p1,op1) : w [Cin, Cout, KS, KS]; desc.H,W = input size, OH,OW = output size */
int sg_convT2d_fwd(const sgConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                   void* ws, size_t ws_bytes, sgStream stream);
int sg_convT2d_dgrad(const sgConvDesc* d, const float* gy, const float* w, float* gx, void* ws, size_t ws_bytes,
                     sgStream stream);
int sg_convT2d_wgrad(const sgConvDesc* d, const float* gy, const float* x, float* gw, float* gb,
                     void* ws, size_t ws_bytes, sgStream stream);
/* Interpolate(x2, nearest) + Conv2d(C, Cout, 3, padding=1) (mask_net, reference generators.py:20-21, layers.py:304-314) as a
 * SUB-PIXEL transposed convolution: every output pixel (2i+a, 2j+b) only sees a 2x2 neighbourhood of the stored input, with
 * the 3x3 taps that land on the same source pixel summed -- conv3x3(up2(x); w) == convT(k4, s2, p1)(x; wt) with
 *   wt[ci][co][kh][kw] = sum_{i in R(kh)} sum_{j in R(kw)} w[co][ci][i][j],  R(0)={2} R(1)={1,2} R(2)={0,1} R(3)={0}
 * i.e. 16 instead of 36 multiply-adds per (input pixel, channel pair).  sg_upconv3_fold_weights builds wt; forward, data and
 * weight gradient are sg_convT2d_{fwd,dgrad,wgrad} on a (KS=4, stride 2, pad 1) desc; sg_upconv3_unfold_wgrad is the adjoint
 * of the fold: gw[co][ci][i][j] = sum_{kh: i in R(kh)} sum_{kw: j in R(kw)} gwt[ci][co][kh][kw]. */
int sg_upconv3_fold_weights(const float* w, float* wt, int Cout, int Cin, sgStream stream);
int sg_upconv3_unfold_wgrad(const float* gwt, float* gw, int Cout, int Cin, sgStream stream);
/* fold a dgrad taken w.r.t. the reflect-padded and/or x2-upsampled logical input back onto the stored
 * input: gx[NC,H,W] = sum of gp[NC, H*up+2p, W*up+2p] over reflected / replicated positions */
int sg_pad_upsample_bwd(const float* gp, float* gx, int NC, int H, int W, int pad, int upsample, sgStream stream);
/* per-channel sum over (N, HW): bias gradients.  ws (optional, sg_channel_sum_ws_bytes) enables the two-stage form */
size_t sg_channel_sum_ws_bytes(int C);
int sg_channel_sum(const float* g, float* out, int N, int C, int HW, void* ws, size_t ws_bytes, sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers (nn.Linear inside build_mlp layers.py:215-231, generators.py:45, discriminators.py:23-27)
 * ------------------------------------------------------------------------------------------- */
int sg_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in_f, int out_f,
                  int act, float slope, sgStream stream);                 /* y = act(x w^T + b) */
int sg_linear_bwd_data(const float* gy, const float* w, float* gx, int rows, int in_f, int out_f, sgStream stream);
int sg_linear_bwd_weight(const float* gy, const float* x, float* gw, float* gb, int rows, int in_f, int out_f,
                         sgStream stream);
/* gx = gy * act'(.) evaluated from the activation OUTPUT y (relu / leaky / tanh / sigmoid) */
int sg_act_bwd(const float* y, const float* gy, float* gx, int64_t n, int act, float slope, sgStream stream);
int sg_act_fwd(const float* x, float* y, int64_t n, int act, float slope, sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Graph convolution (graph.py:58-122) + embeddings (model.py:131-132)
 * ------------------------------------------------------------------------------------------- */
/* destination-major CSR of the 2T (pass, t) entries: pass 0 = subject column, pass 1 = object column,
 * entries of a row ordered (pass, t ascending) == the order CPU scatter_add applies them (graph.py:98-101).
 * csr_off[O+1], csr_ent[2T] (t | pass<<30). */
int sg_build_csr(const int64_t* edges /*T,2*/, int T, int O, int32_t* csr_off, int32_t* csr_ent, sgStream stream);
/* Range check of an index operand on the device: 0 if lo <= idx[i] < hi for every i < n, else SG_ERR_INDEX with the first
 * offending position and value in sg_last_error_string() -- what the reference's indexing raises as IndexError
 * (obj_vecs[s_idx], graph.py:79-80; nn.Embedding, model.py:131-132; feats[bbox_to_feats], bilinear.py:36).  Synchronises the
 * stream (a debugging aid, not part of the hot path); inside a stream capture it returns 0 unchecked.  The kernels
 * themselves do NOT validate indices: with the option check_indices = 1 (sg_set_option / SG_CHECK_INDICES=1) sg_build_csr
 * checks its edges this way, and the host mirror checks objs / obj_to_img / bbox_to_feats before the launches that use them. */
int sg_check_indices(const int64_t* idx, int64_t n, int64_t lo, int64_t hi, const char* what, sgStream stream);
/* out[t] = [obj[s_t], pred[t], obj[o_t]]  (graph.py:79-84) */
int sg_gather_concat_fwd(const float* obj, const float* pred, const int64_t* edges, float* out,
                         int T, int Do, int Dp, sgStream stream);
/* y[t] = act([obj[s_t], pred[t], obj[o_t]] W^T + b): the gather of graph.py:79-84 and the first nn.Linear (+ReLU) of net1
 * (graph.py:58-60,86) in ONE launch -- the A loader of the register-streaming GEMM reads the node / edge feature rows directly,
 * the (T, 2 Do + Dp) matrix is never written.  W: [out_f][2 Do + Dp].  Graphs too large for that kernel (the same size rule
 * as sg_linear_fwd) take gather + GEMM through ws (sg_gconv_gather_linear_ws_bytes: 0 for the fused form). */
size_t sg_gconv_gather_linear_ws_bytes(int T, int Do, int Dp, int out_f);
int sg_gconv_gather_linear_fwd(const float* obj, const float* pred, const int64_t* edges, const float* w, const float* b,
                               float* y, int T, int Do, int Dp, int out_f, int act, float slope, void* ws, size_t ws_bytes,
                               sgStream stream);
/* dst[i, :] = (sum over CSR entries e of row i of src[t_e, col_off[pass_e] : +width]) / (avg ? max(deg_i,1) : 1)
 * forward pool: src=new_t, col_off={0, H+Dout}; gather backward: src=g_cur_t, col_off={0, Do+Dp}. Bit-exact
 * w.r.t. sequential CPU scatter_add (deterministic, no atomics). */
int sg_segment_sum(const float* src, int src_ld, int col_off0, int col_off1, int width, const int32_t* csr_off,
                   const int32_t* csr_ent, float* dst, int O, int avg, sgStream stream);
/* g_new_t[t] = [g_pooled[s_t]/cnt_s, g_new_p[t], g_pooled[o_t]/cnt_o]  (dual of the pool, graph.py:89-116) */
int sg_pool_bwd(const float* g_pooled, const float* g_new_p, const int64_t* edges, const int32_t* csr_off,
                float* g_new_t, int T, int H, int Dout, int avg, sgStream stream);
int sg_embedding_fwd(const float* table, const int64_t* idx, float* out, int n, int dim, sgStream stream);
int sg_embedding_bwd(const float* g, const int64_t* idx, float* g_table, int n, int num_rows, int dim, sgStream stream);
/* strided 2-D copy: dst[r, dst_off : dst_off+width] = src[r, src_off : src_off+width] (concat / split glue) */
int sg_copy_cols(const float* src, int src_ld, int src_off, float* dst, int dst_ld, int dst_off, int rows, int width,
                 sgStream stream);
int sg_one_hot(const int64_t* idx, float* out, int n, int classes, int ld, int col_off, sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Normalisation / pooling (nn.InstanceNorm2d, nn.BatchNorm2d, nn.AvgPool2d(3,2,1,count_include_pad=False),
 * GlobalAvgPool layers.py:82-85)
 * ------------------------------------------------------------------------------------------- */
/* y = act((x-mean)*rstd) [+ skip]; biased variance per (n,c) plane, eps inside sqrt (layers.py:296) */
int sg_instnorm_fwd(const float* x, const float* skip, float* y, float* mean, float* rstd, int NC, int HW, float eps,
                    int act, float slope, sgStream stream);
int sg_instnorm_bwd(const float* x, const float* gy, const float* mean, const float* rstd, float* gx, int NC, int HW,
                    int act, float slope, sgStream stream);
/* training: batch stats (biased var) + running-stat update (unbiased var, momentum) + num_batches_tracked++;
   ws (sg_batchnorm_ws_bytes) holds the per-slice partial statistics of the multi-workgroup reduction */
size_t sg_batchnorm_ws_bytes(int N, int C, int HW);
int sg_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                     float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches, int N, int C,
                     int HW, float eps, float momentum, int training, int act, float slope, void* ws, size_t ws_bytes,
                     sgStream stream);
/* beta is needed to rebuild the pre-activation gamma*z+beta for the fused activation mask */
int sg_batchnorm_bwd(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                     const float* save_rstd, float* gx, float* ggamma, float* gbeta, int N, int C, int HW, int training,
                     int act, float slope, void* ws, size_t ws_bytes, sgStream stream);
int sg_avgpool3s2_fwd(const float* x, float* y, int NC, int H, int W, int OH, int OW, sgStream stream);
int sg_avgpool3s2_bwd(const float* gy, float* gx, int NC, int H, int W, int OH, int OW, sgStream stream);
/* nn.MaxPool2d(2, 2) of the VGG19 feature extractor behind VGGLoss (losses.py:179-224; torchvision vgg19.features[4,9,18,27]):
 * y [NC, H/2, W/2]; _bwd routes gy to the first maximum of each window (recomputed from x), zero elsewhere */
int sg_maxpool2_fwd(const float* x, float* y, int NC, int H, int W, sgStream stream);
int sg_maxpool2_bwd(const float* x, const float* gy, float* gx, int NC, int H, int W, sgStream stream);
/* build_cnn's 'P<k>' layers for any window (layers.py:181-189): nn.MaxPool2d(k, k) (avg = 0; _bwd recomputes the first
 * maximum of each window from x) / nn.AvgPool2d(k, k) (avg = 1; x may be null in _bwd).  y [NC, H/k, W/k] */
int sg_pool2d_fwd(const float* x, float* y, int NC, int H, int W, int k, int avg, sgStream stream);
int sg_pool2d_bwd(const float* x, const float* gy, float* gx, int NC, int H, int W, int k, int avg, sgStream stream);
/* nn.ReplicationPad2d(pad) of ResnetBlock(padding_type='replicate') (layers.py:245-246,258-259) and its adjoint (gp is the
 * gradient on the padded grid [NC, H+2pad, W+2pad]; deterministic gather, no atomics) */
int sg_replicate_pad_fwd(const float* x, float* y, int NC, int H, int W, int pad, sgStream stream);
int sg_replicate_pad_bwd(const float* gp, float* gx, int NC, int H, int W, int pad, sgStream stream);
int sg_gap_fwd(const float* x, float* y, int NC, int HW, sgStream stream);
int sg_gap_bwd(const float* gy, float* gx, int NC, int HW, sgStream stream);
int sg_upsample2_fwd(const float* x, float* y, int NC, int H, int W, sgStream stream);   /* nearest x2 */
int sg_reflect_pad_fwd(const float* x, float* y, int NC, int H, int W, int pad, sgStream stream);
int sg_concat_channels(const float* a, const float* b, float* out, int N, int Ca, int Cb, int HW, sgStream stream);
/* Conv over [x1 || cond row expanded over the grid] with the broadcast source folded into a per-(n, m, tap) term
 * (MultiscaleMaskDiscriminator.singleD_forward, discriminators.py:107-110: cat([feat, cond.expand(...)], 1) -> Conv2d):
 *   y = conv(x1, W[:, :C1]) + sum_{taps inside the plane at (oh, ow)} P[n][m][tap],  P = cond x W2r^T (sg_linear_fwd).
 * split_w: W [M][C1+C2][R] -> W1 [M][C1][R], W2r [M*R][C2];  merge_w: its adjoint (a NULL source reads as zeros);
 * bias_act: y[NM][OH][OW] = act(y + window sum of P[NM][KS*KS]) in place;  window_sums: its adjoint, gP[NM][KS*KS] (KS 1, 3, 4) */
int sg_cond_conv_split_w(const float* w, float* w1, float* w2r, int M, int C1, int C2, int R, sgStream stream);
int sg_cond_conv_merge_w(const float* gw1, const float* gw2r, float* gw, int M, int C1, int C2, int R, sgStream stream);
int sg_cond_conv_bias_act(float* y, const float* p, int NM, int OH, int OW, int H, int W, int KS, int stride, int pad,
                          int act, float slope, sgStream stream);
int sg_cond_conv_window_sums(const float* g, float* gp, int NM, int OH, int OW, int H, int W, int KS, int stride, int pad,
                             sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Layout scatter and bilinear crops (layout.py:64-155, bilinear.py:67-130)
 * ------------------------------------------------------------------------------------------- */
/* grid_sample geometry of the bilinear operators below (masks_to_layout, crop_bbox): 0 (default) = align_corners=False,
 * what torch >= 1.3 executes for the reference's calls (layout.py:51,86,88; bilinear.py:130); 1 = align_corners=True, the
 * default of the PyTorch 1.0 the reference was written for (requirements.txt:8) -- needed to run its released checkpoints
 * with the geometry they were trained with.  Process-wide switch (the second piece of global state next to the profiler). */
int sg_set_legacy_align_corners(int on);
int sg_get_legacy_align_corners(void);
/* seg_off[N+1] from a sorted, gap-free obj_to_img */
int sg_segment_offsets(const int64_t* obj_to_img, int O, int N, int32_t* seg_off, sgStream stream);
/* out[n,d,h,w] = sum_{o in image n, ascending} vecs[o,d] * bilinear(mask_o, box_o)(h,w)   (align_corners=False,
 * zeros padding); masks int64 0/1 (gt) or fp32 (predicted); never materialises (O,D,H,W). */
int sg_masks_to_layout_fwd(const float* vecs, const float* boxes, const void* masks, int masks_i64,
                           const int32_t* seg_off, float* out, int N, int O, int D, int M, int H, int W, int avg,
                           int max_per_image /* hint: objects per image the LDS tile is provisioned for; 0 = default */,
                           sgStream stream);
/* test-mode compositing (layout.py:87-92,157-169): per image, objects are visited in ascending mass
 * (sum over d,h,w of vecs[o,d]*sampled_mask_o[h,w]; ties by index) and a pixel takes vecs[o]*sampled_mask_o of the
 * first visited object whose sampled mask exceeds 0.5 (zero if none).  Inference only: no backward. */
size_t sg_masks_to_layout_test_ws_bytes(int O);
int sg_masks_to_layout_test_fwd(const float* vecs, const float* boxes, const void* masks, int masks_i64,
                                const int32_t* seg_off, float* out, void* ws, size_t ws_bytes, int N, int O, int D, int M,
                                int H, int W, int avg, sgStream stream);
/* g_vecs[o, d] for d in [d_begin, D) (columns below d_begin are zero-filled) */
int sg_masks_to_layout_bwd_vecs(const float* gout, const float* boxes, const void* masks, int masks_i64,
                                const int64_t* obj_to_img, const int32_t* seg_off, float* g_vecs, int N, int O, int D,
                                int M, int H, int W, int avg, int d_begin, sgStream stream);
/* gradients of masks_to_layout w.r.t. the masks (float masks only) and / or the boxes (layout.py:85-86 is differentiable in
 * both): g_masks [O, M, M], g_boxes [O, 4] (either may be null); ws >= sg_masks_to_layout_bwd_geom_ws_bytes (O x H x W
 * floats: the per-object gradient map sum_d gout[n,d] * vecs[o,d]); gather formulation, deterministic */
size_t sg_masks_to_layout_bwd_geom_ws_bytes(int O, int H, int W);
int sg_masks_to_layout_bwd_geom(const float* gout, const float* vecs, const float* boxes, const void* masks, int masks_i64,
                                const int64_t* obj_to_img, const int32_t* seg_off, float* g_masks, float* g_boxes, void* ws,
                                size_t ws_bytes, int N, int O, int D, int M, int H, int W, int avg, sgStream stream);
int sg_crop_bbox_fwd(const float* feats, const float* boxes, const int64_t* box_to_feat, float* out, int N, int C, int H,
                     int W, int B, int HH, int WW, sgStream stream);
/* g_feats [N, C, H, W] is written completely (no zero fill needed): a gather over the crop pixels whose bilinear
 * footprint covers each image pixel, summed in (box, crop row, crop column) order => bit-reproducible */
int sg_crop_bbox_bwd(const float* gout, const float* boxes, const int64_t* box_to_feat, float* g_feats, int N, int C,
                     int H, int W, int B, int HH, int WW, sgStream stream);
/* crop_bbox(feats, bbox, HH, WW, backend='jj') (bilinear.py:101-130 with bilinear_sample, bilinear.py:188-243): one box per
 * image, pixel coordinate X * W without the half-pixel shift, floor / floor + 1 taps clamped to the plane.  No caller of the
 * reference reaches this sampler (crop_bbox_batch never forwards its backend); _bwd = the gradient w.r.t. feats, deterministic. */
int sg_crop_bbox_jj_fwd(const float* feats, const float* boxes, float* out, int N, int C, int H, int W, int HH, int WW,
                        sgStream stream);
int sg_crop_bbox_jj_bwd(const float* gout, const float* boxes, float* g_feats, int N, int C, int H, int W, int HH, int WW,
                        sgStream stream);
/* Per-image filters of the factored layout convs: layout = sum_o [one_hot(class_o) | repr_o] (x) S_o (model.py:165-168,
 * layout.py:85-86) => conv(layout | x2, w)[n] = sum_j wimg[n][:, j] (*) plane_j with
 *   wimg[n][m][j][t] = w[m][class_o][t] + sum_d repr[o][d] w[m][C + d][t]   for the j-th object o of image n (j < cnt_n)
 *                    = w[m][C + R + (j - cnt_n)][t]                         for the C2 channels of the second source
 * w [M][C + R + C2][KS2], repr [O][R], objs [O] class ids, seg_off [N + 1] object offsets per image, wimg [N][M][L][KS2].
 * _bwd: gw [M][C + R + C2][KS2] (written completely) and / or grepr [O][R] from gwimg; sums in index order. */
int sg_factored_weights_fwd(const float* w, const float* repr, const int64_t* objs, const int32_t* seg_off, float* wimg,
                            int N, int O, int M, int L, int KS2, int C, int R, int C2, sgStream stream);
int sg_factored_weights_bwd(const float* gwimg, const float* w, const float* repr, const int64_t* objs,
                            const int32_t* seg_off, const int64_t* img_idx, float* gw, float* grepr, int N, int O, int M,
                            int L, int KS2, int C, int R, int C2, sgStream stream);
/* VectorPool.query on device (utils.py:62-90): plan = int32[4][O] rows {class, src_kind, src_idx, slot}
 * out[i] = src_kind ? pool[class][src_idx] : vectors[src_idx]; then pool[class][slot] = vectors[i] (slot>=0) */
int sg_vector_pool_exchange(float* pool, const float* vectors, const int32_t* plan, float* out, int O, int R,
                            int pool_size, sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Losses (losses.py:26-90,135-175; trainer.py:215,331-340; discriminators.py:35) and Adam (trainer.py:60,80,106,133)
 * ------------------------------------------------------------------------------------------- */
size_t sg_loss_ws_bytes(int64_t n);
/* out[0] (+)= scale * sum_i l(a_i, b_i or target); deterministic two-stage reduction */
int sg_loss_fwd(int kind, const float* a, const float* b, float target, int64_t n, float scale, float* out,
                int accumulate, void* ws, size_t ws_bytes, sgStream stream);
/* ga_i = gout[0] * scale * dl/da_i */
int sg_loss_bwd(int kind, const float* a, const float* b, float target, int64_t n, float scale, const float* gout,
                float* ga, sgStream stream);
/* Several scalar losses of ONE kind and their weighted sum, one launch forward and one backward:
 *   out[0] = sum_t weight[t] * (scale[t] * sum_i l(a_t[i], b_t[i] | target[t])),   t < nterms <= SG_WSUM_MAX
 * -- the feature-matching L1 terms of calculate_features_loss (trainer.py:331-340), the per-scale terms of GANLoss
 * (losses.py:166-172), the five VGG terms (losses.py:220-224).  Same arithmetic in the same order as sg_loss_fwd per term
 * followed by sg_weighted_sum_fwd (bit-identical).  *_host = HOST arrays of nterms entries (a / b / ga: device pointers;
 * b_host, target_host may be NULL; a NULL ga entry skips that term's gradient); terms_out (optional): the nterms scaled
 * terms.  _bwd: ga_t[i] = weight[t] * gout[0] * scale[t] * dl/da. */
size_t sg_multi_loss_ws_bytes(int nterms);
int sg_multi_loss_fwd(int kind, int nterms, const void* const* a_host, const void* const* b_host, const int64_t* n_host,
                      const float* scale_host, const float* weight_host, const float* target_host, float* out,
                      float* terms_out, void* ws, size_t ws_bytes, sgStream stream);
int sg_multi_loss_bwd(int kind, int nterms, const void* const* a_host, const void* const* b_host, const int64_t* n_host,
                      const float* scale_host, const float* weight_host, const float* target_host, const float* gout,
                      void* const* ga_host, sgStream stream);
/* mean over rows of -log softmax(logits)[target]; per-row loss kept in row_loss[rows] */
int sg_cross_entropy_fwd(const float* logits, const int64_t* target, int rows, int classes, float* row_loss,
                         float* out, sgStream stream);
int sg_cross_entropy_bwd(const float* logits, const int64_t* target, int rows, int classes, const float* gout,
                         float* glogits, sgStream stream);
/* torch.optim.Adam step (no weight decay / amsgrad) over one flat fp32 buffer.  The gradient enters as g * grad_scale:
 * 1 on one GPU; 1 / world under data parallelism, where g holds the all-reduced SUM (the mean's scaling pass over the flat
 * gradient buffer -- 8 B per parameter -- is folded into this kernel's read) */
int sg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, float bias_corr1, float bias_corr2_sqrt, float grad_scale, sgStream stream);
int sg_fill(float* p, float value, int64_t n, sgStream stream);
int sg_scale(float* p, float alpha, int64_t n, sgStream stream);
/* dst (device) <- src (PAGE-LOCKED host memory, read through its device mapping) by a kernel on ``stream``: how a collated host
 * batch reaches the device (replaces the eight ``tensor.cuda()`` of train.py:192 for batches packed into one page-locked buffer;
 * scene_generation_amd/pipeline.py).  Both pointers 16-byte aligned, nbytes a multiple of 16. */
int sg_stage_copy(void* dst, const void* src_host_mapped, int64_t nbytes, sgStream stream);

/* y += alpha * x : a second gradient contribution to a parameter slice of the flat gradient buffer (the first one is
 * written in place by the weight-gradient kernels; replaces autograd's AccumulateGrad add, trainer.py:262,278,299,324) */
int sg_axpy(float* y, const float* x, float alpha, int64_t n, sgStream stream);
/* y += x ; x = 0 : folds a SPILL gradient buffer into the flat gradient buffer and clears it for the next step.  The k-th
 * (k >= 2) contribution a parameter receives between zero_grad() and step() -- the real / wrong-texture passes of a
 * discriminator, trainer.py:281-325 -- is written by its weight-gradient kernel straight into spill buffer k-2 (same layout as
 * the gradient buffer); one launch per spill buffer and optimiser step replaces one temporary + sg_axpy per parameter and pass */
int sg_add_clear(float* y, float* x, int64_t n, sgStream stream);
/* out = a + b : the shortcut add of build_cnn's 'R' residual blocks (reference layers.py:84-118);
 * out = alpha * a * b : nn.Dropout's mask multiply (layers.py:230, build_mlp(dropout=...)); the mask itself is drawn by the host
 * framework's device RNG.  Neither is on the benchmark path. */
int sg_add(const float* a, const float* b, float* out, int64_t n, sgStream stream);
int sg_mul(const float* a, const float* b, float alpha, float* out, int64_t n, sgStream stream);
/* total_loss = sum_i weight_i * loss_i over <= SG_WSUM_MAX device scalars (LossManager.add_loss, utils.py:50-57, and the
 * scale sums of GANLoss / calculate_features_loss, losses.py:166-172, trainer.py:331-340); terms_host = HOST array of
 * DEVICE pointers, weights_host = HOST array.  _bwd: gterms[i] = weights[i] * gout[0] */
int sg_weighted_sum_fwd(const void* const* terms_host, const float* weights_host, int n, float* out, sgStream stream);
int sg_weighted_sum_bwd(const float* weights_host, int n, const float* gout, float* gterms, sgStream stream);

/* ---------------------------------------------------------------------------------------------
 * Opt-in per-kernel timing with HIP events on the launch stream (bench.py roofline leg).
 * ------------------------------------------------------------------------------------------- */
int sg_prof_enable(int on);
int sg_prof_reset(void);
int sg_prof_num_kinds(void);
const char* sg_prof_kind_name(int kind);
/* synchronises the recorded events; totals since the last reset */
int sg_prof_read(int kind, double* total_ms, int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SG2IM_HIP_H */
