// Layout scatter + bilinear crops (HBM-bound).
//
// masks_to_layout (layout.py:64-93,96-128,131-155): the reference materialises vecs (x) masks as an
// (O, D, M, M) tensor, grid_samples it to (O, D, H, W) (3.85 GB at N=32/128^2) and sums per image in a
// Python loop.  Here each workgroup owns one image x one tile of pixels: it samples every object's
// mask ONCE per pixel into LDS (S_o[h,w], the factored form of SURVEY appendix D.2), then streams the
// D output channels out as float4 rows, out[n,d,px] = sum_o vecs[o,d] * S_o[px]  (ascending o).
// Algorithmic traffic = the 4*N*D*H*W output bytes.
//
// crop_bbox_batch (bilinear.py:67-130): one gather kernel indexed by box_to_feat -- no per-image
// nonzero()/expand/cat/inverse-permutation.
#include "common.h"
#include <stdlib.h>

namespace {

// torch.linspace(0,1,n)[j] as ATen evaluates it (symmetric halves)
__device__ __forceinline__ float lin01(int j, int n) {
  if (n == 1) return 0.f;
  const float step = 1.f / (float)(n - 1);
  return j < n / 2 ? step * (float)j : 1.f - step * (float)(n - 1 - j);
}
// torch.linspace(1,0,n)[j]
__device__ __forceinline__ float lin10(int j, int n) {
  if (n == 1) return 1.f;
  const float step = -1.f / (float)(n - 1);
  return j < n / 2 ? 1.f + step * (float)j : 0.f - step * (float)(n - 1 - j);
}

struct Tap { int i0, i1; float w0, w1; };

// un-normalisation (align_corners=False: what torch >= 1.3 executes for the reference's grid_sample calls; ac != 0: the
// align_corners=True geometry of the PyTorch 1.0 the reference was written for) + bilinear taps with zeros padding
__device__ __forceinline__ float tap_coord(float g, int size, int ac) {
  return ac ? (g + 1.f) * 0.5f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) * 0.5f;
}
__device__ __forceinline__ Tap make_tap(float g, int size, int ac = 0) {
  Tap t;
  const float p = tap_coord(g, size, ac);
  if (!(fabsf(p) < 1e8f)) {              // NaN / inf (degenerate box): propagate NaN like grid_sample
    t.i0 = t.i1 = 0;
    t.w0 = t.w1 = (p != p) ? p : 0.f;    // inf coordinate => fully outside => 0
    return t;
  }
  const float f = floorf(p);
  const int i0 = (int)f;
  t.w1 = p - f;
  t.w0 = (f + 1.f) - p;
  t.i0 = i0; t.i1 = i0 + 1;
  if (t.i0 < 0 || t.i0 >= size) { t.w0 = 0.f; t.i0 = 0; }
  if (t.i1 < 0 || t.i1 >= size) { t.w1 = 0.f; t.i1 = 0; }
  return t;
}

template <bool I64>
__device__ __forceinline__ float mask_at(const void* masks, size_t o, int M, int y, int x) {
  if (I64) return (float)reinterpret_cast<const int64_t*>(masks)[(o * M + y) * M + x];
  return reinterpret_cast<const float*>(masks)[(o * M + y) * M + x];
}

template <bool I64>
__device__ __forceinline__ float sample_mask(const void* masks, size_t o, int M, const Tap& ty, const Tap& tx) {
  // same tap order as grid_sample: nw, ne, sw, se
  float v = mask_at<I64>(masks, o, M, ty.i0, tx.i0) * (ty.w0 * tx.w0);
  v += mask_at<I64>(masks, o, M, ty.i0, tx.i1) * (ty.w0 * tx.w1);
  v += mask_at<I64>(masks, o, M, ty.i1, tx.i0) * (ty.w1 * tx.w0);
  v += mask_at<I64>(masks, o, M, ty.i1, tx.i1) * (ty.w1 * tx.w1);
  return v;
}

__device__ __forceinline__ float box_coord(float lin, float lo, float hi) {   // layout.py:111-126
  return ((lin - lo) / (hi - lo)) * 2.f - 1.f;
}

__global__ void segment_offsets_kernel(const int64_t* __restrict__ o2i, int O, int N, int32_t* __restrict__ off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > O) return;
  if (i == 0) { for (int n = 0; n <= (O > 0 ? (int)o2i[0] : N); ++n) off[n] = 0; }
  if (i == O) { for (int n = (O > 0 ? (int)o2i[O - 1] + 1 : 0); n <= N; ++n) off[n] = O; return; }
  if (i > 0) {
    const int a = (int)o2i[i - 1], b = (int)o2i[i];
    for (int n = a + 1; n <= b; ++n) off[n] = i;
  }
}

// PXT pixels per workgroup tile, VEC pixels per thread (VEC=4 needs W%4==0).  LDS holds `cap` objects; images
// with more objects are processed in chunks of `cap` (read-modify-write of the tile for chunks after the first).
template <bool I64, int VEC>
__global__ void __launch_bounds__(256) layout_fwd_kernel(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                        const void* __restrict__ masks, const int32_t* __restrict__ seg,
                                                        float* __restrict__ out, int D, int M, int H, int W, int avg,
                                                        int cap, int ac) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PXT = 256 * VEC;
  const int n = blockIdx.y;
  const int o_beg = seg[n], cnt = seg[n + 1] - o_beg;
  float* S = lds;                       // [cap][PXT]
  float* V = lds + (size_t)cap * PXT;   // [cap][D]
  const int tid = threadIdx.x;
  const int HW = H * W;
  const int px0 = blockIdx.x * PXT + tid * VEC;
  const bool live = px0 < HW;
  const int h = live ? px0 / W : 0, w0 = live ? px0 - h * W : 0;
  const float Y = lin01(h, H);
  const float denom = (float)(cnt > 1 ? cnt : 1);
  float* op = out + (size_t)n * D * HW + (live ? px0 : 0);

  if (cnt == 0) {                       // image without objects: all-zero layout
    if (live)
      for (int d = 0; d < D; ++d)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (w0 + v < W) op[(size_t)d * HW + v] = 0.f;
    return;
  }
  for (int c0 = 0; c0 < cnt; c0 += cap) {
    const int nc = min(cap, cnt - c0);
    __syncthreads();
    for (int i = tid; i < nc * D; i += 256) V[i] = vecs[(size_t)(o_beg + c0) * D + i];
    for (int j = 0; j < nc; ++j) {
      const size_t o = (size_t)(o_beg + c0 + j);
      const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
      const Tap ty = make_tap(box_coord(Y, y0, y1), M, ac);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float sv = 0.f;
        if (live && w0 + v < W) {
          const Tap tx = make_tap(box_coord(lin01(w0 + v, W), x0, x1), M, ac);
          sv = sample_mask<I64>(masks, o, M, ty, tx);
        }
        S[(size_t)j * PXT + tid * VEC + v] = sv;
      }
    }
    __syncthreads();
    if (!live) continue;
    const bool first = c0 == 0, last = c0 + cap >= cnt;
    for (int d = 0; d < D; ++d) {
      float acc[VEC];
      if (first) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
      } else if (VEC == 4) {
        const float4 r = *reinterpret_cast<const float4*>(op + (size_t)d * HW);
        acc[0] = r.x; acc[1] = r.y; acc[2] = r.z; acc[3] = r.w;
      } else {
        acc[0] = op[(size_t)d * HW];
      }
      for (int j = 0; j < nc; ++j) {
        const float c = V[j * D + d];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float term = c * S[(size_t)j * PXT + tid * VEC + v];
          acc[v] = (first && j == 0) ? term : acc[v] + term;   // ascending o, same association as the reference sum
        }
      }
      if (avg && last) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = acc[v] / denom;
      }
      if (VEC == 4) *reinterpret_cast<float4*>(op + (size_t)d * HW) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      else op[(size_t)d * HW] = acc[0];
    }
  }
}

// Register-resident form for W % 4 == 0 (the training path): a thread owns 4 consecutive pixels and keeps the sampled masks of
// up to CAP objects for them in REGISTERS; only the coefficients live in LDS, transposed to Vt[d][CAP] so that one channel costs
// CAP/4 broadcast ds_read_b128 instead of 2 LDS reads per (object, channel) -- the kernel above spends ~3.7 k LDS
// instructions per thread on a 1024-pixel tile and runs at 2.0 TB/s (0.25 of the HBM peak on its 428 MB of output).  Objects
// beyond an image's count contribute exact zeros (0 * 0 added in the same ascending order).  Images with more than CAP
// objects take further passes that read-modify-write the tile, like the chunks of the kernel above.
template <bool I64, int CAP>
__global__ void __launch_bounds__(256) layout_fwd_reg_kernel(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                            const void* __restrict__ masks, const int32_t* __restrict__ seg,
                                                            float* __restrict__ out, int D, int M, int H, int W, int avg, int ac,
                                                            int dchunk) {
  extern __shared__ __attribute__((aligned(16))) float Vt[];        // [dchunk][CAP]: this workgroup's channel range
  static_assert(CAP % 4 == 0, "CAP is read as float4s");
  const int n = blockIdx.y, tid = threadIdx.x;
  const int o_beg = seg[n], cnt = seg[n + 1] - o_beg;
  const int HW = H * W;
  const int px0 = (blockIdx.x * 256 + tid) * 4;
  const bool live = px0 < HW;
  const int h = live ? px0 / W : 0, w0 = live ? px0 - h * W : 0;
  const float Y = lin01(h, H);
  const float denom = (float)(cnt > 1 ? cnt : 1);
  // blockIdx.z splits the channels: the sampled masks are recomputed per chunk (cheap) so that a 32-image batch puts
  // ~2000 workgroups of stores in flight instead of 512
  const int d_lo = blockIdx.z * dchunk, d_n = min(dchunk, D - d_lo);
  float* op = out + ((size_t)n * D + d_lo) * HW + (live ? px0 : 0);
  for (int c0 = 0; c0 == 0 || c0 < cnt; c0 += CAP) {
    const int nc = min(CAP, cnt - c0);                 // (<= 0 for an image without objects: an all-zero layout)
    __syncthreads();
    for (int i = tid; i < d_n * CAP; i += 256) {
      const int d = i / CAP, j = i - d * CAP;
      Vt[i] = j < nc ? vecs[(size_t)(o_beg + c0 + j) * D + d_lo + d] : 0.f;
    }
    float sv[CAP][4];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      const bool on = live && j < nc;
      // an image without objects (nc <= 0) must not touch boxes at all: o_beg may equal O there, i.e. one row past the end
      // (block-uniform branch)
      const size_t o = (size_t)(o_beg + c0 + (j < nc ? j : 0));
      float x0 = 0.f, y0 = 0.f, x1 = 1.f, y1 = 1.f;
      if (nc > 0) { x0 = boxes[o * 4 + 0]; y0 = boxes[o * 4 + 1]; x1 = boxes[o * 4 + 2]; y1 = boxes[o * 4 + 3]; }
      const Tap ty = make_tap(box_coord(Y, y0, y1), M, ac);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const Tap tx = make_tap(box_coord(lin01(w0 + v, W), x0, x1), M, ac);
        sv[j][v] = on ? sample_mask<I64>(masks, o, M, ty, tx) : 0.f;
      }
    }
    __syncthreads();
    if (!live) continue;
    const bool first = c0 == 0, last = c0 + CAP >= cnt;
    for (int d = 0; d < d_n; ++d) {
      float4 acc = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(op + (size_t)d * HW);
      const float4* cp = reinterpret_cast<const float4*>(Vt + d * CAP);
#pragma unroll
      for (int q = 0; q < CAP / 4; ++q) {
        const float4 c4 = cp[q];
        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = q * 4 + e;
          acc.x += c[e] * sv[j][0]; acc.y += c[e] * sv[j][1]; acc.z += c[e] * sv[j][2]; acc.w += c[e] * sv[j][3];
        }
      }
      if (avg && last) { acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom; }
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      nt_f4* dst = reinterpret_cast<nt_f4*>(op + (size_t)d * HW);
      const nt_f4 val = {acc.x, acc.y, acc.z, acc.w};
      if (last) __builtin_nontemporal_store(val, dst);           // 428 MB nobody re-reads soon: keep it out of the caches
      else *dst = val;
    }
  }
}

// g_vecs[o,d] = sum_{h,w} gout[n,d,h,w] * S_o[h,w] : one workgroup per (d, image), S recomputed, block reduce
template <bool I64>
__global__ void __launch_bounds__(256) layout_bwd_vecs_kernel(const float* __restrict__ gout, const float* __restrict__ boxes,
                                                             const void* __restrict__ masks, const int32_t* __restrict__ seg,
                                                             float* __restrict__ gv, int D, int M, int H, int W, int avg,
                                                             int d_begin, int ac) {
  __shared__ float red[16];
  constexpr int OC = 8;
  const int n = blockIdx.y, d = d_begin + blockIdx.x;
  const int o_beg = seg[n], cnt = seg[n + 1] - o_beg;
  const int HW = H * W;
  const float* gp = gout + ((size_t)n * D + d) * HW;
  const float denom = avg ? (float)(cnt > 1 ? cnt : 1) : 1.f;
  for (int c0 = 0; c0 < cnt; c0 += OC) {
    float acc[OC];
#pragma unroll
    for (int j = 0; j < OC; ++j) acc[j] = 0.f;
    for (int px = threadIdx.x; px < HW; px += 256) {
      const int h = px / W, w = px - h * W;
      const float g = gp[px];
      const float Y = lin01(h, H), X = lin01(w, W);
#pragma unroll
      for (int j = 0; j < OC; ++j) {
        if (c0 + j < cnt) {
          const size_t o = (size_t)(o_beg + c0 + j);
          const Tap ty = make_tap(box_coord(Y, boxes[o * 4 + 1], boxes[o * 4 + 3]), M, ac);
          const Tap tx = make_tap(box_coord(X, boxes[o * 4 + 0], boxes[o * 4 + 2]), M, ac);
          acc[j] += g * sample_mask<I64>(masks, o, M, ty, tx);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < OC; ++j) {
      if (c0 + j < cnt) {                      // uniform across the block
        const float t = sg_block_sum(acc[j], red);
        if (threadIdx.x == 0) gv[(size_t)(o_beg + c0 + j) * D + d] = t / denom;
      }
    }
  }
}

__global__ void zero_cols_kernel(float* __restrict__ p, int rows, int ld, int width) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * width) return;
  const int r = i / width, c = i - (size_t)r * width;
  p[(size_t)r * ld + c] = 0.f;
}

// ---------------- crops ---------------------------------------------------------------------------
__device__ __forceinline__ void crop_taps(const float* __restrict__ boxes, int b, int y, int x, int HH, int WW, int H, int W,
                                          Tap& ty, Tap& tx, int ac) {
  const float x0 = 2.f * boxes[b * 4 + 0] - 1.f, y0 = 2.f * boxes[b * 4 + 1] - 1.f;
  const float x1 = 2.f * boxes[b * 4 + 2] - 1.f, y1 = 2.f * boxes[b * 4 + 3] - 1.f;
  const float gx = lin10(x, WW) * x0 + lin01(x, WW) * x1;     // bilinear.py:266-274
  const float gy = lin10(y, HH) * y0 + lin01(y, HH) * y1;
  tx = make_tap(gx, W, ac);
  ty = make_tap(gy, H, ac);
}

__global__ void crop_fwd_kernel(const float* __restrict__ feats, const float* __restrict__ boxes,
                                const int64_t* __restrict__ b2f, float* __restrict__ out, int C, int H, int W, int B, int HH,
                                int WW, int ac) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * HH * WW) return;
  const int x = i % WW;
  const int y = (i / WW) % HH;
  const int b = i / ((size_t)WW * HH);
  Tap ty, tx;
  crop_taps(boxes, b, y, x, HH, WW, H, W, ty, tx, ac);
  const float* fp = feats + (size_t)b2f[b] * C * H * W;
  float* op = out + (size_t)b * C * HH * WW + (size_t)y * WW + x;
  for (int c = 0; c < C; ++c) {
    const float* f = fp + (size_t)c * H * W;
    float v = f[ty.i0 * W + tx.i0] * (ty.w0 * tx.w0);
    v += f[ty.i0 * W + tx.i1] * (ty.w0 * tx.w1);
    v += f[ty.i1 * W + tx.i0] * (ty.w1 * tx.w0);
    v += f[ty.i1 * W + tx.i1] * (ty.w1 * tx.w1);
    op[(size_t)c * HH * WW] = v;
  }
}

// Gradient w.r.t. feats as a GATHER (bit-reproducible; the first version scattered with fp32 atomics, whose order of
// additions changed from run to run).  Workgroup = (image n, 256 pixels); the boxes cropping image n are compacted, 256 at a
// time and in ascending order, into LDS; every pixel then visits, box by box, the few crop pixels whose bilinear footprint
// covers it: the sample coordinate is affine in the crop index, so the candidates per axis are an index range (taken with one
// spare element on each side and confirmed with the exact forward taps).  Additions happen in (box, crop row, crop column)
// order.
__device__ __forceinline__ void crop_axis_range(float c0, float c1, int n_crop, int size, int u, int& lo, int& hi, int ac) {
  // pixel coordinate of crop index j: p(j) ~ p0 + s*j with p0 = p(0), s = (p(n-1) - p(0)) / (n-1)
  const float p0 = tap_coord(c0, size, ac);
  const float p1 = tap_coord(c1, size, ac);
  lo = 0; hi = n_crop - 1;
  if (n_crop == 1 || !(fabsf(p0) < 1e8f) || !(fabsf(p1) < 1e8f)) return;      // degenerate: test every index
  const float s = (p1 - p0) / (float)(n_crop - 1);
  if (fabsf(s) < 1e-6f) return;
  // taps of j touch pixel u iff u-1 <= p(j) < u+1
  float a = ((float)(u - 1) - p0) / s, b = ((float)(u + 1) - p0) / s;
  if (a > b) { const float t = a; a = b; b = t; }
  const float fl = floorf(a) - 1.f, fh = ceilf(b) + 1.f;
  lo = fl < 0.f ? 0 : (fl > (float)(n_crop - 1) ? n_crop : (int)fl);
  hi = fh > (float)(n_crop - 1) ? n_crop - 1 : (fh < 0.f ? -1 : (int)fh);
}

template <int CT>
__global__ void __launch_bounds__(256) crop_bwd_gather_kernel(const float* __restrict__ gout, const float* __restrict__ boxes,
                                                             const int64_t* __restrict__ b2f, float* __restrict__ gf, int C,
                                                             int H, int W, int B, int HH, int WW, int ac) {
  __shared__ int list[256];
  __shared__ int wcnt[4];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int pix = blockIdx.x * 256 + tid;
  const bool live = pix < H * W;
  const int v = live ? pix / W : 0, u = live ? pix - (pix / W) * W : 0;
  for (int c0 = 0; c0 < C; c0 += CT) {
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 256) {
      // ordered compaction of the boxes of image n among [b0, b0 + 256)
      const int b = b0 + tid;
      const bool mine = b < B && (int)b2f[b] == n;
      const unsigned long long bal = __ballot(mine);
      if (lane == 0) wcnt[wid] = __popcll(bal);
      __syncthreads();
      int base = 0, total = 0;
      for (int w = 0; w < 4; ++w) { if (w < wid) base += wcnt[w]; total += wcnt[w]; }
      if (mine) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = b;
      __syncthreads();
      if (live) {
        for (int k = 0; k < total; ++k) {
          const int bb = list[k];
          const float x0 = 2.f * boxes[bb * 4 + 0] - 1.f, y0 = 2.f * boxes[bb * 4 + 1] - 1.f;
          const float x1 = 2.f * boxes[bb * 4 + 2] - 1.f, y1 = 2.f * boxes[bb * 4 + 3] - 1.f;
          int jlo, jhi, ilo, ihi;
          crop_axis_range(x0, x1, WW, W, u, jlo, jhi, ac);
          if (jlo > jhi) continue;
          crop_axis_range(y0, y1, HH, H, v, ilo, ihi, ac);
          const float* gp = gout + ((size_t)bb * C + c0) * HH * WW;
          for (int i = ilo; i <= ihi; ++i) {
            const Tap ty = make_tap(lin10(i, HH) * y0 + lin01(i, HH) * y1, H, ac);
            const float wy = (ty.i0 == v ? ty.w0 : 0.f) + (ty.i1 == v ? ty.w1 : 0.f);
            if (wy == 0.f) continue;
            for (int j = jlo; j <= jhi; ++j) {
              const Tap tx = make_tap(lin10(j, WW) * x0 + lin01(j, WW) * x1, W, ac);
              const float wx = (tx.i0 == u ? tx.w0 : 0.f) + (tx.i1 == u ? tx.w1 : 0.f);
              if (wx == 0.f) continue;
              const float wgt = wy * wx;
#pragma unroll
              for (int c = 0; c < CT; ++c)
                if (c0 + c < C) acc[c] += gp[((size_t)c * HH + i) * WW + j] * wgt;
            }
          }
        }
      }
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c0 + c < C) gf[((size_t)n * C + c0 + c) * H * W + pix] = acc[c];
    }
  }
}

// ---------------- VectorPool -----------------------------------------------------------------------
__global__ void pool_gather_kernel(const float* __restrict__ pool, const float* __restrict__ vec,
                                   const int32_t* __restrict__ plan, float* __restrict__ out, int O, int R, int P) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)O * R) return;
  const int o = i / R, r = i - (size_t)o * R;
  const int cls = plan[o], kind = plan[O + o], idx = plan[2 * O + o];
  out[i] = kind ? pool[((size_t)cls * P + idx) * R + r] : vec[(size_t)idx * R + r];
}

__global__ void pool_scatter_kernel(float* __restrict__ pool, const float* __restrict__ vec,
                                    const int32_t* __restrict__ plan, int O, int R, int P) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)O * R) return;
  const int o = i / R, r = i - (size_t)o * R;
  const int cls = plan[o], slot = plan[3 * O + o];
  if (slot >= 0) pool[((size_t)cls * P + slot) * R + r] = vec[i];
}


// ---- test-mode compositing (layout.py:87-92,157-169) ---------------------------------------------------------
// The reference visits the objects of an image in ascending "mass" (sum over the sampled vecs (x) mask tensor, a host
// loop with one .item() per object and numpy argsort) and gives every pixel to the first visited object whose
// sampled mask exceeds 0.5.  Here: mass per object (factored: sum(vecs) * sum(sampled mask), fp64), rank inside the
// image, then one pass over the pixels.  No host round trip.
template <bool I64>
__global__ void __launch_bounds__(256) layout_mass_kernel(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                         const void* __restrict__ masks, double* __restrict__ mass, int D,
                                                         int M, int H, int W, int ac) {
  __shared__ double red[256];
  const size_t o = blockIdx.x;
  const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
  double s = 0.0;
  for (int p = threadIdx.x; p < H * W; p += 256) {
    const int h = p / W, w = p - h * W;
    const Tap ty = make_tap(box_coord(lin01(h, H), y0, y1), M, ac);
    const Tap tx = make_tap(box_coord(lin01(w, W), x0, x1), M, ac);
    s += (double)sample_mask<I64>(masks, o, M, ty, tx);
  }
  double v = 0.0;
  for (int d = threadIdx.x; d < D; d += 256) v += (double)vecs[o * D + d];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  const double S = red[0];
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) mass[o] = S * red[0];
}

// order[seg[n] + r] = local index of the object visited r-th in image n (ascending mass, ties by index, NaN last)
__global__ void layout_order_kernel(const double* __restrict__ mass, const int32_t* __restrict__ seg,
                                    int32_t* __restrict__ order) {
  const int n = blockIdx.x;
  const int beg = seg[n], cnt = seg[n + 1] - beg;
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    const double mj = mass[beg + j];
    const bool nj = mj != mj;
    int r = 0;
    for (int k = 0; k < cnt; ++k) {
      const double mk = mass[beg + k];
      const bool nk = mk != mk;
      const bool before = nj ? (!nk || k < j) : (!nk && (mk < mj || (mk == mj && k < j)));
      r += before ? 1 : 0;
    }
    order[beg + r] = j;
  }
}

template <bool I64>
__global__ void __launch_bounds__(256) layout_test_fwd_kernel(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                             const void* __restrict__ masks, const int32_t* __restrict__ seg,
                                                             const int32_t* __restrict__ order, float* __restrict__ out,
                                                             int D, int M, int H, int W, int avg, int ac) {
  const int n = blockIdx.y;
  const int beg = seg[n], cnt = seg[n + 1] - beg;
  const int HW = H * W;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const int h = px / W, w = px - h * W;
  const float Y = lin01(h, H), X = lin01(w, W);
  int win = -1;
  float sv = 0.f;
  for (int r = 0; r < cnt; ++r) {
    const size_t o = (size_t)(beg + order[beg + r]);
    const Tap ty = make_tap(box_coord(Y, boxes[o * 4 + 1], boxes[o * 4 + 3]), M, ac);
    const Tap tx = make_tap(box_coord(X, boxes[o * 4 + 0], boxes[o * 4 + 2]), M, ac);
    const float v = sample_mask<I64>(masks, o, M, ty, tx);
    if (v > 0.5f) { win = (int)o; sv = v; break; }
  }
  if (avg) sv /= (float)(cnt > 1 ? cnt : 1);
  float* op = out + (size_t)n * D * HW + px;
  const float* vr = vecs + (size_t)(win < 0 ? 0 : win) * D;
  for (int d = 0; d < D; ++d) op[(size_t)d * HW] = win < 0 ? 0.f : vr[d] * sv;
}

}  // namespace

// grid_sample geometry of every bilinear operator of this file: 0 = align_corners=False (what torch >= 1.3 executes for the
// reference's calls, the default and the parity target), 1 = align_corners=True (the PyTorch 1.0 the reference was written
// and its released checkpoints were trained with: requirements.txt:8).  Process-wide; set it before the first step.
static int g_align_corners = 0;
extern "C" int sg_set_legacy_align_corners(int on) { g_align_corners = on ? 1 : 0; return 0; }
extern "C" int sg_get_legacy_align_corners(void) { return g_align_corners; }

extern "C" int sg_segment_offsets(const int64_t* obj_to_img, int O, int N, int32_t* seg_off, sgStream stream) {
  SG_ARG_CHECK(obj_to_img && seg_off && O >= 0 && N > 0, "sg_segment_offsets: bad arguments");
  hipLaunchKernelGGL(segment_offsets_kernel, dim3(sg_cdiv(O + 1, 256)), dim3(256), 0, (hipStream_t)stream, obj_to_img, O, N,
                     seg_off);
  SG_LAUNCH_CHECK("sg_segment_offsets");
  return 0;
}

extern "C" int sg_masks_to_layout_fwd(const float* vecs, const float* boxes, const void* masks, int masks_i64,
                                      const int32_t* seg_off, float* out, int N, int O, int D, int M, int H, int W, int avg,
                                      int max_per_image, sgStream stream) {
  SG_ARG_CHECK(vecs && boxes && masks && seg_off && out && N > 0 && O > 0 && D > 0 && M > 0 && H > 0 && W > 0,
               "sg_masks_to_layout_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // LDS is provisioned for `cap` objects per image (the caller's hint, clamped to what fits); denser images are
  // still correct (chunked), only slower.  ~60 KB/workgroup keeps two workgroups per CU.
  const int use_vec = (W % 4 == 0) ? 4 : 1;
  const size_t per_obj = (size_t)(256 * use_vec + D) * sizeof(float);
  const size_t budget = 160 * 1024 - 1024;
  int cap = max_per_image > 0 ? max_per_image : 12;
  if (cap > O) cap = O;
  const int max_fit = (int)(budget / per_obj);
  SG_ARG_CHECK(max_fit >= 1, "sg_masks_to_layout_fwd: D=%d too large for LDS", D);
  if (cap > max_fit) cap = max_fit;
  const size_t lds_bytes = (size_t)cap * per_obj;
  SgProfScope prof(SG_K_LAYOUT_FWD, s, 0, 4.0 * N * D * (double)H * W + 4.0 * O * D + (masks_i64 ? 8.0 : 4.0) * O * M * M);
  const int regform = sg_opt(SG_OPT_LAYOUT_REG);      // 0: the LDS-staged kernel
  if (regform && use_vec == 4 && (size_t)D * 12 * sizeof(float) <= 64 * 1024) {
    const int tiles = sg_cdiv(H * W, 1024);
    // channel chunks per tile (grid.z): measured SLOWER on MI355X (4 chunks: 78 vs 69 us averaged over the kind, the whole
    // dense-layout step 47.5 vs 43.5 ms) -- interleaved store streams of different chunks -- so one workgroup writes all D
    int dsplit = sg_opt(SG_OPT_LAYOUT_DSPLIT);
    dsplit = dsplit < 1 ? 1 : (dsplit > 8 ? 8 : dsplit);
    if (dsplit > D / 16) dsplit = D / 16 > 0 ? D / 16 : 1;
    const int dchunk = sg_cdiv(D, dsplit);
    const dim3 g4(tiles, N, sg_cdiv(D, dchunk));
    const size_t lds = (size_t)dchunk * 12 * sizeof(float);
    if (masks_i64) hipLaunchKernelGGL((layout_fwd_reg_kernel<true, 12>), g4, dim3(256), lds, s, vecs, boxes, masks, seg_off, out, D, M, H, W, avg, g_align_corners, dchunk);
    else hipLaunchKernelGGL((layout_fwd_reg_kernel<false, 12>), g4, dim3(256), lds, s, vecs, boxes, masks, seg_off, out, D, M, H, W, avg, g_align_corners, dchunk);
    SG_LAUNCH_CHECK("sg_masks_to_layout_fwd");
    return 0;
  }
  const dim3 grid(sg_cdiv(H * W, 256 * use_vec), N);
#define LAUNCH_LAYOUT(I64, VEC)                                                                                         \
  do {                                                                                                                 \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&layout_fwd_kernel<I64, VEC>),                                    \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                                    \
    hipLaunchKernelGGL((layout_fwd_kernel<I64, VEC>), grid, dim3(256), lds_bytes, s, vecs, boxes, masks, seg_off, out, \
                       D, M, H, W, avg, cap, g_align_corners);                                                          \
  } while (0)
  if (masks_i64) { if (use_vec == 4) LAUNCH_LAYOUT(true, 4); else LAUNCH_LAYOUT(true, 1); }
  else { if (use_vec == 4) LAUNCH_LAYOUT(false, 4); else LAUNCH_LAYOUT(false, 1); }
#undef LAUNCH_LAYOUT
  SG_LAUNCH_CHECK("sg_masks_to_layout_fwd");
  return 0;
}

extern "C" size_t sg_masks_to_layout_test_ws_bytes(int O) { return (size_t)(O > 0 ? O : 1) * (sizeof(double) + sizeof(int32_t)) + 16; }

extern "C" int sg_masks_to_layout_test_fwd(const float* vecs, const float* boxes, const void* masks, int masks_i64,
                                           const int32_t* seg_off, float* out, void* ws, size_t ws_bytes, int N, int O, int D,
                                           int M, int H, int W, int avg, sgStream stream) {
  SG_ARG_CHECK(vecs && boxes && masks && seg_off && out && ws && N > 0 && O > 0 && D > 0 && M > 0 && H > 0 && W > 0,
               "sg_masks_to_layout_test_fwd: bad arguments");
  SG_ARG_CHECK(ws_bytes >= sg_masks_to_layout_test_ws_bytes(O), "sg_masks_to_layout_test_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  double* mass = reinterpret_cast<double*>(ws);
  int32_t* order = reinterpret_cast<int32_t*>(mass + O);
  SgProfScope prof(SG_K_LAYOUT_FWD, s, 0, 4.0 * N * D * (double)H * W);
  const dim3 grid(sg_cdiv(H * W, 256), N);
  if (masks_i64) {
    hipLaunchKernelGGL((layout_mass_kernel<true>), dim3(O), dim3(256), 0, s, vecs, boxes, masks, mass, D, M, H, W, g_align_corners);
    hipLaunchKernelGGL(layout_order_kernel, dim3(N), dim3(64), 0, s, (const double*)mass, seg_off, order);
    hipLaunchKernelGGL((layout_test_fwd_kernel<true>), grid, dim3(256), 0, s, vecs, boxes, masks, seg_off,
                       (const int32_t*)order, out, D, M, H, W, avg, g_align_corners);
  } else {
    hipLaunchKernelGGL((layout_mass_kernel<false>), dim3(O), dim3(256), 0, s, vecs, boxes, masks, mass, D, M, H, W, g_align_corners);
    hipLaunchKernelGGL(layout_order_kernel, dim3(N), dim3(64), 0, s, (const double*)mass, seg_off, order);
    hipLaunchKernelGGL((layout_test_fwd_kernel<false>), grid, dim3(256), 0, s, vecs, boxes, masks, seg_off,
                       (const int32_t*)order, out, D, M, H, W, avg, g_align_corners);
  }
  SG_LAUNCH_CHECK("sg_masks_to_layout_test_fwd");
  return 0;
}

extern "C" int sg_masks_to_layout_bwd_vecs(const float* gout, const float* boxes, const void* masks, int masks_i64,
                                           const int64_t* obj_to_img, const int32_t* seg_off, float* g_vecs, int N, int O,
                                           int D, int M, int H, int W, int avg, int d_begin, sgStream stream) {
  SG_ARG_CHECK(gout && boxes && masks && seg_off && g_vecs && N > 0 && O > 0 && D > 0 && d_begin >= 0 && d_begin < D,
               "sg_masks_to_layout_bwd_vecs: bad arguments");
  (void)obj_to_img;
  hipStream_t s = (hipStream_t)stream;
  if (d_begin > 0)
    hipLaunchKernelGGL(zero_cols_kernel, dim3(sg_cdiv((size_t)O * d_begin, 256)), dim3(256), 0, s, g_vecs, O, D, d_begin);
  SgProfScope prof(SG_K_LAYOUT_BWD, s, 0, 4.0 * N * (D - d_begin) * (double)H * W);
  const dim3 grid(D - d_begin, N);
  if (masks_i64) hipLaunchKernelGGL(layout_bwd_vecs_kernel<true>, grid, dim3(256), 0, s, gout, boxes, masks, seg_off, g_vecs, D, M, H, W, avg, d_begin, g_align_corners);
  else hipLaunchKernelGGL(layout_bwd_vecs_kernel<false>, grid, dim3(256), 0, s, gout, boxes, masks, seg_off, g_vecs, D, M, H, W, avg, d_begin, g_align_corners);
  SG_LAUNCH_CHECK("sg_masks_to_layout_bwd_vecs");
  return 0;
}

extern "C" int sg_crop_bbox_fwd(const float* feats, const float* boxes, const int64_t* box_to_feat, float* out, int N, int C,
                                int H, int W, int B, int HH, int WW, sgStream stream) {
  SG_ARG_CHECK(feats && boxes && box_to_feat && out && N > 0 && C > 0 && B >= 0 && HH > 0 && WW > 0,
               "sg_crop_bbox_fwd: bad arguments");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_CROP, s, 0, 4.0 * B * C * (double)HH * WW + 4.0 * N * C * (double)H * W);
  hipLaunchKernelGGL(crop_fwd_kernel, dim3(sg_cdiv((size_t)B * HH * WW, 256)), dim3(256), 0, s, feats, boxes, box_to_feat, out,
                     C, H, W, B, HH, WW, g_align_corners);
  SG_LAUNCH_CHECK("sg_crop_bbox_fwd");
  return 0;
}

extern "C" int sg_crop_bbox_bwd(const float* gout, const float* boxes, const int64_t* box_to_feat, float* g_feats, int N, int C,
                                int H, int W, int B, int HH, int WW, sgStream stream) {
  SG_ARG_CHECK(gout && boxes && box_to_feat && g_feats && N > 0 && C > 0 && B >= 0, "sg_crop_bbox_bwd: bad arguments");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  SgProfScope prof(SG_K_CROP, s, 0, 4.0 * B * C * (double)HH * WW + 4.0 * N * C * (double)H * W);
  // writes EVERY element of g_feats (no zero fill needed), additions in a fixed order
  if (C <= 1)
    hipLaunchKernelGGL(crop_bwd_gather_kernel<1>, dim3(sg_cdiv(H * W, 256), N), dim3(256), 0, s, gout, boxes, box_to_feat,
                       g_feats, C, H, W, B, HH, WW, g_align_corners);
  else
    hipLaunchKernelGGL(crop_bwd_gather_kernel<4>, dim3(sg_cdiv(H * W, 256), N), dim3(256), 0, s, gout, boxes, box_to_feat,
                       g_feats, C, H, W, B, HH, WW, g_align_corners);
  SG_LAUNCH_CHECK("sg_crop_bbox_bwd");
  return 0;
}

// ---- crop_bbox(backend='jj'): the bilinear_sample geometry (bilinear.py:127-128,188-243) ----------------------------------------
// One box per image; grid x[j] = lin10(j) x0 + lin01(j) x1 in [0, 1] (tensor_linspace), pixel coordinate X = x * W (no half-pixel
// shift), taps floor(X) and floor(X) + 1 clamped to [0, W - 1], weights (x1 - X) and (X - x0) from the CLAMPED tap positions (at
// X >= W - 1 both taps are W - 1 and the weights cancel: the reference's own edge behaviour).  No caller of the reference reaches
// this sampler (crop_bbox_batch never forwards its backend); built for completeness of the bilinear.py surface, not for speed.
struct JJTap { int i0, i1; float w0, w1; };
__device__ __forceinline__ JJTap jj_tap(float c0, float c1, int j, int n, int size) {
  const float X = (lin10(j, n) * c0 + lin01(j, n) * c1) * (float)size;
  const float hi = (float)(size - 1);
  const float f0 = fminf(fmaxf(floorf(X), 0.f), hi), f1 = fminf(fmaxf(f0 + 1.f, 0.f), hi);
  JJTap t;
  t.i0 = (int)f0; t.i1 = (int)f1; t.w0 = f1 - X; t.w1 = X - f0;
  return t;
}
__global__ void crop_jj_fwd_kernel(const float* __restrict__ feats, const float* __restrict__ boxes, float* __restrict__ out, int N,
                                   int C, int H, int W, int HH, int WW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * C * HH * WW) return;
  const int x = i % WW, y = (i / WW) % HH, c = (i / ((size_t)WW * HH)) % C, n = i / ((size_t)WW * HH * C);
  const JJTap tx = jj_tap(boxes[n * 4 + 0], boxes[n * 4 + 2], x, WW, W), ty = jj_tap(boxes[n * 4 + 1], boxes[n * 4 + 3], y, HH, H);
  const float* f = feats + ((size_t)n * C + c) * H * W;
  // w1 v(y0,x0) + w2 v(y1,x0) + w3 v(y0,x1) + w4 v(y1,x1), in the reference's order of additions (bilinear.py:242)
  float v = (tx.w0 * ty.w0) * f[ty.i0 * W + tx.i0];
  v += (tx.w0 * ty.w1) * f[ty.i1 * W + tx.i0];
  v += (tx.w1 * ty.w0) * f[ty.i0 * W + tx.i1];
  v += (tx.w1 * ty.w1) * f[ty.i1 * W + tx.i1];
  out[i] = v;
}
// gradient w.r.t. feats as a gather: one thread per input pixel walks the crop rows / columns whose taps name it, in (row, column)
// order -- deterministic, no atomics; O(HH WW) per pixel, which is fine for a path nobody trains through
__global__ void crop_jj_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ boxes, float* __restrict__ gf, int N,
                                   int C, int H, int W, int HH, int WW) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * C * H * W) return;
  const int u = i % W, r = (i / W) % H, c = (i / ((size_t)W * H)) % C, n = i / ((size_t)W * H * C);
  const float bx0 = boxes[n * 4 + 0], by0 = boxes[n * 4 + 1], bx1 = boxes[n * 4 + 2], by1 = boxes[n * 4 + 3];
  const float* g = gout + ((size_t)n * C + c) * HH * WW;
  float acc = 0.f;
  for (int y = 0; y < HH; ++y) {
    const JJTap ty = jj_tap(by0, by1, y, HH, H);
    const float wy = (ty.i0 == r ? ty.w0 : 0.f) + (ty.i1 == r ? ty.w1 : 0.f);
    if (ty.i0 != r && ty.i1 != r) continue;
    for (int x = 0; x < WW; ++x) {
      const JJTap tx = jj_tap(bx0, bx1, x, WW, W);
      if (tx.i0 != u && tx.i1 != u) continue;
      const float wx = (tx.i0 == u ? tx.w0 : 0.f) + (tx.i1 == u ? tx.w1 : 0.f);
      acc += (wx * wy) * g[y * WW + x];
    }
  }
  gf[i] = acc;
}

extern "C" int sg_crop_bbox_jj_fwd(const float* feats, const float* boxes, float* out, int N, int C, int H, int W, int HH, int WW,
                                   sgStream stream) {
  SG_ARG_CHECK(feats && boxes && out && N > 0 && C > 0 && H > 0 && W > 0 && HH > 0 && WW > 0, "sg_crop_bbox_jj_fwd: bad arguments");
  hipLaunchKernelGGL(crop_jj_fwd_kernel, dim3(sg_cdiv((size_t)N * C * HH * WW, 256)), dim3(256), 0, (hipStream_t)stream, feats, boxes,
                     out, N, C, H, W, HH, WW);
  SG_LAUNCH_CHECK("sg_crop_bbox_jj_fwd");
  return 0;
}
extern "C" int sg_crop_bbox_jj_bwd(const float* gout, const float* boxes, float* g_feats, int N, int C, int H, int W, int HH, int WW,
                                   sgStream stream) {
  SG_ARG_CHECK(gout && boxes && g_feats && N > 0 && C > 0 && H > 0 && W > 0 && HH > 0 && WW > 0, "sg_crop_bbox_jj_bwd: bad arguments");
  hipLaunchKernelGGL(crop_jj_bwd_kernel, dim3(sg_cdiv((size_t)N * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, gout, boxes,
                     g_feats, N, C, H, W, HH, WW);
  SG_LAUNCH_CHECK("sg_crop_bbox_jj_bwd");
  return 0;
}

extern "C" int sg_vector_pool_exchange(float* pool, const float* vectors, const int32_t* plan, float* out, int O, int R,
                                       int pool_size, sgStream stream) {
  SG_ARG_CHECK(pool && vectors && plan && out && O >= 0 && R > 0 && pool_size > 0, "sg_vector_pool_exchange: bad arguments");
  if (O == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(sg_cdiv((size_t)O * R, 256));
  hipLaunchKernelGGL(pool_gather_kernel, grid, dim3(256), 0, s, (const float*)pool, vectors, plan, out, O, R, pool_size);
  hipLaunchKernelGGL(pool_scatter_kernel, grid, dim3(256), 0, s, pool, vectors, plan, O, R, pool_size);
  SG_LAUNCH_CHECK("sg_vector_pool_exchange");
  return 0;
}

// ================================================================================================
// Per-image filters of the factored layout convs (ops.factored_layout_conv).
//   layout = sum_o vecs[o] (x) S_o with vecs[o] = [one_hot(class_o) | repr_o] (model.py:165-168, layout.py:85-86), so
//   conv(layout, W)[n] = sum_{o in n} W_eff[o] (*) S_o,   W_eff[o] = W[:, class_o] + sum_d repr[o, d] W[:, C + d]
// wimg[n][m][j][t]: j < cnt_n -> W_eff of the j-th object of image n; cnt_n <= j < cnt_n + C2 -> W[m][C + R + (j - cnt_n)][t]
// (the channels of a concatenated second source, e.g. the image next to the layout in the image discriminator); else 0.
// Replaces an embedding lookup + GEMM + index_put + cat chain of ATen launches; sums run in index order (deterministic).
// ================================================================================================
namespace {

__global__ void factored_weights_fwd_kernel(const float* __restrict__ W, const float* __restrict__ repr,
                                            const int64_t* __restrict__ objs, const int32_t* __restrict__ seg, float* __restrict__ wimg,
                                            int N, int M, int L, int KS2, int C, int R, int C2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * M * L * KS2) return;
  const int t = (int)(i % KS2);
  const int j = (int)((i / KS2) % L);
  const int m = (int)((i / ((size_t)KS2 * L)) % M);
  const int n = (int)(i / ((size_t)KS2 * L * M));
  const int cnt = seg[n + 1] - seg[n];
  const int Ct = C + R + C2;
  const float* Wm = W + (size_t)m * Ct * KS2 + t;
  float v = 0.f;
  if (j < cnt) {
    const int o = seg[n] + j;
    v = Wm[(size_t)objs[o] * KS2];
    const float* r = repr + (size_t)o * R;
    for (int d = 0; d < R; ++d) v += r[d] * Wm[(size_t)(C + d) * KS2];
  } else if (j < cnt + C2) {
    v = Wm[(size_t)(C + R + (j - cnt)) * KS2];
  }
  wimg[i] = v;
}

// gW[m][c][t] for the three channel blocks: classes (sum over the objects of that class, ascending), appearance dims
// (sum_o repr[o][d] g[o]), second-source channels (sum over images)
__global__ void factored_weights_bwd_w_kernel(const float* __restrict__ gwimg, const float* __restrict__ repr,
                                              const int64_t* __restrict__ objs, const int32_t* __restrict__ seg,
                                              const int64_t* __restrict__ img_idx, float* __restrict__ gW, int N, int O, int M,
                                              int L, int KS2, int C, int R, int C2) {
  const int Ct = C + R + C2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Ct * KS2) return;
  const int t = (int)(i % KS2);
  const int c = (int)((i / KS2) % Ct);
  const int m = (int)(i / ((size_t)KS2 * Ct));
  float s = 0.f;
  if (c < C + R) {
    for (int o = 0; o < O; ++o) {
      float wgt;
      if (c < C) { if ((int)objs[o] != c) continue; wgt = 1.f; }
      else wgt = repr[(size_t)o * R + (c - C)];
      const int n = (int)img_idx[o], j = o - seg[n];
      s += wgt * gwimg[(((size_t)n * M + m) * L + j) * KS2 + t];
    }
  } else {
    const int e = c - C - R;
    for (int n = 0; n < N; ++n) s += gwimg[(((size_t)n * M + m) * L + (seg[n + 1] - seg[n]) + e) * KS2 + t];
  }
  gW[i] = s;
}

// Same sums, same order, with the per-object class id and gwimg slot staged in LDS once per workgroup (the kernel above
// re-reads objs / img_idx / seg from global memory in a serial, branchy loop over all objects: ~100 us per call).
__global__ void __launch_bounds__(256) factored_weights_bwd_w_lds_kernel(
    const float* __restrict__ gwimg, const float* __restrict__ repr, const int64_t* __restrict__ objs,
    const int32_t* __restrict__ seg, const int64_t* __restrict__ img_idx, float* __restrict__ gW, int N, int O, int M, int L,
    int KS2, int C, int R, int C2) {
  extern __shared__ int ws_lds[];
  int* cls = ws_lds;            // [O] class of object o
  int* base = ws_lds + O;       // [O] n_o*M*L + j_o
  for (int o = threadIdx.x; o < O; o += 256) {
    const int n = (int)img_idx[o];
    cls[o] = (int)objs[o];
    base[o] = n * M * L + (o - seg[n]);
  }
  __syncthreads();
  const int Ct = C + R + C2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Ct * KS2) return;
  const int t = (int)(i % KS2);
  const int c = (int)((i / KS2) % Ct);
  const int m = (int)(i / ((size_t)KS2 * Ct));
  const int mL = m * L;
  float s = 0.f;
  if (c < C) {
    for (int o = 0; o < O; ++o)
      if (cls[o] == c) s += gwimg[(size_t)(base[o] + mL) * KS2 + t];
  } else if (c < C + R) {
    const float* rp = repr + (c - C);
#pragma unroll 4
    for (int o = 0; o < O; ++o) s += rp[(size_t)o * R] * gwimg[(size_t)(base[o] + mL) * KS2 + t];
  } else {
    const int e = c - C - R;
    for (int n = 0; n < N; ++n) s += gwimg[(((size_t)n * M + m) * L + (seg[n + 1] - seg[n]) + e) * KS2 + t];
  }
  gW[i] = s;
}

// grepr[o][d] = sum_{m,t} gwimg[n_o][m][j_o][t] * W[m][C + d][t]: one wave per (o, d)
__global__ void __launch_bounds__(256) factored_weights_bwd_repr_kernel(const float* __restrict__ gwimg, const float* __restrict__ W,
                                                                       const int32_t* __restrict__ seg,
                                                                       const int64_t* __restrict__ img_idx,
                                                                       float* __restrict__ grepr, int O, int M, int L, int KS2,
                                                                       int C, int R, int C2) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= O * R) return;
  const int o = w / R, d = w - o * R;
  const int n = (int)img_idx[o], j = o - seg[n];
  const int Ct = C + R + C2;
  float s = 0.f;
  for (int q = lane; q < M * KS2; q += 64) {
    const int m = q / KS2, t = q - m * KS2;
    s += gwimg[(((size_t)n * M + m) * L + j) * KS2 + t] * W[((size_t)m * Ct + C + d) * KS2 + t];
  }
  s = sg_wave_sum(s);
  if (lane == 0) grepr[w] = s;
}

}  // namespace

extern "C" int sg_factored_weights_fwd(const float* w, const float* repr, const int64_t* objs, const int32_t* seg_off,
                                       float* wimg, int N, int O, int M, int L, int KS2, int C, int R, int C2, sgStream stream) {
  SG_ARG_CHECK(w && repr && objs && seg_off && wimg && N > 0 && O >= 0 && M > 0 && L > 0 && KS2 > 0 && C >= 0 && R >= 0 && C2 >= 0,
               "sg_factored_weights_fwd: bad arguments");
  const size_t n = (size_t)N * M * L * KS2;
  hipLaunchKernelGGL(factored_weights_fwd_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, repr, objs, seg_off,
                     wimg, N, M, L, KS2, C, R, C2);
  SG_LAUNCH_CHECK("sg_factored_weights_fwd");
  return 0;
}

extern "C" int sg_factored_weights_bwd(const float* gwimg, const float* w, const float* repr, const int64_t* objs,
                                       const int32_t* seg_off, const int64_t* img_idx, float* gw, float* grepr, int N, int O,
                                       int M, int L, int KS2, int C, int R, int C2, sgStream stream) {
  SG_ARG_CHECK(gwimg && w && repr && objs && seg_off && img_idx && (gw || grepr) && N > 0 && O > 0,
               "sg_factored_weights_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (gw) {
    const size_t n = (size_t)M * (C + R + C2) * KS2;
    if (O <= 4096 && (size_t)N * M * L < (1u << 30))
      hipLaunchKernelGGL(factored_weights_bwd_w_lds_kernel, dim3(sg_cdiv(n, 256)), dim3(256), (size_t)O * 2 * sizeof(int), s, gwimg,
                         repr, objs, seg_off, img_idx, gw, N, O, M, L, KS2, C, R, C2);
    else
      hipLaunchKernelGGL(factored_weights_bwd_w_kernel, dim3(sg_cdiv(n, 256)), dim3(256), 0, s, gwimg, repr, objs, seg_off, img_idx,
                         gw, N, O, M, L, KS2, C, R, C2);
  }
  if (grepr && R > 0)
    hipLaunchKernelGGL(factored_weights_bwd_repr_kernel, dim3(sg_cdiv((size_t)O * R, 4)), dim3(256), 0, s, gwimg, w, seg_off, img_idx,
                       grepr, O, M, L, KS2, C, R, C2);
  SG_LAUNCH_CHECK("sg_factored_weights_bwd");
  return 0;
}

// ================================================================================================
// masks_to_layout: gradients w.r.t. the masks and the boxes (layout.py:85-86 is differentiable in both; no loss of the
// training step uses them -- pred_layout only feeds TensorBoard -- but the operator surface offers them).
//   out[n,d,h,w] = sum_o vecs[o,d] * S_o[h,w],  S_o = bilinear(mask_o; px(w; x0,x1), py(h; y0,y1))
// Stage 1: T_o[h,w] = sum_d gout[n_o,d,h,w] * vecs[o,d]  (/ objects of the image for 'avg' pooling).
// Stage 2 (masks): g_mask[o,y,x] = sum_{h,w} T_o[h,w] * wy(h -> y) * wx(w -> x) as a GATHER over the pixels whose footprint
//   covers the mask element (the sample coordinate is affine in the pixel index: candidate ranges + exact tap check), summed
//   in (h, w) order: deterministic.
// Stage 3 (boxes): g_box[o] = sum_{h,w} T_o[h,w] * (dS/dpx * dpx/dx0, dS/dpy * dpy/dy0, dS/dpx * dpx/dx1, dS/dpy * dpy/dy1)
//   with grid_sample's own derivative (zeros padding: only in-range taps contribute); block reduction in fixed order.
// ================================================================================================
namespace {

__global__ void __launch_bounds__(256) layout_objmap_kernel(const float* __restrict__ gout, const float* __restrict__ vecs,
                                                           const int64_t* __restrict__ o2i, const int32_t* __restrict__ seg,
                                                           float* __restrict__ T, int D, int HW, int avg) {
  const int o = blockIdx.y;
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const int n = (int)o2i[o];
  const float* g = gout + (size_t)n * D * HW + px;
  const float* v = vecs + (size_t)o * D;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += g[(size_t)d * HW] * v[d];
  if (avg) { const int cnt = seg[n + 1] - seg[n]; s = s / (float)(cnt > 1 ? cnt : 1); }
  T[(size_t)o * HW + px] = s;
}

// candidate pixel range [lo, hi] whose bilinear footprint can touch mask index y along one axis
__device__ __forceinline__ void layout_axis_range(float b0, float b1, int n_pix, int M, int y, int& lo, int& hi, int ac) {
  const float p0 = tap_coord(box_coord(0.f, b0, b1), M, ac), p1 = tap_coord(box_coord(1.f, b0, b1), M, ac);
  lo = 0; hi = n_pix - 1;
  if (n_pix == 1 || !(fabsf(p0) < 1e8f) || !(fabsf(p1) < 1e8f)) return;
  const float s = (p1 - p0) / (float)(n_pix - 1);
  if (fabsf(s) < 1e-6f) return;
  float a = ((float)(y - 1) - p0) / s, b = ((float)(y + 1) - p0) / s;
  if (a > b) { const float t = a; a = b; b = t; }
  const float fl = floorf(a) - 1.f, fh = ceilf(b) + 1.f;
  lo = fl < 0.f ? 0 : (fl > (float)(n_pix - 1) ? n_pix : (int)fl);
  hi = fh > (float)(n_pix - 1) ? n_pix - 1 : (fh < 0.f ? -1 : (int)fh);
}

__global__ void __launch_bounds__(256) layout_bwd_masks_kernel(const float* __restrict__ T, const float* __restrict__ boxes,
                                                              float* __restrict__ gm, int M, int H, int W, int ac) {
  const int o = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= M * M) return;
  const int y = e / M, x = e - y * M;
  const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
  int hlo, hhi, wlo, whi;
  layout_axis_range(y0, y1, H, M, y, hlo, hhi, ac);
  layout_axis_range(x0, x1, W, M, x, wlo, whi, ac);
  const float* t = T + (size_t)o * H * W;
  float s = 0.f;
  for (int h = hlo; h <= hhi; ++h) {
    const Tap ty = make_tap(box_coord(lin01(h, H), y0, y1), M, ac);
    const float wy = (ty.i0 == y ? ty.w0 : 0.f) + (ty.i1 == y ? ty.w1 : 0.f);
    if (wy == 0.f) continue;
    for (int w = wlo; w <= whi; ++w) {
      const Tap tx = make_tap(box_coord(lin01(w, W), x0, x1), M, ac);
      const float wx = (tx.i0 == x ? tx.w0 : 0.f) + (tx.i1 == x ? tx.w1 : 0.f);
      if (wx != 0.f) s += t[h * W + w] * (wy * wx);
    }
  }
  gm[(size_t)o * M * M + e] = s;
}

template <bool I64>
__global__ void __launch_bounds__(256) layout_bwd_boxes_kernel(const float* __restrict__ T, const float* __restrict__ boxes,
                                                              const void* __restrict__ masks, float* __restrict__ gb, int M, int H,
                                                              int W, int ac) {
  __shared__ float red[16];
  const size_t o = blockIdx.x;
  const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
  const float dx = x1 - x0, dy = y1 - y0;
  const float scale = ac ? 0.5f * (float)(M - 1) : 0.5f * (float)M;          // d(pixel coordinate) / d(normalised coordinate)
  const float* t = T + o * H * W;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
  for (int p = threadIdx.x; p < H * W; p += 256) {
    const int h = p / W, w = p - h * W;
    const float Y = lin01(h, H), X = lin01(w, W);
    const Tap ty = make_tap(box_coord(Y, y0, y1), M, ac), tx = make_tap(box_coord(X, x0, x1), M, ac);
    // taps with zero weight because they are outside the mask contribute nothing to the derivative either (zeros padding)
    const float m00 = mask_at<I64>(masks, o, M, ty.i0, tx.i0), m01 = mask_at<I64>(masks, o, M, ty.i0, tx.i1);
    const float m10 = mask_at<I64>(masks, o, M, ty.i1, tx.i0), m11 = mask_at<I64>(masks, o, M, ty.i1, tx.i1);
    const float px = tap_coord(box_coord(X, x0, x1), M, ac), py = tap_coord(box_coord(Y, y0, y1), M, ac);
    const bool fin = fabsf(px) < 1e8f && fabsf(py) < 1e8f;
    const float flx = floorf(px), fly = floorf(py);
    // in-range indicators of the four taps (grid_sample's zeros padding: an out-of-range corner contributes nothing)
    const float ix0 = (flx >= 0.f && flx < (float)M) ? 1.f : 0.f, ix1 = (flx + 1.f >= 0.f && flx + 1.f < (float)M) ? 1.f : 0.f;
    const float iy0 = (fly >= 0.f && fly < (float)M) ? 1.f : 0.f, iy1 = (fly + 1.f >= 0.f && fly + 1.f < (float)M) ? 1.f : 0.f;
    const float dSdx = ((m01 * ix1 - m00 * ix0) * ty.w0 + (m11 * ix1 - m10 * ix0) * ty.w1);
    const float dSdy = ((m10 * iy1 - m00 * iy0) * tx.w0 + (m11 * iy1 - m01 * iy0) * tx.w1);
    const float tv = fin ? t[p] : 0.f;
    const float cx = tv * dSdx * scale * 2.f / (dx * dx), cy = tv * dSdy * scale * 2.f / (dy * dy);
    g0 += cx * (X - x1); g2 += -cx * (X - x0);
    g1 += cy * (Y - y1); g3 += -cy * (Y - y0);
  }
  g0 = sg_block_sum(g0, red); g1 = sg_block_sum(g1, red); g2 = sg_block_sum(g2, red); g3 = sg_block_sum(g3, red);
  if (threadIdx.x == 0) { gb[o * 4 + 0] = g0; gb[o * 4 + 1] = g1; gb[o * 4 + 2] = g2; gb[o * 4 + 3] = g3; }
}

}  // namespace

extern "C" size_t sg_masks_to_layout_bwd_geom_ws_bytes(int O, int H, int W) { return (size_t)(O > 0 ? O : 1) * H * W * sizeof(float); }

extern "C" int sg_masks_to_layout_bwd_geom(const float* gout, const float* vecs, const float* boxes, const void* masks, int masks_i64,
                                           const int64_t* obj_to_img, const int32_t* seg_off, float* g_masks, float* g_boxes,
                                           void* ws, size_t ws_bytes, int N, int O, int D, int M, int H, int W, int avg,
                                           sgStream stream) {
  SG_ARG_CHECK(gout && vecs && boxes && masks && obj_to_img && seg_off && ws && (g_masks || g_boxes) && N > 0 && O > 0 && D > 0,
               "sg_masks_to_layout_bwd_geom: bad arguments");
  SG_ARG_CHECK(ws_bytes >= sg_masks_to_layout_bwd_geom_ws_bytes(O, H, W), "sg_masks_to_layout_bwd_geom: workspace too small");
  SG_ARG_CHECK(!(g_masks && masks_i64), "sg_masks_to_layout_bwd_geom: integer masks have no gradient");
  hipStream_t s = (hipStream_t)stream;
  float* T = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(layout_objmap_kernel, dim3(sg_cdiv(H * W, 256), O), dim3(256), 0, s, gout, vecs, obj_to_img, seg_off, T, D, H * W, avg);
  if (g_masks)
    hipLaunchKernelGGL(layout_bwd_masks_kernel, dim3(sg_cdiv(M * M, 256), O), dim3(256), 0, s, (const float*)T, boxes, g_masks, M, H, W,
                       g_align_corners);
  if (g_boxes) {
    if (masks_i64) hipLaunchKernelGGL(layout_bwd_boxes_kernel<true>, dim3(O), dim3(256), 0, s, (const float*)T, boxes, masks, g_boxes, M, H, W, g_align_corners);
    else hipLaunchKernelGGL(layout_bwd_boxes_kernel<false>, dim3(O), dim3(256), 0, s, (const float*)T, boxes, masks, g_boxes, M, H, W, g_align_corners);
  }
  SG_LAUNCH_CHECK("sg_masks_to_layout_bwd_geom");
  return 0;
}
