/* ORACLE (test infrastructure, NOT product code): plain-C restatement of the three "indexing" operators of the
 * scene-graph -> image hot path, written from the reference's behaviour (paths relative to
 * /root/reference/scene_generation/).  Scalar, single-threaded, no dependencies.  Used only by tests/ (bit-exact
 * checker of the graph pool; independent fp32 statement of the layout / crop arithmetic).  Pinned against the
 * reference's golden vectors by tests/test_oracle_golden.py::test_c_oracle_*.
 *
 *   ora_pool_triples    graph.py:94-116   scatter_add s-pass THEN o-pass, t ascending; clamp(min=1); true division
 *   ora_masks_to_layout layout.py:64-93,96-128,131-155   (factored form)
 *   ora_masks_to_layout_test layout.py:87-92,157-169    test-mode compositing (ascending-mass order, first hit wins)
 *   ora_crop_bbox       bilinear.py:67-130,246-275       
 * grid_sample semantics = torch >= 1.3 default: bilinear, zeros padding, align_corners=False.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void ora_pool_triples(const float* new_t, const int64_t* edges, int T, int O, int H, int Dout, int avg, float* pooled) {
  const int ld = 2 * H + Dout;
  float* cnt = (float*)calloc((size_t)O, sizeof(float));
  memset(pooled, 0, sizeof(float) * (size_t)O * H);
  for (int t = 0; t < T; ++t) {                       /* pass 1: subjects */
    float* dst = pooled + (size_t)edges[2 * t] * H;
    const float* src = new_t + (size_t)t * ld;
    for (int c = 0; c < H; ++c) dst[c] += src[c];
  }
  for (int t = 0; t < T; ++t) {                       /* pass 2: objects */
    float* dst = pooled + (size_t)edges[2 * t + 1] * H;
    const float* src = new_t + (size_t)t * ld + H + Dout;
    for (int c = 0; c < H; ++c) dst[c] += src[c];
  }
  if (avg) {
    for (int t = 0; t < T; ++t) cnt[edges[2 * t]] += 1.0f;
    for (int t = 0; t < T; ++t) cnt[edges[2 * t + 1]] += 1.0f;
    for (int i = 0; i < O; ++i) {
      const float d = cnt[i] < 1.0f ? 1.0f : cnt[i];
      for (int c = 0; c < H; ++c) pooled[(size_t)i * H + c] = pooled[(size_t)i * H + c] / d;
    }
  }
  free(cnt);
}

/* torch.linspace(0,1,n)[j] as ATen computes it (symmetric halves) */
static float lin01(int j, int n) {
  if (n == 1) return 0.0f;
  const float step = 1.0f / (float)(n - 1);
  return j < n / 2 ? step * (float)j : 1.0f - step * (float)(n - 1 - j);
}
static float lin10(int j, int n) {
  if (n == 1) return 1.0f;
  const float step = -1.0f / (float)(n - 1);
  return j < n / 2 ? 1.0f + step * (float)j : 0.0f - step * (float)(n - 1 - j);
}

typedef struct { int i0, i1; float w0, w1; } tap_t;

static tap_t make_tap(float g, int size) {          /* ((g+1)*size-1)/2, floor, zeros outside */
  tap_t t;
  const float p = ((g + 1.0f) * (float)size - 1.0f) * 0.5f;
  const float f = floorf(p);
  t.i0 = (int)f; t.i1 = t.i0 + 1;
  t.w1 = p - f; t.w0 = (f + 1.0f) - p;
  if (t.i0 < 0 || t.i0 >= size) { t.w0 = 0.0f; t.i0 = 0; }
  if (t.i1 < 0 || t.i1 >= size) { t.w1 = 0.0f; t.i1 = 0; }
  return t;
}

void ora_masks_to_layout(const float* vecs, const float* boxes, const float* masks, const int64_t* obj_to_img, int O,
                         int D, int M, int N, int H, int W, int avg, float* out) {
  float* S = (float*)malloc(sizeof(float) * (size_t)H * W);
  int* count = (int*)calloc((size_t)N, sizeof(int));
  int* seen = (int*)calloc((size_t)N, sizeof(int));
  for (int o = 0; o < O; ++o) count[obj_to_img[o]]++;
  for (int o = 0; o < O; ++o) {                       /* ascending o inside each image */
    const int n = (int)obj_to_img[o];
    const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
    const float* mk = masks + (size_t)o * M * M;
    for (int h = 0; h < H; ++h) {
      const tap_t ty = make_tap(((lin01(h, H) - y0) / (y1 - y0)) * 2.0f - 1.0f, M);
      for (int w = 0; w < W; ++w) {
        const tap_t tx = make_tap(((lin01(w, W) - x0) / (x1 - x0)) * 2.0f - 1.0f, M);
        float v = mk[ty.i0 * M + tx.i0] * (ty.w0 * tx.w0);
        v += mk[ty.i0 * M + tx.i1] * (ty.w0 * tx.w1);
        v += mk[ty.i1 * M + tx.i0] * (ty.w1 * tx.w0);
        v += mk[ty.i1 * M + tx.i1] * (ty.w1 * tx.w1);
        S[h * W + w] = v;
      }
    }
    for (int d = 0; d < D; ++d) {
      float* dst = out + ((size_t)n * D + d) * H * W;
      const float c = vecs[(size_t)o * D + d];
      if (!seen[n]) for (int p = 0; p < H * W; ++p) dst[p] = c * S[p];
      else for (int p = 0; p < H * W; ++p) dst[p] = dst[p] + c * S[p];
    }
    seen[n] = 1;
  }
  for (int n = 0; n < N; ++n) {
    if (!seen[n]) memset(out + (size_t)n * D * H * W, 0, sizeof(float) * (size_t)D * H * W);
    if (avg && count[n] > 1)
      for (size_t p = 0; p < (size_t)D * H * W; ++p) out[(size_t)n * D * H * W + p] /= (float)count[n];
  }
  free(S); free(count); free(seen);
}

/* bilinear-sampled mask of object o at every pixel (the factor S_o of the layout) */
static void sample_plane(const float* boxes, const float* masks, int o, int M, int H, int W, float* S) {
  const float x0 = boxes[o * 4 + 0], y0 = boxes[o * 4 + 1], x1 = boxes[o * 4 + 2], y1 = boxes[o * 4 + 3];
  const float* mk = masks + (size_t)o * M * M;
  for (int h = 0; h < H; ++h) {
    const tap_t ty = make_tap(((lin01(h, H) - y0) / (y1 - y0)) * 2.0f - 1.0f, M);
    for (int w = 0; w < W; ++w) {
      const tap_t tx = make_tap(((lin01(w, W) - x0) / (x1 - x0)) * 2.0f - 1.0f, M);
      float v = mk[ty.i0 * M + tx.i0] * (ty.w0 * tx.w0);
      v += mk[ty.i0 * M + tx.i1] * (ty.w0 * tx.w1);
      v += mk[ty.i1 * M + tx.i0] * (ty.w1 * tx.w0);
      v += mk[ty.i1 * M + tx.i1] * (ty.w1 * tx.w1);
      S[h * W + w] = v;
    }
  }
}

/* layout.py:157-169: per image the objects are visited in ascending mass = sum_{d,h,w} vecs[o,d] * S_o[h,w] (stable for
 * ties, like numpy's insertion sort on the short lists involved); a pixel takes vecs[o] * S_o of the first visited object
 * whose sampled mask exceeds 0.5, zero if none.  obj_to_img sorted (layout.py:153-154). */
void ora_masks_to_layout_test(const float* vecs, const float* boxes, const float* masks, const int64_t* obj_to_img, int O,
                              int D, int M, int N, int H, int W, int avg, float* out) {
  const int HW = H * W;
  float* S = (float*)malloc(sizeof(float) * (size_t)O * HW);
  double* mass = (double*)malloc(sizeof(double) * (size_t)O);
  int* order = (int*)malloc(sizeof(int) * (size_t)O);
  memset(out, 0, sizeof(float) * (size_t)N * D * HW);
  for (int o = 0; o < O; ++o) {
    sample_plane(boxes, masks, o, M, H, W, S + (size_t)o * HW);
    double sv = 0.0, ss = 0.0;
    for (int d = 0; d < D; ++d) sv += (double)vecs[(size_t)o * D + d];
    for (int p = 0; p < HW; ++p) ss += (double)S[(size_t)o * HW + p];
    mass[o] = sv * ss;
  }
  int beg = 0;
  while (beg < O) {
    const int n = (int)obj_to_img[beg];
    int end = beg;
    while (end < O && obj_to_img[end] == n) ++end;
    const int cnt = end - beg;
    for (int j = 0; j < cnt; ++j) order[j] = beg + j;
    for (int j = 1; j < cnt; ++j) {                  /* insertion sort: stable */
      const int key = order[j];
      int k = j - 1;
      while (k >= 0 && mass[order[k]] > mass[key]) { order[k + 1] = order[k]; --k; }
      order[k + 1] = key;
    }
    const float div = (avg && cnt > 1) ? (float)cnt : 1.0f;
    for (int p = 0; p < HW; ++p) {
      for (int j = 0; j < cnt; ++j) {
        const int o = order[j];
        const float s = S[(size_t)o * HW + p];
        if (s > 0.5f) {
          for (int d = 0; d < D; ++d) out[((size_t)n * D + d) * HW + p] = vecs[(size_t)o * D + d] * s / div;
          break;
        }
      }
    }
    beg = end;
  }
  free(S); free(mass); free(order);
}

void ora_crop_bbox(const float* feats, const float* boxes, const int64_t* box_to_feat, int C, int H, int W, int B,
                   int HH, int WW, float* out) {
  for (int b = 0; b < B; ++b) {
    const float x0 = 2.0f * boxes[b * 4 + 0] - 1.0f, y0 = 2.0f * boxes[b * 4 + 1] - 1.0f;
    const float x1 = 2.0f * boxes[b * 4 + 2] - 1.0f, y1 = 2.0f * boxes[b * 4 + 3] - 1.0f;
    const float* f0 = feats + (size_t)box_to_feat[b] * C * H * W;
    for (int y = 0; y < HH; ++y) {
      const tap_t ty = make_tap(lin10(y, HH) * y0 + lin01(y, HH) * y1, H);
      for (int x = 0; x < WW; ++x) {
        const tap_t tx = make_tap(lin10(x, WW) * x0 + lin01(x, WW) * x1, W);
        for (int c = 0; c < C; ++c) {
          const float* f = f0 + (size_t)c * H * W;
          float v = f[ty.i0 * W + tx.i0] * (ty.w0 * tx.w0);
          v += f[ty.i0 * W + tx.i1] * (ty.w0 * tx.w1);
          v += f[ty.i1 * W + tx.i0] * (ty.w1 * tx.w0);
          v += f[ty.i1 * W + tx.i1] * (ty.w1 * tx.w1);
          out[(((size_t)b * C + c) * HH + y) * WW + x] = v;
        }
      }
    }
  }
}
