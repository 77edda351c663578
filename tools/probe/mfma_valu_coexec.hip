// Does VALU / SALU / LDS work overlap with the f32 MFMA on gfx950?  One or two waves per SIMD run 4 independent
// v_mfma_f32_32x32x2_f32 per iteration plus NV independent integer VALU adds (NS scalar adds, NL LDS reads) per MFMA; if the
// extra instructions were free the time would not move.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NS, int NL>
__global__ void __launch_bounds__(512) probe(float* out, int iters, float a0, float b0, int seed) {
  __shared__ float lds[1024];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  lds[threadIdx.x & 1023] = a0;
  __syncthreads();
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  int v[8]; for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  int sc = seed;
  float lsum = 0.f;
  const float* lp = lds + (threadIdx.x & 255);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]));
#pragma unroll
      for (int k = 0; k < NS; ++k) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc));
#pragma unroll
      for (int k = 0; k < NL; ++k) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"((int)(size_t)lp), "i"(k * 4)); asm volatile("" :: "v"(t)); }
    }
  }
  float s = lsum + (float)sc;
  for (int i = 0; i < 8; ++i) s += (float)v[i];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NS, int NL> void run(int threads, const char* what) {
  const int blocks = 256, iters = 4000;
  float* d; hipMalloc(&d, blocks * threads * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NV, NS, NL><<<blocks, threads>>>(d, iters, 1.0f, 0.5f, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NV, NS, NL><<<blocks, threads>>>(d, iters, 1.0f, 0.5f, 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * (threads / 64) * iters * 4;
  printf("%-28s waves/SIMD %d  per MFMA: VALU %d SALU %d LDS %d : %.3f ms  %.1f TFLOP/s  %.1f ns/MFMA/SIMD\n", what, threads / 256, NV, NS, NL, ms,
         mf * 2.0 * 32 * 32 * 2 / ms / 1e9, ms * 1e6 / (iters * 4.0 * (threads / 256)));
  hipFree(d);
}
int main() {
  for (int t = 256; t <= 512; t += 256) {
    run<0, 0, 0>(t, "mfma only");
    run<1, 0, 0>(t, "+1 VALU");
    run<2, 0, 0>(t, "+2 VALU");
    run<4, 0, 0>(t, "+4 VALU");
    run<8, 0, 0>(t, "+8 VALU");
    run<0, 4, 0>(t, "+4 SALU");
    run<0, 8, 0>(t, "+8 SALU");
    run<0, 0, 1>(t, "+1 LDS read");
    run<0, 0, 2>(t, "+2 LDS read");
    run<4, 4, 1>(t, "+4 VALU +4 SALU +1 LDS");
  }
  return 0;
}
