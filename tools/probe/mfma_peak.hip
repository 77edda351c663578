// Pure f32-MFMA ceiling probe: 4 waves/block, NACC independent accumulators per wave, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, int iters) {
  float* d; hipMalloc(&d, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
  printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  run<1>(256, 20000); run<1>(512, 20000); run<1>(1024, 20000);
  run<4>(256, 5000); run<4>(512, 5000);
  return 0;
}
