// Pure v_mfma_f32_16x16x4_f32 issue-rate microbenchmark (operands in registers, NACC independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  f4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int NACC>
void run(int blocks_per_cu, int threads) {
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus * blocks_per_cu, iters = 20000;
  float* out; hipMalloc(&out, (size_t)grid * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, 100, 1.0f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * (threads / 64) * iters * NACC * 2048.0;
  printf("NACC=%2d blocks/CU=%d threads=%d waves/SIMD=%.1f : %8.3f ms  %7.2f TFLOP/s\n", NACC, blocks_per_cu, threads,
         blocks_per_cu * threads / 256.0, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<18>(1, 256); run<18>(2, 256); run<18>(4, 256); run<18>(1, 512); run<8>(2, 256); run<4>(2, 256); run<2>(2, 256); run<2>(1, 256); run<1>(2, 256);
  return 0;
}
