// Operand-layout check of v_mfma_f32_16x16x16_bf16 on gfx950 with exact small-integer data (asymmetric A and B):
// assumed lane map  A[row = l & 15][k = 4*(l >> 4) + j],  B[k = 4*(l >> 4) + j][col = l & 15],  C[row = 4*(l >> 4) + e][col = l & 15].
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_bf16_layout.hip -o scripts/micro/mfma_bf16_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ unsigned short bf(float x) { return (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16); }
__global__ void k(float* out) {
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  s4 a, b;
  for (int j = 0; j < 4; ++j) {
    const int kk = 4 * q + j;
    a[j] = (short)bf((float)(r + 2 * kk + 1));          // A[r][kk] = r + 2kk + 1
    b[j] = (short)bf((float)(3 * kk - r + 5));          // B[kk][c = r] = 3kk - c + 5
  }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  for (int e = 0; e < 4; ++e) out[(4 * q + e) * 16 + r] = c[e];
}
int main() {
  float* d; hipMalloc((void**)&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int c = 0; c < 16; ++c) {
    float ref = 0; for (int kk = 0; kk < 16; ++kk) ref += (float)(i + 2 * kk + 1) * (float)(3 * kk - c + 5);
    if (h[i * 16 + c] != ref) { if (bad < 5) printf("mismatch C[%d][%d] = %g, expected %g\n", i, c, h[i * 16 + c], ref); ++bad; }
  }
  printf("mfma_f32_16x16x16_bf16 layout check: %d mismatches of 256\n", bad);
  return bad != 0;
}
