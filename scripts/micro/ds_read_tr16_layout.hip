#include <hip/hip_runtime.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* in, _Float16* out) {
  __shared__ _Float16 sm[64 * 40];
  for (int i = threadIdx.x; i < 64 * 40; i += 64) sm[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  __attribute__((address_space(3))) fp16x4* a = (__attribute__((address_space(3))) fp16x4*)(sm + (4 * g + q) * 40 + 4 * p);
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(a);
  h4 r; r[0] = (_Float16)v[0]; r[1] = (_Float16)v[1]; r[2] = (_Float16)v[2]; r[3] = (_Float16)v[3];
  *(h4*)(out + l * 4) = r;
}
int main() {
  _Float16 h[64 * 40], o[256]; for (int i = 0; i < 64 * 40; ++i) h[i] = (_Float16)((i / 40) * 100 + (i % 40));
  _Float16 *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d, e); hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 5, 15, 16, 17, 63}) printf("lane %d: %g %g %g %g\n", l, (float)o[l*4], (float)o[l*4+1], (float)o[l*4+2], (float)o[l*4+3]);
}
