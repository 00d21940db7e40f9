// Fused Dice + BCE-with-logits loss (forward sums + gradient), confusion-matrix stats and the
// predict-time threshold, for gfx950.  One read pass over logits/targets produces the four
// batch-global sums (sum p*t, sum p, sum t, sum bce) with wavefront + block reductions in fp64;
// a second elementwise pass writes dL/dlogits.
//
// Reference semantics replaced (SURVEY.md §8 a12,a13,a16,a17, Appendix A.5/A.6):
//   smp.losses.DiceLoss(mode='binary', smooth)      <- /root/reference/src/utils/losses.py:18-19
//   nn.BCEWithLogitsLoss, CombinedLoss              <- /root/reference/src/utils/losses.py:22-23,33-52
//   smp.metrics.get_stats(mode='binary', thr=0.5)   <- /root/reference/src/utils/metrics.py:15-19
//   logits > THRESHOLD -> {0,255}                   <- /root/reference/src/predict.py:614-625
#include "uwm_kernels.h"

namespace uwm {

// target dtypes: 0 = float32, 1 = int64, 2 = uint8, 3 = int32
__device__ __forceinline__ float load_target(const void* t, int dt, size_t i) {
  switch (dt) {
    case 0: return ((const float*)t)[i];
    case 1: return (float)((const long long*)t)[i];
    case 2: return (float)((const unsigned char*)t)[i];
    default: return (float)((const int*)t)[i];
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// scratch: [0]=sum p*t [1]=sum p [2]=sum t [3]=sum bce
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ logits, int ld, const void* __restrict__ target,
                                                          int tdtype, size_t n, double* scratch) {
  float s_pt = 0.f, s_p = 0.f, s_t = 0.f, s_b = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = logits[i * ld];
    const float t = load_target(target, tdtype, i);
    const float sp = log1pf(expf(-fabsf(x)));        // softplus(-|x|)
    const float p = expf(fminf(x, 0.f) - sp);        // exp(logsigmoid(x))
    s_pt += p * t; s_p += p; s_t += t;
    s_b += fmaxf(x, 0.f) - x * t + sp;
  }
  __shared__ double red[4][4];
  double v0 = wave_sum((double)s_pt), v1 = wave_sum((double)s_p), v2 = wave_sum((double)s_t), v3 = wave_sum((double)s_b);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = v0; red[wave][1] = v1; red[wave][2] = v2; red[wave][3] = v3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(scratch + threadIdx.x, s);
  }
}

struct LossCoef { double A, B; double ldice, lbce; };
__device__ __forceinline__ LossCoef loss_coef(const double* s, double n, float smooth, float eps) {
  LossCoef c;
  const double I = s[0], card = s[1] + s[2], T = s[2];
  const double den = card + (double)smooth;
  const double denc = den > (double)eps ? den : (double)eps;
  const double score = (2.0 * I + (double)smooth) / denc;
  const bool on = T > 0.0;
  c.ldice = on ? 1.0 - score : 0.0;
  c.lbce = s[3] / n;
  // d(1-score)/dx_i = -(A*t_i - B) * p_i (1-p_i); the clamp kills the denominator's derivative
  if (on) { c.A = 2.0 / denc; c.B = den > (double)eps ? (2.0 * I + (double)smooth) / (denc * denc) : 0.0; }
  else { c.A = 0.0; c.B = 0.0; }
  return c;
}

__global__ void loss_finalize_kernel(const double* scratch, double n, float w_dice, float w_bce, float smooth, float eps,
                                     float* out3) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const LossCoef c = loss_coef(scratch, n, smooth, eps);
    out3[0] = (float)((double)w_dice * c.ldice + (double)w_bce * c.lbce);
    out3[1] = (float)c.ldice;
    out3[2] = (float)c.lbce;
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ logits, int ld, const void* __restrict__ target,
                                                        int tdtype, size_t n, double ntotal, const double* scratch, float w_dice, float w_bce,
                                                        float smooth, float eps, float* __restrict__ dlogits, int ldd,
                                                        float gscale) {
  // ntotal: pixels behind the sums in `scratch` (== n unless the sums were all-reduced over data-parallel ranks)
  const LossCoef c = loss_coef(scratch, ntotal, smooth, eps);
  const float A = (float)c.A, B = (float)c.B, invn = (float)(1.0 / ntotal);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = logits[i * ld];
    const float t = load_target(target, tdtype, i);
    const float sp = log1pf(expf(-fabsf(x)));
    const float p = expf(fminf(x, 0.f) - sp);
    const float gd = -(A * t - B) * p * (1.f - p);
    const float gb = (p - t) * invn;
    const float g = gscale * (w_dice * gd + w_bce * gb);
    float* o = dlogits + i * ldd;
    o[0] = g;
    for (int k = 1; k < ldd; ++k) o[k] = 0.f;
  }
}

static unsigned loss_blocks(size_t n) { size_t nb = (n + 1023) / 1024; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1; return (unsigned)nb; }
// first half of the loss: scratch4 = local {sum p*t, sum p, sum t, sum bce}
hipError_t launch_loss_sums(const float* logits, int ld, const void* target, int tdtype, size_t n, double* scratch4, hipStream_t st) {
  hipError_t e = hipMemsetAsync(scratch4, 0, 4 * sizeof(double), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(loss_blocks(n)), dim3(256), 0, st, logits, ld, target, tdtype, n, scratch4);
  return hipGetLastError();
}
// second half: loss value and dL/dlogits from the sums in scratch4, which cover ntotal pixels (>= n when all-reduced)
hipError_t launch_loss_apply(const float* logits, int ld, const void* target, int tdtype, size_t n, double ntotal, float w_dice, float w_bce,
                             float smooth, float eps, const double* scratch4, float* loss_out3, float* dlogits, int ldd,
                             float grad_scale, hipStream_t st) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, scratch4, ntotal, w_dice, w_bce, smooth, eps, loss_out3);
  if (dlogits)
    hipLaunchKernelGGL(loss_grad_kernel, dim3(loss_blocks(n)), dim3(256), 0, st, logits, ld, target, tdtype, n, ntotal, scratch4, w_dice,
                       w_bce, smooth, eps, dlogits, ldd, grad_scale);
  return hipGetLastError();
}
hipError_t launch_loss(const float* logits, int ld, const void* target, int tdtype, size_t n, float w_dice, float w_bce,
                       float smooth, float eps, double* scratch4, float* loss_out3, float* dlogits, int ldd,
                       float grad_scale, hipStream_t st) {
  hipError_t e = launch_loss_sums(logits, ld, target, tdtype, n, scratch4, st);
  if (e != hipSuccess) return e;
  return launch_loss_apply(logits, ld, target, tdtype, n, (double)n, w_dice, w_bce, smooth, eps, scratch4, loss_out3, dlogits, ldd, grad_scale, st);
}

// ---------------------------------------------------------------- tp / fp / fn / tn per image (int64)
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ logits, int ld, const void* __restrict__ target,
                                                    int tdtype, size_t hw, float thr, int apply_sigmoid, long long* out4) {
  const int n = blockIdx.y;
  unsigned tp = 0, po = 0, pt = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
    const size_t g = (size_t)n * hw + i;
    float v = logits[g * ld];
    if (apply_sigmoid) v = 1.f / (1.f + expf(-v));
    const unsigned o = v >= thr ? 1u : 0u;
    const unsigned t = load_target(target, tdtype, g) != 0.f ? 1u : 0u;
    tp += o & t; po += o; pt += t;
  }
  __shared__ unsigned red[4][3];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { tp += __shfl_xor(tp, d); po += __shfl_xor(po, d); pt += __shfl_xor(pt, d); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = tp; red[wave][1] = po; red[wave][2] = pt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, b = 0, c = 0;
    for (int w = 0; w < 4; ++w) { a += red[w][0]; b += red[w][1]; c += red[w][2]; }
    unsigned long long* o = (unsigned long long*)(out4 + (size_t)n * 4);
    atomicAdd(o + 0, a);             // tp
    atomicAdd(o + 1, b - a);         // fp
    atomicAdd(o + 2, c - a);         // fn
    if (blockIdx.x == 0) atomicAdd(o + 3, (unsigned long long)hw);   // tn = hw - tp - fp - fn (finished on host side of ABI)
  }
}
__global__ void stats_fix_kernel(long long* out4, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) out4[n * 4 + 3] -= out4[n * 4] + out4[n * 4 + 1] + out4[n * 4 + 2];
}
hipError_t launch_stats(const float* logits, int ld, const void* target, int tdtype, int N, size_t hw, float thr,
                        int apply_sigmoid, long long* out4, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out4, 0, (size_t)N * 4 * sizeof(long long), st);
  if (e != hipSuccess) return e;
  size_t nb = (hw + 2047) / 2048; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(stats_kernel, dim3((unsigned)nb, (unsigned)N), dim3(256), 0, st, logits, ld, target, tdtype, hw, thr,
                     apply_sigmoid, out4);
  hipLaunchKernelGGL(stats_fix_kernel, dim3((N + 63) / 64), dim3(64), 0, st, out4, N);
  return hipGetLastError();
}

__global__ void threshold_kernel(const float* __restrict__ logits, int ld, size_t n, float thr, int apply_sigmoid,
                                 uint8_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = logits[i * ld];
    if (apply_sigmoid) v = 1.f / (1.f + expf(-v));
    out[i] = v > thr ? 255 : 0;
  }
}
hipError_t launch_threshold(const float* logits, int ld, size_t npix, float thr, int apply_sigmoid, uint8_t* out,
                            hipStream_t st) {
  size_t nb = (npix + 1023) / 1024; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)nb), dim3(256), 0, st, logits, ld, npix, thr, apply_sigmoid, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------- bilinear resize (cv2.INTER_LINEAR / torch
// align_corners=False: src = (dst + 0.5) * in/out - 0.5, edge-clamped) of one logit plane per image + threshold
__global__ void resize_threshold_kernel(const float* __restrict__ logits, int ld, int h, int w, int H, int W, float thr,
                                        int apply_sigmoid, uint8_t* __restrict__ out, float* __restrict__ out_f, size_t total) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W); size_t r = i / W; const int Y = (int)(r % H); const int n = (int)(r / H);
    float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
    fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = min(y0, h - 1); x0 = min(x0, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* b = logits + (size_t)n * h * w * ld;
    const float v00 = b[((size_t)y0 * w + x0) * ld], v01 = b[((size_t)y0 * w + x1) * ld];
    const float v10 = b[((size_t)y1 * w + x0) * ld], v11 = b[((size_t)y1 * w + x1) * ld];
    float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    if (apply_sigmoid) v = 1.f / (1.f + expf(-v));
    if (out_f) out_f[i] = v;
    if (out) out[i] = v > thr ? 255 : 0;
  }
}
hipError_t launch_resize_threshold(const float* logits, int ld, int N, int h, int w, int H, int W, float thr, int apply_sigmoid,
                                   uint8_t* out, float* out_f, hipStream_t st) {
  const size_t total = (size_t)N * H * W;
  size_t nb = (total + 1023) / 1024; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(resize_threshold_kernel, dim3((unsigned)nb), dim3(256), 0, st, logits, ld, h, w, H, W, thr, apply_sigmoid, out,
                     out_f, total);
  return hipGetLastError();
}

}  // namespace uwm
