// EfficientNet MBConv plumbing for gfx950 (BASELINE config 4): everything of the block that is not a
// dense 1x1 convolution (those run on conv_igemm / wgrad_igemm): swish, depthwise k x k convolution with TF-"same"
// static padding (forward, dgrad, wgrad), squeeze-and-excitation (pool, two tiny FCs, channel scale) and the block
// output (BatchNorm-apply + drop-connect + identity skip), with their backward passes.  All of it is HBM-bound
// 16-byte-vectorised streaming work; correctness-first kernels (round 1), activations materialised between stages.
//
// Reference semantics: efficientnet_pytorch MBConvBlock as vendored by segmentation_models_pytorch
// (/root/reference/src/models/unet_model.py:64-71 with ENCODER_NAME efficientnet-b4).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
static constexpr int kMaxB = 256 * 8;
// channel-slice width for the column-reduction kernels: the largest divisor of C that is a multiple of 4 and <= 1024
static inline int pick_cw(int C) { for (int d = 1; d <= C; ++d) if (C % d == 0 && C / d <= 1024 && (C / d) % 4 == 0) return C / d; return 0; }
static inline unsigned nb(size_t work, int per) { size_t b = (work + per - 1) / per; if (b > (size_t)kMaxB) b = kMaxB; if (b < 1) b = 1; return (unsigned)b; }

__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + __expf(-z)); }
__device__ __forceinline__ f4 swish4(f4 z) { return (f4){z.x * sigm(z.x), z.y * sigm(z.y), z.z * sigm(z.z), z.w * sigm(z.w)}; }
__device__ __forceinline__ float dswish(float z) { const float s = sigm(z); return s * (1.f + z * (1.f - s)); }
__device__ __forceinline__ f4 dswish4(f4 z) { return (f4){dswish(z.x), dswish(z.y), dswish(z.z), dswish(z.w)}; }

// out = swish(y*scale + shift)
__global__ void swish_fwd_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh, int C,
                                 float* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % (size_t)C);
    *(f4*)(out + i * 4) = swish4(*(const f4*)(y + i * 4) * *(const f4*)(sc + c) + *(const f4*)(sh + c));
  }
}
hipError_t launch_swish_fwd(const float* y, const float* sc, const float* sh, int C, float* out, size_t npix, hipStream_t st) {
  const size_t n4 = npix * C / 4;
  hipLaunchKernelGGL(swish_fwd_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, y, sc, sh, C, out, n4);
  return hipGetLastError();
}
// out = g * swish'(y*scale + shift)  [ * optional extra: (g*s[n][c] + gpool[n][c]*inv_hw) instead of g ]
__global__ void swish_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ sc,
                                 const float* __restrict__ sh, int C, size_t hw, const float* __restrict__ se_s,
                                 const float* __restrict__ gpool, float inv_hw, float* __restrict__ out, size_t n4) {
  const size_t per_img = hw * (size_t)C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % (size_t)C);
    f4 gv = *(const f4*)(g + i * 4);
    if (se_s) {
      const size_t n = i / per_img;
      gv = gv * *(const f4*)(se_s + n * C + c) + *(const f4*)(gpool + n * C + c) * inv_hw;
    }
    const f4 z = *(const f4*)(y + i * 4) * *(const f4*)(sc + c) + *(const f4*)(sh + c);
    *(f4*)(out + i * 4) = gv * dswish4(z);
  }
}
hipError_t launch_swish_bwd(const float* g, const float* y, const float* sc, const float* sh, int C, int N, size_t hw,
                            const float* se_s, const float* gpool, float* out, hipStream_t st) {
  const size_t n4 = (size_t)N * hw * C / 4;
  hipLaunchKernelGGL(swish_bwd_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, g, y, sc, sh, C, hw, se_s, gpool,
                     (float)(1.0 / (double)hw), out, n4);
  return hipGetLastError();
}

// ---------------------------------------------------------------- depthwise convolution
// x [N][H][W][C] (plain), w [C][Kpad] with tap t of channel c at w[c*Kpad + t*4]; pb = pad at the begin of H and W
__global__ void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int Kpad, int k, int stride, int pb,
                              int H, int W, int C, int Ho, int Wo, float* __restrict__ y, size_t total) {
  const int C4 = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4; size_t p = i / C4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r) {
      const int hi = ho * stride - pb + r;
      if (hi < 0 || hi >= H) continue;
      for (int s = 0; s < k; ++s) {
        const int wi = wo * stride - pb + s;
        if (wi < 0 || wi >= W) continue;
        const f4 xv = *(const f4*)(x + (((size_t)n * H + hi) * W + wi) * C + c);
        const int t = (r * k + s) * 4;
        const f4 wv = {w[(size_t)c * Kpad + t], w[(size_t)(c + 1) * Kpad + t], w[(size_t)(c + 2) * Kpad + t], w[(size_t)(c + 3) * Kpad + t]};
        acc += xv * wv;
      }
    }
    *(f4*)(y + i * 4) = acc;
  }
}
hipError_t launch_dw_fwd(const float* x, const float* w, int Kpad, int k, int stride, int pb, int N, int H, int W, int C,
                         int Ho, int Wo, float* y, hipStream_t st) {
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(dw_fwd_kernel, dim3(nb(total, 256)), dim3(256), 0, st, x, w, Kpad, k, stride, pb, H, W, C, Ho, Wo, y, total);
  return hipGetLastError();
}
// dx[n][h][w][c] = sum over taps with (h + pb - r) divisible by stride of dy[n][(h+pb-r)/stride][..][c] * w[c][r][s]
__global__ void dw_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, int Kpad, int k, int stride, int pb,
                                int H, int W, int C, int Ho, int Wo, const float* __restrict__ addend, float* __restrict__ dx, size_t total) {
  const int C4 = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4; size_t p = i / C4;
    const int wi = (int)(p % W); p /= W;
    const int hi = (int)(p % H); const int n = (int)(p / H);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r) {
      const int hn = hi + pb - r;
      if (hn < 0 || (hn % stride) != 0) continue;
      const int ho = hn / stride;
      if (ho >= Ho) continue;
      for (int s = 0; s < k; ++s) {
        const int wn = wi + pb - s;
        if (wn < 0 || (wn % stride) != 0) continue;
        const int wo = wn / stride;
        if (wo >= Wo) continue;
        const f4 gv = *(const f4*)(dy + (((size_t)n * Ho + ho) * Wo + wo) * C + c);
        const int t = (r * k + s) * 4;
        const f4 wv = {w[(size_t)c * Kpad + t], w[(size_t)(c + 1) * Kpad + t], w[(size_t)(c + 2) * Kpad + t], w[(size_t)(c + 3) * Kpad + t]};
        acc += gv * wv;
      }
    }
    if (addend) acc += *(const f4*)(addend + i * 4);
    *(f4*)(dx + i * 4) = acc;
  }
}
hipError_t launch_dw_dgrad(const float* dy, const float* w, int Kpad, int k, int stride, int pb, int N, int H, int W, int C,
                           int Ho, int Wo, const float* addend, float* dx, hipStream_t st) {
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(dw_dgrad_kernel, dim3(nb(total, 256)), dim3(256), 0, st, dy, w, Kpad, k, stride, pb, H, W, C, Ho, Wo, addend, dx, total);
  return hipGetLastError();
}
// dw[c][tap] += sum over output pixels dy[pix][c] * x[pix*stride - pb + tap][c]; blockIdx.y = tap; a block strides over
// output pixels with (256 / (C/4)) pixel lanes per channel quad, LDS reduce, one atomic per (channel, tap) per block
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int Kpad, int k, int stride,
                                                       int pb, int H, int W, int C, int Ho, int Wo, int N, int CW, float* __restrict__ dw) {
  __shared__ float red[256 * 4];
  const int tap = blockIdx.y, r = tap / k, s = tap - r * k;
  const int c0 = blockIdx.z * CW;
  const int tc = CW / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t npix = (size_t)N * Ho * Wo;
  if (rx < tr) {
    for (size_t p = (size_t)blockIdx.x * tr + rx; p < npix; p += (size_t)gridDim.x * tr) {
      const int wo = (int)(p % Wo); size_t q = p / Wo;
      const int ho = (int)(q % Ho); const int n = (int)(q / Ho);
      const int hi = ho * stride - pb + r, wi = wo * stride - pb + s;
      if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
      acc += *(const f4*)(dy + p * C + c) * *(const f4*)(x + (((size_t)n * H + hi) * W + wi) * C + c);
    }
  }
  float* rr = red + threadIdx.x * 4;
  rr[0] = acc.x; rr[1] = acc.y; rr[2] = acc.z; rr[3] = acc.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 4; t += 256) {
    const int q = t / 4, e = t % 4;
    float sum = 0.f;
    for (int kk = 0; kk < tr; ++kk) sum += red[(kk * tc + q) * 4 + e];
    atomicAdd(dw + (size_t)(c0 + q * 4 + e) * Kpad + tap * 4, sum);
  }
}
hipError_t launch_dw_wgrad(const float* x, const float* dy, int Kpad, int k, int stride, int pb, int N, int H, int W, int C,
                           int Ho, int Wo, float* dw, hipStream_t st) {
  const int CW = pick_cw(C);
  if (!CW) return hipErrorInvalidValue;
  const int tr = 256 / (CW / 4);
  const size_t npix = (size_t)N * Ho * Wo;
  unsigned bx = (unsigned)((npix + (size_t)tr * 64 - 1) / ((size_t)tr * 64));
  if (bx > 512) bx = 512; if (bx < 1) bx = 1;
  hipLaunchKernelGGL(dw_wgrad_kernel, dim3(bx, k * k, C / CW), dim3(256), 0, st, x, dy, Kpad, k, stride, pb, H, W, C, Ho, Wo, N, CW, dw);
  return hipGetLastError();
}

// ---------------------------------------------------------------- per-channel sum / sum of squares (BatchNorm statistics)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ y, size_t npix, int C, int CW, double* ssum, double* ssq) {
  __shared__ float red[256 * 8];
  const int c0 = blockIdx.y * CW;
  const int tc = CW / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (rx < tr)
    for (size_t p = (size_t)blockIdx.x * tr + rx; p < npix; p += (size_t)gridDim.x * tr) {
      const f4 v = *(const f4*)(y + p * C + c);
      s1 += v; s2 += v * v;
    }
  float* r = red + threadIdx.x * 8;
  r[0] = s1.x; r[1] = s1.y; r[2] = s1.z; r[3] = s1.w; r[4] = s2.x; r[5] = s2.y; r[6] = s2.z; r[7] = s2.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 8; t += 256) {
    const int q = t / 8, e = t % 8;
    double acc = 0.0;
    for (int k = 0; k < tr; ++k) acc += (double)red[(k * tc + q) * 8 + e];
    if (e < 4) atomicAdd(ssum + c0 + q * 4 + e, acc); else atomicAdd(ssq + c0 + q * 4 + (e - 4), acc);
  }
}
hipError_t launch_colstats(const float* y, size_t npix, int C, double* ssum, double* ssq, hipStream_t st) {
  const int CW = pick_cw(C);
  if (!CW) return hipErrorInvalidValue;
  const int tr = 256 / (CW / 4);
  hipLaunchKernelGGL(colstats_kernel, dim3(nb(npix, tr * 8), C / CW), dim3(256), 0, st, y, npix, C, CW, ssum, ssq);
  return hipGetLastError();
}

// ---------------------------------------------------------------- squeeze-and-excitation
// mode 0: out[n][c] = mean over hw of a[n][hw][c] ; mode 1: out[n][c] = sum over hw of a * b
__global__ __launch_bounds__(256) void se_reduce_hw_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t hw, int C, int CW,
                                                           float scale, float* __restrict__ out) {
  __shared__ float red[256 * 4];
  const int n = blockIdx.y;
  const int c0 = blockIdx.z * CW;
  const int tc = CW / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 acc = {0, 0, 0, 0};
  if (rx < tr)
    for (size_t p = (size_t)blockIdx.x * tr + rx; p < hw; p += (size_t)gridDim.x * tr) {
      const size_t o = ((size_t)n * hw + p) * C + c;
      f4 v = *(const f4*)(a + o);
      if (b) v = v * *(const f4*)(b + o);
      acc += v;
    }
  float* r = red + threadIdx.x * 4;
  r[0] = acc.x; r[1] = acc.y; r[2] = acc.z; r[3] = acc.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 4; t += 256) {
    const int q = t / 4, e = t % 4;
    float sum = 0.f;
    for (int k = 0; k < tr; ++k) sum += red[(k * tc + q) * 4 + e];
    atomicAdd(out + (size_t)n * C + c0 + q * 4 + e, sum * scale);
  }
}
// out must be zeroed by the caller
hipError_t launch_se_reduce_hw(const float* a, const float* b, int N, size_t hw, int C, float scale, float* out, hipStream_t st) {
  const int CW = pick_cw(C);
  if (!CW) return hipErrorInvalidValue;
  const int tr = 256 / (CW / 4);
  unsigned bx = (unsigned)((hw + (size_t)tr * 32 - 1) / ((size_t)tr * 32));
  if (bx > 64) bx = 64; if (bx < 1) bx = 1;
  hipLaunchKernelGGL(se_reduce_hw_kernel, dim3(bx, N, C / CW), dim3(256), 0, st, a, b, hw, C, CW, scale, out);
  return hipGetLastError();
}
// one block per sample: hid = swish(W1 pool + b1) (pre-activation kept in `hpre`), s = sigmoid(W2 hid + b2)
// W1 [nsq][K1pad] over C channels, W2 [C][K2pad] over nsq
__global__ __launch_bounds__(256) void se_fc_fwd_kernel(const float* __restrict__ pool, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        int K1pad, const float* __restrict__ w2, const float* __restrict__ b2, int K2pad,
                                                        int C, int nsq, float* __restrict__ hpre, float* __restrict__ s) {
  extern __shared__ float sh[];        // [nsq] hidden (post-swish)
  const int n = blockIdx.x;
  const float* p = pool + (size_t)n * C;
  for (int j = threadIdx.x; j < nsq; j += blockDim.x) {
    float acc = b1[j];
    for (int c = 0; c < C; ++c) acc += w1[(size_t)j * K1pad + c] * p[c];
    hpre[(size_t)n * nsq + j] = acc;
    sh[j] = acc * sigm(acc);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = b2[c];
    for (int j = 0; j < nsq; ++j) acc += w2[(size_t)c * K2pad + j] * sh[j];
    s[(size_t)n * C + c] = sigm(acc);
  }
}
hipError_t launch_se_fc_fwd(const float* pool, const float* w1, const float* b1, int K1pad, const float* w2, const float* b2,
                            int K2pad, int N, int C, int nsq, float* hpre, float* s, hipStream_t st) {
  hipLaunchKernelGGL(se_fc_fwd_kernel, dim3(N), dim3(256), nsq * sizeof(float), st, pool, w1, b1, K1pad, w2, b2, K2pad, C, nsq, hpre, s);
  return hipGetLastError();
}
// backward of the two FCs for one sample per block: gs[n][c] = dL/ds ; writes gpool[n][c] = dL/dpool and accumulates
// the four parameter gradients with atomics
__global__ __launch_bounds__(256) void se_fc_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ s, const float* __restrict__ hpre,
                                                        const float* __restrict__ pool, const float* __restrict__ w1, int K1pad,
                                                        const float* __restrict__ w2, int K2pad, int C, int nsq, float* __restrict__ gpool,
                                                        float* gw1, float* gb1, float* gw2, float* gb2) {
  extern __shared__ float sh[];        // [C] gz2 ; [nsq] hid ; [nsq] gz1
  float* gz2 = sh; float* hid = sh + C; float* gz1 = hid + nsq;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float sv = s[(size_t)n * C + c];
    const float g = gs[(size_t)n * C + c] * sv * (1.f - sv);        // through the sigmoid
    gz2[c] = g;
    atomicAdd(gb2 + c, g);
  }
  for (int j = threadIdx.x; j < nsq; j += blockDim.x) { const float z = hpre[(size_t)n * nsq + j]; hid[j] = z * sigm(z); }
  __syncthreads();
  for (int i = threadIdx.x; i < C * nsq; i += blockDim.x) {           // gW2[c][j] += gz2[c] * hid[j]
    const int c = i / nsq, j = i - c * nsq;
    atomicAdd(gw2 + (size_t)c * K2pad + j, gz2[c] * hid[j]);
  }
  for (int j = threadIdx.x; j < nsq; j += blockDim.x) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += w2[(size_t)c * K2pad + j] * gz2[c];
    const float g = acc * dswish(hpre[(size_t)n * nsq + j]);
    gz1[j] = g;
    atomicAdd(gb1 + j, g);
  }
  __syncthreads();
  const float* p = pool + (size_t)n * C;
  for (int i = threadIdx.x; i < nsq * C; i += blockDim.x) {           // gW1[j][c] += gz1[j] * pool[c]
    const int j = i / C, c = i - j * C;
    atomicAdd(gw1 + (size_t)j * K1pad + c, gz1[j] * p[c]);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < nsq; ++j) acc += w1[(size_t)j * K1pad + c] * gz1[j];
    gpool[(size_t)n * C + c] = acc;
  }
}
hipError_t launch_se_fc_bwd(const float* gs, const float* s, const float* hpre, const float* pool, const float* w1, int K1pad,
                            const float* w2, int K2pad, int N, int C, int nsq, float* gpool, float* gw1, float* gb1, float* gw2,
                            float* gb2, hipStream_t st) {
  hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(N), dim3(256), (C + 2 * nsq) * sizeof(float), st, gs, s, hpre, pool, w1, K1pad, w2, K2pad,
                     C, nsq, gpool, gw1, gb1, gw2, gb2);
  return hipGetLastError();
}
// out[n][hw][c] = a[n][hw][c] * s[n][c]
__global__ void se_scale_kernel(const float* __restrict__ a, const float* __restrict__ s, size_t hw, int C, float* __restrict__ out, size_t n4) {
  const size_t per_img = hw * (size_t)C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % (size_t)C);
    const size_t n = i / per_img;
    *(f4*)(out + i * 4) = *(const f4*)(a + i * 4) * *(const f4*)(s + n * C + c);
  }
}
hipError_t launch_se_scale(const float* a, const float* s, int N, size_t hw, int C, float* out, hipStream_t st) {
  const size_t n4 = (size_t)N * hw * C / 4;
  hipLaunchKernelGGL(se_scale_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, a, s, hw, C, out, n4);
  return hipGetLastError();
}

// ---------------------------------------------------------------- block output
// out = (y*scale + shift) * rowscale[n] + id      (rowscale: drop-connect keep/(1-p) per sample or nullptr; id or nullptr)
__global__ void mb_out_kernel(const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                              const float* __restrict__ rowscale, const float* __restrict__ id, size_t hw, int C,
                              float* __restrict__ out, size_t n4) {
  const size_t per_img = hw * (size_t)C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % (size_t)C);
    f4 v = *(const f4*)(y + i * 4) * *(const f4*)(sc + c) + *(const f4*)(sh + c);
    if (rowscale) v = v * rowscale[i / per_img];
    if (id) v += *(const f4*)(id + i * 4);
    *(f4*)(out + i * 4) = v;
  }
}
hipError_t launch_mb_out(const float* y, const float* sc, const float* sh, const float* rowscale, const float* id, int N, size_t hw,
                         int C, float* out, hipStream_t st) {
  const size_t n4 = (size_t)N * hw * C / 4;
  hipLaunchKernelGGL(mb_out_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, y, sc, sh, rowscale, id, hw, C, out, n4);
  return hipGetLastError();
}
// out = g * rowscale[n]   (+ optional second destination: acc += g, the identity-skip gradient)
__global__ void rowscale_kernel(const float* __restrict__ g, const float* __restrict__ rowscale, size_t hw, int C, float* __restrict__ out,
                                size_t n4) {
  const size_t per_img = hw * (size_t)C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    *(f4*)(out + i * 4) = *(const f4*)(g + i * 4) * rowscale[i / per_img];
}
hipError_t launch_rowscale(const float* g, const float* rowscale, int N, size_t hw, int C, float* out, hipStream_t st) {
  const size_t n4 = (size_t)N * hw * C / 4;
  hipLaunchKernelGGL(rowscale_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, g, rowscale, hw, C, out, n4);
  return hipGetLastError();
}
// acc += g
__global__ void accum_kernel(const float* __restrict__ g, float* __restrict__ acc, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    *(f4*)(acc + i * 4) = *(const f4*)(acc + i * 4) + *(const f4*)(g + i * 4);
}
hipError_t launch_accum(const float* g, float* acc, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(accum_kernel, dim3(nb(n4, 256)), dim3(256), 0, st, g, acc, n4);
  return hipGetLastError();
}

}  // namespace uwm
