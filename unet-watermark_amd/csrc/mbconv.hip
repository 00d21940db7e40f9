// EfficientNet MBConv plumbing for gfx950 (BASELINE config 4): everything of the block that is not a
// dense 1x1 convolution (those run on conv_igemm / wgrad_igemm): swish, depthwise k x k convolution with TF-"same"
// static padding (forward, dgrad, wgrad), squeeze-and-excitation (pool, two tiny FCs, channel scale) and the block
// output (BatchNorm-apply + drop-connect + identity skip), with their backward passes.  All of it is HBM-bound
// 16-byte-vectorised streaming work; correctness-first kernels (round 1), activations materialised between stages.
//
// Reference semantics: efficientnet_pytorch MBConvBlock as vendored by segmentation_models_pytorch
// (/root/reference/src/models/unet_model.py:64-71 with ENCODER_NAME efficientnet-b4).
#include "uwm_kernels.h"
#include <algorithm>

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
// ---------------------------------------------------------------- depthwise convolution
// x [N][H][W][C] plain NHWC; w [k*k][C] (tap-major: tap t of channel c at w[t*C + c], the parameter-arena layout of a
// depthwise layer, so a thread's four channels of one tap are ONE 16-byte load); pb = zero pad at the begin of H and W
// (the end pad is implied by Ho / Wo).  HBM-bound: a thread owns (channel quad, output column), keeps its k*k*4 weights
// in registers and walks down the image in bands of TH output rows, so every input element is loaded ~(TH*S+K-S)/(TH*S)
// times per column tap instead of K times; lanes run along (column, channel) = contiguous memory.
template <int K, int S, int TH, int MINB>
__global__ __launch_bounds__(256, MINB) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, int flip, int pb, int H, int W,
                                                     int C, int Ho, int Wo, const float* __restrict__ addend, float* __restrict__ y) {
  const int C4 = C >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wo * C4) return;
  const int wo = idx / C4, c = (idx - wo * C4) * 4;
  const int n = blockIdx.z;
  f4 wr[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wr[t] = *(const f4*)(w + (size_t)(flip ? K * K - 1 - t : t) * C + c);
  const float* xn = x + (size_t)n * H * W * C + c;
  const int wi0 = wo * S - pb;
  constexpr int R = (TH - 1) * S + K;
  for (int ho0 = blockIdx.y * TH; ho0 < Ho; ho0 += gridDim.y * TH) {
    f4 acc[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
    const int hi0 = ho0 * S - pb;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int hi = hi0 + rr;
      if (hi < 0 || hi >= H) continue;               // uniform over the workgroup
      const float* row = xn + (size_t)hi * W * C;
      f4 xv[K];
#pragma unroll
      for (int s_ = 0; s_ < K; ++s_) {
        const int wi = wi0 + s_;
        xv[s_] = (wi >= 0 && wi < W) ? *(const f4*)(row + (size_t)wi * C) : (f4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int r = rr - j * S;                    // compile-time after unrolling
        if (r >= 0 && r < K) {
#pragma unroll
          for (int s_ = 0; s_ < K; ++s_) acc[j] += xv[s_] * wr[r * K + s_];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int ho = ho0 + j;
      if (ho < Ho) {
        const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
        f4 v = acc[j];
        if (addend) v += *(const f4*)(addend + o);
        *(f4*)(y + o) = v;
      }
    }
  }
}
static inline void dw_grid(int cols_c4, int nbands, int N, dim3* g) {
  const unsigned gx = (unsigned)((cols_c4 + 255) / 256);
  unsigned gy = (unsigned)std::max(1, std::min(nbands, (int)(4096 / std::max(1u, gx * (unsigned)N))));
  *g = dim3(gx, gy, (unsigned)N);
}
template <int K, int S, int TH, int MINB = 1>
static hipError_t dw_fwd_launch(const float* x, const float* w, int flip, int pb, int N, int H, int W, int C, int Ho, int Wo,
                                const float* addend, float* y, hipStream_t st) {
  dim3 g; dw_grid(Wo * (C / 4), (Ho + TH - 1) / TH, N, &g);
  hipLaunchKernelGGL((dw_fwd_kernel<K, S, TH, MINB>), g, dim3(256), 0, st, x, w, flip, pb, H, W, C, Ho, Wo, addend, y);
  return hipGetLastError();
}
// bands of TH = 4 output rows (8 rows: 256 VGPRs / 1 wave per SIMD, slower; 2 rows: the halo is re-loaded too often)
template <int K, int S>
static hipError_t dw_any(const float* x, const float* w, int flip, int pb, int N, int H, int W, int C, int Ho, int Wo, const float* addend,
                         float* y, hipStream_t st) {
  return dw_fwd_launch<K, S, 4, 1>(x, w, flip, pb, N, H, W, C, Ho, Wo, addend, y, st);
}
hipError_t launch_dw_fwd(const float* x, const float* w, int k, int stride, int pb, int N, int H, int W, int C, int Ho, int Wo,
                         float* y, hipStream_t st) {
  if (C & 3) return hipErrorInvalidValue;
  if (k == 3 && stride == 1) return dw_any<3, 1>(x, w, 0, pb, N, H, W, C, Ho, Wo, nullptr, y, st);
  if (k == 5 && stride == 1) return dw_any<5, 1>(x, w, 0, pb, N, H, W, C, Ho, Wo, nullptr, y, st);
  if (k == 3 && stride == 2) return dw_any<3, 2>(x, w, 0, pb, N, H, W, C, Ho, Wo, nullptr, y, st);
  if (k == 5 && stride == 2) return dw_any<5, 2>(x, w, 0, pb, N, H, W, C, Ho, Wo, nullptr, y, st);
  return hipErrorInvalidValue;
}

// stride-2 dgrad: dx[hi][wi] = sum over (r, s) with (hi + pb - r), (wi + pb - s) even of dy[(hi+pb-r)/2][(wi+pb-s)/2] * w[r][s].
// A thread owns (channel quad, input column wi) and bands of TH = 8 dx rows; per band it visits the NQ dy rows that
// touch the band; with hi0 even, r = j + E - 2q for dx row j and dy row q, E = K-1 + (pb & 1) (template PBODD).
template <int K, int PBODD, int TH>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, int pb, int H, int W, int C,
                                                          int Ho, int Wo, const float* __restrict__ addend, float* __restrict__ dx) {
  constexpr int E = K - 1 + PBODD, NQ = (TH - 1 + E) / 2 + 1;
  const int C4 = C >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * C4) return;
  const int wi = idx / C4, c = (idx - wi * C4) * 4;
  const int n = blockIdx.z;
  f4 wr[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) wr[t] = *(const f4*)(w + (size_t)t * C + c);
  const float* gn = dy + (size_t)n * Ho * Wo * C + c;
  for (int hi0 = blockIdx.y * TH; hi0 < H; hi0 += gridDim.y * TH) {
    f4 acc[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) acc[j] = (f4){0.f, 0.f, 0.f, 0.f};
    const int hob = (hi0 + pb - (K - 1)) >> 1;       // arithmetic shift: floor
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ho = hob + q;
      if (ho < 0 || ho >= Ho) continue;              // uniform
      const float* row = gn + (size_t)ho * Wo * C;
#pragma unroll
      for (int s_ = 0; s_ < K; ++s_) {
        const int wn = wi + pb - s_;
        const bool ok = wn >= 0 && !(wn & 1) && (wn >> 1) < Wo;
        const f4 g = ok ? *(const f4*)(row + (size_t)(wn >> 1) * C) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TH; ++j) {
          const int r = j + E - 2 * q;               // compile-time
          if (r >= 0 && r < K) acc[j] += g * wr[r * K + s_];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int hi = hi0 + j;
      if (hi < H) {
        const size_t o = (((size_t)n * H + hi) * W + wi) * C + c;
        f4 v = acc[j];
        if (addend) v += *(const f4*)(addend + o);
        *(f4*)(dx + o) = v;
      }
    }
  }
}
hipError_t launch_dw_dgrad(const float* dy, const float* w, int k, int stride, int pb, int N, int H, int W, int C, int Ho, int Wo,
                           const float* addend, float* dx, hipStream_t st) {
  if (C & 3) return hipErrorInvalidValue;
  if (stride == 1) {                                 // correlation with the flipped filter, pad k-1-pb
    if (k == 3) return dw_any<3, 1>(dy, w, 1, k - 1 - pb, N, Ho, Wo, C, H, W, addend, dx, st);
    if (k == 5) return dw_any<5, 1>(dy, w, 1, k - 1 - pb, N, Ho, Wo, C, H, W, addend, dx, st);
    return hipErrorInvalidValue;
  }
  if (stride != 2 || (k != 3 && k != 5)) return hipErrorInvalidValue;
  dim3 g; dw_grid(W * (C / 4), (H + 7) / 8, N, &g);
#define DW_S2(KK, PO) hipLaunchKernelGGL((dw_dgrad_s2_kernel<KK, PO, 8>), g, dim3(256), 0, st, dy, w, pb, H, W, C, Ho, Wo, addend, dx)
  if (k == 3 && !(pb & 1)) DW_S2(3, 0); else if (k == 3) DW_S2(3, 1); else if (!(pb & 1)) DW_S2(5, 0); else DW_S2(5, 1);
#undef DW_S2
  return hipGetLastError();
}

// dw[tap][c] += sum over output pixels dy[pix][c] * x[pix*S - pb + tap][c], two launches.
// (1) same walk as the forward kernel with the k*k*4 partial sums in registers.  A workgroup is CQ channel quads x
// (256 / CQ) output columns (CQ <= 32 divides C/4: every pixel still gives CQ*16 contiguous bytes) and a share of the row
// bands; its columns are combined in LDS (ds_add_f32 into [k*k][CQ*4] floats) and stored as ONE partial
// part[(column block, band share, image)][tap][c].  (2) a thread per (tap, channel) and share of the partials adds them
// (coalesced along c) and issues one atomic into dw.  (Global float atomics from every workgroup — 6 M per launch on the 960-channel layers — took 4x the time of
// the data pass itself.)
template <int K, int S, int TH>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int pb, int H, int W, int C,
                                                       int Ho, int Wo, int CQ, int ncb, float* __restrict__ part) {
  constexpr int TG = K == 5 ? 5 : 9;                 // taps combined per round: [TG][256] 16-byte slots of LDS
  __shared__ f4 buf[TG * 256];
  const int cpb = 256 / CQ;                          // columns per workgroup
  const int cq = threadIdx.x % CQ, col = threadIdx.x / CQ;
  const int cb = blockIdx.x / ncb, wb = blockIdx.x - cb * ncb;
  const int wo = wb * cpb + col;
  const int c = (cb * CQ + cq) * 4;
  const int n = blockIdx.z;
  f4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
  if (col < cpb && wo < Wo) {
    const float* xn = x + (size_t)n * H * W * C + c;
    const float* gn = dy + (size_t)n * Ho * Wo * C + (size_t)wo * C + c;
    const int wi0 = wo * S - pb;
    constexpr int R = (TH - 1) * S + K;
    for (int ho0 = blockIdx.y * TH; ho0 < Ho; ho0 += gridDim.y * TH) {
      f4 gv[TH];
#pragma unroll
      for (int j = 0; j < TH; ++j) gv[j] = (ho0 + j < Ho) ? *(const f4*)(gn + (size_t)(ho0 + j) * Wo * C) : (f4){0.f, 0.f, 0.f, 0.f};
      const int hi0 = ho0 * S - pb;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int hi = hi0 + rr;
        if (hi < 0 || hi >= H) continue;
        const float* row = xn + (size_t)hi * W * C;
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
          const int wi = wi0 + s_;
          const f4 xv = (wi >= 0 && wi < W) ? *(const f4*)(row + (size_t)wi * C) : (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < TH; ++j) {
            const int r = rr - j * S;
            if (r >= 0 && r < K) acc[r * K + s_] += gv[j] * xv;
          }
        }
      }
    }
  }
  // combine the workgroup's columns: TG taps per round through LDS (plain stores, then CQ*TG threads add cpb slots each;
  // LDS float atomics here cost a fixed ~270 us per 5x5 launch, whatever the tensor size)
  float* dst = part + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * ncb + wb) * (K * K) * C + cb * CQ * 4;
#pragma unroll
  for (int t0 = 0; t0 < K * K; t0 += TG) {
    if (t0) __syncthreads();
#pragma unroll
    for (int i = 0; i < TG; ++i) buf[i * 256 + threadIdx.x] = acc[t0 + i];       // (threads without a column hold zeros)
    __syncthreads();
    for (int idx = threadIdx.x; idx < TG * CQ; idx += 256) {
      const int i = idx / CQ, q = idx - i * CQ;
      f4 sum = buf[i * 256 + q];
      for (int cl = 1; cl < cpb; ++cl) sum += buf[i * 256 + cl * CQ + q];
      *(f4*)(dst + (size_t)(t0 + i) * C + q * 4) = sum;
    }
  }
}
__global__ void dw_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int per = (nparts + gridDim.y - 1) / gridDim.y;          // blockIdx.y: a share of the partials
  const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int p = p0;
  for (; p + 3 < p1; p += 4) {
    a0 += part[(size_t)p * n + i]; a1 += part[(size_t)(p + 1) * n + i]; a2 += part[(size_t)(p + 2) * n + i]; a3 += part[(size_t)(p + 3) * n + i];
  }
  for (; p < p1; ++p) a0 += part[(size_t)p * n + i];
  if (p1 > p0) atomicAdd(dw + i, (a0 + a1) + (a2 + a3));
}
struct DwWgradCfg { int CQ, ncb; unsigned gx, gy; size_t parts; };
template <int TH>
static DwWgradCfg dw_wgrad_cfg(int N, int C, int Ho, int Wo) {
  DwWgradCfg g; const int C4 = C / 4;
  g.CQ = 1;
  for (int d = 1; d <= 32 && d <= C4; ++d) if (C4 % d == 0) g.CQ = d;
  const int cpb = 256 / g.CQ; g.ncb = (Wo + cpb - 1) / cpb;
  g.gx = (unsigned)((C4 / g.CQ) * g.ncb);
  const int nbands = (Ho + TH - 1) / TH;
  g.gy = (unsigned)std::max(1, std::min(nbands, (int)(2048 / std::max(1u, g.gx * (unsigned)N))));
  g.parts = (size_t)g.ncb * g.gy * N;
  return g;
}
// scratch floats the two-stage depthwise weight gradient needs for this layer
size_t dw_wgrad_scratch_floats(int k, int N, int C, int Ho, int Wo) { return dw_wgrad_cfg<4>(N, C, Ho, Wo).parts * (size_t)k * k * C; }
template <int K, int S, int TH>
static hipError_t dw_wgrad_launch(const float* x, const float* dy, int pb, int N, int H, int W, int C, int Ho, int Wo, float* dw,
                                  float* scratch, hipStream_t st) {
  const DwWgradCfg g = dw_wgrad_cfg<TH>(N, C, Ho, Wo);
  hipLaunchKernelGGL((dw_wgrad_kernel<K, S, TH>), dim3(g.gx, g.gy, (unsigned)N), dim3(256), 0, st, x, dy, pb, H, W, C, Ho, Wo, g.CQ, g.ncb, scratch);
  const int n = K * K * C;
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((n + 255) / 256, (unsigned)std::max<size_t>(1, std::min<size_t>(16, g.parts / 8))), dim3(256), 0, st, scratch, (int)g.parts, n, dw);
  return hipGetLastError();
}
hipError_t launch_dw_wgrad(const float* x, const float* dy, int k, int stride, int pb, int N, int H, int W, int C, int Ho, int Wo,
                           float* dw, float* scratch, hipStream_t st) {
  if ((C & 3) || !scratch) return hipErrorInvalidValue;
  if (k == 3 && stride == 1) return dw_wgrad_launch<3, 1, 4>(x, dy, pb, N, H, W, C, Ho, Wo, dw, scratch, st);
  if (k == 3 && stride == 2) return dw_wgrad_launch<3, 2, 4>(x, dy, pb, N, H, W, C, Ho, Wo, dw, scratch, st);
  if (k == 5 && stride == 1) return dw_wgrad_launch<5, 1, 4>(x, dy, pb, N, H, W, C, Ho, Wo, dw, scratch, st);
  if (k == 5 && stride == 2) return dw_wgrad_launch<5, 2, 4>(x, dy, pb, N, H, W, C, Ho, Wo, dw, scratch, st);
  return hipErrorInvalidValue;
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

// ---------------------------------------------------------------- BatchNorm backward behind a swish (and the SE product)
// The gradient that reaches a BatchNorm output through swish [and the squeeze-and-excitation product] is
//   g' = (SE ? g * s[n][c] + gpool[n][c] / hw : g) * swish'(y*scale + shift)
// Both BatchNorm-backward passes read g and y anyway, so g' is formed on the fly in each instead of in a pass of its own
// (saves one read and one write of g and one read of y per MBConv stage).  Same reduction / apply structure as
// bn_bwd_reduce_kernel / bn_bwd_apply_kernel (elementwise.hip).
template <bool SE>
__device__ __forceinline__ f4 act_grad(f4 gv, f4 yv, f4 sc, f4 sf, const float* se_s, const float* gpool, float inv_hw, size_t nc) {
  if (SE) gv = gv * *(const f4*)(se_s + nc) + *(const f4*)(gpool + nc) * inv_hw;
  return gv * dswish4(yv * sc + sf);
}
template <bool SE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_act_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ se_s,
                                                                const float* __restrict__ gpool, float inv_hw, unsigned hw, double* dgamma,
                                                                double* dbeta, size_t npix, int C, int CW) {
  __shared__ float red[256 * 8];
  const int c0 = blockIdx.y * CW;
  const int tc = CW / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 sg = {0, 0, 0, 0}, sgy = {0, 0, 0, 0};
  if (rx < tr) {
    const f4 mu = *(const f4*)(mean + c), rs = *(const f4*)(rstd + c), sc = *(const f4*)(scale + c), sf = *(const f4*)(shift + c);
    const size_t ps = (size_t)gridDim.x * tr;
    size_t p = (size_t)blockIdx.x * tr + rx;
    f4 s2 = {0, 0, 0, 0};
    for (; p + ps < npix; p += 2 * ps) {
      const f4 g0 = *(const f4*)(g + p * C + c), g1 = *(const f4*)(g + (p + ps) * C + c);
      const f4 y0 = *(const f4*)(y + p * C + c), y1 = *(const f4*)(y + (p + ps) * C + c);
      const f4 a0 = act_grad<SE>(g0, y0, sc, sf, se_s, gpool, inv_hw, SE ? (size_t)((unsigned)p / hw) * C + c : 0);
      const f4 a1 = act_grad<SE>(g1, y1, sc, sf, se_s, gpool, inv_hw, SE ? (size_t)((unsigned)(p + ps) / hw) * C + c : 0);
      sg += a0 + a1;
      s2 += a0 * (y0 - mu) + a1 * (y1 - mu);
    }
    for (; p < npix; p += ps) {
      const f4 yv = *(const f4*)(y + p * C + c);
      const f4 a0 = act_grad<SE>(*(const f4*)(g + p * C + c), yv, sc, sf, se_s, gpool, inv_hw, SE ? (size_t)((unsigned)p / hw) * C + c : 0);
      sg += a0; s2 += a0 * (yv - mu);
    }
    sgy = s2 * rs;
  }
  float* r = red + threadIdx.x * 8;
  r[0] = sg.x; r[1] = sg.y; r[2] = sg.z; r[3] = sg.w; r[4] = sgy.x; r[5] = sgy.y; r[6] = sgy.z; r[7] = sgy.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 8; t += 256) {
    const int q = t / 8, e = t % 8;
    double acc = 0.0;
    for (int k = 0; k < tr; ++k) acc += (double)red[(k * tc + q) * 8 + e];
    if (e < 4) atomicAdd(dbeta + c0 + q * 4 + e, acc); else atomicAdd(dgamma + c0 + q * 4 + (e - 4), acc);
  }
}
template <bool SE>
__global__ void bn_bwd_apply_act_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ scale,
                                        const float* __restrict__ shift, const float* __restrict__ se_s, const float* __restrict__ gpool,
                                        float inv_hw, size_t per_img4, const double* __restrict__ dgamma, const double* __restrict__ dbeta,
                                        float* __restrict__ dy, float* gamma_grad, float* beta_grad, size_t n4, int C, float invM) {
  if (blockIdx.x == 0 && gamma_grad)
    for (int c = threadIdx.x; c < C; c += blockDim.x) { gamma_grad[c] = (float)dgamma[c]; beta_grad[c] = (float)dbeta[c]; }
  const size_t stride = (size_t)gridDim.x * blockDim.x;       // a multiple of C/4: a thread stays on one channel quad
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % (size_t)C);
  const f4 mu = *(const f4*)(mean + c), rs = *(const f4*)(rstd + c), gm = *(const f4*)(gamma + c), sc = *(const f4*)(scale + c), sf = *(const f4*)(shift + c);
  f4 dg, db;
  dg.x = (float)dgamma[c] * invM; dg.y = (float)dgamma[c + 1] * invM; dg.z = (float)dgamma[c + 2] * invM; dg.w = (float)dgamma[c + 3] * invM;
  db.x = (float)dbeta[c] * invM; db.y = (float)dbeta[c + 1] * invM; db.z = (float)dbeta[c + 2] * invM; db.w = (float)dbeta[c + 3] * invM;
  const f4 A = gm * rs, B = -(gm * rs * rs * dg), K = -(A * db) - B * mu;
  for (; i + stride < n4; i += 2 * stride) {
    const f4 g0 = *(const f4*)(g + i * 4), g1 = *(const f4*)(g + (i + stride) * 4);
    const f4 y0 = *(const f4*)(y + i * 4), y1 = *(const f4*)(y + (i + stride) * 4);
    *(f4*)(dy + i * 4) = A * act_grad<SE>(g0, y0, sc, sf, se_s, gpool, inv_hw, SE ? (i / per_img4) * C + c : 0) + B * y0 + K;
    *(f4*)(dy + (i + stride) * 4) = A * act_grad<SE>(g1, y1, sc, sf, se_s, gpool, inv_hw, SE ? ((i + stride) / per_img4) * C + c : 0) + B * y1 + K;
  }
  for (; i < n4; i += stride) {
    const f4 yv = *(const f4*)(y + i * 4);
    *(f4*)(dy + i * 4) = A * act_grad<SE>(*(const f4*)(g + i * 4), yv, sc, sf, se_s, gpool, inv_hw, SE ? (i / per_img4) * C + c : 0) + B * yv + K;
  }
}
// BatchNorm backward of y -> swish (-> SE product): g = gradient wrt the swish output (SE: wrt the SE-scaled tensor)
hipError_t launch_bn_bwd_act(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma, const float* scale,
                             const float* shift, const float* se_s, const float* gpool, int N, size_t hw, double* dgamma, double* dbeta,
                             float* dy, float* gamma_grad, float* beta_grad, int C, hipStream_t st) {
  const size_t npix = (size_t)N * hw;
  const int CW = pick_cw(C);
  if (!CW || npix >= (1ull << 32)) return hipErrorInvalidValue;
  const int tr = 256 / (CW / 4);
  const float inv_hw = (float)(1.0 / (double)hw);
  const dim3 gr(nb(npix, tr * 8), C / CW);
  if (se_s) hipLaunchKernelGGL((bn_bwd_reduce_act_kernel<true>), gr, dim3(256), 0, st, g, y, mean, rstd, scale, shift, se_s, gpool, inv_hw, (unsigned)hw, dgamma, dbeta, npix, C, CW);
  else hipLaunchKernelGGL((bn_bwd_reduce_act_kernel<false>), gr, dim3(256), 0, st, g, y, mean, rstd, scale, shift, se_s, gpool, inv_hw, (unsigned)hw, dgamma, dbeta, npix, C, CW);
  const size_t n4 = npix * C / 4;
  unsigned nbk = nb(n4, 256 * 4);
  const unsigned q = (unsigned)(C / 4);
  unsigned unit;                                   // smallest block count with (blocks*256) % q == 0
  { unsigned a_ = q, b_ = 256; while (b_) { unsigned t = a_ % b_; a_ = b_; b_ = t; } unit = q / a_; }
  nbk = ((nbk + unit - 1) / unit) * unit;
  const size_t per_img4 = hw * (size_t)C / 4;
  const float invM = (float)(1.0 / (double)npix);
  if (se_s) hipLaunchKernelGGL((bn_bwd_apply_act_kernel<true>), dim3(nbk), dim3(256), 0, st, g, y, mean, rstd, gamma, scale, shift, se_s, gpool, inv_hw, per_img4, dgamma, dbeta, dy, gamma_grad, beta_grad, n4, C, invM);
  else hipLaunchKernelGGL((bn_bwd_apply_act_kernel<false>), dim3(nbk), dim3(256), 0, st, g, y, mean, rstd, gamma, scale, shift, se_s, gpool, inv_hw, per_img4, dgamma, dbeta, dy, gamma_grad, beta_grad, n4, C, invM);
  return hipGetLastError();
}

// ---------------------------------------------------------------- squeeze-and-excitation
// out[n][c] = scale * sum over hw of v,  v = a * b (b given: the SE backward's sum g*a1), v = a (plain pooling), or
// v = swish(a*sc + sh) which is ALSO written to act_out (the forward: the activation pass and the squeeze pooling are one
// pass over the tensor).  Two stages, no atomics: workgroup (bx, n, cz) stores ONE partial per channel into
// part[bx][n][c]; se_reduce_finish adds the partials of a (n, c) in block order.  The split of hw over workgroups depends
// on hw and C only — never on N — so the result is bit-identical from run to run AND across batch sizes (the eval forward
// of an image does not depend on its batch mates: BASELINE config 5's "equal to the batch-1 path" property).
constexpr int kSeMaxParts = 64;
__global__ __launch_bounds__(256) void se_reduce_hw_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, float* __restrict__ act_out, size_t hw, int C, int CW,
                                                           float* __restrict__ part) {
  __shared__ float red[256 * 4];
  const int n = blockIdx.y;
  const int c0 = blockIdx.z * CW;
  const int tc = CW / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 acc = {0, 0, 0, 0};
  if (rx < tr) {
    f4 scv = {1, 1, 1, 1}, shv = {0, 0, 0, 0};
    if (sc) { scv = *(const f4*)(sc + c); shv = *(const f4*)(sh + c); }
    const size_t ps = (size_t)gridDim.x * tr;
    size_t p = (size_t)blockIdx.x * tr + rx;
    for (; p + ps < hw; p += 2 * ps) {
      const size_t o0 = ((size_t)n * hw + p) * C + c, o1 = ((size_t)n * hw + p + ps) * C + c;
      f4 v0 = *(const f4*)(a + o0), v1 = *(const f4*)(a + o1);
      if (b) { v0 = v0 * *(const f4*)(b + o0); v1 = v1 * *(const f4*)(b + o1); }
      if (sc) { v0 = swish4(v0 * scv + shv); v1 = swish4(v1 * scv + shv); *(f4*)(act_out + o0) = v0; *(f4*)(act_out + o1) = v1; }
      acc += v0 + v1;
    }
    for (; p < hw; p += ps) {
      const size_t o = ((size_t)n * hw + p) * C + c;
      f4 v = *(const f4*)(a + o);
      if (b) v = v * *(const f4*)(b + o);
      if (sc) { v = swish4(v * scv + shv); *(f4*)(act_out + o) = v; }
      acc += v;
    }
  }
  float* r = red + threadIdx.x * 4;
  r[0] = acc.x; r[1] = acc.y; r[2] = acc.z; r[3] = acc.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 4; t += 256) {
    const int q = t / 4, e = t % 4;
    float sum = 0.f;
    for (int k = 0; k < tr; ++k) sum += red[(k * tc + q) * 4 + e];
    part[((size_t)blockIdx.x * gridDim.y + n) * C + c0 + q * 4 + e] = sum;
  }
}
// 64 of these per EfficientNet-b4 step, each with almost nothing to do: what matters is the length of the dependent chain.  Workgroup =
// 32 (n, c) entries x 8 partial groups: thread (entry, group) adds partials group, group + 8, ... in order, the 8 group sums are
// combined in group order through LDS (fixed association: bit-reproducible and independent of the batch size).  One thread per
// entry walking up to 64 partials measured 16 us per launch.
__global__ __launch_bounds__(256) void se_reduce_finish_kernel(const float* __restrict__ part, int nparts, size_t nc, float scale, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int u = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + u;
  float s = 0.f;
  if (i < nc)
    for (int k = grp; k < nparts; k += 8) s += part[(size_t)k * nc + i];
  red[grp][u] = s;
  __syncthreads();
  if (grp == 0 && i < nc) {
    float t = red[0][u];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][u];
    out[i] = t * scale;
  }
}
static int se_parts(size_t hw, int C) {
  const int CW = pick_cw(C);
  if (!CW) return 0;
  const int tr = 256 / (CW / 4);
  size_t bx = (hw + (size_t)tr * 16 - 1) / ((size_t)tr * 16);
  if (bx > (size_t)kSeMaxParts) bx = kSeMaxParts;
  return bx < 1 ? 1 : (int)bx;
}
size_t se_reduce_scratch_floats(int N, int C) { return (size_t)kSeMaxParts * N * C; }
static hipError_t se_reduce_launch(const float* a, const float* b, const float* sc, const float* sh, float* act_out, int N, size_t hw, int C,
                                   float scale, float* out, float* part, hipStream_t st) {
  const int CW = pick_cw(C);
  if (!CW || !part) return hipErrorInvalidValue;
  const int bx = se_parts(hw, C);
  hipLaunchKernelGGL(se_reduce_hw_kernel, dim3(bx, N, C / CW), dim3(256), 0, st, a, b, sc, sh, act_out, hw, C, CW, part);
  const size_t nc = (size_t)N * C;
  hipLaunchKernelGGL(se_reduce_finish_kernel, dim3((unsigned)((nc + 31) / 32)), dim3(256), 0, st, part, bx, nc, scale, out);
  return hipGetLastError();
}
hipError_t launch_se_reduce_hw(const float* a, const float* b, int N, size_t hw, int C, float scale, float* out, float* part, hipStream_t st) {
  return se_reduce_launch(a, b, nullptr, nullptr, nullptr, N, hw, C, scale, out, part, st);
}
// act_out = swish(y*sc + sh) and pool[n][c] = mean over hw of it, in one pass (+ the tiny finish launch)
hipError_t launch_swish_pool(const float* y, const float* sc, const float* sh, float* act_out, int N, size_t hw, int C, float* pool,
                             float* part, hipStream_t st) {
  return se_reduce_launch(y, nullptr, sc, sh, act_out, N, hw, C, (float)(1.0 / (double)hw), pool, part, st);
}
// SE FCs, forward, two small launches.  (1) grid (N, ceil(nsq/4)): a wave per hidden unit, lanes along the C inputs
// (coalesced row of W1 [nsq][K1pad]), shuffle reduce: hpre = W1 pool + b1 (kept for the backward), hid = swish(hpre).
// (2) grid (N, ceil(C/256)): a thread per output channel (row of W2 [C][K2pad], 16-byte loads): s = sigmoid(W2 hid + b2).
__global__ __launch_bounds__(256) void se_fc1_kernel(const float* __restrict__ pool, const float* __restrict__ w1, const float* __restrict__ b1,
                                                     int K1pad, int C, int nsq, float* __restrict__ hpre, float* __restrict__ hid) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (j >= nsq) return;
  const float* p = pool + (size_t)n * C;
  const float* row = w1 + (size_t)j * K1pad;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc += row[c] * p[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) { acc += b1[j]; hpre[(size_t)n * nsq + j] = acc; hid[(size_t)n * nsq + j] = acc * sigm(acc); }
}
__global__ __launch_bounds__(256) void se_fc2_kernel(const float* __restrict__ hid, const float* __restrict__ w2, const float* __restrict__ b2,
                                                     int K2pad, int C, int nsq, float* __restrict__ s) {
  extern __shared__ float sh[];        // [rup(nsq,4)] hidden, zero padded
  const int n = blockIdx.x;
  const int nsqP = (nsq + 3) & ~3;
  for (int j = threadIdx.x; j < nsqP; j += 256) sh[j] = j < nsq ? hid[(size_t)n * nsq + j] : 0.f;
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < C) {
    float acc = b2[c];
    const float* row = w2 + (size_t)c * K2pad;
    for (int j = 0; j < nsqP; j += 4) { const f4 wv4 = *(const f4*)(row + j); acc += wv4.x * sh[j] + wv4.y * sh[j + 1] + wv4.z * sh[j + 2] + wv4.w * sh[j + 3]; }
    s[(size_t)n * C + c] = sigm(acc);
  }
}
// hid: scratch [N][nsq]
hipError_t launch_se_fc_fwd(const float* pool, const float* w1, const float* b1, int K1pad, const float* w2, const float* b2,
                            int K2pad, int N, int C, int nsq, float* hpre, float* hid, float* s, hipStream_t st) {
  hipLaunchKernelGGL(se_fc1_kernel, dim3(N, (nsq + 3) / 4), dim3(256), 0, st, pool, w1, b1, K1pad, C, nsq, hpre, hid);
  hipLaunchKernelGGL(se_fc2_kernel, dim3(N, (C + 255) / 256), dim3(256), ((nsq + 3) & ~3) * sizeof(float), st, hid, w2, b2, K2pad, C, nsq, s);
  return hipGetLastError();
}
// SE FCs, backward, two launches.  (A) grid (N, ceil(C/256)), a thread per channel: gz2 = gs * s(1-s) (in place over
// gs); its row of W2 times gz2 is added into nsq LDS accumulators (each lane starts at a different unit, so the
// ds_add_f32 of a wave hit different addresses), then one global atomic per unit and workgroup into acc1 [N][nsq]
// (zeroed by the caller) = W2^T gz2.  (B) over all samples, with gz1 = acc1 * swish'(hpre) and hid = swish(hpre) formed
// on the fly: gW2 = sum_n gz2 hid^T, gW1 = sum_n gz1 pool^T, gb2, gb1 (plain stores: the gradient arena is zeroed per
// backward and nothing else writes these tensors) and gpool = W1^T gz1.
__global__ __launch_bounds__(256) void se_fc_bwd_a_kernel(float* __restrict__ gs, const float* __restrict__ s, const float* __restrict__ w2, int K2pad,
                                                          int C, int nsq, float* __restrict__ acc1) {
  extern __shared__ float sh[];        // [nsq] accumulators
  const int n = blockIdx.x;
  for (int j = threadIdx.x; j < nsq; j += 256) sh[j] = 0.f;
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < C) {
    const float sv = s[(size_t)n * C + c];
    const float g = gs[(size_t)n * C + c] * sv * (1.f - sv);
    gs[(size_t)n * C + c] = g;
    const float* row = w2 + (size_t)c * K2pad;
    int j = threadIdx.x % nsq;
    for (int jj = 0; jj < nsq; ++jj) { atomicAdd(&sh[j], row[j] * g); if (++j == nsq) j = 0; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nsq; j += 256) atomicAdd(acc1 + (size_t)n * nsq + j, sh[j]);
}
__global__ __launch_bounds__(256) void se_fc_bwd_b_kernel(const float* __restrict__ gz2, const float* __restrict__ hpre, const float* __restrict__ acc1,
                                                          const float* __restrict__ pool, const float* __restrict__ w1, int K1pad, int K2pad,
                                                          int N, int C, int nsq, float* __restrict__ gpool, float* __restrict__ gw1,
                                                          float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < C * nsq) {
    { const int c = i / nsq, j = i - c * nsq; float a = 0.f;
      for (int n = 0; n < N; ++n) { const float z = hpre[(size_t)n * nsq + j]; a += gz2[(size_t)n * C + c] * (z * sigm(z)); }
      gw2[(size_t)c * K2pad + j] = a; }
    { const int j = i / C, c = i - j * C; float a = 0.f;
      for (int n = 0; n < N; ++n) a += acc1[(size_t)n * nsq + j] * dswish(hpre[(size_t)n * nsq + j]) * pool[(size_t)n * C + c];
      gw1[(size_t)j * K1pad + c] = a; }
  }
  if (i < C) { float a = 0.f; for (int n = 0; n < N; ++n) a += gz2[(size_t)n * C + i]; gb2[i] = a; }
  if (i < nsq) { float a = 0.f; for (int n = 0; n < N; ++n) a += acc1[(size_t)n * nsq + i] * dswish(hpre[(size_t)n * nsq + i]); gb1[i] = a; }
  if (i < N * C) {
    const int n = i / C, c = i - n * C;
    float a = 0.f;
    for (int j = 0; j < nsq; ++j) a += w1[(size_t)j * K1pad + c] * (acc1[(size_t)n * nsq + j] * dswish(hpre[(size_t)n * nsq + j]));
    gpool[i] = a;
  }
}
// gs [N][C] is overwritten with gz2; acc1: scratch [N][nsq], zeroed by the caller
hipError_t launch_se_fc_bwd(float* gs, const float* s, const float* hpre, const float* pool, const float* w1, int K1pad,
                            const float* w2, int K2pad, int N, int C, int nsq, float* gpool, float* acc1, float* gw1,
                            float* gb1, float* gw2, float* gb2, hipStream_t st) {
  hipLaunchKernelGGL(se_fc_bwd_a_kernel, dim3(N, (C + 255) / 256), dim3(256), nsq * sizeof(float), st, gs, s, w2, K2pad, C, nsq, acc1);
  const int work = std::max(C * nsq, N * C);
  hipLaunchKernelGGL(se_fc_bwd_b_kernel, dim3((work + 255) / 256), dim3(256), 0, st, gs, hpre, acc1, pool, w1, K1pad, K2pad, N, C, nsq,
                     gpool, gw1, gb1, gw2, gb2);
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
}  // namespace uwm
