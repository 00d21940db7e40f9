// HBM-bound helper kernels of the U-Net path (gfx950): layout change, BatchNorm finalize /
// backward, residual add, max-pool, upsample-concat gradient split, weight repack, Adam.
// All are 16-byte vectorised over the NHWC channel dimension (C % 4 == 0 everywhere).
// Reference semantics replaced: torch BatchNorm2d / ReLU / MaxPool2d / interpolate+cat autograd
// and optim.Adam as driven by /root/reference/src/train.py:96-105 (SURVEY.md §8 a4,a9,a10,a14,a15).
#include "uwm_kernels.h"
#include <vector>
#include <mutex>
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
static constexpr int kMaxBlocks = 256 * 8;

static inline unsigned nblocks(size_t work, int per_block) {
  size_t b = (work + per_block - 1) / per_block;
  if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ------------------------------------------------------------------ NCHW -> NHWC(pad 4)
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, size_t HW,
                                     size_t total, int CP) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const float* xp = x + n * C * HW + p;
    float* yp = y + i * CP;
    for (int c0 = 0; c0 < CP; c0 += 4) {
      f4 v;
      v.x = c0 + 0 < C ? xp[(size_t)(c0 + 0) * HW] : 0.f;
      v.y = c0 + 1 < C ? xp[(size_t)(c0 + 1) * HW] : 0.f;
      v.z = c0 + 2 < C ? xp[(size_t)(c0 + 2) * HW] : 0.f;
      v.w = c0 + 3 < C ? xp[(size_t)(c0 + 3) * HW] : 0.f;
      *(f4*)(yp + c0) = v;
    }
  }
}
hipError_t launch_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t st) {
  const size_t HW = (size_t)H * W, total = (size_t)N * HW;
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, x, y, C, HW, total, CP);
  return hipGetLastError();
}

// ------------------------------------------------------------------ BatchNorm finalize
// 46 launches per resnet34 step sit on the main stream's critical path and each has almost nothing to do: what matters is
// the length of the dependent-load chain.  A workgroup = 16 channels x 16 replica lanes: lane r adds replicas r, r+16, ...
// (independent loads, all in flight at once), the 16 partials are combined by a fixed xor tree (bit-reproducible), lane 0
// of each channel finishes.  (One thread per channel walking 32 replicas measured 7.3 us per launch.)
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ssum, const double* __restrict__ ssq, const float* gamma, const float* beta,
                                   float* run_mean, float* run_var, float* mean, float* rstd, float* scale,
                                   float* shift, int C, double count, float eps, float momentum, int upd, int nrep,
                                   int rep_stride) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), r0 = threadIdx.x & 15;
  const bool live = c < C;
  double s1 = 0.0, s2 = 0.0;
  if (live)
    for (int r = r0; r < nrep; r += 16) { s1 += ssum[(size_t)r * rep_stride + c]; s2 += ssq[(size_t)r * rep_stride + c]; }
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) { s1 += shfl_xor_f64(s1, d); s2 += shfl_xor_f64(s2, d); }
  if (!live || r0) return;
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const float g = gamma[c], b = beta[c];
  mean[c] = (float)mu; rstd[c] = (float)rs;
  const float sc = (float)((double)g * rs);
  scale[c] = sc; shift[c] = (float)((double)b - mu * (double)g * rs);
  if (upd) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mu);
    run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unb);
  }
}
hipError_t launch_bn_finalize(const double* ssum, const double* ssq, const float* gamma, const float* beta,
                              float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift,
                              int C, double count, float eps, float momentum, int update_running, hipStream_t st,
                              int nrep, int rep_stride) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, ssum, ssq, gamma, beta, run_mean,
                     run_var, mean, rstd, scale, shift, C, count, eps, momentum, update_running, nrep < 1 ? 1 : nrep, rep_stride);
  return hipGetLastError();
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float* scale,
                               float* shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double rs = 1.0 / sqrt((double)rv[c] + (double)eps);
  scale[c] = (float)((double)gamma[c] * rs);
  shift[c] = (float)((double)beta[c] - (double)rm[c] * (double)gamma[c] * rs);
}
hipError_t launch_bn_eval(const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                          float* scale, float* shift, int C, float eps, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, st, gamma, beta, run_mean, run_var, scale,
                     shift, C, eps);
  return hipGetLastError();
}

// ------------------------------------------------------------------ residual add + ReLU
__global__ void residual_kernel(const float* __restrict__ y, const float* __restrict__ s2, const float* __restrict__ b2,
                                const float* __restrict__ id, const float* __restrict__ sd, const float* __restrict__ bd,
                                float* __restrict__ out, size_t n4, int C) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % (size_t)C);
    f4 v = *(const f4*)(y + i * 4) * *(const f4*)(s2 + c) + *(const f4*)(b2 + c);
    f4 r = *(const f4*)(id + i * 4);
    if (sd) r = r * *(const f4*)(sd + c) + *(const f4*)(bd + c);
    v += r;
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    *(f4*)(out + i * 4) = v;
  }
}
hipError_t launch_residual(const float* y, const float* s2, const float* b2, const float* id, const float* sd,
                           const float* bd, float* out, size_t npix, int C, hipStream_t st) {
  const size_t n4 = npix * C / 4;
  hipLaunchKernelGGL(residual_kernel, dim3(nblocks(n4, 256)), dim3(256), 0, st, y, s2, b2, id, sd, bd, out, n4, C);
  return hipGetLastError();
}

// ------------------------------------------------------------------ MaxPool 3x3 s2 p1
__device__ __forceinline__ f4 lazy_val(f4 v, bool has, f4 sc, f4 sh, int relu) {
  if (has) {
    v = v * sc + sh;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  }
  return v;
}

__global__ void maxpool_fwd_kernel(const Src in, float* __restrict__ out, uint8_t* __restrict__ idx, int Ho, int Wo,
                                   size_t total) {
  const int C4 = in.C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t p = i / C4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
    const bool has = in.scale != nullptr;
    f4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (has) { sc = *(const f4*)(in.scale + c); sh = *(const f4*)(in.shift + c); }
    const float ninf = -__builtin_huge_valf();
    f4 best = {ninf, ninf, ninf, ninf};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = ho * 2 - 1 + r;
      if (h < 0 || h >= in.H) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = wo * 2 - 1 + s;
        if (w < 0 || w >= in.W) continue;
        f4 v = lazy_val(*(const f4*)(in.ptr + ((size_t)((size_t)n * in.H + h) * in.W + w) * in.C + c), has, sc, sh, in.relu);
        const int t = r * 3 + s;
        if (v.x > best.x) { best.x = v.x; bi[0] = t; }
        if (v.y > best.y) { best.y = v.y; bi[1] = t; }
        if (v.z > best.z) { best.z = v.z; bi[2] = t; }
        if (v.w > best.w) { best.w = v.w; bi[3] = t; }
      }
    }
    *(f4*)(out + i * 4) = best;
    if (idx) *(uchar4*)(idx + i * 4) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
  }
}
hipError_t launch_maxpool_fwd(const Src& in, float* out, uint8_t* idx, int N, int Ho, int Wo, hipStream_t st) {
  const size_t total = (size_t)N * Ho * Wo * (in.C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, in, out, idx, Ho, Wo, total);
  return hipGetLastError();
}

// bn_mean != nullptr: the masked gradient this kernel writes is the gradient wrt the output of the BatchNorm whose raw input
// it has just read for the ReLU mask (the stem: conv -> bn -> relu -> maxpool), so the BatchNorm-backward sums (dbeta = sum g,
// dgamma = sum g * yhat) are accumulated here, one of `srep` fp64 replicas per workgroup (added up in bn_bwd_apply_kernel's prologue):
// bn_bwd_reduce's pass over both tensors disappears.  Needs gridDim.x * 256 to be a multiple of C/4 (a thread keeps its channels).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gout, const uint8_t* __restrict__ idx,
                                   const float* __restrict__ addend, const Src in, float* __restrict__ gin, int Ho,
                                   int Wo, size_t total, const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd,
                                   double* ssum, double* ssq, int srep, int sstride) {
  const int C4 = in.C / 4;
  f4 sg = {0.f, 0.f, 0.f, 0.f}, sgy = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t p = i / C4;
    const int w = (int)(p % in.W); p /= in.W;
    const int h = (int)(p % in.H); const int n = (int)(p / in.H);
    f4 g = addend ? *(const f4*)(addend + i * 4) : (f4){0.f, 0.f, 0.f, 0.f};
    const int ho0 = h >> 1, ho1 = (h + 1) >> 1, wo0 = w >> 1, wo1 = (w + 1) >> 1;
    for (int a = 0; a < 2; ++a) {
      const int ho = a ? ho1 : ho0;
      if ((a && ho1 == ho0) || ho >= Ho) continue;
      const int r = h - (2 * ho - 1);
      for (int b = 0; b < 2; ++b) {
        const int wo = b ? wo1 : wo0;
        if ((b && wo1 == wo0) || wo >= Wo) continue;
        const int s = w - (2 * wo - 1);
        const size_t o = ((size_t)((size_t)n * Ho + ho) * Wo + wo) * in.C + c;
        const uchar4 id = *(const uchar4*)(idx + o);
        const f4 go = *(const f4*)(gout + o);
        const int t = r * 3 + s;
        if (id.x == t) g.x += go.x;
        if (id.y == t) g.y += go.y;
        if (id.z == t) g.z += go.z;
        if (id.w == t) g.w += go.w;
      }
    }
    if (in.scale && in.relu) {
      const f4 yr = *(const f4*)(in.ptr + i * 4);
      const f4 z = yr * *(const f4*)(in.scale + c) + *(const f4*)(in.shift + c);
      g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f; g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
      if (bn_mean) { sg += g; sgy += g * (yr - *(const f4*)(bn_mean + c)); }
    }
    *(f4*)(gin + i * 4) = g;
  }
  if (bn_mean) {                                 // threads t, t + C4, ... of the workgroup share a channel quad (256 % C4 == 0)
    __shared__ float red[256 * 8];
    const int c = (int)(threadIdx.x % C4) * 4;
    sgy = sgy * *(const f4*)(bn_rstd + c);
    float* r = red + threadIdx.x * 8;
    r[0] = sg.x; r[1] = sg.y; r[2] = sg.z; r[3] = sg.w; r[4] = sgy.x; r[5] = sgy.y; r[6] = sgy.z; r[7] = sgy.w;
    __syncthreads();
    const size_t srep_off = srep > 1 ? (size_t)(blockIdx.x & (unsigned)(srep - 1)) * sstride : 0;
    const int tr = 256 / C4;
    for (int t = threadIdx.x; t < C4 * 8; t += 256) {
      const int q = t / 8, e = t % 8;
      double acc = 0.0;
      for (int k = 0; k < tr; ++k) acc += (double)red[(k * C4 + q) * 8 + e];
      if (e < 4) atomicAdd(ssum + srep_off + q * 4 + e, acc); else atomicAdd(ssq + srep_off + q * 4 + (e - 4), acc);
    }
  }
}
hipError_t launch_maxpool_bwd(const float* gout, const uint8_t* idx, const float* addend, const Src& in, float* gin,
                              int N, int Ho, int Wo, hipStream_t st, const float* bn_mean, const float* bn_rstd, double* ssum,
                              double* ssq, int srep, int sstride) {
  const size_t total = (size_t)N * in.H * in.W * (in.C / 4);
  if (bn_mean && (!bn_rstd || !ssum || !ssq || !in.scale || !in.relu || (in.C & 3) || 256 % (in.C / 4))) return hipErrorInvalidValue;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, gout, idx, addend, in, gin, Ho,
                     Wo, total, bn_mean, bn_rstd, ssum, ssq, srep, sstride);
  return hipGetLastError();
}

// ------------------------------------------------------------------ BatchNorm backward
// per-channel  dbeta = sum g,  dgamma = sum g * (y-mean)*rstd   (fp32 per thread, fp64 across)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* dgamma, double* dbeta, size_t npix, int C, int CW) {
  __shared__ float red[256 * 8];               // CW = channels per blockIdx.y slice (wide layers: resnet50's 2048)
  const int c0 = blockIdx.y * CW;
  const int tc = CW / 4;                      // threads along channels (<= 256)
  const int tr = 256 / tc;                    // pixel rows per pass
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  const int c = c0 + cx * 4;
  f4 sg = {0, 0, 0, 0}, sgy = {0, 0, 0, 0};
  if (rx < tr) {
    const f4 mu = *(const f4*)(mean + c), rs = *(const f4*)(rstd + c);
    // sum g and sum g*y per channel (four 16-byte loads of each tensor in flight); yhat = (y-mu)*rs is applied to the sums
    const size_t ps = (size_t)gridDim.x * tr;
    size_t p = (size_t)blockIdx.x * tr + rx;
    f4 s2 = {0, 0, 0, 0};
    for (; p + 3 * ps < npix; p += 4 * ps) {
      const f4 g0 = *(const f4*)(g + p * C + c), g1 = *(const f4*)(g + (p + ps) * C + c), g2 = *(const f4*)(g + (p + 2 * ps) * C + c),
               g3 = *(const f4*)(g + (p + 3 * ps) * C + c);
      const f4 y0 = *(const f4*)(y + p * C + c), y1 = *(const f4*)(y + (p + ps) * C + c), y2 = *(const f4*)(y + (p + 2 * ps) * C + c),
               y3 = *(const f4*)(y + (p + 3 * ps) * C + c);
      sg += (g0 + g1) + (g2 + g3);
      s2 += (g0 * (y0 - mu) + g1 * (y1 - mu)) + (g2 * (y2 - mu) + g3 * (y3 - mu));
    }
    for (; p < npix; p += ps) {
      const f4 gv = *(const f4*)(g + p * C + c);
      sg += gv; s2 += gv * (*(const f4*)(y + p * C + c) - mu);
    }
    sgy = s2 * rs;
  }
  float* r = red + threadIdx.x * 8;
  r[0] = sg.x; r[1] = sg.y; r[2] = sg.z; r[3] = sg.w; r[4] = sgy.x; r[5] = sgy.y; r[6] = sgy.z; r[7] = sgy.w;
  __syncthreads();
  for (int t = threadIdx.x; t < tc * 8; t += 256) {   // one item per (channel-quad, component)
    const int q = t / 8, e = t % 8;
    double s = 0.0;
    for (int k = 0; k < tr; ++k) s += (double)red[(k * tc + q) * 8 + e];
    if (e < 4) atomicAdd(dbeta + c0 + q * 4 + e, s); else atomicAdd(dgamma + c0 + q * 4 + (e - 4), s);
  }
}
hipError_t launch_bn_bwd_reduce(const float* g, const float* y, const float* mean, const float* rstd, double* dgamma,
                                double* dbeta, size_t npix, int C, hipStream_t st) {
  int CW = 0;                                  // widest slice <= 1024 channels that divides C (1632 -> 816, 2688 -> 896)
  for (int d = 1; d <= C && !CW; ++d) if (C % d == 0 && C / d <= 1024 && (C / d) % 4 == 0) CW = C / d;
  if (!CW) return hipErrorInvalidValue;     // any channel count that is a multiple of 4: idle tail threads
  const int tr = 256 / (CW / 4);
  unsigned nb = nblocks(npix, tr * 8);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, C / CW), dim3(256), 0, st, g, y, mean, rstd, dgamma, dbeta, npix, C, CW);
  return hipGetLastError();
}

// rep != nullptr: the sums arrive as `nrep` per-workgroup replicas {dbeta part[C], dgamma part[C]} a fused dgrad epilogue filled
// (run_dgrad bn_fuse); the workgroup adds them up itself for the channels it touches, in a fixed order (bit-reproducible, the
// same value in every workgroup) — the separate bn_bwd_fold launch between every dgrad and its apply pass (42 per resnet34
// step, each queued behind the other stream's workgroups: 1.2 ms of critical-path latency) is gone.
__global__ __launch_bounds__(256, 6) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const double* __restrict__ dgamma, const double* __restrict__ dbeta, float* __restrict__ dy,
                                    float* gamma_grad, float* beta_grad, size_t n4, int C, float invM,
                                    const double* __restrict__ rep, int nrep, int rep_stride, float* xmax) {
  __shared__ double part[256 * 4];
  __shared__ float wmax[4];
  float amax = 0.f;                                  // max |dy| of this thread's outputs (xmax != nullptr)
  // dy = gm*rs*(g - db - (y-mu)*rs*dg) = A*g + B*y + K per channel; the grid stride is a multiple of C/4, so a thread
  // keeps ONE channel quad: coefficients computed once, four 16-byte loads of each tensor in flight per iteration
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int Q = C >> 2;
  const int c = (int)((i * 4) % (size_t)C);
  // the first iteration's eight 16-byte loads do not depend on the sums: they are in flight while the prologue runs
  f4 g0, g1, g2, g3, y0, y1, y2, y3;
  bool full = i + 3 * stride < n4;
  if (full) {
    g0 = *(const f4*)(g + i * 4); g1 = *(const f4*)(g + (i + stride) * 4); g2 = *(const f4*)(g + (i + 2 * stride) * 4); g3 = *(const f4*)(g + (i + 3 * stride) * 4);
    y0 = *(const f4*)(y + i * 4); y1 = *(const f4*)(y + (i + stride) * 4); y2 = *(const f4*)(y + (i + 2 * stride) * 4); y3 = *(const f4*)(y + (i + 3 * stride) * 4);
  }
  const f4 mu = *(const f4*)(mean + c), rs = *(const f4*)(rstd + c), gm = *(const f4*)(gamma + c);
  // folded sums: dbeta quad, then dgamma quad (one after the other: 8 instead of 16 fp64 accumulators live at a time — the
  // kernel has to stay under 80 VGPRs, see the loop below)
  f4 dbq, dgq;                                         // sums as floats scaled by 1 / M
  {
    double sq[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (rep) {
        // the workgroup touches quads (q0 + t) % Q, t < 256: nq distinct ones; L replica lanes per quad share the nrep copies
        const int nq = Q < 256 ? Q : 256, L = 256 / nq;
        const int q0 = (int)(((size_t)blockIdx.x * 256) % (size_t)Q);
        const int j = threadIdx.x % nq, l = threadIdx.x / nq;
        if (half) __syncthreads();
        if (l < L) {
          const int cj = ((q0 + j) % Q) * 4 + half * C;
          double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
          for (int r = l; r < nrep; r += L) {
            const double* pr = rep + (size_t)r * rep_stride + cj;
            b0 += pr[0]; b1 += pr[1]; b2 += pr[2]; b3 += pr[3];
          }
          double* o = part + (l * nq + j) * 4;
          o[0] = b0; o[1] = b1; o[2] = b2; o[3] = b3;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) sq[e] = 0.0;
        for (int k = 0; k < L; ++k) {                  // fixed order over the replica lanes
          const double* o = part + (k * nq + j) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) sq[e] += o[e];
        }
      } else {
        const double* src = half ? dgamma : dbeta;
#pragma unroll
        for (int e = 0; e < 4; ++e) sq[e] = src[c + e];
      }
      float* gout = half ? gamma_grad : beta_grad;
      if (gout && i < (size_t)Q) {                     // the first Q threads of the grid hold every channel quad once
#pragma unroll
        for (int e = 0; e < 4; ++e) gout[c + e] = (float)sq[e];
      }
      f4 q4; q4.x = (float)sq[0] * invM; q4.y = (float)sq[1] * invM; q4.z = (float)sq[2] * invM; q4.w = (float)sq[3] * invM;
      if (half) dgq = q4; else dbq = q4;
    }
  }
  if (i >= n4 && !xmax) return;
  const f4 dg = dgq, db = dbq;
  const f4 A = gm * rs, B = -(gm * rs * rs * dg), K = -(A * db) - B * mu;
  auto amax4 = [&](const f4& v) { amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); };
  // (no software prefetch across iterations: the kernel is kept under 80 VGPRs so that its workgroups fit beside the 768-thread
  // weight-gradient workgroups of the other stream — 3 x 144 of a SIMD's 512 registers — instead of waiting for one to retire:
  // 89.6 us per launch in the step against 29.8 alone before)
  while (full) {
    g0 = A * g0 + B * y0 + K; g1 = A * g1 + B * y1 + K; g2 = A * g2 + B * y2 + K; g3 = A * g3 + B * y3 + K;
    if (xmax) { amax4(g0); amax4(g1); amax4(g2); amax4(g3); }
    *(f4*)(dy + i * 4) = g0;
    *(f4*)(dy + (i + stride) * 4) = g1;
    *(f4*)(dy + (i + 2 * stride) * 4) = g2;
    *(f4*)(dy + (i + 3 * stride) * 4) = g3;
    i += 4 * stride;
    full = i + 3 * stride < n4;
    if (full) {
      g0 = *(const f4*)(g + i * 4); g1 = *(const f4*)(g + (i + stride) * 4); g2 = *(const f4*)(g + (i + 2 * stride) * 4); g3 = *(const f4*)(g + (i + 3 * stride) * 4);
      y0 = *(const f4*)(y + i * 4); y1 = *(const f4*)(y + (i + stride) * 4); y2 = *(const f4*)(y + (i + 2 * stride) * 4); y3 = *(const f4*)(y + (i + 3 * stride) * 4);
    }
  }
  for (; i < n4; i += stride) {
    const f4 o = A * *(const f4*)(g + i * 4) + B * *(const f4*)(y + i * 4) + K;
    if (xmax) amax4(o);
    *(f4*)(dy + i * 4) = o;
  }
  if (xmax) {                                        // wave max -> workgroup max -> one atomic on one of 32 slots (non-negative floats order as integers)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (mx > 0.f) atomicMax((unsigned*)xmax + (blockIdx.x & 31), __float_as_uint(mx));
    }
  }
}
hipError_t launch_bn_bwd_apply(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma,
                               const double* dgamma, const double* dbeta, float* dy, float* gamma_grad, float* beta_grad,
                               size_t npix, int C, hipStream_t st, const double* rep, int nrep, int rep_stride, hipEvent_t done, float* xmax) {
  const size_t n4 = npix * C / 4;
  // grid stride (blocks x 256) must be a multiple of C/4 so that every thread stays on one channel quad
  unsigned nb = nblocks(n4, 256 * 4);
  const unsigned q = (unsigned)(C / 4);
  unsigned unit = q;                               // smallest block count with (nb*256) % q == 0: q / gcd(q, 256)
  { unsigned a = q, b = 256; while (b) { unsigned t = a % b; a = b; b = t; } unit = q / a; }
  nb = ((nb + unit - 1) / unit) * unit;
  if ((size_t)nb * 256 < q) nb = ((q + 255) / 256 + unit - 1) / unit * unit;      // (tiny tensors: the grid must still hold every channel quad for the gamma / beta gradients)
  // done: an event attached to THIS dispatch's completion signal (hipExtLaunchKernelGGL) — the weight-gradient stream forks
  // from it without a barrier packet of its own on this stream (a hipEventRecord between this kernel and the next dgrad cost
  // ~8 us of queue-processing latency on the dependent chain, 40 times per step)
  if (done)
    hipExtLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb), dim3(256), 0, st, nullptr, done, 0, g, y, mean, rstd, gamma, dgamma,
                          dbeta, dy, gamma_grad, beta_grad, n4, C, (float)(1.0 / (double)npix), rep, rep ? (nrep < 1 ? 1 : nrep) : 0, rep_stride, xmax);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb), dim3(256), 0, st, g, y, mean, rstd, gamma, dgamma,
                       dbeta, dy, gamma_grad, beta_grad, n4, C, (float)(1.0 / (double)npix), rep, rep ? (nrep < 1 ? 1 : nrep) : 0, rep_stride, xmax);
  return hipGetLastError();
}

// ------------------------------------------------------------------ upsample+concat gradient split
__global__ void upsplit_prev_kernel(const float* __restrict__ dcat, int H, int W, int C0, int Ct, float* __restrict__ gprev,
                                    const float* __restrict__ pmask, const float* __restrict__ pscale,
                                    const float* __restrict__ pshift, size_t total, int accumulate) {
  const int C4 = C0 / 4, H2 = H / 2, W2 = W / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t p = i / C4;
    const int w2 = (int)(p % W2); p /= W2;
    const int h2 = (int)(p % H2); const int n = (int)(p / H2);
    const float* b = dcat + ((size_t)((size_t)n * H + 2 * h2) * W + 2 * w2) * Ct + c;
    f4 g = *(const f4*)b + *(const f4*)(b + Ct) + *(const f4*)(b + (size_t)W * Ct) + *(const f4*)(b + (size_t)W * Ct + Ct);
    if (pmask) {
      f4 z = *(const f4*)(pmask + i * 4);
      if (pscale) z = z * *(const f4*)(pscale + c) + *(const f4*)(pshift + c);
      g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f; g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
    }
    if (accumulate) g += *(const f4*)(gprev + i * 4);
    *(f4*)(gprev + i * 4) = g;
  }
}
__global__ void upsplit_skip_kernel(const float* __restrict__ dcat, int C0, int C1, float* __restrict__ gskip, size_t total) {
  const int C4 = C1 / 4, Ct = C0 + C1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t p = i / C4;
    *(f4*)(gskip + i * 4) = *(const f4*)(dcat + p * Ct + C0 + c);
  }
}
hipError_t launch_upsplit(const float* dcat, int N, int H, int W, int C0, int C1, float* gprev, const float* pmask,
                          const float* pscale, const float* pshift, float* gskip, hipStream_t st, int accumulate_prev) {
  const size_t t0 = (size_t)N * (H / 2) * (W / 2) * (C0 / 4);
  hipLaunchKernelGGL(upsplit_prev_kernel, dim3(nblocks(t0, 256)), dim3(256), 0, st, dcat, H, W, C0, C0 + C1, gprev, pmask,
                     pscale, pshift, t0, accumulate_prev);
  if (C1 > 0 && gskip) {
    const size_t t1 = (size_t)N * H * W * (C1 / 4);
    hipLaunchKernelGGL(upsplit_skip_kernel, dim3(nblocks(t1, 256)), dim3(256), 0, st, dcat, C0, C1, gskip, t1);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ input pipeline on the device
// uint8 HWC image (and its uint8 mask) -> the tensors the train step takes: Normalize(mean, std) of x/255 in NCHW
// fp32, mask > thr as uint8 {0,1}; the exact geometric augmentations of dataset.py's transforms (HorizontalFlip,
// VerticalFlip, RandomRotate90) are index arithmetic in the same pass.  flags[n]: bit0 hflip, bit1 vflip,
// bits 2-3 = k of rot90 (counter-clockwise, numpy/torch convention; needs H == W), applied in albumentations'
// pipeline order: flips first, then the rotation.
struct PreArgs { float mul[4], add[4]; };
__device__ __forceinline__ void aug_src(int flags, int H, int W, int y, int x, int& sy, int& sx) {
  // output (y, x) of rot90^k(flip(img)) -> coordinates in the flipped image, then undo the flips
  const int k = (flags >> 2) & 3;
  int fy = y, fx = x;
  if (k == 1) { fy = x; fx = W - 1 - y; }            // torch.rot90(a, 1)[y][x] = a[x][W-1-y]
  else if (k == 2) { fy = H - 1 - y; fx = W - 1 - x; }
  else if (k == 3) { fy = H - 1 - x; fx = y; }
  if (flags & 1) fx = W - 1 - fx;
  if (flags & 2) fy = H - 1 - fy;
  sy = fy; sx = fx;
}
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ img, int H, int W, int C, PreArgs pa,
                                     const int* __restrict__ flags, float* __restrict__ out, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); size_t r = i / W;
    const int y = (int)(r % H); const int n = (int)(r / H);
    int sy, sx; aug_src(flags ? flags[n] : 0, H, W, y, x, sy, sx);
    const uint8_t* p = img + (((size_t)n * H + sy) * W + sx) * C;
    for (int c = 0; c < C; ++c) out[(((size_t)n * C + c) * H + y) * W + x] = (float)p[c] * pa.mul[c] + pa.add[c];
  }
}
__global__ void preprocess_mask_kernel(const uint8_t* __restrict__ m, int H, int W, int thr, const int* __restrict__ flags,
                                       uint8_t* __restrict__ out, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); size_t r = i / W;
    const int y = (int)(r % H); const int n = (int)(r / H);
    int sy, sx; aug_src(flags ? flags[n] : 0, H, W, y, x, sy, sx);
    out[i] = m[((size_t)n * H + sy) * W + sx] > thr ? 1 : 0;
  }
}
hipError_t launch_preprocess_u8(const uint8_t* img, int N, int H, int W, int C, const float* mean, const float* std,
                                const int* flags, float* out, hipStream_t st) {
  if (C < 1 || C > 4) return hipErrorInvalidValue;
  PreArgs pa;
  for (int c = 0; c < 4; ++c) { pa.mul[c] = 0.f; pa.add[c] = 0.f; }
  for (int c = 0; c < C; ++c) { pa.mul[c] = 1.f / (255.f * std[c]); pa.add[c] = -mean[c] / std[c]; }
  const size_t total = (size_t)N * H * W;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, img, H, W, C, pa, flags, out, total);
  return hipGetLastError();
}
hipError_t launch_preprocess_mask(const uint8_t* m, int N, int H, int W, int thr, const int* flags, uint8_t* out, hipStream_t st) {
  const size_t total = (size_t)N * H * W;
  hipLaunchKernelGGL(preprocess_mask_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, m, H, W, thr, flags, out, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------ UNet++ dense skip plumbing
// dst[pix][dst_off + c] = act(src[pix][c])   (act = lazy BatchNorm scale/shift + ReLU of the producer, or identity)
__global__ void concat_copy_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int relu, int C, float* __restrict__ dst, int Cd,
                                   int dst_off, size_t total) {
  const int C4 = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t p = i / C4;
    f4 v = *(const f4*)(src + p * C + c);
    if (scale) {
      v = v * *(const f4*)(scale + c) + *(const f4*)(shift + c);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    *(f4*)(dst + p * Cd + dst_off + c) = v;
  }
}
hipError_t launch_concat_copy(const Src& s, size_t npix, float* dst, int Cd, int dst_off, hipStream_t st) {
  const size_t total = npix * (size_t)(s.C / 4);
  hipLaunchKernelGGL(concat_copy_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, s.ptr, s.scale, s.shift, s.relu, s.C, dst,
                     Cd, dst_off, total);
  return hipGetLastError();
}
// dst[pix][c] (+)= mask * gcat[pix][off + c],  mask = (m[pix][c]*mscale[c]+mshift[c] > 0) of the tensor the
// gradient belongs to (nullptr: no mask — the consumer applies its own)
__global__ void split_accum_kernel(const float* __restrict__ gcat, int Cc, int off, int C, float* __restrict__ dst,
                                   const float* __restrict__ m, const float* __restrict__ mscale,
                                   const float* __restrict__ mshift, int accumulate, size_t total) {
  const int C4 = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t p = i / C4;
    f4 g = *(const f4*)(gcat + p * Cc + off + c);
    if (m) {
      f4 z = *(const f4*)(m + p * C + c);
      if (mscale) z = z * *(const f4*)(mscale + c) + *(const f4*)(mshift + c);
      g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f; g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
    }
    float* d = dst + p * C + c;
    if (accumulate) g += *(const f4*)d;
    *(f4*)d = g;
  }
}
hipError_t launch_split_accum(const float* gcat, int Cc, int off, int C, size_t npix, float* dst, const float* m,
                              const float* mscale, const float* mshift, int accumulate, hipStream_t st) {
  const size_t total = npix * (size_t)(C / 4);
  hipLaunchKernelGGL(split_accum_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, gcat, Cc, off, C, dst, m, mscale, mshift,
                     accumulate, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------ dgrad weight repack
// wd[ci][tap*CoutP + co] = w[co][tap*Cin + ci]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, int Cout, int Kpad, int ntaps, int Cin, float* __restrict__ wd,
                                  int KpadD, int CoutP, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kd = (int)(i % KpadD), ci = (int)(i / KpadD);
    const int tap = kd / CoutP, co = kd - tap * CoutP;
    float v = 0.f;
    if (tap < ntaps && co < Cout) v = w[(size_t)co * Kpad + tap * Cin + ci];
    wd[i] = v;
  }
}
hipError_t launch_pack_dgrad(const float* w, int Cout, int Kpad, int ntaps, int Cin, float* wd, int KpadD, int CoutP,
                             hipStream_t st) {
  const size_t total = (size_t)Cin * KpadD;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(nblocks(total, 256)), dim3(256), 0, st, w, Cout, Kpad, ntaps, Cin, wd, KpadD,
                     CoutP, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------ per-channel column sum (bias gradient)
// Two stages, no float atomics (bit-reproducible): a workgroup stores ONE fp64 partial per channel into scratch[block][C];
// colsum_finish_kernel adds the partials in block order.  (scratch == nullptr: the single-launch form with float atomics.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, size_t npix, int C, float* out, double* scratch) {
  __shared__ float red[256 * 4];
  const int tc = C / 4, tr = 256 / tc;
  const int cx = threadIdx.x % tc, rx = threadIdx.x / tc;
  f4 s = {0, 0, 0, 0};
  if (rx < tr)
    for (size_t p = (size_t)blockIdx.x * tr + rx; p < npix; p += (size_t)gridDim.x * tr) s += *(const f4*)(g + p * C + cx * 4);
  float* r = red + threadIdx.x * 4;
  r[0] = s.x; r[1] = s.y; r[2] = s.z; r[3] = s.w;
  __syncthreads();
  if (threadIdx.x < C) {
    const int q = threadIdx.x / 4, e = threadIdx.x % 4;
    double acc = 0.0;
    for (int k = 0; k < tr; ++k) acc += (double)red[(k * tc + q) * 4 + e];
    if (scratch) scratch[(size_t)blockIdx.x * C + threadIdx.x] = acc;
    else atomicAdd(out + threadIdx.x, (float)acc);
  }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const double* __restrict__ scratch, int nparts, int C, float* out) {
  // thread (channel c, group g of G = 256 / C): partials g, g + G, ... (independent loads), then the G group sums in group order
  __shared__ double red[256];
  const int G = 256 / C, c = threadIdx.x % C, g = threadIdx.x / C;
  double acc = 0.0;
  if (g < G)
    for (int k = g; k < nparts; k += G) acc += scratch[(size_t)k * C + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double t = red[threadIdx.x];
    for (int k = 1; k < G; ++k) t += red[k * C + threadIdx.x];
    out[threadIdx.x] += (float)t;
  }
}
size_t colsum_scratch_doubles(int C) { return (size_t)256 * C; }
hipError_t launch_colsum(const float* g, size_t npix, int C, float* out, double* scratch, hipStream_t st) {
  if ((C & 3) || C > 256 || (256 % (C / 4))) return hipErrorInvalidValue;
  const int tr = 256 / (C / 4);
  unsigned nb = nblocks(npix, tr * 16);
  if (nb > (scratch ? 256u : 1024u)) nb = scratch ? 256u : 1024u;
  hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, st, g, npix, C, out, scratch);
  if (scratch) hipLaunchKernelGGL(colsum_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)scratch, (int)nb, C, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam, coupled L2)
// hyp != nullptr (hipGraph-captured train step, uwm_adam_graph): every hyper-parameter comes from DEVICE memory, so one captured
// launch serves every step — hyp = {lr, beta1, beta2, eps, weight_decay, grad_scale, max_norm, step, bc1, sqrt(bc2)}; the
// step counter and the two bias corrections are advanced by adam_hyper_kernel (one thread) in front of every adam launch
__global__ void adam_hyper_kernel(float* hyp) {
  const float step = hyp[7] + 1.f;
  hyp[7] = step;
  hyp[8] = 1.f - powf(hyp[1], step);
  hyp[9] = sqrtf(1.f - powf(hyp[2], step));
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n4, size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s,
                            float gscale, const double* sumsq, float max_norm, const float* __restrict__ hyp) {
  if (hyp) { lr = hyp[0]; b1 = hyp[1]; b2 = hyp[2]; eps = hyp[3]; wd = hyp[4]; gscale = hyp[5]; max_norm = hyp[6]; bc1 = hyp[8]; bc2s = hyp[9]; }
  if (sumsq) {   // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float tn = (float)sqrt(*sumsq) * gscale;
    gscale *= fminf(1.f, max_norm / (tn + 1e-6f));
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i * 4;
    if (o + 4 <= n) {
      f4 pv = *(f4*)(p + o), gv = *(const f4*)(g + o) * gscale, mv = *(f4*)(m + o), vv = *(f4*)(v + o);
      gv += wd * pv;
      mv = mv + (1.f - b1) * (gv - mv);          // torch: m.lerp_(g, 1-b1)
      vv = b2 * vv + (1.f - b2) * gv * gv;
      f4 den;
      den.x = sqrtf(vv.x) / bc2s + eps; den.y = sqrtf(vv.y) / bc2s + eps; den.z = sqrtf(vv.z) / bc2s + eps; den.w = sqrtf(vv.w) / bc2s + eps;
      pv -= (lr / bc1) * (mv / den);
      *(f4*)(p + o) = pv; *(f4*)(m + o) = mv; *(f4*)(v + o) = vv;
    } else {
      for (size_t j = o; j < n; ++j) {
        float gj = g[j] * gscale + wd * p[j];
        float mj = m[j] + (1.f - b1) * (gj - m[j]);
        float vj = b2 * v[j] + (1.f - b2) * gj * gj;
        p[j] -= (lr / bc1) * (mj / (sqrtf(vj) / bc2s + eps));
        m[j] = mj; v[j] = vj;
      }
    }
  }
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, float bc1, float bc2, float gscale, hipStream_t st, const double* sumsq, float max_norm) {
  const size_t n4 = (n + 3) / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n4, 256)), dim3(256), 0, st, p, g, m, v, n4, n, lr, b1, b2, eps, wd, bc1,
                     sqrtf(bc2), gscale, sumsq, max_norm, (const float*)nullptr);
  return hipGetLastError();
}
hipError_t launch_adam_graph(float* p, const float* g, float* m, float* v, size_t n, float* hyp, const double* sumsq, hipStream_t st) {
  const size_t n4 = (n + 3) / 4;
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(1), 0, st, hyp);
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n4, 256)), dim3(256), 0, st, p, g, m, v, n4, n, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f,
                     sumsq, 0.f, (const float*)hyp);
  return hipGetLastError();
}

// ------------------------------------------------------------------ SGD with momentum (torch.optim.SGD: coupled L2, dampening 0, no Nesterov)
// g' = g*gscale + wd*p ; buf = first ? g' : momentum*buf + g' ; p -= lr*buf
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n4, size_t n, float lr,
                           float momentum, float wd, int first, float gscale, const double* sumsq, float max_norm) {
  if (sumsq) {
    const float tn = (float)sqrt(*sumsq) * gscale;
    gscale *= fminf(1.f, max_norm / (tn + 1e-6f));
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i * 4;
    if (o + 4 <= n) {
      f4 pv = *(f4*)(p + o), gv = *(const f4*)(g + o) * gscale;
      gv += wd * pv;
      f4 bv = gv;
      if (!first) bv = momentum * *(f4*)(buf + o) + gv;
      pv -= lr * bv;
      *(f4*)(p + o) = pv; *(f4*)(buf + o) = bv;
    } else {
      for (size_t j = o; j < n; ++j) {
        const float gj = g[j] * gscale + wd * p[j];
        const float bj = first ? gj : momentum * buf[j] + gj;
        p[j] -= lr * bj; buf[j] = bj;
      }
    }
  }
}
hipError_t launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int first, float gscale,
                      hipStream_t st, const double* sumsq, float max_norm) {
  const size_t n4 = (n + 3) / 4;
  hipLaunchKernelGGL(sgd_kernel, dim3(nblocks(n4, 256)), dim3(256), 0, st, p, g, buf, n4, n, lr, momentum, wd, first, gscale, sumsq, max_norm);
  return hipGetLastError();
}

// sum of squares of a flat range (global gradient norm), fp64 across workgroups
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n4, size_t n, double* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i * 4;
    if (o + 4 <= n) { const f4 v = *(const f4*)(g + o); acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    else for (size_t j = o; j < n; ++j) acc += g[j] * g[j];
  }
  double d = (double)acc;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
hipError_t launch_sumsq(const float* g, size_t n, double* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
  if (e != hipSuccess) return e;
  const size_t n4 = (n + 3) / 4;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks(n4, 1024)), dim3(256), 0, st, g, n4, n, out);
  return hipGetLastError();
}

__global__ void scale_kernel(float* __restrict__ p, size_t n, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] *= s;
}
__global__ __launch_bounds__(256) void absmax32_kernel(const float* __restrict__ x, size_t n, float* out32) {
  __shared__ float wmax[4];
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (mx > 0.f) atomicMax((unsigned*)out32 + (blockIdx.x & 31), __float_as_uint(mx));
  }
}
hipError_t launch_absmax32(const float* x, size_t n, float* out32, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out32, 0, 32 * sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(absmax32_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, st, x, n, out32);
  return hipGetLastError();
}
hipError_t launch_scale(float* p, size_t n, float s, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(nblocks(n, 256)), dim3(256), 0, st, p, n, s);
  return hipGetLastError();
}

// ------------------------------------------------------------------ HIP-event profiler (host side)
bool dbg_flag(const char* name) {
  static const bool on = [] { const char* e = getenv("UWM_DEBUG"); return e && e[0] == '1'; }();
  return on && getenv(name) != nullptr;
}
int dbg_int(const char* name, int dflt) {
  if (!dbg_flag(name)) return dflt;
  return atoi(getenv(name));
}

struct ProfRec { int cls; double flops, bytes; hipEvent_t e0, e1; };
static thread_local const char* g_route_last = "";
void route_note(const char* kernel) { g_route_last = kernel; }
const char* route_last() { return g_route_last; }
static std::atomic<bool> g_prof{false};
static std::mutex g_prof_mu;                      // records / event pool: launches may come from one host thread per GPU
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t prof_event() {                  // g_prof_mu held
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void prof_enable(bool on) { g_prof = on; }
bool prof_on() { return g_prof; }
void prof_pair(int cls, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r; r.cls = cls; r.flops = flops; r.bytes = bytes; r.e0 = prof_event(); r.e1 = prof_event();
  *e0 = r.e0; *e1 = r.e1;
  g_recs.push_back(r);
}
int prof_collect(double* out) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < kProfClasses * 4; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && r.cls >= 0 && r.cls < kProfClasses) {
      out[r.cls * 4 + 0] += 1.0; out[r.cls * 4 + 1] += (double)ms; out[r.cls * 4 + 2] += r.flops; out[r.cls * 4 + 3] += r.bytes;
    }
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  const int n = (int)g_recs.size();
  g_recs.clear();
  return n;
}
const char* prof_class_name(int cls) {
  static const char* names[kProfClasses] = {
      "conv_igemm_kernel<128,128,2,2>", "conv_igemm_kernel<128,64,2,2>", "conv_igemm_kernel<128,32,4,1>",
      "conv_igemm_kernel<128,16,4,1>",  "conv_igemm_kernel<64,64,2,2>",  "conv_igemm_kernel<64,128,1,4>",
      "wgrad_igemm_kernel<64,128,2,2>", "wgrad_igemm_kernel<128,128,2,2>", "wgrad_igemm_kernel<16,256,1,4>",
      "wgrad_igemm_kernel<32,256,1,4>", "conv_patch_kernel<128,2,2>", "conv_patch_kernel<64,2,2>",
      "conv_patch_kernel<32,4,1>",      "conv_patch_kernel<16,4,1>",      "wgrad_patch_kernel<16>",
      "wgrad_patch_kernel<32>",         "wgrad_patch_kernel<64>",         "conv_patch16_kernel<16>",
      "conv_patch16_kernel<32>",        "conv_wino_kernel<64>",           "conv_wino_kernel<32>",
      "conv_wino_kernel<16>",           "wgrad_wino2_kernel<64>",         "wgrad_wino2_kernel<32>",
      "wgrad_wino_kernel<16>",          "conv_wino8_kernel",              "wgrad_igemm_kernel<128,32,2,2>",
      "wgrad_igemm_kernel<128,64,2,2>",  "wgrad_igemm_kernel<32,64,2,2>",  "wgrad_igemm_kernel<32,128,1,4>",
      "conv_head_kernel",               "conv_wino_x3_kernel",            "wgrad_c16_kernel",
      "wgrad_head_kernel",              "conv_up2_kernel",                "conv_up2_dgrad_kernel",
      "wgrad_up2_kernel",               "conv_gemm_kernel<128>",          "conv_gemm_kernel<64>",
      "wgrad_gemm_kernel<128>",         "wgrad_gemm_kernel<64>",          "wgrad_stem_kernel",
      "conv_f16x3_kernel",              "wgrad_f16x3_kernel",             "conv_stem_f16x3_kernel",
      "conv_up2_f16x3_kernel",          "conv_up2_dgrad_f16x3_kernel",    "wgrad_up2_f16x3_kernel",
      "conv_c16_f16x3_kernel",          "wgrad_c16_f16x3_kernel",         "wgrad_stem_f16x3_kernel",
      "conv_gemm_f16x3_kernel<128>",    "conv_gemm_f16x3_kernel<64>",
      "conv_igemm_f16x3_kernel<128,128,2,2>", "conv_igemm_f16x3_kernel<128,64,2,2>", "conv_igemm_f16x3_kernel<128,32,4,1>",
      "conv_igemm_f16x3_kernel<128,16,4,1>",  "conv_igemm_f16x3_kernel<64,64,2,2>",  "conv_igemm_f16x3_kernel<64,128,1,4>",
      "?", "wgrad_igemm_f16x3_kernel<128,128,2,2>", "wgrad_igemm_f16x3_kernel<128,64,2,2>", "wgrad_igemm_f16x3_kernel<64,128,2,2>", "?"};
  return (cls >= 0 && cls < kProfClasses) ? names[cls] : "?";
}

}  // namespace uwm
