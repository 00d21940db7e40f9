// Implicit-GEMM convolution for gfx950 (MI355X) on the exact-fp32 matrix core
// instruction v_mfma_f32_16x16x4_f32.
//
//   D[co][pixel] += sum_k W[co][k] * X[pixel][k],   k = tap*Ctot + c
//
// * A operand = weights (rows = output channels), B operand = gathered input pixels, so each
//   lane ends up holding 4 CONSECUTIVE output channels of one pixel -> one 16-byte NHWC store.
// * X is gathered on the fly (no im2col buffer): per 16-byte unit the loader derives
//   (tap, channel, source) and applies the producer's BatchNorm scale/shift + ReLU, nearest x2
//   upsampling and channel concat while staging the tile (register-staged, double-buffered LDS).
// * LDS tile rows are 32 floats (128 B) with the 16-byte unit index XOR-swizzled by (row>>1)&7:
//   conflict-free for the ds_read_b128 lane groups of gfx950.
// * The same kernel runs dgrad: transposed gather (rmul=-1, sdiv=stride) over a [Cin][tap][Cout]
//   repack of the weights; its epilogue can add a residual gradient and apply a ReLU mask.
// * Optional per-channel sum / sum-of-squares of the output (BatchNorm batch statistics) are
//   reduced in-wave (DPP), across waves through LDS and across blocks with fp64 atomics.
//
// Reference semantics replaced: torch.nn.functional.conv2d as used by smp.Unet
// (/root/reference/src/models/unet_model.py:64-71 -> smp; SURVEY.md §8 a3-a11,a14).
#include "uwm_kernels.h"
#include <cstdlib>
#include <cstdio>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) {
  return f.d <= 1 ? n : __umulhi(n, f.mg);
}

// PC = stride-2 dgrad by output-pixel parity class (class 2*pc_h + pc_w = 3 - blockIdx.y: the four-tap class of a 3x3 is dispatched first): a workgroup covers only the pixels
// (2i + pc_h, 2j + pc_w) and walk only the taps that reach an input pixel from that class (1, 2, 2 or 4 of the
// nine; pc_taps lists them) — the plain transposed gather spends 3/4 of its MFMAs on taps its pixels cannot use.
// F16 (round 4) = the fp16x3 arithmetic of conv_f16x3.hip on this kernel's tiles: every staged fp32 value is split into hi + lo fp16
// halves on its way into LDS (a row keeps its 128 bytes: 32 hi halfs | 32 lo halfs, 16-byte slots XOR-swizzled as before), a
// 32-deep chunk is ONE k-step of v_mfma_f32_16x16x32_f16 and a product is hi*hi' + hi*lo' + lo*hi' — 3 MFMAs of 16 cycles where
// the fp32 form issues 8 of 32.  Range: weights times 2^12 (kaiming-scale weights land near 2^8; |w| < 16 cannot overflow, smaller
// ones keep their low half normal), a dgrad's dY times the power of two that puts max|dY| into [2^13, 2^14) (ConvArgs::xmax),
// activations as they are, clamped to +-65504; undone on the accumulator (exact powers of two).
constexpr float kIgWScale = 4096.f;
typedef _Float16 ig_h8 __attribute__((ext_vector_type(8)));
template <int BM, int BN, int WM, int WN, bool PC = false, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int MI = BM / WM / 16;      // 16-pixel MFMA tiles per wave
  constexpr int NI = BN / WN / 16;      // 16-channel MFMA tiles per wave
  constexpr int XR = BM / 32;           // X rows staged per thread
  constexpr int WR = (BN + 31) / 32;    // W rows staged per thread
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Xs = smem;                       // [2][BM][32]
  float* const Ws = smem + 2 * BM * 32;         // [2][BN][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware bijective block remap: blocks that share an XCD (bid % 8) get a contiguous range
  // of tiles, so halo rows / the weight panel stay in that XCD's L2.
  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int unit = tid & 7, r0 = tid >> 3;

  // per staged row: pixel -> (n, ho, wo)
  int rn[XR], rh[XR], rw[XR];
  unsigned rvalid = 0;
  const int Wg = PC ? (a.Wo >> 1) : a.Wo;                    // pixel grid this launch walks
  const int HoWo = PC ? (a.Ho >> 1) * Wg : a.Ho * a.Wo;
  const int Mg = PC ? a.N * HoWo : a.M;
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    int m = m0 + r0 + 32 * i;
    bool v = m < Mg;
    int mm = v ? m : 0;
    int n = mm / HoWo, rem = mm - n * HoWo;
    int ho = rem / Wg, wo = rem - ho * Wg;
    if (PC) { ho = 2 * ho + (int)((3u - blockIdx.y) >> 1); wo = 2 * wo + (int)((3u - blockIdx.y) & 1); }
    rn[i] = n; rh[i] = ho * a.smul + a.off; rw[i] = wo * a.smul + a.off;
    rvalid |= (v ? 1u : 0u) << i;
  }

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  float xs = 1.f;                                      // F16: power-of-two scale of a dgrad's dY
  if (F16 && a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  f4 xr[XR], wr[WR], tsc, tsh;
  unsigned xvalid = 0; int trelu = 0; bool thas = false;

  const int cpt = PC ? (a.Ctot >> 5) : 1;                    // PC: 32-float chunks per tap (Ctot % 32 == 0)
  auto load_chunk = [&](int kci) {
    int kc = kci;
    if (PC) { const int ti = kci / cpt; kc = ((a.pc_taps[3u - blockIdx.y] >> (4 * ti)) & 15) * cpt + (kci - ti * cpt); }
    const unsigned k = kc * 32 + unit * 4;
    const unsigned tap = fdiv(k, a.dv_ctot);
    const int c = k - tap * a.Ctot;
    const unsigned r = fdiv(tap, a.dv_kw);
    const int s = tap - r * a.kw;
    const bool tv = tap < (unsigned)a.ntaps;
    const bool first = c < a.C0;
    const float* sp = first ? a.s0.ptr : a.s1.ptr;
    const float* ssc = first ? a.s0.scale : a.s1.scale;
    const float* ssh = first ? a.s0.shift : a.s1.shift;
    const int sC = first ? a.s0.C : a.s1.C, sH = first ? a.s0.H : a.s1.H, sW = first ? a.s0.W : a.s1.W;
    const int sup = first ? a.s0.up : a.s1.up;
    trelu = first ? a.s0.relu : a.s1.relu;
    const int cc = first ? c : c - a.C0;
    thas = (ssc != nullptr) && tv;
    if (thas) { tsc = *(const f4*)(ssc + cc); tsh = *(const f4*)(ssh + cc); }
    const int dr = (int)r * a.rmul, ds = s * a.rmul;
    xvalid = 0;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      int hn = rh[i] + dr, wq = rw[i] + ds;
      bool v = tv && ((rvalid >> i) & 1u);
      if (a.sdiv == 2) { v = v && (((hn | wq) & 1) == 0); hn >>= 1; wq >>= 1; }
      v = v && hn >= 0 && hn < a.Hl && wq >= 0 && wq < a.Wl;
      hn >>= sup; wq >>= sup;
      const float* p = sp + ((size_t)((size_t)rn[i] * sH + hn) * sW + wq) * sC + cc;
      xr[i] = v ? *(const f4*)p : (f4){0.f, 0.f, 0.f, 0.f};
      xvalid |= (v ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      int row = n0 + r0 + 32 * i;
      bool v = (row < a.wrows) && (BN >= 32 || r0 < BN);
      wr[i] = v ? *(const f4*)(a.w + (size_t)row * a.Kpad + kc * 32 + unit * 4) : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };

  const float xs_ = xs;
  auto store_chunk = [&](int buf) {
    float* xs = Xs + buf * BM * 32;
    float* ws = Ws + buf * BN * 32;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      f4 v = xr[i];
      if (thas) {
        v = v * tsc + tsh;
        if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (!((xvalid >> i) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
      }
      const int row = r0 + 32 * i;
      if (F16) {          // hi halfs of k = 4 unit .. +3 -> bytes [8 unit, 8 unit + 8) of the row's hi half, lo halfs -> the same place in the lo half
        uwm_u2 hi, lo;
        const float sc = xs_;
        uwm_split4(__builtin_amdgcn_fmed3f(v.x * sc, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y * sc, -65504.f, 65504.f),
                   __builtin_amdgcn_fmed3f(v.z * sc, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w * sc, -65504.f, 65504.f), hi, lo);
        char* rb = (char*)(xs + row * 32);
        const int sw = (row >> 1) & 7;
        *(uwm_u2*)(rb + (((unit >> 1) ^ sw) << 4) + (unit & 1) * 8) = hi;
        *(uwm_u2*)(rb + ((((unit >> 1) + 4) ^ sw) << 4) + (unit & 1) * 8) = lo;
      } else
      *(f4*)(xs + row * 32 + ((unit ^ ((row >> 1) & 7)) << 2)) = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int row = r0 + 32 * i;
      if (F16) {
        if (BN >= 32 || r0 < BN) {
          uwm_u2 hi, lo;
          const f4 w4 = wr[i] * kIgWScale;
          uwm_split4(__builtin_amdgcn_fmed3f(w4.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w4.y, -65504.f, 65504.f),
                     __builtin_amdgcn_fmed3f(w4.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(w4.w, -65504.f, 65504.f), hi, lo);
          char* rb = (char*)(ws + row * 32);
          const int sw = (row >> 1) & 7;
          *(uwm_u2*)(rb + (((unit >> 1) ^ sw) << 4) + (unit & 1) * 8) = hi;
          *(uwm_u2*)(rb + ((((unit >> 1) + 4) ^ sw) << 4) + (unit & 1) * 8) = lo;
        }
      } else
      if (BN >= 32 || r0 < BN) *(f4*)(ws + row * 32 + ((unit ^ ((row >> 1) & 7)) << 2)) = wr[i];
    }
  };

  const int nk = PC ? a.pc_ntaps[3u - blockIdx.y] * cpt : (a.Kpad >> 5);
  if (nk > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();

  const int lrow = lane & 15, lq = lane >> 4;
  for (int kc = 0; kc < nk; ++kc) {
    const int cur = kc & 1;
    load_chunk(kc + 1 < nk ? kc + 1 : kc);     // unconditional prefetch: single-basic-block loop body
    const float* xs = Xs + cur * BM * 32;
    const float* ws = Ws + cur * BN * 32;
    if (F16) {
      // lane (row lrow, k group lq): 8 consecutive k = one 16-byte slot of the row's hi half, the same slot + 4 of its lo half
      ig_h8 xh[MI], xl[MI], wh[NI], wl[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = (wm * MI + i) * 16 + lrow;
        const char* rb = (const char*)(xs + row * 32);
        const int sw = (row >> 1) & 7;
        xh[i] = *(const ig_h8*)(rb + ((lq ^ sw) << 4)); xl[i] = *(const ig_h8*)(rb + (((lq + 4) ^ sw) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = (wn * NI + j) * 16 + lrow;
        const char* rb = (const char*)(ws + row * 32);
        const int sw = (row >> 1) & 7;
        wh[j] = *(const ig_h8*)(rb + ((lq ^ sw) << 4)); wl[j] = *(const ig_h8*)(rb + (((lq + 4) ^ sw) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
    } else
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      f4 xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = (wm * MI + i) * 16 + lrow;
        xf[i] = *(const f4*)(xs + row * 32 + (((k16 * 4 + lq) ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = (wn * NI + j) * 16 + lrow;
        wf[j] = *(const f4*)(ws + row * 32 + (((k16 * 4 + lq) ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[i][e], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // consumers of this step's global loads stay below the MFMA block
    store_chunk(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue: lane (p = lane&15 -> pixel, q = lane>>4 -> 4 channels) --------------
  const bool do_stats = a.ssum != nullptr;
  // BatchNorm statistics go to one of a.srep copies (few-channel layers launch tens of thousands of workgroups:
  // fp64 atomics on the same 2*C addresses serialise), bn_finalize adds the copies
  const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
  f4 ps[NI], pq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { ps[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq[j] = ps[j]; }
  // fused BatchNorm-backward sums (uwm_kernels.h ConvArgs::bnb_*): second sum = v * yhat of the mask tensor (round 2: the 1x1 and
  // stride-2 dgrads of the Bottleneck encoder feed bn2 / bn1 this way)
  const bool bnb = a.bnb_mean != nullptr;
  const float unscale = 1.f / (kIgWScale * xs);        // (F16: exact powers of two)
  f4 bmu[NI], brs[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int co = n0 + (wn * NI + j) * 16 + lq * 4;
    bmu[j] = brs[j] = (f4){0.f, 0.f, 0.f, 0.f};
    if (bnb && co < a.Cout) { bmu[j] = *(const f4*)(a.bnb_mean + co); brs[j] = *(const f4*)(a.bnb_rstd + co); }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = m0 + (wm * MI + i) * 16 + lrow;
    const bool mv = m < Mg;
    if (PC && mv) {                                // class-grid index -> real output pixel
      const int n = m / HoWo, rem = m - n * HoWo;
      const int hi = rem / Wg, wi = rem - hi * Wg;
      m = (n * a.Ho + 2 * hi + (int)((3u - blockIdx.y) >> 1)) * a.Wo + 2 * wi + (int)((3u - blockIdx.y) & 1);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = n0 + (wn * NI + j) * 16 + lq * 4;
      if (mv && co < a.Cout) {
        f4 v = acc[i][j];
        if (F16) v = v * unscale;
        const size_t o = (size_t)m * a.Cout + co;
        if (a.bias) v += *(const f4*)(a.bias + co);
        if (a.addend) v += *(const f4*)(a.addend + o);
        f4 yr = {0.f, 0.f, 0.f, 0.f};
        if (a.mask) {
          f4 mk = *(const f4*)(a.mask + o);
          yr = mk;
          if (a.mscale) mk = mk * *(const f4*)(a.mscale + co) + *(const f4*)(a.mshift + co);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        *(f4*)(a.out + o) = v;
        if (a.bnb_y) yr = *(const f4*)(a.bnb_y + o);
        ps[j] += v; pq[j] += bnb ? v * ((yr - bmu[j]) * brs[j]) : v * v;
      }
    }
  }
  if (do_stats) {
    // reduce over the 16 pixel lanes (xor 1,2,4,8 stays inside a 16-lane row)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = ps[j][e], q = pq[j][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
        ps[j][e] = s; pq[j][e] = q;
      }
    float* red = smem;                       // [WM][BN][2]; main-loop LDS is dead after the last barrier
    if (lrow == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = (wn * NI + j) * 16 + lq * 4 + e;
          red[(wm * BN + cl) * 2 + 0] = ps[j][e];
          red[(wm * BN + cl) * 2 + 1] = pq[j][e];
        }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = n0 + tid;
      if (co < a.Cout) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s += (double)red[(w * BN + tid) * 2]; q += (double)red[(w * BN + tid) * 2 + 1]; }
        atomicAdd(a.ssum + srep_off + co, s);
        atomicAdd(a.ssq + srep_off + co, q);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool F16>
static hipError_t launch_cfg_(const ConvArgs& a, hipStream_t st, int cls) {
  const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_igemm_kernel<BM, BN, WM, WN, false, F16>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(F16 ? 53 + cls : cls, a.flops, a.bytes, (conv_igemm_kernel<BM, BN, WM, WN, false, F16>), dim3((unsigned)(tilesM * tilesN)), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t st, int cls) {
  return a.ig16 ? launch_cfg_<BM, BN, WM, WN, true>(a, st, cls) : launch_cfg_<BM, BN, WM, WN, false>(a, st, cls);
}

// stride-2 dgrad as four parity-class launches (see the PC template flag)
bool conv_s2_dgrad_applicable(const ConvArgs& a) {
  return a.rmul == -1 && a.sdiv == 2 && a.smul == 1 && (a.ntaps == 9 || a.ntaps == 1) && (a.Ctot & 31) == 0 && a.C0 == a.Ctot &&
         (a.Ho & 1) == 0 && (a.Wo & 1) == 0 && a.s0.up == 0;
}
template <int BM, int BN, int WM, int WN, bool F16>
static hipError_t launch_s2_dgrad_(const ConvArgs& a0, hipStream_t st, int cls);
template <int BM, int BN, int WM, int WN>
static hipError_t launch_s2_dgrad(const ConvArgs& a0, hipStream_t st, int cls) {
  return a0.ig16 ? launch_s2_dgrad_<BM, BN, WM, WN, true>(a0, st, cls) : launch_s2_dgrad_<BM, BN, WM, WN, false>(a0, st, cls);
}
template <int BM, int BN, int WM, int WN, bool F16>
static hipError_t launch_s2_dgrad_(const ConvArgs& a0, hipStream_t st, int cls) {
  const int Mg = a0.N * (a0.Ho >> 1) * (a0.Wo >> 1);
  const int tilesM = (Mg + BM - 1) / BM, tilesN = (a0.Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM + BN) * 32 * sizeof(float);
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_igemm_kernel<BM, BN, WM, WN, true, F16>, lds); if (e != hipSuccess) return e; }
  ConvArgs a = a0;
  for (int pc = 0; pc < 4; ++pc) {
    a.pc_taps[pc] = 0; a.pc_ntaps[pc] = 0;
    for (int r = 0; r < a.kw; ++r)
      for (int s = 0; s < a.kw; ++s)         // tap (r, s) reaches an input pixel iff (ho - r + off) and (wo - s + off) are even
        if (((((pc >> 1) - r + a.off) | ((pc & 1) - s + a.off)) & 1) == 0) { a.pc_taps[pc] |= (unsigned)(r * a.kw + s) << (4 * a.pc_ntaps[pc]); ++a.pc_ntaps[pc]; }
  }
  UWM_LAUNCH(F16 ? 53 + cls : cls, a.flops, a.bytes, (conv_igemm_kernel<BM, BN, WM, WN, true, F16>), dim3((unsigned)(tilesM * tilesN), 4), dim3(256), lds, st, a);
  return hipGetLastError();
}

static int g_winograd = -1;
bool winograd_enabled() {
  if (g_winograd < 0) { const char* e = getenv("UWM_WINOGRAD"); g_winograd = (e && e[0] == '0') ? 0 : 1; }
  return g_winograd != 0;
}
void winograd_set_mode(int mode) { g_winograd = mode; }
int winograd_mode() { (void)winograd_enabled(); return g_winograd; }
int device_cu_count() {
  static int cus[64] = {0};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return 256;
  if (!cus[dev] && (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] <= 0)) cus[dev] = 256;
  return cus[dev];
}

// mirrors the auto routing below: true when launch_conv(a, st) ends on a kernel whose epilogue can carry the fused
// BatchNorm-backward sums (ConvArgs::bnb_*): conv_wino_kernel<NI>, conv_wino_x3_kernel, conv_patch16_kernel, conv_head_dgrad_kernel,
// conv_up2_dgrad_kernel, conv_igemm_kernel, conv_c16_f16_kernel
// (with a.bnb_y set: only the epilogues that read yhat from a separate tensor count — conv_wino_kernel<NI> and conv_igemm_kernel)
bool conv_epilogue_carries_bnb(const ConvArgs& a) {
  if (a.ig16 && a.prec != 2 && (conv_c16_f16_applicable(a) || conv_c32_f16_applicable(a))) return true;      // conv_c16_f16_kernel / conv_c32_f16_kernel (dgrad epilogue)
  if (a.prec == 2) return !a.out_up && (a.wu_layout == 1 ? conv_f16x3v2_applicable(a) : conv_f16x3_applicable(a));
  if (a.bnb_y) {
    if (a.out_up || a.prec == 1) return false;
    static const bool no_y = dbg_flag("UWM_NO_BNB_Y");
    if (no_y || conv_up2_applicable(a) || conv_head_applicable(a) || conv_head_dgrad_applicable(a) || conv_patch16_applicable(a)) return false;
    if (wino_mode_of(a.wino) != 0 && conv_wino_applicable(a)) return true;      // (launch_conv_wino skips the 8-wave variant when bnb_mean is set)
    return !conv_patch_applicable(a);
  }
  static const bool no_up2 = dbg_flag("UWM_NO_UP2");
  if (a.out_up) return (!no_up2 && conv_up2_dgrad_applicable(a)) || (a.prec == 1 ? conv_wino_x3_applicable(a) : conv_wino_applicable(a));
  static const bool no_head = dbg_flag("UWM_NO_CONV_HEAD");
  if (!no_up2 && conv_up2_applicable(a)) return false;
  if (!no_head && conv_head_applicable(a)) return false;
  if (!no_head && conv_head_dgrad_applicable(a)) return true;
  if (wino_mode_of(a.wino) != 0 && conv_wino_applicable(a) && !conv_patch16_applicable(a))
    return a.prec == 1 || !conv_wino8_applicable(a);      // (the 8-wave variant has no fused BatchNorm-backward sums)
  if (conv_patch16_applicable(a)) return true;
  if (conv_patch_applicable(a)) return false;             // (direct 3x3 kernels of Winograd mode 0: no bnb epilogue)
  static const bool no_igemm_bnb = dbg_flag("UWM_NO_IGEMM_BNB");
  return !no_igemm_bnb;                                   // flattened implicit GEMM / stride-2 parity classes (conv_igemm.hip)
}

// tile configurations: {BM, BN}: 0:{128,128} 1:{128,64} 2:{128,32} 3:{128,16} 4:{64,64} 5:{64,128}
hipError_t launch_conv(const ConvArgs& a, hipStream_t st, int force_cfg) {
  if (a.M <= 0 || a.Cout <= 0 || (a.Cout & 3) || (a.Kpad & 31) || (a.Ctot & 3) || (a.C0 & 3)) return hipErrorInvalidValue;
  int cfg = force_cfg;
  static const bool trace = dbg_flag("UWM_TRACE_CONV");
  if (trace)
    fprintf(stderr, "conv %s N=%d Ctot=%d(C0=%d up=%d) Cout=%d Ho=%d Wo=%d Hl=%d Wl=%d taps=%d smul=%d sdiv=%d wino=%d gflop=%.2f\n",
            a.rmul < 0 ? "dgrad" : "fwd", a.N, a.Ctot, a.C0, a.s0.up, a.Cout, a.Ho, a.Wo, a.Hl, a.Wl, a.ntaps, a.smul, a.sdiv,
            (int)(cfg < 0 && wino_mode_of(a.wino) != 0 && conv_wino_applicable(a)), a.flops * 1e-9);
  if (cfg == 700) return launch_conv_up2(a, st);
  if (cfg == 710 || (cfg < 0 && a.ig16 && a.prec != 2 && conv_c16_f16_applicable(a))) return launch_conv_c16_f16(a, st);
  if (cfg == 711 || (cfg < 0 && a.ig16 && a.prec != 2 && conv_c32_f16_applicable(a))) return launch_conv_c32_f16(a, st);      // 32 -> 32, fp16x3      // 16 -> 16 at full resolution, fp16x3
  if (cfg >= 800 && cfg < 1000) return launch_conv_gemm(a, st, cfg - 800);
  if (cfg == 500) return launch_conv_head(a, st);
  if ((cfg >= 600 && cfg <= 607) || (cfg < 0 && a.prec == 2)) return launch_conv_f16x3(a, st, cfg >= 600 ? cfg - 600 : 0);      // fp16x3 direct form: the bank behind a.wu is a conv_f16x3.hip one
  if (cfg == 400) return launch_conv_wino_x3(a, st);
  if (cfg >= 300) return launch_conv_wino(a, st, cfg - 300);
  if (a.out_up) {                                                  // fused concat split: Winograd epilogues (and the sub-pixel dgrad of conv_up2.hip)
    static const bool no_up2d = dbg_flag("UWM_NO_UP2");
    if (cfg < 0 && !no_up2d && conv_up2_dgrad_applicable(a)) return launch_conv_up2_dgrad(a, st);
    if (a.prec == 1) return launch_conv_wino_x3(a, st);
    return conv_wino_applicable(a) ? launch_conv_wino(a, st) : hipErrorInvalidValue;
  }
  if (cfg == 200) return launch_conv_patch16(a, st);
  static const bool no_up2 = dbg_flag("UWM_NO_UP2");
  if (cfg < 0 && !no_up2 && conv_up2_applicable(a)) return launch_conv_up2(a, st);      // sub-pixel decomposition: Winograd's 2.25x without transforms
  static const bool no_head = dbg_flag("UWM_NO_CONV_HEAD");
  if (cfg < 0 && !no_head && conv_head_applicable(a)) return launch_conv_head(a, st);
  if (cfg < 0 && !no_head && conv_head_dgrad_applicable(a)) return launch_conv_head_dgrad(a, st);      // few channels -> <= 4 classes: HBM streaming kernel
  // 16-channel inputs at full resolution are HBM-bound: the one-barrier direct kernel beats the Winograd pipeline there
  if (cfg < 0 && wino_mode_of(a.wino) != 0 && conv_wino_applicable(a) && !conv_patch16_applicable(a))
    return a.prec == 1 ? launch_conv_wino_x3(a, st) : launch_conv_wino(a, st);      // prec 1: the bank behind a.wu is a bf16x3 one
  if (cfg >= 100) return launch_conv_patch(a, st, cfg - 100);
  if (cfg < 0 && conv_patch16_applicable(a)) return launch_conv_patch16(a, st);
  if (cfg < 0 && conv_patch_applicable(a)) {
    // patch-tiled 3x3: pick the channel tile so the launch has >= 512 workgroups when it can
    const long sp = (long)route_N(a) * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
    int bn = a.Cout >= 128 ? 128 : (a.Cout > 32 ? 64 : (a.Cout > 16 ? 32 : 16));
    if (bn == 128 && sp * ((a.Cout + 127) / 128) < 512) bn = 64;
    return launch_conv_patch(a, st, bn);
  }
  if (cfg < 0 && conv_s2_dgrad_applicable(a))
  {
    // the four classes carry 1 / 2 / 2 / 4 of a 3x3's taps, so the launch is as long as its four-tap class: when that class alone
    // has fewer 128-wide tiles than CUs (layer3 / layer4 at batch 16: 128 / 64 workgroups, 169 / 233 us), 64 x 64 tiles
    const long t128 = (long)((route_N(a) * (a.Ho >> 1) * (a.Wo >> 1) + 127) / 128) * ((a.Cout + 127) / 128);
    if (a.Cout > 64 && t128 < device_cu_count()) return launch_s2_dgrad<64, 64, 2, 2>(a, st, 4);
    return a.Cout <= 64 ? launch_s2_dgrad<128, 64, 2, 2>(a, st, 1) : launch_s2_dgrad<128, 128, 2, 2>(a, st, 0);
  }
  if (cfg < 0 && conv_gemm_preferred(a)) return launch_conv_gemm(a, st, 0);      // 1x1 / stride 1, Cin % 32 == 0: persistent LDS-DMA GEMM
  if (cfg < 0) {
    const long tiles128 = (long)((route_M(a) + 127) / 128);
    if (a.Cout <= 16) cfg = 3;
    else if (a.Cout <= 32) cfg = 2;
    else if (a.Cout <= 64) cfg = (tiles128 >= 512) ? 1 : 4;
    else {
      const long b0 = tiles128 * ((a.Cout + 127) / 128);
      // 1x1 layers with channel counts like 144 or 192 (MBConv expand / project dgrad): the 128-wide tile pads them to 256;
      // the 64-wide tile wastes far fewer MFMAs and LDS reads
      static const bool narrow = !dbg_flag("UWM_NO_NARROW_1X1");
      const int pad128 = ((a.Cout + 127) / 128) * 128, pad64 = ((a.Cout + 63) / 64) * 64;
      if (narrow && a.ntaps == 1 && pad128 * 100 > pad64 * 115 && tiles128 * (pad64 / 64) >= 512) cfg = 1;
      else if (b0 >= 512) cfg = 0;
      else if (tiles128 * ((a.Cout + 63) / 64) >= 512) cfg = 1;
      else cfg = 4;
    }
  }
  switch (cfg) {
    case 0: return launch_cfg<128, 128, 2, 2>(a, st, 0);
    case 1: return launch_cfg<128, 64, 2, 2>(a, st, 1);
    case 2: return launch_cfg<128, 32, 4, 1>(a, st, 2);
    case 3: return launch_cfg<128, 16, 4, 1>(a, st, 3);
    case 4: return launch_cfg<64, 64, 2, 2>(a, st, 4);
    case 5: return launch_cfg<64, 128, 1, 4>(a, st, 5);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace uwm
