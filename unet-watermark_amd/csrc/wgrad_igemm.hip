// Weight-gradient implicit GEMM for gfx950 on v_mfma_f32_16x16x4_f32 (exact fp32).
//
//   dW[co][k] = sum_m dY[m][co] * X[m][k],   m = output pixel, k = tap*Ctot + c
//
// The reduction dimension is the pixel index, so both operands are staged in LDS as
// [32 pixels][tile columns] exactly as they lie in NHWC memory (columns contiguous) and the MFMA
// fragments are read with ds_read_b32 (row stride == 16 mod 32 floats: conflict-free).
// X is gathered with the same lazy BatchNorm+ReLU / upsample / concat transform as the forward
// kernel; each thread's k-column (hence tap, channel, source, scale, shift) is fixed for the
// whole pixel loop.  The pixel range is split over `nsplit` workgroups per output tile; every split
// stores its own dW-shaped partial image (plain stores) and wgrad_reduce_kernel (wgrad_wino.hip) adds the
// images in a fixed order — no float atomics: the gradient is bit-reproducible (round 3; the atomics also
// cost 1.76x the algorithmic HBM traffic).  nsplit == 1 adds straight into the pre-zeroed dW.
//
// F16 = true (round 4; WgradArgs::prec == 2 with xmax: the stride-2 layers in the f16x3_all modes): the same walk on
// v_mfma_f32_16x16x32_f16 — a 32-pixel step is ONE k-step, three split products per tile (48 MFMAs of 16 cycles per wave and step at
// 128 x 128 where the fp32 form issues 128 of 32 cycles).  Both operands are split into fp16 hi / lo planes while they are staged
// ([32 px][columns] rows of halfs, 32 bytes of padding per row) and read with the transposing load ds_read_b64_tr_b16; dY is scaled
// by the power of two its maximum calls for (WgradArgs::xmax), which leaves with the partial image.
//
// Replaces the weight-gradient half of autograd's conv2d backward.
#include "uwm_kernels.h"
#include <cstdio>
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fdivw(unsigned n, FastDiv f) {
  return f.d <= 1 ? n : __umulhi(n, f.mg);
}

typedef _Float16 wi_h8 __attribute__((ext_vector_type(8)));
typedef __fp16 wi_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) wi_fp16x4 wi_lds_fp16x4;
__device__ __forceinline__ wi_h8 wi_tr_pair(const char* base, int o0, int o1) {      // two transposed reads -> one 8-half operand fragment
  const wi_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wi_lds_fp16x4*)(uintptr_t)(base + o0));
  const wi_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wi_lds_fp16x4*)(uintptr_t)(base + o1));
  typedef __fp16 fp16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const fp16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(wi_h8, v);
}

template <int TA, int TB, int WA, int WB, bool F16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_igemm_kernel(const WgradArgs a) {
  constexpr int LA = TA + 16, LB = TB + 16;          // LDS row strides (== 16 mod 32)
  constexpr int RSA = TA * 2 + 32, RSB = TB * 2 + 32;      // F16: bytes per pixel row of a half-plane
  constexpr int PLA = 32 * RSA, PLB = 32 * RSB;            // F16: bytes per plane; a stage buffer = [A hi | A lo | B hi | B lo]
  constexpr int STG = 2 * PLA + 2 * PLB;
  constexpr int UA = TA / 4, RA = (256 / UA) > 32 ? 32 : (256 / UA), PA = 32 / RA;
  constexpr int UB = TB / 4, RB = (256 / UB) > 32 ? 32 : (256 / UB), PB = 32 / RB;
  constexpr int MI = TA / WA / 16, NI = TB / WB / 16;
  static_assert(WA * WB == 4 && MI >= 1 && NI >= 1, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;                  // [2][32][LA]
  float* const Bs = smem + 2 * 32 * LA;    // [2][32][LB]
  int* const Rw = F16 ? (int*)((char*)smem + 2 * STG) : (int*)(Bs + 2 * 32 * LB);   // [2][32][4]: per pixel row {n, ho*stride-pad, wo*stride-pad, valid}
  char* const Hs = (char*)smem;            // F16: [2 stages][STG bytes]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesA = (a.wrows + TA - 1) / TA, tilesB = (a.Kpad + TB - 1) / TB;
  const int split = t / (tilesA * tilesB);
  const int tt = t - split * (tilesA * tilesB);
  const int ta = tt / tilesB, tb = tt - ta * tilesB;
  const int a0 = ta * TA, b0 = tb * TB;
  const int mbeg = split * a.msplit;
  const int mend = min(a.M, mbeg + a.msplit);
  const int nsteps = (mend - mbeg + 31) >> 5;

  // ---- fixed per-thread column info
  const int ua = tid % UA, ra = tid / UA;
  const bool a_act = tid < UA * RA;
  const int ub = tid % UB, rb = tid / UB;
  const bool b_act = tid < UB * RB;
  const int co = a0 + ua * 4;
  const bool cov = a_act && co < a.Cout;
  const unsigned kcol = b0 + ub * 4;
  const unsigned tap = fdivw(kcol, a.dv_ctot);
  const int c = kcol - tap * a.Ctot;
  const unsigned r = fdivw(tap, a.dv_kw);
  const int s = tap - r * a.kw;
  const bool tv = b_act && (tap < (unsigned)a.ntaps) && (kcol < (unsigned)a.Kpad);
  const bool first = c < a.C0;
  const float* sp = first ? a.s0.ptr : a.s1.ptr;
  const float* ssc = first ? a.s0.scale : a.s1.scale;
  const float* ssh = first ? a.s0.shift : a.s1.shift;
  const int sC = first ? a.s0.C : a.s1.C, sH = first ? a.s0.H : a.s1.H, sW = first ? a.s0.W : a.s1.W;
  const int sup = first ? a.s0.up : a.s1.up;
  const int trelu = first ? a.s0.relu : a.s1.relu;
  const int cc = first ? c : c - a.C0;
  const bool thas = (ssc != nullptr) && tv;
  f4 tsc = {1.f, 1.f, 1.f, 1.f}, tsh = {0.f, 0.f, 0.f, 0.f};
  if (thas) { tsc = *(const f4*)(ssc + cc); tsh = *(const f4*)(ssh + cc); }
  const int HoWo = a.Ho * a.Wo;

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  f4 ar[PA], br[PB];
  unsigned bvalid = 0;
  float xs = 1.f;                                    // F16: power-of-two scale of dY
  if (F16 && a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  // one thread per pixel row decomposes m -> (n, ho, wo) for a whole step; everyone else reads LDS
  auto row_info = [&](int st) {
    if (tid < 32) {
      const int m = mbeg + st * 32 + tid;
      const bool v = m < mend;
      const int mm = v ? m : 0;
      const int n = mm / HoWo, rem = mm - n * HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      int4 ri = make_int4(n, ho * a.stride - a.pad, wo * a.stride - a.pad, v ? 1 : 0);
      *(int4*)(Rw + ((st & 1) * 32 + tid) * 4) = ri;
    }
  };

  auto load_step = [&](int st) {
    const int mb = mbeg + st * 32;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = mb + ra + RA * i;
      const bool v = cov && m < mend;
      ar[i] = v ? *(const f4*)(a.dy + (size_t)m * a.Cout + co) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    bvalid = 0;
    const int* rw = Rw + (st & 1) * 32 * 4;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int4 ri = *(const int4*)(rw + (rb + RB * i) * 4);
      int hn = ri.y + (int)r, wq = ri.z + s;
      bool v = tv && ri.w && hn >= 0 && hn < a.Hl && wq >= 0 && wq < a.Wl;
      hn >>= sup; wq >>= sup;
      const float* p = sp + ((size_t)((size_t)ri.x * sH + hn) * sW + wq) * sC + cc;
      br[i] = v ? *(const f4*)p : (f4){0.f, 0.f, 0.f, 0.f};
      bvalid |= (v ? 1u : 0u) << i;
    }
  };
  auto store_step = [&](int buf) {
    if (F16) {
      char* const hb = Hs + buf * STG;
      if (a_act) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const f4 v = ar[i] * xs;                     // (below 2^14 by construction: no clamp)
          uwm_u2 hi, lo;
          uwm_split4(v.x, v.y, v.z, v.w, hi, lo);
          char* const p_ = hb + (ra + RA * i) * RSA + ua * 8;
          *(uwm_u2*)p_ = hi; *(uwm_u2*)(p_ + PLA) = lo;
        }
      }
      if (b_act) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          f4 v = br[i];
          if (thas) {
            v = v * tsc + tsh;
            if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!((bvalid >> i) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
          }
          uwm_u2 hi, lo;
          uwm_split4(__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f),
                     __builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f), hi, lo);
          char* const p_ = hb + 2 * PLA + (rb + RB * i) * RSB + ub * 8;
          *(uwm_u2*)p_ = hi; *(uwm_u2*)(p_ + PLB) = lo;
        }
      }
      return;
    }
    float* as = As + buf * 32 * LA;
    float* bs = Bs + buf * 32 * LB;
    if (a_act) {
#pragma unroll
      for (int i = 0; i < PA; ++i) *(f4*)(as + (ra + RA * i) * LA + ua * 4) = ar[i];
    }
    if (b_act) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        f4 v = br[i];
        if (thas) {
          v = v * tsc + tsh;
          if (trelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (!((bvalid >> i) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
        }
        *(f4*)(bs + (rb + RB * i) * LB + ub * 4) = v;
      }
    }
  };

  row_info(0);
  row_info(1);
  __syncthreads();
  if (nsteps > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  const int li = lane & 15, lq = lane >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    load_step(st + 1);                               // reads row table slot (st+1)&1 (rows past mend are masked)
    row_info(st + 2);                                // writes slot st&1 (last read while loading step st)
    if (F16) {
      // lane = (k-group kg, row-in-group q, column quad p): pixel rows 8 kg + q (+4), 8 bytes = 4 consecutive columns
      const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
      const char* const hb = Hs + cur * STG;
      const char* const pa = hb + (8 * kg + q) * RSA + (wa * (TA / WA)) * 2 + p * 8;
      const char* const pb = hb + 2 * PLA + (8 * kg + q) * RSB + (wb * (TB / WB)) * 2 + p * 8;
      wi_h8 ah[MI], al[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) { ah[i] = wi_tr_pair(pa + i * 32, 0, 4 * RSA); al[i] = wi_tr_pair(pa + i * 32 + PLA, 0, 4 * RSA); }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const wi_h8 bh = wi_tr_pair(pb + j * 32, 0, 4 * RSB), bl = wi_tr_pair(pb + j * 32 + PLB, 0, 4 * RSB);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_step(cur ^ 1);
      __syncthreads();
      continue;
    }
    const float* as = As + cur * 32 * LA + wa * (TA / WA) + li;
    const float* bs = Bs + cur * 32 * LB + wb * (TB / WB) + li;
    // fragments are read one k4-step ahead of the MFMAs that use them (register double buffer)
    float af[2][MI], bf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[0][i] = as[lq * LA + i * 16];
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[0][j] = bs[lq * LB + j * 16];
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const int cb = k4 & 1, nb = cb ^ 1;
      if (k4 + 1 < 8) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[nb][i] = as[((k4 + 1) * 4 + lq) * LA + i * 16];
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[nb][j] = bs[((k4 + 1) * 4 + lq) * LB + j * 16];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb][i], bf[cb][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // consumers of this step's global loads stay below the MFMA block
    store_step(cur ^ 1);
    __syncthreads();
  }

  // D[i = co][j = kcol]: lane reg e -> co = lq*4 + e, kcol = li
  const float ixs = 1.f / xs;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int kc = b0 + wb * (TB / WB) + j * 16 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = a0 + wa * (TA / WA) + i * 16 + lq * 4 + e;
        if (row < a.wrows && kc < a.Kpad) {
          if (a.nsplit > 1) a.part[((size_t)split * a.wrows + row) * a.Kpad + kc] = acc[i][j][e] * ixs;      // this split's partial image
          else a.dw[(size_t)row * a.Kpad + kc] += acc[i][j][e] * ixs;                                       // the only writer of this element
        }
      }
    }
}

template <int TA, int TB, int WA, int WB, bool F16 = false>
static hipError_t launch_w(const WgradArgs& a, hipStream_t st, int cls) {
  const int tilesA = (a.wrows + TA - 1) / TA, tilesB = (a.Kpad + TB - 1) / TB;
  const size_t lds = (F16 ? (size_t)2 * (2 * 32 * (TA * 2 + 32) + 2 * 32 * (TB * 2 + 32)) : (size_t)2 * 32 * (TA + 16 + TB + 16) * sizeof(float)) + 2 * 32 * 4 * sizeof(int);
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)wgrad_igemm_kernel<TA, TB, WA, WB, F16>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (wgrad_igemm_kernel<TA, TB, WA, WB, F16>), dim3((unsigned)(tilesA * tilesB * a.nsplit)), dim3(256), lds, st, a);
  if (a.nsplit > 1) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_wgrad_reduce(a.part, a.nsplit, (size_t)a.wrows * a.Kpad / 4, a.dw, st, a.rq);
  }
  return hipGetLastError();
}

// tile (output channels x k-columns) chosen from the layer's Cout: 16x256, 32x256, 64x128, 128x128
hipError_t launch_wgrad(const WgradArgs& a0, hipStream_t st) {
  static const bool trace = dbg_flag("UWM_TRACE_CONV");
  if (trace)
    fprintf(stderr, "wgrad N=%d Ctot=%d(C0=%d) Cout=%d wrows=%d Ho=%d Wo=%d Hl=%d Wl=%d taps=%d stride=%d wino=%d patch=%d gflop=%.2f\n", a0.N, a0.Ctot,
            a0.C0, a0.Cout, a0.wrows, a0.Ho, a0.Wo, a0.Hl, a0.Wl, a0.ntaps, a0.stride, (int)(wino_mode_of(a0.wino) != 0 && wgrad_wino_applicable(a0)),
            (int)wgrad_patch_applicable(a0), a0.flops * 1e-9);
  // force_igemm: 0 auto (Winograd-domain -> patch -> flattened), 1 flattened implicit GEMM only, 2 no Winograd, 4 = wgrad_gemm.hip wherever applicable
  if ((a0.force_igemm & 0xff) == 4) return launch_wgrad_gemm(a0, st);
  static const bool no_up2 = dbg_flag("UWM_NO_UP2") || dbg_flag("UWM_NO_UP2_WGRAD");
  if (((a0.force_igemm & 0xff) == 0 || (a0.force_igemm & 0xff) == 6) && a0.prec == 2 && a0.xmax && !no_up2 && wgrad_up2_applicable(a0) && wgrad_up2_f16_shape(a0))
    return launch_wgrad_up2(a0, st);      // sub-pixel form of conv-after-upsample on its fp16x3 kernel (2.25x fewer products than the direct form below)
  if (((a0.force_igemm & 0xff) == 0 || (a0.force_igemm & 0xff) == 6) && a0.prec == 2 && wgrad_f16x3_applicable(a0)) return launch_wgrad_f16x3(a0, st);      // fp16x3 direct form
  if ((a0.force_igemm & 0xff) == 6 && a0.prec == 2 && wgrad_stem_applicable(a0)) return launch_wgrad_stem(a0, st);      // (its fp16x3 kernel)
  if ((a0.force_igemm & 0xff) == 6 && a0.prec == 2 && a0.Cout == 16 && wgrad_c16_applicable(a0)) return launch_wgrad_c16(a0, st);      // (likewise)
  if ((a0.force_igemm & 0xff) == 6) return hipErrorInvalidValue;
  if ((a0.force_igemm & 0xff) == 0 && !no_up2 && wgrad_up2_applicable(a0)) return launch_wgrad_up2(a0, st);      // sub-pixel form of conv-after-upsample
  if ((a0.force_igemm & 0xff) == 0 && wgrad_stem_applicable(a0)) return launch_wgrad_stem(a0, st);         // the ResNet stem: compact K = 147
  if ((a0.force_igemm & 0xff) == 0 && wgrad_gemm_preferred(a0)) return launch_wgrad_gemm(a0, st);        // 1x1 / stride 1: persistent LDS-DMA GEMM, deterministic
  if ((a0.force_igemm & 0xff) == 0 && wgrad_c16_applicable(a0)) return launch_wgrad_c16(a0, st);      // 16-channel full-resolution layers, head
  if ((a0.force_igemm & 0xff) == 0 && wino_mode_of(a0.wino) != 0 && wgrad_wino_applicable(a0)) return launch_wgrad_wino(a0, st);
  if ((a0.force_igemm & 0xff) != 1 && (a0.force_igemm & 0xff) != 7 && wgrad_patch_applicable(a0)) return launch_wgrad_patch(a0, st);      // (7 = tests: the flattened implicit GEMM in its fp16x3 form)
  WgradArgs a = a0;
  if (a.M <= 0 || (a.Cout & 3) || (a.Kpad & 31) || (a.Ctot & 3) || (a.C0 & 3)) return hipErrorInvalidValue;
  int TA, TB;
  // 1x1 layers (MBConv expand / project, Bottleneck): the k extent is just the input channel count (Kpad 32..2688), so a
  // narrower k tile is taken when it cuts the padded columns by >= 15 % (a 128-column tile on Kpad = 32 spends 3/4 of its
  // MFMAs and LDS traffic on zeros)
  static const bool narrow = !dbg_flag("UWM_NO_NARROW_WGRAD");
  auto padded = [&](int tb) { return (a.Kpad + tb - 1) / tb * tb; };
  if (a.wrows <= 16) { TA = 16; TB = 256; }
  else if (a.wrows <= 32) {
    TA = 32; TB = 256;
    if (narrow && a.ntaps == 1) {
      if (padded(128) * 100 <= padded(TB) * 85) TB = 128;
      if (padded(64) * 100 <= padded(TB) * 85) TB = 64;
    }
  }
  else if (a.wrows <= 64) { TA = 64; TB = 128; }
  else {
    TA = 128; TB = 128;
    if (narrow && a.ntaps == 1) {
      if (padded(64) * 100 <= padded(TB) * 85) TB = 64;
      if (a.Kpad <= 96 && padded(32) * 100 <= padded(TB) * 85) TB = 32;
    }
  }
  const int tiles = ((a.wrows + TA - 1) / TA) * ((a.Kpad + TB - 1) / TB);
  // aim for ~1024 workgroups, at least 256 pixels (8 steps) per split; every split owns a partial image in the scratch
  int nsplit = (1024 + tiles - 1) / tiles;
  int maxsplit = (a.M + 255) / 256;
  if (nsplit > maxsplit) nsplit = maxsplit;
  const size_t image = (size_t)a.wrows * a.Kpad;
  if (nsplit > 1 && (!a.part || a.part_floats < 2 * image)) { a.part = wgrad_op_scratch(); a.part_floats = wgrad_wino_scratch_floats(); a.rq = nullptr; }      // single-operator entry points
  if (nsplit > 1 && (!a.part || (size_t)nsplit * image > a.part_floats)) nsplit = a.part ? (int)(a.part_floats / image) : 1;
  if (nsplit < 1) nsplit = 1;
  int msplit = (a.M + nsplit - 1) / nsplit;
  msplit = (msplit + 31) & ~31;
  nsplit = (a.M + msplit - 1) / msplit;
  a.nsplit = nsplit; a.msplit = msplit;
  // fp16x3 form (the stride-2 layers in the f16x3_all modes; channel counts in whole 32s): 128- and 64-row tiles
  static const bool no_f16 = dbg_flag("UWM_NO_WGRAD_IG16");
  const bool f16 = !no_f16 && a.prec == 2 && a.xmax && ((a.force_igemm & 0xff) == 0 || (a.force_igemm & 0xff) == 7) && (a.Ctot & 31) == 0 && (a.Cout & 31) == 0;
  if (f16 && TA == 128 && TB == 128) return launch_w<128, 128, 2, 2, true>(a, st, 60);
  if (f16 && TA == 128 && TB == 64) return launch_w<128, 64, 2, 2, true>(a, st, 61);
  if (f16 && TA == 64 && TB == 128) return launch_w<64, 128, 2, 2, true>(a, st, 62);
  switch (TA) {
    case 16: return launch_w<16, 256, 1, 4>(a, st, 8);
    case 32:
      if (TB == 64) return launch_w<32, 64, 2, 2>(a, st, 28);
      if (TB == 128) return launch_w<32, 128, 1, 4>(a, st, 29);
      return launch_w<32, 256, 1, 4>(a, st, 9);
    case 64: return launch_w<64, 128, 2, 2>(a, st, 6);
    default:
      if (TB == 32) return launch_w<128, 32, 2, 2>(a, st, 26);
      if (TB == 64) return launch_w<128, 64, 2, 2>(a, st, 27);
      return launch_w<128, 128, 2, 2>(a, st, 7);
  }
}

}  // namespace uwm
