// fp16x3 direct 3x3 / stride-1 convolution, second form (round 4): v_mfma_f32_32x32x16_f16, 8 x 32-pixel tiles, 16-channel chunks.
//
// Same arithmetic as conv_f16x3.hip (every fp32 operand split once into hi + lo fp16 halves, a*b = hi*hi' + hi*lo' + lo*hi' with
// fp32 accumulation, exact power-of-two range scaling of filter rows and of a dgrad's dY) — what changes is how the work meets
// the hardware.  The counters of the first form (profiles/r03_t_pmc_conv_f16x3.txt) named three losses:
//   * LDS bank conflicts on half of all fragment-read cycles: a ds_read_b128 is served in four NON-contiguous 16-lane groups
//     ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS), and the 80-byte pixel pitch of a 16-pixel-wide fragment put two lanes of
//     every group on one 16-byte slot.  Here a fragment is 32 CONSECUTIVE pixels of one image row x 8 channels, stored as planes
//     [channel octet][hi | lo][pixel][16 bytes]: the 16 lanes of any group read 16 different consecutive slots — conflict-free for
//     every tap shift and any row pitch;
//   * 10 % of the MFMAs multiplied zeros (a 32-deep k-step = 2 taps x 16 channels, 9 taps -> 10 slots).  The 32x32x16 instruction
//     is 16 deep: one tap x 16 channels per k-step, 9 exact k-steps per chunk;
//   * the VALU work of staging (lazy BatchNorm + ReLU, clamp, hi / lo split: 1.9-2.9 instructions per MFMA) competed with the
//     MFMAs for the SIMD's issue port: a 16x16x32 MFMA holds the port for 8 of its 16 cycles, a 32x32x16 MFMA for 8 of its 32 —
//     three times the issue room per FLOP — and the split itself drops from ~10 to 3.5 instructions per element:
//     one v_med3 does ReLU + range clamp + halo zeroing, v_cvt_pk_f16_f32 rounds two values at once, and the low halves come from
//     v_fma_mixlo/mixhi_f16 (f16(x - hi) in ONE instruction, reading hi straight out of the packed register).
//
// Work split: workgroup = 8 x 32 output pixels x 64 (NCF = 2) or 32 (NCF = 1) output channels, 4 waves; wave w = pixel rows
// 2w, 2w+1 (two 32-pixel B fragments) x NCF 32-channel A fragments = 2 NCF accumulator tiles of 32 x 32 (64 VGPRs at NCF = 2).
// Filter fragments come from global memory / L1 in MFMA lane order, one tap ahead (f16x3v2 bank: [chunk][tap][fragment][hi | lo]
// [64 lanes][8 halfs]); the 10 x 34-pixel halo patch of a chunk is staged through registers (double-buffered LDS, one barrier per
// chunk), its conversion rounds riding between the MFMAs of the previous chunk's taps.  Epilogue contract = conv_f16x3_kernel's.
//
// Reference semantics replaced: the 3x3 convolutions of smp.Unet's encoder / decoder forward and their input gradients
// (/root/reference/src/models/unet_model.py:64-71 -> SURVEY.md §8 a5-a8, a10, a14).
#include "uwm_kernels.h"
#include <type_traits>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#ifndef UWM_F16V2_ABL
#define UWM_F16V2_ABL 0       // compile-time timing ablations: 1 no MFMA, 2 no filter loads, 4 no pixel-fragment LDS reads, 8 no patch loads / stores, (8-wave kernel) 16 no global patch loads, 32 no patch conversion / LDS stores, 64 no filter DMA (LDS reads stay); 0 in the product build
#endif
#ifndef UWM_V2_COSLOW_BYTES
#define UWM_V2_COSLOW_BYTES (2u << 20)      // filter bytes of all channel tiles beyond which the tile walk keeps one channel tile per XCD
#endif
#ifndef UWM_V2_AD
#define UWM_V2_AD 3           // filter-fragment register sets of the 4-wave kernel (prefetch distance UWM_V2_AD - 1 taps; 3 over 2: layer3 69 -> 57 us, decoder block 0 conv1 171 -> 152)
#endif
constexpr int kVH = 8, kVW = 32;                       // output tile
constexpr int kVPW = kVW + 2, kVPH = kVH + 2, kVPP = kVPW * kVPH;      // 34 x 10 = 340 patch pixels
constexpr int kVPlane = kVPP * 16;                     // bytes per plane: [pixel][8 halfs]
constexpr int kVBuf = 4 * kVPlane;                     // [octet 0 hi][octet 1 hi][octet 0 lo][octet 1 lo] = 21 760 bytes
constexpr int kVUnits = kVPP * 4;                      // 16-byte fp32 units of a chunk (pixel x channel quad): 1360
constexpr int kVRounds = (kVUnits + 255) / 256;        // 6 (the last one: 80 units)
template <int NCF> struct VQ { static constexpr int kLd = 32 * NCF + 4; };      // epilogue block: floats per pixel (32 or 64 + 4 pad)

// ---------------------------------------------------------------- filter bank (layout 1 of f16x3_weights_multi: WinoJob::pad_ == 1)
// bank = [C/16 chunks][9 taps][nF 32-row fragments][2 planes][64 lanes][8 halfs]; lane l of a fragment holds row (l & 31),
// channels 8 (l >> 5) .. +7 of the chunk — the A operand of v_mfma_f32_32x32x16_f16.  rinv[] sits where layout 0 keeps it
// (f16x3_rinv_off), so the two layouts share one workspace slot (this one is 10 % smaller).
__host__ __device__ inline int f16x3v2_nf_(int rows) { return ((rows + 63) / 64) * 2; }
int f16x3v2_nf(int rows) { return f16x3v2_nf_(rows); }

// shapes the second form takes: whole 8 x 32 tiles, 32-channel output fragments (the launcher and whoever packs the bank agree
// through ConvArgs::wu_layout, which carries this function's verdict)
bool f16x3v2_shape(int Ho, int Wo, int rows, int chans, int dgrad) {
  static const bool off = dbg_flag("UWM_F16X3_V1");
  static const int mode = dbg_int("UWM_V2_MODE", 1);      // experiments: 0 never | 1 32-row-fragment layers only | 2 + forward convolutions | 3 wherever the shape allows
  if (off || mode == 0 || (Wo % kVW) != 0 || (Ho % kVH) != 0 || (chans & 31) != 0 || rows < 32 || (rows & 31) != 0) return false;
  if (mode == 1) return (rows & 63) != 0;
  if (mode == 2) return (rows & 63) != 0 || !dgrad;
  return true;
}

__global__ __launch_bounds__(256) void f16x3v2_weights_multi_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  if (jb.pad_ != 1) return;
  const int nF = f16x3v2_nf_(jb.rows);
  const size_t total = (size_t)(jb.chans / 16) * 9 * nF * 64;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int f = (int)((i >> 6) % nF);
  const int t = (int)((i >> 6) / nF);                 // chunk * 9 + tap
  const int tap = t % 9, chunk = t / 9;
  const int row = f * 32 + (lane & 31), ch0 = chunk * 16 + 8 * (lane >> 5);
  const float* rinv = jb.ut + (size_t)(jb.chans / 16) * 5 * (size_t)(((jb.rows + 63) / 64) * 4) * 512;      // = f16x3_rinv_off(rows, chans)
  const float s = 1.f / rinv[row];
  h8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ch0 + e;
    float v = 0.f;
    if (row < jb.rows && ch < jb.chans)
      v = jb.mode == 0 ? jb.w[(size_t)row * jb.Kpad + (size_t)tap * jb.chans + ch]
                       : (ch < jb.src_rows ? jb.w[(size_t)ch * jb.Kpad + (size_t)(8 - tap) * jb.rows + row] : 0.f);
    v *= s;
    const _Float16 h = (_Float16)v;
    hi[e] = h; lo[e] = (_Float16)(v - (float)h);
  }
  _Float16* bank = (_Float16*)jb.ut;
  h8* dst = (h8*)(bank + ((size_t)t * nF + f) * 1024 + lane * 8);
  dst[0] = hi;
  dst[64] = lo;                                       // plane 1: +512 halfs
}
hipError_t launch_f16x3v2_weights_multi(const WinoJobs& jobs, hipStream_t st) {      // (after the row scales of launch_f16x3_weights_multi: same rinv array)
  size_t mx = 0;
  for (int i = 0; i < jobs.n; ++i) {
    if (jobs.j[i].pad_ != 1) continue;
    const size_t t = (size_t)(jobs.j[i].chans / 16) * 9 * f16x3v2_nf(jobs.j[i].rows) * 64;
    if (t > mx) mx = t;
  }
  if (mx == 0) return hipSuccess;
  hipLaunchKernelGGL(f16x3v2_weights_multi_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

// ---------------------------------------------------------------- split of four fp32 values into hi / lo fp16 halves
// hi = rn_f16(x) (v_cvt_pk_f16_f32, two values per instruction), lo = rn_f16(x - hi) as ONE v_fma_mix{lo,hi}_f16 per value:
// fma(hi_as_f32, -1.0, x) rounded to fp16 into one half of the destination, the fp16 operand selected out of the packed register
// by op_sel (lane-exact: the difference of an fp32 value and its fp16 rounding is representable in fp32).
__device__ __forceinline__ void split4(f4 x, u2& hi, u2& lo) {
  const h2 a = {(_Float16)x.x, (_Float16)x.y}, b = {(_Float16)x.z, (_Float16)x.w};
  hi.x = __builtin_bit_cast(unsigned, a); hi.y = __builtin_bit_cast(unsigned, b);
  unsigned l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0) : "v"(hi.x), "v"(x.x), "v"(x.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l1) : "v"(hi.y), "v"(x.z), "v"(x.w));
  lo.x = l0; lo.y = l1;
}

// ---------------------------------------------------------------- epilogue (shared by the 4-wave and the 8-wave kernel)
// R = one MMA wave's block [64 pixels = 2 image rows x 32][32 NCF + 4 floats] (raw accumulators); this wave finishes pixels
// [p0, p0 + np) of it with lanes along the channels (whole 128- / 256-byte pixel rows per store): row un-scale, bias, residual
// addend, ReLU mask, BatchNorm statistics / fused BatchNorm-backward sums — conv_f16x3_kernel's contract.  hb = image row of the
// block's first row.  The concat-split form (ConvArgs::out_up, NCF = 2 only) sums 2 x 2 pixel blocks and needs np = 64.
template <int NCF>
__device__ __forceinline__ void v2_epilogue(const ConvArgs& a, const float* R, int p0, int np, int n, int hb, int w0, int n0, int lane,
                                            float ixs, f4& ps_, f4& pq_) {
  constexpr int kCo = 32 * NCF, kCQ = kCo / 4, kSub = 64 / kCQ;
  const float* rinv = (const float*)a.wu + a.wu_rinv_off;
  const bool bnb = a.bnb_mean != nullptr;
  const int cq = lane & (kCQ - 1), sub = lane / kCQ;
  const int co = n0 + cq * 4;
  const bool cok = co < a.Cout;
  if (NCF == 2 && a.out_up != nullptr) {
    f4 rs = {0.f, 0.f, 0.f, 0.f};
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    if (cok) {
      if (n0 < a.up_c0) {
        f4 bmu = {0.f, 0.f, 0.f, 0.f}, brs = bmu, msc = {1.f, 1.f, 1.f, 1.f}, msh = bmu;
        if (bnb) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
        if (a.up_mscale) { msc = *(const f4*)(a.up_mscale + co); msh = *(const f4*)(a.up_mshift + co); }
#pragma unroll
        for (int r = 0; r < 16 / kSub; ++r) {
          const int bx = r * kSub + sub;                   // block column 0..15
          const float* q = R + (2 * bx) * VQ<NCF>::kLd + cq * 4;
          f4 v = (*(const f4*)q + *(const f4*)(q + VQ<NCF>::kLd) + *(const f4*)(q + 32 * VQ<NCF>::kLd) + *(const f4*)(q + 33 * VQ<NCF>::kLd)) * rs;
          const size_t o2 = (((size_t)n * (a.Ho >> 1) + (hb >> 1)) * (a.Wo >> 1) + ((w0 >> 1) + bx)) * a.up_c0 + co;
          if (a.up_mask) {
            f4 mk = *(const f4*)(a.up_mask + o2);
            const f4 yr = mk;
            if (a.up_mscale) mk = mk * msc + msh;
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
            v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            if (bnb) { ps_ += v; pq_ += v * ((yr - bmu) * brs); }
          }
          if (a.up_accum) v += *(const f4*)(a.out_up + o2);
          *(f4*)(a.out_up + o2) = v;
        }
      } else {
        const int c1n = a.Cout - a.up_c0;
#pragma unroll 4
        for (int r = 0; r < 64 / kSub; ++r) {
          const int p = r * kSub + sub;
          const int ho = hb + (p >> 5), wo = w0 + (p & 31);
          *(f4*)(a.out + (((size_t)n * a.Ho + ho) * a.Wo + wo) * c1n + (co - a.up_c0)) = *(const f4*)(R + p * VQ<NCF>::kLd + cq * 4) * rs;
        }
      }
    }
    return;
  }
  f4 rs = {0.f, 0.f, 0.f, 0.f}, bmu = rs, brs = rs, bia = rs, msc = {1.f, 1.f, 1.f, 1.f}, msh = rs;
  if (cok) rs = *(const f4*)(rinv + co) * ixs;
  if (bnb && cok) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
  if (a.bias && cok) bia = *(const f4*)(a.bias + co);
  if (a.mscale && cok) { msc = *(const f4*)(a.mscale + co); msh = *(const f4*)(a.mshift + co); }
#pragma unroll 4
  for (int r = 0; r < np / kSub; ++r) {
    const int p = p0 + r * kSub + sub;
    const int ho = hb + (p >> 5), wo = w0 + (p & 31);
    if (cok) {
      const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
      f4 v = *(const f4*)(R + p * VQ<NCF>::kLd + cq * 4) * rs + bia;
      if (a.addend) v += *(const f4*)(a.addend + o);
      f4 yr = {0.f, 0.f, 0.f, 0.f};
      if (a.mask) {
        f4 mk = *(const f4*)(a.mask + o);
        yr = mk;
        if (a.mscale) mk = mk * msc + msh;
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      }
      *(f4*)(a.out + o) = v;
      if (a.bnb_y) yr = *(const f4*)(a.bnb_y + o);
      ps_ += v; pq_ += bnb ? v * ((yr - bmu) * brs) : v * v;
    }
  }
}
// per-channel (sum, sum of squares | BatchNorm-backward sums) of the workgroup: pixel sub-rows of a wave by shuffles, the NW waves
// through LDS (red: dead LDS, [NW][64][2] floats), one fp64 atomic pair per channel on one of the replicas
template <int NCF, int NW>
__device__ __forceinline__ void v2_stats(const ConvArgs& a, float* red, f4 ps_, f4 pq_, int n0, int tid, int lane, int wave) {
  constexpr int kCo = 32 * NCF, kCQ = kCo / 4;
  const int cq = lane & (kCQ - 1), sub = lane / kCQ;
  const int stat_c = a.out_up != nullptr ? a.up_c0 : a.Cout;       // channels the statistics cover
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float sv = ps_[e], qv = pq_[e];
#pragma unroll
    for (int d = kCQ; d < 64; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
    ps_[e] = sv; pq_[e] = qv;
  }
  __syncthreads();                             // every wave is done with the blocks
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[(wave * 64 + cq * 4 + e) * 2] = ps_[e]; red[(wave * 64 + cq * 4 + e) * 2 + 1] = pq_[e]; }
  }
  __syncthreads();
  if (tid < kCo) {
    const int c1 = n0 + tid;
    if (c1 < stat_c) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sv += (double)red[(w * 64 + tid) * 2]; qv += (double)red[(w * 64 + tid) * 2 + 1]; }
      const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
      atomicAdd(a.ssum + srep_off + c1, sv);
      atomicAdd(a.ssq + srep_off + c1, qv);
    }
  }
}

// ---------------------------------------------------------------- main kernel
// NP = split products per tile (ConvArgs::nprod): 3 = hi*hi' + hi*lo' + lo*hi'; 2 = without the pixel operand's low half; 1 = hi*hi' only
template <int NCF, int NP = 3>
__global__ __launch_bounds__(256, 2) void conv_f16x3v2_kernel(const ConvArgs a) {
  constexpr int kCo = 32 * NCF;                        // output channels per workgroup
  constexpr int kCQ = kCo / 4, kSub = 64 / kCQ;        // epilogue: lanes along the channel quads x pixel sub-rows
  constexpr int dbg = UWM_F16V2_ABL;
  extern __shared__ __attribute__((aligned(16))) char vsm[];      // [2][kVBuf] patch buffers, then the second source's offsets

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pcol = lane & 31, kh = lane >> 5;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + kCo - 1) / kCo;
  const int tilesW = a.Wo / kVW, tilesH = a.Ho / kVH;
  // tile order inside an XCD's contiguous range: channel tile fastest (the channel tiles of one pixel tile share the patch in L2)
  // while all their filter slices fit the XCD's 4-MB L2 beside the streams; else channel tile SLOWEST — an XCD then works on one
  // filter slice (Ctot x 2304 bytes per 64 channels), which stays L2-resident, and streams the patches
  int tn, tw, th, n;
  if ((size_t)tilesN * a.Ctot * (36 * kCo) > (size_t)UWM_V2_COSLOW_BYTES) {
    const int pixt = tilesW * tilesH * a.N;
    tn = tile / pixt; tile -= tn * pixt;
    tw = tile % tilesW; tile /= tilesW;
    th = tile % tilesH; n = tile / tilesH;
  } else {
    tn = tile % tilesN; tile /= tilesN;
    tw = tile % tilesW; tile /= tilesW;
    th = tile % tilesH; n = tile / tilesH;
  }
  const int n0 = tn * kCo, h0 = th * kVH, w0 = tw * kVW;
  const int nF = a.wu_ncb;

  f16v acc[2][NCF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCF; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- patch staging through registers: unit u = rd * 256 + tid -> pixel u >> 2 of the 10 x 34 patch, channel quad u & 3 (= tid & 3).
  // A round's source offset is chunk-invariant; the second source's offsets (a concat whose halves differ in size when one of them
  // is up-sampled) wait in LDS until the chunk walk crosses the boundary
  int goff[kVRounds];
  unsigned gflags = 0;
  int* const goff_s1 = (int*)(vsm + 2 * kVBuf);
  const bool has_x = tid < kVUnits - (kVRounds - 1) * 256;      // the last round: 80 units
#pragma unroll
  for (int rd = 0; rd < kVRounds; ++rd) {
    const int u = rd * 256 + tid;
    const bool act = u < kVUnits;
    const int pp = act ? (u >> 2) : 0;
    const int py = pp / kVPW, pxx = pp - py * kVPW;
    const int hl = h0 - 1 + py, wl = w0 - 1 + pxx;
    const bool ok = act && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
    const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
    goff[rd] = ((n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C;
    if (a.C0 < a.Ctot) goff_s1[rd * 256 + tid] = ((n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C;      // (read back by this thread only)
    gflags |= (ok ? 1u : 0u) << rd;
  }
  // dgrad: dY is staged times the power of two that puts max|dY| into [2^13, 2^14) (exact; undone in the epilogue)
  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }
  f4 pv[kVRounds];
  f4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  float pfloor = -65504.f;                               // ReLU + range clamp in one v_med3: [0 | -65504, 65504]; a halo pixel outside the image: [0, 0]
  const float* lsp = nullptr;                            // source pointer of the chunk whose loads are being issued
  auto chunk_src = [&](int cc) {
    const int c = cc * 16;
    const bool first = c < a.C0;
    if (c == a.C0 && a.C0 < a.Ctot) {                    // the walk crosses the concat boundary (chunks are visited in order)
#pragma unroll
      for (int rd = 0; rd < kVRounds; ++rd) goff[rd] = goff_s1[rd * 256 + tid];
    }
    lsp = (first ? a.s0.ptr : a.s1.ptr) + (first ? c : c - a.C0) + (tid & 3) * 4;
  };
  auto coef_load = [&](int cc) {                         // lazy transform of the chunk being STORED, range scale folded in (xs > 0 commutes with the ReLU)
    const int c = cc * 16;
    const bool first = c < a.C0;
    const Src& s = first ? a.s0 : a.s1;
    const int cl = (first ? c : c - a.C0) + (tid & 3) * 4;
    if (s.scale != nullptr) { psc = *(const f4*)(s.scale + cl) * xs; psh = *(const f4*)(s.shift + cl) * xs; pfloor = s.relu ? 0.f : -65504.f; }
    else { psc = (f4){xs, xs, xs, xs}; psh = (f4){0.f, 0.f, 0.f, 0.f}; pfloor = -65504.f; }
  };
  auto store_unit = [&](int buf, int rd, f4 raw) {
    const bool ok = (gflags >> rd) & 1u;
    const float lo_c = ok ? pfloor : 0.f, hi_c = ok ? 65504.f : 0.f;
    f4 v;
    v.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.x, psc.x, psh.x), lo_c, hi_c);
    v.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.y, psc.y, psh.y), lo_c, hi_c);
    v.z = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.z, psc.z, psh.z), lo_c, hi_c);
    v.w = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.w, psc.w, psh.w), lo_c, hi_c);
    u2 hi, lo;
    split4(v, hi, lo);
    const int u = rd * 256 + tid;
    char* d = vsm + buf * kVBuf + ((u >> 1) & 1) * kVPlane + (u >> 2) * 16 + (u & 1) * 8;
    *(u2*)d = hi;
    *(u2*)(d + 2 * kVPlane) = lo;
  };

  // ---- filter fragments: global -> registers, one tap ahead.  step t = chunk * 9 + tap
  const _Float16* const wb = (const _Float16*)a.wu + (size_t)(n0 / 32) * 1024 + lane * 8;
  const int nchunk = a.Ctot >> 4, nsteps = nchunk * 9;
  auto w_load = [&](int t, h8 (&whi)[NCF], h8 (&wlo)[NCF]) {
    if ((dbg & 2) && t > 0) return;
    const _Float16* p = wb + (size_t)t * nF * 1024;
#pragma unroll
    for (int j = 0; j < NCF; ++j) { whi[j] = *(const h8*)(p + j * 1024); wlo[j] = *(const h8*)(p + j * 1024 + 512); }
  };
  // ---- pixel fragments: lane (pixel column pcol, channel octet kh) of row 2 * wave + i, tap (dy, dx): one 16-byte read per plane
  const int pbase = kh * kVPlane + ((2 * wave) * kVPW + pcol) * 16;
  auto x_load = [&](int tap, const char* pc, h8 (&xh)[2], h8 (&xl)[2]) {
    const int off = ((tap / 3) * kVPW + tap % 3) * 16;
    const char* pp = pc + pbase + off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (dbg & 4) { xh[i] = (h8){1, 1, 1, 1, 1, 1, 1, 1}; xl[i] = xh[i]; continue; }
      xh[i] = *(const h8*)(pp + i * kVPW * 16); xl[i] = *(const h8*)(pp + i * kVPW * 16 + 2 * kVPlane);
    }
  };
  auto mma = [&](const h8 (&whi)[NCF], const h8 (&wlo)[NCF], const h8 (&xh)[2], const h8 (&xl)[2]) {
    if (dbg & 1) { acc[0][0][0] += (float)xh[0][0] + (float)xl[1][1] + (float)whi[0][2] + (float)wlo[NCF - 1][3] + (float)xh[1][0]; return; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[j], xh[i], acc[i][j], 0, 0, 0);
    if (NP >= 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
    }
    if (NP >= 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
    }
  };

  // filter-fragment ring: kAD register sets, the loads of tap t + kAD - 1 are issued while tap t is multiplied (9 taps per chunk:
  // set = tap % kAD with kAD = 3 divides evenly, so one chunk per loop trip)
  constexpr int kAD = UWM_V2_AD;
  h8 w_hi[kAD][NCF], w_lo[kAD][NCF];
  h8 x_h[2][2], x_l[2][2];
#pragma unroll
  for (int d = 0; d < kAD - 1; ++d) w_load(d < nsteps ? d : nsteps - 1, w_hi[d], w_lo[d]);
  chunk_src(0);
#pragma unroll
  for (int rd = 0; rd < kVRounds - 1; ++rd) pv[rd] = *(const f4*)(lsp + goff[rd]);
  if (has_x) pv[kVRounds - 1] = *(const f4*)(lsp + goff[kVRounds - 1]);
  coef_load(0);
#pragma unroll
  for (int rd = 0; rd < kVRounds - 1; ++rd) store_unit(0, rd, pv[rd]);
  if (has_x) store_unit(0, kVRounds - 1, pv[kVRounds - 1]);
  if (nchunk > 1) {                                      // chunk 1 (it may already be the second source)
    chunk_src(1);
#pragma unroll
    for (int rd = 0; rd < kVRounds - 1; ++rd) pv[rd] = *(const f4*)(lsp + goff[rd]);
    if (has_x) pv[kVRounds - 1] = *(const f4*)(lsp + goff[kVRounds - 1]);
  }
  __syncthreads();

  // Iteration c multiplies chunk c (buffer c & 1) and, between the MFMAs of taps 0-5, converts and stores the six rounds of
  // chunk c+1 (raw values fetched during iteration c-1) into the other buffer, re-issuing each round's global load for chunk c+2.
  // kTrip chunks per loop trip so that every register set index is static (taps per trip % kAD == 0, chunks per trip even).
  constexpr int kTrip = (kAD == 3) ? 2 : 2;              // (9 * 2) % 2 == 0 and (9 * 2) % 3 == 0
  for (int cc = 0; cc < nchunk; cc += kTrip) {
#pragma unroll
    for (int hh = 0; hh < kTrip; ++hh) {
      const int c = cc + hh, cur = hh & 1, nxt = cur ^ 1;
      const bool more1 = c + 1 < nchunk;
      coef_load(more1 ? c + 1 : c);
      chunk_src(c + 2 < nchunk ? c + 2 : nchunk - 1);      // (past the end: a harmless re-fetch)
      const char* pc = vsm + cur * kVBuf;
      x_load(0, pc, x_h[0], x_l[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) {
        const int t = c * 9 + ks;
        const int tp = t + kAD - 1 < nsteps ? t + kAD - 1 : nsteps - 1;
        const int ws = (hh * 9 + ks) % kAD, wl = (hh * 9 + ks + kAD - 1) % kAD, xsn = ks & 1;
        // operands ahead first: filter fragments from L1 / L2 (kAD - 1 taps ahead), pixel fragments from LDS (next tap)
        w_load(tp, w_hi[wl], w_lo[wl]);
        if (ks < 8) x_load(ks + 1, pc, x_h[xsn ^ 1], x_l[xsn ^ 1]);
        mma(w_hi[ws], w_lo[ws], x_h[xsn], x_l[xsn]);
        if (!(dbg & 8) && ks < kVRounds) {
          if (ks < kVRounds - 1) {
            store_unit(nxt, ks, pv[ks]);                   // (past the last chunk: into the dead buffer)
            pv[ks] = *(const f4*)(lsp + goff[ks]);
          } else if (has_x) {
            store_unit(nxt, ks, pv[ks]);
            pv[ks] = *(const f4*)(lsp + goff[ks]);
          }
        }
        // schedule of the tap: operand loads first, then the round's VALU work spread under the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * NCF, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int q = 0; q < 2 * NP * NCF; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue: D[row = channel][col = pixel]; lane (pixel pcol, half kh) holds channels 8q + 4kh .. +3 (q = 0..3) of
  // each 32-channel fragment.  Every wave passes its 64 px x kCo block through LDS ([pixel][68] floats, its own region; the patch
  // buffers are dead) and reads it back with lanes along the channels: whole 128- / 256-byte pixel rows per store instruction
  float* const R = (float*)vsm + wave * 64 * VQ<NCF>::kLd;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCF; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f4*)(R + (i * 32 + pcol) * VQ<NCF>::kLd + j * 32 + q * 8 + kh * 4) = (f4){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  __syncthreads();
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_;
  v2_epilogue<NCF>(a, R, 0, 64, n, h0 + 2 * wave, w0, n0, lane, 1.f / xs, ps_, pq_);
  if (a.ssum != nullptr) v2_stats<NCF, 4>(a, (float*)vsm, ps_, pq_, n0, tid, lane, wave);
}

// ---------------------------------------------------------------- 8-wave kernel: MMA waves and loader waves
// What holds the 4-wave kernel back (compile-time ablations, profiles/r04_e_ablate_v2.txt: the MFMAs alone 85 us on 768 -> 256 at
// 32^2, + pixel-fragment LDS reads 93, + filter loads 103, + patch staging alone 98 — but all of them together 152): a wave's
// vector-memory operations retire IN ORDER, so the filter fragments of the next tap (an L2 hit) cannot be consumed before the HBM
// loads of the patch staging issued ahead of them have landed; six taps out of nine wait for an HBM round trip.  Here ONE
// 512-thread workgroup per CU gives each SIMD an MMA wave and a LOADER wave: waves 0-3 only load filter fragments (kSD taps ahead,
// nothing else on their memory counter), read pixel fragments and issue MFMAs; waves 4-7 only stage patches (global -> registers
// two chunks ahead -> lazy BatchNorm / ReLU / split -> LDS).  One barrier per chunk; all eight waves share the epilogue.
#ifndef UWM_V2_ALDS
#define UWM_V2_ALDS 1         // 8-wave kernel: 1 = the filter fragments of a chunk reach the MMA waves through LDS (loader waves' LDS-DMA, one copy per workgroup); 0 = every MMA wave loads them from L1 / L2 itself
#endif
typedef __attribute__((address_space(3))) void v2_lds_void;
typedef const __attribute__((address_space(1))) void v2_gbl_void;
#ifndef UWM_V2_W128
#define UWM_V2_W128 1         // 8-wave kernel: loader units of 8 channels, 16-byte LDS stores (0: 4 channels, 8-byte stores)
#endif
#ifndef UWM_V2_PD
#define UWM_V2_PD 1           // 8-wave kernel, filters through LDS: fragment prefetch distance of the MMA waves (taps)
#endif
#ifndef UWM_V2_LS
#define UWM_V2_LS 2           // 8-wave kernel: register stages of the loader waves (chunks of load latency budget)
#endif
#ifndef UWM_V2_SD
#define UWM_V2_SD 4           // filter-fragment register sets of the MMA waves (prefetch distance UWM_V2_SD - 1 taps)
#endif
template <int NCF, int NP = 3>
__global__ __launch_bounds__(512, 1) void conv_f16x3v2s_kernel(const ConvArgs a) {
  constexpr int kCo = 32 * NCF;
  constexpr int kSD = UWM_V2_SD;
  constexpr int dbg = UWM_F16V2_ABL;
  constexpr bool kALds = UWM_V2_ALDS != 0;
  constexpr int kAPieces = 9 * NCF * 2;                // 1-KB fragment planes of a chunk's filter block: [tap][fragment][hi | lo][64 lanes][16 B]
  constexpr int kABuf = kAPieces * 1024;               // 36 864 bytes at NCF = 2

  constexpr int kTrip = (kSD == 4 || kSD == 2) ? (kSD == 4 ? 4 : 2) : (kSD == 3 ? 1 : 0);      // chunks per loop trip: (9 * kTrip) % kSD == 0
  static_assert(kTrip > 0 && (9 * kTrip) % kSD == 0, "UWM_V2_SD must be 2, 3 or 4");
  extern __shared__ __attribute__((aligned(16))) char vsm[];      // [2][kVBuf] patch buffers; the epilogue's [4][64][VQ<NCF>::kLd] floats
  char* const asm_ = vsm + 2 * kVBuf;                  // [2][kABuf] behind the patch buffers (kALds)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_mma = wave < 4;
  const int mw = wave & 3;                            // MMA wave index / loader wave index
  const int ltid = tid & 255;                         // thread index inside its role
  const int pcol = lane & 31, kh = lane >> 5;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + kCo - 1) / kCo;
  const int tilesW = a.Wo / kVW, tilesH = a.Ho / kVH;
  // tile order inside an XCD's contiguous range: channel tile fastest (the channel tiles of one pixel tile share the patch in L2)
  // while all their filter slices fit the XCD's 4-MB L2 beside the streams; else channel tile SLOWEST — an XCD then works on one
  // filter slice (Ctot x 2304 bytes per 64 channels), which stays L2-resident, and streams the patches
  int tn, tw, th, n;
  if ((size_t)tilesN * a.Ctot * (36 * kCo) > (size_t)UWM_V2_COSLOW_BYTES) {
    const int pixt = tilesW * tilesH * a.N;
    tn = tile / pixt; tile -= tn * pixt;
    tw = tile % tilesW; tile /= tilesW;
    th = tile % tilesH; n = tile / tilesH;
  } else {
    tn = tile % tilesN; tile /= tilesN;
    tw = tile % tilesW; tile /= tilesW;
    th = tile % tilesH; n = tile / tilesH;
  }
  const int n0 = tn * kCo, h0 = th * kVH, w0 = tw * kVW;
  const int nF = a.wu_ncb;
  const int nchunk = a.Ctot >> 4, nsteps = nchunk * 9;

  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  f16v acc[2][NCF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCF; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (!is_mma) {
    // ================= loader waves =================
#if UWM_V2_W128
    // unit u = rd * 256 + ltid -> pixel u >> 1 of the 10 x 34 patch, channel OCTET u & 1 (= ltid & 1): two 16-byte loads, and the
    // eight hi / lo halfs leave as ONE 16-byte LDS store each (half the store instructions of the quad form: the LDS stores of the
    // loader waves, not their loads or their arithmetic, are what the MMA waves feel — profiles/r04_t_*)
    constexpr int kUnits8 = kVPP * 2, kR8 = (kUnits8 + 255) / 256;      // 680 units, 3 rounds (the last one: 168)
    int goff0[kR8], goff1[kR8];
    unsigned gflags = 0;
    const bool has_x = ltid < kUnits8 - (kR8 - 1) * 256;
#pragma unroll
    for (int rd = 0; rd < kR8; ++rd) {
      const int u = rd * 256 + ltid;
      const bool act = u < kUnits8;
      const int pp = act ? (u >> 1) : 0;
      const int py = pp / kVPW, pxx = pp - py * kVPW;
      const int hl = h0 - 1 + py, wl = w0 - 1 + pxx;
      const bool ok = act && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      goff0[rd] = ((n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C;
      goff1[rd] = a.C0 < a.Ctot ? ((n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C : goff0[rd];
      gflags |= (ok ? 1u : 0u) << rd;
    }
    struct Stage { f4 pv[kR8][2]; f4 sc[2], sh[2]; float floor_; };
    auto patch_load = [&](int cc, Stage& st) {
      if (dbg & 8) return;
      const int c = cc * 16;
      const bool first = c < a.C0;
      const Src& s = first ? a.s0 : a.s1;
      const int cl = (first ? c : c - a.C0) + (ltid & 1) * 8;
      if (s.scale != nullptr) {
        st.sc[0] = *(const f4*)(s.scale + cl) * xs; st.sc[1] = *(const f4*)(s.scale + cl + 4) * xs;
        st.sh[0] = *(const f4*)(s.shift + cl) * xs; st.sh[1] = *(const f4*)(s.shift + cl + 4) * xs; st.floor_ = s.relu ? 0.f : -65504.f;
      } else { st.sc[0] = st.sc[1] = (f4){xs, xs, xs, xs}; st.sh[0] = st.sh[1] = (f4){0.f, 0.f, 0.f, 0.f}; st.floor_ = -65504.f; }
      const float* sp = s.ptr + cl;
#pragma unroll
      for (int rd = 0; rd < kR8; ++rd) {
        const float* q = sp + (first ? goff0[rd] : goff1[rd]);
        st.pv[rd][0] = *(const f4*)q; st.pv[rd][1] = *(const f4*)(q + 4);
      }
    };
    auto patch_store = [&](int buf, const Stage& st) {
      if (dbg & 8) return;
#pragma unroll
      for (int rd = 0; rd < kR8; ++rd) {
        if (rd == kR8 - 1 && !has_x) break;
        const bool ok = (gflags >> rd) & 1u;
        const float lo_c = ok ? st.floor_ : 0.f, hi_c = ok ? 65504.f : 0.f;
        u2 hi[2], lo[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f4 raw = st.pv[rd][e];
          f4 v;
          v.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.x, st.sc[e].x, st.sh[e].x), lo_c, hi_c);
          v.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.y, st.sc[e].y, st.sh[e].y), lo_c, hi_c);
          v.z = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.z, st.sc[e].z, st.sh[e].z), lo_c, hi_c);
          v.w = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.w, st.sc[e].w, st.sh[e].w), lo_c, hi_c);
          split4(v, hi[e], lo[e]);
        }
        const int u = rd * 256 + ltid;
        char* d = vsm + buf * kVBuf + (u & 1) * kVPlane + (u >> 1) * 16;
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        *(u4*)d = (u4){hi[0].x, hi[0].y, hi[1].x, hi[1].y};
        *(u4*)(d + 2 * kVPlane) = (u4){lo[0].x, lo[0].y, lo[1].x, lo[1].y};
      }
    };
#else
    int goff0[kVRounds], goff1[kVRounds];                // chunk-invariant source offsets of a round, per source of the concat
    unsigned gflags = 0;
    const bool has_x = ltid < kVUnits - (kVRounds - 1) * 256;
#pragma unroll
    for (int rd = 0; rd < kVRounds; ++rd) {
      const int u = rd * 256 + ltid;
      const bool act = u < kVUnits;
      const int pp = act ? (u >> 2) : 0;
      const int py = pp / kVPW, pxx = pp - py * kVPW;
      const int hl = h0 - 1 + py, wl = w0 - 1 + pxx;
      const bool ok = act && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      goff0[rd] = ((n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C;
      goff1[rd] = a.C0 < a.Ctot ? ((n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C : goff0[rd];
      gflags |= (ok ? 1u : 0u) << rd;
    }
    struct Stage { f4 pv[kVRounds]; f4 sc, sh; float floor_; };
    auto patch_load = [&](int cc, Stage& st) {
      if (dbg & 8) return;
      if ((dbg & 16) && cc > 2) return;
      const int c = cc * 16;
      const bool first = c < a.C0;
      const Src& s = first ? a.s0 : a.s1;
      const int cl = (first ? c : c - a.C0) + (ltid & 3) * 4;
      if (s.scale != nullptr) { st.sc = *(const f4*)(s.scale + cl) * xs; st.sh = *(const f4*)(s.shift + cl) * xs; st.floor_ = s.relu ? 0.f : -65504.f; }
      else { st.sc = (f4){xs, xs, xs, xs}; st.sh = (f4){0.f, 0.f, 0.f, 0.f}; st.floor_ = -65504.f; }
      const float* sp = s.ptr + cl;
#pragma unroll
      for (int rd = 0; rd < kVRounds; ++rd) st.pv[rd] = *(const f4*)(sp + (first ? goff0[rd] : goff1[rd]));      // (every wave issues all six: the counted wait below relies on it; the offsets of inactive units are valid duplicates)
    };
    auto patch_store = [&](int buf, const Stage& st) {
      if (dbg & 8) return;
      if ((dbg & 32) && buf >= 0) { asm volatile("" :: "v"(st.pv[0].x), "v"(st.pv[5].w)); return; }
#pragma unroll
      for (int rd = 0; rd < kVRounds; ++rd) {
        if (rd == kVRounds - 1 && !has_x) break;
        const bool ok = (gflags >> rd) & 1u;
        const float lo_c = ok ? st.floor_ : 0.f, hi_c = ok ? 65504.f : 0.f;
        const f4 raw = st.pv[rd];
        f4 v;
        v.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.x, st.sc.x, st.sh.x), lo_c, hi_c);
        v.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.y, st.sc.y, st.sh.y), lo_c, hi_c);
        v.z = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.z, st.sc.z, st.sh.z), lo_c, hi_c);
        v.w = __builtin_amdgcn_fmed3f(__builtin_fmaf(raw.w, st.sc.w, st.sh.w), lo_c, hi_c);
        u2 hi, lo;
        if (dbg & 256) { hi = (u2){__builtin_bit_cast(unsigned, raw.x), __builtin_bit_cast(unsigned, raw.y)}; lo = (u2){__builtin_bit_cast(unsigned, raw.z), __builtin_bit_cast(unsigned, raw.w)}; }
        else split4(v, hi, lo);
        const int u = rd * 256 + ltid;
        char* d = vsm + buf * kVBuf + ((u >> 1) & 1) * kVPlane + (u >> 2) * 16 + (u & 1) * 8;
        if (dbg & 128) { asm volatile("" :: "v"(hi.x), "v"(hi.y), "v"(lo.x), "v"(lo.y)); continue; }
        *(u2*)d = hi;
        *(u2*)(d + 2 * kVPlane) = lo;
      }
    };
#endif
    if (dbg & 512) __builtin_amdgcn_s_setprio(3);
    // kLS register stages: chunk c+1 is stored while chunk c is multiplied, its loads were issued kLS - 1 iterations earlier
    // (UWM_V2_LS = 2: two chunks of latency budget; 3: three)
    constexpr int kLS = UWM_V2_LS;
    Stage st[kLS];
#pragma unroll
    for (int k = 0; k < kLS; ++k) patch_load(k < nchunk ? k : nchunk - 1, st[k]);
    patch_store(0, st[0]);
    patch_load(kLS < nchunk ? kLS : nchunk - 1, st[0]);
    __syncthreads();                                       // chunk 0 staged
    // iteration c: store chunk c+1, load chunk c+1+kLS into the freed registers (plain loads only in these waves: hipcc then waits
    // with counted vmcnt — beside an LDS-DMA it drains the whole queue at every use of a loaded register)
    for (int cc = 0; cc < nchunk; cc += kLS) {
#pragma unroll
      for (int k = 0; k < kLS; ++k) {
        const int c = cc + k;
        if (c < nchunk) {                                  // (uniform: every wave of the workgroup meets the same nchunk barriers)
          Stage& sg = st[(k + 1) % kLS];
          if (c + 1 < nchunk) patch_store((c + 1) & 1, sg);
          patch_load(c + 1 + kLS < nchunk ? c + 1 + kLS : nchunk - 1, sg);
          __syncthreads();
        }
      }
    }
  } else {
    // ================= MMA waves =================
    const _Float16* const wb = (const _Float16*)a.wu + (size_t)(n0 / 32) * 1024 + lane * 8;
    auto w_load = [&](int t, h8 (&whi)[NCF], h8 (&wlo)[NCF]) {
      if ((dbg & 2) && t >= kSD) return;
      const _Float16* p = wb + (size_t)t * nF * 1024;
#pragma unroll
      for (int j = 0; j < NCF; ++j) { whi[j] = *(const h8*)(p + j * 1024); wlo[j] = *(const h8*)(p + j * 1024 + 512); }
    };
    // kALds: filter fragments of chunk cc -> LDS block cc & 1 by LDS-DMA, piece k (1 KB = one fragment plane, lane-linear on both
    // sides) through MMA wave k & 3.  Issued at the START of chunk cc - 1: the MMA waves have no other vector-memory traffic, so the
    // pieces have a whole chunk of MFMAs to land and the vmcnt(0) of the chunk's closing __syncthreads() finds them done
    auto a_dma = [&](int cc) {
      if (!kALds || (dbg & 2) || (dbg & 64)) return;
      char* const dst = asm_ + (cc & 1) * kABuf;
#pragma unroll
      for (int i = 0; i < (kAPieces + 3) / 4; ++i) {
        const int k = i * 4 + mw;                          // (wave-uniform)
        if (k < kAPieces) {
          const int tap = k / (2 * NCF), rest = k % (2 * NCF);      // rest = fragment * 2 + plane
          const _Float16* g = wb + ((size_t)(cc * 9 + tap) * nF) * 1024 + rest * 512;
          __builtin_amdgcn_global_load_lds((v2_gbl_void*)g, (v2_lds_void*)(uintptr_t)(dst + k * 1024), 16, 0, 0);
        }
      }
    };
    auto w_lds = [&](int c, int tap, h8 (&whi)[NCF], h8 (&wlo)[NCF]) {      // kALds: the same fragments out of the chunk's LDS block
      if (dbg & 2) { whi[0] = (h8){1, 1, 1, 1, 1, 1, 1, 1}; wlo[0] = whi[0]; if (NCF > 1) { whi[NCF - 1] = whi[0]; wlo[NCF - 1] = whi[0]; } return; }
      const char* p = asm_ + (c & 1) * kABuf + tap * (2 * NCF) * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < NCF; ++j) { whi[j] = *(const h8*)(p + (2 * j) * 1024); wlo[j] = *(const h8*)(p + (2 * j + 1) * 1024); }
    };
    const int pbase = kh * kVPlane + ((2 * mw) * kVPW + pcol) * 16;
    auto x_load = [&](int tap, const char* pc, h8 (&xh)[2], h8 (&xl)[2]) {
      const int off = ((tap / 3) * kVPW + tap % 3) * 16;
      const char* pp = pc + pbase + off;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (dbg & 4) { xh[i] = (h8){1, 1, 1, 1, 1, 1, 1, 1}; xl[i] = xh[i]; continue; }
        xh[i] = *(const h8*)(pp + i * kVPW * 16); xl[i] = *(const h8*)(pp + i * kVPW * 16 + 2 * kVPlane);
      }
    };
    auto mma = [&](const h8 (&whi)[NCF], const h8 (&wlo)[NCF], const h8 (&xh)[2], const h8 (&xl)[2]) {
      if (dbg & 1) { acc[0][0][0] += (float)xh[0][0] + (float)xl[1][1] + (float)whi[0][2] + (float)wlo[NCF - 1][3] + (float)xh[1][0]; return; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[j], xh[i], acc[i][j], 0, 0, 0);
      if (NP >= 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
      }
      if (NP >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
      }
    };
    constexpr int kPD = UWM_V2_PD;                       // LDS prefetch distance of the MMA waves in taps (1 or 2)
    h8 w_hi[kSD][NCF], w_lo[kSD][NCF];
    h8 x_h[3][2], x_l[3][2];
    if (!kALds) {
#pragma unroll
      for (int d = 0; d < kSD - 1; ++d) w_load(d < nsteps ? d : nsteps - 1, w_hi[d], w_lo[d]);
    } else a_dma(0);
    __syncthreads();                                       // chunk 0 staged (patch and filter block)
    // one chunk of the walk; `base` = (taps done so far) % kSD, a compile-time constant after unrolling
    auto chunk = [&](int c, auto base_c) {
      constexpr int base = decltype(base_c)::value;
      const char* pc = vsm + (c & 1) * kVBuf;
      if (kALds && c + 1 < nchunk) a_dma(c + 1);
      x_load(0, pc, x_h[0], x_l[0]);
      if (kALds) w_lds(c, 0, w_hi[0], w_lo[0]);
      if (kALds && kPD == 2) { x_load(1, pc, x_h[1], x_l[1]); w_lds(c, 1, w_hi[1], w_lo[1]); }
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) {
        const int t = c * 9 + ks;
        const int xsn = ks & 1;
        if (kALds && kPD == 2) {                           // both operands from LDS, TWO taps ahead (three register sets: 9 taps = 3 x 3)
          const int cur = ks % 3, nx2 = (ks + 2) % 3;
          if (ks < 7) { w_lds(c, ks + 2, w_hi[nx2], w_lo[nx2]); x_load(ks + 2, pc, x_h[nx2], x_l[nx2]); }
          __builtin_amdgcn_sched_barrier(0);
          mma(w_hi[cur], w_lo[cur], x_h[cur], x_l[cur]);
        } else if (kALds) {                                // both operands from LDS, one tap ahead
          if (ks < 8) { w_lds(c, ks + 1, w_hi[xsn ^ 1], w_lo[xsn ^ 1]); x_load(ks + 1, pc, x_h[xsn ^ 1], x_l[xsn ^ 1]); }
          __builtin_amdgcn_sched_barrier(0);
          mma(w_hi[xsn], w_lo[xsn], x_h[xsn], x_l[xsn]);
        } else {
          const int tp = t + kSD - 1 < nsteps ? t + kSD - 1 : nsteps - 1;
          const int ws = (base + ks) % kSD, wl = (base + ks + kSD - 1) % kSD;
          w_load(tp, w_hi[wl], w_lo[wl]);
          if (ks < 8) x_load(ks + 1, pc, x_h[xsn ^ 1], x_l[xsn ^ 1]);
          __builtin_amdgcn_sched_barrier(0);               // (operand loads are ISSUED ahead of the MFMA block, not sunk behind it)
          mma(w_hi[ws], w_lo[ws], x_h[xsn], x_l[xsn]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    int c = 0;
    if (kTrip == 4) {
      for (; c + 4 <= nchunk; c += 4) {
        chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, 9 % kSD>{});
        chunk(c + 2, std::integral_constant<int, 18 % kSD>{}); chunk(c + 3, std::integral_constant<int, 27 % kSD>{});
      }
      if (c < nchunk) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, 9 % kSD>{}); }      // (nchunk is even)
    } else if (kTrip == 2) {
      for (; c < nchunk; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, 9 % kSD>{}); }
    } else {
      for (; c < nchunk; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, 0>{}); }
    }
  }

  // ---------------- epilogue (the 4-wave kernel's, spread over eight waves): an MMA wave's 64 px x kCo block goes through its LDS
  // region; waves w and w + 4 finish its first and its second image row (the concat-split form sums 2 x 2 blocks across the two
  // rows: the MMA waves do it alone)
  if (is_mma) {
    float* const Rw = (float*)vsm + mw * 64 * VQ<NCF>::kLd;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f4*)(Rw + (i * 32 + pcol) * VQ<NCF>::kLd + j * 32 + q * 8 + kh * 4) = (f4){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  }
  __syncthreads();
  const float* const R = (const float*)vsm + mw * 64 * VQ<NCF>::kLd;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_;
  if (NCF == 2 && a.out_up != nullptr) { if (is_mma) v2_epilogue<NCF>(a, R, 0, 64, n, h0 + 2 * mw, w0, n0, lane, 1.f / xs, ps_, pq_); }
  else v2_epilogue<NCF>(a, R, (wave >> 2) * 32, 32, n, h0 + 2 * mw, w0, n0, lane, 1.f / xs, ps_, pq_);
  if (a.ssum != nullptr) v2_stats<NCF, 8>(a, (float*)vsm, ps_, pq_, n0, tid, lane, wave);
}

// 3x3 / stride 1 / pad 1 over whole 8 x 32-pixel tiles, 16-channel chunks in PAIRS on either side of a concat (the loop is
// unrolled two chunks per trip), 32-channel output fragments; the fused concat split with the boundary on a 64-channel tile
bool conv_f16x3v2_applicable(const ConvArgs& a) {
  return a.wu != nullptr && a.wu_layout == 1 && a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.rmul == 1 ? a.off == -1 : a.off == 1) &&
         (a.Ctot & 31) == 0 && (a.C0 & 15) == 0 && (a.s0.C & 3) == 0 && (a.s1.C & 3) == 0 && (a.Cout & 31) == 0 && a.Cout >= 32 &&
         a.Hl == a.Ho && a.Wl == a.Wo && (a.Ho % kVH) == 0 && (a.Wo % kVW) == 0 && a.Hl < 32768 && a.Wl < 32768 &&
         (!a.out_up || (((a.Ho | a.Wo) & 1) == 0 && (a.up_c0 & 63) == 0 && a.up_c0 <= a.Cout && (a.Cout & 63) == 0)) &&
         (size_t)a.N * a.s0.H * a.s0.W * a.s0.C < (1ull << 31) && (size_t)a.N * a.s1.H * a.s1.W * a.s1.C < (1ull << 31);
}

template <int NCF, int NP>
static hipError_t launch_v2(const ConvArgs& a, hipStream_t st, unsigned grid, size_t lds) {
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_f16x3v2_kernel<NCF, NP>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3v2_kernel<NCF, NP>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int NCF, int NP>
static hipError_t launch_v2s(const ConvArgs& a, hipStream_t st, unsigned grid, size_t lds) {
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_f16x3v2s_kernel<NCF, NP>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3v2s_kernel<NCF, NP>), dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_conv_f16x3v2(const ConvArgs& a, hipStream_t st, int variant) {      // variant: 0 auto | 4 / 5 = 4-wave kernel, 64- / 32-channel tiles | 6 / 7 = 8-wave kernel, 64- / 32-channel tiles
  if (!conv_f16x3v2_applicable(a)) return hipErrorInvalidValue;
  if (a.out_up && (a.addend || a.mask || a.bias || a.bnb_y || (a.ssum && !a.bnb_mean) || (a.up_c0 < a.Cout && !a.out))) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.out_up ? a.up_mask : (a.bnb_y ? a.bnb_y : a.mask)) || a.up_accum)) return hipErrorInvalidValue;
  const int tiles = a.N * (a.Ho / kVH) * (a.Wo / kVW);
  const size_t main_lds = (size_t)2 * kVBuf + (size_t)kVRounds * 256 * sizeof(int);
  // which kernel (kernel-alone timings at 16 x 512^2, profiles/r04_*): 32-channel tiles -> the 8-wave kernel (128 -> 32 at 256^2: 260 us
  // against 275 on the 4-wave kernel and 394 on conv_f16x3.hip's); 64-channel tiles -> the 4-wave kernel while the launch gives every
  // CU two workgroups (they cover each other's prologue and epilogue: 64 -> 64 at 128^2 71 vs 85 us), the 8-wave kernel below that
  // (256 -> 256 at 32^2: 56 vs 57-69 us)
  int v = variant;
  if (v == 0) {
    const long wgs64 = (long)route_N(a) * (a.Ho / kVH) * (a.Wo / kVW) * ((a.Cout + 63) / 64);
    if ((a.Cout & 63) != 0) v = 7;
    else v = (wgs64 >= 2L * device_cu_count() || a.out_up) ? 4 : 6;
  }
  if ((v == 4 || v == 6) && (a.Cout & 63)) return hipErrorInvalidValue;
  if ((v == 5 || v == 7) && a.out_up) return hipErrorInvalidValue;
  const int ncf = (v == 4 || v == 6) ? 2 : 1;
  // epilogue blocks: 4 waves x 64 pixels x (32 ncf + 4) floats — 36.9 KB at 32-channel tiles, so the 4-wave kernel's 49.6 KB of
  // staging LDS (and its 168 VGPRs) admit THREE workgroups per CU there
  const size_t q_lds = (size_t)4 * 64 * (32 * ncf + 4) * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  const unsigned grid = (unsigned)(tiles * (a.Cout / (32 * ncf)));
  const int np = (a.nprod >= 1 && a.nprod <= 3) ? a.nprod : 3;
  if (v == 6 || v == 7) {
    const size_t s_lds = UWM_V2_ALDS ? (size_t)2 * kVBuf + (size_t)2 * 9 * ncf * 2 * 1024 : 0;      // patch buffers + two filter-fragment blocks
    const size_t lds8 = s_lds > q_lds ? s_lds : q_lds;
    if (v == 7) return np == 3 ? launch_v2s<1, 3>(a, st, grid, lds8) : np == 2 ? launch_v2s<1, 2>(a, st, grid, lds8) : launch_v2s<1, 1>(a, st, grid, lds8);
    return np == 3 ? launch_v2s<2, 3>(a, st, grid, lds8) : np == 2 ? launch_v2s<2, 2>(a, st, grid, lds8) : launch_v2s<2, 1>(a, st, grid, lds8);
  }
  if (v == 5) return np == 3 ? launch_v2<1, 3>(a, st, grid, lds) : np == 2 ? launch_v2<1, 2>(a, st, grid, lds) : launch_v2<1, 1>(a, st, grid, lds);
  if (v != 4) return hipErrorInvalidValue;
  return np == 3 ? launch_v2<2, 3>(a, st, grid, lds) : np == 2 ? launch_v2<2, 2>(a, st, grid, lds) : launch_v2<2, 1>(a, st, grid, lds);
}

}  // namespace uwm
