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

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#ifndef UWM_F16V2_ABL
#define UWM_F16V2_ABL 0       // compile-time timing ablations: 1 no MFMA, 2 no filter loads, 4 no pixel-fragment LDS reads, 8 no patch loads / stores; 0 in the product build
#endif
constexpr int kVH = 8, kVW = 32;                       // output tile
constexpr int kVPW = kVW + 2, kVPH = kVH + 2, kVPP = kVPW * kVPH;      // 34 x 10 = 340 patch pixels
constexpr int kVPlane = kVPP * 16;                     // bytes per plane: [pixel][8 halfs]
constexpr int kVBuf = 4 * kVPlane;                     // [octet 0 hi][octet 1 hi][octet 0 lo][octet 1 lo] = 21 760 bytes
constexpr int kVUnits = kVPP * 4;                      // 16-byte fp32 units of a chunk (pixel x channel quad): 1360
constexpr int kVRounds = (kVUnits + 255) / 256;        // 6 (the last one: 80 units)
constexpr int kVQLd = 68;                              // epilogue block: floats per pixel (64 + 4 pad)

// ---------------------------------------------------------------- filter bank (layout 1 of f16x3_weights_multi: WinoJob::pad_ == 1)
// bank = [C/16 chunks][9 taps][nF 32-row fragments][2 planes][64 lanes][8 halfs]; lane l of a fragment holds row (l & 31),
// channels 8 (l >> 5) .. +7 of the chunk — the A operand of v_mfma_f32_32x32x16_f16.  rinv[] sits where layout 0 keeps it
// (f16x3_rinv_off), so the two layouts share one workspace slot (this one is 10 % smaller).
__host__ __device__ inline int f16x3v2_nf_(int rows) { return ((rows + 63) / 64) * 2; }
int f16x3v2_nf(int rows) { return f16x3v2_nf_(rows); }

// shapes the second form takes: whole 8 x 32 tiles, 32-channel output fragments (the launcher and whoever packs the bank agree
// through ConvArgs::wu_layout, which carries this function's verdict)
bool f16x3v2_shape(int Ho, int Wo, int rows, int chans) {
  static const bool off = dbg_flag("UWM_F16X3_V1");
  return !off && (Wo % kVW) == 0 && (Ho % kVH) == 0 && (chans & 31) == 0 && rows >= 32 && (rows & 31) == 0;
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

// ---------------------------------------------------------------- main kernel
template <int NCF>
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
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NCF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
  };

  h8 wA_hi[NCF], wA_lo[NCF], wB_hi[NCF], wB_lo[NCF];
  h8 xA_h[2], xA_l[2], xB_h[2], xB_l[2];
  w_load(0, wA_hi, wA_lo);
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
  // Two chunks per loop trip (nchunk is even): register sets alternate statically (9 taps per chunk: odd).
  for (int cc = 0; cc < nchunk; cc += 2) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = cc + hh, cur = hh, nxt = hh ^ 1;
      const bool more1 = c + 1 < nchunk;
      coef_load(more1 ? c + 1 : c);
      chunk_src(c + 2 < nchunk ? c + 2 : nchunk - 1);      // (past the end: a harmless re-fetch)
      const char* pc = vsm + cur * kVBuf;
      x_load(0, pc, xA_h, xA_l);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) {
        const int t = c * 9 + ks;
        const int tnext = t + 1 < nsteps ? t + 1 : t;
        const bool wsel = ((hh * 9 + ks) & 1) == 0, xsel = (ks & 1) == 0;
        // next tap's operands first: filter fragments from L1 / L2, pixel fragments from LDS
        if (wsel) w_load(tnext, wB_hi, wB_lo); else w_load(tnext, wA_hi, wA_lo);
        if (ks < 8) { if (xsel) x_load(ks + 1, pc, xB_h, xB_l); else x_load(ks + 1, pc, xA_h, xA_l); }
        if (wsel) { if (xsel) mma(wA_hi, wA_lo, xA_h, xA_l); else mma(wA_hi, wA_lo, xB_h, xB_l); }
        else { if (xsel) mma(wB_hi, wB_lo, xA_h, xA_l); else mma(wB_hi, wB_lo, xB_h, xB_l); }
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
        for (int q = 0; q < 6 * NCF; ++q) {
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
  float* const R = (float*)vsm + wave * 64 * kVQLd;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NCF; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f4*)(R + (i * 32 + pcol) * kVQLd + j * 32 + q * 8 + kh * 4) = (f4){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
  __syncthreads();
  const float* rinv = (const float*)a.wu + a.wu_rinv_off;
  const float ixs = 1.f / xs;
  const bool do_stats = a.ssum != nullptr;
  const bool bnb = a.bnb_mean != nullptr;
  const int cq = lane & (kCQ - 1), sub = lane / kCQ;
  const int co = n0 + cq * 4;
  const bool cok = co < a.Cout;
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_;
  const int stat_c = a.out_up != nullptr ? a.up_c0 : a.Cout;       // channels the statistics cover
  const int hb = h0 + 2 * wave;                            // image row of this wave's first pixel row
  if (NCF == 2 && a.out_up != nullptr) {
    // concat-split epilogue of a decoder conv1 dgrad (ConvArgs::out_up, conv_wino_kernel's contract): channels [0, up_c0) are summed
    // over each 2 x 2 pixel block (this wave's two rows x 16 column pairs), ReLU-masked by the low-resolution producer, written at
    // half resolution with the fused BatchNorm-backward sums; channels [up_c0, Cout) go to `out` at full resolution
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
          const float* q = R + (2 * bx) * kVQLd + cq * 4;
          f4 v = (*(const f4*)q + *(const f4*)(q + kVQLd) + *(const f4*)(q + 32 * kVQLd) + *(const f4*)(q + 33 * kVQLd)) * rs;
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
          *(f4*)(a.out + (((size_t)n * a.Ho + ho) * a.Wo + wo) * c1n + (co - a.up_c0)) = *(const f4*)(R + p * kVQLd + cq * 4) * rs;
        }
      }
    }
  } else {
    f4 rs = {0.f, 0.f, 0.f, 0.f}, bmu = rs, brs = rs, bia = rs, msc = {1.f, 1.f, 1.f, 1.f}, msh = rs;
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    if (bnb && cok) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
    if (a.bias && cok) bia = *(const f4*)(a.bias + co);
    if (a.mscale && cok) { msc = *(const f4*)(a.mscale + co); msh = *(const f4*)(a.mshift + co); }
#pragma unroll 4
    for (int r = 0; r < 64 / kSub; ++r) {
      const int p = r * kSub + sub;
      const int ho = hb + (p >> 5), wo = w0 + (p & 31);
      if (cok) {
        const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
        f4 v = *(const f4*)(R + p * kVQLd + cq * 4) * rs + bia;
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
  if (do_stats) {
    // the pixel sub-rows of a wave (xor kCQ .. 32) -> 4 waves through LDS -> fp64 atomics on one replica
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
#pragma unroll
      for (int d = kCQ; d < 64; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
      ps_[e] = sv; pq_[e] = qv;
    }
    __syncthreads();                             // every wave is done with its block
    float* red = (float*)vsm;                    // [4 waves][64][2]
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
        for (int w = 0; w < 4; ++w) { sv += (double)red[(w * 64 + tid) * 2]; qv += (double)red[(w * 64 + tid) * 2 + 1]; }
        const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
        atomicAdd(a.ssum + srep_off + c1, sv);
        atomicAdd(a.ssq + srep_off + c1, qv);
      }
    }
  }
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

hipError_t launch_conv_f16x3v2(const ConvArgs& a, hipStream_t st, int variant) {      // variant: 0 auto | 4 = 64-channel tiles | 5 = 32-channel tiles
  if (!conv_f16x3v2_applicable(a)) return hipErrorInvalidValue;
  if (a.out_up && (a.addend || a.mask || a.bias || a.bnb_y || (a.ssum && !a.bnb_mean) || (a.up_c0 < a.Cout && !a.out))) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.out_up ? a.up_mask : (a.bnb_y ? a.bnb_y : a.mask)) || a.up_accum)) return hipErrorInvalidValue;
  const int tiles = a.N * (a.Ho / kVH) * (a.Wo / kVW);
  const size_t main_lds = (size_t)2 * kVBuf + (size_t)kVRounds * 256 * sizeof(int), q_lds = (size_t)4 * 64 * kVQLd * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  // 32-channel tiles: the 32-output layers, and (forward only, where nothing co-runs) launches whose 64-channel tiles cannot give
  // every CU two workgroups
  const long wgs64 = (long)route_N(a) * (a.Ho / kVH) * (a.Wo / kVW) * ((a.Cout + 63) / 64);
  const bool fwd_alone = a.rmul == 1 && !a.xmax;
  const bool narrow = variant == 5 || (variant == 0 && !a.out_up && ((a.Cout & 63) != 0 || (fwd_alone && wgs64 < 2L * device_cu_count())));
  if (narrow) {
    if (a.out_up) return hipErrorInvalidValue;
    static DevOnce lds_attr1;
    { hipError_t e = lds_attr1.set_max_lds((const void*)conv_f16x3v2_kernel<1>, lds); if (e != hipSuccess) return e; }
    UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3v2_kernel<1>), dim3((unsigned)(tiles * (a.Cout / 32))), dim3(256), lds, st, a);
    return hipGetLastError();
  }
  if (a.Cout & 63) return hipErrorInvalidValue;
  static DevOnce lds_attr2;
  { hipError_t e = lds_attr2.set_max_lds((const void*)conv_f16x3v2_kernel<2>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3v2_kernel<2>), dim3((unsigned)(tiles * (a.Cout / 64))), dim3(256), lds, st, a);
  return hipGetLastError();
}

}  // namespace uwm
