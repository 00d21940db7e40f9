// fp16x3 direct 3x3 / stride-1 convolution (forward AND dgrad) for gfx950 on v_mfma_f32_16x16x32_f16 — the CDNA4 half-precision
// matrix instruction, 16x the rate of the fp32 one — with fp32-class accuracy:
//
//   every fp32 operand is split ONCE, when it is staged, into two fp16 halves  x = hi + lo  (hi = fp16(x), lo = fp16(x - hi):
//   22 mantissa bits together) and a product is formed as  hi*hi' + hi*lo' + lo*hi'  with fp32 accumulation inside the MFMA:
//   relative error 2^-22 per product (the dropped lo*lo' term), against 2^-24 for an fp32 multiply — measured per layer in
//   tests/test_ops_gpu.py.  (The bf16x3 mode of conv_wino_x3.hip keeps 16 bits: 2^-16, which is what pushed its forward
//   outside the 1e-3 logit bar.)  fp16's narrow EXPONENT is handled by exact power-of-two scaling: every filter row is scaled so
//   that its largest tap lands in [2^13, 2^14) (f16x3_rowscale_kernel; undone in the epilogue), a dgrad's dY by a per-tensor
//   power of two (ConvArgs::xscale); activations (O(1) behind a BatchNorm) are taken as they are and clamped to +-65504.
//
// Direct form, not Winograd: 3 half-precision MFMAs per 32 products at 16x the fp32 rate is 2.4x the fp32-Winograd MFMA
// floor, and the kernel has no transforms to pay — the workgroup is bound by staging (HBM / LDS), not by the matrix pipe.
//
// Work split: workgroup = 16x16 output pixels x 64 output channels, 4 waves, wave w = pixel rows 4w .. 4w+3 (4 pixel fragments)
// x 4 channel fragments = 16 accumulator tiles (64 VGPRs).  K runs over 16-channel chunks x 5 tap pairs (tap slots 0..9, slot 9
// is zero padding): lane group g = lane >> 4 of an MFMA takes tap slot 2*ks + (g >> 1), channels 8*(g & 1) .. +7.
//   * the 18x18 halo patch of a chunk goes to LDS through registers (lazy BatchNorm + ReLU, nearest x2 upsample, channel concat,
//     zero padding and the hi / lo split applied while staging; 80 bytes per pixel: [hi 16 ch | lo 16 ch | pad] — the pad makes
//     the 16 pixel rows of a ds_read_b128 fragment read conflict-free), double-buffered, one barrier per chunk;
//   * the filter fragments come straight from global memory / L2 in MFMA lane order (f16x3_weights_multi_kernel packs
//     [chunk][tap pair][channel fragment][hi | lo][lane][8 halfs] once per step), one 16-byte load per lane and fragment,
//     prefetched one tap pair ahead in registers: no LDS traffic, no barrier for the weights;
//   * weights are the A operand (rows = output channels), pixels the B operand, so a lane ends with 4 consecutive output
//     channels of one pixel: one 16-byte NHWC store.  Epilogue contract of conv_wino_kernel's plain form: row un-scale, bias,
//     residual addend, ReLU mask, BatchNorm statistics / fused BatchNorm-backward sums (bnb_*, bnb_y).
//
// Reference semantics replaced: the 3x3 convolutions of smp.Unet's encoder / decoder forward and their input gradients
// (/root/reference/src/models/unet_model.py:64-71 -> SURVEY.md §8 a5-a8, a10, a14).
#include "uwm_kernels.h"

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#ifndef UWM_F16_ABL
#define UWM_F16_ABL 0       // compile-time timing ablations (scripts/ablate_f16x3.sh): 1 no MFMA, 2 no filter loads, 4 no pixel-fragment LDS reads, 8 no patch loads / stores, 16 no epilogue; 0 in the product build
#endif
constexpr int kFT = 16, kFP = kFT + 2, kFPP = kFP * kFP;      // 16x16 output pixels, 18x18 = 324 patch pixels
constexpr int kFPix = 40;                                      // halfs per patch pixel: 16 hi + 16 lo + 8 pad (80 bytes)
constexpr int kFBuf = kFPP * kFPix;                            // halfs per patch buffer (25 920 bytes)
constexpr int kFKs = 5;                                        // tap pairs per chunk (10 tap slots, the last one zero)
constexpr int kFRounds = (kFPP * 4 + 255) / 256;               // 16-byte units of a chunk (324 px x 4) over 256 threads: 6 rounds

// ---------------------------------------------------------------- filter bank
// bank = [C/16 chunks][5 tap pairs][nJ fragments][2 planes][64 lanes][8 halfs] halfs, then float rinv[nJ*16] (1 / row scale).
// rows are padded to whole 64-row tiles with zeros.  mode 0: rows x [tap][chans] forward weights (k = tap*chans + c);
// mode 2: the dgrad bank straight from the FORWARD weights: rows = input channels, chans = output channels,
// g'[tap][ch] = w[ch][8 - tap][row] (transposed + mirrored).
__host__ __device__ inline int f16x3_nj_(int rows) { return ((rows + 63) / 64) * 4; }
int f16x3_nj(int rows) { return f16x3_nj_(rows); }
size_t f16x3_bank_floats(int rows, int chans) {          // in floats (the model's workspace unit)
  const size_t halfs = (size_t)(chans / 16) * kFKs * f16x3_nj(rows) * 2 * 64 * 8;
  return halfs / 2 + (size_t)f16x3_nj(rows) * 16;
}
size_t f16x3_rinv_off(int rows, int chans) { return f16x3_rinv_off_floats(rows, chans); }

__device__ __forceinline__ float f16x3_wval(const WinoJob& jb, int row, int slot, int ch) {
  if (row >= jb.rows || slot >= 9 || ch >= jb.chans) return 0.f;
  if (jb.mode == 0) return jb.w[(size_t)row * jb.Kpad + (size_t)slot * jb.chans + ch];
  return ch < jb.src_rows ? jb.w[(size_t)ch * jb.Kpad + (size_t)(8 - slot) * jb.rows + row] : 0.f;
}
// Row maxima -> row scales, two launches: (1) max|w| per filter row into rmax (zeroed; non-negative floats order as integers, so the
// partial maxima of the workgroups that share a row meet in an atomicMax), with lanes along whatever is contiguous in memory —
// the k index of a forward bank (a wave per row, 4 rows per workgroup, the k range split over blockIdx.z), the ROW index of a
// dgrad bank built from the forward weights (mode 2: a lane per row, 64 rows per wave, (tap, channel) split over waves and
// blockIdx.z); (2) rinv[row] = 2^(e - 14) for row max = m * 2^e (m in [0.5, 1)): the largest tap lands in [2^13, 2^14).
// (The first version — one wave per row walking the whole row — took 70-94 us per launch, 0.19 ms per step.)
__global__ __launch_bounds__(256) void f16x3_rowzero_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row < f16x3_nj_(jb.rows) * 16) (jb.ut + f16x3_rinv_off_floats(jb.rows, jb.chans))[row] = 0.f;
}
__global__ __launch_bounds__(256) void f16x3_rowmax_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  const int nJ = f16x3_nj_(jb.rows);
  unsigned* rmax = (unsigned*)(jb.ut + f16x3_rinv_off_floats(jb.rows, jb.chans));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int K = 9 * jb.chans, nz = gridDim.z, z = blockIdx.z;
  if (jb.mode == 0) {
    const int row = blockIdx.x * 4 + wv;
    if (row >= jb.rows) return;
    float mx = 0.f;
    const f4* wr = (const f4*)(jb.w + (size_t)row * jb.Kpad);      // (K = 9 * chans, chans % 16 == 0; rows are 128-byte aligned)
    for (int i = z * 64 + lane; i < K / 4; i += nz * 64) {
      const f4 v = wr[i];
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (lane == 0 && mx > 0.f) atomicMax(rmax + row, __float_as_uint(mx));
  } else {
    const int row = blockIdx.x * 64 + lane;            // (blockIdx.x counts 64-row groups here)
    if (blockIdx.x * 64 >= jb.rows) return;
    float mx = 0.f;
    if (row < jb.rows)
      for (int i = z * 4 + wv; i < 9 * jb.src_rows; i += nz * 4) {      // i = ch * 9 + tap
        const int ch = i / 9, tap = i - ch * 9;
        mx = fmaxf(mx, fabsf(jb.w[(size_t)ch * jb.Kpad + (size_t)tap * jb.rows + row]));
      }
    if (row < jb.rows && mx > 0.f) atomicMax(rmax + row, __float_as_uint(mx));
  }
  (void)nJ;
}
__global__ __launch_bounds__(256) void f16x3_rowscale_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  const int nJ = f16x3_nj_(jb.rows);
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= nJ * 16) return;
  float* rinv = jb.ut + f16x3_rinv_off_floats(jb.rows, jb.chans);
  const float mx = rinv[row];                          // (the maximum, left there by f16x3_rowmax_kernel; 0 for padding rows)
  float s = 1.f;
  if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); s = ldexpf(1.f, 14 - e); }
  rinv[row] = 1.f / s;                                 // (exact: a power of two)
}
__global__ __launch_bounds__(256) void f16x3_weights_multi_kernel(const WinoJobs jobs) {
  const WinoJob jb = jobs.j[blockIdx.y];
  if (jb.pad_ == 1) return;                            // layout 1: f16x3v2_weights_multi_kernel
  const int nJ = f16x3_nj_(jb.rows);
  const size_t total = (size_t)(jb.chans / 16) * kFKs * nJ * 64;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int j = (int)((i >> 6) % nJ);
  const int t = (int)((i >> 6) / nJ);                 // chunk * 5 + ks
  const int ks = t % kFKs, chunk = t / kFKs;
  const int row = j * 16 + (lane & 15), g = lane >> 4;
  const int slot = 2 * ks + (g >> 1), ch0 = chunk * 16 + 8 * (g & 1);
  const float* rinv = jb.ut + f16x3_rinv_off_floats(jb.rows, jb.chans);
  const float s = 1.f / rinv[row];
  h8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = f16x3_wval(jb, row, slot, ch0 + e) * s;
    const _Float16 h = (_Float16)v;
    hi[e] = h; lo[e] = (_Float16)(v - (float)h);
  }
  _Float16* bank = (_Float16*)jb.ut;
  h8* dst = (h8*)(bank + ((size_t)t * nJ + j) * 1024 + lane * 8);
  dst[0] = hi;
  dst[64] = lo;                                       // plane 1: +512 halfs
}
hipError_t launch_f16x3_weights_multi(const WinoJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return hipSuccess;
  size_t mx = 0; int mr = 0;
  for (int i = 0; i < jobs.n; ++i) {
    if (jobs.j[i].chans & 15) return hipErrorInvalidValue;
    const size_t t = (size_t)(jobs.j[i].chans / 16) * kFKs * f16x3_nj(jobs.j[i].rows) * 64;
    if (t > mx) mx = t;
    if (f16x3_nj(jobs.j[i].rows) * 16 > mr) mr = f16x3_nj(jobs.j[i].rows) * 16;
  }
  hipLaunchKernelGGL(f16x3_rowzero_kernel, dim3((unsigned)((mr + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);      // the row maxima meet in atomicMax: start from zero
  hipLaunchKernelGGL(f16x3_rowmax_kernel, dim3((unsigned)((mr + 3) / 4), (unsigned)jobs.n, 8), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(f16x3_rowscale_kernel, dim3((unsigned)((mr + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);
  hipLaunchKernelGGL(f16x3_weights_multi_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)jobs.n), dim3(256), 0, st, jobs);
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) return e; }
  return launch_f16x3v2_weights_multi(jobs, st);
}

// Concat-split epilogue of a decoder conv1 dgrad (ConvArgs::out_up; the contract of conv_wino_kernel's): the output channels
// [0, up_c0) are the gradient wrt the nearest-x2 up-sampled tensor — summed over each 2x2 pixel block, ReLU-masked by the
// low-resolution producer, (optionally) accumulated, written at half resolution, with the fused BatchNorm-backward sums —
// and the channels [up_c0, Cout) the gradient wrt the skip tensor, written at full resolution with its own channel count.
// up_c0 % 64 == 0, so a 64-channel tile lies entirely on one side.  R = a [4 rows x 16 columns][kQLd] pixel block in LDS
// (rows row0 .. row0 + nrows - 1 of it are this wave's), (hb, wb) = image coordinates of the block's pixel (0, 0).
template <int NROWS>
__device__ __forceinline__ void f16x3_split_epilogue(const ConvArgs& a, const float* R, int kQLd, int row0, int n, int hb, int wb, int n0, int lane,
                                                     f4 rs, f4& ps_, f4& pq_) {
  const int cq = lane & 15, sub = lane >> 4;
  const int co = n0 + cq * 4;
  if (co >= a.Cout) return;
  if (n0 < a.up_c0) {
    const bool bnb = a.bnb_mean != nullptr;
    f4 bmu = {0.f, 0.f, 0.f, 0.f}, brs = bmu, msc = {1.f, 1.f, 1.f, 1.f}, msh = bmu;
    if (bnb) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
    if (a.up_mscale) { msc = *(const f4*)(a.up_mscale + co); msh = *(const f4*)(a.up_mshift + co); }
#pragma unroll
    for (int r = 0; r < NROWS; ++r) {                      // NROWS / 2 block rows x 8 block columns = NROWS * 4 blocks over 4 sub-lanes
      const int b = r * 4 + sub;
      const int by = b >> 3, bx = b & 7;
      const int y = row0 + 2 * by, x = 2 * bx;
      const int ho = hb + y, wo = wb + x;
      if (ho < a.Ho && wo < a.Wo) {
        const float* q = R + (y * 16 + x) * kQLd + cq * 4;
        f4 v = (*(const f4*)q + *(const f4*)(q + kQLd) + *(const f4*)(q + 16 * kQLd) + *(const f4*)(q + 17 * kQLd)) * rs;
        const size_t o2 = (((size_t)n * (a.Ho >> 1) + (ho >> 1)) * (a.Wo >> 1) + (wo >> 1)) * a.up_c0 + co;
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
    }
  } else {
    const int c1n = a.Cout - a.up_c0;
#pragma unroll 4
    for (int r = 0; r < NROWS * 4; ++r) {                  // NROWS * 16 pixels over 4 sub-lanes
      const int p = r * 4 + sub;
      const int y = row0 + (p >> 4), x = p & 15;
      const int ho = hb + y, wo = wb + x;
      if (ho < a.Ho && wo < a.Wo)
        *(f4*)(a.out + (((size_t)n * a.Ho + ho) * a.Wo + wo) * c1n + (co - a.up_c0)) = *(const f4*)(R + (y * 16 + x) * kQLd + cq * 4) * rs;
    }
  }
}

// ---------------------------------------------------------------- main kernel
__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }      // (one v_med3_f32; fminf(fmaxf()) adds a canonicalising v_max per value)

// NJ = 16-channel fragments per workgroup: 4 (64 output channels) or 2 (32: the 32-output layers of decoder block 3 — half of a
// 64-channel tile's MFMAs and filter loads were padding there: 400 / 162 us forward against 416 / 152 on the fp32 Winograd kernel)
// NJ = 1: the 16 -> 16-channel full-resolution layer of decoder block 4 (conv2, forward and dgrad): ONE 16-channel chunk, so no
// staging inside the loop, one patch buffer, a [64 px][20]-float epilogue block per wave — 26 KB of LDS and ~100 VGPRs: up to
// six short-lived workgroups per CU cover each other's load latency (the fp32 conv_patch16 kernel is bound by its matrix pipe
// there: 123 us of fp32 MFMAs against a 107-us HBM floor)
// NP = split products per tile (ConvArgs::nprod): 3 = hi*hi' + hi*lo' + lo*hi'; 2 = without the pixel operand's low half (hi*lo'); 1 = hi*hi' only
template <int NJ, int NP = 3>
__global__ __launch_bounds__(256, (NJ == 1 ? 4 : 2)) void conv_f16x3_kernel(const ConvArgs a) {
  constexpr bool kOne = NJ == 1;                       // single-chunk form
  constexpr int kCo = 16 * NJ;                         // output channels per workgroup
  constexpr int kCQ = 4 * NJ, kSub = 64 / kCQ;        // epilogue: lanes along the channel quads x pixel sub-rows
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];      // [2][324 px][40 halfs]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px16 = lane & 15, g = lane >> 4;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + kCo - 1) / kCo;
  const int tilesW = (a.Wo + kFT - 1) / kFT, tilesH = (a.Ho + kFT - 1) / kFT;
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int n0 = tn * kCo, h0 = th * kFT, w0 = tw * kFT;
  const int nJ = a.wu_ncb;

  f4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- patch staging through registers: 1296 16-byte units (pixel, channel quad) = 5 whole rounds of 256 + 16 units (threads 0-15).
  // A round's source offset is chunk-invariant: held in registers for the source being read; the second source's (the two
  // differ in size when one of them is up-sampled) wait in LDS until the chunk walk crosses the concat boundary (recomputing
  // them there cost 28 spilled VGPRs); the in-bounds bits sit in one flag word
  int goff[kFRounds];
  unsigned gflags = 0;
  int* const goff_s1 = (int*)(hsm + 2 * kFBuf);          // [6][256] behind the patch buffers: the second source's offsets, parked in LDS
#pragma unroll
  for (int rd = 0; rd < kFRounds; ++rd) {
    const int u = rd * 256 + tid;
    const bool act = u < kFPP * 4;
    const int pp = act ? (u >> 2) : 0;
    const int py = pp / kFP, pxx = pp - py * kFP;
    const int hl = h0 - 1 + py, wl = w0 - 1 + pxx;
    const bool ok = act && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
    const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
    goff[rd] = ((n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C;
    if (!kOne && a.C0 < a.Ctot) goff_s1[rd * 256 + tid] = ((n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C;      // (read back by this thread only)
    gflags |= (ok ? 1u : 0u) << rd;
  }
  // dgrad: dY is tiny (1e-3 ... 1e-9): it is staged times the power of two that puts max|dY| into [2^13, 2^14) (exact; undone in
  // the epilogue).  max|dY| = the maximum of the 32 slots bn_bwd_apply filled; activations are staged as they are
  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }
  // ONE register set of raw patch values (a round's register is re-loaded for chunk c+2 right after its chunk-c+1 contents have been
  // stored) and one set of lazy-transform coefficients (those of the chunk being stored).  The store side is branch-free and cut into rounds, so that a round's
  // conversion work (about 30 VALU instructions + 2 LDS stores) can sit in the shadow of a tap pair's MFMAs — the matrix pipe
  // and the VALU do not co-execute across waves here (PMC: SQ_VALU_MFMA_COEXEC_CYCLES = 6 % of the MFMA-busy cycles with the
  // staging as a phase of its own; 2.8 VALU instructions per MFMA, i.e. 70 % of the MFMA time again, back to back)
  constexpr int kWR = kFRounds - 1;                    // whole rounds (5); the 16 left-over units: pvX, threads 0-15
  f4 pv[kWR], pvX = {0.f, 0.f, 0.f, 0.f}, psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  float pfloor = -3.0e38f;                               // ReLU as max(v, floor): 0 or "-inf"
  const bool has_x = tid < kFPP * 4 - kWR * 256;
  const float* lsp = nullptr;                            // source pointer of the chunk whose loads are being issued
  auto chunk_src = [&](int cc) {
    const int c = cc * 16;
    const bool first = c < a.C0;
    if (c == a.C0 && a.C0 < a.Ctot) {                    // the walk crosses the concat boundary (chunks are visited in order)
#pragma unroll
      for (int rd = 0; rd < kFRounds; ++rd) goff[rd] = goff_s1[rd * 256 + tid];
    }
    lsp = (first ? a.s0.ptr : a.s1.ptr) + (first ? c : c - a.C0) + (tid & 3) * 4;
  };
  auto coef_load = [&](int cc) {
    const int c = cc * 16;
    const bool first = c < a.C0;
    const Src& s = first ? a.s0 : a.s1;
    const int cl = (first ? c : c - a.C0) + (tid & 3) * 4;
    pfloor = s.relu ? 0.f : -3.0e38f;
    if (s.scale != nullptr) { psc = *(const f4*)(s.scale + cl); psh = *(const f4*)(s.shift + cl); }
    else { psc = (f4){1.f, 1.f, 1.f, 1.f}; psh = (f4){0.f, 0.f, 0.f, 0.f}; pfloor = -3.0e38f; }
  };
  auto store_unit = [&](int buf, int rd, f4 raw) {
    f4 v = raw * psc + psh;
    v.x = fmaxf(v.x, pfloor); v.y = fmaxf(v.y, pfloor); v.z = fmaxf(v.z, pfloor); v.w = fmaxf(v.w, pfloor);
    v = v * xs;
    if (!((gflags >> rd) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
    h4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = clamp_h(v[e]);
      const _Float16 h = (_Float16)x;
      hi[e] = h; lo[e] = (_Float16)(x - (float)h);
    }
    const int u = rd * 256 + tid;
    _Float16* d = hsm + buf * kFBuf + (u >> 2) * kFPix + (u & 3) * 4;
    *(h4*)d = hi;
    *(h4*)(d + 16) = lo;
  };

  // ---- filter fragments: global -> registers, one tap pair ahead.  step t = chunk * 5 + ks
  const _Float16* const wb = (const _Float16*)a.wu + (size_t)(n0 / 16) * 1024 + lane * 8;
  const int nchunk = a.Ctot >> 4, nsteps = nchunk * kFKs;
  constexpr int dbg = UWM_F16_ABL;
  auto w_load = [&](int t, h8 (&whi)[NJ], h8 (&wlo)[NJ]) {
    if ((dbg & 2) && t > 0) return;
    const _Float16* p = wb + (size_t)t * nJ * 1024;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { whi[j] = *(const h8*)(p + j * 1024); wlo[j] = *(const h8*)(p + j * 1024 + 512); }
  };
  // ---- pixel fragments: lane (pixel column px16, group g) reads 8 channels of tap slot 2*ks + (g >> 1) at pixel row 4*wave + i
  const int pbase = ((wave * 4) * kFP + px16) * kFPix + (g & 1) * 8;
  const int ghi = g >> 1;
  auto mma_step = [&](int ks, const _Float16* pc, const h8 (&whi)[NJ], const h8 (&wlo)[NJ]) {
    const int slot0 = 2 * ks, slot1 = 2 * ks + 1 > 8 ? 8 : 2 * ks + 1;      // (slot 9: its weights are zero; read a valid address)
    const int off0 = ((slot0 / 3) * kFP + slot0 % 3) * kFPix, off1 = ((slot1 / 3) * kFP + slot1 % 3) * kFPix;
    const _Float16* pp = pc + pbase + (ghi ? off1 : off0);
    h8 xh[4], xl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (dbg & 4) { xh[i] = whi[i % NJ]; xl[i] = wlo[i % NJ]; continue; }
      xh[i] = *(const h8*)(pp + i * kFP * kFPix); xl[i] = *(const h8*)(pp + i * kFP * kFPix + 16);
    }
    if (dbg & 1) { acc[0][0][0] += (float)xh[0][0] + (float)xl[1][1] + (float)whi[NJ > 1 ? NJ - 2 : 0][2] + (float)wlo[NJ - 1][3] + (float)xh[2][0] + (float)xh[3][0]; return; }
    // the three products of a tile go to the SAME accumulator: issue them 16 tiles apart (a dependent MFMA waits ~2 issue
    // slots for its predecessor's result), product type outermost
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xh[i], acc[i][j], 0, 0, 0);
    if (NP >= 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
    }
    if (NP >= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
    }
  };

  h8 wA_hi[NJ], wA_lo[NJ], wB_hi[NJ], wB_lo[NJ];
  w_load(0, wA_hi, wA_lo);
  chunk_src(0);
#pragma unroll
  for (int rd = 0; rd < kWR; ++rd) pv[rd] = *(const f4*)(lsp + goff[rd]);
  if (has_x) pvX = *(const f4*)(lsp + goff[kWR]);
  coef_load(0);
#pragma unroll
  for (int rd = 0; rd < kWR; ++rd) store_unit(0, rd, pv[rd]);
  if (has_x) store_unit(0, kWR, pvX);
  if (!kOne && nchunk > 1) {                             // chunk 1 (it may already be the second source)
    chunk_src(1);
#pragma unroll
    for (int rd = 0; rd < kWR; ++rd) pv[rd] = *(const f4*)(lsp + goff[rd]);
    if (has_x) pvX = *(const f4*)(lsp + goff[kWR]);
  }
  __syncthreads();

  // Iteration c multiplies chunk c (buffer c & 1) and, inside the MFMA stream, one round per tap pair: converts and stores round
  // ks of chunk c+1 (its raw values arrived during iteration c-1) into the other buffer and re-issues that register's global load
  // for chunk c+2 — a full iteration of latency budget with ONE register set.  Two chunks per loop trip: the filter-fragment
  // register sets alternate statically.
  for (int cc = 0; cc < nchunk; cc += 2) {
#pragma unroll
    for (int hh = 0; hh < (kOne ? 1 : 2); ++hh) {
      const int c = cc + hh, cur = hh, nxt = hh ^ 1;       // (nchunk is even: chunk c sits in buffer c & 1 = hh)
      const bool more1 = c + 1 < nchunk;
      if (!kOne) {
        coef_load(more1 ? c + 1 : c);
        chunk_src(c + 2 < nchunk ? c + 2 : nchunk - 1);    // (past the end: a harmless re-fetch)
      }
      const _Float16* pc = hsm + cur * kFBuf;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < kFKs; ++ks) {
        const int t = c * kFKs + ks;
        const int tnext = t + 1 < nsteps ? t + 1 : t;
        if (((hh * kFKs + ks) & 1) == 0) { w_load(tnext, wB_hi, wB_lo); mma_step(ks, pc, wA_hi, wA_lo); }
        else { w_load(tnext, wA_hi, wA_lo); mma_step(ks, pc, wB_hi, wB_lo); }
        if (!(dbg & 8) && !kOne) {
          store_unit(nxt, ks, pv[ks]);                     // (past the last chunk: into the dead buffer)
          pv[ks] = *(const f4*)(lsp + goff[ks]);
        }
        // schedule of the tap pair: filter loads and fragment reads first, then the round's VALU work spread under the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * NJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int q = 0; q < 4 * NJ; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, NP, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(dbg & 8) && !kOne && has_x) {
        if (more1) store_unit(nxt, kWR, pvX);
        pvX = *(const f4*)(lsp + goff[kWR]);
      }
      __syncthreads();
    }
  }
  if (dbg & 16) { if (acc[0][0][0] + acc[1][1][1] + acc[2][NJ > 1 ? NJ - 2 : 0][2] + acc[3][NJ - 1][3] == 123.456f) a.out[0] = 1.f; return; }

  // ---------------- epilogue: D[row = co 4g + e][col = pixel px16].  A lane holds 4 channels of one pixel per tile: stored
  // straight from the accumulators a wave instruction writes sixteen 64-byte pieces (half cache lines: 49 of the 94 us of a
  // layer1 launch).  So every wave passes its 64 px x 64 ch block through LDS (its own region, [pixel][64 + 4 pad] floats; the
  // patch buffers are dead) and reads it back with lanes along the channels: 256 contiguous bytes per pixel, per-thread
  // constant channel quad (row un-scale, bias, mask coefficients and the statistics stay in registers)
  constexpr int kQLd = kOne ? 20 : 68;
  float* const R = (float*)hsm + wave * 64 * kQLd;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) *(f4*)(R + (i * 16 + px16) * kQLd + j * 16 + g * 4) = acc[i][j];
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
  if (NJ == 4 && a.out_up != nullptr) {                 // (the concat split works on 64-channel tiles: the launcher keeps it on NJ = 4)
    f4 rs = {0.f, 0.f, 0.f, 0.f};
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    f16x3_split_epilogue<4>(a, R, kQLd, 0, n, h0 + wave * 4, w0, n0, lane, rs, ps_, pq_);
  } else {
    f4 rs = {0.f, 0.f, 0.f, 0.f}, bmu = rs, brs = rs, bia = rs, msc = {1.f, 1.f, 1.f, 1.f}, msh = rs;
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    if (bnb && cok) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
    if (a.bias && cok) bia = *(const f4*)(a.bias + co);
    if (a.mscale && cok) { msc = *(const f4*)(a.mscale + co); msh = *(const f4*)(a.mshift + co); }
#pragma unroll 4
    for (int r = 0; r < 64 / kSub; ++r) {
      const int p = r * kSub + sub;
      const int ho = h0 + wave * 4 + (p >> 4), wo = w0 + (p & 15);
      if (ho < a.Ho && wo < a.Wo && cok) {
        const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
        f4 v = *(const f4*)(R + p * kQLd + cq * 4) * rs + bia;
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
    // the 4 pixel sub-rows of a wave (xor 16, 32) -> 4 waves through LDS -> fp64 atomics on one replica
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
#pragma unroll
      for (int d = kCQ; d < 64; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
      ps_[e] = sv; pq_[e] = qv;
    }
    __syncthreads();                             // every wave is done with its block
    float* red = (float*)hsm;                    // [4 waves][64][2]
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

// ---------------------------------------------------------------- 8-wave variant: MMA waves and loader waves
// The 4-wave kernel above serialises, inside every wave, the staging of the next chunk (global loads, lazy BatchNorm + ReLU, the
// hi / lo split, LDS stores: ~14 us of a 71-us layer1 launch by the compile-time ablations) with the MFMAs of the current one.
// Here ONE 512-thread workgroup per CU gives each SIMD an MMA wave and a LOADER wave: waves 0-3 only read fragments and issue
// MFMAs (same tiling), waves 4-7 only stage — chunk c+1 into the other buffer while chunk c is multiplied, with the global
// loads of chunk c+2 already in flight in a second register set (two chunks of latency budget).  One barrier per chunk.
// All eight waves share the epilogue (32 pixels x 64 channels each).
template <int NP = 3>
__global__ __launch_bounds__(512, 1) void conv_f16x3s_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];      // [2][324 px][40 halfs]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_mma = wave < 4;
  const int mw = wave & 3;                            // MMA wave index / loader wave index
  const int ltid = tid & 255;                         // thread index inside its role
  const int px16 = lane & 15, g = lane >> 4;

  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + 63) / 64;
  const int tilesW = (a.Wo + kFT - 1) / kFT, tilesH = (a.Ho + kFT - 1) / kFT;
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int n0 = tn * 64, h0 = th * kFT, w0 = tw * kFT;
  const int nJ = a.wu_ncb;
  const int nchunk = a.Ctot >> 4, nsteps = nchunk * kFKs;

  float xs = 1.f;
  if (a.xmax) {
    float mx = a.xmax[lane & 31];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); xs = ldexpf(1.f, 14 - e); }
  }

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  if (!is_mma) {
    // ================= loader waves =================
    int goff0[kFRounds], goff1[kFRounds];                // stage-invariant source offsets of a round, per source of the concat
    unsigned gflags = 0;
#pragma unroll
    for (int rd = 0; rd < kFRounds; ++rd) {
      const int u = rd * 256 + ltid;
      const bool act = u < kFPP * 4;
      const int pp = act ? (u >> 2) : 0;
      const int py = pp / kFP, pxx = pp - py * kFP;
      const int hl = h0 - 1 + py, wl = w0 - 1 + pxx;
      const bool ok = act && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
      const int hc = min(max(hl, 0), a.Hl - 1), wc = min(max(wl, 0), a.Wl - 1);
      goff0[rd] = ((n * a.s0.H + (hc >> a.s0.up)) * a.s0.W + (wc >> a.s0.up)) * a.s0.C;
      goff1[rd] = a.C0 < a.Ctot ? ((n * a.s1.H + (hc >> a.s1.up)) * a.s1.W + (wc >> a.s1.up)) * a.s1.C : goff0[rd];
      gflags |= ((ok ? 1u : 0u) | (act ? 2u : 0u)) << (2 * rd);
    }
    struct Stage { f4 pv[kFRounds]; f4 sc, sh; int relu; bool has; };
    auto patch_load = [&](int cc, Stage& st) {
      const int c = cc * 16;
      const bool first = c < a.C0;
      const Src& s = first ? a.s0 : a.s1;
      st.relu = s.relu;
      const int cl = (first ? c : c - a.C0) + (ltid & 3) * 4;
      st.has = s.scale != nullptr;
      if (st.has) { st.sc = *(const f4*)(s.scale + cl); st.sh = *(const f4*)(s.shift + cl); }
      const float* sp = s.ptr + cl;
#pragma unroll
      for (int rd = 0; rd < kFRounds; ++rd) st.pv[rd] = *(const f4*)(sp + (first ? goff0[rd] : goff1[rd]));
    };
    auto patch_store = [&](int buf, const Stage& st) {
#pragma unroll
      for (int rd = 0; rd < kFRounds; ++rd) {
        f4 v = st.pv[rd];
        if (st.has) {
          v = v * st.sc + st.sh;
          if (st.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        v = v * xs;
        if (!((gflags >> (2 * rd)) & 1u)) v = (f4){0.f, 0.f, 0.f, 0.f};
        if ((gflags >> (2 * rd)) & 2u) {
          h4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = clamp_h(v[e]);
            const _Float16 h = (_Float16)x;
            hi[e] = h; lo[e] = (_Float16)(x - (float)h);
          }
          const int u = rd * 256 + ltid;
          _Float16* d = hsm + buf * kFBuf + (u >> 2) * kFPix + (u & 3) * 4;
          *(h4*)d = hi;
          *(h4*)(d + 16) = lo;
        }
      }
    };
    Stage sA, sB;
    patch_load(0, sA);
    patch_load(nchunk > 1 ? 1 : 0, sB);
    patch_store(0, sA);
    __syncthreads();                                       // chunk 0 staged
    // iteration c (two per loop trip: static register sets): store chunk c+1 (loaded one iteration ago), load chunk c+2
    for (int cc = 0; cc < nchunk; cc += 2) {
      {                                                    // c = cc: chunk c+1 sits in sB
        const int c2 = cc + 2 < nchunk ? cc + 2 : nchunk - 1;
        if (cc + 1 < nchunk) patch_store(1, sB);
        patch_load(c2, sA);
        __syncthreads();
      }
      {                                                    // c = cc + 1: chunk c+1 = cc+2 sits in sA
        const int c3 = cc + 3 < nchunk ? cc + 3 : nchunk - 1;
        if (cc + 2 < nchunk) patch_store(0, sA);
        patch_load(c3, sB);
        __syncthreads();
      }
    }
  } else {
    // ================= MMA waves =================
    const _Float16* const wb = (const _Float16*)a.wu + (size_t)(n0 / 16) * 1024 + lane * 8;
    auto w_load = [&](int t, h8 (&whi)[4], h8 (&wlo)[4]) {
      const _Float16* p = wb + (size_t)t * nJ * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j) { whi[j] = *(const h8*)(p + j * 1024); wlo[j] = *(const h8*)(p + j * 1024 + 512); }
    };
    const int pbase = ((mw * 4) * kFP + px16) * kFPix + (g & 1) * 8;
    const int ghi = g >> 1;
    auto x_load = [&](int ks, const _Float16* pc, h8 (&xh)[4], h8 (&xl)[4]) {
      const int slot0 = 2 * ks, slot1 = 2 * ks + 1 > 8 ? 8 : 2 * ks + 1;
      const int off0 = ((slot0 / 3) * kFP + slot0 % 3) * kFPix, off1 = ((slot1 / 3) * kFP + slot1 % 3) * kFPix;
      const _Float16* pp = pc + pbase + (ghi ? off1 : off0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { xh[i] = *(const h8*)(pp + i * kFP * kFPix); xl[i] = *(const h8*)(pp + i * kFP * kFPix + 16); }
    };
    auto mma = [&](const h8 (&whi)[4], const h8 (&wlo)[4], const h8 (&xh)[4], const h8 (&xl)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xh[i], acc[i][j], 0, 0, 0);
      if (NP >= 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[j], xl[i], acc[i][j], 0, 0, 0);
      }
      if (NP >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[j], xh[i], acc[i][j], 0, 0, 0);
      }
    };
    h8 wA_hi[4], wA_lo[4], wB_hi[4], wB_lo[4], xh[4], xl[4];
    w_load(0, wA_hi, wA_lo);
    __syncthreads();                                       // chunk 0 staged
    for (int cc = 0; cc < nchunk; cc += 2) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int c = cc + hh;
        const _Float16* pc = hsm + hh * kFBuf;             // (nchunk is even: chunk c sits in buffer c & 1 = hh)
#pragma unroll
        for (int ks = 0; ks < kFKs; ++ks) {
          const int t = c * kFKs + ks;
          const int tnext = t + 1 < nsteps ? t + 1 : t;
          // the next tap pair's filter loads and this one's fragment reads are ISSUED before the MFMA block (the fence keeps hipcc
          // from sinking the loads behind the MFMAs to save registers: that exposed their latency at every step)
          x_load(ks, pc, xh, xl);
          if (((hh * kFKs + ks) & 1) == 0) { w_load(tnext, wB_hi, wB_lo); __builtin_amdgcn_sched_barrier(0); mma(wA_hi, wA_lo, xh, xl); }
          else { w_load(tnext, wA_hi, wA_lo); __builtin_amdgcn_sched_barrier(0); mma(wB_hi, wB_lo, xh, xl); }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
    }
  }

  // ---------------- epilogue (conv_f16x3_kernel's, spread over eight waves): an MMA wave's 64 px x 64 ch block goes through its
  // LDS region; waves w and w + 4 read back its pixels [0, 32) and [32, 64) with lanes along the channels
  constexpr int kQLd = 68;
  if (is_mma) {
    float* const Rw = (float*)hsm + mw * 64 * kQLd;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f4*)(Rw + (i * 16 + px16) * kQLd + j * 16 + g * 4) = acc[i][j];
  }
  __syncthreads();
  const float* const R = (const float*)hsm + mw * 64 * kQLd;
  const float* rinv = (const float*)a.wu + a.wu_rinv_off;
  const float ixs = 1.f / xs;
  const bool do_stats = a.ssum != nullptr;
  const bool bnb = a.bnb_mean != nullptr;
  const int cq = lane & 15, sub = lane >> 4;
  const int co = n0 + cq * 4;
  const bool cok = co < a.Cout;
  const int phalf = (wave >> 2) * 32;                      // this wave's half of the block's 64 pixels
  f4 ps_ = {0.f, 0.f, 0.f, 0.f}, pq_ = ps_;
  const int stat_c = a.out_up != nullptr ? a.up_c0 : a.Cout;       // channels the statistics cover
  if (a.out_up != nullptr) {
    f4 rs = {0.f, 0.f, 0.f, 0.f};
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    f16x3_split_epilogue<2>(a, R, kQLd, phalf >> 4, n, h0 + mw * 4, w0, n0, lane, rs, ps_, pq_);
  } else {
    f4 rs = {0.f, 0.f, 0.f, 0.f}, bmu = rs, brs = rs, bia = rs, msc = {1.f, 1.f, 1.f, 1.f}, msh = rs;
    if (cok) rs = *(const f4*)(rinv + co) * ixs;
    if (bnb && cok) { bmu = *(const f4*)(a.bnb_mean + co); brs = *(const f4*)(a.bnb_rstd + co); }
    if (a.bias && cok) bia = *(const f4*)(a.bias + co);
    if (a.mscale && cok) { msc = *(const f4*)(a.mscale + co); msh = *(const f4*)(a.mshift + co); }
#pragma unroll 4
    for (int r = 0; r < 8; ++r) {
      const int p = phalf + r * 4 + sub;
      const int ho = h0 + mw * 4 + (p >> 4), wo = w0 + (p & 15);
      if (ho < a.Ho && wo < a.Wo && cok) {
        const size_t o = (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.Cout + co;
        f4 v = *(const f4*)(R + p * kQLd + cq * 4) * rs + bia;
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
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sv = ps_[e], qv = pq_[e];
      sv += __shfl_xor(sv, 16); qv += __shfl_xor(qv, 16);
      sv += __shfl_xor(sv, 32); qv += __shfl_xor(qv, 32);
      ps_[e] = sv; pq_[e] = qv;
    }
    __syncthreads();                             // every wave is done with the blocks
    float* red = (float*)hsm;                    // [8 waves][64][2]
    if (sub == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[(wave * 64 + cq * 4 + e) * 2] = ps_[e]; red[(wave * 64 + cq * 4 + e) * 2 + 1] = pq_[e]; }
    }
    __syncthreads();
    if (tid < 64) {
      const int c1 = n0 + tid;
      if (c1 < stat_c) {
        double sv = 0.0, qv = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { sv += (double)red[(w * 64 + tid) * 2]; qv += (double)red[(w * 64 + tid) * 2 + 1]; }
        const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
        atomicAdd(a.ssum + srep_off + c1, sv);
        atomicAdd(a.ssq + srep_off + c1, qv);
      }
    }
  }
}

// 3x3 / stride 1 / pad 1, 16-channel chunks in pairs on either side of a concat, whole 16x16 tiles not required (edges are
// masked) but at least one; the fused concat split (ConvArgs::out_up) with the boundary on a 64-channel tile
bool conv_f16x3_applicable(const ConvArgs& a) {
  return a.wu != nullptr && a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.rmul == 1 ? a.off == -1 : a.off == 1) &&
         ((a.Ctot & 31) == 0 || (a.Ctot == 16 && a.C0 == 16 && a.Cout <= 16 && !a.out_up)) && (a.C0 & 15) == 0 && (a.s0.C & 3) == 0 && (a.s1.C & 3) == 0 && (a.Cout & 3) == 0 && a.Cout >= 16 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.Ho >= 8 && a.Wo >= 16 && a.Hl < 32768 && a.Wl < 32768 &&
         (!a.out_up || (((a.Ho | a.Wo) & 1) == 0 && (a.up_c0 & 63) == 0 && a.up_c0 <= a.Cout)) &&
         (size_t)a.N * a.s0.H * a.s0.W * a.s0.C < (1ull << 31) && (size_t)a.N * a.s1.H * a.s1.W * a.s1.C < (1ull << 31);
}

template <int NJ, int NP>
static hipError_t launch_f3(const ConvArgs& a, hipStream_t st, dim3 grid, size_t lds) {
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_f16x3_kernel<NJ, NP>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3_kernel<NJ, NP>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int NP>
static hipError_t launch_f3s(const ConvArgs& a, hipStream_t st, dim3 grid, size_t lds) {
  static DevOnce lds_attr;
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_f16x3s_kernel<NP>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(42, a.flops, a.bytes, (conv_f16x3s_kernel<NP>), grid, dim3(512), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_conv_f16x3(const ConvArgs& a, hipStream_t st, int variant) {
  if (a.wu_layout == 1) return launch_conv_f16x3v2(a, st, variant >= 4 ? variant : 0);      // the bank is a conv_f16x3v2.hip one
  if (variant >= 4) return hipErrorInvalidValue;
  if (!conv_f16x3_applicable(a)) return hipErrorInvalidValue;
  if (a.out_up && (a.addend || a.mask || a.bias || a.bnb_y || (a.ssum && !a.bnb_mean) || (a.up_c0 < a.Cout && !a.out))) return hipErrorInvalidValue;
  if (a.bnb_mean && (!a.ssum || !a.ssq || !a.bnb_rstd || !(a.out_up ? a.up_mask : (a.bnb_y ? a.bnb_y : a.mask)) || a.up_accum)) return hipErrorInvalidValue;
  const int tilesN = (a.Cout + 63) / 64;
  const int tilesW = (a.Wo + kFT - 1) / kFT, tilesH = (a.Ho + kFT - 1) / kFT;
  const size_t main_lds = (size_t)2 * kFBuf * sizeof(_Float16) + 6 * 256 * sizeof(int), q_lds = (size_t)4 * 64 * 68 * sizeof(float);
  const size_t lds = main_lds > q_lds ? main_lds : q_lds;
  // 8-wave kernel (one workgroup per CU, loader waves beside the MMA waves) for launches with fewer than two workgroups per CU
  // — long-K deep layers, where staging under the MFMAs pays (layer3 93 -> 79 us, layer4 159 -> 124, 768 -> 256 at 32^2 226 -> 198);
  // the 4-wave kernel (two workgroups per CU hide each other's prologue and epilogue) for the many-tile, short-K ones
  // (layer1 89 vs 93 us, layer2 73 vs 77)
  static const bool force4 = dbg_flag("UWM_F16X3_4WAVE"), force8 = dbg_flag("UWM_F16X3_8WAVE");
  const long wgs = (long)route_N(a) * tilesH * tilesW * tilesN;      // (variant choice: ConvArgs::route_n)
  const int np = (a.nprod >= 1 && a.nprod <= 3) ? a.nprod : 3;
  if (a.Ctot == 16) {                                     // one chunk, 16 outputs: the single-chunk form
    const size_t lds1 = (size_t)kFBuf * sizeof(_Float16);          // one patch buffer (25 920 B) >= the epilogue's 4 x 64 x 20 floats
    return np == 3 ? launch_f3<1, 3>(a, st, dim3((unsigned)(a.N * tilesH * tilesW)), lds1)
         : np == 2 ? launch_f3<1, 2>(a, st, dim3((unsigned)(a.N * tilesH * tilesW)), lds1) : launch_f3<1, 1>(a, st, dim3((unsigned)(a.N * tilesH * tilesW)), lds1);
  }
  const bool four = variant == 1 || variant == 3 || (variant != 2 && (force4 || (!force8 && wgs >= 2L * device_cu_count())));
  if (variant == 3 && a.out_up) return hipErrorInvalidValue;
  // 32-channel tiles: no padded fragments on the 32-output layers — and, in the FORWARD pass only, twice the workgroups where
  // 64-channel tiles cannot give every CU one (layer4 at batch 16: 128 -> 256 workgroups, 94 vs 105 us; in the backward the idle
  // CUs of such a launch are not idle — the weight-gradient stream runs there — and the narrow tiles measured 1067 vs 1074 img/s)
  const bool fwd_alone = a.rmul == 1 && !a.xmax;
  const bool narrow = !a.out_up && (a.Cout <= 32 ? four : (fwd_alone && variant == 0 && wgs < device_cu_count()));
  if (variant == 3 || (variant == 0 && narrow)) {
    const dim3 g((unsigned)(a.N * tilesH * tilesW * ((a.Cout + 31) / 32)));
    return np == 3 ? launch_f3<2, 3>(a, st, g, lds) : np == 2 ? launch_f3<2, 2>(a, st, g, lds) : launch_f3<2, 1>(a, st, g, lds);
  }
  const dim3 g((unsigned)(a.N * tilesH * tilesW * tilesN));
  if (four) return np == 3 ? launch_f3<4, 3>(a, st, g, lds) : np == 2 ? launch_f3<4, 2>(a, st, g, lds) : launch_f3<4, 1>(a, st, g, lds);
  return np == 3 ? launch_f3s<3>(a, st, g, lds) : np == 2 ? launch_f3s<2>(a, st, g, lds) : launch_f3s<1>(a, st, g, lds);
}

}  // namespace uwm
