// 3x3 / stride 1 / pad 1 convolution (forward AND dgrad) for gfx950 on v_mfma_f32_16x16x4_f32 with a
// spatially tiled input PATCH in LDS.
//
// A workgroup owns an 8x16 block of output pixels of one image and BN output channels.  Per 32-channel
// chunk of the (concatenated) input it stages the 10x18-pixel halo patch ONCE — BatchNorm scale/shift +
// ReLU, nearest x2 upsampling, channel concat and zero padding applied while staging — and then takes
// all nine taps from it: tap (r,s) is only an LDS address offset for the MFMA B-fragment reads.
// Compared with the flattened implicit GEMM (conv_igemm.hip) the loader's index/transform work and the
// L2->LDS traffic of the activation operand drop ~6x, which is what bounded that kernel
// (profiles/r01_c_pmc_*: 50-70 % issue-stall at ~55 % MFMA busy).
//
// K order inside the kernel is (channel chunk, tap); the weight panel keeps its [Cout][tap*Ctot + c]
// layout, so a (tap, chunk) slice is still 32 contiguous floats per output channel.
// dgrad: same kernel over the [Cin][tap][Cout] repack with mirrored tap offsets (rr = 2-r, ss = 2-s).
#include "uwm_kernels.h"
#include <cstdlib>

namespace uwm {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 16, kPH = kTH + 2, kPW = kTW + 2, kPP = kPH * kPW;   // 180 patch pixels
constexpr int kPUnits = kPP * 8;                                                // 1440 16-byte units
constexpr int kPRounds = (kPUnits + 255) / 256;                                 // 6

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, BN >= 128 ? 2 : 3) void conv_patch_kernel(const ConvArgs a) {
  constexpr int MI = kTH / WM;            // one MFMA M-tile = one 16-pixel row of the block
  constexpr int NI = BN / WN / 16;
  constexpr int WR = (BN + 31) / 32;
  static_assert(WM * WN == 4 && MI >= 1 && NI >= 1, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ps = smem;                        // [180][32]   (single-buffered)
  float* const Ws = smem + kPP * 32;             // [2][BN][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // one tile per workgroup (measured: a persistent multi-tile walk with cross-tile prefetch was ~8 %
  // slower — dynamic dispatch balances and de-synchronises workgroups better); XCD-aware bijective remap
  const unsigned nblk = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const int tn = tile % tilesN; tile /= tilesN;
  const int tw = tile % tilesW; tile /= tilesW;
  const int th = tile % tilesH; const int n = tile / tilesH;
  const int n0 = tn * BN, h0 = th * kTH, w0 = tw * kTW;

  const int unit = tid & 7, r0 = tid >> 3;
  const bool mirror = a.rmul < 0;                // dgrad: tap (r,s) reads patch (ty+2-r, tx+2-s)

  f4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- patch piece: one 16-byte unit per thread per round
  f4 pv, psc, psh; bool pact = false, pok = false, phas = false; int prelu = 0, ppos = 0;
  auto patch_load = [&](int cc, int round, bool enable) {
    const int u = round * 256 + tid;
    pact = enable && u < kPUnits;
    const int pp = u >> 3, chu = u & 7;
    const int py = pp / kPW, px = pp - py * kPW;
    const int hl = h0 - 1 + py, wl = w0 - 1 + px;
    const int c = cc * 32 + chu * 4;
    const bool first = c < a.C0;
    const float* sp = first ? a.s0.ptr : a.s1.ptr;
    const float* ssc = first ? a.s0.scale : a.s1.scale;
    const float* ssh = first ? a.s0.shift : a.s1.shift;
    const int sC = first ? a.s0.C : a.s1.C, sH = first ? a.s0.H : a.s1.H, sW = first ? a.s0.W : a.s1.W;
    const int sup = first ? a.s0.up : a.s1.up;
    prelu = first ? a.s0.relu : a.s1.relu;
    const int cl = first ? c : c - a.C0;
    pok = pact && hl >= 0 && hl < a.Hl && wl >= 0 && wl < a.Wl;
    phas = (ssc != nullptr) && pok;
    if (phas) { psc = *(const f4*)(ssc + cl); psh = *(const f4*)(ssh + cl); }
    const float* p = sp + ((size_t)((size_t)n * sH + (hl >> sup)) * sW + (wl >> sup)) * sC + cl;
    pv = pok ? *(const f4*)p : (f4){0.f, 0.f, 0.f, 0.f};
    ppos = pp * 32 + ((chu ^ ((pp >> 1) & 7)) << 2);
  };
  auto patch_store = [&](int buf) {
    if (!pact) return;
    f4 v = pv;
    if (phas) {
      v = v * psc + psh;
      if (prelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    *(f4*)(Ps + buf * kPP * 32 + ppos) = v;
  };

  // ---- weight tile of one (chunk, tap) step
  f4 wr[WR];
  auto w_load = [&](int cc, int tap) {
    const int k0 = tap * a.Ctot + cc * 32 + unit * 4;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int row = n0 + r0 + 32 * i;
      const bool v = (row < a.wrows) && (BN >= 32 || r0 < BN);
      wr[i] = v ? *(const f4*)(a.w + (size_t)row * a.Kpad + k0) : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto w_store = [&](int buf) {
    float* ws = Ws + buf * BN * 32;
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int row = r0 + 32 * i;
      if (BN >= 32 || r0 < BN) *(f4*)(ws + row * 32 + ((unit ^ ((row >> 1) & 7)) << 2)) = wr[i];
    }
  };

  const int nchunk = a.Ctot >> 5;
  // prologue: whole patch of chunk 0 (all rounds in flight together) + weights of step 0
  f4 qv[kPRounds], qsc, qsh; int qpos[kPRounds]; unsigned qact = 0, qhas = 0; int qrelu = 0;
  {
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      patch_load(0, rd, true);
      qv[rd] = pv; qpos[rd] = ppos;
      qact |= (pact ? 1u : 0u) << rd; qhas |= (phas ? 1u : 0u) << rd;
    }
    qsc = psc; qsh = psh; qrelu = prelu;          // channel unit (tid & 7) is the same for every round
    w_load(0, 0);
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      pv = qv[rd]; ppos = qpos[rd]; pact = (qact >> rd) & 1u; phas = (qhas >> rd) & 1u;
      patch_store(0);
    }
    w_store(0);
  }
  __syncthreads();

  const int lrow = lane & 15, lq = lane >> 4;
  // The patch is SINGLE-buffered (39 KB of LDS per workgroup for BN=64 -> 4 workgroups per CU instead of 2):
  // the six pieces of the next chunk are fetched into registers during taps 3..8 and written after the
  // chunk's last barrier.  The nine taps are unrolled, so tap offsets and register indices are static.
  auto step_body = [&](int cc, int tap, bool next_chunk) {
    const bool more = next_chunk || tap < 8;
    w_load(tap == 8 ? (next_chunk ? cc + 1 : 0) : cc, tap == 8 ? 0 : (more ? tap + 1 : 0));
    const int r = tap / 3, s = tap - r * 3;
    const int rr = mirror ? 2 - r : r, ss = mirror ? 2 - s : s;
    const float* ws = Ws + ((cc + tap) & 1) * BN * 32;       // step = 9*cc + tap, 9 is odd
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      f4 xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int pp = (wm * MI + i + rr) * kPW + lrow + ss;
        xf[i] = *(const f4*)(Ps + pp * 32 + (((k16 * 4 + lq) ^ ((pp >> 1) & 7)) << 2));
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
    // keep every consumer of this step's global loads BELOW the MFMA block (hipcc otherwise hoists the
    // BatchNorm-transform math above the MFMAs and waits vmcnt(0) right after issuing the loads)
    __builtin_amdgcn_sched_barrier(0);
    w_store((cc + tap + 1) & 1);
    __syncthreads();
  };

  for (int cc = 0; cc < nchunk; ++cc) {
    const bool next_chunk = cc + 1 < nchunk;
#pragma unroll 1
    for (int tap = 0; tap < 6; ++tap) step_body(cc, tap, next_chunk);
    // all six pieces of the NEXT chunk's patch are requested here, as straight-line code into named registers;
    // they are consumed after tap 8's barrier, i.e. three steps (~6k cycles) later
    qact = 0; qhas = 0;
#pragma unroll
    for (int rd = 0; rd < kPRounds; ++rd) {
      patch_load(next_chunk ? cc + 1 : 0, rd, next_chunk);
      qv[rd] = pv; qpos[rd] = ppos;
      qact |= (pact ? 1u : 0u) << rd; qhas |= (phas ? 1u : 0u) << rd;
    }
    qsc = psc; qsh = psh; qrelu = prelu;
#pragma unroll 1
    for (int tap = 6; tap < 9; ++tap) step_body(cc, tap, next_chunk);
    if (next_chunk) {                               // every wave has passed the barrier of tap 8: patch(cc) is dead
#pragma unroll
      for (int rd = 0; rd < kPRounds; ++rd) {
        pv = qv[rd]; ppos = qpos[rd]; psc = qsc; psh = qsh; prelu = qrelu;
        pact = (qact >> rd) & 1u; phas = (qhas >> rd) & 1u;
        patch_store(0);
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue (same contract as conv_igemm_kernel) ----------------
  const bool do_stats = a.ssum != nullptr;
  // BatchNorm statistics go to one of a.srep copies (few-channel layers launch tens of thousands of workgroups:
  // fp64 atomics on the same 2*C addresses serialise), bn_finalize adds the copies
  const size_t srep_off = a.srep > 1 ? (size_t)(blockIdx.x & (unsigned)(a.srep - 1)) * a.sstride : 0;
  f4 ps_[NI], pq_[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { ps_[j] = (f4){0.f, 0.f, 0.f, 0.f}; pq_[j] = ps_[j]; }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int ho = h0 + wm * MI + i, wo = w0 + lrow;
    const bool pin = ho < a.Ho && wo < a.Wo;
    const size_t m = ((size_t)n * a.Ho + ho) * a.Wo + wo;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int co = n0 + (wn * NI + j) * 16 + lq * 4;
      if (pin && co < a.Cout) {
        f4 v = acc[i][j];
        const size_t o = m * a.Cout + co;
        if (a.bias) v += *(const f4*)(a.bias + co);
        if (a.addend) v += *(const f4*)(a.addend + o);
        if (a.mask) {
          f4 mk = *(const f4*)(a.mask + o);
          if (a.mscale) mk = mk * *(const f4*)(a.mscale + co) + *(const f4*)(a.mshift + co);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        *(f4*)(a.out + o) = v;
        ps_[j] += v; pq_[j] += v * v;
      }
    }
  }
  if (do_stats) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sv = ps_[j][e], qv = pq_[j][e];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { sv += __shfl_xor(sv, d); qv += __shfl_xor(qv, d); }
        ps_[j][e] = sv; pq_[j][e] = qv;
      }
    float* red = smem;                       // [WM][BN][2]; main-loop LDS is dead after the last barrier
    if (lrow == 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = (wn * NI + j) * 16 + lq * 4 + e;
          red[(wm * BN + cl) * 2 + 0] = ps_[j][e];
          red[(wm * BN + cl) * 2 + 1] = pq_[j][e];
        }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = n0 + tid;
      if (co < a.Cout) {
        double sv = 0.0, qv = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) { sv += (double)red[(w * BN + tid) * 2]; qv += (double)red[(w * BN + tid) * 2 + 1]; }
        atomicAdd(a.ssum + srep_off + co, sv);
        atomicAdd(a.ssq + srep_off + co, qv);
      }
    }
  }
}

template <int BN, int WM, int WN>
static hipError_t launch_p(const ConvArgs& a, hipStream_t st, int cls) {
  const int tilesN = (a.Cout + BN - 1) / BN;
  const int tilesW = (a.Wo + kTW - 1) / kTW, tilesH = (a.Ho + kTH - 1) / kTH;
  const size_t lds = (size_t)(kPP * 32 + 2 * BN * 32) * sizeof(float);
  static DevOnce lds_attr;                  // hipFuncSetAttribute is per device
  { hipError_t e = lds_attr.set_max_lds((const void*)conv_patch_kernel<BN, WM, WN>, lds); if (e != hipSuccess) return e; }
  UWM_LAUNCH(cls, a.flops, a.bytes, (conv_patch_kernel<BN, WM, WN>), dim3((unsigned)(a.N * tilesH * tilesW * tilesN)), dim3(256), lds, st, a);
  return hipGetLastError();
}

bool conv_patch_applicable(const ConvArgs& a) {
  return a.ntaps == 9 && a.kw == 3 && a.smul == 1 && a.sdiv == 1 && (a.off == -1 || a.off == 1) &&
         (a.rmul == 1 ? a.off == -1 : a.off == 1) && (a.Ctot & 31) == 0 && (a.C0 & 31) == 0 &&
         a.Hl == a.Ho && a.Wl == a.Wo && a.Ho >= kTH && a.Wo >= kTW;
}

// bn: 128 | 64 | 32 | 16 output channels per workgroup
hipError_t launch_conv_patch(const ConvArgs& a, hipStream_t st, int bn) {
  if (!conv_patch_applicable(a) || (a.Cout & 3)) return hipErrorInvalidValue;
  if (a.bnb_mean) return hipErrorInvalidValue;      // no fused BatchNorm-backward sums in this epilogue: it would write (v, v^2) into the replicas
  if (bn <= 0) bn = a.Cout >= 128 ? 128 : (a.Cout > 32 ? 64 : (a.Cout > 16 ? 32 : 16));
  switch (bn) {
    case 128: return launch_p<128, 2, 2>(a, st, 10);
    case 64: return launch_p<64, 2, 2>(a, st, 11);
    case 32: return launch_p<32, 4, 1>(a, st, 12);
    case 16: return launch_p<16, 4, 1>(a, st, 13);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace uwm
