// Internal kernel-launch interface of libuwm (MI355X / gfx950 only).
// Layout conventions:
//   activations  NHWC fp32, channel count padded to a multiple of 4
//   weights      [Cout_rows][Kpad] fp32, k = tap*Ctot + c  (tap = r*kw + s), Kpad % 32 == 0
//   a "lazy" activation = raw conv output + per-channel (scale, shift[, relu]) applied by
//   the CONSUMER when it stages the tile (BatchNorm-apply + ReLU never make their own pass)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

namespace uwm {

struct Src {                 // one (possibly lazily transformed) NHWC activation
  const float* ptr;          // [N][H][W][C]
  const float* scale;        // per-channel scale or nullptr (identity)
  const float* shift;        // per-channel shift (valid when scale != nullptr)
  int C, H, W;               // physical dims
  int up;                    // log2 nearest-upsample factor seen by the consumer (0|1)
  int relu;                  // max(.,0) after the affine
};

struct FastDiv {             // n / d for n*d < 2^32 (k-index arithmetic only)
  unsigned mg, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d; f.mg = (d <= 1) ? 0u : (unsigned)((0x100000000ull / d) + 1ull); return f;
}

// ---- per-device launch state.  hipFuncSetAttribute and the CU count belong to a DEVICE, not to the process: one flag
// per (kernel instantiation, device), so a process that drives several GPUs sets the attribute on each of them.
struct DevOnce {
  std::atomic<unsigned long long> mask{0};          // one host thread per GPU may race here: the attribute call is idempotent, the bit set is atomic
  hipError_t set_max_lds(const void* fn, size_t bytes) {
    int dev = 0; hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release);
    return e;
  }
};
// ---- debug switches: ONE gate.  Every routing / ablation switch of the library (UWM_NO_*, UWM_TRACE_CONV, UWM_WGRAD_V1, UWM_WW_TA:
// INTEGRATION.md 2b) is read through dbg_flag() / dbg_int(), which answer "unset" unless UWM_DEBUG=1 is in the environment — a
// production process cannot be re-routed by a stray variable.  (UWM_WINOGRAD and UWM_SIDE_STREAM are documented process
// defaults of public setters, not debug switches.)
bool dbg_flag(const char* name);                // UWM_DEBUG=1 and `name` set
int dbg_int(const char* name, int dflt);        // UWM_DEBUG=1 and `name` set ? atoi : dflt
int device_cu_count();       // compute units of the CURRENT device (cached per device)

struct ConvArgs {            // implicit-GEMM conv: forward conv AND dgrad (transposed gather)
  Src s0, s1;                // channel-concat of two sources: [0,C0) from s0, [C0,Ctot) from s1
  int C0, Ctot;
  const float* w;            // [wrows][Kpad]
  int wrows, Kpad, ntaps, kw;
  int N, Ho, Wo, Cout, M;    // output [N][Ho][Wo][Cout], M = N*Ho*Wo
  int Hl, Wl;                // logical (post-upsample) input dims used for the bounds check
  int smul, rmul, off, sdiv; // hi_num = ho*smul + r*rmul + off ; hi = hi_num / sdiv (must divide)
  float* out;
  const float* bias;         // [Cout] or nullptr
  const float* addend;       // [M][Cout] added in the epilogue or nullptr
  const float* mask;         // [M][Cout]: out = (mask*mscale+mshift > 0) ? out : 0, or nullptr
  const float* mscale; const float* mshift;
  double* ssum; double* ssq; // per-channel sum / sum of squares of the output, or nullptr
  int srep, sstride;         // statistics replicas: workgroup b adds into copy (b & (srep-1)) at +copy*sstride doubles (srep: power of 2, 0/1 = none)
  // BatchNorm-backward sums fused into a dgrad epilogue (conv_wino.hip / conv_wino_x3.hip only; bnb_mean != nullptr): the
  // epilogue's masked output v IS the gradient wrt a BatchNorm output whose raw input it has just read as the ReLU mask
  // (`mask` / `up_mask`), so it accumulates ssum += v (dbeta) and ssq += v * (mask_raw - mean) * rstd (dgamma) instead of the
  // forward's (v, v^2): bn_bwd_reduce's pass over both tensors disappears
  const float* bnb_mean; const float* bnb_rstd;
  // bnb_y != nullptr: yhat is taken from THIS tensor (same [M][Cout] indexing as the output) instead of the mask tensor: the dgrad
  // that writes the gradient wrt a residual block's OUTPUT (masked by that output) carries the sums of the block's last BatchNorm,
  // whose raw input is a different tensor (conv_wino_kernel<NI> plain epilogue and conv_igemm_kernel only)
  const float* bnb_y;
  FastDiv dv_ctot, dv_kw;
  double flops;              // algorithmic FLOPs of this launch (host-side profiling only)
  double bytes;              // algorithmic HBM bytes of this launch: every operand read once + the output written once (profiling only)
  // decoder dgrad with the concat split fused into the epilogue (conv_wino.hip only): output channels [0, up_c0) are
  // summed over each 2x2 pixel block (nearest-x2 upsample backward), ReLU-masked by the low-resolution producer
  // (up_mask * up_mscale + up_mshift > 0) and written to out_up [N][Ho/2][Wo/2][up_c0]; channels [up_c0, Cout) go to
  // `out` as [N][Ho][Wo][Cout - up_c0] (may be nullptr when there are none)
  float* out_up; int up_c0; const float* up_mask; const float* up_mscale; const float* up_mshift;
  int up_accum;              // out_up += instead of = (a tensor with several consumers: UNet++)
  int pc_ntaps[4]; unsigned pc_taps[4];          // stride-2 dgrad parity classes (conv_igemm.hip; blockIdx.y = class), set by the launcher
  int live_ch;               // dgrad: channels of dY that can be non-zero (0 = all; the head's classes inside its 4 padded channels)
  const float* wu;           // Winograd-transformed weights (conv_wino.hip layout) or nullptr
  int wu_ncb;                // 16-row blocks per xi in wu
  int wino;                  // Winograd mode of this launch: 0 = process default (uwm_set_winograd), else mode + 1 (per-handle: uwm_set_winograd_mode)
  int prec;                  // 1: wu is a bf16x3 bank (conv_wino_x3.hip layout) and the launch goes to the split-bf16 kernel; 2: wu is an fp16x3 bank (conv_f16x3.hip: direct form on v_mfma_f32_16x16x32_f16, fp32-class accuracy); 0: fp32
  int wu_rinv_off;           // prec 2: float offset of the bank's 1 / row-scale array from wu
  const float* xmax;         // prec 2: 32 device floats whose maximum is max|input| (a dgrad's dY, written by bn_bwd_apply), or nullptr: the input is staged times the power of two that puts that maximum in [2^13, 2^14), undone in the epilogue
  int route_n;               // > 0: choose the kernel VARIANT as if the batch were route_n images (uwm_set_routing_batch: a small parity sample on the kernels the full batch takes); 0: N
  int ig16;                  // implicit-GEMM launches (conv_igemm.hip: stride-2 3x3, 1x1 stride 2, their dgrads) and the sub-pixel decoder conv (conv_up2_f16.hip): 1 = fp16x3 split products on v_mfma_f32_16x16x32_f16 (operands split while staging, weights times 2^12, a dgrad's dY by xmax); 0 = exact fp32
  int nprod;                 // prec 2: split products per tile — 0 / 3: hi*hi' + hi*lo' + lo*hi' (fp32-class); 2: the pixel operand (a dgrad's dY) as ONE fp16 (hi*hi' + lo_w*hi'); 1: hi*hi' only (plain fp16 products, the reference's autocast arithmetic)
  int wu_layout;             // prec 2: layout of the fp16x3 bank behind wu — 0: conv_f16x3.hip (tap pairs, 16-row fragments), 1: conv_f16x3v2.hip (taps, 32-row fragments); set by whoever packed the bank (f16x3v2_shape)
};

// batch the VARIANT choice of a launcher is made for (ConvArgs::route_n / WgradArgs::route_n; grids always use the real N)
template <class A> static inline int route_N(const A& a) { return a.route_n > 0 ? a.route_n : a.N; }
template <class A> static inline long route_M(const A& a) { return a.N > 0 ? (long)a.M / a.N * route_N(a) : (long)a.M; }

#if defined(__HIPCC__)
// Split of four fp32 values into hi / lo fp16 halves (the fp16x3 operand form): hi = rn_f16(x) (v_cvt_pk_f16_f32, two values per
// instruction), lo = rn_f16(x - hi) as ONE v_fma_mix{lo,hi}_f16 per value — fma(hi_as_f32, -1.0, x) rounded to fp16 into one half
// of the destination, the fp16 operand selected out of the packed register by op_sel (exact: an fp32 value minus its fp16 rounding
// is representable in fp32).  Results as packed pairs: {x, y}, {z, w}.
typedef unsigned uwm_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void uwm_split4(float x0, float x1, float x2, float x3, uwm_u2& hi, uwm_u2& lo) {
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  const h2_ a = {(_Float16)x0, (_Float16)x1}, b = {(_Float16)x2, (_Float16)x3};
  hi.x = __builtin_bit_cast(unsigned, a); hi.y = __builtin_bit_cast(unsigned, b);
  unsigned l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l0) : "v"(hi.x), "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l1) : "v"(hi.y), "v"(x2), "v"(x3));
  lo.x = l0; lo.y = l1;
}
#endif

// Deferred partial-sum reduces.  A split weight-gradient launch leaves `nsplit` dW-shaped partial images in scratch; adding them
// up (fixed order) used to be one small launch behind every wgrad (37 per resnet34 step, each 3x slower beside the other
// stream's kernels than alone).  With a queue attached (WgradArgs::rq) the launcher only records the job; the owner flushes the
// queue with ONE multi-job launch per backward stage (launch_wgrad_reduce_multi) — same arithmetic per job, so the result is
// bit-identical to the immediate form.  The owner hands every queued job its own scratch region (rq->used_floats).
struct ReduceJob { const float* part; float* dw; unsigned long long n4; int nsplit; unsigned block0; };
struct ReduceQueue {
  enum { kMax = 48 };
  ReduceJob j[kMax]; int n = 0; unsigned blocks = 0; size_t used_floats = 0;
};
struct ReduceJobs { ReduceJob j[ReduceQueue::kMax]; int n; };

struct WgradArgs {           // dW[co][k] += sum_m dY[m][co] * X[m][k]   (k = tap*Ctot + c)
  Src s0, s1; int C0, Ctot;
  const float* dy;           // [M][Cout]
  float* dw;                 // [wrows][Kpad], accumulated with float atomics
  int wrows, Kpad, ntaps, kw;
  int N, Ho, Wo, Cout, M;
  int Hl, Wl, stride, pad;
  int nsplit, msplit;        // pixel range per split (multiple of 32)
  float* part;               // scratch for per-split partial dW tiles (Winograd wgrad: deterministic two-stage sum) or nullptr
  size_t part_floats;        // its capacity
  int force_igemm;           // tests: 1 = never route to wgrad_patch
  FastDiv dv_ctot, dv_kw;
  double flops;              // algorithmic FLOPs of this launch (host-side profiling only)
  double bytes;              // algorithmic HBM bytes of this launch: every operand read once + the output written once (profiling only)
  int wino;                  // as ConvArgs::wino
  int prec;                  // 2: fp16x3 direct weight gradient (wgrad_f16x3.hip) where applicable; 0: fp32 kernels
  const float* xmax;         // prec 2: 32 device floats whose maximum is max|dy| (written by bn_bwd_apply): dY is staged times the power of two that puts it in [2^13, 2^14)
  ReduceQueue* rq;           // HOST pointer (never read on the device) or nullptr: queue the partial-sum reduce instead of launching it
  int cu_share;              // prec 2: 0 = one workgroup per CU; n = per n/4 of the CUs (3 when the launch runs beside the dependent chain: the rest stay free of its 768-thread workgroups)
  int route_n;               // as ConvArgs::route_n
  int nprod;                 // prec 2: as ConvArgs::nprod (2: dY as one fp16: dy_hi*x_hi + dy_hi*x_lo)
};

// ---- optional HIP-event profiler: one (start, stop) event pair per conv / wgrad launch, recorded on
// the launch stream; classes 0..5 = conv_igemm tile configs, 6..9 = wgrad tiles 64x128, 128x128, 16x256, 32x256,
// 10..13 = conv_patch BN 128, 64, 32, 16 ; 14..16 = wgrad_patch TA 16, 32, 64 ; 17..18 = conv_patch16 BN 16, 32
enum { kProfClasses = 64 };   // 60..62 = wgrad_igemm_f16x3 tiles 128x128, 128x64, 64x128 ; 45 = conv_up2_f16x3 ; 46 = conv_up2_dgrad_f16x3 ; 47 = wgrad_up2_f16x3 ; 48 = conv_c16_f16x3 ; 49 = wgrad_c16_f16x3 ; 50 = wgrad_stem_f16x3 ; 51..52 = conv_gemm_f16x3 BN 128, 64 ; 53..58 = conv_igemm_f16x3 tile configurations 0..5 ; 44 = conv_stem_f16x3 ; 43 = wgrad_f16x3 ; 42 = conv_f16x3 ; 41 = wgrad_stem ; 39..40 = wgrad_gemm TA 128, 64 ; 37..38 = conv_gemm BN 128, 64 ; 34 = conv_up2 ; 35 = conv_up2_dgrad ; 36 = wgrad_up2 ; 31 = conv_wino_x3 (bf16x3) ; 32 = wgrad_c16 ; 33 = wgrad_head ; 19..21 = conv_wino BN 64, 32, 16 ; 22..24 = wgrad_wino TA 64, 32, 16 ; 25 = conv_wino8 ; 26..29 = wgrad tiles 128x32, 128x64, 32x64, 32x128 ; 30 = conv_head
void prof_enable(bool on);
bool prof_on();
void prof_pair(int cls, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
// routing record: every UWM_LAUNCH leaves the name of the kernel it dispatched in a thread-local slot; the model copies it into its
// per-handle routing log (uwm_routing_dump) so a test / bench.py can assert WHICH kernel a layer ran on
void route_note(const char* kernel);
const char* route_last();
// One profiled launch = one (start, stop) event pair attached to the KERNEL DISPATCH itself (hipExtLaunchKernelGGL): the
// pair carries the dispatch's own begin / end timestamps — the clock rocprofv3's kernel trace reads — so a launch that
// waits for CUs behind the other stream's kernels is not charged for the wait (events recorded AROUND the launch were).
#define UWM_LAUNCH(cls, flops, bytes, kernel, grid, block, lds, st, ...)                                     \
  do {                                                                                                       \
    route_note(#kernel);                                                                                     \
    if (prof_on()) {                                                                                         \
      hipEvent_t e0_, e1_; prof_pair((cls), (flops), (bytes), &e0_, &e1_);                                   \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0_, e1_, 0, __VA_ARGS__);                         \
    } else {                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                         \
    }                                                                                                        \
  } while (0)
int  prof_collect(double* out /* [kProfClasses][4] = launches, ms, flops, bytes */);
const char* prof_class_name(int cls);

// ---- launchers (all asynchronous on `st`, no host sync, no allocation) ----
hipError_t launch_conv(const ConvArgs& a, hipStream_t st, int force_cfg = -1);
bool conv_epilogue_carries_bnb(const ConvArgs& a);      // launch_conv (auto routing) ends on a kernel whose epilogue carries the fused BatchNorm-backward sums (bnb_*); every launcher that cannot REJECTS a.bnb_mean, so a disagreement with the router is an error, not a wrong dgamma
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t st);
// 3x3 s1 p1 patch-tiled conv (conv_patch.hip); launch_conv routes to it when applicable.
// force_cfg for launch_conv: -1 auto, 0..5 conv_igemm tile config, 100+BN (116,132,164,228) conv_patch, 200 conv_patch16
bool wgrad_patch_applicable(const WgradArgs& a);
hipError_t launch_wgrad_patch(const WgradArgs& a, hipStream_t st);
bool wgrad_wino_applicable(const WgradArgs& a);            // Winograd-domain wgrad (wgrad_wino.hip)
hipError_t launch_wgrad_wino(const WgradArgs& a, hipStream_t st);
size_t wgrad_wino_scratch_floats();                         // workspace the model plans for WgradArgs::part
float* wgrad_op_scratch();                                  // the same, for the single-operator entry points (cached per device)
bool wgrad_c16_applicable(const WgradArgs& a);              // 16-channel full-resolution layers and the head (wgrad_c16.hip)
hipError_t launch_wgrad_c16(const WgradArgs& a, hipStream_t st);
hipError_t launch_wgrad_reduce(const float* part, int nsplit, size_t n4, float* dw, hipStream_t st, ReduceQueue* rq = nullptr);      // dw += sum of nsplit full-size partial images, fixed order; rq: queued, not launched
hipError_t launch_wgrad_reduce_multi(ReduceQueue& q, hipStream_t st);        // every queued job in one launch; empties the queue
bool wgrad_f16x3_applicable(const WgradArgs& a);            // fp16x3 direct weight gradient of the 3x3 / stride-1 layers (wgrad_f16x3.hip); needs a.xmax
hipError_t launch_wgrad_f16x3(const WgradArgs& a, hipStream_t st);
bool wgrad_stem_applicable(const WgradArgs& a);             // 7x7 / stride-2 / 3(4)-channel stem: compact-column wgrad (wgrad_stem.hip)
hipError_t launch_wgrad_stem(const WgradArgs& a, hipStream_t st);
bool wgrad_gemm_applicable(const WgradArgs& a);             // 1x1 / stride-1 weight gradient as a persistent LDS-DMA GEMM (wgrad_gemm.hip)
hipError_t launch_wgrad_gemm(const WgradArgs& a, hipStream_t st);
bool wgrad_gemm_preferred(const WgradArgs& a);            // applicable AND not a many-pixel / tiny-dW layer (those stay on wgrad_igemm)
bool wgrad_up2_applicable(const WgradArgs& a);              // wgrad of conv_up2.hip's layer (nearest-x2 upsampled 32-channel input, 16 outputs)
hipError_t launch_wgrad_up2(const WgradArgs& a, hipStream_t st);
bool conv_gemm_applicable(const ConvArgs& a);             // 1x1 / stride-1 conv as a persistent LDS-DMA GEMM (conv_gemm.hip); force_cfg 800 (auto tile) / 864 / 928
hipError_t launch_conv_gemm(const ConvArgs& a, hipStream_t st, int bn);
bool conv_gemm_preferred(const ConvArgs& a);              // applicable AND enough tiles to fill the chip (else the implicit GEMM's 64x64 tiles)
bool conv_up2_applicable(const ConvArgs& a);              // 3x3 over a nearest-x2 upsampled 32-channel input, 16 outputs (conv_up2.hip)
hipError_t launch_conv_up2(const ConvArgs& a, hipStream_t st);
bool conv_up2_dgrad_applicable(const ConvArgs& a);        // its dgrad wrt the low-resolution input, concat-split epilogue contract (ConvArgs::out_up)
hipError_t launch_conv_up2_dgrad(const ConvArgs& a, hipStream_t st);
// conv_up2_f16.hip: the same two launches on v_mfma_f32_16x16x32_f16 with fp16x3 split products (taken when ConvArgs::ig16 is set)
hipError_t launch_conv_up2_f16(const ConvArgs& a, hipStream_t st);
hipError_t launch_conv_up2_dgrad_f16(const ConvArgs& a, hipStream_t st);
// conv_c16_f16.hip: 3x3 from 16 to 16 channels at full resolution (decoder block 4 conv2), forward and dgrad, fp16x3 (taken when ConvArgs::ig16 is set); force_cfg 710
bool conv_c16_f16_applicable(const ConvArgs& a);
hipError_t launch_conv_c16_f16(const ConvArgs& a, hipStream_t st);
bool conv_c32_f16_applicable(const ConvArgs& a);          // the same for 32 -> 32 channels (decoder block 3 conv2); force_cfg 711
hipError_t launch_conv_c32_f16(const ConvArgs& a, hipStream_t st);
bool wgrad_up2_f16_shape(const WgradArgs& a);             // its weight gradient: 2 x 32 low-resolution tiles
int wgrad_up2_f16_parts(const WgradArgs& a);              // workgroup partials of that launch (wgrad_up2_kernel's layout)
hipError_t launch_wgrad_up2_f16(const WgradArgs& a, hipStream_t st);
bool conv_patch_applicable(const ConvArgs& a);
bool conv_patch16_applicable(const ConvArgs& a);          // 16-channel inputs: whole K in LDS (conv_patch16.hip)
hipError_t launch_conv_patch16(const ConvArgs& a, hipStream_t st);
hipError_t launch_conv_patch(const ConvArgs& a, hipStream_t st, int bn);
// Winograd F(2x2,3x3) (conv_wino.hip): needs a.wu = launch_wino_weights(a.w ...) output; force_cfg 300 (auto tile) / 300+BN
bool conv_wino_applicable(const ConvArgs& a);
hipError_t launch_conv_wino(const ConvArgs& a, hipStream_t st, int bn = 0);   // bn 8 = conv_wino8
// segmentation head (3x3, 8|16|32 channels -> <= 4 classes, bias): HBM streaming kernel (conv_head.hip); force_cfg 500
bool conv_head_applicable(const ConvArgs& a);
hipError_t launch_conv_head(const ConvArgs& a, hipStream_t st);
bool conv_head_dgrad_applicable(const ConvArgs& a);          // its dgrad (4 padded classes -> 8|16|32 channels, ReLU mask)
hipError_t launch_conv_head_dgrad(const ConvArgs& a, hipStream_t st);
bool conv_wino8_applicable(const ConvArgs& a);              // 512-thread, 16x16-pixel, 64-channel variant (conv_wino8.hip)
hipError_t launch_conv_wino8(const ConvArgs& a, hipStream_t st);
size_t wino_weights_floats(int wrows, int Ctot);
int wino_ncb(int wrows);
hipError_t launch_wino_weights(const float* w, int wrows, int Kpad, int Ctot, int mirror, float* ut, hipStream_t st);
struct WinoJob { const float* w; float* ut; int rows, chans, Kpad, mode, src_rows, pad_; };
struct WinoJobs { WinoJob j[40]; int n; };
hipError_t launch_wino_weights_multi(const WinoJobs& jobs, hipStream_t st);   // every layer's transform in one launch
// bf16x3 precision mode (conv_wino_x3.hip): split-bf16 filter banks (same size as the fp32 ones) and the conv kernel; force_cfg 400
hipError_t launch_wino_weights_x3_multi(const WinoJobs& jobs, hipStream_t st);
// fp16x3 direct convolution (conv_f16x3.hip): split-fp16 filter banks and the conv kernel; force_cfg 600
int f16x3_nj(int rows);
size_t f16x3_bank_floats(int rows, int chans);             // bank size in floats, 1 / row-scale array included
size_t f16x3_rinv_off(int rows, int chans);                 // float offset of that array in the bank
hipError_t launch_f16x3_weights_multi(const WinoJobs& jobs, hipStream_t st);      // WinoJob::pad_ = bank layout (ConvArgs::wu_layout)
bool conv_f16x3_applicable(const ConvArgs& a);
static inline __host__ __device__ size_t f16x3_rinv_off_floats(int rows, int chans) { return (size_t)(chans / 16) * 5 * (size_t)(((rows + 63) / 64) * 4) * 512; }
// conv_f16x3v2.hip: the same arithmetic on v_mfma_f32_32x32x16_f16, 8 x 32-pixel tiles (bank layout 1)
bool f16x3v2_shape(int Ho, int Wo, int rows, int chans, int dgrad);      // layers it takes (whole tiles, 32-row fragments; which of them: measured, see the function): decides the bank layout
int f16x3v2_nf(int rows);
hipError_t launch_f16x3v2_weights_multi(const WinoJobs& jobs, hipStream_t st);      // the layout-1 jobs of a job table (row scales already made)
bool conv_f16x3v2_applicable(const ConvArgs& a);
hipError_t launch_conv_f16x3v2(const ConvArgs& a, hipStream_t st, int variant = 0);      // variant: 0 auto | 4 64-channel tiles | 5 32-channel tiles
// conv_stem_f16x3.hip: the 7x7 / stride-2 ResNet stem on the fp16x3 arithmetic (one MFMA k-step per kernel row); a.wu = its bank
size_t stem_f16x3_bank_floats();
hipError_t launch_stem_f16x3_weights(const float* w, int Kpad, int cin_p, float* bank, hipStream_t st);
bool conv_stem_f16x3_applicable(const ConvArgs& a);
hipError_t launch_conv_stem_f16x3(const ConvArgs& a, hipStream_t st);
hipError_t launch_conv_f16x3(const ConvArgs& a, hipStream_t st, int variant = 0);      // variant: 0 auto | 1 four-wave kernel | 2 eight-wave kernel | 3 four-wave, 32-channel tiles (tests)
bool conv_wino_x3_applicable(const ConvArgs& a);
hipError_t launch_conv_wino_x3(const ConvArgs& a, hipStream_t st);
// process default (UWM_WINOGRAD / uwm_set_winograd): used by the single-operator entry points and by handles created later
bool winograd_enabled();
void winograd_set_mode(int mode);   // 0 off, 1 auto, 2 = tests: take the 8-wave variant wherever its shape rules allow
int winograd_mode();
// mode of ONE launch: the handle's own mode when its args carry one (wino = mode + 1), else the process default
static inline int wino_mode_of(int wino) { return wino > 0 ? wino - 1 : winograd_mode(); }

hipError_t launch_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, int CP, hipStream_t st);
hipError_t launch_bn_finalize(const double* ssum, const double* ssq, const float* gamma, const float* beta,
                              float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift,
                              int C, double count, float eps, float momentum, int update_running, hipStream_t st,
                              int nrep = 1, int rep_stride = 0);
hipError_t launch_bn_eval(const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                          float* scale, float* shift, int C, float eps, hipStream_t st);
// xn = relu(y*s2+b2 + idn), idn = id (materialised) or id*sd+bd (lazy)
hipError_t launch_residual(const float* y, const float* s2, const float* b2, const float* id, const float* sd,
                           const float* bd, float* out, size_t npix, int C, hipStream_t st);
hipError_t launch_maxpool_fwd(const Src& in, float* out, uint8_t* idx, int N, int Ho, int Wo, hipStream_t st);
// g_in = (maxpool_bwd(g_out, idx) + addend) masked by relu(in.raw*scale+shift) > 0
hipError_t launch_maxpool_bwd(const float* gout, const uint8_t* idx, const float* addend, const Src& in, float* gin,
                              int N, int Ho, int Wo, hipStream_t st, const float* bn_mean = nullptr, const float* bn_rstd = nullptr,
                              double* ssum = nullptr, double* ssq = nullptr, int srep = 0, int sstride = 0);   // bn_*: fused BatchNorm-backward sums of the masked output
// BatchNorm backward: g = grad wrt BN output (already ReLU-masked), y = raw conv output
hipError_t launch_bn_bwd_reduce(const float* g, const float* y, const float* mean, const float* rstd,
                                double* dgamma, double* dbeta, size_t npix, int C, hipStream_t st);
hipError_t launch_bn_bwd_apply(const float* g, const float* y, const float* mean, const float* rstd,
                               const float* gamma, const double* dgamma, const double* dbeta, float* dy,
                               float* gamma_grad, float* beta_grad, size_t npix, int C, hipStream_t st,
                               const double* rep = nullptr, int nrep = 0, int rep_stride = 0, hipEvent_t done = nullptr,
                               float* xmax = nullptr);   // xmax: 32 zero-initialised floats; slot (workgroup & 31) receives max|dy| of the workgroup (ConvArgs::xmax of the fp16x3 dgrad that reads dy)   // rep: fold the nrep replicas {dbeta part[C], dgamma part[C]} of a fused dgrad epilogue in the prologue (dgamma / dbeta unused)
// dcat [N][H][W][C0+C1] -> gprev [N][H/2][W/2][C0] = mask(sum 2x2), gskip [N][H][W][C1] (copy)
hipError_t launch_upsplit(const float* dcat, int N, int H, int W, int C0, int C1, float* gprev,
                          const float* pmask, const float* pscale, const float* pshift, float* gskip,
                          hipStream_t st, int accumulate_prev = 0);
// UNet++ dense skips: materialise act(src) into a channel range of a concat buffer / scatter-accumulate a channel
// range of a concat gradient back (optionally ReLU-masked by the tensor it belongs to)
hipError_t launch_concat_copy(const Src& s, size_t npix, float* dst, int Cd, int dst_off, hipStream_t st);
hipError_t launch_split_accum(const float* gcat, int Cc, int off, int C, size_t npix, float* dst, const float* m,
                              const float* mscale, const float* mshift, int accumulate, hipStream_t st);
hipError_t launch_pack_dgrad(const float* w, int Cout, int Kpad, int ntaps, int Cin, float* wd, int KpadD,
                             int CoutP, hipStream_t st);
hipError_t launch_colsum(const float* g, size_t npix, int C, float* out, double* scratch, hipStream_t st);   // out[c] += sum over pixels; scratch (colsum_scratch_doubles(C)): two-stage, fixed order; nullptr: float atomics
size_t colsum_scratch_doubles(int C);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                       float eps, float wd, float bc1, float bc2, float gscale, hipStream_t st,
                       const double* sumsq = nullptr, float max_norm = 0.f);
hipError_t launch_adam_graph(float* p, const float* g, float* m, float* v, size_t n, float* hyp /* device, 10 floats */, const double* sumsq, hipStream_t st);
hipError_t launch_sumsq(const float* g, size_t n, double* out, hipStream_t st);
hipError_t launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int first, float gscale,
                      hipStream_t st, const double* sumsq = nullptr, float max_norm = 0.f);
hipError_t launch_resize_threshold(const float* logits, int ld, int N, int h, int w, int H, int W, float thr,
                                   int apply_sigmoid, uint8_t* out, float* out_f, hipStream_t st);
hipError_t launch_scale(float* p, size_t n, float s, hipStream_t st);
hipError_t launch_absmax32(const float* x, size_t n, float* out32, hipStream_t st);      // out32[0..31] <- max|x| (slot workgroup & 31; zeroed first): the xmax contract of the fp16x3 kernels
hipError_t launch_preprocess_u8(const uint8_t* img, int N, int H, int W, int C, const float* mean, const float* std,
                                const int* flags, float* out, hipStream_t st);
hipError_t launch_preprocess_mask(const uint8_t* m, int N, int H, int W, int thr, const int* flags, uint8_t* out, hipStream_t st);

// EfficientNet MBConv pieces (mbconv.hip): swish, depthwise k x k conv (weights tap-major [k*k][C]) with static "same"
// padding (pb = pad at the begin of H and W; the end pad is implied by Ho/Wo), squeeze-and-excitation, block output with drop-connect
hipError_t launch_swish_fwd(const float* y, const float* sc, const float* sh, int C, float* out, size_t npix, hipStream_t st);
hipError_t launch_dw_fwd(const float* x, const float* w, int k, int stride, int pb, int N, int H, int W, int C,
                         int Ho, int Wo, float* y, hipStream_t st);
hipError_t launch_dw_dgrad(const float* dy, const float* w, int k, int stride, int pb, int N, int H, int W, int C,
                           int Ho, int Wo, const float* addend, float* dx, hipStream_t st);
// two-stage (partials + reduce, no global atomics): scratch holds dw_wgrad_scratch_floats(...) floats; dw += result
size_t dw_wgrad_scratch_floats(int k, int N, int C, int Ho, int Wo);
hipError_t launch_dw_wgrad(const float* x, const float* dy, int k, int stride, int pb, int N, int H, int W, int C,
                           int Ho, int Wo, float* dw, float* scratch, hipStream_t st);
// BatchNorm backward behind swish [and the SE product]: swish_bwd fused into both BatchNorm-backward passes
hipError_t launch_bn_bwd_act(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma, const float* scale,
                             const float* shift, const float* se_s, const float* gpool, int N, size_t hw, double* dgamma, double* dbeta,
                             float* dy, float* gamma_grad, float* beta_grad, int C, hipStream_t st);
hipError_t launch_colstats(const float* y, size_t npix, int C, double* ssum, double* ssq, hipStream_t st);
// out[n][c] = scale * sum_hw a[n][hw][c] (* b[n][hw][c]); deterministic two-stage sum, part: se_reduce_scratch_floats(N, C) floats
size_t se_reduce_scratch_floats(int N, int C);
hipError_t launch_se_reduce_hw(const float* a, const float* b, int N, size_t hw, int C, float scale, float* out, float* part, hipStream_t st);
hipError_t launch_swish_pool(const float* y, const float* sc, const float* sh, float* act_out, int N, size_t hw, int C, float* pool,
                             float* part, hipStream_t st);
hipError_t launch_se_fc_fwd(const float* pool, const float* w1, const float* b1, int K1pad, const float* w2, const float* b2,
                            int K2pad, int N, int C, int nsq, float* hpre, float* hid /* scratch [N][nsq] */, float* s, hipStream_t st);
// gs [N][C] is overwritten (gz2); acc1: scratch [N][nsq] zeroed by the caller; parameter gradients are plain stores
hipError_t launch_se_fc_bwd(float* gs, const float* s, const float* hpre, const float* pool, const float* w1, int K1pad,
                            const float* w2, int K2pad, int N, int C, int nsq, float* gpool, float* acc1, float* gw1,
                            float* gb1, float* gw2, float* gb2, hipStream_t st);
hipError_t launch_se_scale(const float* a, const float* s, int N, size_t hw, int C, float* out, hipStream_t st);
// out = (y*sc+sh) * rowscale[n] + id   (rowscale / id may be nullptr)
hipError_t launch_mb_out(const float* y, const float* sc, const float* sh, const float* rowscale, const float* id, int N, size_t hw,
                         int C, float* out, hipStream_t st);
hipError_t launch_rowscale(const float* g, const float* rowscale, int N, size_t hw, int C, float* out, hipStream_t st);

// loss / metrics (loss.hip)
hipError_t launch_loss(const float* logits, int ld, const void* target, int tdtype, size_t npix_total,
                       float w_dice, float w_bce, float smooth, float eps, double* scratch4, float* loss_out3,
                       float* dlogits, int ldd, float grad_scale, hipStream_t st);
hipError_t launch_loss_sums(const float* logits, int ld, const void* target, int tdtype, size_t n, double* scratch4, hipStream_t st);
hipError_t launch_loss_apply(const float* logits, int ld, const void* target, int tdtype, size_t n, double ntotal, float w_dice, float w_bce,
                             float smooth, float eps, const double* scratch4, float* loss_out3, float* dlogits, int ldd,
                             float grad_scale, hipStream_t st);
hipError_t launch_stats(const float* logits, int ld, const void* target, int tdtype, int N, size_t hw,
                        float thr, int apply_sigmoid, long long* out4, hipStream_t st);
hipError_t launch_threshold(const float* logits, int ld, size_t npix, float thr, int apply_sigmoid,
                            uint8_t* out, hipStream_t st);

}  // namespace uwm
