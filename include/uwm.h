/* libuwm — C ABI of the MI355X-native U-Net watermark-segmentation hot path.
 *
 * This is the drop-in boundary for ONE path of Dave-he/unet-watermark: the forward / backward of
 * smp.Unet | smp.UnetPlusPlus (encoders resnet18 | resnet34 | resnet50 | efficientnet-b4; UnetPlusPlus is the
 * reference's default MODEL.NAME, src/configs/config.py:15) and its Dice + BCE loss, which the reference reaches through
 *   model = smp.Unet(**kwargs)                /root/reference/src/models/unet_model.py:17-27,64-71,93-120
 *   outputs = model(images)                    /root/reference/src/train.py:91,100,142 ; src/predict.py:339,611
 *   loss = criterion(outputs, masks)           /root/reference/src/train.py:94,103 ; src/utils/losses.py:11-52
 *   loss.backward(); optimizer.step()          /root/reference/src/train.py:96-98,104-105
 *   metrics(sigmoid(outputs), masks)           /root/reference/src/train.py:110-117 ; src/utils/metrics.py:11-37
 *   (mask > THRESHOLD) * 255                   /root/reference/src/predict.py:614-625
 * plus the data-parallel gradient exchange the reference lacks: uwm_allreduce_grads over RCCL.
 * The reference is pure Python and has no FFI of its own; INTEGRATION.md shows the ctypes stub a
 * maintainer adds (unet-watermark_amd/_lib.py is that stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every device buffer is CALLER-OWNED (e.g. a torch tensor's
 *     data_ptr()); the library never allocates or frees device memory, never synchronises the
 *     host and enqueues all work on the caller's stream (a hipStream_t passed as void*).
 *   - every function returns 0 on success, non-zero on failure; uwm_last_error() then returns a
 *     thread-local message.  A handle is bound to one device — the device its arenas live on (uwm_bind reads it from
 *     the parameter pointer); uwm_forward / uwm_backward / uwm_allreduce_grads make that device current for the
 *     duration of the call, so the caller's current device need not match.  The free functions (uwm_loss, uwm_adam,
 *     uwm_stats ...) launch on the caller's stream and expect that stream's device to be current.  Not re-entrant.
 *   - activations are NHWC fp32 with channels padded to a multiple of 4; logits are returned as
 *     [N][H][W][CP], CP = uwm_logits_channels() (class k at channel k).
 *   - parameters live in ONE flat fp32 arena (caller-owned) whose layout the library defines:
 *     uwm_tensor_info_get() gives, per smp-compatible state_dict key, the arena offset plus logical
 *     OIHW shape and element strides (convolution weights are stored [O][kh][kw][I] with each
 *     output-channel row padded to a multiple of 32 floats).  Gradients use the same layout in a
 *     second arena; BatchNorm running statistics live in a third ("buffer") arena.
 */
#ifndef UWM_H
#define UWM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uwm_model* uwm_handle;
typedef void* uwm_stream;              /* hipStream_t */

enum { UWM_ENC_RESNET18 = 18, UWM_ENC_RESNET34 = 34, UWM_ENC_RESNET50 = 50,     /* 50: Bottleneck blocks (unet_watermark_large.yaml) */
       UWM_ENC_EFFICIENTNET_B4 = 104 };   /* MBConv blocks (README.md:173-176 / BASELINE config 4) */
enum { UWM_ARCH_UNET = 0, UWM_ARCH_UNETPLUSPLUS = 1 };   /* smp.Unet | smp.UnetPlusPlus (the reference's default MODEL.NAME, src/configs/config.py:15) */
enum { UWM_T_F32 = 0, UWM_T_I64 = 1, UWM_T_U8 = 2, UWM_T_I32 = 3 };          /* target dtypes */
enum { UWM_KIND_CONV_W = 0, UWM_KIND_BIAS = 1, UWM_KIND_BN_GAMMA = 2, UWM_KIND_BN_BETA = 3,
       UWM_KIND_BN_MEAN = 4, UWM_KIND_BN_VAR = 5 };
enum { UWM_ARENA_PARAM = 0, UWM_ARENA_BUFFER = 1 };
enum { UWM_PREC_F32 = 0, UWM_PREC_BF16X3 = 1, UWM_PREC_BF16X3_ALL = 2, UWM_PREC_F16X3 = 3, UWM_PREC_F16X3_ALL = 4,
       UWM_PREC_F16X1 = 5, UWM_PREC_F16X3_BWD2 = 6 };       /* uwm_set_precision */

/* mirrors smp.Unet(encoder_name, encoder_depth=5, decoder_channels, in_channels, classes) */
typedef struct {
  int encoder;                 /* UWM_ENC_* */
  int in_channels;             /* 1..4 */
  int classes;                 /* >= 1 */
  int decoder_channels[5];     /* e.g. 256,128,64,32,16 ; each a multiple of 4 */
  float bn_eps;                /* 1e-5 */
  float bn_momentum;           /* 0.1 */
  int arch;                    /* UWM_ARCH_* ; decoder of src/models/unet_model.py:19 (create_model_from_config) */
} uwm_unet_desc;

typedef struct {
  char name[96];               /* smp state_dict key, e.g. "encoder.layer1.0.conv1.weight" */
  int kind;                    /* UWM_KIND_* */
  int arena;                   /* UWM_ARENA_PARAM | UWM_ARENA_BUFFER */
  int ndim;                    /* 4 for conv weights, 1 otherwise */
  long long offset;            /* element offset into the arena */
  long long shape[4];          /* logical shape (OIHW for conv weights) */
  long long stride[4];         /* element strides of that logical view */
} uwm_tensor_info;

const char* uwm_last_error(void);
int uwm_version(void);

int  uwm_create(const uwm_unet_desc* desc, uwm_handle* out);
void uwm_destroy(uwm_handle h);

long long uwm_param_arena_floats(uwm_handle h);     /* incl. padding; padding must stay 0 */
long long uwm_buffer_arena_floats(uwm_handle h);
long long uwm_param_count(uwm_handle h);            /* logical number of trainable scalars */
int  uwm_num_tensors(uwm_handle h);
int  uwm_tensor_info_get(uwm_handle h, int index, uwm_tensor_info* out);
int  uwm_logits_channels(uwm_handle h);

/* number of backward stages (gradient buckets) and the arena range [begin,end) each one
 * completes; stage 0 = head+decoder, then the encoder from its deepest group to the stem (ResNet: layer4, layer3,
 * layer2, layer1+stem; EfficientNet-b4: blocks 22-31, 10-21, 6-9, 0-5+stem). */
int  uwm_num_stages(uwm_handle h);
int  uwm_stage_range(uwm_handle h, int stage, long long* begin, long long* end);

/* Bind caller-owned device arenas.  grads may be NULL for inference-only use. */
int  uwm_bind(uwm_handle h, float* params, float* grads, float* buffers);

size_t uwm_workspace_bytes(uwm_handle h, int N, int H, int W, int training);
/* algorithmic (direct-convolution) FLOPs per image at H x W: forward, and forward + backward (= 3x forward minus the
 * stem's dgrad) — SURVEY.md 8(d)'s roofline numerator (62.512 / 186.303 GFLOP for Unet-resnet34 at 512x512) */
int  uwm_conv_flops(uwm_handle h, int H, int W, double* fwd, double* fwd_bwd);

/* logits[N][H][W][CP] = Unet(x[N][Cin][H][W]).  training!=0: BatchNorm uses batch statistics,
 * updates running stats, and the workspace keeps what uwm_backward needs.  H, W % 32 == 0. */
int  uwm_forward(uwm_handle h, const float* x_nchw, float* logits, void* workspace, size_t workspace_bytes,
                 int N, int H, int W, int training, uwm_stream stream);

/* Backward of the last training forward held in `workspace`; writes (overwrites) the gradient arena
 * ranges of stages [stage_begin, stage_end).  Call with (0, uwm_num_stages) for everything, or stage
 * by stage to overlap gradient all-reduce with the rest of the backward. */
int  uwm_backward(uwm_handle h, const float* dlogits, void* workspace, int stage_begin, int stage_end,
                  uwm_stream stream);

/* loss = w_dice*Dice(logits,target) + w_bce*BCEWithLogits(logits,target) over class channel 0.
 * logits [npix][ld]; target [npix] of dtype target_dtype; scratch: >= 64 bytes device memory;
 * loss_out: 3 device floats {total, dice, bce}; dlogits [npix][ldd] (may be NULL) = grad_scale*dL/dlogits. */
int  uwm_loss(const float* logits, int ld, const void* target, int target_dtype, long long npix,
              float w_dice, float w_bce, float smooth, float eps, void* scratch, float* loss_out,
              float* dlogits, int ldd, float grad_scale, uwm_stream stream);

/* The two halves of uwm_loss, for the data-parallel "global Dice" of SURVEY.md 8(e) (Dice is a ratio of batch sums, so the
 * mean of per-rank Dice losses is NOT the Dice of the global batch): uwm_loss_sums leaves the local sums
 * {sum p*t, sum p, sum t, sum bce} as four doubles in scratch[0..3]; the host all-reduces (SUM) those 32 bytes over the ranks
 * (ncclAllReduce / torch.distributed on the same stream); uwm_loss_apply then evaluates loss_out and dlogits of the GLOBAL
 * batch, npix_total = pixels over all ranks.  The parameter gradients of the ranks must then be SUMMED, not averaged: pass
 * grad_scale = world_size here when the exchange averages.  npix_total == npix reproduces uwm_loss. */
int  uwm_loss_sums(const float* logits, int ld, const void* target, int target_dtype, long long npix, void* scratch,
                   uwm_stream stream);
int  uwm_loss_apply(const float* logits, int ld, const void* target, int target_dtype, long long npix, long long npix_total,
                    float w_dice, float w_bce, float smooth, float eps, const void* scratch, float* loss_out,
                    float* dlogits, int ldd, float grad_scale, uwm_stream stream);

/* out[N][4] int64 = tp, fp, fn, tn of (v >= threshold), v = sigmoid(logit) if apply_sigmoid else logit */
int  uwm_stats(const float* logits, int ld, const void* target, int target_dtype, int N, long long hw,
               float threshold, int apply_sigmoid, long long* out, uwm_stream stream);
/* mask[npix] uint8 = (v > threshold) ? 255 : 0 */
int  uwm_threshold(const float* logits, int ld, long long npix, float threshold, int apply_sigmoid,
                   uint8_t* mask, uwm_stream stream);

/* torch.optim.Adam (coupled weight decay) over a flat range; step = 1-based step count */
int  uwm_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
              float eps, float weight_decay, long long step, float grad_scale, uwm_stream stream);
/* same, preceded by global-norm gradient clipping (torch.nn.utils.clip_grad_norm_(params, max_norm)): the norm of
 * grad_scale*g over the whole range is reduced on the device (scratch: >= 8 bytes) and folded into the update */
int  uwm_adam_clip(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, long long step, float grad_scale, float max_norm, void* scratch,
                   uwm_stream stream);
/* The same update with every hyper-parameter in DEVICE memory, for a hipGraph-captured train step (one captured launch must
 * serve every step): hyper = 10 floats {lr, beta1, beta2, eps, weight_decay, grad_scale, max_norm, step, -, -}; the call
 * advances hyper[7] (the step count of the update it performs: write step - 1 there before the first call / after a restore)
 * and fills the two bias-correction slots itself.  clip_scratch != NULL (>= 8 bytes): global-norm clipping to hyper[6]. */
int  uwm_adam_graph(float* p, const float* g, float* m, float* v, long long n, float* hyper, void* clip_scratch,
                    uwm_stream stream);
/* torch.optim.SGD(lr, momentum, weight_decay) (coupled L2, dampening 0; the reference's OPTIMIZER.NAME == "SGD" branch,
 * /root/reference/src/train.py:272-278) over a flat range: buf = step == 1 ? g' : momentum*buf + g', p -= lr*buf with
 * g' = grad_scale*g + weight_decay*p; max_norm > 0 adds global-norm clipping as uwm_adam_clip (scratch >= 8 bytes). */
int  uwm_sgd(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay,
             long long step, float grad_scale, float max_norm, void* scratch, uwm_stream stream);
int  uwm_scale(float* p, long long n, float s, uwm_stream stream);
/* Staged backward under data parallelism: uwm_backward runs its weight-gradient kernels on an internal side stream and,
 * by default, makes the caller's stream wait for them before it returns.  With a join stream set (the stream the
 * gradient all-reduces are issued on), the stages before the last make THAT stream wait instead, so the caller's
 * stream continues into the next stage while the side stream drains; the last stage joins both.  NULL restores the
 * default. */
int  uwm_set_join_stream(uwm_handle h, uwm_stream stream);
/* The data-parallel exchange itself (nothing comparable exists in the reference, which has no DDP):
 * SUM all-reduce, in place, of the gradient-arena ranges of backward stages [stage_begin, stage_end) over the RCCL
 * communicator `nccl_comm` (an ncclComm_t), one ncclAllReduce per stage, enqueued on `stream` (the communication
 * stream — order it behind uwm_backward of those stages with an event, or hand it to uwm_set_join_stream).  Averaging
 * is the optimizer's grad_scale = 1/world.  RCCL is not linked: ncclAllReduce is looked up in the host process (the
 * library that created the communicator), else librccl.so.1 is opened. */
int  uwm_allreduce_grads(uwm_handle h, void* nccl_comm, int stage_begin, int stage_end, uwm_stream stream);
float* uwm_grad_arena(uwm_handle h);     /* the bound gradient arena (NULL before uwm_bind) */
/* Input pipeline on the device (src/utils/dataset.py:298-395 get_*_transform tails): uint8 HWC images [N][H][W][C] ->
 * Normalize(mean, std) of x/255 as NCHW fp32 (what uwm_forward takes); uint8 masks [N][H][W] -> (m > threshold) as
 * uint8 {0,1} (what uwm_loss takes).  flags (device int[N] or NULL): bit0 HorizontalFlip, bit1 VerticalFlip, bits 2-3
 * k of RandomRotate90 (counter-clockwise, needs H == W), applied flips first, then the rotation — the same flags on
 * image and mask keep them aligned.  mean/std are host pointers (C values). */
int  uwm_preprocess_u8(const uint8_t* images, int N, int H, int W, int C, const float* mean, const float* std,
                       const int* flags, float* out_nchw, uwm_stream stream);
int  uwm_preprocess_mask_u8(const uint8_t* masks, int N, int H, int W, int threshold, const int* flags, uint8_t* out,
                            uwm_stream stream);
/* 3x3/stride-1 convolutions (forward, dgrad and weight gradient) run as Winograd F(2x2,3x3) on the fp32 MFMA path by
 * default (2.25x fewer multiplies, results within a few fp32 ulps of the direct form); mode 0 selects the direct
 * kernels everywhere; 2 (tests) prefers the 512-thread Winograd variant wherever its shape rules allow, whatever the
 * launch size.  uwm_set_winograd_mode is PER HANDLE.  uwm_set_winograd sets the process default, which the
 * single-operator entry points (uwm_op_*) use and which a handle takes at uwm_create (also UWM_WINOGRAD=0 in the
 * environment); it does not change existing handles. */
int  uwm_set_winograd(int on);
int  uwm_set_winograd_mode(uwm_handle h, int mode);
int  uwm_get_winograd_mode(uwm_handle h);
/* Precision mode, per handle.  UWM_PREC_F32 (default): every product on the exact-fp32 matrix instruction.
 * "bf16x3" arithmetic takes a product as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi over bf16 halves (hi = bf16(x), lo = bf16(x - hi))
 * of the fp32 operands with fp32 accumulation: ~17 mantissa bits per operand on 3 bf16 matrix instructions at 16x the fp32
 * rate (conv_wino_x3.hip).  Parameters, activations, accumulators, BatchNorm, loss, weight gradients and optimizer state
 * stay fp32 in every mode.
 *   UWM_PREC_BF16X3      the BACKWARD data-gradient (dgrad) products of the 3x3 stride-1 layers with whole 16-channel
 *                        chunks.  The forward is untouched, so logits are the fp32 mode's bit for bit and the 1e-3 bar
 *                        against the fp32 CPU reference holds at every depth; gradients meet the fp32 mode's bars.
 *   UWM_PREC_BF16X3_ALL  forward products of those layers as well.  Measured logit error vs the fp32 CPU reference:
 *                        < 1e-3 on resnet18 / efficientnet-b4, 1.6e-3 on resnet34 (2x256x192) — outside BASELINE's bar on
 *                        the deeper encoders, offered for what the reference itself does on a GPU: reduced-precision
 *                        training (fp16 autocast + GradScaler, /root/reference/src/train.py:75,89-98).
 * "fp16x3" arithmetic (conv_f16x3.hip) is the same three-term product over FP16 halves (hi = fp16(x), lo = fp16(x - hi): 22
 * mantissa bits per operand, relative error 2^-22 per product against 2^-24 for fp32) on gfx950's
 * v_mfma_f32_16x16x32_f16, direct-form convolution, fp32 accumulation; fp16's exponent range is covered by exact
 * power-of-two scaling of every filter row (and of a dgrad's dY).  fp32-class accuracy: it meets the fp32 mode's bars.
 *   UWM_PREC_F16X3       the FORWARD products of the 3x3 stride-1 layers with channels % 32 == 0
 *   UWM_PREC_F16X3_ALL   their data-gradient and weight-gradient products as well
 * Two REDUCED-precision modes on the same kernels (never the default, never bench.py's headline; reported under alt_modes):
 *   UWM_PREC_F16X1       one product per tile, hi * hi' — plain fp16 products with fp32 accumulation, fp32 storage, BatchNorm and
 *                        optimizer: the arithmetic of the reference's own GPU path (torch.autocast fp16 + GradScaler,
 *                        /root/reference/src/train.py:75,89-98) with the exact power-of-two range scaling in place of a loss
 *                        scaler.  Forward, dgrad and wgrad.  Logits differ from the fp32 CPU reference by 2e-2 (resnet18) to
 *                        7e-2 (resnet34; tests/test_model_gpu.py, bench.py alt_modes): OUTSIDE BASELINE's 1e-3 bar.
 *   UWM_PREC_F16X3_BWD2  forward as UWM_PREC_F16X3_ALL (logits identical, inside the bar); in the backward the gradient operand dY
 *                        enters as ONE fp16 (two products per tile: dy_hi * w_hi + dy_hi * w_lo), i.e. gradients carry 11-bit dY
 *                        against 22-bit weights / activations — still above the reference's GPU arithmetic. */
int  uwm_set_precision(uwm_handle h, int mode);
int  uwm_get_precision(uwm_handle h);
/* The fp16x3 forward / dgrad kernels work in 16x16-pixel x 64-channel workgroups and are taken for launches of at least
 * `min_workgroups` of them (0 = the default: one per two compute units — measured: layer4 at batch 16, 128 workgroups on 256 CUs,
 * is still 2 % of the step faster there than on the fp32 Winograd kernels; smaller launches stay on those, which tile finer).  1 = wherever the shape allows (tests). */
int  uwm_set_precision_fill(uwm_handle h, int min_workgroups);
/* Routing batch: with batch > 0 every size-dependent kernel choice (fp16x3 fill rule, 4- / 8-wave and tile-width variants, tile
 * configurations of the implicit GEMM) is made as if the batch were `batch` images, whatever N uwm_forward gets; grids and split
 * counts follow the real N.  A 2-image parity sample then runs on exactly the kernels the 16-image step takes (bench.py, tests).
 * 0 (default) = the real batch. */
int  uwm_set_routing_batch(uwm_handle h, int batch);
/* Routing record: while enabled, every convolution-class launch of uwm_forward / uwm_backward appends one line
 * "<fwd|dgrad|wgrad> <layer name> <kernel>" (launch order) to a per-handle text.  uwm_routing_dump copies it (NUL-terminated, at
 * most cap bytes) and returns the size the whole text needs including the NUL; clear != 0 empties it afterwards.  Host-side
 * bookkeeping only: no device work, no effect on results. */
int  uwm_routing_enable(uwm_handle h, int on);
long long uwm_routing_dump(uwm_handle h, char* buf, long long cap, int clear);
/* EfficientNet encoders only: stochastic depth ("drop connect") of the MBConv blocks in training mode.  `rowscale` is a
 * device array [uwm_num_mbconv_blocks][N] holding, per block and sample, keep/(1 - p_block) with keep in {0,1}; the host
 * draws it each step (uwm_mbconv_drop_rate gives p_block; blocks without identity skip ignore their row).  NULL (the
 * default) disables it.  The pointer is read by the next uwm_forward(training=1) and its uwm_backward. */
int  uwm_set_drop_connect(uwm_handle h, const float* rowscale);
/* Single-operator entry point of the depthwise k x k convolution (k 3|5, stride 1|2; tests and kernel timing).  NHWC
 * activations, weights tap-major [k*k][C]; pad_begin = zero pad at the top/left (efficientnet_pytorch's static "same"
 * padding: the bottom/right pad follows from Ho, Wo).  mode 0: out[N][Ho][Wo][C] = conv(a = x[N][H][W][C], b = w);
 * mode 1: out[N][H][W][C] = dgrad(a = dy[N][Ho][Wo][C], b = w) (+ addend); mode 2: out[k*k][C] += wgrad(a = x, b = dy),
 * scratch = uwm_op_depthwise_scratch_floats(...) floats. */
int  uwm_op_depthwise(int mode, const float* a, const float* b, int k, int stride, int pad_begin, int N, int H, int W, int C,
                      int Ho, int Wo, const float* addend, float* out, float* scratch, uwm_stream stream);
long long uwm_op_depthwise_scratch_floats(int k, int N, int C, int Ho, int Wo);
int  uwm_num_mbconv_blocks(uwm_handle h);
float uwm_mbconv_drop_rate(uwm_handle h, int block);
/* predict.py:620-625 on the device: bilinear resize (cv2.INTER_LINEAR convention) of each image's logit plane
 * [N][h][w] (element stride ld) to [N][H][W], then (v > threshold) ? 255 : 0.  mask and/or resized may be NULL. */
int  uwm_resize_threshold(const float* logits, int ld, int N, int h, int w, int H, int W, float threshold,
                          int apply_sigmoid, uint8_t* mask, float* resized, uwm_stream stream);

/* Weight-gradient kernels run on an internal side stream (forked from / joined to the caller's stream with events,
 * per backward stage) so they overlap the dgrad chain; this switches that off/on at run time (default on). */
int  uwm_set_side_stream(uwm_handle h, int on);

/* Optional HIP-event profiler: while enabled every conv / wgrad launch carries a hipEvent pair attached to the kernel
 * dispatch itself (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, the clock rocprofv3's kernel trace
 * reads).  uwm_prof_collect waits for the events and returns, per kernel class, {launches, total ms, total algorithmic
 * FLOPs, total algorithmic HBM bytes} in out[class*4 + 0..3] (out: >= 4*max_classes doubles); returns the number of
 * classes. */
int  uwm_prof_enable(int on);
int  uwm_prof_collect(double* out, int max_classes);
const char* uwm_prof_class_name(int cls);

/* Workspace introspection for parity tests: element offset (in floats from the workspace base) and
 * element count of a planned intermediate.  Keys: "y:<conv>", "g:<conv>" (raw conv output / its
 * gradient; <conv> = state_dict prefix such as "encoder.layer1.0.conv1"), "xn:<i>", "gx:<i>"
 * (encoder block i output / masked gradient), "pool", "g_pool", "x4", "dcat:<i>", "gskip:<i>". */
int  uwm_debug_lookup(uwm_handle h, const char* key, long long* offset, long long* count);

/* ---- single-operator entry points (used by the parity tests) ---- */
typedef struct {
  const float* ptr; const float* scale; const float* shift;   /* NHWC fp32, optional lazy affine */
  int C, H, W, up, relu;
} uwm_src;
/* y[N][Ho][Wo][Cout] = conv(cat(s0,s1), w) ; w [Cout][Kpad] packed (k = tap*Ctot + c).
 * stats (2*Cout doubles: sum, sumsq; pre-zeroed) may be NULL.
 * cfg: -1 = the library's routing; 0..5 implicit-GEMM tile configs; 100+BN direct patch kernel; 200 16-channel patch
 * kernel; 300 (+BN, 308 = 8-wave) Winograd F(2x2,3x3); 400 Winograd in bf16x3 arithmetic; 500 segmentation-head streaming
 * kernel; 700 sub-pixel kernel for a 3x3 over a nearest-x2 upsampled 32-channel source with 16 outputs (conv_up2.hip);
 * 800 (+64 | +128 = channel tile) persistent LDS-DMA GEMM for 1x1 / stride-1 layers with Cin % 32 == 0 (conv_gemm.hip)
 * (tests / timing). */
int  uwm_op_set_igemm_f16x3(int on);   /* tests / kernel timing: uwm_op_conv / uwm_op_dgrad launches that end on the implicit GEMM (stride 2, 1x1) use its fp16x3 split-product form (what the model does for the stride-2 layers in the fp16x3 precision modes) */
int  uwm_op_conv(const uwm_src* s0, const uwm_src* s1, const float* w, int wrows, int Kpad, int kh, int kw, int stride,
                 int pad, int N, int Cout, const float* bias, float* y, double* stats, int cfg, uwm_stream stream);
/* dx[N][H][W][Cin] = conv_transpose(dy[N][Ho][Wo][Cout], wd) (+addend, *relu-mask) ; wd [Cin][KpadD] */
int  uwm_op_dgrad(const float* dy, int N, int Ho, int Wo, int Cout, const float* wd, int Cin, int KpadD, int kh, int kw,
                  int stride, int pad, int H, int W, const float* addend, const float* mask, const float* mscale,
                  const float* mshift, float* dx, uwm_stream stream);
/* force_igemm: 0 = the library's routing (Winograd-domain / sub-pixel / 16-channel / stem / 1x1-GEMM kernels where they apply);
 * 1 = flattened implicit GEMM only; 2 = no Winograd; 4 = wgrad_gemm.hip wherever applicable; 6 = the fp16x3 direct weight
 * gradient (wgrad_f16x3.hip: 3x3 stride 1, channels % 32 == 0, Wo % 32 == 0, Ho % 4 == 0) (tests / timing) */
int  uwm_op_wgrad(const uwm_src* s0, const uwm_src* s1, const float* dy, int N, int Ho, int Wo, int Cout, int wrows,
                  int Kpad, int kh, int kw, int stride, int pad, float* dw, int force_igemm, uwm_stream stream);
int  uwm_op_pack_dgrad(const float* w, int Cout, int Kpad, int ntaps, int Cin, float* wd, int KpadD, int CoutP,
                       uwm_stream stream);
int  uwm_op_maxpool(const uwm_src* in, int N, float* out, uint8_t* idx, uwm_stream stream);
/* gin[N][H][W][C] = (maxpool3x3s2_backward(gout, idx) + addend) * [relu(in) > 0] */
int  uwm_op_maxpool_backward(const float* gout, const uint8_t* idx, const float* addend, const uwm_src* in, int N,
                             float* gin, uwm_stream stream);
/* BatchNorm backward (batch statistics): g = grad wrt the BN output, y = BN input; scratch2c: 2*C doubles.
 * dy = gamma*rstd*(g - mean(g) - yhat*mean(g*yhat)), dgamma = sum g*yhat, dbeta = sum g */
int  uwm_op_bn_backward(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma,
                        double* scratch2c, float* dy, float* dgamma, float* dbeta, long long npix, int C, uwm_stream stream);
/* gradient of cat(nearest_x2(prev), skip): gprev[N][H/2][W/2][C0] = mask(sum 2x2 dcat[..., :C0]), gskip = dcat[..., C0:] */
/* decoder block conv1 dgrad with the concat split fused (Winograd epilogue): gprev [N][H/2][W/2][C0] = ReLU-masked
 * 2x2 sums of the first C0 gradient channels, gskip [N][H][W][C1] the rest; wd = uwm_op_pack_dgrad output */
int  uwm_op_dgrad_upsplit(const float* dy, int N, int H, int W, int Cout, const float* wd, int C0, int C1, int KpadD,
                          float* gprev, const float* pmask, const float* pscale, const float* pshift, float* gskip,
                          uwm_stream stream);
int  uwm_op_upsplit(const float* dcat, int N, int H, int W, int C0, int C1, float* gprev, const float* pmask,
                    const float* pscale, const float* pshift, float* gskip, uwm_stream stream);
/* out = relu(y*s2+b2 + (sd ? id*sd+bd : id)) */
int  uwm_op_residual(const float* y, const float* s2, const float* b2, const float* id, const float* sd, const float* bd,
                     float* out, long long npix, int C, uwm_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* UWM_H */
