// libuwm host side: smp.Unet(resnet18|34) graph description, parameter-arena and workspace
// planning, forward / staged-backward orchestration (kernel launches only — no host syncs, no
// allocation) and the C ABI declared in include/uwm.h.
//
// Graph follows SURVEY.md Appendix A.2/A.3 (the published smp.Unet algorithm reached from
// /root/reference/src/models/unet_model.py:64-71); state_dict key names are smp-compatible.
#include "../../include/uwm.h"
#include "uwm_kernels.h"

#include <cstdarg>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

using namespace uwm;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return 1;
}
#define HIPCHK(expr)                                                                           \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, \
       hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static inline long long rup(long long v, long long a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------ model description
struct BNL {
  std::string name; int C; int stage;
  long long g_off, b_off;          // param arena
  long long rm_off, rv_off;        // buffer arena
  size_t d_off;                    // workspace doubles: dgamma[C], dbeta[C], then nrep copies of {sum[C], sq[C]}, then 32 floats of max|dy|
  int nrep;                        // statistics copies (power of 2): spreads the conv epilogues' fp64 atomics
  size_t dcount() const { return 2 * (size_t)C + (size_t)nrep * 2 * C + 16; }     // (+ 16 doubles = the 32 float slots of max|dy|: fp16x3 dgrad scaling, zeroed with the sums)
  size_t xmax_off() const { return d_off + dcount() - 16; }                         // in doubles
  size_t f_off;                    // workspace floats: mean, rstd, scale, shift (4*C)
  float eps = 0.f, mom = 0.f;      // 0 = the descriptor's bn_eps / bn_momentum (EfficientNet encoder layers carry their own)
};
struct ConvL {
  std::string name; int Cin, CinP, Cout, CoutP, k, stride, pad, Kpad, KpadD, stage;
  long long w_off, bias_off; int bn; bool dgrad;
  bool dw = false;                 // depthwise layer (Cin = 1 per group): weights tap-major [k*k][Cout] in the arena
  long long wfloats() const { return dw ? (long long)k * k * CoutP : (long long)Cout * Kpad; }
  size_t wd_off;                   // workspace floats: [CinP][KpadD] dgrad repack
  size_t wu_off, wud_off;          // workspace floats: Winograd-transformed weights (forward / dgrad); 0 = none
  bool wino() const { return k == 3 && stride == 1 && pad == 1 && (CinP & 7) == 0; }
  bool wino_d() const { return dgrad && k == 3 && stride == 1 && pad == 1 && (CoutP & 7) == 0; }
  int c0 = 0;                      // channels of the FIRST source of this conv's input (decoder conv1: the up-sampled tensor; else CinP)
  // bf16x3 precision mode: whole 16-channel chunks on either side of the concat (conv_wino_x3.hip)
  bool x3() const { return wino() && (CinP & 15) == 0 && (c0 & 15) == 0; }
  bool x3_d() const { return wino_d() && (CoutP & 15) == 0; }
  // fp16x3 direct form (conv_f16x3.hip): chunk pairs of 16 channels on either side of the concat
  // (Cout >= 32: the 16-output full-resolution layers have their own kernels — conv_up2 / conv_patch16 — which the 64-channel tile
  // of conv_f16x3 cannot match: 565 vs 176 us on decoder block 4 conv1 — except the single-chunk 16 -> 16 form, decoder block 4 conv2, forward and dgrad)
  bool f3() const { return wino() && (((CinP & 31) == 0 && (c0 & 15) == 0 && Cout >= 32) || (CinP == 16 && c0 == 16 && Cout == 16)); }
  // the 7x7 / stride-2 stem of the ResNet encoders (3 input channels stored as 4): conv_stem_f16x3.hip in the fp16x3 modes
  bool stem7() const { return k == 7 && stride == 2 && pad == 3 && CinP == 4 && Cout == 64 && CoutP == 64 && !dw; }
  bool f3_d() const { return wino_d() && (((CoutP & 31) == 0 && CinP >= 16) || (CoutP == 16 && CinP == 16 && Cout == 16)); }
};
// encoder residual block.  BasicBlock: c1 3x3(stride) -> c2 3x3, c3 = -1.  Bottleneck: c1 1x1 -> c2 3x3(stride) -> c3 1x1(x4).
struct BlockL { int c1, c2, cd, c3 = -1, stride = 1, Cin = 0, Cout = 0; int last() const { return c3 >= 0 ? c3 : c2; } };
struct DecL { int c1, c2, C0, C1; };
// UnetPlusPlus decoder block.  Tensor ids: 0..4 = encoder features f1..f5 (f1 = stem, f5 = deepest), 5 + i = output
// of node i.  lvl = log2 of the down-scale of the node's OUTPUT (f1: 1 ... f5: 5, final node: 0).
struct NodeL { int c1, c2, prev; std::vector<int> skips; int C0, C1, lvl; };
// EfficientNet MBConv block: [expand 1x1 -> BN -> swish] -> depthwise k x k (static same pad) -> BN -> swish -> SE ->
// project 1x1 -> BN [-> drop-connect + identity].  ce = -1 when expand_ratio == 1.
struct MBL { int ce = -1, cdw = -1, cr = -1, cx = -1, cp = -1, Cin = 0, Cout = 0, mid = 0, nsq = 0, k = 3, stride = 1, pb = 0; bool skip = false; float drop = 0.f; };

constexpr size_t kWgParts = 8;      // partial-sum scratch: room for this many worst-case split launches between two flushes of the reduce queue
struct Plan {                       // workspace layout for one (N,H,W,training)
  int N = 0, H = 0, W = 0, training = -1;
  size_t bytes = 0;
  size_t x4 = 0, pool = 0, pool_idx = 0, g_pool = 0, tmp = 0, loss_scr = 0;
  std::vector<size_t> y, g;         // per conv: raw output / its gradient buffer (float offsets)
  std::vector<int> oh, ow;          // per conv: output height / width
  int wino_mode = 1;                // the handle's Winograd mode when this plan was made (0 off, 1 auto, 2 = 8-wave variant wherever allowed)
  int prec = 0;                     // the handle's precision mode (UWM_PREC_*: 0 fp32, 1 bf16x3 dgrad convs, 2 bf16x3 forward + dgrad convs)
  bool wino_ok(size_t ci) const { return wino_mode != 0 && oh[ci] >= 8 && ow[ci] >= 16; }   // conv_wino tile fits
  std::vector<size_t> xn, gx;       // per encoder block: residual output / its gradient
  std::vector<size_t> dcat, gskip;  // per decoder block (UnetPlusPlus: dcat[0] = shared scratch, gskip[0..3] = f4,f3,f2,f1 accumulators)
  size_t stem_a = 0;                // EfficientNet: materialised stem feature f1 = swish(bn(conv_stem))
  std::vector<size_t> a0, a1, a2, se;   // EfficientNet per block: swish(bn0(expand)), swish(bn1(dw)), SE-scaled, {pool[N][mid], s[N][mid], hpre[N][nsqP]}
  std::vector<int> mh, mw;          // EfficientNet per block: output height / width
  size_t dw_part = 0;               // EfficientNet: partial sums of the two-stage depthwise weight gradient (largest layer)
  size_t se_g = 0;                  // EfficientNet: SE backward gpool scratch [N][max mid]
  size_t se_part = 0;               // EfficientNet: partial sums of the two-stage (deterministic) SE pooling, largest layer
  std::vector<size_t> se_pool, se_gs;   // per block: pooled mean [N][mid] (forward) / {gs [N][mid], acc1 [N][nsq]} (backward), each set contiguous:
  size_t se_pool_all = 0, se_pool_floats = 0, se_gs_all = 0, se_gs_floats = 0;   // ONE memset per forward / backward instead of one per block
  std::vector<size_t> cat;          // UnetPlusPlus: per node, materialised skip concat (0 = none)
  size_t gcat = 0;                  // UnetPlusPlus: shared scratch for a node's skip-concat gradient
  size_t stat_d = 0, stat_d_count = 0;   // BN double region
  size_t colsum_scr = 0;            // per-workgroup fp64 partials of the head-bias gradient (two-stage, fixed order), in floats
  size_t wg_part = 0;               // per-split partial sums of the Winograd weight gradient (deterministic two-stage reduce)
};

struct uwm_model {
  uwm_unet_desc desc;
  std::vector<ConvL> convs;
  std::vector<BNL> bns;
  std::vector<std::vector<BlockL>> stages;   // 4 encoder stages
  std::vector<DecL> dec;
  std::vector<NodeL> nodes;          // UnetPlusPlus (arch 1) decoder in forward order; empty for Unet
  int featC[4] = {64, 128, 256, 512};  // channels of the encoder features f2..f5
  int f1C = 64;                      // channels of f1 (stem feature)
  std::vector<MBL> mb;               // EfficientNet encoder blocks (empty for ResNets)
  int feat_blk[4] = {0, 0, 0, 0};    // EfficientNet: index of the block whose output is f2..f5
  const float* keep = nullptr;       // uwm_set_drop_connect: device [mb.size()][N] row scales, or nullptr
  const float* keep_fwd = nullptr;   // the pointer the last training forward used (the backward must use the same)
  int stem = -1, head = -1, CP = 4, CinP = 4;
  long long param_floats = 0, buffer_floats = 0, param_count = 0;
  long long stage_begin[6] = {0, 0, 0, 0, 0, 0};
  std::vector<uwm_tensor_info> infos;
  float *params = nullptr, *grads = nullptr, *buffers = nullptr;
  size_t fixed_floats = 0;           // fixed workspace region (BN scratch + dgrad packs), in floats
  Plan plan;
  bool have_fwd = false;
  // weight-gradient kernels run on an internal side stream, forked from / joined to the caller's stream
  // with events (capturable fork-join; no host synchronisation)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pack = nullptr;
  hipEvent_t ev_disp = nullptr;       // attached to the completion of the latest bn_bwd_apply dispatch: the weight-gradient fork without a record of its own
  std::vector<const float*> disp_cov; // gradient buffers whose producing apply carried ev_disp in this uwm_backward call (later dispatches cover earlier ones: in-order queue)
  int disp_fork = 1;
  int use_side = 1;
  hipStream_t join_stream = nullptr;  // uwm_set_join_stream: stream that waits for the side stream at the end of uwm_backward (default: the caller's)
  bool packed_in_fwd = false;         // dgrad weight repacks were enqueued on the side stream by the last forward
  int pack_mode = -1;                 // Winograd mode those repacks were made for
  int wino_mode = 1;                  // per-handle Winograd mode (uwm_set_winograd_mode); starts as the process default
  int prec = 0;                       // per-handle precision mode (uwm_set_precision): UWM_PREC_F32 | UWM_PREC_BF16X3 | UWM_PREC_BF16X3_ALL
  int device = -1;                    // HIP device the bound arenas live on (uwm_bind)
  int nstages = 5;                    // backward stages = gradient buckets (head+decoder, then four encoder groups)
  bool hwq_warned = false;
  ReduceQueue rq;                    // partial-sum reduces of the split weight gradients, flushed once per backward stage (uwm_kernels.h)
  int f3_min_wgs = 0;                // fp16x3 kernels: smallest launch (workgroups) they take; 0 = one per CU (uwm_set_precision_fill)
  int route_n = 0;                   // uwm_set_routing_batch: every size-dependent kernel choice is made as if the batch were this many images (0: the real batch)
  bool route_log_on = false;         // uwm_routing_enable: record (pass, layer, kernel) of every conv / dgrad / wgrad launch
  std::string route_log;             // text of the record since the last uwm_routing_dump(.., clear)
  bool prec_from_env = false;        // the precision mode came from UWM_PRECISION (logged once at the first forward)
  std::vector<char> out_sums;        // per residual block: the BatchNorm-backward sums of its last BatchNorm were made by the dgrad that wrote its output gradient (run_dgrad bn_y)
};

static int add_bn(uwm_model* m, const std::string& name, int C, int stage, float eps = 0.f, float mom = 0.f) {
  BNL b; b.eps = eps; b.mom = mom; b.name = name; b.C = C; b.stage = stage; b.nrep = C <= 64 ? 32 : (C <= 128 ? 16 : (C <= 256 ? 8 : 4)); b.g_off = b.b_off = b.rm_off = b.rv_off = -1; b.d_off = b.f_off = 0;
  m->bns.push_back(b); return (int)m->bns.size() - 1;
}
static int add_conv(uwm_model* m, const std::string& name, int Cin, int Cout, int k, int stride, int pad, int stage,
                    bool dgrad, const std::string& bn_name, bool bias = false, float bn_eps = 0.f, float bn_mom = 0.f) {
  ConvL c; c.name = name; c.Cin = Cin; c.CinP = (int)rup(Cin, 4); c.Cout = Cout; c.CoutP = (int)rup(Cout, 4);
  c.k = k; c.stride = stride; c.pad = pad; c.stage = stage; c.dgrad = dgrad;
  c.Kpad = (int)rup((long long)k * k * c.CinP, 32);
  c.KpadD = (int)rup((long long)k * k * c.CoutP, 32);
  c.w_off = -1; c.bias_off = bias ? 0 : -1; c.wd_off = 0; c.wu_off = c.wud_off = 0; c.c0 = c.CinP;
  c.bn = bn_name.empty() ? -1 : add_bn(m, bn_name, Cout, stage, bn_eps, bn_mom);
  m->convs.push_back(c); return (int)m->convs.size() - 1;
}

static void push_info(uwm_model* m, const std::string& name, int kind, int arena, long long off, int ndim,
                      const long long* shape, const long long* stride) {
  uwm_tensor_info t; memset(&t, 0, sizeof(t));
  snprintf(t.name, sizeof(t.name), "%s", name.c_str());
  t.kind = kind; t.arena = arena; t.ndim = ndim; t.offset = off;
  for (int i = 0; i < ndim; ++i) { t.shape[i] = shape[i]; t.stride[i] = stride[i]; }
  m->infos.push_back(t);
}

static int build_model(uwm_model* m) {
  const uwm_unet_desc& d = m->desc;
  int nb[4] = {0, 0, 0, 0}; int expn = 1;
  const bool effnet = d.encoder == UWM_ENC_EFFICIENTNET_B4;
  if (d.encoder == UWM_ENC_RESNET18) { nb[0] = 2; nb[1] = 2; nb[2] = 2; nb[3] = 2; }
  else if (d.encoder == UWM_ENC_RESNET34) { nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; }
  else if (d.encoder == UWM_ENC_RESNET50) { nb[0] = 3; nb[1] = 4; nb[2] = 6; nb[3] = 3; expn = 4; }
  else if (!effnet) return fail("unsupported encoder %d (supported: resnet18, resnet34, resnet50, efficientnet-b4)", d.encoder);
  if (d.in_channels < 1 || d.in_channels > 4) return fail("in_channels must be 1..4, got %d", d.in_channels);
  if (d.classes < 1 || d.classes > 4) return fail("classes must be 1..4, got %d", d.classes);
  for (int i = 0; i < 5; ++i)
    if (d.decoder_channels[i] < 4 || (d.decoder_channels[i] & 3)) return fail("decoder_channels[%d]=%d must be a positive multiple of 4", i, d.decoder_channels[i]);
  m->CP = (int)rup(d.classes, 4); m->CinP = (int)rup(d.in_channels, 4);
  const int widths[4] = {64, 128, 256, 512};
  m->stages.resize(4);
  int encc[5] = {512 * expn, 256 * expn, 128 * expn, 64 * expn, 64};
  // backward stages: 0 head+decoder, 1 layer4, 2 layer3, 3 layer2, 4 layer1+stem
  if (!effnet) {
    m->stem = add_conv(m, "encoder.conv1", d.in_channels, 64, 7, 2, 3, 4, false, "encoder.bn1");
    int cin = 64;
    for (int s = 0; s < 4; ++s) {
      const int stage = 4 - s;
      for (int b = 0; b < nb[s]; ++b) {
        const int stride = (b == 0 && s > 0) ? 2 : 1;
        char pre[64]; snprintf(pre, sizeof(pre), "encoder.layer%d.%d", s + 1, b);
        const std::string P(pre);
        BlockL bl; bl.stride = stride; bl.Cin = cin; bl.Cout = widths[s] * expn;
        if (expn == 1) {
          bl.c1 = add_conv(m, P + ".conv1", cin, widths[s], 3, stride, 1, stage, true, P + ".bn1");
          bl.c2 = add_conv(m, P + ".conv2", widths[s], widths[s], 3, 1, 1, stage, true, P + ".bn2");
        } else {            // torchvision Bottleneck v1.5: the stride sits on the 3x3
          bl.c1 = add_conv(m, P + ".conv1", cin, widths[s], 1, 1, 0, stage, true, P + ".bn1");
          bl.c2 = add_conv(m, P + ".conv2", widths[s], widths[s], 3, stride, 1, stage, true, P + ".bn2");
          bl.c3 = add_conv(m, P + ".conv3", widths[s], bl.Cout, 1, 1, 0, stage, true, P + ".bn3");
        }
        bl.cd = -1;
        if (stride != 1 || cin != bl.Cout)
          bl.cd = add_conv(m, P + ".downsample.0", cin, bl.Cout, 1, stride, 0, stage, true, P + ".downsample.1");
        m->stages[s].push_back(bl);
        cin = bl.Cout;
      }
    }
  } else {
    // efficientnet_pytorch "efficientnet-b4" as smp wraps it: width 1.4 / depth 1.8 scaling of the b0
    // stage table; BatchNorm eps 1e-3, momentum 0.01; Conv2dStaticSamePadding pads computed along the 380-pixel size chain
    // (asymmetric: begin = total / 2); SE ratio 0.25 of the block INPUT channels; drop-connect 0.2 * idx / 32.
    struct St { int rep, k, stride, expand, cin, cout; };
    const St stg[7] = {{2, 3, 1, 1, 48, 24}, {4, 3, 2, 6, 24, 32}, {4, 5, 2, 6, 32, 56}, {6, 3, 2, 6, 56, 112},
                       {6, 5, 1, 6, 112, 160}, {8, 5, 2, 6, 160, 272}, {2, 3, 1, 6, 272, 448}};
    const float be = 1e-3f, bm = 0.01f;
    auto same_pad = [](int size, int k, int st, int* out) { const int o = (size + st - 1) / st; int t = (o - 1) * st + k - size; if (t < 0) t = 0; *out = o; return t / 2; };
    int size = 380, osz = 0;
    const int spb = same_pad(size, 3, 2, &osz); size = osz;
    m->stem = add_conv(m, "encoder._conv_stem", d.in_channels, 48, 3, 2, spb, 4, false, "encoder._bn0", false, be, bm);
    const int feat_after[4] = {6, 10, 22, 32};             // smp: features after blocks[:6], [:10], [:22], [:32]
    int idx = 0;
    for (int sg = 0; sg < 7; ++sg)
      for (int r = 0; r < stg[sg].rep; ++r, ++idx) {
        MBL b; b.Cin = r == 0 ? stg[sg].cin : stg[sg].cout; b.Cout = stg[sg].cout; b.k = stg[sg].k;
        b.stride = r == 0 ? stg[sg].stride : 1; b.mid = b.Cin * stg[sg].expand; b.nsq = std::max(1, b.Cin / 4);
        b.skip = b.stride == 1 && b.Cin == b.Cout; b.drop = 0.2f * (float)idx / 32.f;
        int fs = 0; while (idx >= feat_after[fs]) ++fs;     // feature stage this block belongs to (0..3)
        const int stage = 4 - fs;
        char pre[64]; snprintf(pre, sizeof(pre), "encoder._blocks.%d", idx);
        const std::string P(pre);
        if (stg[sg].expand != 1) b.ce = add_conv(m, P + "._expand_conv", b.Cin, b.mid, 1, 1, 0, stage, true, P + "._bn0", false, be, bm);
        b.pb = same_pad(size, b.k, b.stride, &osz); size = osz;
        b.cdw = add_conv(m, P + "._depthwise_conv", 1, b.mid, b.k, b.stride, b.pb, stage, false, P + "._bn1", false, be, bm);
        m->convs[b.cdw].dw = true;
        b.cr = add_conv(m, P + "._se_reduce", b.mid, b.nsq, 1, 1, 0, stage, false, "", true);
        b.cx = add_conv(m, P + "._se_expand", b.nsq, b.mid, 1, 1, 0, stage, false, "", true);
        b.cp = add_conv(m, P + "._project_conv", b.mid, b.Cout, 1, 1, 0, stage, true, P + "._bn2", false, be, bm);
        m->mb.push_back(b);
        if (idx + 1 == feat_after[fs]) m->feat_blk[fs] = idx;
      }
    encc[0] = 448; encc[1] = 160; encc[2] = 56; encc[3] = 32; encc[4] = 48;
  }
  m->f1C = encc[4];
  for (int j = 0; j < 4; ++j) m->featC[j] = encc[3 - j];            // f2..f5
  int prev = encc[0];
  if (d.arch == UWM_ARCH_UNET) {
    for (int i = 0; i < 5; ++i) {
      const int skip = i < 4 ? encc[i + 1] : 0, out = d.decoder_channels[i];
      char pre[64]; snprintf(pre, sizeof(pre), "decoder.blocks.%d", i);
      DecL dl; dl.C0 = prev; dl.C1 = skip;
      dl.c1 = add_conv(m, std::string(pre) + ".conv1.0", prev + skip, out, 3, 1, 1, 0, true, std::string(pre) + ".conv1.1");
      m->convs[dl.c1].c0 = prev;
      dl.c2 = add_conv(m, std::string(pre) + ".conv2.0", out, out, 3, 1, 1, 0, true, std::string(pre) + ".conv2.1");
      m->dec.push_back(dl);
      prev = out;
    }
  } else if (d.arch == UWM_ARCH_UNETPLUSPLUS) {
    // smp UnetPlusPlusDecoder (src/configs/config.py:15 default MODEL.NAME): blocks x_{depth}_{layer}
    const int in_ch[5] = {encc[0], d.decoder_channels[0], d.decoder_channels[1], d.decoder_channels[2], d.decoder_channels[3]};
    const int skip_ch[5] = {encc[1], encc[2], encc[3], encc[4], 0};
    int blk_c1[5][5], blk_c2[5][5], blk_in[5][5], blk_skip[5][5];
    auto add_block = [&](int dep, int lay, int cin_, int cskip, int cout_) {
      char pre[64]; snprintf(pre, sizeof(pre), "decoder.blocks.x_%d_%d", dep, lay);
      blk_in[dep][lay] = cin_; blk_skip[dep][lay] = cskip;
      blk_c1[dep][lay] = add_conv(m, std::string(pre) + ".conv1.0", cin_ + cskip, cout_, 3, 1, 1, 0, true, std::string(pre) + ".conv1.1");
      m->convs[blk_c1[dep][lay]].c0 = cin_;
      blk_c2[dep][lay] = add_conv(m, std::string(pre) + ".conv2.0", cout_, cout_, 3, 1, 1, 0, true, std::string(pre) + ".conv2.1");
    };
    for (int lay = 0; lay < 4; ++lay)            // registration order of smp's ModuleDict = state_dict order
      for (int dep = 0; dep <= lay; ++dep) {
        if (dep == 0) add_block(0, lay, in_ch[lay], skip_ch[lay] * (lay + 1), d.decoder_channels[lay]);
        else add_block(dep, lay, skip_ch[lay - 1], skip_ch[lay] * (lay + 1 - dep), skip_ch[lay]);
      }
    add_block(0, 4, in_ch[4], 0, d.decoder_channels[4]);
    // forward order (smp UnetPlusPlusDecoder.forward); features[k] = f_{5-k} = tensor id 4 - k
    int node_of[5][5];
    auto add_node = [&](int dep, int lay, int prev_id, const std::vector<int>& skips, int lvl) {
      NodeL nd; nd.c1 = blk_c1[dep][lay]; nd.c2 = blk_c2[dep][lay]; nd.prev = prev_id; nd.skips = skips;
      nd.C0 = blk_in[dep][lay]; nd.C1 = blk_skip[dep][lay]; nd.lvl = lvl;
      node_of[dep][lay] = (int)m->nodes.size(); m->nodes.push_back(nd);
    };
    for (int li = 0; li < 4; ++li)
      for (int dep = 0; dep < 4 - li; ++dep) {
        if (li == 0) add_node(dep, dep, 4 - dep, {4 - (dep + 1)}, 4 - dep);
        else {
          const int L2 = dep + li;
          std::vector<int> sk;
          for (int idx = dep + 1; idx <= L2; ++idx) sk.push_back(5 + node_of[idx][L2]);
          sk.push_back(4 - (L2 + 1));
          add_node(dep, L2, 5 + node_of[dep][L2 - 1], sk, 4 - L2);
        }
      }
    add_node(0, 4, 5 + node_of[0][3], {}, 0);
    prev = d.decoder_channels[4];
  } else return fail("unsupported architecture %d (0 = Unet, 1 = UnetPlusPlus)", d.arch);
  m->head = add_conv(m, "segmentation_head.0", prev, d.classes, 3, 1, 1, 0, true, "", true);

  // ---- parameter arena: grouped by backward stage so each stage's gradients are one range
  long long off = 0;
  for (int st = 0; st < 5; ++st) {
    m->stage_begin[st] = off;
    for (auto& c : m->convs) if (c.stage == st) {
      c.w_off = off; off += c.wfloats();
      if (c.bias_off == 0) { c.bias_off = off; off += c.CoutP; }
    }
    for (auto& b : m->bns) if (b.stage == st) { b.g_off = off; off += b.C; b.b_off = off; off += b.C; }
    off = rup(off, 64);
  }
  m->stage_begin[5] = off;
  m->param_floats = off;
  long long boff = 0;
  for (auto& b : m->bns) { b.rm_off = boff; boff += b.C; b.rv_off = boff; boff += b.C; }
  m->buffer_floats = rup(boff, 64);

  // ---- fixed workspace region: BN scratch, dgrad weight repacks
  size_t dcount = 0;
  for (auto& b : m->bns) { b.d_off = dcount; dcount += b.dcount(); }
  size_t f = dcount * 2;                        // doubles first (in float units)
  f = (size_t)rup((long long)f, 64);
  for (auto& b : m->bns) { b.f_off = f; f += 4 * (size_t)b.C; }
  f = (size_t)rup((long long)f, 64);
  for (auto& c : m->convs) if (c.dgrad) { c.wd_off = f; f += (size_t)c.CinP * c.KpadD; f = (size_t)rup((long long)f, 64); }
  for (auto& c : m->convs) {
    // (one slot per direction holds whichever bank the precision mode asks for: fp32 Winograd, bf16x3 Winograd or fp16x3 direct)
    if (c.wino()) { c.wu_off = f; f += std::max(wino_weights_floats(c.Cout, c.CinP), c.f3() ? f16x3_bank_floats(c.Cout, c.CinP) : 0); f = (size_t)rup((long long)f, 64); }
    if (c.stem7()) { c.wu_off = f; f += stem_f16x3_bank_floats(); f = (size_t)rup((long long)f, 64); }
    if (c.wino_d()) { c.wud_off = f; f += std::max(wino_weights_floats(c.CinP, c.CoutP), c.f3_d() ? f16x3_bank_floats(c.CinP, c.CoutP) : 0); f = (size_t)rup((long long)f, 64); }
  }
  m->fixed_floats = f;

  // ---- tensor infos in smp state_dict order
  m->param_count = 0;
  auto conv_info = [&](const ConvL& c) {
    long long shape[4] = {c.Cout, c.Cin, c.k, c.k};
    long long stride[4] = {c.Kpad, 1, (long long)c.k * c.CinP, c.CinP};
    if (c.dw) { stride[0] = 1; stride[1] = 1; stride[2] = (long long)c.k * c.CoutP; stride[3] = c.CoutP; }
    push_info(m, c.name + ".weight", UWM_KIND_CONV_W, UWM_ARENA_PARAM, c.w_off, 4, shape, stride);
    m->param_count += (long long)c.Cout * c.Cin * c.k * c.k;
    if (c.bias_off >= 0) {
      long long s1[1] = {c.Cout}, st1[1] = {1};
      push_info(m, c.name + ".bias", UWM_KIND_BIAS, UWM_ARENA_PARAM, c.bias_off, 1, s1, st1);
      m->param_count += c.Cout;
    }
  };
  auto bn_info = [&](const BNL& b) {
    long long s1[1] = {b.C}, st1[1] = {1};
    push_info(m, b.name + ".weight", UWM_KIND_BN_GAMMA, UWM_ARENA_PARAM, b.g_off, 1, s1, st1);
    push_info(m, b.name + ".bias", UWM_KIND_BN_BETA, UWM_ARENA_PARAM, b.b_off, 1, s1, st1);
    push_info(m, b.name + ".running_mean", UWM_KIND_BN_MEAN, UWM_ARENA_BUFFER, b.rm_off, 1, s1, st1);
    push_info(m, b.name + ".running_var", UWM_KIND_BN_VAR, UWM_ARENA_BUFFER, b.rv_off, 1, s1, st1);
    m->param_count += 2LL * b.C;
  };
  for (auto& c : m->convs) { conv_info(c); if (c.bn >= 0) bn_info(m->bns[c.bn]); }
  return 0;
}

// The rest of this translation unit lives in five parts (the 1 800-line monolith split by concern):
#include "uwm_plan.inc"        // workspace plan + routing predicates
#include "uwm_launch.inc"      // per-layer launch helpers, filter-bank jobs, routing record
#include "uwm_forward.inc"     // uwm_forward's walk
#include "uwm_backward.inc"    // uwm_backward's staged walk
#include "uwm_abi.inc"         // extern "C" entry points
