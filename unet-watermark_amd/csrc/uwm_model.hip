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

// ------------------------------------------------------------------------------ workspace plan
static void make_plan(uwm_model* m, int N, int H, int W, int training) {
  Plan p; p.N = N; p.H = H; p.W = W; p.training = training; p.wino_mode = m->wino_mode; p.prec = m->prec;
  size_t off = m->fixed_floats;
  auto alloc = [&](size_t floats) { size_t o = off; off += (size_t)rup((long long)floats, 64); return o; };
  const size_t nc = m->convs.size();
  p.y.assign(nc, 0); p.g.assign(nc, 0); p.oh.assign(nc, 0); p.ow.assign(nc, 0);
  p.stat_d = 0; p.stat_d_count = 0; for (auto& b : m->bns) p.stat_d_count += b.dcount();
  p.loss_scr = alloc(64);
  p.x4 = alloc((size_t)N * H * W * m->CinP);
  int h = H / 2, w = W / 2;
  const int stemC = m->convs[m->stem].CoutP;
  p.y[m->stem] = alloc((size_t)N * h * w * stemC); p.oh[m->stem] = h; p.ow[m->stem] = w;
  size_t nblk = m->mb.size(); for (auto& s : m->stages) nblk += s.size();
  p.xn.assign(nblk, 0); p.gx.assign(nblk, 0);
  size_t bi = 0; size_t max_in = 0, max_mid = 0, max_nsq = 0;
  auto place = [&](int ci, int oh, int ow) { p.y[ci] = alloc((size_t)N * oh * ow * m->convs[ci].CoutP); p.oh[ci] = oh; p.ow[ci] = ow; };
  if (m->mb.empty()) {
    h /= 2; w /= 2;
    p.pool = alloc((size_t)N * h * w * 64);
    p.pool_idx = alloc((size_t)N * h * w * 64 / 4 + 64);
    max_in = (size_t)N * h * w * 64;
    for (int s = 0; s < 4; ++s) {
      for (auto& bl : m->stages[s]) {
        const int hi = h, wi = w;
        h /= bl.stride; w /= bl.stride;
        if (bl.c3 < 0) { place(bl.c1, h, w); place(bl.c2, h, w); }
        else { place(bl.c1, hi, wi); place(bl.c2, h, w); place(bl.c3, h, w); }
        if (bl.cd >= 0) place(bl.cd, h, w);
        const size_t sz = (size_t)N * h * w * bl.Cout;
        p.xn[bi++] = alloc(sz);
        if (sz > max_in) max_in = sz;
        if ((size_t)N * hi * wi * bl.Cin > max_in) max_in = (size_t)N * hi * wi * bl.Cin;
      }
    }
  } else {
    p.stem_a = alloc((size_t)N * h * w * stemC);
    const size_t nb = m->mb.size();
    p.a0.assign(nb, 0); p.a1.assign(nb, 0); p.a2.assign(nb, 0); p.se.assign(nb, 0); p.mh.assign(nb, 0); p.mw.assign(nb, 0);
    p.se_pool.assign(nb, 0); p.se_gs.assign(nb, 0);
    for (size_t i = 0; i < nb; ++i) { p.se_pool[i] = p.se_pool_floats; p.se_pool_floats += (size_t)rup((long long)N * m->mb[i].mid, 64); }
    p.se_pool_all = alloc(p.se_pool_floats);
    { size_t mm = 0; for (auto& b : m->mb) mm = std::max(mm, (size_t)b.mid); p.se_part = alloc(se_reduce_scratch_floats(N, (int)mm)); }
    for (size_t i = 0; i < nb; ++i) p.se_pool[i] += p.se_pool_all;
    for (; bi < nb; ++bi) {
      const MBL& b = m->mb[bi];
      const int hi = h, wi = w;
      h /= b.stride; w /= b.stride;
      if (b.ce >= 0) { place(b.ce, hi, wi); p.a0[bi] = alloc((size_t)N * hi * wi * b.mid); }
      place(b.cdw, h, w);
      p.a1[bi] = alloc((size_t)N * h * w * b.mid); p.a2[bi] = alloc((size_t)N * h * w * b.mid);
      p.se[bi] = alloc((size_t)N * (b.mid + 2 * (size_t)rup(b.nsq, 4)));        // {s [N][mid], hpre [N][nsq], hid [N][nsq]}
      place(b.cp, h, w);
      p.oh[b.cr] = p.ow[b.cr] = p.oh[b.cx] = p.ow[b.cx] = 1;
      p.xn[bi] = alloc((size_t)N * h * w * b.Cout);
      p.mh[bi] = h; p.mw[bi] = w;
      max_mid = std::max(max_mid, (size_t)b.mid); max_nsq = std::max(max_nsq, (size_t)rup(b.nsq, 4));
    }
  }
  for (size_t i = 0; i < m->dec.size(); ++i) {
    h *= 2; w *= 2;
    const size_t sz = (size_t)N * h * w * m->convs[m->dec[i].c1].Cout;
    p.y[m->dec[i].c1] = alloc(sz); p.y[m->dec[i].c2] = alloc(sz);
    p.oh[m->dec[i].c1] = p.oh[m->dec[i].c2] = h; p.ow[m->dec[i].c1] = p.ow[m->dec[i].c2] = w;
  }
  p.cat.assign(m->nodes.size(), 0);
  size_t max_gcat = 0, max_dcat = 0;
  for (size_t i = 0; i < m->nodes.size(); ++i) {
    const NodeL& nd = m->nodes[i];
    const int nh = H >> nd.lvl, nw = W >> nd.lvl;
    const size_t sz = (size_t)N * nh * nw * m->convs[nd.c1].Cout;
    p.y[nd.c1] = alloc(sz); p.y[nd.c2] = alloc(sz);
    p.oh[nd.c1] = p.oh[nd.c2] = nh; p.ow[nd.c1] = p.ow[nd.c2] = nw;
    if (nd.skips.size() > 1) p.cat[i] = alloc((size_t)N * nh * nw * nd.C1);
    max_gcat = std::max(max_gcat, (size_t)N * nh * nw * nd.C1);
    max_dcat = std::max(max_dcat, (size_t)N * nh * nw * (nd.C0 + nd.C1));
  }
  p.oh[m->head] = H; p.ow[m->head] = W;
  if (training) {
    p.colsum_scr = alloc(2 * colsum_scratch_doubles(m->CP));
    p.wg_part = alloc(kWgParts * wgrad_wino_scratch_floats());      // several layers' partial images wait for one batched reduce
    // gradient buffers (same shapes as their activations)
    h = H / 2; w = W / 2;
    p.g[m->stem] = alloc((size_t)N * h * w * stemC);
    if (m->mb.empty()) {
      h /= 2; w /= 2;
      p.g_pool = alloc((size_t)N * h * w * 64);
      bi = 0;
      for (int s = 0; s < 4; ++s)
        for (auto& bl : m->stages[s]) {
          for (int ci : {bl.c1, bl.c2, bl.c3, bl.cd})
            if (ci >= 0) p.g[ci] = alloc((size_t)N * p.oh[ci] * p.ow[ci] * m->convs[ci].Cout);
          p.gx[bi++] = alloc((size_t)N * p.oh[bl.c2] * p.ow[bl.c2] * bl.Cout);
        }
      p.tmp = alloc(max_in);
    } else {
      for (size_t i = 0; i < m->mb.size(); ++i) {
        const MBL& b = m->mb[i];
        for (int ci : {b.ce, b.cdw, b.cp})
          if (ci >= 0) p.g[ci] = alloc((size_t)N * p.oh[ci] * p.ow[ci] * m->convs[ci].CoutP);
        p.gx[i] = alloc((size_t)N * p.mh[i] * p.mw[i] * b.Cout);
      }
      size_t max_part = 0;
      for (size_t i = 0; i < m->mb.size(); ++i) max_part = std::max(max_part, dw_wgrad_scratch_floats(m->mb[i].k, N, m->mb[i].mid, p.mh[i], p.mw[i]));
      p.dw_part = alloc(max_part);
      p.se_g = alloc((size_t)N * max_mid);
      for (size_t i = 0; i < m->mb.size(); ++i) { p.se_gs[i] = p.se_gs_floats; p.se_gs_floats += (size_t)rup((long long)N * (m->mb[i].mid + m->mb[i].nsq), 64); }
      p.se_gs_all = alloc(p.se_gs_floats);
      for (size_t i = 0; i < m->mb.size(); ++i) p.se_gs[i] += p.se_gs_all;
    }
    h = H / 32; w = W / 32;                       // deepest feature: the decoder loop below doubles from here
    p.dcat.assign(m->dec.size(), 0); p.gskip.assign(m->dec.size(), 0);
    for (size_t i = 0; i < m->dec.size(); ++i) {
      h *= 2; w *= 2;
      const DecL& d = m->dec[i];
      const size_t sz = (size_t)N * h * w * m->convs[d.c1].Cout;
      p.g[d.c1] = alloc(sz); p.g[d.c2] = alloc(sz);
      p.dcat[i] = alloc((size_t)N * h * w * (d.C0 + d.C1));
      if (d.C1 > 0) p.gskip[i] = alloc((size_t)N * h * w * d.C1);
    }
    if (!m->nodes.empty()) {
      for (auto& nd : m->nodes) {
        const size_t sz = (size_t)N * (H >> nd.lvl) * (W >> nd.lvl) * m->convs[nd.c1].Cout;
        p.g[nd.c1] = alloc(sz); p.g[nd.c2] = alloc(sz);
      }
      p.dcat.assign(1, alloc(max_dcat));
      p.gcat = alloc(max_gcat);
      const int fc[4] = {m->featC[2], m->featC[1], m->featC[0], m->f1C};       // gradient accumulators of f4, f3, f2, f1
      p.gskip.assign(4, 0);
      for (int j = 0; j < 4; ++j) p.gskip[j] = alloc((size_t)N * (H >> (4 - j)) * (W >> (4 - j)) * fc[j]);
    }
  }
  p.bytes = off * sizeof(float);
  m->plan = p;
}

// a conv takes the fp16x3 direct kernel (forward) when its shape is eligible AND the launch fills the chip: 16x16-pixel x 64-channel
// workgroups, at least one per two CUs (measured: 973 vs 952 img/s with layer4's 128-workgroup launches on it; smaller launches
// — small batch / resolution — stay on the Winograd kernels)
static bool f3_fwd_on(const uwm_model* m, size_t ci) {
  const ConvL& cv = m->convs[ci]; const Plan& p = m->plan;
  if (p.prec < UWM_PREC_F16X3 || !cv.f3() || !p.wino_ok(ci)) return false;
  const long wgs = (long)(m->route_n > 0 ? m->route_n : p.N) * ((p.oh[ci] + 15) / 16) * ((p.ow[ci] + 15) / 16) * ((cv.Cout + 63) / 64);
  return wgs >= (m->f3_min_wgs > 0 ? m->f3_min_wgs : device_cu_count() / 2);
}
// a decoder conv1's dgrad splits the concat gradient in its epilogue (ConvArgs::out_up): the fp16x3 kernel takes it when the
// boundary sits on a 64-channel tile
// the stem on conv_stem_f16x3: fp16x3 forward modes, the same fill rule over its 16x16-pixel workgroups
static bool stem_f3_on(const uwm_model* m) {
  if (m->stem < 0) return false;
  const ConvL& cv = m->convs[m->stem]; const Plan& p = m->plan;
  if (p.prec < UWM_PREC_F16X3 || !cv.stem7() || !cv.wu_off || p.wino_mode == 0 || dbg_flag("UWM_NO_STEM_F16X3")) return false;
  const long wgs = (long)(m->route_n > 0 ? m->route_n : p.N) * ((p.oh[m->stem] + 15) / 16) * ((p.ow[m->stem] + 15) / 16);
  return wgs >= (m->f3_min_wgs > 0 ? m->f3_min_wgs : device_cu_count() / 2);
}
static bool f3d_plain(const uwm_model* m, int ci) {
  for (auto& d : m->dec) if (d.c1 == ci) return (d.C0 & 63) == 0;
  for (auto& nd : m->nodes) if (nd.c1 == ci) return (nd.C0 & 63) == 0;
  return true;
}
// fp16x3 weight gradient: the map tiled by whole 4 x 32- or 8 x 16-pixel stages, 32-channel tiles on either side of the concat
static bool f3_wgrad_on(const uwm_model* m, size_t ci) {
  const ConvL& cv = m->convs[ci]; const Plan& p = m->plan;
  return p.prec >= UWM_PREC_F16X3_ALL && cv.wino() && cv.bn >= 0 && cv.Cout >= 32 && (cv.CinP & 31) == 0 && (cv.c0 & 31) == 0 && cv.Kpad == 9 * cv.CinP &&
         (((p.ow[ci] % 32) == 0 && (p.oh[ci] % 4) == 0) || ((p.ow[ci] % 16) == 0 && (p.oh[ci] % 8) == 0)) && !dbg_flag("UWM_NO_F16X3_WGRAD");
}
static bool f3_dgrad_on(const uwm_model* m, size_t ci) {
  const ConvL& cv = m->convs[ci]; const Plan& p = m->plan;
  if (p.prec < UWM_PREC_F16X3_ALL || !cv.f3_d() || cv.bn < 0 || !p.wino_ok(ci) || !f3d_plain(m, (int)ci)) return false;
  const long wgs = (long)(m->route_n > 0 ? m->route_n : p.N) * ((p.oh[ci] + 15) / 16) * ((p.ow[ci] + 15) / 16) * ((cv.CinP + 63) / 64);      // (stride 1: the input has the output's size)
  return wgs >= (m->f3_min_wgs > 0 ? m->f3_min_wgs : device_cu_count() / 2);
}
// split products per tile (ConvArgs::nprod / WgradArgs::nprod) under the handle's precision mode: f16x1 = hi*hi' everywhere;
// f16x3_bwd2 = the backward's dY operand as ONE fp16 (two products), forward unchanged; the fp32-class modes: three
static int f3_nprod(const uwm_model* m, bool backward) {
  const int pm = m->plan.prec;
  if (pm == UWM_PREC_F16X1) return 1;
  if (pm == UWM_PREC_F16X3_BWD2 && backward) return 2;
  return 3;
}
// layout of a layer's fp16x3 bank (ConvArgs::wu_layout): 1 = conv_f16x3v2.hip (32x32x16 MFMA, 8 x 32-pixel tiles) where the map is
// tiled by whole tiles and the output rows by 32-row fragments (a decoder conv1 dgrad with the fused concat split: 64-row tiles)
static int f3_layout(const uwm_model* m, size_t ci, bool dgrad) {
  const ConvL& cv = m->convs[ci]; const Plan& p = m->plan;
  const int rows = dgrad ? cv.CinP : cv.Cout, chans = dgrad ? cv.CoutP : cv.CinP;
  if (rows != (dgrad ? cv.CinP : cv.CoutP)) return 0;
  if (dgrad && !f3d_plain(m, (int)ci)) return 0;
  bool split = false;
  if (dgrad) { for (auto& d : m->dec) if (d.c1 == (int)ci) split = true; for (auto& nd : m->nodes) if (nd.c1 == (int)ci) split = true; }
  if (split && (rows & 63)) return 0;
  return f16x3v2_shape(p.oh[ci], p.ow[ci], rows, chans, dgrad ? 1 : 0) ? 1 : 0;
}
// everything the choice of a dgrad filter bank's FORM depends on (Winograd mode, precision mode, fp16x3 fill rule, routing batch):
// a change between a forward and its backward re-packs the banks at the start of the backward
static int pack_key(const uwm_model* m) {
  return ((m->plan.wino_mode * 8 + m->plan.prec) * 4099 + (m->f3_min_wgs & 0xfff)) * 257 + (m->route_n & 0xff);
}
// ------------------------------------------------------------------------------ launch helpers
struct Ctx {
  uwm_model* m; float* ws; hipStream_t st; int N;
  hipStream_t wst = nullptr;          // stream for wgrad launches (== st when the side stream is off)
  float* F(size_t off) const { return ws + off; }
  double* D(size_t doff) const { return (double*)ws + doff; }
};
static Src mk_src(const float* ptr, int C, int H, int W, const float* scale = nullptr, const float* shift = nullptr,
                  int relu = 0, int up = 0) {
  Src s; s.ptr = ptr; s.scale = scale; s.shift = shift; s.C = C; s.H = H; s.W = W; s.up = up; s.relu = relu; return s;
}
static Src lazy_src(const Ctx& c, int conv, int H, int W, int relu = 1, int up = 0) {
  const ConvL& cv = c.m->convs[conv];
  const BNL& b = c.m->bns[cv.bn];
  return mk_src(c.F(c.m->plan.y[conv]), cv.CoutP, H, W, c.F(b.f_off) + 2 * b.C, c.F(b.f_off) + 3 * b.C, relu, up);
}

// routing record (uwm_routing_enable / uwm_routing_dump): "<pass> <layer> <kernel>" per conv-class launch, in launch order
static hipError_t route_rec(const Ctx& c, const char* pass, int ci, hipError_t e) {
  if (c.m->route_log_on && e == hipSuccess) { c.m->route_log += pass; c.m->route_log += ' '; c.m->route_log += c.m->convs[ci].name; c.m->route_log += ' '; c.m->route_log += route_last(); c.m->route_log += '\n'; }
  return e;
}

static hipError_t run_conv_fwd(const Ctx& c, int ci, const Src& s0, const Src* s1, int Ho, int Wo, float* out,
                               bool stats, int cfg = -1) {
  const ConvL& cv = c.m->convs[ci];
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.s0 = s0; a.C0 = s0.C;
  if (s1) { a.s1 = *s1; a.Ctot = s0.C + s1->C; } else { a.s1 = s0; a.Ctot = s0.C; }
  a.w = c.m->params + cv.w_off; a.wrows = cv.Cout; a.Kpad = cv.Kpad; a.ntaps = cv.k * cv.k; a.kw = cv.k;
  a.N = c.N; a.Ho = Ho; a.Wo = Wo; a.Cout = cv.CoutP; a.M = c.N * Ho * Wo;
  a.Hl = s0.H << s0.up; a.Wl = s0.W << s0.up;
  a.smul = cv.stride; a.rmul = 1; a.off = -cv.pad; a.sdiv = 1;
  a.out = out; a.bias = cv.bias_off >= 0 ? c.m->params + cv.bias_off : nullptr;
  if (stats && cv.bn >= 0) {
    const BNL& b = c.m->bns[cv.bn];
    a.ssum = c.D(b.d_off) + 2 * b.C; a.ssq = a.ssum + b.C; a.srep = b.nrep; a.sstride = 2 * b.C;
  }
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  a.flops = 2.0 * (double)a.M * cv.Cout * cv.Cin * cv.k * cv.k;
  a.bytes = 4.0 * ((double)c.N * s0.H * s0.W * s0.C + (s1 ? (double)c.N * s1->H * s1->W * s1->C : 0.0) + (double)cv.Cout * cv.Kpad +
                   (double)a.M * cv.CoutP);
  if (ci == c.m->stem && cfg < 0 && stem_f3_on(c.m)) {
    a.wu = c.F(cv.wu_off);
    if (conv_stem_f16x3_applicable(a)) return route_rec(c, "fwd", ci, launch_conv_stem_f16x3(a, c.st));
    a.wu = nullptr;
  }
  if (cv.wu_off && !cv.stem7() && a.Ctot == cv.CinP && c.m->plan.wino_ok((size_t)ci)) {
    a.wu = c.F(cv.wu_off); a.wu_ncb = wino_ncb(cv.Cout);
    if (c.m->plan.prec == UWM_PREC_BF16X3_ALL && cv.x3()) {
      if (a.C0 != cv.c0 && a.C0 != a.Ctot) return hipErrorInvalidValue;       // the bank was split for this concat boundary
      a.prec = 1;
    }
    if (f3_fwd_on(c.m, (size_t)ci)) {
      a.prec = 2; a.wu_layout = f3_layout(c.m, (size_t)ci, false); a.nprod = f3_nprod(c.m, false);
      a.wu_ncb = a.wu_layout == 1 ? f16x3v2_nf(cv.Cout) : f16x3_nj(cv.Cout); a.wu_rinv_off = (int)f16x3_rinv_off(cv.Cout, cv.CinP);
    }
  }
  a.wino = c.m->plan.wino_mode + 1; a.route_n = c.m->route_n;
  return route_rec(c, "fwd", ci, launch_conv(a, c.st, cfg));
}

struct UpSplit { float* gprev; int C0; const float* pmask; const float* pscale; const float* pshift; int accumulate = 0; };
// dX = dgrad(dY) (+addend) (*mask); with `us`: decoder concat split fused into the epilogue (dx = gskip or nullptr)
// bn_fuse >= 0: the masked output of this dgrad is the gradient wrt the output of BatchNorm `bn_fuse`, whose raw input is the
// ReLU mask the epilogue reads anyway: where the launch runs on a Winograd epilogue the BatchNorm-backward sums (dbeta,
// dgamma partials, one of nrep replicas per workgroup) are accumulated there and *fused = true tells run_bn_bwd to skip
// its reduce pass.  bn_y: yhat comes from this tensor instead of the mask tensor (gradient wrt a residual block's output: masked by the
// output, but the BatchNorm in question is the block's last one, whose raw input is bn_y)
static hipError_t run_dgrad(const Ctx& c, int ci, const float* dy, int Ho, int Wo, int Hin, int Win, float* dx,
                            const float* addend, const float* mask, const float* mscale, const float* mshift,
                            const UpSplit* us = nullptr, int bn_fuse = -1, bool* fused = nullptr, const float* bn_y = nullptr) {
  const ConvL& cv = c.m->convs[ci];
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.s0 = mk_src(dy, cv.CoutP, Ho, Wo); a.s1 = a.s0; a.C0 = cv.CoutP; a.Ctot = cv.CoutP;
  a.w = c.F(cv.wd_off); a.wrows = cv.CinP; a.Kpad = cv.KpadD; a.ntaps = cv.k * cv.k; a.kw = cv.k;
  a.N = c.N; a.Ho = Hin; a.Wo = Win; a.Cout = cv.CinP; a.M = c.N * Hin * Win;
  a.Hl = Ho; a.Wl = Wo; a.smul = 1; a.rmul = -1; a.off = cv.pad; a.sdiv = cv.stride;
  a.out = dx; a.addend = addend; a.mask = mask; a.mscale = mscale; a.mshift = mshift; a.live_ch = cv.Cout;
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  a.flops = 2.0 * (double)c.N * Ho * Wo * cv.Cout * cv.Cin * cv.k * cv.k;   // same MACs as the forward conv
  a.bytes = 4.0 * ((double)c.N * Ho * Wo * cv.CoutP + (double)cv.CinP * cv.KpadD +
                   (double)a.M * cv.CinP * (1.0 + (addend ? 1.0 : 0.0) + (mask ? 1.0 : 0.0)));
  if (cv.wud_off && c.m->plan.wino_ok((size_t)ci)) {
    a.wu = c.F(cv.wud_off); a.wu_ncb = wino_ncb(cv.CinP);
    const int pm = c.m->plan.prec;
    if ((pm == UWM_PREC_BF16X3 || pm == UWM_PREC_BF16X3_ALL) && cv.x3_d()) a.prec = 1;
    if (f3_dgrad_on(c.m, (size_t)ci) && (!us || (us->C0 & 63) == 0)) {        // fp16x3 direct form: dY scaled by the power of two bn_bwd_apply's max|dy| calls for
      a.prec = 2; a.wu_layout = f3_layout(c.m, (size_t)ci, true); a.nprod = f3_nprod(c.m, true);
      a.wu_ncb = a.wu_layout == 1 ? f16x3v2_nf(cv.CinP) : f16x3_nj(cv.CinP); a.wu_rinv_off = (int)f16x3_rinv_off(cv.CinP, cv.CoutP);
      a.xmax = (const float*)c.D(c.m->bns[cv.bn].xmax_off());
    }
  }
  a.wino = c.m->plan.wino_mode + 1;
  if (us) { a.out_up = us->gprev; a.up_c0 = us->C0; a.up_mask = us->pmask; a.up_mscale = us->pscale; a.up_mshift = us->pshift; a.up_accum = us->accumulate; }
  if (fused) *fused = false;
  static const bool no_fuse = dbg_flag("UWM_NO_BN_FUSE");
  a.bnb_y = bn_y;                                  // (consulted by conv_epilogue_carries_bnb; cleared again when the sums are not fused)
  if (bn_fuse >= 0 && fused && !no_fuse && (us ? (us->pmask != nullptr && !us->accumulate) : (mask != nullptr || bn_y != nullptr)) && conv_epilogue_carries_bnb(a)) {
    const BNL& b = c.m->bns[bn_fuse];
    if (b.C == (us ? us->C0 : cv.CinP)) {
      const float* f = c.F(b.f_off);
      a.bnb_mean = f; a.bnb_rstd = f + b.C;
      a.ssum = c.D(b.d_off) + 2 * b.C; a.ssq = a.ssum + b.C; a.srep = b.nrep; a.sstride = 2 * b.C;     // the forward's (dead, re-zeroed) replicas
      *fused = true;
    }
  }
  if (!fused || !*fused) a.bnb_y = nullptr;
  a.route_n = c.m->route_n;
  return route_rec(c, "dgrad", ci, launch_conv(a, c.st));
}

static hipError_t run_wgrad(const Ctx& c, int ci, const Src& s0, const Src* s1, const float* dy, int Ho, int Wo) {
  const ConvL& cv = c.m->convs[ci];
  WgradArgs a; memset(&a, 0, sizeof(a));
  a.s0 = s0; a.C0 = s0.C;
  if (s1) { a.s1 = *s1; a.Ctot = s0.C + s1->C; } else { a.s1 = s0; a.Ctot = s0.C; }
  a.dy = dy; a.dw = c.m->grads + cv.w_off; a.wrows = cv.Cout; a.Kpad = cv.Kpad; a.ntaps = cv.k * cv.k; a.kw = cv.k;
  a.N = c.N; a.Ho = Ho; a.Wo = Wo; a.Cout = cv.CoutP; a.M = c.N * Ho * Wo;
  a.Hl = s0.H << s0.up; a.Wl = s0.W << s0.up; a.stride = cv.stride; a.pad = cv.pad;
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  a.flops = 2.0 * (double)a.M * cv.Cout * cv.Cin * cv.k * cv.k;
  a.bytes = 4.0 * ((double)a.M * cv.CoutP + (double)c.N * s0.H * s0.W * s0.C + (s1 ? (double)c.N * s1->H * s1->W * s1->C : 0.0) +
                   (double)cv.Cout * cv.Kpad);
  a.wino = c.m->plan.wino_mode + 1; a.route_n = c.m->route_n;
  if (f3_wgrad_on(c.m, (size_t)ci)) { a.prec = 2; a.nprod = f3_nprod(c.m, true); a.xmax = (const float*)c.D(c.m->bns[cv.bn].xmax_off()); a.cu_share = (c.wst && c.wst != c.st) ? 3 : 0; }
  // partial images of a split launch: the next free slice of the scratch; their reduce is queued and runs with the other layers'
  // in one launch (flush_reduces: when the scratch / queue fills up and at the end of every backward stage)
  static const bool no_defer = dbg_flag("UWM_NO_DEFER_REDUCE");
  ReduceQueue& rq = c.m->rq;
  const size_t cap = wgrad_wino_scratch_floats();
  if (rq.n + 1 >= ReduceQueue::kMax || rq.used_floats + cap > kWgParts * cap) {
    hipError_t e = launch_wgrad_reduce_multi(rq, (c.wst && c.wst != c.st) ? c.wst : c.st);
    if (e != hipSuccess) return e;
  }
  a.part = c.F(c.m->plan.wg_part) + rq.used_floats; a.part_floats = cap; a.rq = no_defer ? nullptr : &rq;
  if (c.wst && c.wst != c.st) {
    // fork: the side stream must see everything enqueued so far on the main stream (dy, activations).  Where dy came out of a
    // bn_bwd_apply dispatch that carried ev_disp, that event IS the fork point
    const bool covered = std::find(c.m->disp_cov.begin(), c.m->disp_cov.end(), dy) != c.m->disp_cov.end();
    hipError_t e = hipSuccess;
    if (!covered) e = hipEventRecord(c.m->ev_fork, c.st);
    if (e != hipSuccess) return e;
    e = hipStreamWaitEvent(c.wst, covered ? c.m->ev_disp : c.m->ev_fork, 0);
    if (e != hipSuccess) return e;
    return route_rec(c, "wgrad", ci, launch_wgrad(a, c.wst));
  }
  return route_rec(c, "wgrad", ci, launch_wgrad(a, c.st));
}

static hipError_t run_bn_finalize(const Ctx& c, int bi, size_t count, int training) {
  uwm_model* m = c.m; const BNL& b = m->bns[bi];
  float* f = c.F(b.f_off);
  const float eps = b.eps > 0.f ? b.eps : m->desc.bn_eps, mom = b.mom > 0.f ? b.mom : m->desc.bn_momentum;
  if (training)
    return launch_bn_finalize(c.D(b.d_off) + 2 * b.C, c.D(b.d_off) + 3 * b.C, m->params + b.g_off, m->params + b.b_off,
                              m->buffers + b.rm_off, m->buffers + b.rv_off, f, f + b.C, f + 2 * b.C, f + 3 * b.C, b.C,
                              (double)count, eps, mom, 1, c.st, b.nrep, 2 * b.C);
  return launch_bn_eval(m->params + b.g_off, m->params + b.b_off, m->buffers + b.rm_off, m->buffers + b.rv_off,
                        f + 2 * b.C, f + 3 * b.C, b.C, eps, c.st);
}

// g (masked grad wrt BN output) -> dy ; also writes gamma/beta gradients
static hipError_t run_bn_bwd(const Ctx& c, int ci, const float* g, float* dy, size_t npix, bool sums_fused = false) {
  uwm_model* m = c.m; const ConvL& cv = m->convs[ci]; const BNL& b = m->bns[cv.bn];
  const float* f = c.F(b.f_off); const float* y = c.F(m->plan.y[ci]);
  double* dg = c.D(b.d_off); double* db = c.D(b.d_off) + b.C;
  // sums_fused: the dgrad that produced g left per-workgroup-replica partial sums behind (run_dgrad bn_fuse): the apply pass adds
  // the replicas up in its own prologue (no fold launch between the dgrad and the apply)
  hipEvent_t done = nullptr;
  if (c.wst && c.wst != c.st && m->ev_disp && m->disp_fork) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c.st, &cs) != hipSuccess) { cs = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    if (cs == hipStreamCaptureStatusNone) done = m->ev_disp;          // (a capturing stream takes the plain record / wait pair: graph edges)
  }
  hipError_t e = hipSuccess;
  float* xmax = (f3_dgrad_on(m, (size_t)ci) || f3_wgrad_on(m, (size_t)ci)) ? (float*)c.D(b.xmax_off()) : nullptr;      // the fp16x3 dgrad / wgrad of this conv scale dy by its maximum
  if (sums_fused)
    e = launch_bn_bwd_apply(g, y, f, f + b.C, m->params + b.g_off, nullptr, nullptr, dy, m->grads + b.g_off, m->grads + b.b_off,
                            npix, b.C, c.st, c.D(b.d_off) + 2 * b.C, b.nrep, 2 * b.C, done, xmax);
  else {
    e = launch_bn_bwd_reduce(g, y, f, f + b.C, dg, db, npix, b.C, c.st);
    if (e != hipSuccess) return e;
    e = launch_bn_bwd_apply(g, y, f, f + b.C, m->params + b.g_off, dg, db, dy, m->grads + b.g_off, m->grads + b.b_off,
                            npix, b.C, c.st, nullptr, 0, 0, done, xmax);
  }
  if (e == hipSuccess && done) m->disp_cov.push_back(dy);
  return e;
}

// BatchNorm backward of conv ci whose output went through swish [and the SE product]: g = grad wrt that activation
static hipError_t run_bn_bwd_act(const Ctx& c, int ci, const float* g, float* dy, int N, size_t hw, const float* se_s, const float* gpool) {
  uwm_model* m = c.m; const ConvL& cv = m->convs[ci]; const BNL& b = m->bns[cv.bn];
  const float* f = c.F(b.f_off);
  return launch_bn_bwd_act(g, c.F(m->plan.y[ci]), f, f + b.C, m->params + b.g_off, f + 2 * b.C, f + 3 * b.C, se_s, gpool, N, hw,
                           c.D(b.d_off), c.D(b.d_off) + b.C, dy, m->grads + b.g_off, m->grads + b.b_off, b.C, c.st);
}

#define LCHK(expr)                                                                                     \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("launch failed: %s at %s:%d (%s)",  \
       hipGetErrorString(e_), __FILE__, __LINE__, #expr); } while (0)

// Winograd filter transforms of every eligible layer (forward banks, or dgrad banks straight from the forward
// weights), at most 40 layers per launch
static hipError_t wino_jobs(const Ctx& c, bool dgrad, hipStream_t st) {
  if (c.m->plan.wino_mode == 0) return hipSuccess;
  const uwm_model* m = c.m;
  // three passes: fp32 banks, then (bf16x3 modes) the split-bf16 banks of the layers that run on conv_wino_x3, then (fp16x3
  // modes) the split-fp16 banks of the layers that run on conv_f16x3
  const int prec = m->plan.prec;
  const bool bf = prec == UWM_PREC_BF16X3 || prec == UWM_PREC_BF16X3_ALL;
  for (int x3 = 0; x3 <= 2; ++x3) {
    WinoJobs jobs; jobs.n = 0;
    auto flush = [&]() { hipError_t e = x3 == 2 ? launch_f16x3_weights_multi(jobs, st) : (x3 ? launch_wino_weights_x3_multi(jobs, st) : launch_wino_weights_multi(jobs, st)); jobs.n = 0; return e; };
    for (size_t ci = 0; ci < m->convs.size(); ++ci) {
      const ConvL& cv = m->convs[ci];
      if (!m->plan.wino_ok(ci) || !(dgrad ? cv.wud_off : cv.wu_off) || cv.stem7()) continue;      // (the stem's slot holds conv_stem_f16x3's bank, built by its own kernel)
      int kind = 0;
      if (dgrad) { if (bf && cv.x3_d()) kind = 1; if (f3_dgrad_on(m, ci)) kind = 2; }
      else { if (prec == UWM_PREC_BF16X3_ALL && cv.x3()) kind = 1; if (f3_fwd_on(m, ci)) kind = 2; }
      if (kind != x3) continue;
      WinoJob& j = jobs.j[jobs.n++];
      j.w = m->params + cv.w_off; j.Kpad = cv.Kpad; j.pad_ = kind == 2 ? f3_layout(m, ci, dgrad) : 0;
      if (dgrad) { j.ut = c.F(cv.wud_off); j.rows = cv.CinP; j.chans = cv.CoutP; j.mode = 2; j.src_rows = cv.Cout; }
      else { j.ut = c.F(cv.wu_off); j.rows = cv.Cout; j.chans = cv.CinP; j.mode = 0; j.src_rows = cv.Cout; }
      if (jobs.n == 40) { hipError_t e = flush(); if (e != hipSuccess) return e; }
    }
    hipError_t e = flush();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------ forward
static int do_forward(uwm_model* m, const float* x, float* logits, float* ws, int N, int H, int W, int training,
                      hipStream_t st) {
  const Plan& p = m->plan;
  Ctx c{m, ws, st, N};
  if (m->prec_from_env) {               // say once which arithmetic a process default switched on (a stray variable must not go unnoticed)
    static const char* names[7] = {"f32", "bf16x3", "bf16x3_all", "f16x3", "f16x3_all", "f16x1", "f16x3_bwd2"};
    fprintf(stderr, "libuwm: precision mode %s for this handle comes from UWM_PRECISION (uwm_set_precision overrides it)\n", names[m->prec]);
    m->prec_from_env = false;
  }
  // a previous training forward that was never followed by a backward left its dgrad repacks on the side stream with
  // nothing joined to them: this stream must not touch the workspace (re-planned, re-used or re-allocated) before they land
  if (m->packed_in_fwd) HIPCHK(hipStreamWaitEvent(st, m->ev_pack, 0));
  if (training) HIPCHK(hipMemsetAsync(c.D(p.stat_d), 0, p.stat_d_count * sizeof(double), st));
  m->packed_in_fwd = false;
  if (training && m->use_side && m->side && m->grads) {
    // the [Cin][tap][Cout] weight repacks the backward's dgrads need depend only on the parameters: enqueue them
    // on the side stream now (after everything already on the caller's stream, i.e. after the last optimizer step)
    // so they cost nothing; uwm_backward waits for ev_pack
    HIPCHK(hipEventRecord(m->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    for (size_t ci = 0; ci < m->convs.size(); ++ci) {
      const ConvL& cv = m->convs[ci];
      if (cv.dgrad && (!(cv.wud_off && p.wino_ok(ci)) || cv.CoutP == 16))      // Winograd dgrads take their filters from wino_jobs below; 16-channel dY goes to conv_patch16 (packed bank)
        LCHK(launch_pack_dgrad(m->params + cv.w_off, cv.Cout, cv.Kpad, cv.k * cv.k, cv.CinP, c.F(cv.wd_off), cv.KpadD,
                               cv.CoutP, m->side));
    }
    LCHK(wino_jobs(c, true, m->side));
    HIPCHK(hipEventRecord(m->ev_pack, m->side));
    m->packed_in_fwd = true; m->pack_mode = pack_key(m);
  }
  LCHK(wino_jobs(c, false, st));
  if (stem_f3_on(m)) { const ConvL& sv = m->convs[m->stem]; LCHK(launch_stem_f16x3_weights(m->params + sv.w_off, sv.Kpad, sv.CinP, c.F(sv.wu_off), st)); }
  LCHK(launch_nchw_to_nhwc4(x, c.F(p.x4), N, m->desc.in_channels, H, W, m->CinP, st));
  // eval: all BN scale/shift come from running stats and are known up front
  if (!training) for (size_t i = 0; i < m->bns.size(); ++i) LCHK(run_bn_finalize(c, (int)i, 1, 0));
  auto conv_bn = [&](int ci, const Src& s0, const Src* s1, int Ho, int Wo) -> int {
    LCHK(run_conv_fwd(c, ci, s0, s1, Ho, Wo, c.F(p.y[ci]), training != 0));
    if (training) LCHK(run_bn_finalize(c, m->convs[ci].bn, (size_t)N * Ho * Wo, 1));
    return 0;
  };
  int h = H / 2, w = W / 2;
  Src x4 = mk_src(c.F(p.x4), m->CinP, H, W);
  if (conv_bn(m->stem, x4, nullptr, h, w)) return 1;
  Src f1 = lazy_src(c, m->stem, h, w);
  const int h1 = h, w1 = w;
  Src feats[4];
  if (m->mb.empty()) {
  h /= 2; w /= 2;
  LCHK(launch_maxpool_fwd(f1, c.F(p.pool), training ? (uint8_t*)c.F(p.pool_idx) : nullptr, N, h, w, st));
  Src cur = mk_src(c.F(p.pool), 64, h, w);
  size_t bi = 0;
  for (int s = 0; s < 4; ++s) {
    for (auto& bl : m->stages[s]) {
      const int ho = h / bl.stride, wo = w / bl.stride;
      if (bl.c3 < 0) {
        if (conv_bn(bl.c1, cur, nullptr, ho, wo)) return 1;
        Src a1 = lazy_src(c, bl.c1, ho, wo);
        if (conv_bn(bl.c2, a1, nullptr, ho, wo)) return 1;
      } else {
        if (conv_bn(bl.c1, cur, nullptr, h, w)) return 1;
        Src a1 = lazy_src(c, bl.c1, h, w);
        if (conv_bn(bl.c2, a1, nullptr, ho, wo)) return 1;
        Src a2 = lazy_src(c, bl.c2, ho, wo);
        if (conv_bn(bl.c3, a2, nullptr, ho, wo)) return 1;
      }
      const int lc = bl.last();
      const BNL& b2 = m->bns[m->convs[lc].bn];
      const float *idp = cur.ptr, *sd = nullptr, *bd = nullptr;
      if (bl.cd >= 0) {
        if (conv_bn(bl.cd, cur, nullptr, ho, wo)) return 1;
        const BNL& bdn = m->bns[m->convs[bl.cd].bn];
        idp = c.F(p.y[bl.cd]); sd = c.F(bdn.f_off) + 2 * bdn.C; bd = c.F(bdn.f_off) + 3 * bdn.C;
      }
      LCHK(launch_residual(c.F(p.y[lc]), c.F(b2.f_off) + 2 * b2.C, c.F(b2.f_off) + 3 * b2.C, idp, sd, bd,
                           c.F(p.xn[bi]), (size_t)N * ho * wo, bl.Cout, st));
      h = ho; w = wo;
      cur = mk_src(c.F(p.xn[bi]), bl.Cout, h, w);
      ++bi;
    }
    feats[s] = cur;
  }
  } else {
    // ---- EfficientNet encoder: every stage of an MBConv block materialised (round-1 correctness-first path)
    auto bn_ss = [&](int ci, const float** sc, const float** sh_) { const BNL& b = m->bns[m->convs[ci].bn]; *sc = c.F(b.f_off) + 2 * b.C; *sh_ = c.F(b.f_off) + 3 * b.C; };
    const float *sc, *sf;
    bn_ss(m->stem, &sc, &sf);
    LCHK(launch_swish_fwd(c.F(p.y[m->stem]), sc, sf, m->f1C, c.F(p.stem_a), (size_t)N * h * w, st));
    f1 = mk_src(c.F(p.stem_a), m->f1C, h, w);
    m->keep_fwd = training ? m->keep : nullptr;
    Src cur = f1;
    for (size_t bi = 0; bi < m->mb.size(); ++bi) {
      const MBL& b = m->mb[bi];
      const int ho = h / b.stride, wo = w / b.stride;
      if (ho != p.mh[bi] || wo != p.mw[bi]) return fail("internal: MBConv block %d plan mismatch", (int)bi);
      const float* dwin = cur.ptr;
      if (b.ce >= 0) {
        if (conv_bn(b.ce, cur, nullptr, h, w)) return 1;
        bn_ss(b.ce, &sc, &sf);
        LCHK(launch_swish_fwd(c.F(p.y[b.ce]), sc, sf, b.mid, c.F(p.a0[bi]), (size_t)N * h * w, st));
        dwin = c.F(p.a0[bi]);
      }
      const ConvL& dw = m->convs[b.cdw];
      const size_t npo = (size_t)N * ho * wo;
      LCHK(launch_dw_fwd(dwin, m->params + dw.w_off, b.k, b.stride, b.pb, N, h, w, b.mid, ho, wo, c.F(p.y[b.cdw]), st));
      if (training) {
        const BNL& b1 = m->bns[dw.bn];
        LCHK(launch_colstats(c.F(p.y[b.cdw]), npo, b.mid, c.D(b1.d_off) + 2 * b1.C, c.D(b1.d_off) + 3 * b1.C, st));
        LCHK(run_bn_finalize(c, dw.bn, npo, 1));
      }
      bn_ss(b.cdw, &sc, &sf);
      float* pool = c.F(p.se_pool[bi]); float* sv = c.F(p.se[bi]); float* hpre = sv + (size_t)N * b.mid;
      LCHK(launch_swish_pool(c.F(p.y[b.cdw]), sc, sf, c.F(p.a1[bi]), N, (size_t)ho * wo, b.mid, pool, c.F(p.se_part), st));
      const ConvL& cr = m->convs[b.cr]; const ConvL& cx = m->convs[b.cx];
      LCHK(launch_se_fc_fwd(pool, m->params + cr.w_off, m->params + cr.bias_off, cr.Kpad, m->params + cx.w_off,
                            m->params + cx.bias_off, cx.Kpad, N, b.mid, b.nsq, hpre, hpre + (size_t)N * rup(b.nsq, 4), sv, st));
      LCHK(launch_se_scale(c.F(p.a1[bi]), sv, N, (size_t)ho * wo, b.mid, c.F(p.a2[bi]), st));
      Src a2 = mk_src(c.F(p.a2[bi]), b.mid, ho, wo);
      if (conv_bn(b.cp, a2, nullptr, ho, wo)) return 1;
      bn_ss(b.cp, &sc, &sf);
      const float* rs = (b.skip && b.drop > 0.f && m->keep_fwd) ? m->keep_fwd + bi * (size_t)N : nullptr;
      LCHK(launch_mb_out(c.F(p.y[b.cp]), sc, sf, rs, b.skip ? cur.ptr : nullptr, N, (size_t)ho * wo, b.Cout, c.F(p.xn[bi]), st));
      h = ho; w = wo;
      cur = mk_src(c.F(p.xn[bi]), b.Cout, h, w);
      for (int fs = 0; fs < 4; ++fs) if ((int)bi == m->feat_blk[fs]) feats[fs] = cur;
    }
  }
  Src d = feats[3];
  for (size_t i = 0; i < m->dec.size(); ++i) {
    const DecL& dl = m->dec[i];
    Src up = d; up.up = 1;
    h *= 2; w *= 2;
    Src skip; const Src* sp = nullptr;
    if (i < 3) { skip = feats[2 - i]; sp = &skip; }
    else if (i == 3) { skip = f1; sp = &skip; }
    if (sp && (skip.H != h || skip.W != w)) return fail("internal: skip shape mismatch at decoder block %d", (int)i);
    if (conv_bn(dl.c1, up, sp, h, w)) return 1;
    Src a1 = lazy_src(c, dl.c1, h, w);
    if (conv_bn(dl.c2, a1, nullptr, h, w)) return 1;
    d = lazy_src(c, dl.c2, h, w);
  }
  // UnetPlusPlus: dense grid of the same blocks; a node with several skip tensors gets them materialised (activation
  // applied) into one concat buffer, so every conv still sees two sources
  auto tensor_src = [&](int id) -> Src {
    if (id == 0) return f1;
    if (id < 5) return feats[id - 1];
    const NodeL& nd = m->nodes[id - 5];
    return lazy_src(c, nd.c2, H >> nd.lvl, W >> nd.lvl);
  };
  for (size_t i = 0; i < m->nodes.size(); ++i) {
    const NodeL& nd = m->nodes[i];
    h = H >> nd.lvl; w = W >> nd.lvl;
    Src up = tensor_src(nd.prev); up.up = 1;
    if (up.C != nd.C0 || (up.H << 1) != h || (up.W << 1) != w) return fail("internal: UnetPlusPlus node %d input mismatch", (int)i);
    Src skip; const Src* sp = nullptr;
    if (nd.skips.size() == 1) { skip = tensor_src(nd.skips[0]); sp = &skip; }
    else if (nd.skips.size() > 1) {
      int coff = 0;
      for (int id : nd.skips) {
        const Src t = tensor_src(id);
        if (t.H != h || t.W != w) return fail("internal: UnetPlusPlus node %d skip shape mismatch", (int)i);
        LCHK(launch_concat_copy(t, (size_t)N * h * w, c.F(p.cat[i]), nd.C1, coff, st));
        coff += t.C;
      }
      if (coff != nd.C1) return fail("internal: UnetPlusPlus node %d skip channels %d != %d", (int)i, coff, nd.C1);
      skip = mk_src(c.F(p.cat[i]), nd.C1, h, w); sp = &skip;
    }
    if (sp && (sp->C != nd.C1)) return fail("internal: UnetPlusPlus node %d skip channel mismatch", (int)i);
    if (conv_bn(nd.c1, up, sp, h, w)) return 1;
    Src a1 = lazy_src(c, nd.c1, h, w);
    if (conv_bn(nd.c2, a1, nullptr, h, w)) return 1;
    d = lazy_src(c, nd.c2, h, w);
  }
  (void)h1; (void)w1;
  LCHK(run_conv_fwd(c, m->head, d, nullptr, h, w, logits, false));
  return 0;
}

// ------------------------------------------------------------------------------ backward
static int do_backward(uwm_model* m, const float* dlogits, float* ws, int sb, int se, hipStream_t st) {
  const Plan& p = m->plan;
  const int N = p.N, H = p.H, W = p.W;
  Ctx c{m, ws, st, N};
  c.wst = (m->use_side && m->side) ? m->side : st;
  // geometry
  const int h1 = H / 2, w1 = W / 2;          // f1
  int sh[4], sw[4];
  { int h = H / 4, w = W / 4; for (int s = 0; s < 4; ++s) { if (s > 0) { h /= 2; w /= 2; } sh[s] = h; sw[s] = w; } }
  std::vector<size_t> first_blk(4); { size_t b = 0; for (int s = 0; s < 4; ++s) { first_blk[s] = b; b += m->stages[s].size(); } }
  const bool effnet = !m->mb.empty();
  auto feat_blk = [&](int s) -> size_t { return effnet ? (size_t)m->feat_blk[s] : first_blk[s] + m->stages[s].size() - 1; };
  auto feat_src = [&](int s) {      // materialised output of encoder stage s (f2..f5)
    return mk_src(c.F(p.xn[feat_blk(s)]), m->featC[s], sh[s], sw[s]);
  };
  // ResNet features are ReLU outputs (the consumer's gradient is masked by feature > 0); EfficientNet's are linear
  auto feat_mask = [&](const Src& f) -> const float* { return effnet ? nullptr : f.ptr; };
  Src f1 = effnet ? mk_src(c.F(p.stem_a), m->f1C, h1, w1) : lazy_src(c, m->stem, h1, w1);

  std::vector<char>& out_sums = m->out_sums;
  if (sb <= 0 || out_sums.size() != first_blk[3] + m->stages[3].size()) out_sums.assign(first_blk[3] + m->stages[3].size(), 0);
  m->disp_cov.clear();
  if (sb <= 0) m->rq = ReduceQueue();                   // (a failed earlier call may have left jobs behind)
  if (sb <= 0 && se > 0) {
    HIPCHK(hipMemsetAsync(m->grads, 0, (size_t)m->param_floats * sizeof(float), st));
    // one memset for every BatchNorm's double scratch (the forward's sum/sumsq halves are dead after bn_finalize)
    HIPCHK(hipMemsetAsync(c.D(p.stat_d), 0, p.stat_d_count * sizeof(double), st));
    if (p.se_gs_floats) HIPCHK(hipMemsetAsync(c.F(p.se_gs_all), 0, p.se_gs_floats * sizeof(float), st));
    if (m->packed_in_fwd && m->pack_mode == pack_key(m)) {      // (a mode switch between forward and backward: redo them)
      HIPCHK(hipStreamWaitEvent(st, m->ev_pack, 0));
      m->packed_in_fwd = false;           // joined
    } else {
      if (m->packed_in_fwd) { HIPCHK(hipStreamWaitEvent(st, m->ev_pack, 0)); m->packed_in_fwd = false; }     // stale packs must have landed before they are overwritten
      for (size_t ci = 0; ci < m->convs.size(); ++ci) {
        const ConvL& cv = m->convs[ci];
        if (cv.dgrad && (!(cv.wud_off && p.wino_ok(ci)) || cv.CoutP == 16))
          LCHK(launch_pack_dgrad(m->params + cv.w_off, cv.Cout, cv.Kpad, cv.k * cv.k, cv.CinP, c.F(cv.wd_off), cv.KpadD,
                                 cv.CoutP, st));
      }
      LCHK(wino_jobs(c, true, st));
    }
    // ---------------- head
    const ConvL& hd = m->convs[m->head];
    const int last_c2 = m->nodes.empty() ? m->dec.back().c2 : m->nodes.back().c2;
    Src d4 = lazy_src(c, last_c2, H, W);
    LCHK(run_wgrad(c, m->head, d4, nullptr, dlogits, H, W));
    LCHK(launch_colsum(dlogits, (size_t)N * H * W, hd.CoutP, m->grads + hd.bias_off, (double*)c.F(p.colsum_scr), st));
    bool head_sums = false;                               // BatchNorm-backward sums of the last decoder conv, made by the head's dgrad
    LCHK(run_dgrad(c, m->head, dlogits, H, W, H, W, c.F(p.g[last_c2]), nullptr, d4.ptr, d4.scale, d4.shift, nullptr,
                   m->convs[last_c2].bn, &head_sums));
    // ---------------- decoder blocks, last to first
    int h = H, w = W;
    bool c2_sums = head_sums;                              // BatchNorm-backward sums of dl.c2 already made by the dgrad that wrote its gradient
    for (int i = (int)m->dec.size() - 1; i >= 0; --i) {
      const DecL& dl = m->dec[i];
      const size_t npix = (size_t)N * h * w;
      LCHK(run_bn_bwd(c, dl.c2, c.F(p.g[dl.c2]), c.F(p.g[dl.c2]), npix, c2_sums));
      c2_sums = false;
      Src a1 = lazy_src(c, dl.c1, h, w);
      LCHK(run_wgrad(c, dl.c2, a1, nullptr, c.F(p.g[dl.c2]), h, w));
      bool c1_sums = false;
      LCHK(run_dgrad(c, dl.c2, c.F(p.g[dl.c2]), h, w, h, w, c.F(p.g[dl.c1]), nullptr, a1.ptr, a1.scale, a1.shift, nullptr,
                     m->convs[dl.c1].bn, &c1_sums));
      LCHK(run_bn_bwd(c, dl.c1, c.F(p.g[dl.c1]), c.F(p.g[dl.c1]), npix, c1_sums));
      // the block's input: cat(up(prev), skip)
      Src prev = (i == 0) ? feat_src(3) : lazy_src(c, m->dec[i - 1].c2, h / 2, w / 2);
      prev.up = 1;
      Src skip; const Src* sp = nullptr;
      if (i < 3) { skip = feat_src(2 - i); sp = &skip; } else if (i == 3) { skip = f1; sp = &skip; }
      LCHK(run_wgrad(c, dl.c1, prev, sp, c.F(p.g[dl.c1]), h, w));
      float* gprev = (i == 0) ? c.F(p.gx[feat_blk(3)]) : c.F(p.g[m->dec[i - 1].c2]);
      const float* pm = (i == 0) ? feat_mask(prev) : prev.ptr;
      const ConvL& c1v = m->convs[dl.c1];
      if (c1v.wud_off && p.wino_ok((size_t)dl.c1) && !(h & 1) && !(w & 1)) {
        // Winograd dgrad writes the 2x2-pooled, ReLU-masked gradient of up(prev) and the skip gradient directly:
        // the full-resolution dcat buffer and the upsplit pass never exist
        UpSplit us{gprev, dl.C0, pm, prev.scale, prev.shift, 0};
        // i > 0: gprev is the gradient wrt relu(bn(conv2 of block i-1)), masked by that conv's raw output: its BN sums ride along
        LCHK(run_dgrad(c, dl.c1, c.F(p.g[dl.c1]), h, w, h, w, dl.C1 > 0 ? c.F(p.gskip[i]) : nullptr, nullptr, nullptr, nullptr,
                       nullptr, &us, i > 0 ? m->convs[m->dec[i - 1].c2].bn : -1, &c2_sums));
      } else {
        LCHK(run_dgrad(c, dl.c1, c.F(p.g[dl.c1]), h, w, h, w, c.F(p.dcat[i]), nullptr, nullptr, nullptr, nullptr));
        LCHK(launch_upsplit(c.F(p.dcat[i]), N, h, w, dl.C0, dl.C1, gprev, pm, prev.scale, prev.shift,
                            dl.C1 > 0 ? c.F(p.gskip[i]) : nullptr, st));
      }
      h /= 2; w /= 2;
    }
    // ---------------- UnetPlusPlus nodes, last to first.  A tensor feeds several nodes: the first contribution to its
    // gradient buffer (in this order) writes, the others accumulate; ReLU masks are 0/1 factors, so masking each
    // contribution separately equals masking the sum.
    if (!m->nodes.empty()) {
      std::vector<char> ginit(5 + m->nodes.size(), 0);
      ginit[5 + m->nodes.size() - 1] = 1;                    // the head dgrad wrote the last node's gradient
      auto tensor_src = [&](int id) -> Src {
        if (id == 0) return f1;
        if (id < 5) return feat_src(id - 1);
        const NodeL& nd = m->nodes[id - 5];
        return lazy_src(c, nd.c2, H >> nd.lvl, W >> nd.lvl);
      };
      auto tensor_grad = [&](int id) -> float* {             // f1..f4 -> gskip[3..0], f5 -> gx of the last encoder block
        if (id < 4) return c.F(p.gskip[3 - id]);
        if (id == 4) return c.F(p.gx[feat_blk(3)]);
        return c.F(p.g[m->nodes[id - 5].c2]);
      };
      for (int i = (int)m->nodes.size() - 1; i >= 0; --i) {
        const NodeL& nd = m->nodes[i];
        const int nh = H >> nd.lvl, nw = W >> nd.lvl;
        const size_t npix = (size_t)N * nh * nw;
        if (!ginit[5 + i]) return fail("internal: UnetPlusPlus node %d has no consumer", i);
        LCHK(run_bn_bwd(c, nd.c2, c.F(p.g[nd.c2]), c.F(p.g[nd.c2]), npix, i == (int)m->nodes.size() - 1 && head_sums));
        Src a1 = lazy_src(c, nd.c1, nh, nw);
        LCHK(run_wgrad(c, nd.c2, a1, nullptr, c.F(p.g[nd.c2]), nh, nw));
        bool c1_sums = false;
        LCHK(run_dgrad(c, nd.c2, c.F(p.g[nd.c2]), nh, nw, nh, nw, c.F(p.g[nd.c1]), nullptr, a1.ptr, a1.scale, a1.shift, nullptr,
                       m->convs[nd.c1].bn, &c1_sums));
        LCHK(run_bn_bwd(c, nd.c1, c.F(p.g[nd.c1]), c.F(p.g[nd.c1]), npix, c1_sums));
        Src prev = tensor_src(nd.prev); prev.up = 1;
        Src skip; const Src* sp = nullptr;
        if (nd.skips.size() == 1) { skip = tensor_src(nd.skips[0]); sp = &skip; }
        else if (nd.skips.size() > 1) { skip = mk_src(c.F(p.cat[i]), nd.C1, nh, nw); sp = &skip; }
        LCHK(run_wgrad(c, nd.c1, prev, sp, c.F(p.g[nd.c1]), nh, nw));
        float* gprev = tensor_grad(nd.prev);
        const float* pm = nd.prev < 5 ? feat_mask(prev) : prev.ptr;
        const int acc_prev = ginit[nd.prev]; ginit[nd.prev] = 1;
        const ConvL& c1v = m->convs[nd.c1];
        const float* gcat; int gcc, gco;                      // where the skip part of the gradient lands
        if (c1v.wud_off && p.wino_ok((size_t)nd.c1) && !(nh & 1) && !(nw & 1)) {
          UpSplit us{gprev, nd.C0, pm, prev.scale, prev.shift, acc_prev};
          LCHK(run_dgrad(c, nd.c1, c.F(p.g[nd.c1]), nh, nw, nh, nw, nd.C1 > 0 ? c.F(p.gcat) : nullptr, nullptr, nullptr, nullptr,
                         nullptr, &us));
          gcat = c.F(p.gcat); gcc = nd.C1; gco = 0;
        } else {
          LCHK(run_dgrad(c, nd.c1, c.F(p.g[nd.c1]), nh, nw, nh, nw, c.F(p.dcat[0]), nullptr, nullptr, nullptr, nullptr));
          LCHK(launch_upsplit(c.F(p.dcat[0]), N, nh, nw, nd.C0, nd.C1, gprev, pm, prev.scale, prev.shift, nullptr, st, acc_prev));
          gcat = c.F(p.dcat[0]); gcc = nd.C0 + nd.C1; gco = nd.C0;
        }
        int coff = 0;
        for (int id : nd.skips) {
          const Src t = tensor_src(id);
          const bool lazy_mask = id >= 5;                     // node outputs: mask by their own ReLU; encoder features: the encoder masks
          LCHK(launch_split_accum(gcat, gcc, gco + coff, t.C, npix, tensor_grad(id), lazy_mask ? t.ptr : nullptr,
                                  lazy_mask ? t.scale : nullptr, lazy_mask ? t.shift : nullptr, ginit[id], st));
          ginit[id] = 1;
          coff += t.C;
        }
      }
      for (int id = 0; id < 5; ++id) if (!ginit[id]) return fail("internal: UnetPlusPlus feature %d received no gradient", id);
    }
  }
  // ---------------- EfficientNet encoder: backward stage k handles the blocks between features f_{5-k} and f_{6-k}
  for (int k = (sb < 1 ? 1 : sb); effnet && k < se && k <= 4; ++k) {
    const int s = 4 - k;
    const int lo = s == 0 ? 0 : m->feat_blk[s - 1] + 1;
    for (int bi = m->feat_blk[s]; bi >= lo; --bi) {
      const MBL& b = m->mb[bi];
      const int ho = p.mh[bi], wo = p.mw[bi], hi = ho * b.stride, wi = wo * b.stride;
      const size_t npo = (size_t)N * ho * wo, npi = (size_t)N * hi * wi;
      const float* dz = c.F(p.gx[bi]);                       // grad wrt the block output (complete)
      float* gO = c.F(p.g[b.cp]); float* gM = c.F(p.g[b.cdw]);
      const float* rs = (b.skip && b.drop > 0.f && m->keep_fwd) ? m->keep_fwd + (size_t)bi * N : nullptr;
      const float* g2 = dz;
      if (rs) { LCHK(launch_rowscale(dz, rs, N, (size_t)ho * wo, b.Cout, gO, st)); g2 = gO; }
      LCHK(run_bn_bwd(c, b.cp, g2, gO, npo));
      Src a2 = mk_src(c.F(p.a2[bi]), b.mid, ho, wo);
      LCHK(run_wgrad(c, b.cp, a2, nullptr, gO, ho, wo));
      LCHK(run_dgrad(c, b.cp, gO, ho, wo, ho, wo, gM, nullptr, nullptr, nullptr, nullptr));
      // squeeze-and-excitation: gs = sum_hw g*a1 ; FC backward ; g_a1 = g*s + gpool/hw, then through swish(bn1(.))
      float* pool = c.F(p.se_pool[bi]); float* sv = c.F(p.se[bi]); float* hpre = sv + (size_t)N * b.mid;
      float* gs = c.F(p.se_gs[bi]); float* acc1 = gs + (size_t)N * b.mid; float* gpool = c.F(p.se_g);      // gs / acc1 zeroed at the start of the backward
      LCHK(launch_se_reduce_hw(gM, c.F(p.a1[bi]), N, (size_t)ho * wo, b.mid, 1.f, gs, c.F(p.se_part), st));
      const ConvL& cr = m->convs[b.cr]; const ConvL& cx = m->convs[b.cx];
      LCHK(launch_se_fc_bwd(gs, sv, hpre, pool, m->params + cr.w_off, cr.Kpad, m->params + cx.w_off, cx.Kpad, N, b.mid, b.nsq,
                            gpool, acc1, m->grads + cr.w_off, m->grads + cr.bias_off, m->grads + cx.w_off, m->grads + cx.bias_off, st));
      LCHK(run_bn_bwd_act(c, b.cdw, gM, gM, N, (size_t)ho * wo, sv, gpool));
      // block input and where its gradient goes
      Src in = bi == 0 ? f1 : mk_src(c.F(p.xn[bi - 1]), b.Cin, hi, wi);
      float* gin = bi == 0 ? c.F(p.g[m->stem]) : c.F(p.gx[bi - 1]);
      const float* addend = nullptr;
      if (b.skip) addend = dz;                                // identity shortcut
      else if (bi == 0) addend = c.F(p.gskip[3]);             // f1's gradient from the decoder
      else for (int fs = 0; fs < 3; ++fs) if (bi - 1 == m->feat_blk[fs]) addend = c.F(p.gskip[2 - fs]);   // f2..f4
      const ConvL& dw = m->convs[b.cdw];
      const float* dwin = b.ce >= 0 ? c.F(p.a0[bi]) : in.ptr;
      {
        hipStream_t ws_ = st;
        if (c.wst && c.wst != st) { HIPCHK(hipEventRecord(m->ev_fork, st)); HIPCHK(hipStreamWaitEvent(c.wst, m->ev_fork, 0)); ws_ = c.wst; }
        LCHK(launch_dw_wgrad(dwin, gM, b.k, b.stride, b.pb, N, hi, wi, b.mid, ho, wo, m->grads + dw.w_off, c.F(p.dw_part), ws_));
      }
      if (b.ce >= 0) {
        float* gI = c.F(p.g[b.ce]);
        LCHK(launch_dw_dgrad(gM, m->params + dw.w_off, b.k, b.stride, b.pb, N, hi, wi, b.mid, ho, wo, nullptr, gI, st));
        LCHK(run_bn_bwd_act(c, b.ce, gI, gI, N, (size_t)hi * wi, nullptr, nullptr));
        LCHK(run_wgrad(c, b.ce, in, nullptr, gI, hi, wi));
        LCHK(run_dgrad(c, b.ce, gI, hi, wi, hi, wi, gin, addend, nullptr, nullptr, nullptr));
      } else {
        LCHK(launch_dw_dgrad(gM, m->params + dw.w_off, b.k, b.stride, b.pb, N, hi, wi, b.mid, ho, wo, addend, gin, st));
      }
    }
    if (s == 0) {
      float* g = c.F(p.g[m->stem]);
      LCHK(run_bn_bwd_act(c, m->stem, g, g, N, (size_t)h1 * w1, nullptr, nullptr));
      Src x4 = mk_src(c.F(p.x4), m->CinP, H, W);
      LCHK(run_wgrad(c, m->stem, x4, nullptr, g, h1, w1));
    }
  }
  // ---------------- encoder stages: backward stage k handles encoder stage s = 4 - k
  for (int k = (sb < 1 ? 1 : sb); !effnet && k < se && k <= 4; ++k) {
    const int s = 4 - k;
    const int h = sh[s], w = sw[s];
    const size_t npix = (size_t)N * h * w;
    for (int b = (int)m->stages[s].size() - 1; b >= 0; --b) {
      const BlockL& bl = m->stages[s][b];
      const size_t bi = first_blk[s] + b;
      const float* dz = c.F(p.gx[bi]);                       // masked grad wrt the block output
      const int lc = bl.last();
      const int hin = h * bl.stride, win = w * bl.stride;
      LCHK(run_bn_bwd(c, lc, dz, c.F(p.g[lc]), npix, out_sums[bi] != 0));
      if (bl.cd >= 0) LCHK(run_bn_bwd(c, bl.cd, dz, c.F(p.g[bl.cd]), npix));
      int hc1 = h, wc1 = w;                                  // resolution of c1's output
      if (bl.c3 >= 0) {                                      // Bottleneck tail: conv3 (1x1) <- relu(bn2(conv2))
        Src a2 = lazy_src(c, bl.c2, h, w);
        LCHK(run_wgrad(c, bl.c3, a2, nullptr, c.F(p.g[bl.c3]), h, w));
        bool c2_sums = false;                                // conv3's 1x1 dgrad (implicit GEMM epilogue) carries bn2's backward sums
        LCHK(run_dgrad(c, bl.c3, c.F(p.g[bl.c3]), h, w, h, w, c.F(p.g[bl.c2]), nullptr, a2.ptr, a2.scale, a2.shift, nullptr,
                       m->convs[bl.c2].bn, &c2_sums));
        LCHK(run_bn_bwd(c, bl.c2, c.F(p.g[bl.c2]), c.F(p.g[bl.c2]), npix, c2_sums));
        hc1 = hin; wc1 = win;
      }
      Src a1 = lazy_src(c, bl.c1, hc1, wc1);
      LCHK(run_wgrad(c, bl.c2, a1, nullptr, c.F(p.g[bl.c2]), h, w));
      bool c1_sums = false;
      LCHK(run_dgrad(c, bl.c2, c.F(p.g[bl.c2]), h, w, hc1, wc1, c.F(p.g[bl.c1]), nullptr, a1.ptr, a1.scale, a1.shift, nullptr,
                     m->convs[bl.c1].bn, &c1_sums));
      LCHK(run_bn_bwd(c, bl.c1, c.F(p.g[bl.c1]), c.F(p.g[bl.c1]), (size_t)N * hc1 * wc1, c1_sums));
      // block input
      Src in; float* gin; const float* in_mask;
      if (b > 0) { in = mk_src(c.F(p.xn[bi - 1]), bl.Cin, hin, win); gin = c.F(p.gx[bi - 1]); in_mask = in.ptr; }
      else if (s > 0) { in = feat_src(s - 1); gin = c.F(p.gx[first_blk[s] - 1]); in_mask = in.ptr; }
      else { in = mk_src(c.F(p.pool), 64, hin, win); gin = c.F(p.g_pool); in_mask = nullptr; }
      LCHK(run_wgrad(c, bl.c1, in, nullptr, c.F(p.g[bl.c1]), hc1, wc1));
      const float* addend;
      if (bl.cd >= 0) {
        LCHK(run_wgrad(c, bl.cd, in, nullptr, c.F(p.g[bl.cd]), h, w));
        // skip-connection gradient from the decoder for this feature (f2,f3,f4 <- dec blocks 2,1,0)
        const float* gs = (b == 0 && s > 0) ? c.F(p.gskip[3 - s]) : nullptr;
        LCHK(run_dgrad(c, bl.cd, c.F(p.g[bl.cd]), h, w, hin, win, c.F(p.tmp), gs, nullptr, nullptr, nullptr));
        addend = c.F(p.tmp);
      } else {
        addend = dz;                                         // identity shortcut
      }
      // gin is the (masked) gradient wrt the PREVIOUS block's output = the gradient wrt that block's last BatchNorm output: its
      // backward sums ride in this epilogue, yhat read from that BatchNorm's raw input (run_dgrad bn_y)
      int pbn = -1; const float* py = nullptr; size_t pbi = 0;
      if (b > 0) { pbi = bi - 1; const int plc = m->stages[s][b - 1].last(); pbn = m->convs[plc].bn; py = c.F(p.y[plc]); }
      else if (s > 0) { pbi = first_blk[s] - 1; const int plc = m->stages[s - 1].back().last(); pbn = m->convs[plc].bn; py = c.F(p.y[plc]); }
      bool psums = false;
      LCHK(run_dgrad(c, bl.c1, c.F(p.g[bl.c1]), hc1, wc1, hin, win, gin, addend, in_mask, nullptr, nullptr, nullptr, pbn, &psums, py));
      if (pbn >= 0) out_sums[pbi] = psums ? 1 : 0;
    }
    if (s == 0) {
      // maxpool backward (+ decoder skip gradient for f1) -> stem BN backward -> stem wgrad
      // the masked gradient it writes is the gradient wrt the stem BatchNorm's output and it reads that BatchNorm's raw input
      // for the mask anyway: the BatchNorm-backward sums ride along (replicas of the forward statistics, re-zeroed above)
      const BNL& sb = m->bns[m->convs[m->stem].bn];
      static const bool no_fuse = dbg_flag("UWM_NO_BN_FUSE");
      const bool pool_sums = !no_fuse && f1.scale && f1.relu && sb.C == f1.C && (256 % (f1.C / 4)) == 0;
      LCHK(launch_maxpool_bwd(c.F(p.g_pool), (const uint8_t*)c.F(p.pool_idx), c.F(p.gskip[3]), f1, c.F(p.g[m->stem]), N,
                              sh[0], sw[0], st, pool_sums ? c.F(sb.f_off) : nullptr, pool_sums ? c.F(sb.f_off) + sb.C : nullptr,
                              c.D(sb.d_off) + 2 * sb.C, c.D(sb.d_off) + 3 * sb.C, sb.nrep, 2 * sb.C));
      LCHK(run_bn_bwd(c, m->stem, c.F(p.g[m->stem]), c.F(p.g[m->stem]), (size_t)N * h1 * w1, pool_sums));
      Src x4 = mk_src(c.F(p.x4), m->CinP, H, W);
      LCHK(run_wgrad(c, m->stem, x4, nullptr, c.F(p.g[m->stem]), h1, w1));
    }
  }
  LCHK(launch_wgrad_reduce_multi(m->rq, c.wst));      // the queued partial-sum reduces of these stages: one launch
  if (c.wst != st) {                    // join: the caller's stream waits for every wgrad of these stages
    HIPCHK(hipEventRecord(m->ev_join, c.wst));
    // data-parallel training hands in its communication stream: the bucket's all-reduce waits for this stage's weight
    // gradients, the caller's stream runs on into the next stage's dgrad chain (nothing there reads what the side
    // stream still writes).  The LAST stage always joins the caller's stream: the optimizer comes next.
    HIPCHK(hipStreamWaitEvent((m->join_stream && se < 5) ? m->join_stream : st, m->ev_join, 0));
    if (m->join_stream && se >= 5) HIPCHK(hipStreamWaitEvent(m->join_stream, m->ev_join, 0));
  }
  return 0;
}

// ------------------------------------------------------------------------------ C ABI
extern "C" {

const char* uwm_last_error(void) { return g_err; }
int uwm_version(void) { return 1; }

int uwm_create(const uwm_unet_desc* desc, uwm_handle* out) {
  if (!desc || !out) return fail("uwm_create: null argument");
  uwm_model* m = new uwm_model();
  m->desc = *desc;
  if (m->desc.bn_eps <= 0.f) m->desc.bn_eps = 1e-5f;
  if (m->desc.bn_momentum <= 0.f) m->desc.bn_momentum = 0.1f;
  if (build_model(m)) { delete m; return 1; }
  const char* e = getenv("UWM_SIDE_STREAM");
  m->use_side = e ? atoi(e) : 1;
  if (const char* pe = getenv("UWM_PRECISION")) {      // process default of the precision mode (like UWM_WINOGRAD): f32 | bf16x3 | bf16x3_all | f16x3 | f16x3_all or 0..4
    static const char* names[7] = {"f32", "bf16x3", "bf16x3_all", "f16x3", "f16x3_all", "f16x1", "f16x3_bwd2"};
    int found = -1;
    for (int i = 0; i < 7; ++i) if (!strcmp(pe, names[i]) || (pe[0] == '0' + i && !pe[1])) found = i;
    if (found < 0) { delete m; return fail("uwm_create: UWM_PRECISION=%s is not a precision mode (f32 | bf16x3 | bf16x3_all | f16x3 | f16x3_all | f16x1 | f16x3_bwd2 or 0..6)", pe); }
    m->prec = found; m->plan.prec = found; m->prec_from_env = found != UWM_PREC_F32;
  }
  if (const char* pf = getenv("UWM_F16X3_MIN_WGS")) m->f3_min_wgs = atoi(pf) > 0 ? atoi(pf) : 0;      // process default of uwm_set_precision_fill
  m->wino_mode = winograd_mode();       // process default (UWM_WINOGRAD / uwm_set_winograd) at creation; then per handle
  m->plan.wino_mode = m->wino_mode;
  *out = m; return 0;
}
void uwm_destroy(uwm_handle h) {
  if (!h) return;
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_pack) (void)hipEventDestroy(h->ev_pack);
  if (h->ev_disp) (void)hipEventDestroy(h->ev_disp);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

long long uwm_param_arena_floats(uwm_handle h) { return h ? h->param_floats : 0; }
long long uwm_buffer_arena_floats(uwm_handle h) { return h ? h->buffer_floats : 0; }
long long uwm_param_count(uwm_handle h) { return h ? h->param_count : 0; }
int uwm_num_tensors(uwm_handle h) { return h ? (int)h->infos.size() : 0; }
int uwm_tensor_info_get(uwm_handle h, int i, uwm_tensor_info* out) {
  if (!h || !out || i < 0 || i >= (int)h->infos.size()) return fail("uwm_tensor_info_get: bad index %d", i);
  *out = h->infos[i]; return 0;
}
int uwm_logits_channels(uwm_handle h) { return h ? h->CP : 0; }
int uwm_num_stages(uwm_handle h) { return h ? h->nstages : 0; }
int uwm_stage_range(uwm_handle h, int stage, long long* b, long long* e) {
  if (!h || stage < 0 || stage >= h->nstages || !b || !e) return fail("uwm_stage_range: bad stage %d", stage);
  *b = h->stage_begin[stage]; *e = h->stage_begin[stage + 1]; return 0;
}
// RAII: make `dev` the current HIP device for the scope of one ABI call (a handle is bound to the device its arenas
// live on; the caller's current device may be another one)
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); return; }
    if (prev != dev && hipSetDevice(dev) == hipSuccess) switched = true;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

int uwm_bind(uwm_handle h, float* params, float* grads, float* buffers) {
  if (!h || !params || !buffers) return fail("uwm_bind: params and buffers must be non-null");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)buffers) & 15) return fail("uwm_bind: arenas must be 16-byte aligned");
  hipPointerAttribute_t at;
  int dev = -1;
  if (hipPointerGetAttributes(&at, params) == hipSuccess) dev = at.device; else (void)hipGetLastError();
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
  if (h->device >= 0 && h->device != dev && h->side) {      // arenas moved to another device: the side stream moves with them
    DeviceGuard g0(h->device);
    (void)hipStreamSynchronize(h->side);
    (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); (void)hipEventDestroy(h->ev_pack);
    if (h->ev_disp) (void)hipEventDestroy(h->ev_disp);
    (void)hipStreamDestroy(h->side);
    h->side = nullptr; h->ev_fork = h->ev_join = h->ev_pack = h->ev_disp = nullptr; h->packed_in_fwd = false;
  }
  h->device = dev;
  h->params = params; h->grads = grads; h->buffers = buffers;
  DeviceGuard guard(dev);
  if (h->use_side && !h->side) {        // created on the device the arenas live on
    // LOWEST queue priority: the weight gradients are off the critical path, the dgrad / BatchNorm-backward chain on the caller's
    // stream is it — when a CU slot frees up, the dependent chain's workgroups must get it first
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { prio_lo = 0; (void)hipGetLastError(); }
    if (dbg_flag("UWM_SIDE_PRIO_NORMAL")) prio_lo = 0;
    if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_lo) != hipSuccess) { h->side = nullptr; (void)hipGetLastError(); }
    else if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(h->side); h->side = nullptr; (void)hipGetLastError();
    }
    if (h->side && hipEventCreate(&h->ev_disp) != hipSuccess) { h->ev_disp = nullptr; (void)hipGetLastError(); }
    h->disp_fork = dbg_flag("UWM_NO_DISPATCH_FORK") ? 0 : 1;
    // ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  In a process that
    // has already created several streams (RCCL, torch) the side stream can alias the compute stream's queue: the
    // weight-gradient overlap is then silently lost (measured -5 %).  The variable is read when the HIP
    // runtime initialises, so all the library can do at this point is say so.
    if (h->side && !h->hwq_warned) {
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      if (!q || atoi(q) < 8) {
        fprintf(stderr, "libuwm: warning: GPU_MAX_HW_QUEUES=%s (< 8): the weight-gradient side stream may share a hardware queue with "
                        "the compute stream and lose its overlap; export GPU_MAX_HW_QUEUES=8 before the HIP runtime starts\n", q ? q : "unset");
        h->hwq_warned = true;
      }
    }
  }
  return 0;
}

static int check_shape(int N, int H, int W) {
  if (N < 1) return fail("batch size must be >= 1, got %d", N);
  if (H < 32 || W < 32 || (H % 32) || (W % 32))
    return fail("Wrong input shape height=%d, width=%d. Expected image height and width divisible by 32.", H, W);
  if ((long long)N * H * W >= (1LL << 31) / 16) return fail("N*H*W too large for 32-bit pixel indexing: %lld", (long long)N * H * W);
  return 0;
}

size_t uwm_workspace_bytes(uwm_handle h, int N, int H, int W, int training) {
  if (!h || check_shape(N, H, W)) return 0;
  if (h->plan.N != N || h->plan.H != H || h->plan.W != W || h->plan.training != (training ? 1 : 0)) {
    make_plan(h, N, H, W, training ? 1 : 0);
    h->have_fwd = false;
  }
  return h->plan.bytes;
}

// Algorithmic (direct-convolution) FLOPs per IMAGE of this model at H x W — SURVEY.md 8(d)'s roofline numerator:
// forward = sum over conv layers of 2*Ho*Wo*Cout*Cin_per_group*k*k ; forward+backward = 3x that minus the stem's dgrad
// (the input image needs no gradient).  BatchNorm / elementwise / loss / optimizer work is not counted.
int uwm_conv_flops(uwm_handle h, int H, int W, double* fwd, double* fwd_bwd) {
  if (!h || !fwd || !fwd_bwd) return fail("uwm_conv_flops: null argument");
  if (check_shape(1, H, W)) return 1;
  const Plan keep = h->plan;
  make_plan(h, 1, H, W, 0);
  double f = 0.0, fb = 0.0;
  for (size_t ci = 0; ci < h->convs.size(); ++ci) {
    const ConvL& cv = h->convs[ci];
    const double macs = (double)h->plan.oh[ci] * h->plan.ow[ci] * cv.Cout * (cv.dw ? 1 : cv.Cin) * cv.k * cv.k;
    f += 2.0 * macs; fb += 2.0 * macs * ((int)ci == h->stem ? 2.0 : 3.0);
  }
  h->plan = keep;
  *fwd = f; *fwd_bwd = fb; return 0;
}

int uwm_forward(uwm_handle h, const float* x, float* logits, void* ws, size_t ws_bytes, int N, int H, int W, int training,
                uwm_stream stream) {
  if (!h || !x || !logits || !ws) return fail("uwm_forward: null argument");
  if (!h->params || !h->buffers) return fail("uwm_forward: call uwm_bind first");
  if (check_shape(N, H, W)) return 1;
  if (((uintptr_t)x | (uintptr_t)logits | (uintptr_t)ws) & 15) return fail("uwm_forward: pointers must be 16-byte aligned");
  const size_t need = uwm_workspace_bytes(h, N, H, W, training);
  if (ws_bytes < need) return fail("uwm_forward: workspace too small (%zu < %zu bytes)", ws_bytes, need);
  h->have_fwd = false;
  DeviceGuard guard(h->device);
  if (do_forward(h, x, logits, (float*)ws, N, H, W, training ? 1 : 0, (hipStream_t)stream)) return 1;
  h->have_fwd = training != 0;
  return 0;
}

int uwm_backward(uwm_handle h, const float* dlogits, void* ws, int sb, int se, uwm_stream stream) {
  if (!h || !dlogits || !ws) return fail("uwm_backward: null argument");
  if (!h->grads) return fail("uwm_backward: no gradient arena bound");
  if (!h->have_fwd) return fail("uwm_backward: no training-mode forward is held in the workspace");
  if (sb < 0 || se > h->nstages || sb >= se) return fail("uwm_backward: bad stage range [%d,%d)", sb, se);
  DeviceGuard guard(h->device);
  return do_backward(h, dlogits, (float*)ws, sb, se, (hipStream_t)stream);
}

int uwm_loss(const float* logits, int ld, const void* target, int tdt, long long npix, float w_dice, float w_bce,
             float smooth, float eps, void* scratch, float* loss_out, float* dlogits, int ldd, float grad_scale,
             uwm_stream stream) {
  if (!logits || !target || !scratch || !loss_out || npix <= 0 || ld < 1) return fail("uwm_loss: bad argument");
  if (tdt < 0 || tdt > 3) return fail("uwm_loss: unsupported target dtype %d", tdt);
  if (dlogits && ldd < 1) return fail("uwm_loss: bad dlogits stride");
  LCHK(launch_loss(logits, ld, target, tdt, (size_t)npix, w_dice, w_bce, smooth, eps, (double*)scratch, loss_out, dlogits,
                   ldd, grad_scale, (hipStream_t)stream));
  return 0;
}
int uwm_loss_sums(const float* logits, int ld, const void* target, int tdt, long long npix, void* scratch, uwm_stream stream) {
  if (!logits || !target || !scratch || npix <= 0 || ld < 1) return fail("uwm_loss_sums: bad argument");
  if (tdt < 0 || tdt > 3) return fail("uwm_loss_sums: unsupported target dtype %d", tdt);
  LCHK(launch_loss_sums(logits, ld, target, tdt, (size_t)npix, (double*)scratch, (hipStream_t)stream));
  return 0;
}
int uwm_loss_apply(const float* logits, int ld, const void* target, int tdt, long long npix, long long npix_total, float w_dice,
                   float w_bce, float smooth, float eps, const void* scratch, float* loss_out, float* dlogits, int ldd,
                   float grad_scale, uwm_stream stream) {
  if (!logits || !target || !scratch || !loss_out || npix <= 0 || npix_total < npix || ld < 1) return fail("uwm_loss_apply: bad argument");
  if (tdt < 0 || tdt > 3) return fail("uwm_loss_apply: unsupported target dtype %d", tdt);
  if (dlogits && ldd < 1) return fail("uwm_loss_apply: bad dlogits stride");
  LCHK(launch_loss_apply(logits, ld, target, tdt, (size_t)npix, (double)npix_total, w_dice, w_bce, smooth, eps, (const double*)scratch,
                         loss_out, dlogits, ldd, grad_scale, (hipStream_t)stream));
  return 0;
}
int uwm_stats(const float* logits, int ld, const void* target, int tdt, int N, long long hw, float thr, int sig,
              long long* out, uwm_stream stream) {
  if (!logits || !target || !out || N < 1 || hw < 1) return fail("uwm_stats: bad argument");
  if (tdt < 0 || tdt > 3) return fail("uwm_stats: unsupported target dtype %d", tdt);
  LCHK(launch_stats(logits, ld, target, tdt, N, (size_t)hw, thr, sig, out, (hipStream_t)stream));
  return 0;
}
int uwm_threshold(const float* logits, int ld, long long npix, float thr, int sig, uint8_t* mask, uwm_stream stream) {
  if (!logits || !mask || npix < 1) return fail("uwm_threshold: bad argument");
  LCHK(launch_threshold(logits, ld, (size_t)npix, thr, sig, mask, (hipStream_t)stream));
  return 0;
}
int uwm_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
             long long step, float gscale, uwm_stream stream) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return fail("uwm_adam: bad argument");
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  LCHK(launch_adam(p, g, m, v, (size_t)n, lr, b1, b2, eps, wd, bc1, bc2, gscale, (hipStream_t)stream));
  return 0;
}
int uwm_adam_clip(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
                  long long step, float gscale, float max_norm, void* scratch, uwm_stream stream) {
  if (!p || !g || !m || !v || !scratch || n < 1 || step < 1 || max_norm <= 0.f) return fail("uwm_adam_clip: bad argument");
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  LCHK(launch_sumsq(g, (size_t)n, (double*)scratch, (hipStream_t)stream));
  LCHK(launch_adam(p, g, m, v, (size_t)n, lr, b1, b2, eps, wd, bc1, bc2, gscale, (hipStream_t)stream, (const double*)scratch, max_norm));
  return 0;
}
int uwm_adam_graph(float* p, const float* g, float* m, float* v, long long n, float* hyper, void* clip_scratch, uwm_stream stream) {
  if (!p || !g || !m || !v || !hyper || n < 1) return fail("uwm_adam_graph: bad argument");
  if (clip_scratch) LCHK(launch_sumsq(g, (size_t)n, (double*)clip_scratch, (hipStream_t)stream));
  LCHK(launch_adam_graph(p, g, m, v, (size_t)n, hyper, (const double*)clip_scratch, (hipStream_t)stream));
  return 0;
}
int uwm_sgd(float* p, const float* g, float* buf, long long n, float lr, float momentum, float wd, long long step, float gscale,
            float max_norm, void* scratch, uwm_stream stream) {
  if (!p || !g || !buf || n < 1 || step < 1 || (max_norm > 0.f && !scratch)) return fail("uwm_sgd: bad argument");
  if (max_norm > 0.f) LCHK(launch_sumsq(g, (size_t)n, (double*)scratch, (hipStream_t)stream));
  LCHK(launch_sgd(p, g, buf, (size_t)n, lr, momentum, wd, step == 1, gscale, (hipStream_t)stream,
                  max_norm > 0.f ? (const double*)scratch : nullptr, max_norm));
  return 0;
}
int uwm_resize_threshold(const float* logits, int ld, int N, int h, int w, int H, int W, float threshold, int apply_sigmoid,
                         uint8_t* mask, float* resized, uwm_stream stream) {
  if (!logits || (!mask && !resized) || N < 1 || h < 1 || w < 1 || H < 1 || W < 1 || ld < 1) return fail("uwm_resize_threshold: bad argument");
  LCHK(launch_resize_threshold(logits, ld, N, h, w, H, W, threshold, apply_sigmoid, mask, resized, (hipStream_t)stream));
  return 0;
}
int uwm_preprocess_u8(const uint8_t* images, int N, int H, int W, int C, const float* mean, const float* std, const int* flags,
                      float* out_nchw, uwm_stream stream) {
  if (!images || !mean || !std || !out_nchw || N < 1 || H < 1 || W < 1 || C < 1 || C > 4) return fail("uwm_preprocess_u8: bad argument");
  for (int c = 0; c < C; ++c) if (!(std[c] > 0.f)) return fail("uwm_preprocess_u8: std[%d] must be positive", c);
  LCHK(launch_preprocess_u8(images, N, H, W, C, mean, std, flags, out_nchw, (hipStream_t)stream));
  return 0;
}
int uwm_preprocess_mask_u8(const uint8_t* masks, int N, int H, int W, int threshold, const int* flags, uint8_t* out,
                           uwm_stream stream) {
  if (!masks || !out || N < 1 || H < 1 || W < 1) return fail("uwm_preprocess_mask_u8: bad argument");
  LCHK(launch_preprocess_mask(masks, N, H, W, threshold, flags, out, (hipStream_t)stream));
  return 0;
}
int uwm_scale(float* p, long long n, float s, uwm_stream stream) {
  if (!p || n < 1) return fail("uwm_scale: bad argument");
  LCHK(launch_scale(p, (size_t)n, s, (hipStream_t)stream));
  return 0;
}

// ---- run-time switch for the internal weight-gradient side stream (default on; UWM_SIDE_STREAM=0 disables it)
int uwm_set_side_stream(uwm_handle h, int on) {
  if (!h) return fail("uwm_set_side_stream: null handle");
  h->use_side = on != 0; return 0;
}

// ---- HIP-event profiler (bench.py's roofline leg)
int uwm_prof_enable(int on) { prof_enable(on != 0); return 0; }
int uwm_prof_collect(double* out, int max_classes) {
  if (!out || max_classes < kProfClasses) return fail("uwm_prof_collect: need room for %d classes (4 doubles each)", (int)kProfClasses);
  prof_collect(out); return kProfClasses;
}
const char* uwm_prof_class_name(int cls) { return prof_class_name(cls); }

// ---- workspace introspection (parity tests): float offset + element count of a planned buffer.
// keys: "y:<conv>", "g:<conv>" (<conv> = state_dict prefix, e.g. encoder.layer1.0.conv1),
//       "xn:<i>", "gx:<i>" (encoder block i), "pool", "g_pool", "x4", "dcat:<i>", "gskip:<i>"
int uwm_debug_lookup(uwm_handle h, const char* key, long long* off, long long* count) {
  if (!h || !key || !off || !count) return fail("uwm_debug_lookup: null argument");
  const Plan& p = h->plan;
  if (p.N == 0) return fail("uwm_debug_lookup: no plan yet");
  const std::string k(key);
  const int N = p.N, H = p.H, W = p.W;
  auto conv_geo = [&](int ci, long long* cnt) { *cnt = (long long)N * p.oh[ci] * p.ow[ci] * h->convs[ci].CoutP; };
  if (k.rfind("y:", 0) == 0 || k.rfind("g:", 0) == 0) {
    const std::string name = k.substr(2);
    for (size_t i = 0; i < h->convs.size(); ++i) if (h->convs[i].name == name) {
      if ((int)i == h->head) return fail("uwm_debug_lookup: head output is the caller's logits buffer");
      *off = (long long)(k[0] == 'y' ? p.y[i] : p.g[i]); conv_geo((int)i, count); return 0;
    }
    return fail("uwm_debug_lookup: unknown conv %s", name.c_str());
  }
  auto blk_geo = [&](size_t bi, long long* cnt) {
    if (!h->mb.empty()) { *cnt = bi < h->mb.size() ? (long long)N * p.mh[bi] * p.mw[bi] * h->mb[bi].Cout : 0; return; }
    int hh = H / 4, ww = W / 4; size_t b = 0;
    for (int s = 0; s < 4; ++s) for (auto& bl : h->stages[s]) {
      hh /= bl.stride; ww /= bl.stride;
      if (b == bi) { *cnt = (long long)N * hh * ww * bl.Cout; return; }
      ++b;
    }
    *cnt = 0;
  };
  if (k.rfind("xn:", 0) == 0 || k.rfind("gx:", 0) == 0) {
    const size_t bi = (size_t)atoi(k.c_str() + 3);
    if (bi >= p.xn.size()) return fail("uwm_debug_lookup: bad block index");
    *off = (long long)(k[0] == 'x' ? p.xn[bi] : p.gx[bi]); blk_geo(bi, count); return 0;
  }
  if (k == "stem_a") { *off = (long long)p.stem_a; *count = (long long)N * (H / 2) * (W / 2) * h->f1C; return 0; }
  if (k == "pool" || k == "g_pool") { *off = (long long)(k == "pool" ? p.pool : p.g_pool); *count = (long long)N * (H / 4) * (W / 4) * 64; return 0; }
  if (k == "x4") { *off = (long long)p.x4; *count = (long long)N * H * W * h->CinP; return 0; }
  if (k.rfind("dcat:", 0) == 0 || k.rfind("gskip:", 0) == 0) {
    const bool dc = k[0] == 'd';
    const size_t i = (size_t)atoi(k.c_str() + (dc ? 5 : 6));
    if (i >= h->dec.size() || p.dcat.empty()) return fail("uwm_debug_lookup: bad decoder index / eval plan");
    const int hh = (H / 32) << (i + 1), ww = (W / 32) << (i + 1);
    *off = (long long)(dc ? p.dcat[i] : p.gskip[i]);
    *count = (long long)N * hh * ww * (dc ? h->dec[i].C0 + h->dec[i].C1 : h->dec[i].C1);
    return 0;
  }
  return fail("uwm_debug_lookup: unknown key %s", key);
}

// ---- single-operator entry points
// Winograd weights for the op-level entry points (tests): transformed into a cached scratch buffer
static int op_wino_prepare(ConvArgs& a, int mirror, hipStream_t st, bool x3 = false) {
  static float* buf = nullptr; static size_t cap = 0;
  const size_t need = wino_weights_floats(a.wrows, a.Ctot);
  if (need > cap) {
    HIPCHK(hipDeviceSynchronize());
    if (buf) HIPCHK(hipFree(buf));
    HIPCHK(hipMalloc((void**)&buf, need * sizeof(float))); cap = need;
  }
  if (x3) {           // bf16x3 bank (forward layout only: the model derives dgrad banks from the forward weights itself)
    if (mirror || (a.Ctot & 15)) return fail("uwm_op_conv: cfg 400 (bf16x3 Winograd) takes forward weights with channels %% 16 == 0");
    WinoJobs jobs; jobs.n = 1;
    WinoJob& j = jobs.j[0];
    j.w = a.w; j.ut = buf; j.rows = a.wrows; j.chans = a.Ctot; j.Kpad = a.Kpad; j.mode = 0; j.src_rows = a.wrows; j.pad_ = 0;
    LCHK(launch_wino_weights_x3_multi(jobs, st));
    a.prec = 1;
  } else {
    LCHK(launch_wino_weights(a.w, a.wrows, a.Kpad, a.Ctot, mirror, buf, st));
  }
  a.wu = buf; a.wu_ncb = wino_ncb(a.wrows);
  return 0;
}
// fp16x3 bank for the op-level entry point (tests / timing): cfg 600
static int op_f16x3_prepare(ConvArgs& a, hipStream_t st, int variant = 0, bool reuse = false) {
  static float* buf = nullptr; static size_t cap = 0;
  static const float* last_w = nullptr; static int last_rows = 0, last_chans = 0, last_layout = -1;      // kernel timing (cfg + 1000): the bank of the previous call is reused when it matches
  if ((a.Ctot & 31) && !(a.Ctot == 16 && a.Cout <= 16)) return fail("uwm_op_conv: cfg 600 (fp16x3) needs channels %% 32 == 0 (or the 16 -> 16 single-chunk layer)");
  const size_t need = f16x3_bank_floats(a.wrows, a.Ctot);
  if (need > cap) {
    HIPCHK(hipDeviceSynchronize());
    if (buf) HIPCHK(hipFree(buf));
    HIPCHK(hipMalloc((void**)&buf, need * sizeof(float))); cap = need;
  }
  WinoJobs jobs; jobs.n = 1;
  WinoJob& j = jobs.j[0];
  // bank layout: 601-603 force a conv_f16x3.hip kernel (layout 0), 604 / 605 a conv_f16x3v2.hip one (layout 1), 600 = what the model would take
  const int layout = variant >= 4 ? 1 : (variant == 0 && f16x3v2_shape(a.Ho, a.Wo, a.wrows, a.Ctot, 0) && a.wrows == a.Cout ? 1 : 0);
  j.w = a.w; j.ut = buf; j.rows = a.wrows; j.chans = a.Ctot; j.Kpad = a.Kpad; j.mode = 0; j.src_rows = a.wrows; j.pad_ = layout;
  if (!(reuse && last_w == a.w && last_rows == a.wrows && last_chans == a.Ctot && last_layout == layout)) LCHK(launch_f16x3_weights_multi(jobs, st));
  last_w = a.w; last_rows = a.wrows; last_chans = a.Ctot; last_layout = layout;
  a.wu = buf; a.wu_layout = layout; a.wu_ncb = layout == 1 ? f16x3v2_nf(a.wrows) : f16x3_nj(a.wrows); a.wu_rinv_off = (int)f16x3_rinv_off(a.wrows, a.Ctot); a.prec = 2;
  return 0;
}
static bool op_wino_shape(const ConvArgs& a, int kh, int kw, int stride, int pad) {
  return kh == 3 && kw == 3 && stride == 1 && pad == 1 && (a.Ctot & 7) == 0 && (a.C0 & 7) == 0 && a.Ho >= 8 && a.Wo >= 16;
}
int uwm_set_join_stream(uwm_handle h, uwm_stream stream) {
  if (!h) return fail("uwm_set_join_stream: null handle");
  h->join_stream = (hipStream_t)stream; return 0;
}
int uwm_set_drop_connect(uwm_handle h, const float* rowscale) {
  if (!h) return fail("uwm_set_drop_connect: null handle");
  if (rowscale && h->mb.empty()) return fail("uwm_set_drop_connect: the encoder has no MBConv blocks");
  h->keep = rowscale; return 0;
}
int uwm_num_mbconv_blocks(uwm_handle h) { return h ? (int)h->mb.size() : 0; }
float uwm_mbconv_drop_rate(uwm_handle h, int block) {
  if (!h || block < 0 || block >= (int)h->mb.size()) return 0.f;
  return h->mb[block].skip ? h->mb[block].drop : 0.f;
}
int uwm_op_depthwise(int mode, const float* a, const float* b, int k, int stride, int pb, int N, int H, int W, int C, int Ho, int Wo,
                     const float* addend, float* out, float* scratch, uwm_stream stream) {
  if (!a || !b || !out || N < 1 || (C & 3) || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return fail("uwm_op_depthwise: bad argument");
  if (pb < 0 || pb >= k || Ho < 1 || Wo < 1 || (Ho - 1) * stride - pb >= H || (Wo - 1) * stride - pb >= W) return fail("uwm_op_depthwise: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) LCHK(launch_dw_fwd(a, b, k, stride, pb, N, H, W, C, Ho, Wo, out, st));
  else if (mode == 1) LCHK(launch_dw_dgrad(a, b, k, stride, pb, N, H, W, C, Ho, Wo, addend, out, st));
  else if (mode == 2) { if (!scratch) return fail("uwm_op_depthwise: wgrad needs scratch"); LCHK(launch_dw_wgrad(a, b, k, stride, pb, N, H, W, C, Ho, Wo, out, scratch, st)); }
  else return fail("uwm_op_depthwise: bad mode %d", mode);
  return 0;
}
long long uwm_op_depthwise_scratch_floats(int k, int N, int C, int Ho, int Wo) { return (long long)dw_wgrad_scratch_floats(k, N, C, Ho, Wo); }
int uwm_set_winograd(int on) { winograd_set_mode(on < 0 ? 0 : (on > 2 ? 1 : on)); return 0; }
int uwm_set_winograd_mode(uwm_handle h, int mode) {
  if (!h) return fail("uwm_set_winograd_mode: null handle");
  if (mode < 0 || mode > 2) return fail("uwm_set_winograd_mode: mode must be 0 (direct kernels), 1 (auto) or 2 (8-wave variant wherever allowed), got %d", mode);
  h->wino_mode = mode; h->plan.wino_mode = mode;       // the workspace layout does not depend on the mode
  return 0;
}
int uwm_get_winograd_mode(uwm_handle h) { return h ? h->wino_mode : -1; }
int uwm_set_precision(uwm_handle h, int mode) {
  if (!h) return fail("uwm_set_precision: null handle");
  if (mode < UWM_PREC_F32 || mode > UWM_PREC_F16X3_BWD2)
    return fail("uwm_set_precision: mode must be UWM_PREC_F32 (0), UWM_PREC_BF16X3 (1), UWM_PREC_BF16X3_ALL (2), UWM_PREC_F16X3 (3), UWM_PREC_F16X3_ALL (4), UWM_PREC_F16X1 (5) or UWM_PREC_F16X3_BWD2 (6), got %d", mode);
  h->prec = mode; h->plan.prec = mode;          // the workspace layout does not depend on the mode (the banks have one size)
  return 0;
}
int uwm_get_precision(uwm_handle h) { return h ? h->prec : -1; }
int uwm_set_precision_fill(uwm_handle h, int min_workgroups) {
  if (!h || min_workgroups < 0) return fail("uwm_set_precision_fill: bad argument");
  h->f3_min_wgs = min_workgroups; return 0;
}
int uwm_set_routing_batch(uwm_handle h, int batch) {
  if (!h || batch < 0 || batch > 255) return fail("uwm_set_routing_batch: batch must be 0 (= the real batch) .. 255");
  h->route_n = batch; return 0;
}
int uwm_routing_enable(uwm_handle h, int on) {
  if (!h) return fail("uwm_routing_enable: null handle");
  h->route_log_on = on != 0; if (!on) h->route_log.clear();
  return 0;
}
long long uwm_routing_dump(uwm_handle h, char* buf, long long cap, int clear) {
  if (!h) return -1;
  const long long need = (long long)h->route_log.size() + 1;
  if (buf && cap > 0) {
    const long long n = need <= cap ? need - 1 : cap - 1;
    memcpy(buf, h->route_log.data(), (size_t)n); buf[n] = 0;
  }
  if (clear) h->route_log.clear();
  return need;
}

// ---- data-parallel exchange on the C ABI: SUM all-reduce of the gradient arena ranges of backward
// stages [stage_begin, stage_end) over an RCCL communicator, one collective per stage (= bucket), enqueued on `stream`.
// RCCL is resolved at run time from the process (the library the caller's communicator came from), never linked.
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static nccl_allreduce_fn g_allreduce = nullptr;
static int resolve_rccl() {
  if (g_allreduce) return 0;
  void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
  if (!sym) {
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib) sym = dlsym(lib, "ncclAllReduce");
  }
  if (!sym) return fail("uwm_allreduce_grads: ncclAllReduce not found (load RCCL in the host process first)");
  g_allreduce = (nccl_allreduce_fn)sym; return 0;
}
int uwm_allreduce_grads(uwm_handle h, void* comm, int sb, int se, uwm_stream stream) {
  if (!h || !comm) return fail("uwm_allreduce_grads: null argument");
  if (!h->grads) return fail("uwm_allreduce_grads: no gradient arena bound");
  if (sb < 0 || se > h->nstages || sb >= se) return fail("uwm_allreduce_grads: bad stage range [%d,%d)", sb, se);
  if (resolve_rccl()) return 1;
  DeviceGuard guard(h->device);
  for (int k = sb; k < se; ++k) {
    const long long b = h->stage_begin[k], e = h->stage_begin[k + 1];
    if (e <= b) continue;
    const int rc = g_allreduce(h->grads + b, h->grads + b, (size_t)(e - b), /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    if (rc != 0) return fail("uwm_allreduce_grads: ncclAllReduce failed on bucket %d (ncclResult_t %d)", k, rc);
  }
  return 0;
}
float* uwm_grad_arena(uwm_handle h) { return h ? h->grads : nullptr; }
static Src to_src(const uwm_src* s) { return mk_src(s->ptr, s->C, s->H, s->W, s->scale, s->shift, s->relu, s->up); }

int uwm_op_conv(const uwm_src* s0, const uwm_src* s1, const float* w, int wrows, int Kpad, int kh, int kw, int stride,
                int pad, int N, int Cout, const float* bias, float* y, double* stats, int cfg, uwm_stream stream) {
  if (!s0 || !w || !y) return fail("uwm_op_conv: null argument");
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.s0 = to_src(s0); a.C0 = a.s0.C;
  if (s1) { a.s1 = to_src(s1); a.Ctot = a.C0 + a.s1.C; } else { a.s1 = a.s0; a.Ctot = a.C0; }
  a.Hl = a.s0.H << a.s0.up; a.Wl = a.s0.W << a.s0.up;
  a.w = w; a.wrows = wrows; a.Kpad = Kpad; a.ntaps = kh * kw; a.kw = kw;
  a.N = N; a.Ho = (a.Hl + 2 * pad - kh) / stride + 1; a.Wo = (a.Wl + 2 * pad - kw) / stride + 1;
  a.Cout = Cout; a.M = N * a.Ho * a.Wo;
  a.smul = stride; a.rmul = 1; a.off = -pad; a.sdiv = 1;
  a.out = y; a.bias = bias;
  if (stats) { a.ssum = stats; a.ssq = stats + Cout; }
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  if (cfg == 610) {                                     // the ResNet stem on conv_stem_f16x3.hip
    static float* sbuf = nullptr;
    if (!sbuf) HIPCHK(hipMalloc((void**)&sbuf, stem_f16x3_bank_floats() * sizeof(float)));
    if (kh != 7 || kw != 7 || a.Ctot != 4 || wrows != 64) return fail("uwm_op_conv: cfg 610 is the 7x7 stem (4 stored input channels, 64 outputs)");
    LCHK(launch_stem_f16x3_weights(w, Kpad, a.Ctot, sbuf, (hipStream_t)stream));
    a.wu = sbuf;
    LCHK(launch_conv_stem_f16x3(a, (hipStream_t)stream));
    return 0;
  }
  const bool reuse_bank = cfg >= 1600 && cfg <= 1607;      // 16xx = 6xx without re-packing the filter bank (kernel-only timing: the previous call must have been the same layer and layout)
  if (reuse_bank) cfg -= 1000;
  if (cfg >= 600 && cfg <= 607) {                     // 606 / 607: conv_f16x3v2 8-wave kernel; 600 auto | 601 four-wave kernel | 602 eight-wave kernel | 603 four-wave, 32-channel tiles | 604 / 605 conv_f16x3v2 64- / 32-channel tiles
    if (!op_wino_shape(a, kh, kw, stride, pad)) return fail("uwm_op_conv: cfg 600 (fp16x3) needs 3x3 s1 p1, Ho >= 8, Wo >= 16");
    if (op_f16x3_prepare(a, (hipStream_t)stream, cfg - 600, reuse_bank)) return 1;
  } else if (((cfg >= 300 && cfg < 500) || (cfg < 0 && winograd_mode() != 0)) && op_wino_shape(a, kh, kw, stride, pad)) {
    if (op_wino_prepare(a, 0, (hipStream_t)stream, cfg == 400)) return 1;
  } else if (cfg >= 300 && cfg < 500) return fail("uwm_op_conv: cfg 300 (Winograd) needs 3x3 s1 p1, channels %% 8 == 0, Ho >= 8, Wo >= 16");
  LCHK(launch_conv(a, (hipStream_t)stream, cfg));
  return 0;
}
int uwm_op_dgrad(const float* dy, int N, int Ho, int Wo, int Cout, const float* wd, int Cin, int KpadD, int kh, int kw,
                 int stride, int pad, int H, int W, const float* addend, const float* mask, const float* mscale,
                 const float* mshift, float* dx, uwm_stream stream) {
  if (!dy || !wd || !dx) return fail("uwm_op_dgrad: null argument");
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.s0 = mk_src(dy, Cout, Ho, Wo); a.s1 = a.s0; a.C0 = Cout; a.Ctot = Cout;
  a.w = wd; a.wrows = Cin; a.Kpad = KpadD; a.ntaps = kh * kw; a.kw = kw;
  a.N = N; a.Ho = H; a.Wo = W; a.Cout = Cin; a.M = N * H * W;
  a.Hl = Ho; a.Wl = Wo; a.smul = 1; a.rmul = -1; a.off = pad; a.sdiv = stride;
  a.out = dx; a.addend = addend; a.mask = mask; a.mscale = mscale; a.mshift = mshift;
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  if (winograd_mode() != 0 && op_wino_shape(a, kh, kw, stride, pad) && op_wino_prepare(a, 1, (hipStream_t)stream)) return 1;
  LCHK(launch_conv(a, (hipStream_t)stream));
  return 0;
}
int uwm_op_dgrad_upsplit(const float* dy, int N, int H, int W, int Cout, const float* wd, int C0, int C1, int KpadD,
                         float* gprev, const float* pmask, const float* pscale, const float* pshift, float* gskip,
                         uwm_stream stream) {
  if (!dy || !wd || !gprev || (C1 > 0 && !gskip)) return fail("uwm_op_dgrad_upsplit: null argument");
  ConvArgs a; memset(&a, 0, sizeof(a));
  a.s0 = mk_src(dy, Cout, H, W); a.s1 = a.s0; a.C0 = Cout; a.Ctot = Cout;
  a.w = wd; a.wrows = C0 + C1; a.Kpad = KpadD; a.ntaps = 9; a.kw = 3;
  a.N = N; a.Ho = H; a.Wo = W; a.Cout = C0 + C1; a.M = N * H * W;
  a.Hl = H; a.Wl = W; a.smul = 1; a.rmul = -1; a.off = 1; a.sdiv = 1;
  a.out = gskip; a.out_up = gprev; a.up_c0 = C0; a.up_mask = pmask; a.up_mscale = pscale; a.up_mshift = pshift;
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  if (!op_wino_shape(a, 3, 3, 1, 1) || (H & 1) || (W & 1)) return fail("uwm_op_dgrad_upsplit: needs even H >= 8, W >= 16, channels %% 8 == 0");
  if (op_wino_prepare(a, 1, (hipStream_t)stream)) return 1;
  LCHK(launch_conv(a, (hipStream_t)stream));
  return 0;
}
int uwm_op_wgrad(const uwm_src* s0, const uwm_src* s1, const float* dy, int N, int Ho, int Wo, int Cout, int wrows, int Kpad,
                 int kh, int kw, int stride, int pad, float* dw, int force_igemm, uwm_stream stream) {
  if (!s0 || !dy || !dw) return fail("uwm_op_wgrad: null argument");
  WgradArgs a; memset(&a, 0, sizeof(a));
  a.s0 = to_src(s0); a.C0 = a.s0.C;
  if (s1) { a.s1 = to_src(s1); a.Ctot = a.C0 + a.s1.C; } else { a.s1 = a.s0; a.Ctot = a.C0; }
  a.dy = dy; a.dw = dw; a.wrows = wrows; a.Kpad = Kpad; a.ntaps = kh * kw; a.kw = kw;
  a.N = N; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.M = N * Ho * Wo;
  a.Hl = a.s0.H << a.s0.up; a.Wl = a.s0.W << a.s0.up; a.stride = stride; a.pad = pad;
  a.dv_ctot = make_fastdiv(a.Ctot); a.dv_kw = make_fastdiv(a.kw);
  a.force_igemm = force_igemm;
  if ((force_igemm & 0xff) == 6) {               // fp16x3 direct weight gradient (tests / timing): max|dy| through a one-off reduction
    static float* xm = nullptr;
    if (!xm) HIPCHK(hipMalloc((void**)&xm, 32 * sizeof(float)));
    LCHK(launch_absmax32(dy, (size_t)N * Ho * Wo * Cout, xm, (hipStream_t)stream));
    a.prec = 2; a.xmax = xm;
  }
  LCHK(launch_wgrad(a, (hipStream_t)stream));
  return 0;
}
int uwm_op_pack_dgrad(const float* w, int Cout, int Kpad, int ntaps, int Cin, float* wd, int KpadD, int CoutP,
                      uwm_stream stream) {
  if (!w || !wd) return fail("uwm_op_pack_dgrad: null argument");
  LCHK(launch_pack_dgrad(w, Cout, Kpad, ntaps, Cin, wd, KpadD, CoutP, (hipStream_t)stream));
  return 0;
}
int uwm_op_bn_backward(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma,
                       double* scratch2c, float* dy, float* dgamma, float* dbeta, long long npix, int C, uwm_stream stream) {
  if (!g || !y || !mean || !rstd || !gamma || !scratch2c || !dy || !dgamma || !dbeta || npix < 1) return fail("uwm_op_bn_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(scratch2c, 0, 2 * (size_t)C * sizeof(double), st));
  LCHK(launch_bn_bwd_reduce(g, y, mean, rstd, scratch2c, scratch2c + C, (size_t)npix, C, st));
  LCHK(launch_bn_bwd_apply(g, y, mean, rstd, gamma, scratch2c, scratch2c + C, dy, dgamma, dbeta, (size_t)npix, C, st));
  return 0;
}
int uwm_op_upsplit(const float* dcat, int N, int H, int W, int C0, int C1, float* gprev, const float* pmask,
                   const float* pscale, const float* pshift, float* gskip, uwm_stream stream) {
  if (!dcat || !gprev) return fail("uwm_op_upsplit: null argument");
  LCHK(launch_upsplit(dcat, N, H, W, C0, C1, gprev, pmask, pscale, pshift, gskip, (hipStream_t)stream));
  return 0;
}
int uwm_op_residual(const float* y, const float* s2, const float* b2, const float* id, const float* sd, const float* bd,
                    float* out, long long npix, int C, uwm_stream stream) {
  if (!y || !s2 || !b2 || !id || !out) return fail("uwm_op_residual: null argument");
  LCHK(launch_residual(y, s2, b2, id, sd, bd, out, (size_t)npix, C, (hipStream_t)stream));
  return 0;
}
int uwm_op_maxpool_backward(const float* gout, const uint8_t* idx, const float* addend, const uwm_src* in, int N, float* gin,
                            uwm_stream stream) {
  if (!gout || !idx || !in || !gin) return fail("uwm_op_maxpool_backward: null argument");
  Src s = to_src(in);
  LCHK(launch_maxpool_bwd(gout, idx, addend, s, gin, N, (s.H - 1) / 2 + 1, (s.W - 1) / 2 + 1, (hipStream_t)stream));
  return 0;
}
int uwm_op_maxpool(const uwm_src* in, int N, float* out, uint8_t* idx, uwm_stream stream) {
  if (!in || !out) return fail("uwm_op_maxpool: null argument");
  Src s = to_src(in);
  LCHK(launch_maxpool_fwd(s, out, idx, N, (s.H - 1) / 2 + 1, (s.W - 1) / 2 + 1, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
