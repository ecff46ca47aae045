// Plan construction: parameter tables (TF variable order of the reference), layer tables,
// channel-slab layouts and the workspace arena.
//
// Reference being restated structurally (no code shared): models/PWCNet/model_pwcnet.py:149-168,
// 476-506,559-576,599-649 (PWC-Net lg-6-2), models/nets.py:4-42 (generator), :45-110 (recover).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "conv_host.h"
#include "elementwise.h"
#include "plan.h"

namespace udet {

// ------------------------------------------------------------------ params ----
int NetParams::add(const std::string& name, int a, int b, int c, int d) {
  ParamDesc q;
  q.name = name;
  q.shape[0] = a; q.shape[1] = b; q.shape[2] = c; q.shape[3] = d;
  q.rank = d ? 4 : (c ? 3 : (b ? 2 : 1));
  q.count = (size_t)a * (b ? b : 1) * (c ? c : 1) * (d ? d : 1);
  q.offset = total;
  total += q.count;
  p.push_back(q);
  return (int)p.size() - 1;
}
int NetParams::find(const std::string& name) const {
  for (size_t i = 0; i < p.size(); ++i)
    if (p[i].name == name) return (int)i;
  return -1;
}

static const int PWC_CH[7] = {0, 16, 32, 64, 96, 128, 196};
static const int EST_CO[5] = {128, 128, 96, 64, 32};
static const int CTX_CO[7] = {128, 128, 128, 96, 64, 32, 2};
static const int CTX_DIL[7] = {1, 2, 4, 8, 16, 1, 1};

static std::string S(const char* fmt, ...) {
  char b[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof(b), fmt, ap);
  va_end(ap);
  return b;
}

static NetParams build_pwc_params() {
  NetParams n;
  int cin = 3;
  for (int l = 1; l <= 6; ++l) {
    const int f = PWC_CH[l];
    const char* suf[3] = {"a", "aa", "b"};
    for (int j = 0; j < 3; ++j) {
      n.add(S("pwcnet/featpyr/conv%d%s/kernel", l, suf[j]), 3, 3, j == 0 ? cin : f, f);
      n.add(S("pwcnet/featpyr/conv%d%s/bias", l, suf[j]), f);
    }
    cin = f;
  }
  for (int l = 6; l >= 2; --l) {
    int x = l == 6 ? 81 : 81 + PWC_CH[l] + 4;
    for (int i = 0; i < 5; ++i) {
      n.add(S("pwcnet/predict_flow/conv%d_%d/kernel", l, i), 3, 3, x, EST_CO[i]);
      n.add(S("pwcnet/predict_flow/conv%d_%d/bias", l, i), EST_CO[i]);
      x += EST_CO[i];
    }
    n.add(S("pwcnet/predict_flow/flow%d/kernel", l), 3, 3, x, 2);
    n.add(S("pwcnet/predict_flow/flow%d/bias", l), 2);
    int ci = x;
    for (int i = 0; i < 7; ++i) {
      n.add(S("pwcnet/ctxt/dc_conv%d%d/kernel", l, i + 1), 3, 3, ci, CTX_CO[i]);
      n.add(S("pwcnet/ctxt/dc_conv%d%d/bias", l, i + 1), CTX_CO[i]);
      ci = CTX_CO[i];
    }
    if (l != 2) {
      n.add(S("pwcnet/upsample/up_flow%d/kernel", l), 4, 4, 2, 2);
      n.add(S("pwcnet/upsample/up_flow%d/bias", l), 2);
      n.add(S("pwcnet/upsample/up_feat%d/kernel", l), 4, 4, 2, x);
      n.add(S("pwcnet/upsample/up_feat%d/bias", l), 2);
    }
  }
  return n;
}

struct GenSpec { const char* name; int cin, cout, k, s, d; bool up; int act; };
static const GenSpec GEN[17] = {
    {"conv1", 5, 32, 5, 1, 1, false, ACT_ELU},           {"conv2_downsample", 32, 64, 3, 2, 1, false, ACT_ELU},
    {"conv3", 64, 64, 3, 1, 1, false, ACT_ELU},          {"conv4_downsample", 64, 128, 3, 2, 1, false, ACT_ELU},
    {"conv5", 128, 128, 3, 1, 1, false, ACT_ELU},        {"conv6", 128, 128, 3, 1, 1, false, ACT_ELU},
    {"conv7_atrous", 128, 128, 3, 1, 2, false, ACT_ELU}, {"conv8_atrous", 128, 128, 3, 1, 4, false, ACT_ELU},
    {"conv9_atrous", 128, 128, 3, 1, 8, false, ACT_ELU}, {"conv10_atrous", 128, 128, 3, 1, 16, false, ACT_ELU},
    {"conv11", 128, 128, 3, 1, 1, false, ACT_ELU},       {"conv12", 128, 128, 3, 1, 1, false, ACT_ELU},
    {"conv13_upsample", 128, 64, 3, 1, 1, true, ACT_ELU}, {"conv14", 64, 64, 3, 1, 1, false, ACT_ELU},
    {"conv15_upsample", 64, 32, 3, 1, 1, true, ACT_ELU},  {"conv16", 32, 16, 3, 1, 1, false, ACT_ELU},
    {"conv17", 16, 2, 3, 1, 1, false, ACT_NONE}};

static NetParams build_gen_params() {
  NetParams n;
  for (int i = 0; i < 17; ++i) {
    const GenSpec& g = GEN[i];
    const std::string pre = g.up ? S("MaskNet/%s/%s_conv", g.name, g.name) : S("MaskNet/%s", g.name);
    n.add(pre + "/kernel", g.k, g.k, g.cin, g.cout);
    n.add(pre + "/bias", g.cout);
    n.add(S("MaskNet/%s/bn/gamma", g.name), g.cout);
    n.add(S("MaskNet/%s/bn/beta", g.name), g.cout);
  }
  return n;
}

struct EncSpec { const char* suf; int cin, cout, k, s; };
static const EncSpec ENC[9] = {{"conv1", 0, 16, 7, 2},   {"conv2", 16, 32, 5, 2},   {"conv3", 32, 64, 5, 2},
                               {"conv31", 64, 64, 3, 1}, {"conv4", 64, 128, 3, 2},  {"conv41", 128, 128, 3, 1},
                               {"conv5", 128, 128, 3, 2}, {"conv51", 128, 128, 3, 1}, {"conv6", 128, 128, 3, 2}};
struct DecSpec { const char* name; int k, cin, cout; };
static const DecSpec DEC[14] = {{"deconv5", 4, 256, 128}, {"flow5", 3, 384, 2},   {"deconv4", 4, 384, 128}, {"upflow4", 4, 2, 2},
                                {"flow4", 3, 386, 2},     {"deconv3", 4, 386, 64}, {"upflow3", 4, 2, 2},     {"flow3", 3, 194, 2},
                                {"deconv2", 4, 194, 32},  {"upflow2", 4, 2, 2},    {"flow2", 3, 98, 2},      {"deconv1", 4, 98, 16},
                                {"upflow1", 4, 2, 2},     {"flow1", 5, 50, 2}};

static NetParams build_rec_params() {
  NetParams n;
  for (int e = 0; e < 2; ++e)
    for (int i = 0; i < 9; ++i) {
      const int ci = ENC[i].cin ? ENC[i].cin : (e == 0 ? 3 : 4);
      n.add(S("FlownetS/%c%s/weights", e == 0 ? 'a' : 'b', ENC[i].suf), ENC[i].k, ENC[i].k, ci, ENC[i].cout);
      n.add(S("FlownetS/%c%s/biases", e == 0 ? 'a' : 'b', ENC[i].suf), ENC[i].cout);
    }
  for (int i = 0; i < 14; ++i) {
    n.add(S("FlownetS/%s/weights", DEC[i].name), DEC[i].k, DEC[i].k, DEC[i].cin, DEC[i].cout);
    n.add(S("FlownetS/%s/biases", DEC[i].name), DEC[i].cout);
  }
  return n;
}

const NetParams& net_params(int net) {
  static const NetParams pwc = build_pwc_params(), gen = build_gen_params(), rec = build_rec_params();
  return net == NET_PWC ? pwc : (net == NET_GEN ? gen : rec);
}

// ------------------------------------------------------------------- plan ----
int Plan::add_buf(const std::string& name, int n, int h, int w, int ld) {
  Buf b;
  b.name = name; b.n = n; b.h = h; b.w = w; b.ld = ld; b.off = 0;
  bufs.push_back(b);
  buf_by_name[name] = (int)bufs.size() - 1;
  return (int)bufs.size() - 1;
}
int Plan::bid(const std::string& name) const {
  auto it = buf_by_name.find(name);
  return it == buf_by_name.end() ? -1 : it->second;
}

static Layer mk_layer(int net, const std::string& name, const std::string& wname, const std::string& bname, int k, int cin,
                      int cout, int stride, int dil, int act, float alpha) {
  Layer L;
  L.name = name; L.net = net;
  L.kh = L.kw = k; L.cin = cin; L.cout = cout; L.stride = stride; L.dil = dil;
  L.act = act; L.alpha = alpha;
  const NetParams& np = net_params(net);
  L.w_idx = np.find(wname);
  L.b_idx = np.find(bname);
  L.Kc = round_up(cin, cin <= 4 ? 4 : 8);  // 2..4-channel inputs (RGB, flows, recover's 4-channel input) are stored with ld = 4
  L.ldw = round_up(cout, 4);
  L.k_split = L.Kc;
  L.k_gap = 0;
  return L;
}

static size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

Plan::~Plan() {
  for (auto& st : cand)
    if (st) (void)hipStreamDestroy(st);
  for (auto& e : ev_pool) (void)hipEventDestroy(e);
  for (auto& e : ev_pool_prefetch) (void)hipEventDestroy(e);
  if (prefetch_ev) (void)hipEventDestroy(prefetch_ev);
  for (auto* r : prof) delete r;
  for (auto& ev : grad_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ovf_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (ovf_host) (void)hipHostFree(ovf_host);
}

// recover decoder's up-conv algebra: fewest low-resolution source pixels (batch included) of a level whose backward-data pass uses it
// too (measured on the benchmark plan: 12 x 12 x 24 pixels lose to the up-sampled form -- too few tiles for the chip --, 12 x 24 x 48 win);
// the tests set 0 so that small plans run the backward form on every level (udet_debug_upb_min_pixels)
static long g_upb_bwd_min = 8192;
void plan_debug_upb_min_pixels(long v) { g_upb_bwd_min = v < 0 ? 8192 : v; }
#ifdef UDET_EXPERIMENT  // libudet_exp.so only (plan.h): the release library has neither the table nor the setter
static long g_knob[UDET_KNOB_COUNT] = {};
void plan_debug_knob(int id, long v) { if (id >= 0 && id < UDET_KNOB_COUNT) g_knob[id] = v; }
long plan_knob(int id) { return id >= 0 && id < UDET_KNOB_COUNT ? g_knob[id] : 0; }
extern "C" void udet_exp_knob(int id, long v) { plan_debug_knob(id, v); }
#endif

// Segments and tap tables of the recover decoder's up-conv algebra for a low-resolution source of h x w (plan_exec.hip has the algebra).
// Merge rows that are all zero in a variant (0 interior, 1 last row / column) carry no tap.
static inline bool upb_used(int parity, int last, int a) { return !last || (parity == 0 ? a < 2 : a < 1); }
static void build_upb_launches(Layer& L, int h, int w) {
  // forward: 4 output regions {interior, last row, last column, corner} x 4 parity classes; class (py, px), merged tap (a, b) of the
  // quotient pixel q reads x^[q + p + (a, b)] (ring coordinates) with weight set = region
  Layer::SegLaunch& F = L.upb_f;
  F.nseg = 0; F.taps.clear();
  for (int r = 0; r < 4; ++r) {
    const int rv = r & 1, cv = r >> 1;
    const int r0 = rv ? h - 1 : 0, c0 = cv ? w - 1 : 0, nr = rv ? 1 : h - 1, nc = cv ? 1 : w - 1;
    for (int c = 0; c < 4; ++c) {
      const int py = c >> 1, px = c & 1;
      ConvSeg& g = F.seg[F.nseg];
      memset(&g, 0, sizeof(g));
      g.oy = 2 * r0 + py; g.ox = 2 * c0 + px; g.h = nr; g.w = nc;
      F.seg_tap[F.nseg++] = (int)F.taps.size();
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          if (!upb_used(py, rv, a) || !upb_used(px, cv, b)) continue;
          F.taps.push_back(ConvTap{r0 + a + py, c0 + b + px, r * 36 + 9 * c + 3 * a + b});
        }
    }
  }
  F.seg_tap[F.nseg] = (int)F.taps.size();
  // backward-data: the ringed gradient x^[R] collects dU[2R - p - 2a] W~[class p][a]^T where the quotient pixel q = R - p - a exists;
  // q == h - 1 takes the last-row set.  Rows split into {R < h - 1} (every q interior) and R = h - 1, h, h + 1 (a fixed mix per row);
  // columns alike: 16 segments, each with its own (tap, weight set) list.
  Layer::SegLaunch& B = L.upb_b;
  B.nseg = 0; B.taps.clear();
  struct Combo { int p, a, last; };
  auto combos = [](int grp, std::vector<Combo>& out) {  // grp 0: R < n - 1; grp k: R = n - 2 + k
    out.clear();
    for (int p = 0; p < 2; ++p)
      for (int a = 0; a < 3; ++a) {
        int last = 0;
        if (grp > 0) {
          const int rp = grp - 1;  // q = n - 1 + rp - p - a
          if (p + a < rp) continue;
          last = (p + a == rp);
        }
        if (upb_used(p, last, a)) out.push_back(Combo{p, a, last});
      }
  };
  std::vector<Combo> rc, cc;
  for (int gr = 0; gr < 4; ++gr)
    for (int gc = 0; gc < 4; ++gc) {
      const int R0 = gr == 0 ? 0 : h - 2 + gr, C0 = gc == 0 ? 0 : w - 2 + gc;
      ConvSeg& g = B.seg[B.nseg];
      memset(&g, 0, sizeof(g));
      g.oy = R0; g.ox = C0; g.h = gr == 0 ? h - 1 : 1; g.w = gc == 0 ? w - 1 : 1;
      B.seg_tap[B.nseg++] = (int)B.taps.size();
      combos(gr, rc);
      combos(gc, cc);
      for (const Combo& y : rc)
        for (const Combo& x : cc)
          B.taps.push_back(ConvTap{2 * R0 - y.p - 2 * y.a, 2 * C0 - x.p - 2 * x.a, (y.last + 2 * x.last) * 36 + 9 * (2 * y.p + x.p) + 3 * y.a + x.a});
    }
  B.seg_tap[B.nseg] = (int)B.taps.size();
}

Plan* plan_build(const Config& cfg) {
  if (cfg.batch > 16) {
    set_error("plan: batch %d > 16 per GPU is not laid out (reduction scratch); shard over more ranks", cfg.batch);
    return nullptr;
  }
  if (cfg.batch < 1 || cfg.in_h % 64 || cfg.in_w % 64 || cfg.img_h % 64 || cfg.img_w % 64 || cfg.in_h < 64 ||
      cfg.in_w < 64 || cfg.img_h < 64 || cfg.img_w < 64) {
    set_error("plan: batch>=1 and in_h,in_w,img_h,img_w multiples of 64 required (got B=%d %dx%d -> %dx%d)", cfg.batch,
              cfg.in_h, cfg.in_w, cfg.img_h, cfg.img_w);
    return nullptr;
  }
  Plan* P = new Plan();
  P->cfg = cfg;
  const int B = cfg.batch;

  // ======================= PWC-Net =======================
  {
    int h = cfg.in_h, w = cfg.in_w;
    P->add_buf("pwc.x8", 2 * B, h, w, 4);
    int prev = P->bid("pwc.x8"), prev_c = 3;
    for (int l = 1; l <= 6; ++l) {
      h /= 2; w /= 2;
      const int f = PWC_CH[l];
      const int ta = P->add_buf(S("pwc.t%da", l), 2 * B, h, w, f);
      const int tb = P->add_buf(S("pwc.t%db", l), 2 * B, h, w, f);
      const int c = P->add_buf(S("pwc.c%d", l), 2 * B, h, w, f);
      const char* suf[3] = {"a", "aa", "b"};
      const int xs[3] = {prev, ta, tb}, ys[3] = {ta, tb, c};
      for (int j = 0; j < 3; ++j) {
        const std::string nm = S("pwcnet/featpyr/conv%d%s", l, suf[j]);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 3, j == 0 ? prev_c : f, f, j == 0 ? 2 : 1, 1,
                           ACT_LEAKY, 0.1f);
        L.x = xs[j]; L.y = ys[j];
        L.H = j == 0 ? h * 2 : h; L.W = j == 0 ? w * 2 : w;
        P->pwc.push_back(L);
      }
      prev = c; prev_c = f;
    }
    for (int l = 6; l >= 2; --l) {
      const int hh = cfg.in_h >> l, ww = cfg.in_w >> l;
      const int C = PWC_CH[l];
      const int ld = l == 6 ? 536 : 536 + C;
      const int slab = P->add_buf(S("pwc.slab%d", l), B, hh, ww, ld);
      if (l != 6) P->add_buf(S("pwc.warp%d", l), B, hh, ww, C);
      const int F = P->add_buf(S("pwc.flow%d", l), B, hh, ww, 4);
      const int FR = P->add_buf(S("pwc.rflow%d", l), B, hh, ww, 4);
      const int starts[5] = {448, 320, 192, 96, 32}, outs[5] = {320, 192, 96, 32, 0};
      int x = l == 6 ? 81 : 81 + C + 4;
      for (int i = 0; i < 5; ++i) {
        const std::string nm = S("pwcnet/predict_flow/conv%d_%d", l, i);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 3, x, EST_CO[i], 1, 1, ACT_LEAKY, 0.1f);
        L.x = slab; L.x_coff = starts[i]; L.y = slab; L.y_coff = outs[i];
        L.Kc = ld - starts[i];
        L.k_split = 448 - starts[i] + 81;
        L.k_gap = l == 6 ? 0 : 3;
        if (l == 6) L.k_split = L.Kc;
        L.H = hh; L.W = ww;
        P->pwc.push_back(L);
        x += EST_CO[i];
      }
      {
        const std::string nm = S("pwcnet/predict_flow/flow%d", l);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 3, x, 2, 1, 1, ACT_NONE, 0.f);
        L.x = slab; L.x_coff = 0; L.y = F; L.Kc = ld;
        L.k_split = l == 6 ? ld : 448 + 81; L.k_gap = l == 6 ? 0 : 3;
        L.H = hh; L.W = ww;
        P->pwc.push_back(L);
      }
      int ci = x, xb = slab;
      for (int i = 0; i < 7; ++i) {
        const std::string nm = S("pwcnet/ctxt/dc_conv%d%d", l, i + 1);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 3, ci, CTX_CO[i], 1, CTX_DIL[i], i < 6 ? ACT_LEAKY : ACT_NONE,
                           0.1f);
        L.x = xb; L.x_coff = 0;
        if (i == 0) { L.Kc = ld; L.k_split = l == 6 ? ld : 448 + 81; L.k_gap = l == 6 ? 0 : 3; }
        if (i < 6) L.y = P->add_buf(S("pwc.dc%d_%d", l, i + 1), B, hh, ww, CTX_CO[i]);
        else { L.y = FR; L.res = F; }  // refined flow = flow + dc_conv7 (model_pwcnet.py:576)
        L.H = hh; L.W = ww;
        P->pwc.push_back(L);
        ci = CTX_CO[i]; xb = L.y;
      }
      (void)x;
    }
    // learned x2 upsampling into the next level's slab (needs that slab to exist)
    for (int l = 6; l >= 3; --l) {
      const int hh = cfg.in_h >> l, ww = cfg.in_w >> l;
      const int Cn = PWC_CH[l - 1];
      const int ld = l == 6 ? 536 : 536 + PWC_CH[l];
      const int x = l == 6 ? 529 : 529 + PWC_CH[l] + 4 - 0;  // real channel count of upfeat
      const int nslab = P->bid(S("pwc.slab%d", l - 1));
      {
        const std::string nm = S("pwcnet/upsample/up_flow%d", l);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 4, 2, 2, 2, 1, ACT_NONE, 0.f);
        L.transposed = true;
        L.x = P->bid(S("pwc.rflow%d", l)); L.y = nslab; L.y_coff = 532 + Cn;
        L.H = hh; L.W = ww;
        P->pwc.push_back(L);
      }
      {
        const std::string nm = S("pwcnet/upsample/up_feat%d", l);
        Layer L = mk_layer(NET_PWC, nm, nm + "/kernel", nm + "/bias", 4, x, 2, 2, 1, ACT_NONE, 0.f);
        L.transposed = true;
        L.x = P->bid(S("pwc.slab%d", l)); L.y = nslab; L.y_coff = 534 + Cn;
        L.Kc = ld; L.k_split = l == 6 ? ld : 448 + 81; L.k_gap = l == 6 ? 0 : 3;
        L.H = hh; L.W = ww;
        P->pwc.push_back(L);
      }
    }
    P->add_buf("flow_full", B, cfg.in_h, cfg.in_w, 2);
  }

  // ======================= shared 192x384 tensors =======================
  const int H = cfg.img_h, W = cfg.img_w;
  P->add_buf("image", B, H, W, 3);
  P->add_buf("flow", B, H, W, 2);
  P->add_buf("image.next", B, H, W, 3);  // staging of the prefetched next step (udet_prefetch_flow)
  P->add_buf("flow.next", B, H, W, 2);
  P->add_buf("mask", B, H, W, 1);
  P->add_buf("pred", 3 * B, H, W, 2);
  P->add_buf("d.pred", 3 * B, H, W, 2);
  P->add_buf("d.mask", B, H, W, 1);

  // ======================= generator =======================
  {
    int x = P->add_buf("gen.in", B, H, W, 8);
    int h = H, w = W;
    for (int i = 0; i < 17; ++i) {
      const GenSpec& g = GEN[i];
      const std::string pre = g.up ? S("MaskNet/%s/%s_conv", g.name, g.name) : S("MaskNet/%s", g.name);
      Layer L = mk_layer(NET_GEN, g.name, pre + "/kernel", pre + "/bias", g.k, g.cin, g.cout, g.s, g.d, g.act, 0.f);
      L.g_idx = net_params(NET_GEN).find(S("MaskNet/%s/bn/gamma", g.name));
      L.be_idx = net_params(NET_GEN).find(S("MaskNet/%s/bn/beta", g.name));
      L.up = g.up;
      L.x = x; L.H = h; L.W = w;
      if (g.up) { h *= 2; w *= 2; }
      if (g.s == 2) { h /= 2; w /= 2; }
      const int ldo = round_up(g.cout, 8);
      L.y = P->add_buf(S("gen.a%d", i + 1), B, h, w, ldo);
      P->add_buf(S("gen.d%d", i + 1), B, h, w, ldo);  // gradient w.r.t. this layer's (summed) output
      P->add_buf(S("gen.u%d", i + 1), B, h, w, ldo);  // ... times act'(output): what the layer's own backward launches read
      if (i == 10 || i == 13 || i == 14) {          // conv11 + x2, conv14 + x1, conv15_upsample + x0 (nets.py:29,32,33)
        L.y2 = L.y;                                  // a_k keeps the pre-skip activation
        L.y = P->add_buf(S("gen.s%d", i + 1), B, h, w, ldo);
        L.res = P->bid(i == 10 ? "gen.a6" : (i == 13 ? "gen.a3" : "gen.a1"));
      }
      if (g.up) P->add_buf(S("gen.dup%d", i + 1), B, h, w, g.cin);  // gradient w.r.t. the upsampled input
      x = L.y;
      P->gen.push_back(L);
    }
  }

  // ======================= recover (3 calls batched) =======================
  {
    const int N = 3 * B;
    const int imgin = P->add_buf("rec.imgin", N, H, W, 4);
    const int fin = P->add_buf("rec.fin", N, H, W, 4);
    P->add_buf("rec.d.fin", N, H, W, 4);
    const int hs[7] = {H, H / 2, H / 4, H / 8, H / 16, H / 32, H / 64}, wsz[7] = {W, W / 2, W / 4, W / 8, W / 16, W / 32, W / 64};
    // concat slabs: [deconv | bconv | aconv | upflow(2)+pad(6)]
    const int cc[6] = {0, 16, 32, 64, 128, 128};  // channels of the skip at level k (bconv1,2,31,41,51)
    int concat[6];
    for (int k = 1; k <= 5; ++k) {
      const int ld = k == 5 ? 384 : 3 * cc[k] + 8;
      concat[k] = P->add_buf(S("rec.concat%d", k), N, hs[k], wsz[k], ld);
      P->add_buf(S("rec.d.concat%d", k), N, hs[k], wsz[k], ld);
    }
    const int conv6 = P->add_buf("rec.conv6", N, hs[6], wsz[6], 256);
    P->add_buf("rec.d.conv6", N, hs[6], wsz[6], 256);
    // encoder-only intermediates
    const char* mid[3] = {"3", "4", "5"};
    const int midlvl[3] = {3, 4, 5}, midc[3] = {64, 128, 128};
    for (int e = 0; e < 2; ++e)
      for (int j = 0; j < 3; ++j) {
        P->add_buf(S("rec.%c%s", e ? 'b' : 'a', mid[j]), N, hs[midlvl[j]], wsz[midlvl[j]], midc[j]);
        P->add_buf(S("rec.d.%c%s", e ? 'b' : 'a', mid[j]), N, hs[midlvl[j]], wsz[midlvl[j]], midc[j]);
      }
    for (int e = 0; e < 2; ++e) {
      const char ec = e ? 'b' : 'a';
      // (x buffer, x_coff) -> (y buffer, y_coff) per encoder conv
      struct IO { int x, xc, y, yc, lvl_in; };
      const int so = e ? 1 : 2;  // slab segment index: bconv at [cc, 2cc), aconv at [2cc, 3cc)
      IO io[9] = {
          {e ? fin : imgin, 0, concat[1], so * cc[1], 0},
          {concat[1], so * cc[1], concat[2], so * cc[2], 1},
          {concat[2], so * cc[2], P->bid(S("rec.%c3", ec)), 0, 2},
          {P->bid(S("rec.%c3", ec)), 0, concat[3], so * cc[3], 3},
          {concat[3], so * cc[3], P->bid(S("rec.%c4", ec)), 0, 3},
          {P->bid(S("rec.%c4", ec)), 0, concat[4], so * cc[4], 4},
          {concat[4], so * cc[4], P->bid(S("rec.%c5", ec)), 0, 4},
          {P->bid(S("rec.%c5", ec)), 0, concat[5], so * cc[5], 5},
          {concat[5], so * cc[5], conv6, e ? 128 : 0, 5},
      };
      for (int i = 0; i < 9; ++i) {
        const int ci = ENC[i].cin ? ENC[i].cin : (e == 0 ? 3 : 4);
        const std::string nm = S("%c%s", ec, ENC[i].suf);
        Layer L = mk_layer(NET_REC, nm, "FlownetS/" + nm + "/weights", "FlownetS/" + nm + "/biases", ENC[i].k, ci, ENC[i].cout,
                           ENC[i].s, 1, ACT_LEAKY, 0.2f);
        L.x = io[i].x; L.x_coff = io[i].xc; L.y = io[i].y; L.y_coff = io[i].yc;
        L.H = hs[io[i].lvl_in]; L.W = wsz[io[i].lvl_in];
        P->rec.push_back(L);
      }
    }
    // decoder
    int src = conv6, src_lvl = 6;
    int flow_prev = -1;
    for (int k = 5; k >= 1; --k) {
      const int ldsrc = P->buf(src).ld;
      const int r = P->add_buf(S("rec.r%d", k + 1), N, hs[k], wsz[k], ldsrc);  // resize(src) to level k
      P->add_buf(S("rec.d.r%d", k + 1), N, hs[k], wsz[k], ldsrc);
      {
        const DecSpec& d = DEC[k == 5 ? 0 : (k == 4 ? 2 : (k == 3 ? 5 : (k == 2 ? 8 : 11)))];
        Layer L = mk_layer(NET_REC, d.name, S("FlownetS/%s/weights", d.name), S("FlownetS/%s/biases", d.name), 4, d.cin, d.cout, 1,
                           1, ACT_LEAKY, 0.2f);
        L.x = r; L.y = concat[k]; L.y_coff = 0; L.Kc = ldsrc; L.H = hs[k]; L.W = wsz[k];
        // levels 1-4 (exact x2, source at least 4x4): four 3x3 convolutions on the ringed low-resolution source instead of the
        // 4x4 convolution over the up-sampled tensor (9 of 16 tap products; plan_exec.hip).  The filter gradient keeps the up-sampled form.
        // (fp16 plans too since round 5: 4.13 -> 4.06 ms per step at batch 2, the mode's parity tests unchanged -- the negated ring's
        // cancellation against the merged centre weight leaves an fp16 rounding of the weights, like every other product of the mode)
        if (k <= 4 && hs[k] == 2 * hs[k + 1] && wsz[k] == 2 * wsz[k + 1] && hs[k + 1] >= 4 && wsz[k + 1] >= 4) {
          L.upb = true;
          L.upb_bwd = (long)N * hs[k + 1] * wsz[k + 1] >= g_upb_bwd_min;
          L.upb_split = d.cout <= 16;
          L.src = src;
          L.xhat = P->add_buf(S("rec.p%d", k + 1), N, hs[k + 1] + 2, wsz[k + 1] + 2, ldsrc);
          // (the ringed gradient grid exists only where the level's backward-data pass takes the low-resolution form)
          if (L.upb_bwd) P->add_buf(S("rec.d.p%d", k + 1), N, hs[k + 1] + 2, wsz[k + 1] + 2, ldsrc);
          build_upb_launches(L, hs[k + 1], wsz[k + 1]);
        }
        P->rec.push_back(L);
      }
      if (k < 5) {
        const int rf = P->add_buf(S("rec.rf%d", k + 1), N, hs[k], wsz[k], 4);
        P->add_buf(S("rec.d.rf%d", k + 1), N, hs[k], wsz[k], 4);
        const std::string nm = S("upflow%d", k);
        Layer L = mk_layer(NET_REC, nm, "FlownetS/" + nm + "/weights", "FlownetS/" + nm + "/biases", 4, 2, 2, 1, 1, ACT_NONE, 0.f);
        L.x = rf; L.y = concat[k]; L.y_coff = 3 * cc[k]; L.H = hs[k]; L.W = wsz[k];
        P->rec.push_back(L);
        (void)flow_prev;
      }
      {
        const std::string nm = S("flow%d", k);
        const int cin = k == 5 ? 384 : 3 * cc[k] + 2;
        Layer L = mk_layer(NET_REC, nm, "FlownetS/" + nm + "/weights", "FlownetS/" + nm + "/biases", k == 1 ? 5 : 3, cin, 2, 1, 1,
                           ACT_NONE, 0.f);
        L.x = concat[k]; L.Kc = P->buf(concat[k]).ld;
        L.y = P->add_buf(S("rec.flow%d", k), N, hs[k], wsz[k], 4);
        P->add_buf(S("rec.d.flow%d", k), N, hs[k], wsz[k], 4);
        L.H = hs[k]; L.W = wsz[k];
        flow_prev = L.y;
        P->rec.push_back(L);
      }
      src = concat[k]; src_lvl = k;
    }
    (void)src_lvl;
  }

  // The generator-loss backward pass runs concurrently with the recover-loss pass: it gets its own copy of the
  // recover-side gradient buffers ("rec.e.*", "e.pred" mirror "rec.d.*", "d.pred").
  {
    const size_t nb = P->bufs.size();
    for (size_t i = 0; i < nb; ++i) {
      const Buf b = P->bufs[i];
      if (b.name.rfind("rec.d.", 0) == 0) {
        P->add_buf("rec.e." + b.name.substr(6), b.n, b.h, b.w, b.ld);
        P->add_buf("rec.ud." + b.name.substr(6), b.n, b.h, b.w, b.ld);  // dU = gradient * act'(activation) mirrors
        P->add_buf("rec.ue." + b.name.substr(6), b.n, b.h, b.w, b.ld);
      }
      if (b.name == "d.pred") P->add_buf("e.pred", b.n, b.h, b.w, b.ld);
    }
  }

  // 2-channel heads whose K (input channels) is deep: GEMM + gather formulation (see run_fwd)
  auto mark_heads = [&](std::vector<Layer>& v, int N) {
    for (auto& L : v) {
      const bool head = L.cout == 2 && L.cin >= 32 && L.kh * L.kw * L.cout <= 64 && L.dil == 1 && !L.up && L.res < 0 && L.y2 < 0 &&
                        ((L.transposed && L.kh == 4) || (!L.transposed && L.stride == 1));
      if (!head) continue;
      L.col2im = true;
      L.ldz = round_up(L.kh * L.kw * L.cout, 32);
      L.zbuf = P->add_buf("z." + L.name, N, L.H, L.W, L.ldz);
    }
  };
  mark_heads(P->pwc, B);
  mark_heads(P->rec, 3 * B);

  // ======================= packed weights =======================
  size_t off = 0;
  auto place = [&](Layer& L, bool trainable) {
    const int T = L.kh * L.kw;
    L.wp_off = off; off = align64(off + (size_t)T * L.Kc * L.ldw);
    L.bias_f_off = off; off = align64(off + L.cout);
    L.scale_off = off; off = align64(off + L.cout);
    if (L.col2im) { L.wz_off = off; off = align64(off + (size_t)L.Kc * L.ldz); }
    if (trainable) {
      L.KcT = round_up(L.cout, (L.cout <= 4 && L.net == NET_REC) ? 4 : 8);  // recover's 2-channel outputs live in ld = 4 buffers
      L.ldwT = round_up(L.cin, 4);
      L.wpT_off = off; off = align64(off + (size_t)T * L.KcT * L.ldwT);
    }
    if (L.up) {
      L.wu_off = off; off = align64(off + (size_t)16 * L.Kc * L.ldw);
      L.wuT_off = off; off = align64(off + (size_t)16 * L.KcT * L.ldwT);
    }
    if (L.upb) {  // four weight sets back to back each way (a tap's widx = set * 36 + class * 9 + merged tap), the two tap tables
      L.wupb_off = off; off = align64(off + (size_t)4 * 36 * L.Kc * L.ldw);
      L.upb_f.tab_off = off; off = align64(off + L.upb_f.taps.size() * (sizeof(ConvTap) / sizeof(float)));
      if (L.upb_bwd) {  // (a forward-only level reads neither the transposed sets nor the backward table)
        L.wupbT_off = off; off = align64(off + (size_t)4 * 36 * L.KcT * L.ldwT);
        L.upb_b.tab_off = off; off = align64(off + L.upb_b.taps.size() * (sizeof(ConvTap) / sizeof(float)));
      }
    }
    // Winograd operands: 3x3 stride-1 layers whose K extent is whole 8-channel stages (the tuner decides per shape whether the family runs).
    // Not in fp16 plans: conv_wino_ok rejects p.f16, the operands would be re-packed every step for nothing.
    if (!cfg.conv_fp16 && L.kh == 3 && L.kw == 3 && L.stride == 1 && !L.up && !L.transposed && !L.col2im && L.cout >= 16 && L.H * L.W >= 512) {
      if (L.Kc % 8 == 0) {
        L.wino_np = conv_wino_np(L.cout);
        L.wino_off = off; off = align64(off + conv_wino_floats(L.Kc, L.cout));
      }
      if (trainable && L.KcT % 8 == 0 && L.cin >= 16) {
        L.winoT_np = conv_wino_np(L.cin);
        L.winoT_off = off; off = align64(off + conv_wino_floats(L.KcT, L.cin));
      }
    }
  };
  for (auto& L : P->pwc) place(L, false);
  for (auto& L : P->gen) place(L, true);
  for (auto& L : P->rec) place(L, true);
  P->packed_floats = off;

  // ======================= arena =======================
  for (auto& b : P->bufs) { b.off = off; off = align64(off + b.floats()); }
  P->scratch_floats = (size_t)24 << 20;  // 96 MiB per lane: split-K slabs
  P->wgrad_floats = (size_t)48 << 20;    // 192 MiB per lane: wgrad split partials
  for (int l = 0; l < Plan::NLANE; ++l) {
    P->scratch_off[l] = off;
    off = align64(off + P->scratch_floats);
    P->wgrad_off[l] = off;
    if (l < 4) off = align64(off + P->wgrad_floats);  // lanes 4/5 (prefetch) never run filter gradients
  }
  P->small_off = off;
  off = align64(off + 65536);
  P->ticket_off = off;  // split-K tickets, one table per lane (outside the region udet_autotune fills with probe data)
  off = align64(off + (size_t)Plan::NLANE * UDET_MAX_TICKETS);
  for (int net = 1; net <= 2; ++net) {
    P->seg_off[net] = off;
    off = align64(off + 4 * net_params(net).p.size() + 64);  // two long tables (offset, len) = 4 floats per entry
  }
  for (int net = 1; net <= 2; ++net) {  // PackJob tables: UDET_PACKJOB_CAP(layers) entries -- an ordinary layer has at most 6 jobs (forward, transposed,
                                        // taps-into-N | the two NN x2 sets, the two Winograd operands, bias), each of the up to four up-conv decoder
                                        // levels 8 more (four forward + four backward-data weight sets); plan_init_workspace checks the count
    const size_t nl = net == NET_GEN ? P->gen.size() : P->rec.size();
    P->jobs_off[net] = off;
    off = align64(off + UDET_PACKJOB_CAP(nl) * (sizeof(PackJob) / sizeof(float)) + 64);
  }
  P->arena_floats = off;
  // UDET_SERIAL=1 (read once, here; documented in include/udet.h next to udet_plan_set_concurrent): every lane collapses onto
  // the caller's stream
  if (const char* e = getenv("UDET_SERIAL")) P->concurrent = atoi(e) == 0;
  // Side streams, all at the default priority (measured: any priority split between the lanes costs 4-5 ms per step).
  // (candidates: the lanes are placed on them at first use, see Plan::Placement)
  for (int i = 0; i < Plan::NCAND; ++i) {
    hipStream_t c = nullptr;
    if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) == hipSuccess) P->cand.push_back(c);
  }
  if (P->cand.empty()) P->concurrent = false;
  // views into the small region (read by the host wrapper)
  struct { const char* n; int a, d; size_t o; } views[5] = {{"losses", 1, 8, 0}, {"loss_coef", B, 4, 16}, {"noise_flag", 1, 2, UDET_SMALL_NOISE},
                                                            {"fp16_overflow", 2, 2, UDET_SMALL_OVF}, {"loss_sums", B, 5, UDET_SMALL_SUMS}};
  for (auto& v : views) {
    const int id = P->add_buf(v.n, v.a, 1, 1, v.d);
    P->bufs[id].off = P->small_off + v.o;
  }
  return P;
}

}  // namespace udet
