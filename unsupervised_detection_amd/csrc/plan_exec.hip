// Plan execution: enqueues the kernels of one step on the caller's stream.
//
// Reference being replaced: the single sess.run of models/adversarial_learner.py:396 over the graph
// built by build_train_graph (:72-258) / build_test_graph (:450-523).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "conv_host.h"
#include "elementwise.h"
#include "lanes.h"
#include "plan.h"

namespace udet {

static const float BN_C = 0.99950037468777316f;  // 1/sqrt(1+1e-3): inference-mode BN with moving stats (0,1)

static const Layer* find_layer(const std::vector<Layer>& v, const std::string& name) {
  for (const auto& L : v)
    if (L.name == name) return &L;
  return nullptr;
}
static std::string S(const char* fmt, int a, int b = 0) {
  char buf[128];
  snprintf(buf, sizeof(buf), fmt, a, b);
  return buf;
}

// ------------------------------------------------------------ profiling ----
void prof_begin(Plan* P, int cat, double flops, double bytes, hipStream_t s, const char* name) {
  if (!P->profiling) return;
  Plan::ProfRec* r = new Plan::ProfRec();
  r->cat = cat; r->flops = flops; r->bytes = bytes; r->name = name ? name : "";
  r->sink.ev = r->kev; r->sink.n = 0; r->sink.cap = Plan::ProfRec::MAXK;
  (void)hipEventCreate(&r->a);
  (void)hipEventCreate(&r->b);
  (void)hipEventRecord(r->a, s);
  P->prof.push_back(r);
  g_launch_sink = &r->sink;  // kernels launched until prof_end carry their own start / stop events
}
void prof_end(Plan* P, hipStream_t s) {
  if (!P->profiling) return;
  g_launch_sink = nullptr;
  (void)hipEventRecord(P->prof.back()->b, s);
}
// algorithmic FLOPs of the convolution a layer stands for (2*MACs, unpadded channels)
static double layer_flops(const Layer& L, int N) {
  const int up = L.up ? 1 : 0;
  double oh, ow;
  if (L.transposed) { oh = 2.0 * L.H; ow = 2.0 * L.W; }
  else { oh = (double)(((L.H << up) + L.stride - 1) / L.stride); ow = (double)(((L.W << up) + L.stride - 1) / L.stride); }
  double taps = (double)L.kh * L.kw;
  if (L.transposed) taps /= 4.0;  // each output pixel of a k4 s2 transposed conv sees 2x2 taps
  return 2.0 * N * oh * ow * L.cout * L.cin * taps;
}
// algorithmic HBM bytes of the same convolution (SURVEY Appendix A's definition: input + output + weights + bias in fp32 if nothing is
// fused; up-sampling layers count their post-resize input, as the appendix does).  The backward-data (dY, dX, W) and backward-filter
// (X, dY, dW) passes of a layer move the same three tensors, so one figure serves all three directions.
static double layer_bytes(const Layer& L, int N) {
  const int up = L.up ? 1 : 0;
  const double ih = (double)(L.H << up), iw = (double)(L.W << up);
  double oh, ow;
  if (L.transposed) { oh = 2.0 * L.H; ow = 2.0 * L.W; }
  else { oh = (double)(((L.H << up) + L.stride - 1) / L.stride); ow = (double)(((L.W << up) + L.stride - 1) / L.stride); }
  return 4.0 * (N * ih * iw * L.cin + N * oh * ow * L.cout + (double)L.kh * L.kw * L.cin * L.cout + L.cout);
}

// timing-only ablation (experiment knob UDET_KNOB_SKIP: libudet_exp.so only, a constant false in the release build; results are wrong on purpose): bit 0 generator filter gradients, 1 recover
// filter gradients, 2 generator backward-data, 3 recover backward-data, 4 PWC-Net forward, 5 generator forward, 6 recover forward
static inline bool skip_launch(int kind, int net) {
  const long k = plan_knob(UDET_KNOB_SKIP);
  if (!k) return false;
  const int bit = kind == 2 ? (net == NET_GEN ? 0 : 1) : (kind == 1 ? (net == NET_GEN ? 2 : 3) : (net == NET_PWC ? 4 : (net == NET_GEN ? 5 : 6)));
  return (k >> bit) & 1;
}
// ---------------------------------------------------------------- lanes ----
// Lane 0 is the caller's stream; lanes 1..5 are placed on the plan's candidate streams by place_lanes (lanes that share a hardware
// queue are the same stream).  While profiling (per-kernel timing) or with UDET_SERIAL=1 every lane collapses onto the caller's
// stream, which reproduces the plain program order.
struct Lane {
  hipStream_t s;
  int slot;
};
// first use from `main`: find candidate streams that run concurrently with it and with each other (one probe each, ~0.1 ms; the
// device is synchronised once) and lay the lanes out on them -- see Plan::Placement
static const Plan::Placement& place_lanes(Plan* P, hipStream_t main) {
  if (P->placed >= 0 && P->placements[P->placed].main == main) return P->placements[P->placed];
  for (size_t k = 0; k < P->placements.size(); ++k)
    if (P->placements[k].main == main) { P->placed = (int)k; return P->placements[k]; }
  if (P->placements.size() >= 16) { P->placements.clear(); P->placed = -1; }  // (a caller that keeps making new streams: start over)
  (void)hipDeviceSynchronize();
  hipStream_t pick[3] = {nullptr, nullptr, nullptr};
  int np = 0;
  for (size_t k = 0; np < 3; ++k) {
    if (k == P->cand.size()) {  // every candidate so far shares a queue with the caller or a pick: draw another stream
      hipStream_t c = nullptr;
      if (k >= (size_t)Plan::MAXCAND || hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
      P->cand.push_back(c);
    }
    hipStream_t c = P->cand[k];
    bool ok = false;
    if (streams_concurrent(main, c, &ok) != UDET_OK || !ok) continue;
    for (int j = 0; j < np && ok; ++j) {
      bool cc = false;
      if (streams_concurrent(pick[j], c, &cc) != UDET_OK || !cc) ok = false;
    }
    if (ok) pick[np++] = c;
  }
  Plan::Placement pl;
  pl.main = main;
  pl.nqueues = np + 1;
  const int q1 = np > 0 ? 1 : 0, q3 = np > 1 ? (np > 2 ? 2 : 1) : q1, q4 = np > 2 ? 3 : (np > 1 ? 2 : q1);
  // measured alternatives (ms per step; this one 10.90): lane 3 on lane 1's queue 11.20, lane 2 on its own queue and 3 with 1 11.12,
  // lane 2 with 1 11.42, lane 5 with 3 10.96 / with 1 or 0 12.2, lanes 2 and 3 swapped 10.94 (profiles/r03_hop_bench.txt)
  const int queue[Plan::NLANE] = {0, q1, 0, q3, q4, q4};
  for (int i = 0; i < Plan::NLANE; ++i) {
    pl.queue[i] = queue[i];
    pl.lane[i] = queue[i] == 0 ? main : pick[queue[i] - 1];
  }
  P->placements.push_back(pl);
  P->placed = (int)P->placements.size() - 1;
  return P->placements[P->placed];
}
// the host pins the layout for `main`: `n` (0..3) streams it knows to sit on distinct hardware queues, distinct from main's -- no probe
int plan_pin_lanes(Plan* P, hipStream_t main, hipStream_t const* streams, int n) {
  if (n < 0 || n > 3) { set_error("plan_pin_lanes: 0..3 side streams"); return UDET_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    if (!streams[i] || streams[i] == main) { set_error("plan_pin_lanes: side streams must be non-null and differ from the caller's"); return UDET_ERR_ARG; }
    for (int j = 0; j < i; ++j)
      if (streams[j] == streams[i]) { set_error("plan_pin_lanes: duplicate side stream"); return UDET_ERR_ARG; }
  }
  Plan::Placement pl;
  pl.main = main;
  pl.nqueues = n + 1;
  const int q1 = n > 0 ? 1 : 0, q3 = n > 1 ? (n > 2 ? 2 : 1) : q1, q4 = n > 2 ? 3 : (n > 1 ? 2 : q1);
  const int queue[Plan::NLANE] = {0, q1, 0, q3, q4, q4};  // (the layout of place_lanes)
  for (int i = 0; i < Plan::NLANE; ++i) {
    pl.queue[i] = queue[i];
    pl.lane[i] = queue[i] == 0 ? main : streams[queue[i] - 1];
  }
  for (size_t k = 0; k < P->placements.size(); ++k)
    if (P->placements[k].main == main) { P->placements[k] = pl; P->placed = (int)k; return UDET_OK; }
  if (P->placements.size() >= 16) { P->placements.clear(); P->placed = -1; }
  P->placements.push_back(pl);
  P->placed = (int)P->placements.size() - 1;
  return UDET_OK;
}
static Lane lane_of(Plan* P, hipStream_t main, int i) {
  if (i == 0 || !P->concurrent || P->profiling) return Lane{main, 0};
  return Lane{place_lanes(P, main).lane[i], i};
}
int plan_lane_queues(Plan* P, hipStream_t s, int* queue) {
  if (!P->concurrent) {  // every lane is the caller's stream
    for (int i = 0; i < Plan::NLANE; ++i) queue[i] = 0;
    return 1;
  }
  const Plan::Placement& pl = place_lanes(P, s);
  for (int i = 0; i < Plan::NLANE; ++i) queue[i] = pl.queue[i];
  return pl.nqueues;
}
static hipEvent_t next_event(Plan* P) {
  std::vector<hipEvent_t>& pool = P->in_prefetch ? P->ev_pool_prefetch : P->ev_pool;
  size_t& nx = P->in_prefetch ? P->ev_next_prefetch : P->ev_next;
  if (nx == pool.size()) {
    hipEvent_t e;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    pool.push_back(e);
  }
  return pool[nx++];
}
// work enqueued on `to` after this call also waits for everything enqueued on `from` so far
static void order_after(Plan* P, const Lane& from, const Lane& to) {
  if (from.s == to.s) return;
  hipEvent_t e = next_event(P);
  (void)hipEventRecord(e, from.s);
  (void)hipStreamWaitEvent(to.s, e, 0);
}

// ------------------------------------------------------------- runners ----
// tf.image.resize_nearest_neighbor(x2) followed by a SAME 3x3 convolution (gen_deconv, convolution_utils.py:58-75) reads, for
// the output pixel (2q+p), up-sampled rows 2q+p-1 .. 2q+p+1 = low-resolution rows {q-1, q, q} (p = 0) or {q, q, q+1} (p = 1): per
// output parity class it is a 2x2 convolution over the low-resolution grid whose weights are sums of the 3x3 taps that read the
// same pixel -- 16 instead of 36 tap products per low-resolution pixel, exactly the arithmetic of the reference up to the order of
// the additions.  Tap t of class (py,px): row offset d(py,ty), column offset d(px,tx), weight matrix class*4 + t.
static inline int up_off(int parity, int t) { return parity == 0 ? (t == 0 ? -1 : 0) : (t == 0 ? 0 : 1); }
// forward: ncls = 4 launch on the low-resolution grid writing the four output sub-lattices
static void setup_up_fwd(ConvParams& p, int N, int H, int W) {
  p.N = N; p.H = H; p.W = W;
  p.OH = 2 * H; p.OW = 2 * W; p.OHq = H; p.OWq = W;
  p.osy = p.osx = 2; p.ooy = p.oox = 0; p.isy = p.isx = 1;
  p.ncls = 4; p.ntaps = 16;
  for (int c = 0; c < 4; ++c) {
    p.cls_tap[c] = 4 * c;
    for (int t = 0; t < 4; ++t) {
      ConvTap& tp = p.taps[4 * c + t];
      tp.dy = up_off(c >> 1, t >> 1); tp.dx = up_off(c & 1, t & 1); tp.widx = 4 * c + t;
    }
  }
  p.cls_tap[4] = 16;
}
// backward-data: dX[q] = sum_{class, tap} dU[2(q - d) + p] Weff[class][tap]^T -- a stride-2 walk over the full-resolution dU
static void setup_up_dgrad(ConvParams& p, int N, int H, int W) {
  p.N = N; p.H = 2 * H; p.W = 2 * W;
  p.OH = H; p.OW = W; p.OHq = H; p.OWq = W;
  p.osy = p.osx = 1; p.ooy = p.oox = 0; p.isy = p.isx = 2;
  p.ncls = 1; p.ntaps = 16;
  for (int c = 0; c < 4; ++c)
    for (int t = 0; t < 4; ++t) {
      ConvTap& tp = p.taps[4 * c + t];
      tp.dy = (c >> 1) - 2 * up_off(c >> 1, t >> 1); tp.dx = (c & 1) - 2 * up_off(c & 1, t & 1); tp.widx = 4 * c + t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Recover decoder, levels 1-4 (Layer::upb; backward-data form on the levels with enough source pixels, Layer::upb_bwd): legacy bilinear x2 (out[2j] = x[j], out[2j+1] = (x[j] + x[j+1]) / 2, clamped at the end;
// nets.py:91-110 through tf.image.resize_bilinear) followed by the 4x4 SAME convolution (padding 1 before, 2 after) is, per output
// parity class, a 3x3 convolution over the LOW-resolution source:
//   even rows 2j   read u[2j-1 .. 2j+2] = (x[j-1]+x[j])/2, x[j], (x[j]+x[j+1])/2, x[j+1]      -> x[j-1], x[j], x[j+1] through E
//   odd rows  2j+1 read u[2j .. 2j+3]   = x[j], (x[j]+x[j+1])/2, x[j+1], (x[j+1]+x[j+2])/2    -> x[j], x[j+1], x[j+2] through O
// (E / O: the merge matrices of pack modes 9 / 10, conv_host.hip) -- 9 instead of 16 tap products per output pixel and a source read
// at a quarter of the size.  The borders are exact, not approximated: the source carries a one-pixel ring x^ (upb_ring_kernel) that is
// the NEGATED edge at the top / left (the two halves of the merged even row 0 then cancel to the zero padding of the up-sampled grid)
// and the CLAMPED edge at the bottom / right (the resize's clamp for row 2H-3); only the LAST low-resolution row / column merges
// differently (rows 2H-2, 2H-1 see the clamp and the padding).  Both passes are ONE segmented launch (ConvParams::nseg, 16 segments,
// tables built by plan_build's build_upb_launches): forward = {interior, last row, last column, corner} x 4 parity classes, each
// segment with the weight set of its region; backward-data = the ringed gradient x^[R] collects dU[2R - p - 2a] over the quotient
// pixels q = R - p - a, with the last-row set where q is the last row -- a fixed mix per output row R >= H - 1, hence row groups
// {R < H - 1}, H - 1, H, H + 1 times the same for columns -- followed by the adjoint of the ring (upb_ring_fold), which replaces the
// resize adjoint.  The filter gradient keeps the up-sampled operand.
// ------------------------------------------------------------------------------------------------------------------------------
static void fill_segments(ConvParams& p, const Layer::SegLaunch& g, float* ws, int first = 0) {  // segments [first, nseg)
  p.nseg = g.nseg - first;
  memcpy(p.seg, g.seg + first, sizeof(ConvSeg) * p.nseg);
  memcpy(p.seg_tap, g.seg_tap + first, sizeof(int) * (p.nseg + 1));  // (offsets into the whole table)
  p.tap_tab = reinterpret_cast<const ConvTap*>(ws + g.tab_off);
}

// fp16 mode: gradient operands are multiplied by this power of two before the fp16 conversion (and the fp32 accumulators divided by it)
#define UDET_F16_GRAD_SCALE 4096.f
static void fill_common(Plan* P, ConvParams& p, float* ws, int slot) {
  p.partial = ws + P->scratch_off[slot];
  p.partial_cap = P->scratch_floats;
  p.zero16 = ws + P->small_off + 60000;  // never written after udet_plan_init's memset
  p.tickets = reinterpret_cast<int*>(ws + P->ticket_off) + (size_t)slot * UDET_MAX_TICKETS;
  p.f16 = P->cfg.conv_fp16;
  p.f16_xscale = 1.f;  // (backward-data launches: UDET_F16_GRAD_SCALE, their x operand is a gradient)
}

static int run_fwd_upb(Plan* P, const Layer& L, int N, float* ws, const Lane& ln) {
  hipStream_t s = ln.s;
  const Buf &bs = P->buf(L.src), &bp = P->buf(L.xhat), &by = P->buf(L.y);
  const int h = bs.h, w = bs.w;
  prof_begin(P, PROF_CONV_FWD, layer_flops(L, N) * 9.0 / 16.0, layer_bytes(L, N), s, L.name.c_str());
  UDET_TRY(launch_upb_ring(ws + bs.off, bs.ld, N, h, w, ws + bp.off, s));
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.N = N; p.H = h + 2; p.W = w + 2;
  p.OH = 2 * h; p.OW = 2 * w;
  p.osy = p.osx = 2; p.isy = p.isx = 1;
  p.x = ws + bp.off; p.ldx = bp.ld; p.x_coff = 0;
  p.wp = ws + L.wupb_off; p.Kc = L.Kc; p.ldw = L.ldw; p.bias = ws + L.bias_f_off;
  p.y = ws + by.off; p.ldy = by.ld; p.y_coff = L.y_coff; p.Cout = L.cout;
  p.act = L.act; p.alpha = L.alpha;
  fill_common(P, p, ws, ln.slot);
  if (L.upb_split) {
    // <= 16 output channels: the tile-resident kernel (conv_tile.hip) is the fast family and takes no segments -- the interior as an
    // ordinary four-class launch (segments 0-3 of the table: weight set 0), the twelve border segments in a second, short launch
    ConvParams q = p;
    q.OHq = h - 1; q.OWq = w - 1;
    q.ncls = 4;
    for (int c = 0; c < 4; ++c) {
      q.cls_tap[c] = q.ntaps;
      for (int t = L.upb_f.seg_tap[c]; t < L.upb_f.seg_tap[c + 1]; ++t) q.taps[q.ntaps++] = L.upb_f.taps[t];
    }
    q.cls_tap[4] = q.ntaps;
    UDET_TRY(launch_conv(q, s));
    fill_segments(p, L.upb_f, ws, 4);
  } else {
    fill_segments(p, L.upb_f, ws);
  }
  UDET_TRY(launch_conv(p, s));
  prof_end(P, s);
  return UDET_OK;
}

// gradient w.r.t. the low-resolution source of an upb level: dsrc (written) from dU (`du` buffer, channels [0, KcT))
static int run_dgrad_upb(Plan* P, const Layer& L, int N, int du, int dxhat, int dsrc, float* ws, const Lane& ln) {
  hipStream_t s = ln.s;
  if (skip_launch(1, L.net)) return UDET_OK;
  const Buf &bu = P->buf(du), &bp = P->buf(dxhat), &bd = P->buf(dsrc);
  const int h = bd.h, w = bd.w;
  prof_begin(P, PROF_CONV_DGRAD, layer_flops(L, N) * 9.0 / 16.0, layer_bytes(L, N), s, L.name.c_str());
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.N = N; p.H = 2 * h; p.W = 2 * w;
  p.OH = h + 2; p.OW = w + 2;
  p.osy = p.osx = 1; p.isy = p.isx = 2;
  fill_segments(p, L.upb_b, ws);
  p.x = ws + bu.off; p.ldx = bu.ld; p.x_coff = L.y_coff;
  p.wp = ws + L.wupbT_off; p.Kc = L.KcT; p.ldw = L.ldwT; p.kreal = L.cout;
  p.y = ws + bp.off; p.ldy = bp.ld; p.y_coff = 0; p.Cout = L.cin;
  fill_common(P, p, ws, ln.slot);
  p.f16_xscale = UDET_F16_GRAD_SCALE;
  UDET_TRY(launch_conv(p, s));
  UDET_TRY(launch_upb_ring_fold(ws + bp.off, bp.ld, N, h, w, ws + bd.off, s));
  prof_end(P, s);
  return UDET_OK;
}

static int run_fwd(Plan* P, const Layer& L, int N, float* ws, const Lane& ln, size_t x_extra = 0, size_t y_extra = 0) {
  hipStream_t s = ln.s;
  if (skip_launch(0, L.net)) return UDET_OK;
  if (L.upb && x_extra == 0 && y_extra == 0) return run_fwd_upb(P, L, N, ws, ln);
  if (L.upb) {
    // BUFFER CONTRACT (round 5 on): the forward never builds the up-sampled input `rec.r{k+1}` of an up-conv level -- Layer::x of such a
    // layer holds whatever the last recover-loss backward left there (rec_backward rebuilds it for the filter gradient only).  The
    // generic path below would silently convolve that stale tensor: refuse.
    set_error("run_fwd(%s): up-conv levels run on their ringed low-resolution source only (rec.r* is not built by the forward)", L.name.c_str());
    return UDET_ERR_UNSUPPORTED;
  }
  const Buf &bx = P->buf(L.x), &by = P->buf(L.y);
  const int ncls = L.transposed ? conv_dgrad_classes(2, 2 * L.H, 2 * L.W) : 1;
  // (the measurement pass books the multiply-adds the launch executes: 16 of the reference's 36 tap products per low-resolution
  // pixel for the up-sampling layers)
  prof_begin(P, PROF_CONV_FWD, layer_flops(L, N) * ((L.up && L.wu_off) ? 4.0 / 9.0 : 1.0), layer_bytes(L, N), s, L.name.c_str());
  // 2-channel heads: the direct kernel (conv_thin.hip) through the ordinary launch below wherever it is eligible; otherwise
  // (fp16 mode) the GEMM + gather formulation
  bool direct_head = false;
  if (L.col2im && !P->cfg.conv_fp16 && L.Kc <= 64) {  // (deep inputs: the 1x1 GEMM already reads them at the memory rate)
    ConvParams q;
    memset(&q, 0, sizeof(q));
    if (L.transposed) (void)conv_setup_dgrad(q, 0, N, 2 * L.H, 2 * L.W, L.kh, L.kw, 2, 1);
    else conv_setup_fwd(q, N, L.H, L.W, L.kh, L.kw, L.stride, L.dil);
    q.Kc = L.Kc; q.ldw = L.ldw; q.Cout = L.cout;
    if (q.ncls != 4) { q.ncls = 1; q.cls_tap[0] = 0; q.cls_tap[1] = q.ntaps; }
    direct_head = conv_thin_n_ok(q);
  }
  if (L.col2im && !direct_head && (!L.transposed || ncls == 1) && x_extra == 0 && y_extra == 0) {
    // 2-channel head: the (tap, output channel) pairs become the N axis of ONE 1x1 GEMM over the deep channel axis
    // (every input byte read once, no 16x padding of the MFMA columns per tap), then a gather-sum over the taps
    const Buf& bz = P->buf(L.zbuf);
    ConvParams p;
    memset(&p, 0, sizeof(p));
    conv_setup_fwd(p, N, L.H, L.W, 1, 1, 1, 1);
    p.x = ws + bx.off; p.ldx = bx.ld; p.x_coff = L.x_coff;
    p.wp = ws + L.wz_off; p.Kc = L.Kc; p.ldw = L.ldz;
    p.y = ws + bz.off; p.ldy = bz.ld; p.y_coff = 0; p.Cout = L.kh * L.kw * L.cout;
    fill_common(P, p, ws, ln.slot);
    UDET_TRY(launch_conv(p, s));
    ConvParams g;
    memset(&g, 0, sizeof(g));
    if (L.transposed) conv_setup_dgrad(g, 0, N, 2 * L.H, 2 * L.W, L.kh, L.kw, 2, 1);
    else conv_setup_fwd(g, N, L.H, L.W, L.kh, L.kw, 1, 1);
    g.H = L.H; g.W = L.W;  // Z lives on the input grid
    g.bias = ws + L.bias_f_off;
    g.y = ws + by.off; g.ldy = by.ld; g.y_coff = L.y_coff; g.Cout = L.cout;
    UDET_TRY(launch_tap_gather(g, ws + bz.off, bz.ld, s));
    prof_end(P, s);
    return UDET_OK;
  }
  for (int cls = 0; cls < ncls; ++cls) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    if (L.transposed) {
      if (!conv_setup_dgrad(p, cls, N, 2 * L.H, 2 * L.W, L.kh, L.kw, 2, 1)) continue;
    } else if (L.up && L.wu_off) {
      setup_up_fwd(p, N, L.H, L.W);
    } else {
      conv_setup_fwd(p, N, L.H << (L.up ? 1 : 0), L.W << (L.up ? 1 : 0), L.kh, L.kw, L.stride, L.dil);
      p.up_shift = L.up ? 1 : 0;
    }
    p.x = ws + bx.off + x_extra; p.ldx = bx.ld; p.x_coff = L.x_coff;
    p.wp = ws + ((L.up && L.wu_off) ? L.wu_off : L.wp_off); p.Kc = L.Kc; p.ldw = L.ldw; p.bias = ws + L.bias_f_off;
    p.y = ws + by.off + y_extra; p.ldy = by.ld; p.y_coff = L.y_coff; p.Cout = L.cout;
    p.act = L.act; p.alpha = L.alpha;
    if (L.res >= 0) { p.res = ws + P->buf(L.res).off; p.ldres = P->buf(L.res).ld; p.res_coff = L.res_coff; }
    if (L.y2 >= 0) { p.y2 = ws + P->buf(L.y2).off; p.ldy2 = P->buf(L.y2).ld; p.y2_coff = 0; }
    if (L.wino_off) { p.wino_u = ws + L.wino_off; p.wino_np = L.wino_np; }
    fill_common(P, p, ws, ln.slot);
    UDET_TRY(launch_conv(p, s));
    if (P->profiling && (conv_last_config() & 0xff) == 9) P->prof.back()->mfma_scale = 4.0 / 9.0;
  }
  prof_end(P, s);
  return UDET_OK;
}

// dU emission of a backward-data launch: output channels [c0,c1) of the result (relative to dx_coff) are also written,
// multiplied by act'(activation `abuf`), into `ubuf` (same layout as the dx buffer)
struct Emit {
  int ubuf = -1, abuf = -1, c0 = 0, c1 = 0, act = ACT_NONE;
  float alpha = 0.f;
};

// gradient w.r.t. the layer input: dX(dx buffer) (=|+=) conv_T(dU) [+ res].  dy_is_du: `dy` already holds
// dU = dY * act'(saved output) (emitted by the launch that finalised dY); otherwise act' is applied on load.
static int run_dgrad(Plan* P, const Layer& L, int N, int dy, bool dy_is_du, int dx, int dx_coff, int accumulate, int res,
                     const Emit& em, float* ws, const Lane& ln) {
  hipStream_t s = ln.s;
  if (skip_launch(1, L.net)) return UDET_OK;
  const int act_buf = L.y2 >= 0 ? L.y2 : L.y;
  const Buf &bdy = P->buf(dy), &bdx = P->buf(dx), &ba = P->buf(act_buf);
  if (ba.ld != bdy.ld) {
    set_error("dgrad(%s): gradient/activation layout mismatch (%d vs %d)", L.name.c_str(), bdy.ld, ba.ld);
    return UDET_ERR_SHAPE;
  }
  const int up = L.up ? 1 : 0;
  prof_begin(P, PROF_CONV_DGRAD, layer_flops(L, N) * ((L.up && L.wuT_off) ? 4.0 / 9.0 : 1.0), layer_bytes(L, N), s, L.name.c_str());
  const bool upeff = L.up && L.wuT_off;  // gradient w.r.t. the LOW-resolution input in one launch (no up-sampled gradient, no pooling)
  for (int cls = 0; cls < (upeff ? 1 : conv_dgrad_classes(L.stride, L.H << up, L.W << up)); ++cls) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    if (upeff) setup_up_dgrad(p, N, L.H, L.W);
    else if (!conv_setup_dgrad(p, cls, N, L.H << up, L.W << up, L.kh, L.kw, L.stride, L.dil)) continue;
    p.x = ws + bdy.off; p.ldx = bdy.ld; p.x_coff = L.y_coff;
    if (L.act != ACT_NONE && !dy_is_du) { p.xa = ws + ba.off; p.xact = L.act; p.xalpha = L.alpha; }
    p.wp = ws + (upeff ? L.wuT_off : L.wpT_off); p.Kc = L.KcT; p.ldw = L.ldwT; p.kreal = L.cout;
    p.y = ws + bdx.off; p.ldy = bdx.ld; p.y_coff = dx_coff; p.Cout = L.cin;
    p.accumulate = accumulate;
    if (res >= 0) { p.res = ws + P->buf(res).off; p.ldres = P->buf(res).ld; p.res_coff = 0; }
    if (em.ubuf >= 0) {
      const Buf &bu = P->buf(em.ubuf), &bua = P->buf(em.abuf);
      p.uo = ws + bu.off; p.ldu = bu.ld; p.u_coff = dx_coff;
      p.ua = ws + bua.off; p.ldua = bua.ld; p.ua_coff = dx_coff;
      p.uact = em.act; p.ualpha = em.alpha; p.u_c0 = em.c0; p.u_c1 = em.c1;
    }
    if (L.winoT_off && !upeff) { p.wino_u = ws + L.winoT_off; p.wino_np = L.winoT_np; }
    fill_common(P, p, ws, ln.slot);
    p.f16_xscale = UDET_F16_GRAD_SCALE;  // the x operand is a gradient
    UDET_TRY(launch_conv(p, s));
    if (P->profiling && (conv_last_config() & 0xff) == 9) P->prof.back()->mfma_scale = 4.0 / 9.0;
  }
  prof_end(P, s);
  return UDET_OK;
}

static int run_wgrad(Plan* P, const Layer& L, int N, int dy, bool dy_is_du, const float* w_flat, float* g_flat, float* ws, const Lane& ln) {
  hipStream_t s = ln.s;
  if (skip_launch(2, L.net)) return UDET_OK;
  const NetParams& np = net_params(L.net);
  const int act_buf = L.y2 >= 0 ? L.y2 : L.y;
  const Buf &bx = P->buf(L.x), &bdy = P->buf(dy), &ba = P->buf(act_buf);
  const int up = L.up ? 1 : 0;
  ConvParams g;
  memset(&g, 0, sizeof(g));
  conv_setup_fwd(g, N, L.H << up, L.W << up, L.kh, L.kw, L.stride, L.dil);
  WgradParams q;
  memset(&q, 0, sizeof(q));
  q.x = ws + bx.off; q.ldx = bx.ld; q.x_coff = L.x_coff;
  q.N = N; q.H = L.H << up; q.W = L.W << up; q.up_shift = up; q.Cin = L.cin;
  q.dy = ws + bdy.off; q.ldy = bdy.ld; q.y_coff = L.y_coff; q.Cout = L.cout;
  if (L.act != ACT_NONE && !dy_is_du) { q.ya = ws + ba.off; q.yact = L.act; q.yalpha = L.alpha; }
  q.OH = g.OH; q.OW = g.OW; q.isy = q.isx = L.stride;
  q.ntaps = g.ntaps;
  memcpy(q.taps, g.taps, sizeof(g.taps));
  q.dw = g_flat + np.p[L.w_idx].offset;
  q.db = g_flat + np.p[L.b_idx].offset;
  q.partial = ws + P->wgrad_off[ln.slot];
  q.zero16 = ws + P->small_off + 60000;
  q.partial_floats = P->wgrad_floats;
  q.f16 = P->cfg.conv_fp16;
  q.f16_yscale = UDET_F16_GRAD_SCALE;
  if (L.g_idx >= 0) {
    q.w = w_flat + np.p[L.w_idx].offset;
    q.b = w_flat + np.p[L.b_idx].offset;
    q.gamma = w_flat + np.p[L.g_idx].offset;
    q.dgamma = g_flat + np.p[L.g_idx].offset;
    q.dbeta = g_flat + np.p[L.be_idx].offset;
    q.bn_c = BN_C;
  }
  if (L.up && L.wu_off && L.g_idx >= 0 && dy_is_du) {
    // NN x2 + 3x3 as four 2x2 convolutions (setup_up_fwd): dWeff[class][tap] = sum_q X[q + d]^T dU[2q + p] over the low-resolution
    // pixels (16 instead of 36 tap products each), then dW[ky][kx] = sum over the classes of the effective tap that contains it,
    // then the BN finalisation
    ConvParams u;
    memset(&u, 0, sizeof(u));
    setup_up_fwd(u, N, L.H, L.W);
    q.H = L.H; q.W = L.W; q.up_shift = 0;
    q.OH = L.H; q.OW = L.W; q.isy = q.isx = 1;
    q.ycls = 1; q.OHf = 2 * L.H; q.OWf = 2 * L.W;
    q.ntaps = 16;
    memcpy(q.taps, u.taps, sizeof(u.taps));
    const size_t reserve = (size_t)16 * L.cin * L.cout + 64 * (size_t)L.cout + 1024;
    float* deff = ws + P->wgrad_off[ln.slot] + (P->wgrad_floats - reserve);  // [16][Cin][Cout] + BN-dot partials
    float* pd = deff + (size_t)16 * L.cin * L.cout;
    q.partial_floats = P->wgrad_floats - reserve;
    float* dw = q.dw;
    float* db = q.db;
    q.dw = deff;  // (q.db: the kernel's own column sums of dU, four class partials per split)
    const float *w_ = q.w, *b_ = q.b, *gamma_ = q.gamma;
    float *dgamma_ = q.dgamma, *dbeta_ = q.dbeta;
    q.w = q.b = q.gamma = nullptr; q.dgamma = q.dbeta = nullptr;
    prof_begin(P, PROF_CONV_WGRAD, layer_flops(L, N) * 4.0 / 9.0, layer_bytes(L, N), s, L.name.c_str());
    int rc = launch_wgrad_T(q, 16, s);
    if (rc == UDET_OK) rc = launch_wgrad_up_combine(deff, dw, L.cin, L.cout, s);
    if (rc == UDET_OK) rc = launch_bn_finalize(dw, 9, L.cin, L.cout, w_, b_, gamma_, BN_C, pd, db, dgamma_, dbeta_, s);
    prof_end(P, s);
    return rc;
  }
  prof_begin(P, PROF_CONV_WGRAD, layer_flops(L, N), layer_bytes(L, N), s, L.name.c_str());
  const int rc = launch_wgrad_T(q, L.kh * L.kw, s);
  if (P->profiling && (wgrad_last_config() >> 20) == 3) P->prof.back()->mfma_scale = 4.0 / 9.0;  // (Winograd-domain family: 16 of 36 products)
  prof_end(P, s);
  return rc;
}

// ---- pair launches (launch_conv_pair, conv_igemm.hip): the recover net's two encoders (nets.py:57-75) -----------------------------
// aconv_k and bconv_k have the same geometry per level, separate weights and (shared image encoder) different batch sizes; their
// forward launches and their backward-data launches are independent of each other.  One launch per level carries both.
// (experiment knob UDET_KNOB_NO_PAIRS, libudet_exp.so only: bit 0 forward pairs off, bit 1 backward-data pairs off)
static bool pairs_on(const Plan* P, int dir = 0) { return !P->cfg.conv_fp16 && !((plan_knob(UDET_KNOB_NO_PAIRS) >> dir) & 1); }
// the parameter block run_fwd launches for a PLAIN forward layer (no transposed / up-sampling / head / up-conv form)
static void fwd_params(Plan* P, const Layer& L, int N, float* ws, int slot, ConvParams& p) {
  const Buf &bx = P->buf(L.x), &by = P->buf(L.y);
  memset(&p, 0, sizeof(p));
  conv_setup_fwd(p, N, L.H, L.W, L.kh, L.kw, L.stride, L.dil);
  p.x = ws + bx.off; p.ldx = bx.ld; p.x_coff = L.x_coff;
  p.wp = ws + L.wp_off; p.Kc = L.Kc; p.ldw = L.ldw; p.bias = ws + L.bias_f_off;
  p.y = ws + by.off; p.ldy = by.ld; p.y_coff = L.y_coff; p.Cout = L.cout;
  p.act = L.act; p.alpha = L.alpha;
  if (L.res >= 0) { p.res = ws + P->buf(L.res).off; p.ldres = P->buf(L.res).ld; p.res_coff = L.res_coff; }
  if (L.y2 >= 0) { p.y2 = ws + P->buf(L.y2).off; p.ldy2 = P->buf(L.y2).ld; p.y2_coff = 0; }
  fill_common(P, p, ws, slot);
}
static bool plain_fwd_layer(const Layer& L) { return !L.transposed && !L.up && !L.upb && !L.col2im; }
static int run_fwd_pair(Plan* P, const Layer& La, int Na, const Layer& Lb, int Nb, float* ws, const Lane& ln) {
  if (!pairs_on(P) || !plain_fwd_layer(La) || !plain_fwd_layer(Lb)) {
    UDET_TRY(run_fwd(P, La, Na, ws, ln));
    return run_fwd(P, Lb, Nb, ws, ln);
  }
  if (skip_launch(0, La.net)) return UDET_OK;
  ConvParams a, b;
  fwd_params(P, La, Na, ws, ln.slot, a);
  fwd_params(P, Lb, Nb, ws, ln.slot, b);
  prof_begin(P, PROF_CONV_FWD, layer_flops(La, Na) + layer_flops(Lb, Nb), layer_bytes(La, Na) + layer_bytes(Lb, Nb), ln.s, (La.name + "+" + Lb.name).c_str());
  const int rc = launch_conv_pair(a, b, ln.s);
  prof_end(P, ln.s);
  return rc;
}
// the parameter block of a layer's backward-data launch when that is ONE launch on a materialised dU (run_dgrad's common case)
static bool dgrad_params(Plan* P, const Layer& L, int N, int dy, int dx, int dx_coff, int accumulate, const Emit& em, float* ws, int slot, ConvParams& p) {
  if (L.up || L.upb || L.transposed || conv_dgrad_classes(L.stride, L.H, L.W) != 1) return false;
  const Buf &bdy = P->buf(dy), &bdx = P->buf(dx);
  memset(&p, 0, sizeof(p));
  if (!conv_setup_dgrad(p, 0, N, L.H, L.W, L.kh, L.kw, L.stride, L.dil)) return false;
  p.x = ws + bdy.off; p.ldx = bdy.ld; p.x_coff = L.y_coff;
  p.wp = ws + L.wpT_off; p.Kc = L.KcT; p.ldw = L.ldwT; p.kreal = L.cout;
  p.y = ws + bdx.off; p.ldy = bdx.ld; p.y_coff = dx_coff; p.Cout = L.cin;
  p.accumulate = accumulate;
  if (em.ubuf >= 0) {
    const Buf &bu = P->buf(em.ubuf), &bua = P->buf(em.abuf);
    p.uo = ws + bu.off; p.ldu = bu.ld; p.u_coff = dx_coff;
    p.ua = ws + bua.off; p.ldua = bua.ld; p.ua_coff = dx_coff;
    p.uact = em.act; p.ualpha = em.alpha; p.u_c0 = em.c0; p.u_c1 = em.c1;
  }
  if (L.winoT_off) { p.wino_u = ws + L.winoT_off; p.wino_np = L.winoT_np; }
  fill_common(P, p, ws, slot);
  p.f16_xscale = UDET_F16_GRAD_SCALE;
  return true;
}
struct DgradJob {
  const Layer* L;
  int N, dy, dx, dx_coff, accumulate;
  Emit em;
};
// both jobs read a materialised dU (dy_is_du) and carry no residual operand
static int run_dgrad_pair(Plan* P, const DgradJob& ja, const DgradJob& jb, float* ws, const Lane& ln) {
  ConvParams a, b;
  if (!pairs_on(P, 1) || !dgrad_params(P, *ja.L, ja.N, ja.dy, ja.dx, ja.dx_coff, ja.accumulate, ja.em, ws, ln.slot, a) ||
      !dgrad_params(P, *jb.L, jb.N, jb.dy, jb.dx, jb.dx_coff, jb.accumulate, jb.em, ws, ln.slot, b)) {
    UDET_TRY(run_dgrad(P, *ja.L, ja.N, ja.dy, true, ja.dx, ja.dx_coff, ja.accumulate, -1, ja.em, ws, ln));
    return run_dgrad(P, *jb.L, jb.N, jb.dy, true, jb.dx, jb.dx_coff, jb.accumulate, -1, jb.em, ws, ln);
  }
  if (skip_launch(1, ja.L->net)) return UDET_OK;
  a.wino_u = b.wino_u = nullptr;  // (pair launches are implicit-GEMM launches)
  prof_begin(P, PROF_CONV_DGRAD, layer_flops(*ja.L, ja.N) + layer_flops(*jb.L, jb.N), layer_bytes(*ja.L, ja.N) + layer_bytes(*jb.L, jb.N), ln.s,
             (ja.L->name + "+" + jb.L->name).c_str());
  const int rc = launch_conv_pair(a, b, ln.s);
  prof_end(P, ln.s);
  return rc;
}

// ------------------------------------------------------- init / packing ----
int plan_init_workspace(Plan* P, float* ws, hipStream_t s) {
  UDET_HIP(hipMemsetAsync(ws, 0, P->arena_floats * sizeof(float), s));
  for (int net = 1; net <= 2; ++net) {
    const NetParams& np = net_params(net);
    std::vector<long> tab(2 * np.p.size());
    for (size_t i = 0; i < np.p.size(); ++i) {
      tab[i] = (long)np.p[i].offset;
      tab[np.p.size() + i] = (long)np.p[i].count;
    }
    UDET_HIP(hipMemcpyAsync(ws + P->seg_off[net], tab.data(), tab.size() * sizeof(long), hipMemcpyHostToDevice, s));
    UDET_HIP(hipStreamSynchronize(s));  // `tab` is a host temporary
  }
  // weight re-layout job tables of the trainable networks (one launch per network and step)
  for (int net = 1; net <= 2; ++net) {
    const NetParams& np = net_params(net);
    const std::vector<Layer>& layers = net == NET_GEN ? P->gen : P->rec;
    std::vector<PackJob> jobs;
    for (const auto& L : layers) {
      const int T = L.kh * L.kw;
      const long goff = L.g_idx >= 0 ? (long)np.p[L.g_idx].offset : -1, beoff = L.be_idx >= 0 ? (long)np.p[L.be_idx].offset : -1;
      PackJob j;
      memset(&j, 0, sizeof(j));
      j.src_off = (long)np.p[L.w_idx].offset; j.gamma_off = goff; j.beta_off = beoff;
      j.T = T;
      // forward operand (conv2d_transpose layers store [t][cout][cin] and never occur in the trainable nets)
      j.dst_off = (long)L.wp_off; j.R = L.cin; j.C = L.cout; j.Kc = L.Kc; j.ldw = L.ldw; j.k_split = L.k_split; j.k_gap = L.k_gap;
      j.mode = 0; j.total = (long)T * L.Kc * L.ldw;
      jobs.push_back(j);
      // transposed operand of the backward-data pass
      j.dst_off = (long)L.wpT_off; j.Kc = L.KcT; j.ldw = L.ldwT; j.k_split = L.KcT; j.k_gap = 0;
      j.mode = 1; j.total = (long)T * L.KcT * L.ldwT;
      jobs.push_back(j);
      if (L.col2im) {  // taps folded into the N axis (GEMM + gather heads)
        j.dst_off = (long)L.wz_off; j.R = L.cin; j.C = L.cout; j.Kc = L.Kc; j.ldw = L.ldz; j.k_split = L.k_split; j.k_gap = L.k_gap;
        j.mode = 3; j.total = (long)L.Kc * L.ldz;
        jobs.push_back(j);
      }
      if (L.up) {  // NN x2 + 3x3 as four 2x2 convolutions on the low-resolution grid (run_fwd / run_dgrad)
        j.T = 16;
        j.dst_off = (long)L.wu_off; j.R = L.cin; j.C = L.cout; j.Kc = L.Kc; j.ldw = L.ldw; j.k_split = L.Kc; j.k_gap = 0;
        j.mode = 5; j.total = (long)16 * L.Kc * L.ldw;
        jobs.push_back(j);
        j.dst_off = (long)L.wuT_off; j.Kc = L.KcT; j.ldw = L.ldwT; j.k_split = L.KcT; j.k_gap = 0;
        j.mode = 6; j.total = (long)16 * L.KcT * L.ldwT;
        jobs.push_back(j);
        j.T = T;
      }
      if (L.upb)  // up-conv algebra of the recover decoder: four forward and four backward-data weight sets (pack modes 9 / 10), set
        for (int r = 0; r < 4; ++r) {  // r = (last row) + 2 (last column) -> job variant (row, column) in {interior, last}
          j.T = 36; j.gamma_off = -1;
          j.dst_off = (long)(L.wupb_off + (size_t)r * 36 * L.Kc * L.ldw); j.R = L.cin; j.C = L.cout; j.Kc = L.Kc; j.ldw = L.ldw; j.k_split = L.k_split; j.k_gap = L.k_gap;
          j.mode = 9; j.beta_off = (long)((r & 1) * 3 + (r >> 1)); j.total = (long)L.Kc * L.ldw;  // (one work item per (k, n))
          jobs.push_back(j);
          if (L.upb_bwd) {  // (a level that keeps the up-sampled form for backward-data never reads these)
            j.dst_off = (long)(L.wupbT_off + (size_t)r * 36 * L.KcT * L.ldwT); j.Kc = L.KcT; j.ldw = L.ldwT; j.k_split = L.KcT; j.k_gap = 0;
            j.mode = 10; j.total = (long)L.KcT * L.ldwT;
            jobs.push_back(j);
          }
          j.T = T; j.beta_off = beoff;
        }
      if (L.wino_off) {  // Winograd operand of the forward pass: K = input channels with the slab's gap map, N = output channels
        j.dst_off = (long)L.wino_off; j.R = L.cin; j.C = L.cout; j.Kc = L.Kc; j.ldw = L.wino_np; j.k_split = L.k_split; j.k_gap = L.k_gap;
        j.mode = 7; j.total = (long)(L.Kc / 8) * 2 * L.wino_np;  // work items (wino_pack.h)
        jobs.push_back(j);
      }
      if (L.winoT_off) {  // ... of the backward-data pass: K = output channels, N = input channels, taps mirrored
        j.dst_off = (long)L.winoT_off; j.R = L.cin; j.C = L.cout; j.Kc = L.KcT; j.ldw = L.winoT_np; j.k_split = L.KcT; j.k_gap = 0;
        j.mode = 8; j.total = (long)(L.KcT / 8) * 2 * L.winoT_np;
        jobs.push_back(j);
      }
      // bias (BN-folded for the generator)
      j.src_off = (long)np.p[L.b_idx].offset; j.dst_off = (long)L.bias_f_off; j.mode = 2; j.total = L.cout;
      jobs.push_back(j);
    }
    if (jobs.size() > UDET_PACKJOB_CAP(layers.size())) {  // (the arena reserves exactly that many entries: plan_build)
      set_error("plan_init: %zu weight re-layout jobs for %zu layers exceed the table's %zu entries", jobs.size(), layers.size(),
                (size_t)UDET_PACKJOB_CAP(layers.size()));
      return UDET_ERR_ARG;
    }
    P->njobs[net] = (int)jobs.size();
    UDET_HIP(hipMemcpyAsync(ws + P->jobs_off[net], jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, s));
    UDET_HIP(hipStreamSynchronize(s));  // `jobs` is a host temporary
  }
  for (const auto& L : P->rec)  // tap tables of the segmented launches (recover decoder's up-conv algebra)
    if (L.upb)
      for (const Layer::SegLaunch* g : {&L.upb_f, &L.upb_b})
        if (g == &L.upb_f || L.upb_bwd)
          UDET_HIP(hipMemcpyAsync(ws + g->tab_off, g->taps.data(), g->taps.size() * sizeof(ConvTap), hipMemcpyHostToDevice, s));
  UDET_HIP(hipStreamSynchronize(s));
  P->pwc_packed = false;
  return UDET_OK;
}

static int pack_layer(const Layer& L, const float* w_flat, float* ws, const float* scale, bool trainable, hipStream_t s) {
  const NetParams& np = net_params(L.net);
  const float* w = w_flat + np.p[L.w_idx].offset;
  const int T = L.kh * L.kw;
  if (L.transposed)  // weights are [t][cout][cin]
    UDET_TRY(launch_pack_weights(w, ws + L.wp_off, T, L.cout, L.cin, L.Kc, L.ldw, L.k_split, L.k_gap, 1, nullptr, s));
  else
    UDET_TRY(launch_pack_weights(w, ws + L.wp_off, T, L.cin, L.cout, L.Kc, L.ldw, L.k_split, L.k_gap, 0, scale, s));
  if (trainable)
    UDET_TRY(launch_pack_weights(w, ws + L.wpT_off, T, L.cin, L.cout, L.KcT, L.ldwT, L.KcT, 0, 1, scale, s));
  if (L.col2im)
    UDET_TRY(launch_pack_taps_into_n(w, ws + L.wz_off, T, L.cin, L.cout, L.Kc, L.ldz, L.k_split, L.k_gap, L.transposed ? 1 : 0, s));
  if (L.wino_off && !trainable)  // (the trainable nets build theirs in the per-step job table, BN folded)
    UDET_TRY(launch_wino_pack(w, ws + L.wino_off, L.cin, L.cout, L.Kc, L.wino_np, L.k_split, L.k_gap, 0, s));
  return UDET_OK;
}

int plan_pack_pwc(Plan* P, const float* w, float* ws, hipStream_t s) {
  const NetParams& np = net_params(NET_PWC);
  for (const auto& L : P->pwc) {
    UDET_TRY(pack_layer(L, w, ws, nullptr, false, s));
    UDET_TRY(launch_copy_channels(w + np.p[L.b_idx].offset, L.cout, 0, ws + L.bias_f_off, L.cout, 0, 1, L.cout, 1.f, 0.f, s));
  }
  P->pwc_packed = true;
  return UDET_OK;
}

int plan_pack_trainable(Plan* P, const float* w_gen, const float* w_rec, float* ws, hipStream_t s) {
  if (w_gen)
    UDET_TRY(launch_pack_jobs(reinterpret_cast<const PackJob*>(ws + P->jobs_off[NET_GEN]), P->njobs[NET_GEN], w_gen, ws, BN_C, s));
  if (w_rec)
    UDET_TRY(launch_pack_jobs(reinterpret_cast<const PackJob*>(ws + P->jobs_off[NET_REC]), P->njobs[NET_REC], w_rec, ws, BN_C, s));
  return UDET_OK;
}

// ------------------------------------------------------------ PWC-Net ----
static const int PWC_CH[7] = {0, 16, 32, 64, 96, 128, 196};

// L0: the lane of the pyramid / estimator / context chain; LH: the 2-channel heads, beside the context network
static int pwc_forward_on(Plan* P, const float* img1, const float* img2, float* ws, const Lane& L0, const Lane& LH) {
  if (!P->pwc_packed) {
    set_error("pwc_forward: call udet_pack_pwc first");
    return UDET_ERR_ARG;
  }
  const Config& c = P->cfg;
  const int B = c.batch;
  hipStream_t s = L0.s;
  UDET_TRY(launch_pack_pwc_input(img1, img2, ws + P->buf(P->bid("pwc.x8")).off, (long)B * c.in_h * c.in_w, s));
  // siamese feature pyramid on the 2B stacked images (model_pwcnet.py:149-168)
  for (int l = 1; l <= 6; ++l)
    for (const char* suf : {"a", "aa", "b"}) {
      char nm[64];
      snprintf(nm, sizeof(nm), "pwcnet/featpyr/conv%d%s", l, suf);
      UDET_TRY(run_fwd(P, *find_layer(P->pwc, nm), 2 * B, ws, L0));
    }
  for (int l = 6; l >= 2; --l) {
    const int h = c.in_h >> l, w = c.in_w >> l, C = PWC_CH[l];
    const Buf& cb = P->buf(P->bid(S("pwc.c%d", l)));
    const Buf& slab = P->buf(P->bid(S("pwc.slab%d", l)));
    const float* c1 = ws + cb.off;
    const float* c2 = ws + cb.off + (size_t)B * h * w * C;
    // warp(c2, up_flow * 20/2^l) -> cost volume -> slab segments [corr 81 | c1] in one launch (model_pwcnet.py:616-623);
    // level 6 correlates c1 with c2 itself and has no c1 / up_flow / up_feat segments
    const bool warped = l != 6;
    const double px = (double)B * h * w;
    prof_begin(P, PROF_CORR, 2.0 * px * 81.0 * C, px * ((warped ? 3.0 * C + 2.0 : 2.0 * C) + 81.0) * 4.0, s, S("warp_costvol%d", l).c_str());
    UDET_TRY(launch_warp_cost_volume(c1, c2, warped ? ws + slab.off : nullptr, slab.ld, 532 + C, 20.0f / (float)(1 << l), ws + slab.off,
                                     slab.ld, 448, warped ? 532 : -1, nullptr, B, h, w, C, s));
    prof_end(P, s);
    for (int i = 0; i < 5; ++i) UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/predict_flow/conv%d_%d", l, i)), B, ws, L0));
    // upfeat (the slab) is complete: the flow head and the learned upsampling of upfeat only read it, so they run on
    // their own lane while the context network's six wide convolutions occupy the caller's stream
    order_after(P, L0, LH);
    UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/predict_flow/flow%d", l)), B, ws, LH));
    if (l != 2) UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/upsample/up_feat%d", l)), B, ws, LH));
    for (int i = 1; i <= 6; ++i) UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/ctxt/dc_conv%d%d", l, i)), B, ws, L0));
    order_after(P, LH, L0);
    UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/ctxt/dc_conv%d%d", l, 7)), B, ws, L0));  // + flow (residual operand)
    if (l != 2) UDET_TRY(run_fwd(P, *find_layer(P->pwc, S("pwcnet/upsample/up_flow%d", l)), B, ws, L0));
  }
  // flow_pred = resize_bilinear(flow2, x4) * 4   (model_pwcnet.py:641-646)
  const Buf& fr = P->buf(P->bid("pwc.rflow2"));
  const Buf& ff = P->buf(P->bid("flow_full"));
  return launch_resize_bilinear_fwd(ws + fr.off, fr.ld, 0, B, fr.h, fr.w, ws + ff.off, 2, 0, c.in_h, c.in_w, 2, 4.0f, 1.f, s);
}

int plan_pwc_forward(Plan* P, const float* img1, const float* img2, float* ws, hipStream_t s) {
  P->ev_next = 0;
  if (P->prefetch_pending) (void)hipStreamWaitEvent(s, P->prefetch_ev, 0);  // the prefetch owns the PWC buffers until it is done
  return pwc_forward_on(P, img1, img2, ws, lane_of(P, s, 0), lane_of(P, s, 2));
}

// image -> img_h x img_w  (adversarial_learner.py:87-90)
static int plan_prepare_image(Plan* P, const float* img1, const char* dst, float* ws, hipStream_t s) {
  const Config& c = P->cfg;
  return launch_resize_bilinear_fwd(img1, 3, 0, c.batch, c.in_h, c.in_w, ws + P->buf(P->bid(dst)).off, 3, 0, c.img_h, c.img_w,
                                    3, 1.f, 1.f, s);
}
// flow -> img_h x img_w, flow / flow_normalizer  (adversarial_learner.py:91-97)
static int plan_prepare_flow(Plan* P, const char* dst, float* ws, hipStream_t s) {
  const Config& c = P->cfg;
  const Buf& ff = P->buf(P->bid("flow_full"));
  return launch_resize_bilinear_fwd(ws + ff.off, 2, 0, c.batch, c.in_h, c.in_w, ws + P->buf(P->bid(dst)).off, 2, 0, c.img_h,
                                    c.img_w, 2, 1.f, c.flow_normalizer, s);
}

int plan_prefetch(Plan* P, const float* img1, const float* img2, float* ws, hipStream_t s) {
  const Lane L0 = lane_of(P, s, 0), LC = lane_of(P, s, 4), LH = lane_of(P, s, 5);
  P->in_prefetch = true;  // own event pool: the step's pool is recycled while this work is still in flight
  P->ev_next_prefetch = 0;
  order_after(P, L0, LC);
  int rc = pwc_forward_on(P, img1, img2, ws, LC, LH);
  P->in_prefetch = false;
  UDET_TRY(rc);
  UDET_TRY(plan_prepare_flow(P, "flow.next", ws, LC.s));
  UDET_TRY(plan_prepare_image(P, img1, "image.next", ws, LC.s));
  if (!P->prefetch_ev) (void)hipEventCreateWithFlags(&P->prefetch_ev, hipEventDisableTiming);
  (void)hipEventRecord(P->prefetch_ev, LC.s);
  P->prefetch_pending = true;
  return UDET_OK;
}

// --------------------------------------------------- generator / recover ----
int plan_generator_forward(Plan* P, float* ws, hipStream_t s) {
  const Config& c = P->cfg;
  const long HW = (long)c.img_h * c.img_w;
  const Lane L0 = lane_of(P, s, 0);
  double* part = reinterpret_cast<double*>(ws + P->small_off + 4096);
  UDET_TRY(launch_gen_input(ws + P->buf(P->bid("image")).off, ws + P->buf(P->bid("flow")).off, part,
                            ws + P->buf(P->bid("gen.in")).off, c.batch, HW, s));
  for (const auto& L : P->gen) UDET_TRY(run_fwd(P, L, c.batch, ws, L0));
  return UDET_OK;
}
// the 17 layers alone, from a caller-packed "gen.in" ([image 3 | standardised flow 2 | 0 0 0]): nets.generator_net's own contract
int plan_generator_layers(Plan* P, float* ws, hipStream_t s) {
  const Lane L0 = lane_of(P, s, 0);
  for (const auto& L : P->gen) UDET_TRY(run_fwd(P, L, P->cfg.batch, ws, L0));
  return UDET_OK;
}

static int rec_resize(Plan* P, const char* src, const char* dst, int N, float* ws, hipStream_t s) {
  const Buf &a = P->buf(P->bid(src)), &b = P->buf(P->bid(dst));
  return launch_resize_bilinear_fwd(ws + a.off, a.ld, 0, N, a.h, a.w, ws + b.off, b.ld, 0, b.h, b.w, a.ld, 1.f, 1.f, s);
}

static const char* ENC_NAMES[9] = {"conv1", "conv2", "conv3", "conv31", "conv4", "conv41", "conv5", "conv51", "conv6"};

// The image branch of recover_net (nets.py:57-65): image replicated for the `ncalls` invocations + encoder A.  It
// depends on nothing but the image, so the step runs it beside PWC-Net / the generator.
// encoder A's skip tensors (the slab segments aconv1/2/31/41/51 and aconv6), computed for the B images, fanned out to the other calls' samples
static int share_enc_a_output(Plan* P, const Layer& L, int ncalls, float* ws, hipStream_t s) {
  const Buf& y = P->buf(L.y);
  if (ncalls > 1 && (y.name.find("concat") != std::string::npos || y.name == "rec.conv6"))
    return launch_share_samples(ws + y.off, (long)P->cfg.batch * y.h * y.w, y.ld, L.y_coff, L.cout, ncalls, s);
  return UDET_OK;
}
// with_layers = false (pair mode, round 6): only the encoder's input is packed here; its nine layers ride in encoder B's launches
// (plan_recover_forward), one pair launch per level
static int plan_rec_image_branch(Plan* P, int ncalls, float* ws, const Lane& ln, bool with_layers = true) {
  const Config& c = P->cfg;
  if (ncalls < 1) return UDET_OK;
  const long Ppix = (long)c.batch * c.img_h * c.img_w;
  // every call sees the same image: encoder A runs once on the B images, then the tensors the decoder reads are fanned out
  UDET_TRY(launch_pack_imgin(ws + P->buf(P->bid("image")).off, ws + P->buf(P->bid("rec.imgin")).off, Ppix, 1, ln.s));
  if (!with_layers) return UDET_OK;
  for (int i = 0; i < 9; ++i) {
    const Layer& L = *find_layer(P->rec, std::string("a") + ENC_NAMES[i]);
    UDET_TRY(run_fwd(P, L, c.batch, ws, ln));
    UDET_TRY(share_enc_a_output(P, L, ncalls, ws, ln.s));
  }
  P->enc_a_shared = true;
  return UDET_OK;
}

// mask, recover inputs, `ncalls` batched recover invocations (nets.py:45-110; adversarial_learner.py:107-131)
int plan_recover_forward(Plan* P, int ncalls, float* ws, hipStream_t s, bool inputs_prepacked, bool skip_enc_a, bool enc_a_input_packed) {
  const Config& c = P->cfg;
  const int B = c.batch, N = ncalls * B;
  const long Ppix = (long)B * c.img_h * c.img_w;
  const Lane L0 = lane_of(P, s, 0);
  if (!inputs_prepacked)
    UDET_TRY(launch_mask_rec_inputs(ws + P->buf(P->bid("gen.a17")).off, ws + P->buf(P->bid("flow")).off,
                                    ws + P->buf(P->bid("mask")).off, ws + P->buf(P->bid("rec.fin")).off, Ppix, ncalls, s));
  if (ncalls < 1) return UDET_OK;
  if (!skip_enc_a && pairs_on(P)) {
    // pair mode: level by level, encoder A (the B images -- or, caller-packed inputs, every sample) and encoder B (the N samples of the
    // batched calls) in ONE launch (run_fwd_pair); the image encoder's input was packed by the caller of this function or is packed here
    if (!inputs_prepacked && !enc_a_input_packed) UDET_TRY(plan_rec_image_branch(P, ncalls, ws, L0, false));
    for (int i = 0; i < 9; ++i) {
      const Layer& La = *find_layer(P->rec, std::string("a") + ENC_NAMES[i]);
      UDET_TRY(run_fwd_pair(P, La, inputs_prepacked ? N : B, *find_layer(P->rec, std::string("b") + ENC_NAMES[i]), N, ws, L0));
      if (!inputs_prepacked) UDET_TRY(share_enc_a_output(P, La, ncalls, ws, s));
    }
    P->enc_a_shared = !inputs_prepacked;
  } else {
    if (!skip_enc_a) {
      if (inputs_prepacked) {  // caller-packed images may differ between the calls: per-sample encoder
        for (int i = 0; i < 9; ++i) UDET_TRY(run_fwd(P, *find_layer(P->rec, std::string("a") + ENC_NAMES[i]), N, ws, L0));
        P->enc_a_shared = false;
      } else {
        UDET_TRY(plan_rec_image_branch(P, ncalls, ws, L0));
      }
    }
    for (int i = 0; i < 9; ++i) UDET_TRY(run_fwd(P, *find_layer(P->rec, std::string("b") + ENC_NAMES[i]), N, ws, L0));
  }
  for (int k = 5; k >= 1; --k) {
    // the up-sampled tensor rec.r{k+1}: the forward of an up-conv level (Layer::upb) reads the ringed low-resolution source instead, and
    // nothing else in the forward or in either backward-data pass reads it -- only the level's filter gradient does, which builds it
    // itself (rec_backward).  Inference, the generator-only schedule steps and the generator-loss pass never pay for it.
    const Layer& dcl = *find_layer(P->rec, S("deconv%d", k));
    if (!dcl.upb) UDET_TRY(rec_resize(P, k == 5 ? "rec.conv6" : S("rec.concat%d", k + 1).c_str(), S("rec.r%d", k + 1).c_str(), N, ws, s));
    UDET_TRY(run_fwd(P, dcl, N, ws, L0));
    if (k < 5) {
      UDET_TRY(rec_resize(P, S("rec.flow%d", k + 1).c_str(), S("rec.rf%d", k + 1).c_str(), N, ws, s));
      UDET_TRY(run_fwd(P, *find_layer(P->rec, S("upflow%d", k)), N, ws, L0));
    }
    UDET_TRY(run_fwd(P, *find_layer(P->rec, S("flow%d", k)), N, ws, L0));
  }
  const Buf& f1 = P->buf(P->bid("rec.flow1"));
  return launch_resize_bilinear_fwd(ws + f1.off, f1.ld, 0, N, f1.h, f1.w, ws + P->buf(P->bid("pred")).off, 2, 0, c.img_h,
                                    c.img_w, 2, 1.f, 1.f, s);
}

// small region layout (floats from small_off): [0,8) losses, [16,16+4B) coef, [256,258) noise flag,
// [1024,1024+5B) sums, [2048,..) per-variable |g| partial sums, [4096,..) flow-stat partials (doubles), [8192,..) loss partials
int plan_losses(Plan* P, float* ws, hipStream_t s) {
  const Config& c = P->cfg;
  float* sm = ws + P->small_off;
  const long HW = (long)c.img_h * c.img_w;
  return launch_losses(ws + P->buf(P->bid("flow")).off, ws + P->buf(P->bid("mask")).off, ws + P->buf(P->bid("pred")).off, HW,
                       c.batch, c.cbn, c.epsilon, (float)(c.img_w * c.img_h * c.batch), sm + 8192, sm, sm + 16, sm + UDET_SMALL_SUMS, s);
}

// join the pending prefetch and move its staging buffers into "flow" / "image"
int plan_prefetch_consume(Plan* P, float* ws, hipStream_t s) {
  if (!P->prefetch_pending) {
    set_error("forward_prefetched: no udet_prefetch_flow is pending");
    return UDET_ERR_ARG;
  }
  (void)hipStreamWaitEvent(s, P->prefetch_ev, 0);
  P->prefetch_pending = false;
  const Buf &fn = P->buf(P->bid("flow.next")), &in = P->buf(P->bid("image.next"));
  UDET_HIP(hipMemcpyAsync(ws + P->buf(P->bid("flow")).off, ws + fn.off, fn.floats() * sizeof(float), hipMemcpyDeviceToDevice, s));
  UDET_HIP(hipMemcpyAsync(ws + P->buf(P->bid("image")).off, ws + in.off, in.floats() * sizeof(float), hipMemcpyDeviceToDevice, s));
  return UDET_OK;
}

// adversarial_learner.py:83-204.  Lane 1 carries the image branch (image resize, recover encoder A) beside
// PWC-Net and the generator on the caller's stream.
int plan_forward(Plan* P, const float* img1, const float* img2, int ncalls, float* ws, hipStream_t s, bool prefetched) {
  P->ev_next = 0;
  const Lane L0 = lane_of(P, s, 0), LI = lane_of(P, s, 1);
  if (prefetched) {
    UDET_TRY(plan_prefetch_consume(P, ws, s));
    img1 = img2 = nullptr;
  } else if (img1 && P->prefetch_pending) {
    // a stand-alone forward (validation between training steps) while a prefetch is in flight: the prefetch owns the
    // PWC buffers until it is done; its staged result stays valid for the next udet_prefetch_consume
    (void)hipStreamWaitEvent(s, P->prefetch_ev, 0);
  }
  order_after(P, L0, LI);
  if (img1) UDET_TRY(plan_prepare_image(P, img1, "image", ws, LI.s));
  hipEvent_t e_img = nullptr;
  if (LI.s != L0.s) {
    e_img = next_event(P);
    (void)hipEventRecord(e_img, LI.s);
  }
  // pair mode: lane 1 only packs the image encoder's input; its layers run inside encoder B's launches (plan_recover_forward)
  const bool paired = pairs_on(P);
  UDET_TRY(plan_rec_image_branch(P, ncalls, ws, LI, !paired));
  if (img1) {
    UDET_TRY(pwc_forward_on(P, img1, img2, ws, L0, lane_of(P, s, 2)));
    UDET_TRY(plan_prepare_flow(P, "flow", ws, s));
  }
  if (e_img) (void)hipStreamWaitEvent(s, e_img, 0);
  UDET_TRY(plan_generator_forward(P, ws, s));
  order_after(P, LI, L0);
  UDET_TRY(plan_recover_forward(P, ncalls, ws, s, false, !paired, paired));
  if (ncalls == 3) UDET_TRY(plan_losses(P, ws, s));
  return UDET_OK;
}

// ------------------------------------------------------------ backward ----
// Recover decoder/encoder backward for the first N samples of the batched calls, seeded by <dp>pred.
// `dp` is the gradient-buffer family ("d" recover-loss pass, "e" generator-loss pass); "rec.u<dp>.*" mirrors it with
// dU = gradient * act'(activation), emitted by whichever launch writes a region last, so that the backward-data and
// backward-filter launches of every activated layer read their operand without an act' on load.
// with_wgrad: parameter gradients into g_rec, each on lane LW right where its output gradient is final.
// need_dfin: propagate to the b-encoder input.
static int rec_backward(Plan* P, int N, const char* dp, bool with_wgrad, bool need_dfin, const float* w_rec, float* g_rec, float* ws,
                        const Lane& LD, const Lane& LW, const Lane* LAp = nullptr, const Lane* LWdecp = nullptr) {
  const Config& c = P->cfg;
  hipStream_t s = LD.s;
  const std::string pre = std::string("rec.") + dp + ".", upre = std::string("rec.u") + dp + ".";
  auto B_ = [&](const std::string& n) { return P->bid(n); };
  auto D_ = [&](const std::string& n) { return P->bid(pre + n); };
  auto U_ = [&](const std::string& n) { return P->bid(upre + n); };
  auto Lr = [&](const std::string& n) { return find_layer(P->rec, n); };
  // (LWdec: the filter-gradient lane of the DECODER layers -- the experiment knob UDET_KNOB_REC_DEC_WGRAD_LANE may send them to another
  // lane than the encoders'; every lane that ran one is joined to LD at the end)
  const Lane LWdec = LWdecp ? *LWdecp : LW;
  auto wgrad_on = [&](const Lane& lw, const Layer& L, int dy, bool is_du, int n) -> int {
    order_after(P, LD, lw);
    return run_wgrad(P, L, n < 0 ? N : n, dy, is_du, w_rec, g_rec, ws, lw);
  };
  auto wgrad = [&](const Layer& L, int dy, bool is_du, int n = -1) -> int { return wgrad_on(LW, L, dy, is_du, n); };
  auto wgrad_dec = [&](const Layer& L, int dy, bool is_du) -> int { return wgrad_on(LWdec, L, dy, is_du, -1); };
  // Encoder A's backward (shared image encoder: B samples, 8 backward-data + 9 filter-gradient launches of 12-25 us each) depends on the
  // decoder's gradients only and touches channel segments no launch of encoder B's chain touches, so it may run as its own chain on
  // another lane (LAp) beside encoder B's instead of inside the recover-loss pass's serial chain; joined to LD at the end.
  const Lane LA = LAp ? *LAp : LD;
  const bool a_own_lane = LA.s != LD.s;
  // shared encoder A (plan_rec_image_branch): its activations exist for the B images only and are identical for every
  // call, so the calls' output gradients are summed (fold) where they enter the encoder and its backward runs on B samples
  const int ncopies = N / c.batch;
  const bool a_shared = P->enc_a_shared && ncopies > 1;
  const Emit none;
  const float LEAK = 0.2f;
  // pred = resize(flow1)
  {
    const Buf& df1 = P->buf(D_("flow1"));
    UDET_TRY(launch_resize_bilinear_bwd(ws + P->buf(B_(std::string(dp) + ".pred")).off, 2, 0, N, c.img_h, c.img_w, ws + df1.off,
                                        df1.ld, 0, df1.h, df1.w, 2, 0, s));
  }
  for (int k = 1; k <= 5; ++k) {
    const std::string cname = S("concat%d", k);
    const int dconcat = D_(cname), uconcat = U_(cname);
    const Layer* fl = Lr(S("flow%d", k));
    const Layer* dc = Lr(S("deconv%d", k));
    // concat_k feeds flow_k (and, for k>1, the resize of the next finer level wrote it first).  flow_k's backward-data
    // launch is the last writer of the deconv_k segment [0, Cout(deconv_k)): it emits that segment's dU.
    Emit em;
    em.ubuf = uconcat; em.abuf = fl->x; em.c0 = 0; em.c1 = dc->cout; em.act = ACT_LEAKY; em.alpha = LEAK;
    UDET_TRY(run_dgrad(P, *fl, N, D_(S("flow%d", k)), false, dconcat, 0, k == 1 ? 0 : 1, -1, em, ws, LD));
    if (with_wgrad) UDET_TRY(wgrad_dec(*fl, D_(S("flow%d", k)), false));
    if (k < 5) {
      const Layer* uf = Lr(S("upflow%d", k));
      if (with_wgrad) UDET_TRY(wgrad_dec(*uf, dconcat, false));  // linear layer: raw gradient
      UDET_TRY(run_dgrad(P, *uf, N, dconcat, false, D_(S("rf%d", k + 1)), 0, 0, -1, none, ws, LD));
      const Buf &drf = P->buf(D_(S("rf%d", k + 1))), &dfn = P->buf(D_(S("flow%d", k + 1)));
      UDET_TRY(launch_resize_bilinear_bwd(ws + drf.off, drf.ld, 0, N, drf.h, drf.w, ws + dfn.off, dfn.ld, 0, dfn.h, dfn.w, drf.ld, 0, s));
    }
    if (with_wgrad) {
      if (dc->upb) {  // the filter gradient's X operand (plan_recover_forward skipped it): built on the filter-gradient lane, right here
        order_after(P, LD, LWdec);
        UDET_TRY(rec_resize(P, k == 5 ? "rec.conv6" : S("rec.concat%d", k + 1).c_str(), S("rec.r%d", k + 1).c_str(), N, ws, LWdec.s));
      }
      UDET_TRY(wgrad_dec(*dc, uconcat, true));
    }
    if (dc->upb_bwd) {
      UDET_TRY(run_dgrad_upb(P, *dc, N, uconcat, D_(S("p%d", k + 1)), D_(S("concat%d", k + 1)), ws, LD));
      continue;
    }
    const int dr = D_(S("r%d", k + 1));
    UDET_TRY(run_dgrad(P, *dc, N, uconcat, true, dr, 0, 0, -1, none, ws, LD));
    const Buf& bdr = P->buf(dr);
    const Buf& dsrc = P->buf(k == 5 ? D_("conv6") : D_(S("concat%d", k + 1)));
    UDET_TRY(launch_resize_bilinear_bwd(ws + bdr.off, bdr.ld, 0, N, bdr.h, bdr.w, ws + dsrc.off, dsrc.ld, 0, dsrc.h, dsrc.w,
                                        bdr.ld, 0, s));
  }
  // conv6's output gradient was finalised by the resize adjoint: emit its dU with an elementwise pass (3x6 grid)
  {
    const Buf &d6 = P->buf(D_("conv6")), &a6 = P->buf(B_("rec.conv6")), &u6 = P->buf(U_("conv6"));
    UDET_TRY(launch_emit_du(ws + d6.off, ws + a6.off, ws + u6.off, (long)N * d6.h * d6.w, d6.ld, 0, d6.ld, ACT_LEAKY, LEAK, s));
  }
  // encoders, deepest first.  gradient buffers mirror the forward buffers of each conv's output / input.
  if (a_own_lane) order_after(P, LD, LA);  // (everything the decoder wrote)
  const bool pair_enc = pairs_on(P, 1) && with_wgrad && !a_own_lane;
  for (int i = 8; i >= 0; --i) {
    DgradJob job[2];
    int njob = 0;
    for (const char* e : {"a", "b"}) {
      // encoder A sees only the image: without parameter gradients (generator-loss pass) nothing upstream needs it
      if (e[0] == 'a' && !with_wgrad) continue;
      const Layer* L = Lr(std::string(e) + ENC_NAMES[i]);
      const int du = U_(P->buf(L->y).name.substr(4));  // dU of this layer's output (emitted by its consumer's dgrad)
      const bool shared = e[0] == 'a' && a_shared;
      const int Ne = shared ? c.batch : N;
      const bool own = e[0] == 'a' && a_own_lane;
      const Lane& LE = own ? LA : LD;  // this encoder's backward-data chain
      hipStream_t se = LE.s;
      if (shared && i == 8) {  // aconv6 half of conv6: dU was emitted per call above
        const Buf& u = P->buf(du);
        UDET_TRY(launch_fold_samples(ws + u.off, (long)c.batch * u.h * u.w, u.ld, L->y_coff, L->cout, ncopies, se));
      }
      if (with_wgrad) {
        if (own) UDET_TRY(run_wgrad(P, *L, Ne, du, true, w_rec, g_rec, ws, LA));  // (same lane: in chain order, no event)
        else UDET_TRY(wgrad(*L, du, true, Ne));
      }
      if (i == 0) {
        if (e[0] == 'b' && need_dfin) UDET_TRY(run_dgrad(P, *L, N, du, true, D_("fin"), 0, 0, -1, none, ws, LD));
        continue;
      }
      const std::string xname = P->buf(L->x).name;
      const int dx = D_(xname.substr(4));
      const bool slab_in = xname.find("concat") != std::string::npos;  // slab inputs already hold the decoder's gradient
      // this launch is the last writer of the previous encoder layer's output gradient: emit its dU
      Emit em;
      em.ubuf = U_(xname.substr(4)); em.abuf = L->x; em.c0 = 0; em.c1 = L->cin; em.act = ACT_LEAKY; em.alpha = LEAK;
      if (shared && slab_in) {  // the decoder's gradient of this skip segment, summed over the calls
        const Buf& d = P->buf(dx);
        UDET_TRY(launch_fold_samples(ws + d.off, (long)c.batch * d.h * d.w, d.ld, L->x_coff, L->cin, ncopies, se));
      }
      if (pair_enc) {  // (both encoders' launches of this level go out together below: everything either of them waits for is enqueued)
        job[njob].L = L; job[njob].N = Ne; job[njob].dy = du; job[njob].dx = dx; job[njob].dx_coff = L->x_coff;
        job[njob].accumulate = slab_in ? 1 : 0; job[njob].em = em;
        ++njob;
        continue;
      }
      UDET_TRY(run_dgrad(P, *L, Ne, du, true, dx, L->x_coff, slab_in ? 1 : 0, -1, em, ws, LE));
    }
    if (njob == 2) UDET_TRY(run_dgrad_pair(P, job[0], job[1], ws, LD));
    else if (njob == 1) UDET_TRY(run_dgrad(P, *job[0].L, job[0].N, job[0].dy, true, job[0].dx, job[0].dx_coff, job[0].accumulate, -1, job[0].em, ws, LD));
  }
  if (a_own_lane) order_after(P, LA, LD);
  if (with_wgrad && LWdec.s != LW.s) order_after(P, LWdec, LD);
  return UDET_OK;
}

// d recover_loss / d FlownetS  (loss_utils.py:18; adversarial_learner.py:230-234)
static int backward_recover(Plan* P, const float* w_rec, float* g_rec, float* ws, const Lane& LD, const Lane& LW, const Lane* LA = nullptr,
                            const Lane* LWdec = nullptr) {
  const Config& c = P->cfg;
  const long BHW = (long)c.batch * c.img_h * c.img_w;
  UDET_TRY(launch_rec_loss_bwd(ws + P->buf(P->bid("flow")).off, ws + P->buf(P->bid("mask")).off, ws + P->buf(P->bid("pred")).off,
                               ws + P->buf(P->bid("d.pred")).off, BHW, c.cbn, 1.0f / (float)(c.img_w * c.img_h * c.batch), LD.s));
  return rec_backward(P, 3 * c.batch, "d", true, false, w_rec, g_rec, ws, LD, LW, LA, LWdec);
}

// d generator_loss / d MaskNet  (adversarial_learner.py:224-228): through recover calls 1 and 2 (data gradient only,
// "e" buffers), the mask, then the generator.
static int backward_generator(Plan* P, const float* w_gen, float* g_gen, float* ws, const Lane& LD, const Lane& LW, const Lane* LWlate = nullptr,
                              int nlate = 0) {
  const Config& c = P->cfg;
  const int B = c.batch;
  const long HW = (long)c.img_h * c.img_w;
  hipStream_t s = LD.s;
  float* sm = ws + P->small_off;
  auto B_ = [&](const std::string& n) { return P->bid(n); };
  UDET_TRY(launch_gen_loss_bwd(ws + P->buf(B_("flow")).off, ws + P->buf(B_("mask")).off, ws + P->buf(B_("pred")).off, sm + 16,
                               ws + P->buf(B_("e.pred")).off, ws + P->buf(B_("d.mask")).off, HW, B, c.cbn, s));
  UDET_TRY(rec_backward(P, 2 * B, "e", false, true, nullptr, nullptr, ws, LD, LD));
  UDET_TRY(launch_mask_bwd(ws + P->buf(B_("d.mask")).off, ws + P->buf(B_("rec.e.fin")).off, ws + P->buf(B_("flow")).off,
                           ws + P->buf(B_("mask")).off, ws + P->buf(B_("gen.d17")).off, B * HW, s));
  // generator, last layer first.  gen.d{k} = gradient w.r.t. layer k's (post-skip) output; gen.u{k} = that times
  // act'(a_k), emitted by the launch that finalises gen.d{k} (the next layer's backward-data launch or the 2x2 pooling).
  for (int i = 16; i >= 0; --i) {
    const Layer& L = P->gen[i];
    const bool has_act = L.act != ACT_NONE;
    const int dy = has_act ? B_(S("gen.u%d", i + 1)) : B_(S("gen.d%d", i + 1));
    const Lane& LWi = (LWlate && i < nlate) ? *LWlate : LW;  // (experiment knob: the last `nlate` layers' filter gradients on another lane)
    order_after(P, LD, LWi);
    UDET_TRY(run_wgrad(P, L, B, dy, has_act, w_gen, g_gen, ws, LWi));
    if (i == 0) break;
    // skip gradients: x2 = a6 (+ d11), x1 = a3 (+ d14), x0 = a1 (+ d15)   (nets.py:29,32,33)
    int res = -1;
    if (i == 6) res = B_("gen.d11");
    if (i == 3) res = B_("gen.d14");
    if (i == 1) res = B_("gen.d15");
    const int dx = B_(S("gen.d%d", i));
    const Layer& Lp = P->gen[i - 1];  // the layer whose output gradient this launch produces
    const int ap = Lp.y2 >= 0 ? Lp.y2 : Lp.y;
    {
      // (up-sampling layers too: their backward-data launch walks the full-resolution dU with stride 2 and produces the gradient of
      // the low-resolution input directly -- see setup_up_dgrad)
      Emit em;
      if (Lp.act != ACT_NONE) { em.ubuf = B_(S("gen.u%d", i)); em.abuf = ap; em.c0 = 0; em.c1 = L.cin; em.act = Lp.act; em.alpha = Lp.alpha; }
      UDET_TRY(run_dgrad(P, L, B, dy, has_act, dx, 0, 0, res, em, ws, LD));
    }
  }
  return UDET_OK;
}

// Both passes only read the forward state, so with which == 3 they run concurrently: the recover-loss pass on the
// caller's stream (its filter gradients on lane 2), the generator-loss pass on lane 1 (filter gradients on lane 3).
int plan_backward(Plan* P, int which, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec, float* ws, hipStream_t s) {
  P->ev_next = 0;
  const Lane L0 = lane_of(P, s, 0), L1 = lane_of(P, s, 1), L2 = lane_of(P, s, 2), L3 = lane_of(P, s, 3);
  // grad_ev[net]: recorded where that network's flat gradient buffer is final, BEFORE the caller's stream joins the other
  // pass -- a communication stream that waits on it (udet_stream_wait_grads) can exchange the recover gradients while the
  // (longer) generator-loss pass is still running
  auto mark = [&](int net) {
    if (!P->grad_ev[net]) (void)hipEventCreateWithFlags(&P->grad_ev[net], hipEventDisableTiming);
    (void)hipEventRecord(P->grad_ev[net], s);
  };
  if (which == 3) {
    order_after(P, L0, L1);
    const int la = (int)plan_knob(UDET_KNOB_ENC_A_LANE);
    const Lane LA = lane_of(P, s, la > 0 && la < Plan::NLANE ? la : 0);
    // The recover DECODER's filter gradients (deconv / flow / upflow of the five levels, 0.65 ms of large launches, ready from the first
    // 0.1 ms of the pass on) run on lane 3 -- the generator's filter-gradient queue, which has nothing to do until the generator-loss pass
    // has walked the recover net (~0.8 ms) -- instead of lane 2, which shares its hardware queue with the recover-loss pass's own
    // backward-data chain: on one queue they executed strictly behind each other, on two the large filter-gradient launches fill the CUs the
    // chain's many small launches leave idle.  8.89 -> 8.70 ms per step (two boxes, alternating runs); the ENCODERS' filter gradients
    // there as well: 8.96 (they then sit in front of the generator's, which are on the step's critical tail).  profiles/NOTES.md, round 5.
    const int lw = (int)plan_knob(UDET_KNOB_REC_DEC_WGRAD_LANE), le = (int)plan_knob(UDET_KNOB_REC_ENC_WGRAD_LANE);
    const Lane LWD = lane_of(P, s, lw > 0 && lw < Plan::NLANE ? lw : 3);
    const Lane LWE = lane_of(P, s, le > 0 && le < Plan::NLANE ? le : 2);
    UDET_TRY(backward_recover(P, w_rec, g_rec, ws, L0, LWE, la > 0 ? &LA : nullptr, &LWD));
    if (LWE.s != L2.s) order_after(P, LWE, L0);
    // The filter gradients of the generator's first four layers (the LAST ones the generator-loss pass reaches: conv4_downsample ... conv1) run
    // on lane 2 instead of lane 3: they are the tail of the step, and lane 2 -- the recover encoders' filter gradients -- has long drained by
    // then, so the two queues finish the tail side by side.  Round 6 sweep, three runs each on one box (ms per step): 0 layers 8.274,
    // 2: 8.224, 3: 8.218, 4: 8.216, 6: 8.236, 8: 8.278.  (Round 3 measured the same move as a loss -- the kernels behind it were slower then.)
    // (experiment knob, libudet_exp.so only: v > 0 that many layers, v < 0 none)
    const long kl = plan_knob(UDET_KNOB_GEN_WGRAD_LATE);
    const int nlate = kl > 0 ? (int)kl : (kl < 0 ? 0 : 4);
    // the recover gradients are final HERE (rec_backward joined its filter-gradient lanes): their event is recorded before the generator's
    // late filter gradients are enqueued on lane 2, so a communication stream waiting on it still starts under the generator-loss pass
    order_after(P, L2, L0);
    mark(NET_REC);
    UDET_TRY(backward_generator(P, w_gen, g_gen, ws, L1, L3, nlate > 0 ? &L2 : nullptr, nlate));
    order_after(P, L2, L0);
    order_after(P, L1, L0);
    order_after(P, L3, L0);
    mark(NET_GEN);
  } else if (which == 2) {
    UDET_TRY(backward_recover(P, w_rec, g_rec, ws, L0, L2, nullptr, &L3));  // (decoder filter gradients on lane 3's queue, as above)
    order_after(P, L2, L0);
    mark(NET_REC);
  } else {
    UDET_TRY(backward_generator(P, w_gen, g_gen, ws, L0, L3));
    order_after(P, L3, L0);
    mark(NET_GEN);
  }
  return UDET_OK;
}

// ------------------------------------------------------------ optimizer ----
static int overflow_consume(Plan* P, int net, bool wait);
int plan_apply(Plan* P, int net, float* w, float* g, float* m, float* v, float* ws, hipStream_t s) {
  if (net != NET_GEN && net != NET_REC) {
    set_error("apply: net must be 1 (generator) or 2 (recover)");
    return UDET_ERR_ARG;
  }
  const Config& c = P->cfg;
  const NetParams& np = net_params(net);
  float* sm = ws + P->small_off;
  const float* flag = nullptr;
  if (net == NET_GEN) {  // can_change=True (adversarial_learner.py:224-228)
    const long* tab = reinterpret_cast<const long*>(ws + P->seg_off[net]);
    UDET_TRY(launch_grad_absmean(g, tab, tab + np.p.size(), (int)np.p.size(), sm + 2048, 1e-5f, sm + UDET_SMALL_NOISE, s));
    flag = sm + UDET_SMALL_NOISE;
  }
  // fp16 mode: the report of this network's PREVIOUS apply is consumed first (normally long over: that apply is a whole step old).  A
  // dropped update gives its step count back there (overflow_consume), so the count -- the shared beta powers, udet_get_adam_step,
  // checkpoints -- equals the number of updates that really happened from the next apply on, instead of running one ahead per dropped
  // update for good.  The host cannot know at enqueue time, so the applies enqueued between a drop and its report (at most the other
  // network's, and this network's next) still use the advanced count: a transient of one step, not a persistent offset.
  if (c.conv_fp16) UDET_TRY(overflow_consume(P, net, true));
  const long t = ++P->adam_t;  // ONE optimizer object: beta powers advance on every apply (:216)
  const double lr_t = (double)c.lr * sqrt(1.0 - pow((double)c.beta2, (double)t)) / (1.0 - pow((double)c.beta1, (double)t));
  const int* skip = nullptr;
  if (c.conv_fp16) {  // overflow guard of the static fp16 gradient scale (see Plan::ovf_host)
    int* cnt = reinterpret_cast<int*>(sm + UDET_SMALL_OVF) + 2 * (net - 1);  // {this apply, running total}; view "fp16_overflow"
    UDET_TRY(launch_nonfinite_count(g, (long)np.total, cnt, s));
    skip = cnt;
  }
  UDET_TRY(launch_adam(w, g, m, v, (long)np.total, (float)lr_t, c.beta1, c.beta2, c.adam_eps, c.clip, flag, c.noise_seed,
                       (uint64_t)t, s, 0, skip));
  if (c.conv_fp16) {
    if (!P->ovf_host) UDET_HIP(hipHostMalloc(reinterpret_cast<void**>(&P->ovf_host), 4 * sizeof(int), hipHostMallocDefault));
    if (!P->ovf_ev[net]) UDET_HIP(hipEventCreateWithFlags(&P->ovf_ev[net], hipEventDisableTiming));
    // (the pinned slot and the event are re-used: the previous report was consumed above)
    UDET_HIP(hipMemcpyAsync(P->ovf_host + 2 * (net - 1), skip, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    UDET_HIP(hipEventRecord(P->ovf_ev[net], s));
    P->ovf_pending[net] = true;
  }
  return UDET_OK;
}

// looks at the pending report of `net` (wait: block until its apply has run; otherwise only if it has) and books a dropped update
static int overflow_consume(Plan* P, int net, bool wait) {
  if (!P->ovf_pending[net]) return UDET_OK;
  if (wait) UDET_HIP(hipEventSynchronize(P->ovf_ev[net]));
  else if (hipEventQuery(P->ovf_ev[net]) != hipSuccess) return UDET_OK;  // still in flight: looked at by a later call
  P->ovf_pending[net] = false;
  const int n = P->ovf_host[2 * (net - 1)];
  if (n > 0) {
    P->ovf_report_values += n; P->ovf_report_nets |= net; ++P->ovf_skipped;
    if (P->adam_t > 0) --P->adam_t;  // the dropped update did not happen: it does not count (see plan_apply)
  }
  return UDET_OK;
}

// fp16 mode: waits for the pending per-network reports and books dropped updates (adam_t gives their step back) WITHOUT raising the
// pending UDET_ERR_OVERFLOW -- that stays for the next call on the plan.  udet_get_adam_step / udet_set_adam_step go through here, so a
// count read for a checkpoint is never one ahead of the updates that really happened, and a count restored from one is not decremented
// by the report of an apply that preceded the restore.
void plan_settle_adam_step(Plan* P) {
  if (!P->cfg.conv_fp16) return;
  for (int net = 1; net <= 2; ++net) (void)overflow_consume(P, net, true);
}

int plan_check_overflow(Plan* P, bool wait) {
  if (!P->cfg.conv_fp16) return UDET_OK;
  for (int net = 1; net <= 2; ++net) UDET_TRY(overflow_consume(P, net, wait));
  if (!P->ovf_report_values) return UDET_OK;
  const int bad = P->ovf_report_values, which = P->ovf_report_nets;
  P->ovf_report_values = 0;
  P->ovf_report_nets = 0;
  set_error("fp16 mode: %d non-finite gradient value(s) in the %s gradients -- a gradient operand times 4096 overflowed fp16 (|dU| > 16); "
            "the optimizer update(s) concerned were NOT applied (weights and Adam slots unchanged)", bad,
            which == 3 ? "generator and recover" : (which == 1 ? "generator" : "recover"));
  return UDET_ERR_OVERFLOW;
}

}  // namespace udet
