// The step plan: static, shape-specialised description of the hot path
// (PWC-Net forward -> resize -> generator -> 3x recover -> losses -> two backward passes -> clipped Adam)
// over one pre-laid-out workspace.  See DESIGN.md.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace udet {

struct ParamDesc {
  std::string name;
  int rank;
  int shape[4];
  size_t offset, count;  // in floats, inside the net's flat weight buffer
};
struct NetParams {
  std::vector<ParamDesc> p;
  size_t total = 0;
  int add(const std::string& name, int a, int b = 0, int c = 0, int d = 0);
  int find(const std::string& name) const;
};
enum { NET_PWC = 0, NET_GEN = 1, NET_REC = 2 };
const NetParams& net_params(int net);

struct Buf {
  std::string name;
  size_t off;  // floats from the workspace base
  int n, h, w, ld;
  size_t floats() const { return (size_t)n * h * w * ld; }
};

struct Layer {
  std::string name;
  int net;
  int kh, kw, cin, cout, stride, dil;
  bool transposed = false;  // tf.layers.conv2d_transpose (k4 s2)
  bool up = false;          // NN x2 upsample fused into the loader
  int act = ACT_NONE;
  float alpha = 0.f;
  int w_idx = -1, b_idx = -1, g_idx = -1, be_idx = -1;
  // forward packing
  int Kc = 0, ldw = 0, k_split = 0, k_gap = 0;
  size_t wp_off = 0;
  // transposed packing for the backward-data pass (trainable nets)
  int KcT = 0, ldwT = 0;
  size_t wpT_off = 0;
  size_t bias_f_off = 0, scale_off = 0;  // BN-folded bias / per-channel scale (generator)
  // NN x2 upsample + 3x3 convolution as four 2x2 convolutions on the low-resolution grid (one per output parity class, weights of
  // taps that read the same low-resolution pixel pre-added): [16 = class*4 + tap][Kc][ldw] and its transpose for backward-data
  size_t wu_off = 0, wuT_off = 0;
  // Winograd F(2x2,3x3) operands of the 3x3 stride-1 layers (conv_wino.hip): U = G g G^T for the forward pass and, trainable nets,
  // for the backward-data pass (mirrored taps, K = output channels); 0: the layer is not eligible
  size_t wino_off = 0, winoT_off = 0;
  int wino_np = 0, winoT_np = 0;
  // Recover decoder levels 1-3 ("up-conv algebra", plan_exec.hip): legacy bilinear x2 + 4x4 convolution as four 3x3 convolutions on the
  // ringed low-resolution source `xhat`, ONE segmented launch per pass (ConvParams::nseg): weight sets {interior, last row, last column,
  // corner} back to back (forward: wupb_off, backward-data: wupbT_off), segments and device tap tables built by plan_build
  // BUFFER CONTRACT: for a upb layer `x` (rec.r{k+1}, the up-sampled tensor) is NOT written by the forward; it is valid only between the
  // rebuild inside rec_backward (recover-loss pass, filter-gradient lane) and the end of that level's filter gradient.  Debug dumps of it
  // after a forward show the previous step's data; run_fwd refuses the generic path for these layers.
  bool upb = false;
  bool upb_bwd = false;      // backward-data too (the deepest level has too few low-resolution pixels to fill the chip: forward only)
  bool upb_split = false;    // forward: interior as an ordinary four-class launch + the border segments (see run_fwd_upb)
  int xhat = -1, src = -1;   // ringed source [N, h + 2, w + 2, ld] / the source it is built from
  size_t wupb_off = 0, wupbT_off = 0;
  struct SegLaunch {
    int nseg = 0;
    ConvSeg seg[UDET_MAX_SEGS];
    int seg_tap[UDET_MAX_SEGS + 1];
    std::vector<ConvTap> taps;
    size_t tab_off = 0;  // device copy of `taps` (floats from the workspace base)
  } upb_f, upb_b;
  // tensors
  int x = -1, x_coff = 0, y = -1, y_coff = 0;
  int res = -1, res_coff = 0, y2 = -1;
  int H = 0, W = 0;  // stored input grid
  // 2-channel heads: forward as ONE 1x1 GEMM Z[p][(tap,co)] = sum_c X[p][c] W[tap][c][co] + a gather-sum over the taps
  bool col2im = false;
  int ldz = 0, zbuf = -1;
  size_t wz_off = 0;
};

struct Config {
  int batch, in_h, in_w, img_h, img_w;
  float flow_normalizer, cbn, epsilon;
  float lr, beta1, beta2, adam_eps, clip;
  unsigned long long noise_seed;
  int conv_fp16 = 0;  // convolution GEMMs multiply in fp16 (fp32 accumulation)
};

struct Plan {
  Config cfg;
  std::vector<Buf> bufs;
  std::map<std::string, int> buf_by_name;
  std::vector<Layer> pwc, gen, rec;
  size_t packed_floats = 0;      // packed-weight region (start of the workspace)
  size_t arena_floats = 0;       // everything
  // Concurrency: independent chains of the step run on side streams forked from / joined to the caller's stream
  // with events (lane 0 = the caller's stream).  Every lane owns its split-K and wgrad scratch.
  // Lane placement: ROCm maps a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues in creation order, two streams on one
  // queue execute in submission order, and which lanes share a queue decides 11.1 vs 13.3 ms per step.  The plan
  // therefore owns candidate streams (NCAND at first; more are created, up to MAXCAND, while the probe has not found three
  // independent queues -- every new stream goes to the queue with the fewest streams, so a process whose queues are unevenly
  // loaded needs more draws) and, the first time it is driven from a given caller stream, PROBES (lanes.h) which of
  // them run concurrently with that stream and with each other; it then lays the lanes out as
  //   {0: caller's stream, 2: the same stream} {1} {3} {4, 5}          -- four queues, the layout the step was tuned on;
  // with fewer independent queues available lanes 3 -> 1, then 4/5 -> 1, then everything -> 0.
  enum { NLANE = 6, NCAND = 8, MAXCAND = 32 };
  std::vector<hipStream_t> cand;
  struct Placement {
    hipStream_t main = nullptr;
    hipStream_t lane[NLANE] = {};
    int queue[NLANE] = {};  // index of the hardware-queue group the lane sits on (0 = the caller's)
    int nqueues = 1;
  };
  std::vector<Placement> placements;  // one per caller stream seen so far
  int placed = -1;                    // index of the placement in use
  hipEvent_t prefetch_ev = nullptr;  // completion of the last udet_prefetch_flow (lane 4)
  hipEvent_t grad_ev[3] = {nullptr, nullptr, nullptr};  // [net]: that network's gradient buffer is final (last udet_backward)
  bool prefetch_pending = false;
  std::vector<hipEvent_t> ev_pool, ev_pool_prefetch;  // recycled per call (step) / per prefetch
  size_t ev_next = 0, ev_next_prefetch = 0;
  bool in_prefetch = false;
  bool concurrent = true;
  // the last recover forward evaluated encoder A once for the B images and fanned its skip tensors out to the batched
  // calls (true), or per sample on caller-packed inputs (false); the backward pass follows suit
  bool enc_a_shared = false;
  size_t scratch_off[NLANE] = {}, scratch_floats = 0;   // split-K slabs
  size_t wgrad_off[NLANE] = {}, wgrad_floats = 0;       // wgrad partials (lanes that run filter gradients)
  ~Plan();
  size_t small_off = 0;          // losses, coefficients, flags, reduction partials
  size_t ticket_off = 0;         // NLANE x UDET_MAX_TICKETS ints: split-K arrival counters (zero between launches)
  size_t seg_off[3] = {0, 0, 0}; // per-variable (offset,len) tables on device (as long)
  size_t jobs_off[3] = {0, 0, 0}; // PackJob tables on device
  int njobs[3] = {0, 0, 0};
  // optional per-category timing with HIP events on the launch stream (bench.py roofline)
  struct ProfRec {
    int cat; double flops, bytes; hipEvent_t a, b; std::string name;
    double mfma_scale = 1.0;  // multiplications the launch really issues / those of the direct form (Winograd F(2x2,3x3): 16 / 36)
    enum { MAXK = 16 };
    hipEvent_t kev[2 * MAXK];  // start / stop pairs of the kernels launched inside the group
    LaunchSink sink;
  };
  bool profiling = false;
  std::vector<ProfRec*> prof;
  long adam_t = 0;               // number of optimizer applies so far (shared beta powers)
  // fp16 mode (cfg.conv_fp16): overflow guard of the static 4096 gradient scale.  plan_apply counts the non-finite values of the
  // flat gradient buffer on the device; the optimizer kernel drops the update when there are any (the weights are never touched
  // by inf / NaN), the count travels to pinned host memory behind an event and the NEXT call on the plan reports it
  // (UDET_ERR_OVERFLOW once per skipped update; udet_fp16_overflow_count is the synchronous query)
  int* ovf_host = nullptr;       // [2 nets][2]: {count of the last apply, running total}
  hipEvent_t ovf_ev[3] = {nullptr, nullptr, nullptr};
  bool ovf_pending[3] = {false, false, false};
  long ovf_skipped = 0;          // optimizer updates dropped so far
  int ovf_report_values = 0, ovf_report_nets = 0;  // dropped updates no call has reported yet (non-finite values, network bits)
  bool pwc_packed = false;
  int add_buf(const std::string& name, int n, int h, int w, int ld);
  const Buf& buf(int id) const { return bufs[id]; }
  int bid(const std::string& name) const;
};

// capacity (entries) of a network's device PackJob table; plan_init_workspace refuses to copy more
#define UDET_PACKJOB_CAP(layers) (8 * (layers) + 32)
// offsets (floats) inside the plan's small region; registered as named views by plan_build
#define UDET_SMALL_NOISE 256   // noise_flag
#define UDET_SMALL_OVF 300     // fp16_overflow: int[2 nets][2] = {non-finite values of the last apply, running total}
#define UDET_SMALL_SUMS 1024   // loss_sums
Plan* plan_build(const Config& cfg);
void plan_debug_upb_min_pixels(long v);  // (libudet_debug)
// Experiment knobs (tools/knob_bench.py; every knob's 0 is the shipped behaviour).  They exist ONLY in libudet_exp.so, the build of these
// sources with -DUDET_EXPERIMENT (`make exp`): in the release library plan_knob() is a constant 0, every branch that reads a knob folds
// away at compile time -- the work-skipping ablation mask (UDET_KNOB_SKIP) included -- and there is no setter to export.
// id 0: lane (1..5) of the recover net's encoder-A backward chain in a which = 3 backward instead of the recover-loss pass's own stream;
// 1: lane of the recover DECODER's filter gradients (shipped: 3); 2: lane of the recover encoders' filter gradients (shipped: 2); 3: the
// generator's last `v` filter gradients on lane 2; 6: the Winograd-domain filter-gradient family off; 7: timing-only ablation mask -- whole
// launch categories skipped, RESULTS WRONG ON PURPOSE (bit 0 generator filter gradients, 1 recover filter gradients, 2 generator
// backward-data, 3 recover backward-data, 4 PWC-Net forward, 5 generator forward, 6 recover forward); 8: 1 = the recover encoders unpaired
// (encoder A on its own launches and lane, as before round 6).
enum { UDET_KNOB_ENC_A_LANE = 0, UDET_KNOB_REC_DEC_WGRAD_LANE = 1, UDET_KNOB_REC_ENC_WGRAD_LANE = 2, UDET_KNOB_GEN_WGRAD_LATE = 3, UDET_KNOB_NO_WGRAD_WINO = 6, UDET_KNOB_SKIP = 7, UDET_KNOB_NO_PAIRS = 8, UDET_KNOB_COUNT = 12 };
#ifdef UDET_EXPERIMENT
void plan_debug_knob(int id, long v);
long plan_knob(int id);
#else
constexpr long plan_knob(int) { return 0; }
#endif

// execution (all asynchronous on `s`)
int plan_init_workspace(Plan* P, float* ws, hipStream_t s);
int plan_pack_pwc(Plan* P, const float* w_pwc, float* ws, hipStream_t s);
int plan_pack_trainable(Plan* P, const float* w_gen, const float* w_rec, float* ws, hipStream_t s);
int plan_pwc_forward(Plan* P, const float* img1, const float* img2, float* ws, hipStream_t s);
// img1 != null: PWC flow + resizes first; then generator, `ncalls` recover invocations, losses (ncalls == 3)
int plan_forward(Plan* P, const float* img1, const float* img2, int ncalls, float* ws, hipStream_t s, bool prefetched = false);
// PWC flow + resizes of the NEXT step's pair into staging buffers, on lanes 4/5, concurrent with whatever the caller
// enqueues next (PWC-Net is frozen: adversarial_learner.py:211-214); consumed by plan_forward(.., prefetched=true)
int plan_prefetch(Plan* P, const float* img1, const float* img2, float* ws, hipStream_t s);
int plan_prefetch_consume(Plan* P, float* ws, hipStream_t s);
int plan_generator_forward(Plan* P, float* ws, hipStream_t s);
int plan_generator_layers(Plan* P, float* ws, hipStream_t s);
int plan_recover_forward(Plan* P, int ncalls, float* ws, hipStream_t s, bool inputs_prepacked = false, bool skip_enc_a = false, bool enc_a_input_packed = false);
int plan_losses(Plan* P, float* ws, hipStream_t s);
// which: 1 generator loss -> MaskNet, 2 recover loss -> FlownetS, 3 both (the two passes run concurrently)
int plan_backward(Plan* P, int which, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec, float* ws, hipStream_t s);
int plan_apply(Plan* P, int net, float* w, float* g, float* m, float* v, float* ws, hipStream_t s);
// fp16 mode: reports (once) an optimizer update that was dropped because its gradients were not finite; wait: synchronise first
int plan_check_overflow(Plan* P, bool wait);
void plan_settle_adam_step(Plan* P);  // fp16 mode: pending overflow reports booked into adam_t (synchronises); no error raised
int plan_lane_queues(Plan* P, hipStream_t s, int* queue);  // places the lanes for `s` if necessary; returns the queues in use
int plan_pin_lanes(Plan* P, hipStream_t main, hipStream_t const* streams, int n);  // host-provided layout for `main` (no probe)
enum { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_CONV_WGRAD = 2, PROF_WARP = 3, PROF_CORR = 4, PROF_NCAT = 5 };
void prof_begin(Plan* P, int cat, double flops, double bytes, hipStream_t s, const char* name = "");
void prof_end(Plan* P, hipStream_t s);

}  // namespace udet
