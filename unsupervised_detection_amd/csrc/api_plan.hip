// extern "C" plan-level surface of libudet.so (see include/udet.h).
#include <stdlib.h>
#include <string.h>

#include "conv_host.h"
#include "elementwise.h"
#include "plan.h"

using namespace udet;

struct udet_plan {
  Plan* p;
};

extern "C" {

int udet_plan_create(const udet_config* c, udet_plan** out) {
  if (!c || !out) { set_error("plan_create: null argument"); return UDET_ERR_ARG; }
  Config k;
  k.batch = c->batch; k.in_h = c->in_h; k.in_w = c->in_w; k.img_h = c->img_h; k.img_w = c->img_w;
  k.flow_normalizer = c->flow_normalizer; k.cbn = c->cbn; k.epsilon = c->epsilon;
  k.lr = c->lr; k.beta1 = c->beta1; k.beta2 = c->beta2; k.adam_eps = c->adam_eps; k.clip = c->clip;
  k.noise_seed = c->noise_seed;
  k.conv_fp16 = c->conv_fp16 ? 1 : 0;
  Plan* P = plan_build(k);
  if (!P) return UDET_ERR_SHAPE;
  *out = new udet_plan{P};
  return UDET_OK;
}
void udet_plan_destroy(udet_plan* h) {
  if (h) { delete h->p; delete h; }
}
size_t udet_workspace_bytes(const udet_plan* h) { return h->p->arena_floats * sizeof(float); }
int udet_plan_init(udet_plan* h, void* ws, void* stream) { return plan_init_workspace(h->p, (float*)ws, (hipStream_t)stream); }

int udet_param_count(int net) { return (net < 0 || net > 2) ? 0 : (int)net_params(net).p.size(); }
size_t udet_param_total(int net) { return (net < 0 || net > 2) ? 0 : net_params(net).total; }
int udet_param_info(int net, int i, const char** name, int* rank, int* shape, size_t* offset) {
  if (net < 0 || net > 2 || i < 0 || i >= (int)net_params(net).p.size()) { set_error("param_info: out of range"); return UDET_ERR_ARG; }
  const ParamDesc& d = net_params(net).p[i];
  *name = d.name.c_str(); *rank = d.rank; *offset = d.offset;
  for (int k = 0; k < 4; ++k) shape[k] = d.shape[k];
  return UDET_OK;
}
int udet_buffer_count(const udet_plan* h) { return (int)h->p->bufs.size(); }
int udet_buffer_info(const udet_plan* h, int i, const char** name, size_t* offset_bytes, int* dims) {
  if (i < 0 || i >= (int)h->p->bufs.size()) { set_error("buffer_info: out of range"); return UDET_ERR_ARG; }
  const Buf& b = h->p->bufs[i];
  *name = b.name.c_str(); *offset_bytes = b.off * sizeof(float);
  dims[0] = b.n; dims[1] = b.h; dims[2] = b.w; dims[3] = b.ld;
  return UDET_OK;
}
int udet_plan_set_concurrent(udet_plan* h, int on) {
  if (!h) { set_error("plan_set_concurrent: null plan"); return UDET_ERR_ARG; }
  if (on)
    if (h->p->cand.empty()) { set_error("plan_set_concurrent: the plan has no side streams"); return UDET_ERR_UNSUPPORTED; }
  h->p->concurrent = on != 0;
  return UDET_OK;
}
int udet_plan_lane_queues(udet_plan* h, void* stream, int* queue) {
  if (!h || !queue) { set_error("plan_lane_queues: null argument"); return UDET_ERR_ARG; }
  return plan_lane_queues(h->p, (hipStream_t)stream, queue);
}
int udet_plan_pin_lanes(udet_plan* h, void* stream, void* const* side_streams, int n) {
  if (!h || (n > 0 && !side_streams)) { set_error("plan_pin_lanes: null argument"); return UDET_ERR_ARG; }
  return plan_pin_lanes(h->p, (hipStream_t)stream, reinterpret_cast<hipStream_t const*>(side_streams), n);
}
long udet_fp16_overflow_count(udet_plan* h) {
  if (!h) return 0;
  (void)plan_check_overflow(h->p, true);  // (the error text stays in udet_last_error; the count is the answer here)
  return h->p->ovf_skipped;
}
long udet_get_adam_step(const udet_plan* h) {
  plan_settle_adam_step(h->p);  // (fp16 mode: a dropped update still in flight gives its step back first; see udet.h)
  return h->p->adam_t;
}
void udet_set_adam_step(udet_plan* h, long t) {
  plan_settle_adam_step(h->p);  // reports of applies that preceded the restore are booked against the OLD count, not the restored one
  h->p->adam_t = t;
}

int udet_pack_pwc(udet_plan* h, const float* w_pwc, void* ws, void* stream) {
  return plan_pack_pwc(h->p, w_pwc, (float*)ws, (hipStream_t)stream);
}
int udet_pack_trainable(udet_plan* h, const float* w_gen, const float* w_rec, void* ws, void* stream) {
  return plan_pack_trainable(h->p, w_gen, w_rec, (float*)ws, (hipStream_t)stream);
}
int udet_pwc_forward(udet_plan* h, const float* img1, const float* img2, void* ws, void* stream) {
  return plan_pwc_forward(h->p, img1, img2, (float*)ws, (hipStream_t)stream);
}
int udet_forward_from_flow(udet_plan* h, int ncalls, void* ws, void* stream) {
  UDET_TRY(plan_check_overflow(h->p, false));  // conv_fp16: report an update the previous step had to drop
  if (ncalls < 0 || ncalls > 3) { set_error("forward: ncalls must be 0..3"); return UDET_ERR_ARG; }
  return plan_forward(h->p, nullptr, nullptr, ncalls, (float*)ws, (hipStream_t)stream);
}
int udet_generator_forward(udet_plan* h, void* ws, void* stream) {
  UDET_TRY(plan_generator_forward(h->p, (float*)ws, (hipStream_t)stream));
  return plan_recover_forward(h->p, 0, (float*)ws, (hipStream_t)stream);  // ncalls=0: only the mask is produced
}
int udet_generator_layers(udet_plan* h, void* ws, void* stream) {
  h->p->ev_next = 0;
  UDET_TRY(plan_generator_layers(h->p, (float*)ws, (hipStream_t)stream));
  return plan_recover_forward(h->p, 0, (float*)ws, (hipStream_t)stream);  // ncalls=0: only the mask is produced
}
int udet_recover_forward(udet_plan* h, int n, void* ws, void* stream) {
  if (n < 1 || n > 3) { set_error("recover_forward: n must be 1..3 (multiples of the plan batch)"); return UDET_ERR_ARG; }
  return plan_recover_forward(h->p, n, (float*)ws, (hipStream_t)stream, true);
}
int udet_forward(udet_plan* h, const float* img1, const float* img2, int ncalls, void* ws, void* stream) {
  UDET_TRY(plan_check_overflow(h->p, false));  // conv_fp16: report an update the previous step had to drop
  if (ncalls < 0 || ncalls > 3) { set_error("forward: ncalls must be 0..3"); return UDET_ERR_ARG; }
  if (!img1 || !img2) { set_error("forward: null image pointer"); return UDET_ERR_ARG; }
  return plan_forward(h->p, img1, img2, ncalls, (float*)ws, (hipStream_t)stream);
}
int udet_prefetch_flow(udet_plan* h, const float* img1, const float* img2, void* ws, void* stream) {
  if (!img1 || !img2) { set_error("prefetch_flow: null image pointer"); return UDET_ERR_ARG; }
  return plan_prefetch(h->p, img1, img2, (float*)ws, (hipStream_t)stream);
}
int udet_prefetch_consume(udet_plan* h, void* ws, void* stream) {
  return plan_prefetch_consume(h->p, (float*)ws, (hipStream_t)stream);
}
int udet_forward_prefetched(udet_plan* h, int ncalls, void* ws, void* stream) {
  UDET_TRY(plan_check_overflow(h->p, false));  // conv_fp16: report an update the previous step had to drop
  if (ncalls < 0 || ncalls > 3) { set_error("forward: ncalls must be 0..3"); return UDET_ERR_ARG; }
  return plan_forward(h->p, nullptr, nullptr, ncalls, (float*)ws, (hipStream_t)stream, true);
}
int udet_backward(udet_plan* h, int which, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec, void* ws,
                  void* stream) {
  UDET_TRY(plan_check_overflow(h->p, false));  // conv_fp16: report an update the previous step had to drop
  if (which < 1 || which > 3) { set_error("backward: which must be 1 (generator), 2 (recover) or 3 (both)"); return UDET_ERR_ARG; }
  return plan_backward(h->p, which, w_gen, w_rec, g_gen, g_rec, (float*)ws, (hipStream_t)stream);
}
int udet_apply(udet_plan* h, int net, float* w, float* g, float* m, float* v, void* ws, void* stream) {
  return plan_apply(h->p, net, w, g, m, v, (float*)ws, (hipStream_t)stream);
}
int udet_train_step(udet_plan* h, int which, const float* img1, const float* img2, float* w_gen, float* w_rec, float* g_gen,
                    float* g_rec, float* m_gen, float* v_gen, float* m_rec, float* v_rec, void* ws, void* stream) {
  UDET_TRY(udet_pack_trainable(h, w_gen, w_rec, ws, stream));
  UDET_TRY(udet_forward(h, img1, img2, 3, ws, stream));
  UDET_TRY(udet_backward(h, which, w_gen, w_rec, g_gen, g_rec, ws, stream));
  if (which & 1) UDET_TRY(udet_apply(h, NET_GEN, w_gen, g_gen, m_gen, v_gen, ws, stream));
  if (which & 2) UDET_TRY(udet_apply(h, NET_REC, w_rec, g_rec, m_rec, v_rec, ws, stream));
  return UDET_OK;
}

/* autotuner: one untimed forward + both backward passes over random data during which every distinct convolution
 * problem of the plan times its candidate kernel configurations (tile, split-K, wave specialisation, wgrad split)
 * and caches the fastest; the activation / gradient regions of the workspace are re-zeroed afterwards. */
int udet_autotune(udet_plan* h, const float* w_gen, const float* w_rec, float* g_gen, float* g_rec, void* ws_, void* stream) {
  Plan* P = h->p;
  float* ws = (float*)ws_;
  hipStream_t s = (hipStream_t)stream;
  if (!P->pwc_packed) { set_error("autotune: pack the PWC-Net and trainable weights first"); return UDET_ERR_ARG; }
  const size_t lo = P->packed_floats, hi = P->small_off;
  UDET_TRY(launch_fill_uniform(ws + lo, (long)(hi - lo), 0x5eedull, -0.5f, 0.5f, s));
  const size_t img_floats = (size_t)P->cfg.batch * P->cfg.in_h * P->cfg.in_w * 3;
  if (P->wgrad_floats < 2 * img_floats) { set_error("autotune: workspace too small for the probe images"); return UDET_ERR_ARG; }
  const float* img1 = ws + P->wgrad_off[0];
  const float* img2 = img1 + img_floats;
  const bool was_concurrent = P->concurrent;
  P->concurrent = false;  // candidates are timed on the caller's stream with nothing else in flight
  conv_set_tuning(1);
  wgrad_set_tuning(1);
  int rc = udet_forward(h, img1, img2, 3, ws_, stream);
  if (rc == UDET_OK) rc = udet_backward(h, 3, w_gen, w_rec, g_gen, g_rec, ws_, stream);
  conv_set_tuning(0);
  wgrad_set_tuning(0);
  P->concurrent = was_concurrent;
  UDET_HIP(hipMemsetAsync(ws + lo, 0, (hi - lo) * sizeof(float), s));
  UDET_HIP(hipStreamSynchronize(s));
  return rc;
}
int udet_tuned_shapes(void) { return conv_tuned_shapes() + wgrad_tuned_shapes() + conv_pair_tuned_shapes(); }
int udet_tune_save(const char* path) {
  FILE* f = path ? fopen(path, "w") : nullptr;
  if (!f) { set_error("tune_save: cannot open %s", path ? path : "(null)"); return UDET_ERR_ARG; }
  fprintf(f, "udet-tune 2 abi %d\n", UDET_TUNE_ABI);
  conv_tune_dump(f);
  wgrad_tune_dump(f);
  conv_pair_tune_dump(f);
  fclose(f);
  return UDET_OK;
}
int udet_tune_load(const char* path) {
  FILE* f = path ? fopen(path, "r") : nullptr;
  if (!f) { set_error("tune_load: cannot open %s", path ? path : "(null)"); return UDET_ERR_ARG; }
  char line[256];
  int n = 0, ver = 0, abi = -1;
  // the header names the build's tuning ABI: keys and configuration codes of another build mean something else
  if (!fgets(line, sizeof(line), f) || sscanf(line, "udet-tune %d abi %d", &ver, &abi) != 2 || ver != 2 || abi != UDET_TUNE_ABI) {
    fclose(f);
    set_error("tune_load: %s is not a udet-tune 2 file of this build (abi %d)", path, UDET_TUNE_ABI);
    return UDET_ERR_ARG;
  }
  while (fgets(line, sizeof(line), f)) {
    unsigned long long key;
    int a, b, c, d, e;
    int g = 0, nf = sscanf(line, "c %llu %d %d %d %d %d %d", &key, &a, &b, &c, &d, &e, &g);
    if (nf == 6 || nf == 7) { conv_tune_put(key, a, b, c, d, e, nf == 7 ? g : 0); ++n; }
    else if (sscanf(line, "w %llu %d", &key, &a) == 2) { wgrad_tune_put(key, a); ++n; }
    else if (sscanf(line, "p %llu %d %d %d %d", &key, &a, &b, &c, &d) == 5) { conv_pair_tune_put(key, a, b, c, d); ++n; }
  }
  fclose(f);
  return n;
}

/* measurement: per-category timing of the conv / warp-cost-volume launches */
int udet_profile_begin(udet_plan* h) {
  for (auto* r : h->p->prof) delete r;
  h->p->prof.clear();
  h->p->profiling = true;
  return UDET_OK;
}
/* out[cat][5] = {launch groups, kernel ms (sum of the kernels' own start -> stop times), algorithmic flops, algorithmic bytes,
 * bracket ms (hipEventRecord before / after each group: adds the event packets and the dispatch gaps)}; synchronises the stream */
int udet_profile_end(udet_plan* h, double* out, int ncat, void* stream) {
  Plan* P = h->p;
  P->profiling = false;
  g_launch_sink = nullptr;
  UDET_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < ncat * 5; ++i) out[i] = 0.0;
  FILE* dump = nullptr;  // UDET_PROF_DUMP=<file>: one CSV line per launch group (category,name,kernel ms,GFLOP,MB,bracket ms,kernels,each kernel's us,GFLOP the matrix pipe issues)
  if (const char* path = getenv("UDET_PROF_DUMP")) dump = fopen(path, "a");
  for (auto* r : P->prof) {
    float wall = 0.f;
    (void)hipEventElapsedTime(&wall, r->a, r->b);
    double kern = 0.0;
    std::string each;  // the kernels' own durations in launch order, microseconds, '+'-separated (last CSV column)
    for (int k = 0; k < r->sink.n; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r->kev[2 * k], r->kev[2 * k + 1]) == hipSuccess) kern += ms;
      char buf[32];
      snprintf(buf, sizeof(buf), "%s%.1f", k ? "+" : "", ms * 1e3f);
      each += buf;
      (void)hipEventDestroy(r->kev[2 * k]);
      (void)hipEventDestroy(r->kev[2 * k + 1]);
    }
    if (r->sink.n == 0) kern = wall;  // (a group whose launches did not go through the sink)
    if (dump)
      fprintf(dump, "%d,%s,%.4f,%.4f,%.4f,%.4f,%d,%s,%.4f\n", r->cat, r->name.c_str(), kern, r->flops * 1e-9, r->bytes * 1e-6, wall, r->sink.n,
              each.c_str(), r->flops * r->mfma_scale * 1e-9);
    if (r->cat < ncat) {
      out[r->cat * 5 + 0] += 1.0;
      out[r->cat * 5 + 1] += kern;
      out[r->cat * 5 + 2] += r->flops;
      out[r->cat * 5 + 3] += r->bytes;
      out[r->cat * 5 + 4] += wall;
    }
    (void)hipEventDestroy(r->a);
    (void)hipEventDestroy(r->b);
    delete r;
  }
  if (dump) fclose(dump);
  P->prof.clear();
  return UDET_OK;
}

/* the two train ops' gradient passes alone (adversarial_learner.py:224-234) */
int udet_generator_backward(udet_plan* h, const float* w_gen, float* g_gen, void* ws, void* stream) {
  if (!w_gen || !g_gen) { set_error("generator_backward: null argument"); return UDET_ERR_ARG; }
  return plan_backward(h->p, 1, w_gen, nullptr, g_gen, nullptr, (float*)ws, (hipStream_t)stream);
}
int udet_recover_backward(udet_plan* h, const float* w_rec, float* g_rec, void* ws, void* stream) {
  if (!w_rec || !g_rec) { set_error("recover_backward: null argument"); return UDET_ERR_ARG; }
  return plan_backward(h->p, 2, nullptr, w_rec, nullptr, g_rec, (float*)ws, (hipStream_t)stream);
}
/* loss_utils.py:19-21: out2 = {mean over the variables of mean|g_v|, (that < 1e-5) ? 1 : 0} */
int udet_grad_absmean(udet_plan* h, int net, const float* g, float* out2, void* ws_, void* stream) {
  if ((net != NET_GEN && net != NET_REC) || !g || !out2) { set_error("grad_absmean: net must be 1 or 2, g / out2 non-null"); return UDET_ERR_ARG; }
  Plan* P = h->p;
  float* ws = (float*)ws_;
  const NetParams& np = net_params(net);
  const long* tab = reinterpret_cast<const long*>(ws + P->seg_off[net]);
  return launch_grad_absmean(g, tab, tab + np.p.size(), (int)np.p.size(), ws + P->small_off + 2048, 1e-5f, out2, (hipStream_t)stream);
}
/* `stream` waits until the gradient buffer of `net` written by the last udet_backward is final (not for the other pass) */
int udet_stream_wait_grads(udet_plan* h, int net, void* stream) {
  if (net != NET_GEN && net != NET_REC) { set_error("stream_wait_grads: net must be 1 or 2"); return UDET_ERR_ARG; }
  if (!h->p->grad_ev[net]) { set_error("stream_wait_grads: no udet_backward has produced that gradient yet"); return UDET_ERR_ARG; }
  UDET_HIP(hipStreamWaitEvent((hipStream_t)stream, h->p->grad_ev[net], 0));
  return UDET_OK;
}
int udet_tune_rejected(void) { return conv_tune_rejected(); }

}  // extern "C"
