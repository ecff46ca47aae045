// Epilogue of the convolution kernels (conv_igemm.hip, conv_wino.hip): bias, activation, second output, residual, accumulate,
// channel-offset store and the optional dU emission of one output element / of four consecutive channels.
#pragma once
#include "common.h"

namespace udet {

// bias, activation, second output, residual, accumulate, store (+ optional dU emission) of one output element
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, int off, int n, float v) {
  if (p.bias) v += p.bias[n];
  v = act_fwd(v, p.act, p.alpha);
  if (p.y2) p.y2[(size_t)off * p.ldy2 + p.y2_coff + n] = v;
  if (p.res) v += p.res[(size_t)off * p.ldres + p.res_coff + n];
  float* dst = p.y + (size_t)off * p.ldy + p.y_coff + n;
  if (p.accumulate) v += *dst;
  *dst = v;
  if (p.uo && n >= p.u_c0 && n < p.u_c1)
    p.uo[(size_t)off * p.ldu + p.u_coff + n] = v * act_dfo(p.ua[(size_t)off * p.ldua + p.ua_coff + n], p.uact, p.ualpha);
}

// Four consecutive channels of one output element row at once (every operand 16-byte aligned: epilogue4_ok).
__device__ __forceinline__ float4 act_fwd4(float4 v, int act, float alpha) {
  return make_float4(act_fwd(v.x, act, alpha), act_fwd(v.y, act, alpha), act_fwd(v.z, act, alpha), act_fwd(v.w, act, alpha));
}
__device__ __forceinline__ void conv_epilogue4(const ConvParams& p, int off, int n, float4 v) {
  if (p.bias) {
    v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3];
  }
  v = act_fwd4(v, p.act, p.alpha);
  if (p.y2) *reinterpret_cast<float4*>(p.y2 + (size_t)off * p.ldy2 + p.y2_coff + n) = v;
  if (p.res) {
    const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)off * p.ldres + p.res_coff + n);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  float4* dst = reinterpret_cast<float4*>(p.y + (size_t)off * p.ldy + p.y_coff + n);
  if (p.accumulate) {
    const float4 o = *dst;
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  *dst = v;
  if (p.uo && n >= p.u_c0 && n < p.u_c1) {  // (u_c0, u_c1 multiples of 4: the quad is inside or outside as a whole)
    const float4 ua = *reinterpret_cast<const float4*>(p.ua + (size_t)off * p.ldua + p.ua_coff + n);
    *reinterpret_cast<float4*>(p.uo + (size_t)off * p.ldu + p.u_coff + n) =
        make_float4(v.x * act_dfo(ua.x, p.uact, p.ualpha), v.y * act_dfo(ua.y, p.uact, p.ualpha), v.z * act_dfo(ua.z, p.uact, p.ualpha),
                    v.w * act_dfo(ua.w, p.uact, p.ualpha));
  }
}
// The same epilogue in two phases, for store loops that keep several quads in flight (round 6).  conv_epilogue4 loads its operands
// (bias, residual, the accumulate target, the saved activation of the dU emission) where it needs them; called from a loop, every
// iteration then waits out a global-load latency before its store, and the stores keep the compiler from hoisting the next iteration's
// loads (they might alias) -- 16 iterations of the Winograd kernels' tile store cost ~10 us of a 55 us launch that way.  Here the
// caller loads the bias quad ONCE (it does not depend on the pixel), requests the per-pixel operands of a batch of quads first
// (epi4_request), and finishes them afterwards (epi4_finish): one latency per batch instead of one per quad.
struct Epi4Req {
  float4 res, acc, ua;
};
// Loop-invariant activation coefficients.  act_fwd / act_dfo select the activation with `if (act == ...)` per ELEMENT; inside an unrolled
// store loop hipcc turns those into scalar branches -- five per element, ~320 per lane in the Winograd kernels' tile store, ~8000 cycles
// of a 16 400-cycle store loop (tools/wino_stamps.py, round 6).  Here the selection happens once: none / leaky are ONE formula
// (v > 0 ? v : v * slope, slope = 1 for none), ELU is a template parameter the caller switches on outside its loop, and the derivative
// through the saved output is a > 0 ? 1 : fma(a, ue, us) with (ue, us) = (0, 1) none, (0, alpha) leaky, (1, 1) ELU.  Same values, bit for bit.
struct EpiAct {
  float slope, ue, us;
};
__device__ __forceinline__ EpiAct epi_act(const ConvParams& p) {
  EpiAct e;
  e.slope = p.act == ACT_LEAKY ? p.alpha : 1.f;
  e.ue = p.uact == ACT_ELU ? 1.f : 0.f;
  e.us = p.uact == ACT_LEAKY ? p.ualpha : 1.f;
  return e;
}
template <bool ELU>
__device__ __forceinline__ float act_fwd_c(float v, float slope) {
  if (ELU) {
    // no divergent branch around the exponential (hipcc wraps `v > 0 ? v : f(v)` in an EXEC-mask branch per element when f is costly):
    // max(v, 0) + expm1(min(v, 0)) -- one of the two terms is exactly 0; a NaN input stays NaN
    const float r = fmaxf(v, 0.f) + elu_negative(fminf(v, 0.f));
    return v != v ? v : r;
  }
  return v > 0.f ? v : v * slope;
}
__device__ __forceinline__ float act_dfo_c(float a, const EpiAct& e) { return a > 0.f ? 1.f : fmaf(a, e.ue, e.us); }
__device__ __forceinline__ float4 epi4_bias(const ConvParams& p, int n) {
  return p.bias ? make_float4(p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void epi4_request(const ConvParams& p, int off, int n, Epi4Req& r) {
  if (p.res) r.res = *reinterpret_cast<const float4*>(p.res + (size_t)off * p.ldres + p.res_coff + n);
  if (p.accumulate) r.acc = *reinterpret_cast<const float4*>(p.y + (size_t)off * p.ldy + p.y_coff + n);
  if (p.uo && n >= p.u_c0 && n < p.u_c1) r.ua = *reinterpret_cast<const float4*>(p.ua + (size_t)off * p.ldua + p.ua_coff + n);
}
template <bool ELU>
__device__ __forceinline__ void epi4_finish(const ConvParams& p, int off, int n, float4 v, const float4& bias, const Epi4Req& r, const EpiAct& ea) {
  v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;  // (same operation order as conv_epilogue4: results are bit-identical)
  v = make_float4(act_fwd_c<ELU>(v.x, ea.slope), act_fwd_c<ELU>(v.y, ea.slope), act_fwd_c<ELU>(v.z, ea.slope), act_fwd_c<ELU>(v.w, ea.slope));
  if (p.y2) *reinterpret_cast<float4*>(p.y2 + (size_t)off * p.ldy2 + p.y2_coff + n) = v;
  if (p.res) { v.x += r.res.x; v.y += r.res.y; v.z += r.res.z; v.w += r.res.w; }
  if (p.accumulate) { v.x += r.acc.x; v.y += r.acc.y; v.z += r.acc.z; v.w += r.acc.w; }
  *reinterpret_cast<float4*>(p.y + (size_t)off * p.ldy + p.y_coff + n) = v;
  if (p.uo && n >= p.u_c0 && n < p.u_c1)
    *reinterpret_cast<float4*>(p.uo + (size_t)off * p.ldu + p.u_coff + n) =
        make_float4(v.x * act_dfo_c(r.ua.x, ea), v.y * act_dfo_c(r.ua.y, ea), v.z * act_dfo_c(r.ua.z, ea), v.w * act_dfo_c(r.ua.w, ea));
}
// The store-only form: launches with neither residual nor accumulate nor dU emission (most forward launches).  No global load at all
// in the store loop -- on gfx950 loads and stores share one counter (vmcnt), so any load the loop waits for also drains every store
// issued before it: the generic form above pays a store round trip per quad (~700 cycles), this one only issues.
__device__ __forceinline__ bool epi4_plain(const ConvParams& p) { return !p.res && !p.accumulate && !p.uo; }
template <bool ELU>
__device__ __forceinline__ void epi4_finish_plain(const ConvParams& p, int off, int n, float4 v, const float4& bias, float slope) {
  v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
  v = make_float4(act_fwd_c<ELU>(v.x, slope), act_fwd_c<ELU>(v.y, slope), act_fwd_c<ELU>(v.z, slope), act_fwd_c<ELU>(v.w, slope));
  if (p.y2) *reinterpret_cast<float4*>(p.y2 + (size_t)off * p.ldy2 + p.y2_coff + n) = v;
  *reinterpret_cast<float4*>(p.y + (size_t)off * p.ldy + p.y_coff + n) = v;
}
// the same with a 32-bit element index (callers whose launcher has checked N * H * W * ldy < 2^31: the Winograd family): one multiply-add
// per address instead of a 64-bit multiply sequence
template <bool ELU>
__device__ __forceinline__ void epi4_finish_plain32(const ConvParams& p, int off, int n, float4 v, const float4& bias, float slope) {
  v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
  v = make_float4(act_fwd_c<ELU>(v.x, slope), act_fwd_c<ELU>(v.y, slope), act_fwd_c<ELU>(v.z, slope), act_fwd_c<ELU>(v.w, slope));
  if (p.y2) *reinterpret_cast<float4*>(p.y2 + (size_t)off * p.ldy2 + p.y2_coff + n) = v;
  *reinterpret_cast<float4*>(p.y + (unsigned)(off * p.ldy + p.y_coff + n)) = v;
}
__device__ __forceinline__ bool epilogue4_out_ok(const ConvParams& p) {  // conv_epilogue4 may be used on this launch's outputs
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = (p.Cout & 3) == 0 && ((p.ldy | p.y_coff) & 3) == 0 && al(p.y);
  if (p.y2) ok = ok && ((p.ldy2 | p.y2_coff) & 3) == 0 && al(p.y2);
  if (p.res) ok = ok && ((p.ldres | p.res_coff) & 3) == 0 && al(p.res);
  if (p.uo) ok = ok && ((p.ldu | p.u_coff | p.ldua | p.ua_coff | p.u_c0 | p.u_c1) & 3) == 0 && al(p.uo) && al(p.ua);
  return ok;
}

}  // namespace udet
