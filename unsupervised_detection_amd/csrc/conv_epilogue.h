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
__device__ __forceinline__ bool epilogue4_out_ok(const ConvParams& p) {  // conv_epilogue4 may be used on this launch's outputs
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = (p.Cout & 3) == 0 && ((p.ldy | p.y_coff) & 3) == 0 && al(p.y);
  if (p.y2) ok = ok && ((p.ldy2 | p.y2_coff) & 3) == 0 && al(p.y2);
  if (p.res) ok = ok && ((p.ldres | p.res_coff) & 3) == 0 && al(p.res);
  if (p.uo) ok = ok && ((p.ldu | p.u_coff | p.ldua | p.ua_coff | p.u_c0 | p.u_c1) & 3) == 0 && al(p.uo) && al(p.ua);
  return ok;
}

}  // namespace udet
