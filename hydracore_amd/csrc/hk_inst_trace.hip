// hk_inst_trace.hip -- the traversal kernels of libhydra_hip.so (rows a/T1, a/T2): every instantiation of k_trace, k_shadow and
// k_trace_dyn, behind the three launchers of hk_kernels.h.
#include "hk_kernels.h"

void hk_launch_trace_static(bool count, bool alpha, const TraceLaunch& a) {
#define HK_L(CNT, AL) hipLaunchKernelGGL((k_trace<CNT, AL>), dim3(a.grid), dim3(HK_TRACE_BLOCK), 0, a.stream, a.s, a.q, a.a4, a.b4, a.hits, a.perRay3, a.totals5, a.carry)
  if (count) { if (alpha) HK_L(true, true); else HK_L(true, false); }
  else { if (alpha) HK_L(false, true); else HK_L(false, false); }
#undef HK_L
}
void hk_launch_shadow_static(bool count, const TraceLaunch& a) {
  if (count) hipLaunchKernelGGL(k_shadow<true>, dim3(a.grid), dim3(HK_TRACE_BLOCK), 0, a.stream, a.s, a.q, a.a4, a.b4, a.vis, a.totals5);
  else hipLaunchKernelGGL(k_shadow<false>, dim3(a.grid), dim3(HK_TRACE_BLOCK), 0, a.stream, a.s, a.q, a.a4, a.b4, a.vis, a.totals5);
}
void hk_launch_trace_dyn(bool anyhit, bool count, bool toptris, bool alpha, const TraceLaunch& a) {
  float4* out = reinterpret_cast<float4*>(a.hits);
#define HK_L(AH, CNT, TT, AL, VT) hipLaunchKernelGGL((k_trace_dyn<AH, CNT, TT, AL, VT>), dim3(a.grid), dim3(HK_TRACE_BLOCK), 0, a.stream, a.s, a.q, a.fetchCounters, a.a4, a.b4, out, a.vis, a.totals5, a.minActive, a.raysPerLane, a.wq, a.wt, a.wi)
#define HK_LV(AH, CNT, TT, AL) do { if (a.vote) HK_L(AH, CNT, TT, AL, true); else HK_L(AH, CNT, TT, AL, false); } while (0)
  if (!anyhit) {
    if (alpha) { if (count) HK_LV(false, true, false, true); else HK_LV(false, false, false, true); }          // the alpha-tested kernels exist without LDS triangles only
    else if (toptris) { if (count) HK_LV(false, true, true, false); else HK_LV(false, false, true, false); }
    else { if (count) HK_LV(false, true, false, false); else HK_LV(false, false, false, false); }
  } else {                                                                                                     // shadow rays: tree 0 without the alpha test (Common.cpp:156-180)
    if (toptris) { if (count) HK_LV(true, true, true, false); else HK_LV(true, false, true, false); }
    else if (count) HK_LV(true, true, false, false);
    else if (a.vote && a.unordered)
      hipLaunchKernelGGL((k_trace_dyn<true, false, false, false, true, true>), dim3(a.grid), dim3(HK_TRACE_BLOCK), 0, a.stream, a.s, a.q, a.fetchCounters, a.a4, a.b4, out, a.vis, a.totals5, a.minActive, a.raysPerLane, a.wq, a.wt, a.wi);
    else HK_LV(true, false, false, false);
  }
#undef HK_LV
#undef HK_L
}
