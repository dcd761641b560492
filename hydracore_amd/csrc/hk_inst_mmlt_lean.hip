// hk_inst_mmlt_lean.hip -- the stage kernels of IntegratorMMLT::F (hk_kernels.h, hk_bidir.h): the sky / delta-light / Oren-Nayar subset and the classic set
#include "hk_kernels.h"

template <int F>
static void launch_mmlt(int kernel, const MmltLaunch& a) {
  if (kernel == 0) hipLaunchKernelGGL((k_mmlt_step<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v, a.currDepth, a.q, a.in, a.hits, a.out, a.outCount);
  else if (kernel == 1) hipLaunchKernelGGL((k_mmlt_connect_begin<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v);
  else hipLaunchKernelGGL((k_mmlt_connect_end<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v);
}
bool hk_launch_mmlt_lean(int kernel, int F, const MmltLaunch& a) {
  if (F == (HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR)) { launch_mmlt<HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR>(kernel, a); return true; }
  if (F == HK_FEAT_CLASSIC) { launch_mmlt<HK_FEAT_CLASSIC>(kernel, a); return true; }
  return false;
}
