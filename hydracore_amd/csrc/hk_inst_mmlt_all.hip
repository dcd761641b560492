// hk_inst_mmlt_all.hip -- the stage kernels of IntegratorMMLT::F (hk_kernels.h, hk_bidir.h): every shading feature
#include "hk_kernels.h"

template <int F>
static void launch_mmlt(int kernel, const MmltLaunch& a) {
  if (kernel == 0) hipLaunchKernelGGL((k_mmlt_step<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v, a.currDepth, a.q, a.in, a.hits, a.out, a.outCount);
  else if (kernel == 1) hipLaunchKernelGGL((k_mmlt_connect_begin<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v);
  else hipLaunchKernelGGL((k_mmlt_connect_end<F>), dim3(a.grid), dim3(256), 0, a.stream, a.s, a.v);
}
bool hk_launch_mmlt_all(int kernel, int F, const MmltLaunch& a) {
  if (F != HK_FEAT_ALL) return false;
  launch_mmlt<HK_FEAT_ALL>(kernel, a);
  return true;
}
