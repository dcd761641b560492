// hk_inst_bounce_all.hip -- k_bounce<W, F, STG> (hk_kernels.h): every shading feature, at the default register budget
#include "hk_kernels.h"

#define HK_LB(W_, F_, G_) hipLaunchKernelGGL((k_bounce<W_, F_, G_>), dim3(a.grid), dim3(HK_BOUNCE_BLOCK), a.ldsBytes, a.stream, a.s, a.stage, a.qIn, a.nextCnt, a.shCnt, a.depth, a.maxDepth, a.A, a.B, a.hits, a.sh, a.contrib, a.gens, a.sortPaths, a.screen)
#define HK_LB_STG(W_, F_) do { switch (STG) { case 3: HK_LB(W_, F_, 3); break; case 2: HK_LB(W_, F_, 2); break; case 1: HK_LB(W_, F_, 1); break; default: HK_LB(W_, F_, 0); break; } } while (0)
bool hk_launch_bounce_all(int W, int F, int STG, const BounceLaunch& a) {
  if (W != 3 || F != HK_FEAT_ALL) return false;
  HK_LB_STG(3, HK_FEAT_ALL);
  return true;
}
#ifdef HK_EXP_BOUNCE_STAMPS
int hk_bounce_stamps_read_all(unsigned long long* acc16, int reset) { return hk_bounce_stamps_read(acc16, reset); }
#endif
