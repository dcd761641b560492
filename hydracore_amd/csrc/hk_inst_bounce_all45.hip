// hk_inst_bounce_all45.hip -- k_bounce<W, F, STG> (hk_kernels.h): every shading feature at the 4- and 5-wave register budgets (option shade_waves: experiment switches; everything staged, or nothing)
#include "hk_kernels.h"

#define HK_LB(W_, F_, G_) hipLaunchKernelGGL((k_bounce<W_, F_, G_>), dim3(a.grid), dim3(HK_BOUNCE_BLOCK), a.ldsBytes, a.stream, a.s, a.stage, a.qIn, a.nextCnt, a.shCnt, a.depth, a.maxDepth, a.A, a.B, a.hits, a.sh, a.contrib, a.gens, a.sortPaths, a.screen)
#define HK_LB_STG(W_, F_) do { switch (STG) { case 3: HK_LB(W_, F_, 3); break; case 2: HK_LB(W_, F_, 2); break; case 1: HK_LB(W_, F_, 1); break; default: HK_LB(W_, F_, 0); break; } } while (0)
bool hk_launch_bounce_all45(int W, int F, int STG, const BounceLaunch& a) {
  if (F != HK_FEAT_ALL || (W != 4 && W != 5) || (STG != 0 && STG != 3)) return false;
  if (W == 5) { if (STG == 3) HK_LB(5, HK_FEAT_ALL, 3); else HK_LB(5, HK_FEAT_ALL, 0); }
  else { if (STG == 3) HK_LB(4, HK_FEAT_ALL, 3); else HK_LB(4, HK_FEAT_ALL, 0); }
  return true;
}
#ifdef HK_EXP_BOUNCE_STAMPS
int hk_bounce_stamps_read_all45(unsigned long long* acc16, int reset) { return hk_bounce_stamps_read(acc16, reset); }
#endif
