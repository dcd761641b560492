// hk_inst_bounce_classic.hip -- k_bounce<W, F, STG> (hk_kernels.h): the classic set (everything but normal maps, translucency, Blinn and the anisotropic lobes) and its two subsets without glass / without GGX
#include "hk_kernels.h"

#define HK_LB(W_, F_, G_) hipLaunchKernelGGL((k_bounce<W_, F_, G_>), dim3(a.grid), dim3(HK_BOUNCE_BLOCK), a.ldsBytes, a.stream, a.s, a.stage, a.qIn, a.nextCnt, a.shCnt, a.depth, a.maxDepth, a.A, a.B, a.hits, a.sh, a.contrib, a.gens, a.sortPaths, a.screen)
#define HK_LB_STG(W_, F_) do { switch (STG) { case 3: HK_LB(W_, F_, 3); break; case 2: HK_LB(W_, F_, 2); break; case 1: HK_LB(W_, F_, 1); break; default: HK_LB(W_, F_, 0); break; } } while (0)
bool hk_launch_bounce_classic(int W, int F, int STG, const BounceLaunch& a) {
  if (W != 3) return false;
  if (F == (HK_FEAT_CLASSIC & ~HK_FEAT_GLASS)) { HK_LB_STG(3, HK_FEAT_CLASSIC & ~HK_FEAT_GLASS); return true; }
  if (F == (HK_FEAT_CLASSIC & ~HK_FEAT_GGX)) { HK_LB_STG(3, HK_FEAT_CLASSIC & ~HK_FEAT_GGX); return true; }
  if (F == HK_FEAT_CLASSIC) { HK_LB_STG(3, HK_FEAT_CLASSIC); return true; }
  return false;
}
#ifdef HK_EXP_BOUNCE_STAMPS
int hk_bounce_stamps_read_classic(unsigned long long* acc16, int reset) { return hk_bounce_stamps_read(acc16, reset); }
#endif
