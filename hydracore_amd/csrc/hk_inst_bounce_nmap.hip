// hk_inst_bounce_nmap.hip -- k_bounce<W, F, STG> (hk_kernels.h): normal maps over the classic set; also k_hit<W> / k_shade<W>, the split form of the bounce (option fused_bounce = 0)
#include "hk_kernels.h"

#define HK_LB(W_, F_, G_) hipLaunchKernelGGL((k_bounce<W_, F_, G_>), dim3(a.grid), dim3(HK_BOUNCE_BLOCK), a.ldsBytes, a.stream, a.s, a.stage, a.qIn, a.nextCnt, a.shCnt, a.depth, a.maxDepth, a.A, a.B, a.hits, a.sh, a.contrib, a.gens, a.sortPaths, a.screen)
#define HK_LB_STG(W_, F_) do { switch (STG) { case 3: HK_LB(W_, F_, 3); break; case 2: HK_LB(W_, F_, 2); break; case 1: HK_LB(W_, F_, 1); break; default: HK_LB(W_, F_, 0); break; } } while (0)
bool hk_launch_bounce_nmap(int W, int F, int STG, const BounceLaunch& a) {
  if (W != 3 || F != (HK_FEAT_CLASSIC | HK_FEAT_NMAP)) return false;
  HK_LB_STG(3, HK_FEAT_CLASSIC | HK_FEAT_NMAP);
  return true;
}
#ifdef HK_EXP_BOUNCE_STAMPS
int hk_bounce_stamps_read_nmap(unsigned long long* acc16, int reset) { return hk_bounce_stamps_read(acc16, reset); }
#endif

void hk_launch_hit(int W, const SplitLaunch& a) {
#define HK_LH(W_) hipLaunchKernelGGL(k_hit<W_>, dim3(a.grid), dim3(256), 0, a.stream, a.s, a.q, a.nextCnt, a.shCnt, a.depth, a.maxDepth, a.S, a.hits, a.M, a.contrib, a.gens)
  switch (W) { case 3: HK_LH(3); break; case 5: HK_LH(5); break; default: HK_LH(4); break; }
#undef HK_LH
}
void hk_launch_shade(int W, const SplitLaunch& a) {
#define HK_LS(W_) hipLaunchKernelGGL(k_shade<W_>, dim3(a.grid), dim3(256), 0, a.stream, a.s, a.q, a.M, a.S)
  switch (W) { case 3: HK_LS(3); break; case 5: HK_LS(5); break; default: HK_LS(4); break; }
#undef HK_LS
}
