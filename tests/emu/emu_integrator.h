// emu_integrator.h -- TEST INFRASTRUCTURE.  One whole path per call: the stage order of IntegratorMISPTLoop2::PathTrace
// (hydra_drv/CPUExp_Integrators_PT_Loop.cpp:264-321) strung together from the product's device functions
// (hydracore_amd/csrc/hk_*.h) so that the host-emulation build can run them under ASan/UBSan.  The product itself only
// has the wavefront split of this loop (hydra_hip.hip: k_trace_dyn / k_bounce).
#pragma once

HK_DEV f3 hk_path_trace_one(const SceneDev& s, HkStack& st, f3 ray_pos, f3 ray_dir, RandomGen& gen, float& rays) {
  TravCounters tc = {0, 0, 0, 0, 0};
  f3 accumColor = mk3(0, 0, 0), thr = mk3(1, 1, 1), currColor = mk3(0, 0, 0);
  float misPdf = 1.0f; bool misSpec = true;
  uint32_t flags = 0;
  const int maxDepth = g_varsI(s)[HV_I_TRACE_DEPTH];
  for (int depth = 0; depth < maxDepth; depth++) {
    const HydraLiteHit hit = hk_traverse<false, false>(make_bvh_view(s.bvh, 0, s.tris, 0), s.haveInst != 0, ray_pos, ray_dir, 0.0f, hk_miss_hit(), st, tc);
    rays += 1.0f;
    if (!HitSome(hit)) { currColor = environmentColor(s, ray_dir, misPdf, misSpec, flags); break; }
    const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
    const float* mat = materialAt(s, surf.matId);
    {
      const int lo0 = (s.globals[HG_LIGHTS_NUM] != 0) ? s.instLightInstId[hit.instId] : -1;
      const float* pL = lightAt(s, lo0);
      const f3 emission = emissionEval(s, ray_pos, ray_dir, surf, flags, misSpec, pL, mat);
      if (dot(emission, emission) > 1e-3f) {
        if (pL != nullptr) {
          const float lgtPdf = pL[HL_PICK_PROB_REV] * lightEvalPDF(s, pL, ray_pos, ray_dir, surf.pos, surf.normal, surf.texCoord);
          float w = misWeightHeuristic(misPdf, lgtPdf);
          if (misSpec) w = 1.0f;
          currColor = emission * w;
        } else currColor = emission;
        break;
      } else if (depth >= maxDepth - 1) { currColor = mk3(0, 0, 0); break; }
    }
    const float4 rl = rndFloat4_Pseudo(gen);
    float pick = 1.0f;
    const int lightOffset = SelectRandomLightRev(rl.z, s, pick);
    f3 explicitColor = mk3(0, 0, 0);
    if (lightOffset >= 0) {
      ShadowSample sam;
      LightSampleRev(s, lightAt(s, lightOffset), mk3(rl.x, rl.y, rl.z), surf.pos, sam);
      const f3 sdir = normalize(sam.pos - surf.pos);
      const f3 spos = OffsShadowRayPos(surf.pos, surf.normal, sdir, surf.sRayOff);
      HydraLiteHit sh = hk_miss_hit();
      sh.t = length(spos - sam.pos) * 0.995f;
      sh = hk_traverse<true, false>(make_bvh_view(s.bvh, 0, s.tris, 0), s.haveInst != 0, spos, sdir, 0.0f, sh, st, tc);
      rays += 1.0f;
      const float shadow = (sh.primId != -1) ? 0.0f : 1.0f;
      ShadeContext sc;
      sc.l = sdir; sc.v = ray_dir * (-1.0f); sc.n = surf.normal; sc.tc = surf.texCoord; sc.fn = surf.flatNormal; sc.tg = surf.tangent; sc.bn = surf.biTangent;
      const BxDFResult ev = materialEval(mat, sc, s);
      const float cos1 = fmaxf(+dot(sdir, surf.normal), 0.0f), cos2 = fmaxf(-dot(sdir, surf.normal), 0.0f);
      const f3 bxdfVal = (ev.brdf * cos1) + (ev.btdf * cos2);
      float w = misWeightHeuristic(sam.pdf * pick, ev.pdfFwd);
      if (sam.isPoint) w = 1.0f;
      const f3 lc = sam.color * (1.0f / fmaxf(sam.pdf, HK_DEPSILON2));
      explicitColor = (((lc * (1.0f / pick)) * bxdfVal) * w) * shadow;
    }
    float rands[10];
    {
      const float4 r4 = rndFloat4_Pseudo(gen);
      rands[0] = r4.x; rands[1] = r4.y; rands[2] = r4.z;
      for (int k = 0; k < 7; k++) rands[3 + k] = rndFloat1_Pseudo(gen);
    }
    MatSample ms;
    MaterialSampleAndEvalBxDF(mat, rands, surf, ray_dir, flags, s, ms);
    const f3 bxdfVal = ms.color * (1.0f / fmaxf(ms.pdf, 1e-20f));
    const float cosTheta = fabsf(dot(ms.direction, surf.normal));
    ray_dir = ms.direction;
    ray_pos = OffsRayPos(surf.pos, surf.normal, ms.direction);
    misSpec = ((ms.flags & HRE_S) != 0 || (ms.flags & HRE_T) != 0);
    misPdf = ms.pdf;
    flags = flagsNextBounceLite(flags, ms, s);
    accumColor = accumColor + (thr * explicitColor);
    thr = thr * (bxdfVal * cosTheta);
  }
  accumColor = accumColor + (thr * currColor);
  return accumColor;
}
