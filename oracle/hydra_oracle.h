/* hydra_oracle.h -- CPU oracle: plain-C restatement of the reference's unidirectional MIS path tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed from the product
 * (hydracore_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the reported CPU baseline.
 *
 * Follows, function by function, hydra_drv/CPUExp_Integrators_PT_Loop.cpp (IntegratorMISPTLoop2),
 * CPUExp_Integrators_Common.cpp (DoPass, rayTrace, shadowTrace) and the inline kernels in hydra_drv/c*.h; each
 * function in hydra_oracle.c cites the file:line it restates.
 *
 * PINNING: the reference's C++ path cannot be compiled in this image (hydra_drv/cglobals.h:324-327 includes
 * HydraAPI's LiteMath.h / HR_HDRImage.h, which are absent, and stand-ins are not allowed), and the reference ships no
 * golden vectors for this path.  The restatement is therefore pinned against the reference's own OpenCL build of the
 * same inline functions (oracle/_ref/<kernel>.hsaco, built by oracle/build_ref.sh from hydra_drv/shaders/<kernel>.cl and run on
 * the GPU box; vectors committed under tests/golden/).  Anything not covered by those vectors is "parity unpinned".
 */
#ifndef HYDRA_ORACLE_H
#define HYDRA_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcScene {
  const int32_t* globals;        /* [EngineGlobals | tables | lights] blob            */
  const float*   matStorage;     /* materials arena (float4 units addressable)        */
  const int32_t* texStorage;     /* textures arena                                    */
  const float*   geomStorage;    /* geometry arena                                    */
  const float*   pdfStorage;     /* pdf tables arena (unused by the supported subset) */
  const float*   bvh;            /* tree 0 nodes, 8 floats per node                   */
  const float*   tris;           /* tree 0 triangle float4 list                       */
  int32_t        haveInst;
  const float*   instMatrices;   /* 16 floats per instance, world -> object, columns  */
  const int32_t* instLightInstId;
  int32_t        instNum;
  const int32_t* remapLists;  int32_t remapListsSize;
  const int32_t* remapTable;  int32_t remapTableSize;   /* int2 entries */
  const int32_t* remapInst;   int32_t remapInstSize;
  /* trees 1..3 of a multi-tree scene (IntegratorCommon::rayTrace walks all of them, Common.cpp:128-150) and the alpha tables */
  int32_t        treesNum;       /* 0 or 1: tree 0 only */
  const float*   bvhN[3];
  const float*   trisN[3];
  int32_t        haveInstN[3];
  const uint32_t* alpha[4];      /* per tree: uint2 per float4 of the triangle list + opacity samplers, NULL = none */
  const int32_t* texAuxStorage;  /* the second texture arena (normal maps), NULL = none */
} OrcScene;

typedef struct OrcHit { float t; int32_t primId, instId, geomId; } OrcHit;

/* R1 */
void orc_random_init(int32_t seed, uint32_t state[2]);
uint32_t orc_next_state(uint32_t state[2]);
void orc_rnd_float4(uint32_t state[2], float out[4]);
float orc_rnd_float1(uint32_t state[2]);

/* P1: xy = 2 ints per ray, offs4 = 4 floats in [-1,1] */
void orc_make_eye_rays(const OrcScene* s, int n, int w, int h, const int32_t* xy, const float* offs4, float* pos4, float* dir4);
/* T1 (counters3 optional: quads visited, instance quads entered, triangles tested; leaves = optional 4th array NULL) */
void orc_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, OrcHit* hits, uint32_t* counters3, uint32_t* leaves1);
/* T2 */
void orc_shadow_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, const float* tfar, float* vis);
void orc_shadow_trace_anyhit(const OrcScene* s, int n, const float* pos4, const float* dir4, const float* tfar, float* vis, uint32_t* c4);
/* H1: 24 floats per hit (layout in include/hydra_hip.h, hydra_hip_stage_eval_surface) */
void orc_eval_surface(const OrcScene* s, int n, const float* pos4, const float* dir4, const OrcHit* hits, float* surf24);
/* whole paths, per-path RandomGen state (updated in place); color4.w = number of rays traced (extension + shadow) */
/* one shading point with the random numbers handed in: light pick + light sample + materialEval + BxDF sampling
   (layout of surf24 = orc_eval_surface output; out28 documented in hydra_oracle.c) */
void orc_shade_point(const OrcScene* s, int n, const float* surf24, const float* dir4, const int32_t* flags, const float* rndLight4,
                     const float* rands10, float* out28);
void orc_path_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, uint32_t* rng2, float* color4);
/* one bounce of n paths with every input handed in: the kernel_* stages of PT_Loop.cpp:9-262 in order (layouts: include/hydra_hip.h, hydra_hip_stage_bounce) */
/* procedural textures: the per-point lists (ids[max_num][n], colours as halfs [max_num][n][4]; HYDRA's INVALID_TEXTURE ends a list) the following orc_stage_bounce
 * calls OF THE SAME n consult -- the oracle restates the consumer of the lists (readProcTex in sample2DExt), not the scene's functions; n = 0 drops them */
/* the miss shader for n rays handed in (include/hydra_hip.h, hydra_hip_stage_environment): with a back-plate in the header the OpenCL layer's environmentColorExtended
 * (cbidir.h:593-629), else environmentColor */
void orc_stage_environment(const OrcScene* s, int n, const float* dir4, const float* in8, float* out4);
/* whole frames: the scene's functions built for the host by the test (tests/proctex_host.py), called by PathTrace for every hit on a material with procedural textures:
 * fn(user, surf19 = wp lp n tg bn tc0 ao ao2, material head, ray direction, &count, ids[16], colours[48]); NULL = none (the lists stay empty) */
void orc_set_proctex_eval(void* fn, void* user);
void orc_stage_set_proctex(int n, int max_num, const int* ids, const uint16_t* halfs4);
/* one accept / reject step of n Markov chains handed in (include/hydra_hip.h, hydra_hip_stage_mmlt_accept) */
void orc_stage_mmlt_accept(int n, const float* old8, const float* new8, uint32_t* gen2, float bkScale, float* out12);
void orc_stage_bounce(const OrcScene* s, int n, int depth, int maxDepth, const float* pos4, const float* dir4, const float* surf24, const float* in16,
                      const float* rands10, float* out40);

/* P0: one sample for every pixel owned by (rank, world, tile) -- IntegratorCommon::DoPass with per-pixel generators.
 * gens: 2 uint32 per pixel, image: float4 running mean (reference semantics) OR sums when sum_mode != 0.
 * Returns the number of rays traced (extension + shadow). threads <= 0: OpenMP default. */
uint64_t orc_render_pass(const OrcScene* s, int w, int h, uint32_t* gens, float* image4, int spp_done, int sum_mode,
                         int rank, int world, int tile, int threads);
void orc_init_generators(int w, int h, int seed, uint32_t* gens);
/* dump the live rays of a given bounce (0 = primary) for the first sample of every pixel: returns count */
int64_t orc_collect_rays(const OrcScene* s, int w, int h, int seed, int bounce, int shadow, float* pos4, float* dir4, float* tfar, int64_t cap);
/* f3 building blocks (MMLT / SBDPT): clight.h:1064-1110, :1117-1175, cbidir.h:78-131, crandom.h:189-210 */
void orc_light_sample_forward(const OrcScene* s, int n, const int32_t* lightIds, const float* rands4, float* out16);
void orc_light_pdf_fwd(const OrcScene* s, int n, const int32_t* lightIds, const float* cosTheta, float* out4);
void orc_camera_connect(const OrcScene* s, int n, const float* pos4, const float* norm4, const float* disk2, float* out8);
void orc_mutate_kelemen(int n, const float* values, const float* rands2, float p2, float p1, float* out);
/* IntegratorMMLT::F (CPUExp_Integrators_MMLT.cpp:146-315): n primary-sample vectors of `stride` floats, path length depth[i] -> out8 = colour, x, y, split, MIS weight, contribFunc */
void orc_mmlt_f(const OrcScene* s, int n, const int32_t* depth, const float* xvec, int stride, float* out8);
/* n Markov chains of IntegratorMMLT (DoPassIndirectMLT :358-461): `mutations` mutate / F / accept steps each from the given states */
void orc_mmlt_run(const OrcScene* s, int n, uint32_t* gens4, const int32_t* depth, int mutations, int w, float* image4, float* chains6, float* xrows, int stride, int32_t* accepted);
/* IntegratorSBDPT::DoPass through F: n samples, generator of sample i = gens4[i][0..1] (advanced), image4 += splats */
void orc_sbdpt_pass(const OrcScene* s, int n, uint32_t* gens4, int maxDepth, int w, float* image4);
/* IntegratorCommon::gbufferEval (CPUExp_GBuffer.cpp:15-113) for the pixel window [x0, x0+nx) x [y0, y0+ny): packGBuffer1 / packGBuffer2 per pixel (+ the unpacked record) */
void orc_gbuffer(const OrcScene* s, int width, int height, int x0, int y0, int nx, int ny, float* data1, float* data2, float* raw14);
/* CPUSharedData::NormalMapFromDisplacement (CPUBilateralFilter2D.cpp:101-246): RGBA8 height map -> RGBA8 normal map; parity unpinned (see the .c file) */
void orc_normal_map_from_displacement(int w, int h, const uint8_t* rgba_in, float bumpAmt, int invHeight, float smoothLvl, uint8_t* rgba_out);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
