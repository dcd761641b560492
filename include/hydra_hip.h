/* hydra_hip.h -- C-ABI of libhydra_hip.so, the MI355X (gfx950) wavefront path-tracing core.
 *
 * This is the drop-in boundary: an IHWLayer subclass in the reference driver
 * (hydra_drv/IHWLayer.h:97-246, factories :256-257, call site RenderDriverRTE.cpp:85-88) forwards each
 * virtual call to one of these entry points.  Plain C types only; every function returns 0 on success
 * or a negative HYDRA_HIP_E* code, with a text in hydra_hip_last_error().  The library never falls
 * back to a CPU path: without a usable HIP device hydra_hip_create fails.
 *
 * Blob/arena/BVH layouts are the reference's own (include/hydra_layouts.h).
 */
#ifndef HYDRA_HIP_H
#define HYDRA_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "hydra_layouts.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hydra_hip_ctx* hydra_hip_handle;

enum {
  HYDRA_HIP_OK = 0,
  HYDRA_HIP_EINVAL = -1,   /* bad argument / size mismatch                  */
  HYDRA_HIP_ENODEV = -2,   /* no usable HIP device                          */
  HYDRA_HIP_ENOMEM = -3,   /* device allocation failed                      */
  HYDRA_HIP_ESTATE = -4,   /* call order violated (e.g. pass before upload) */
  HYDRA_HIP_EDEVICE = -5   /* a HIP runtime call failed                     */
};

/* storage arena kinds -- names RenderDriverRTE::AllocAll passes to IHWLayer::CreateMemStorage
 * (hydra_drv/RenderDriverRTE.cpp:705-709) */
enum {
  HYDRA_STORAGE_TEXTURES = 0,
  HYDRA_STORAGE_TEXTURES_AUX = 1,
  HYDRA_STORAGE_GEOM = 2,
  HYDRA_STORAGE_MATERIALS = 3,
  HYDRA_STORAGE_PDFS = 4,
  HYDRA_STORAGE_KINDS = 5
};

/* ---------------------------------------------------------------- life cycle */
/* replaces CreateOclImpl/CreateCPUExpImpl (IHWLayer.h:256-257) */
int hydra_hip_create(int width, int height, int flags, int device_id, hydra_hip_handle* out);
int hydra_hip_destroy(hydra_hip_handle h);
const char* hydra_hip_last_error(hydra_hip_handle h);            /* valid for h == NULL too (create errors) */
int hydra_hip_device_name(hydra_hip_handle h, char* buf, int n); /* IHWLayer::GetDeviceName  :163 */
int hydra_hip_resize(hydra_hip_handle h, int width, int height); /* IHWLayer::ResizeScreen   :147 */
int hydra_hip_available_memory(hydra_hip_handle h, size_t* free_bytes, size_t* total_bytes); /* GetAvaliableMemoryAmount :152 */
int hydra_hip_finish(hydra_hip_handle h);                        /* IHWLayer::FinishAll      :137 */

/* ---------------------------------------------------------------- scene upload */
/* IHWLayer::PrepareEngineGlobals / PrepareEngineTables (:109-110): the assembled int blob
 * [EngineGlobals | tables | lights] (IHWLayerDataAssembler.cpp:149-170, 326-388). */
int hydra_hip_upload_globals(hydra_hip_handle h, const int32_t* blob, size_t words);
/* only the first `words` words changed (camera matrices, vars, flags); cheaper per-Draw refresh */
int hydra_hip_update_globals_header(hydra_hip_handle h, const int32_t* blob, size_t words);
/* contents of one IMemoryStorage arena (IMemoryStorage.h:16-49) */
int hydra_hip_upload_storage(hydra_hip_handle h, int kind, const void* data, size_t bytes);
/* IHWLayer::SetAllBVH4 (:116): one converted tree (IBVHBuilderAPI.h:7-33). Pointers are only read inside the call. */
int hydra_hip_upload_bvh(hydra_hip_handle h, int tree_id, const HydraBVHNode* nodes, int nodes_num,
                         const float* tri_f4, int tri_f4_num, const uint32_t* alpha_u2, int alpha_num,
                         int have_inst);
int hydra_hip_set_bvh_trees_num(hydra_hip_handle h, int trees_num);
/* IHWLayer::SetAllInstMatrices + SetAllInstLightInstId (:117-118); copied inside the call */
int hydra_hip_upload_instances(hydra_hip_handle h, const float* inv_matrices16, const int32_t* light_inst_id, int inst_num);
/* IHWLayer::SetAllRemapLists + SetAllInstIdToRemapId (:122-123) */
int hydra_hip_upload_remap_lists(hydra_hip_handle h, const int32_t* all_lists, int all_size,
                                 const int32_t* table_int2, int table_size,
                                 const int32_t* inst_to_remap_id, int inst_num);

/* ---------------------------------------------------------------- rendering */
/* image-plane tile partition for multi-GPU: tiles of tile_size x tile_size pixels are ordered along the
 * Morton (Z) curve of their tile coordinates and dealt round-robin: the i-th tile of that order belongs to rank i % world.
 * This context renders only its own tiles.  rank 0 / world 1 = the whole frame.  Changing the partition restarts the image
 * (generator states are kept per owned pixel). */
int hydra_hip_set_tile_partition(hydra_hip_handle h, int rank, int world, int tile_size);
/* which rank owns which tile under that rule: owner_of_tile[ty * tilesX + tx], tiles = tilesX * tilesY (host arithmetic only) */
int hydra_hip_tile_owners(int width, int height, int world, int tile_size, int32_t* owner_of_tile, int tiles);
/* What one rank's render state will hold for a frame size, partition and sampling choice -- host arithmetic only, no device
 * needed, the same routine trace_pass sizes its allocations with.  Returns HYDRA_HIP_EINVAL (text in err) when a limit is
 * exceeded: samples_in_flight * width * height < 2^32 (generator slots), samples_in_flight * owned pixels < 2^31 (path
 * slots).  Per-rank state scales with 1 / world: BASELINE configs[3] (3840x2160, 8 ranks, 512 samples in flight) plans
 * 531 M paths and ~134 GB per rank. */
typedef struct HydraStatePlan {
  int32_t samples_in_flight;     /* K in effect (0 in the request = chosen from the resolution) */
  int32_t pad_;
  int64_t owned_pixels, paths;   /* paths = owned_pixels * K, all in flight in one sub-pass */
  int64_t segments, segment_capacity;
  int64_t path_state_bytes, generator_bytes, contrib_bytes, owned_map_bytes, total_bytes;
} HydraStatePlan;
int hydra_hip_plan_render_state(int width, int height, int rank, int world, int tile_size, int samples_in_flight,
                                int queue_segments, int fused_bounce, HydraStatePlan* out, char* err, int err_len);
/* IHWLayer::SetExternalImageAccumulator (:199): use caller-owned device memory (float4 sums, w*h*16 B)
 * instead of the internal accumulator, e.g. a torch tensor that is later reduced over RCCL. NULL restores. */
int hydra_hip_set_external_accumulator(hydra_hip_handle h, void* dev_float4, size_t bytes);
int hydra_hip_init_path_tracing(hydra_hip_handle h, int seed);   /* IHWLayer::InitPathTracing :139 */
int hydra_hip_clear_accumulated_color(hydra_hip_handle h);       /* IHWLayer::ClearAccumulatedColor :140 */
/* IHWLayer::BeginTracingPass + EndTracingPass (:134-135): adds `spp` samples to every owned pixel.
 * Asynchronous on the context's stream. */
int hydra_hip_trace_pass(hydra_hip_handle h, int spp);
/* overwrite the accumulated sample count (after an external reduce of the accumulator) */
int hydra_hip_set_spp(hydra_hip_handle h, float spp);
float hydra_hip_get_spp(hydra_hip_handle h);                     /* IHWLayer::GetSPP :207 */
/* IHWLayer::GetHDRImage / GetLDRImage (:149-150): mean radiance, row-major, row 0 = bottom as generated;
 * size mismatch returns HYDRA_HIP_EINVAL and leaves the buffer untouched (CPUExpLayer.cpp:133-147). */
int hydra_hip_get_hdr_image(hydra_hip_handle h, float* rgba, int width, int height);
int hydra_hip_get_ldr_image(hydra_hip_handle h, uint32_t* rgba8, int width, int height);
/* the accumulated float4 SUMS themselves (not divided by the sample count): what IHWLayer::ContribToExternalImageAccumulator
 * (:201; GPUOCLLayerOther.cpp:365-429) adds into the shared accumulation image */
int hydra_hip_get_accumulator(hydra_hip_handle h, float* rgba_sums, int width, int height);
/* IHWLayer::GetRaysStat / ResetPerfCounters (:145,155): per-stage HIP-event times and exact ray counters */
int hydra_hip_get_rays_stat(hydra_hip_handle h, HydraRaysStat* out);
int hydra_hip_reset_perf_counters(hydra_hip_handle h);
/* enable per-stage hipEvent timing inside trace_pass (event records only; they are resolved by get_rays_stat) */
int hydra_hip_enable_stage_timing(hydra_hip_handle h, int enable);
/* per-bounce split of the same stage timers since the last reset: out = max_depth x 3 floats (ms):
 * [bounce][0 = closest-hit traversal | 1 = bounce kernel(s) | 2 = shadow traversal]  (traversal rate per bounce class) */
int hydra_hip_get_stage_times_per_bounce(hydra_hip_handle h, float* out, int max_depth);
/* Options.
 * Tuning knobs that never change results: "trace_mode" 1 = persistent traversal kernels with dynamic ray fetch (default),
 * 0 = one ray per lane; "trace_min_active" = refill threshold in lanes (default 48); "trace_blocks_per_cu" (default 12);
 * "shade_waves" 3|4|5 = register budget variant of the bounce kernels (default 3);
 * "shade_blocks_per_cu" (default 256), "static_blocks_per_cu" = grid caps; "shadow_unordered" (default 1): shadow (any-hit) rays take the children of a quad in stored order instead of near to far -- the answer of an any-hit query does not
 * depend on the order (trees deeper than the 80-entry stack aside, where the entries dropped differ), the sorting network is saved; 0 = the order of BVH4InstTraverseShadow; "queue_segments" 1..64 = independent path sub-queues
 * (default 32); "fused_bounce" 1 = one kernel per bounce (default), 0 = hit and shade kernels with an intermediate record;
 * "path_order" 1 = stream-major slots (default), 0 = pixel-major; "leaf_count_links" 1 = the device copy of the node array carries
 * triangle counts in its leaf links (default; read by the next upload_bvh, HYDRA_HIP_LEAF_COUNT_LINKS presets it).
 * "sort_paths" 1 = k_bounce groups the 256 paths of a workgroup by the shading class of the material they hit (labels travel in the hit
 * record; counting sort through LDS) from bounce "sort_paths_from_bounce" (default 1) on, 0 = paths are shaded in queue order;
 * "scene_tables_in_lds" 1 = k_bounce copies the material arena, the material / texture id tables and the lights into LDS when they fit 48 KB,
 * 2 (default) = also the geometry-id -> triangle-record table, the per-instance light ids and inverse matrices when those fit 16 KB, 0 = nothing;
 * "srgb_table" 1 = sRGB texel decode through a 256-entry table filled on the device by the decode function itself (default), 0 = powf per tap;
 * "top_tris_in_lds" 0..16 = size of the triangle pool the same kernels keep in LDS for the most visited leaves (default 0: measured to change nothing,
 * DESIGN.md; read by the next upload_bvh, HYDRA_HIP_TOP_TRIS presets it);
 * "top_quads_in_lds" 0..21 = how many of the most visited BVH quads the persistent traversal kernels keep in LDS (default 21;
 * read by the next upload_bvh, HYDRA_HIP_TOP_QUADS presets it).  HYDRA_HIP_TRACE_MODE / _MIN_ACTIVE / _BLOCKS_PER_CU in the environment preset the first three.
 * Sampling: "samples_in_flight" K = samples per pixel traced concurrently (1..512, 0 = chosen from the resolution:
 * 16 at 1080p).  Sample j of a trace_pass(spp) call draws from generator stream j % K of its pixel, stream k of pixel p
 * being RandomGenInit(seed + k * width * height + p) -- the per-slot seeding of the reference's wavefront layer
 * (shaders/trace.cl:6-13) with K * width * height slots.  K = 1 gives one persistent generator per pixel.  Changing K
 * restarts the image (generators are re-seeded).  K does not depend on the tile partition, so N ranks still sum to the
 * 1-rank frame. */
int hydra_hip_set_option(hydra_hip_handle h, const char* name, int value);
int hydra_hip_get_option(hydra_hip_handle h, const char* name, int* value);   /* "samples_in_flight" returns the K in effect */
/* algorithmic-work counters of the traversal kernels (roofline byte model).  While enabled, trace_pass
 * uses the counting kernel variants (slower).  out = max_depth x 2 x 5 uint64:
 * [bounce][0 = closest-hit | 1 = shadow][rays, quads visited, instance quads entered, leaves visited, triangles tested] */
int hydra_hip_enable_traversal_counters(hydra_hip_handle h, int enable);
int hydra_hip_get_traversal_counters(hydra_hip_handle h, uint64_t* out, int max_depth);
/* The traversal kernels read nodes and triangles through range-checked raw buffer loads, which answer an out-of-range
 * offset with zeros instead of faulting.  The counting variants also count every fetch that WOULD have been out of range,
 * summed here over all bounces since enable_traversal_counters(1): must be 0 for a well-formed tree (tests assert it). */
int hydra_hip_get_traversal_oob(hydra_hip_handle h, uint64_t* out);

/* ---------------------------------------------------------------- multi-GPU exchange
 * One process per GPU, each with a context whose tile partition is (rank, world).  The only exchange per frame is the float4
 * accumulator; these entry points do it over RCCL/xGMI without Python (RCCL is dlopen()ed on first use), on the context's own
 * stream behind the frame's kernels.  Reference precedent: every render process adds its frame into one shared image,
 * hydra_drv/GPUOCLLayerOther.cpp:365-429.
 *   rank 0:      hydra_hip_comm_unique_id(h, id)  -> ship the 128 bytes to the other processes (socket, file, MPI, ...)
 *   every rank:  hydra_hip_comm_init(h, id, rank, world)            (collective; rank/world = set_tile_partition's)
 *   per frame:   hydra_hip_comm_gather_frame(h, root)               (collective) then hydra_hip_finish; root's accumulator is the frame
 * gather_frame: every rank packs the sums of ITS pixels (1/world of the frame) and sends them to root, which drops them into its
 * accumulator -- supports are disjoint, nothing is added, the result has the bits of the one-GPU frame.  reduce_frame is the
 * ncclReduce(SUM) of the whole zero-padded frame instead (world times the bytes; kept for hosts that prefer one call on a caller-owned
 * accumulator).  With world == 1 both return at once. */
int hydra_hip_comm_unique_id(hydra_hip_handle h, char* id128);
int hydra_hip_comm_init(hydra_hip_handle h, const char* id128, int rank, int world);
int hydra_hip_comm_gather_frame(hydra_hip_handle h, int root);
int hydra_hip_comm_reduce_frame(hydra_hip_handle h, int root);
int hydra_hip_comm_destroy(hydra_hip_handle h);
/* test entry point: the pack and unpack kernels of comm_gather_frame applied to this rank's own pixels into a zeroed frame */
int hydra_hip_stage_pack_unpack(hydra_hip_handle h, float* rgba_frame, int width, int height);

/* ---- procedural textures.
 * Replaces IHWLayer::RecompileProcTexShaders (hydra_drv/IHWLayer.h:205; with RECOMPILE_PROCTEX_FROM_STRING, IHWLayer.h:341, its argument is the program text) as
 * GPUOCLLayer implements it (GPUOCLLayer.cpp:788-810: rebuild the texproc program, size the per-ray result buffer).  `source` is that text: shaders/texproc.cl
 * with the scene's functions after its '#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:' line and one generated call per texture after '#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:'
 * (RenderDriverRTE_ProcTex.cpp:446-629).  The two regions are cut out (they end at the reference file's next own line, or at '#HK_END_OF_PROCEDURAL_TEXTURES' /
 * '#HK_END_OF_PROCEDURAL_TEXTURES_EVAL'), compiled for gfx950 with hiprtc inside this library's frame (hydracore_amd/csrc/hk_proctex_rt.h) and run between the
 * closest-hit traversal and the bounce kernel of every bounce (where GPUOCLLayer::runKernel_ComputeHit runs ProcTexExec, GPUOCLKernels.cpp:662-690); results reach the
 * shading code as four halfs per texture, as in the reference (WriteProcTextureList, cglobals.h:2327-2359).  HYDRA_HIP_EINVAL with the compiler's log in
 * hydra_hip_last_error when the text does not compile.  source = NULL or length 0 drops the program.  A scene whose materials carry PLAIN_MATERIAL_HAVE_PROC_TEXTURES
 * cannot be traced without one (trace_pass / eval_gbuffer fail), and runs in the path tracer and the G-buffer pass (mmlt_begin refuses it).  Ambient-occlusion inputs (readAttr_AO) are 1. */
int hydra_hip_proctex_compile(hydra_hip_handle h, const char* source, size_t length);
/* the same compilation without a device or a context (hiprtc builds for gfx950 wherever it runs): HYDRA_HIP_OK, or HYDRA_HIP_EINVAL with the log in hydra_hip_last_error(NULL) */
int hydra_hip_proctex_check(const char* source, size_t length);
/* test entries: the compiled program on n hits handed in -- ids[max_num][n] (HYDRA_INVALID_TEXTURE ends a point's list; what follows it is undefined) and the
 * colours as halfs [max_num][n][4]; and the lists the following stage_shade_point / stage_bounce calls OF THE SAME n consult (n = 0 drops them) */
int hydra_hip_stage_proctex(hydra_hip_handle h, int n, int max_num, const float* ray_pos4, const float* ray_dir4, const HydraLiteHit* hits, int32_t* ids, uint16_t* halfs4);
int hydra_hip_stage_set_proctex(hydra_hip_handle h, int n, int max_num, const int32_t* ids, const uint16_t* halfs4);

/* ---------------------------------------------------------------- stage entry points
 * One call = one wavefront kernel over n host-provided items; used by the parity tests and by the
 * traversal roofline bench.  All pointers are HOST pointers; float4 arrays are n*4 floats.           */
/* P1  MakeRandEyeRay (cfetch.h:877-930): pixel (x,y) + 4 offsets in [-1,1] -> world ray */
int hydra_hip_stage_make_eye_rays(hydra_hip_handle h, int n, const int32_t* xy, const float* offs4,
                                  float* ray_pos4, float* ray_dir4);
/* T1  IntegratorCommon::rayTrace (CPUExp_Integrators_Common.cpp:122-154) -> Lite_Hit;
 * counters (optional, 3 uint32 per ray): quads visited, instance quads entered, triangles tested */
int hydra_hip_stage_trace(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                          HydraLiteHit* hits, uint32_t* counters3);
/* T2  IntegratorCommon::shadowTrace (Common.cpp:156-180) -> visibility 0/1 */
int hydra_hip_stage_shadow_trace(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                                 const float* t_far, float* visibility);
/* H1  kernel_EvalSurface (CPUExp_Integrators_PT_Loop.cpp:35-84) -> 24 floats per hit:
 * pos3 normal3 flatNormal3 tangent3 biTangent3 texCoord2 matId(as int bits) t sRayOff hfi(0/1) pad3 */
int hydra_hip_stage_eval_surface(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                                 const HydraLiteHit* hits, float* surf24);
/* one shading point with the random numbers handed in (parity fixtures for rows a/L1,L2,S1,S2): kernel_LightSelect +
 * kernel_LightSample (PT_Loop.cpp:141-168), materialEval towards the sample (cmaterial.h:2554-2628) and the BxDF sampling
 * of kernel_NextBounce (:218-256).  surf24 as written by stage_eval_surface; out28 = sample pos xyz, pdf, colour xyz,
 * pick prob, light offset (int), isPoint, brdf xyz, pdfFwd, btdf xyz, MatSample colour xyz, pdf, direction xyz, flags (int),
 * flagsNextBounceLite (int), 2 spare */
int hydra_hip_stage_shade_point(hydra_hip_handle h, int n, const float* surf24, const float* ray_dir4, const int32_t* flags,
                                const float* rnd_light4, const float* rands10, float* out28);
/* whole paths for n given primary rays with given per-path RandomGen state (2 uint32 each, updated in place), run
 * through the production wavefront kernels: IntegratorMISPTLoop2::PathTrace (PT_Loop.cpp:264-321) -> rgb, w = 0 */
/* test entry: the miss shader -- what a ray that leaves the scene brings back -- for n rays handed in.  With a back-plate named by the header (HRT_SHADOW_MATTE_BACK: a sky
 * light's or a shadow catcher's <back> texture) this is the OpenCL layer's environmentColorExtended (hydra_drv/cbidir.h:593-629, called by HitEnvOrLightKernel,
 * shaders/material.cl:354): suns of the header's table first, else environmentColor, replaced by the back texture (camera-projected by pixel, or spherical) for camera rays
 * and rays that only crossed transparent surfaces; without one, environmentColor (cbidir.h:492-533) as the CPU integrator calls it (CPUExp_Integrators_PT_Loop.cpp:28).
 * in8 per ray: origin xyz, previous BSDF pdf, previous bounce specular (0/1), ray flags, pixel x, pixel y (the last three as int bits).  out4: colour | 0. */
int hydra_hip_stage_environment(hydra_hip_handle h, int n, const float* ray_dir4, const float* in8, float* out4);

/* test entry: one accept / reject step of n Markov chains through the production kernel -- the acceptance min(1, f(y) / f(x)), the draw from the chain's second generator and
 * the two expected-value contributions (IntegratorMMLT::DoPass, hydra_drv/CPUExp_Integrators_MMLT.cpp:400-446; the wavefront layer's MMLTAcceptReject, shaders/mlt.cl:205-262,
 * whose inputs and outputs tests/golden/ref_mmlt_accept.npz holds).  old8 / new8 per chain: colour xyz, -, -, -, -, contribFunc of the current state / the proposal; gen2: the
 * generator states (in: before the draw, out: after).  out12: contribution at the old state xyz, its weight 1 - a; at the proposal xyz, its weight a (a pair is zero when the
 * contribution is below the reference's 1e-12 threshold); [8] = 1 when the chain took the proposal, [9] = the chain's acceptance counter. */
int hydra_hip_stage_mmlt_accept(hydra_hip_handle h, int n, const float* old8, const float* new8, uint32_t* gen2, float bk_scale, float* out12);

/* One bounce of n paths with every input handed in: the phases of the bounce kernel one after the other -- environment
 * (kernel_HitEnvironment, hydra_drv/CPUExp_Integrators_PT_Loop.cpp:23-33), emission + MIS (kernel_EvalEmission :86-139), light pick and sample with the
 * shadow ray (kernel_LightSelect / kernel_LightSample :141-168), next-event shading (kernel_Shade :181-216), BSDF sampling and the path-state
 * update (kernel_NextBounce :218-256, kernel_AddLastBouceContrib :258-262).  The reference's wavefront layer has a kernel per phase
 * (shaders/material.cl:301 HitEnvOrLightKernel, :578 Shade, :756 NextBounce; shaders/light.cl:140 LightSample): tests/golden/ref_stage_*.npz
 * holds their inputs and outputs, which this entry is checked against.
 * surf24 = the record of hydra_hip_stage_eval_surface (matId < 0: the ray left the scene).  in16 per path: throughput xyz, previous BSDF pdf,
 * radiance so far xyz, previous bounce was specular (0/1), the light's four random numbers (rndLight), the number that picks the light (the CPU
 * path uses the third of the four), visibility of the shadow ray, Lite_Hit.instId and the ray flags (int bits).  rands10 = RndMatAll's numbers.
 * out40: [0..2] environment or emitted radiance after MIS, [3] (int) 1 = left the scene, 2 = ended on an emitter, 4 = ended at the depth limit, 0 = goes on;
 * [4..6] light sample position, [7] pdf, [8..10] radiance, [11] isPoint, [12] pick probability, [13] (int) light offset, [14..16] shadow ray origin,
 * [17] its far end, [18..20] its direction; [21..23] next-event contribution (x visibility); [24..26] next ray origin, [27..29] direction, [30] (int) flags,
 * [31..33] throughput, [34..36] radiance carried on (ended paths: the path's final radiance), [37] BSDF pdf of the sample, [38] sample was specular. */
int hydra_hip_stage_bounce(hydra_hip_handle h, int n, int depth, int max_depth, const float* ray_pos4, const float* ray_dir4, const float* surf24, const float* in16,
                           const float* rands10, float* out40);
int hydra_hip_stage_path_trace(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                               uint32_t* rng_state2, float* color4);
/* Row f3 (MMLT / SBDPT, hydra_drv/CPUExp_Integrators_MMLT.cpp), first milestone: the building blocks one call at a time.
 * LightSampleForward (clight.h:1064-1110): light id + 4 randoms -> out16 = pos xyz, dir xyz, normal xyz, colour xyz, pdfA, pdfW, cosTheta, isPoint */
int hydra_hip_stage_light_sample_forward(hydra_hip_handle h, int n, const int32_t* light_ids, const float* rands4, float* out16);
/* lightPdfFwd (clight.h:1117-1175): light id + cosine at the light -> out4 = pdfA, pdfW, pick probability (forward), 0 */
int hydra_hip_stage_light_pdf_fwd(hydra_hip_handle h, int n, const int32_t* light_ids, const float* cos_theta, float* out4);
/* CameraImageToSurfaceFactor + worldPosToScreenSpace (cbidir.h:78-131): surface point, normal (float4 each), lens offset (2 floats) ->
 * out8 = image-to-surface factor, direction to the camera xyz, distance, screen x, screen y, 0 */
int hydra_hip_stage_camera_connect(hydra_hip_handle h, int n, const float* pos4, const float* norm4, const float* disk2, float* out8);
/* MutateKelemen (crandom.h:189-210): primary-space values + 2 randoms each, step parameters p2 < p1 (defaults 64, 1024) */
int hydra_hip_stage_mutate_kelemen(hydra_hip_handle h, int n, const float* values, const float* rands2, float p2, float p1, float* out);
/* ---- GPU-side BVH build (row f2).  The reference builds through IBVHBuilder2 (hydra_drv/IBVHBuilderAPI.h:35-68: InstanceTriangleMeshes /
 * CommitScene / ConvertMap, Embree 2.17 behind it); this entry builds the tree of ONE mesh on the device -- Morton codes, a hand-written stable
 * radix sort, a binary tree over the sorted triangles (see hydra_hip_bvh_build_mesh_ex), level-synchronous collapse to 4-wide nodes with leaves of <= leaf_max
 * triangles -- and returns it in build form: node 0 is the root; an inner node has count == 0 and up to four children (-1 = none), a leaf owns
 * prim_order[first .. first + count).  Degenerate triangles are dropped (bvh_access_dll2.cpp:354-355).  Emission into the reference's quad /
 * triangle-list layout (ConvertMap, bvh_access_dll2.cpp:604-717) is the host builder's (hydracore_amd/host/bvh4_builder.cpp), which its own
 * SAH build shares.  Needs no layer handle; nodes_out holds up to 2 x triangles nodes, prim_order_out up to `triangles` ids. */
typedef struct HydraBuildNode { float boxMin[3]; int32_t first; float boxMax[3]; int32_t count; int32_t child[4]; } HydraBuildNode;
int hydra_hip_bvh_build_mesh(int device, const float* vert4f, int num_vert, const int32_t* indices, int num_indices, int leaf_max,
                             HydraBuildNode* nodes_out, int32_t* node_count_out, int32_t* prim_order_out, int32_t* prim_count_out, float* build_ms_out);
/* The same with the way the binary tree under the collapse is made chosen: HYDRA_BVH_LBVH = the binary radix tree of the Morton codes alone (Karras
 * 2012); HYDRA_BVH_PLOC (what hydra_hip_bvh_build_mesh uses, radius 128) = parallel locally-ordered clustering over the Morton order (Meister & Bittner
 * 2018): every round each cluster merges with the neighbour within `radius` places whose union with it has the smallest surface area, if that neighbour
 * chose it too.  radius 1..128 (larger: better trees, slower build). */
enum { HYDRA_BVH_LBVH = 0, HYDRA_BVH_PLOC = 1 };
int hydra_hip_bvh_build_mesh_ex(int device, const float* vert4f, int num_vert, const int32_t* indices, int num_indices, int leaf_max, int method, int radius,
                                HydraBuildNode* nodes_out, int32_t* node_count_out, int32_t* prim_order_out, int32_t* prim_count_out, float* build_ms_out);
const char* hydra_hip_bvh_last_error(void);
/* IHWLayer::NormalMapFromDisplacement (hydra_drv/IHWLayer.h:197; host form CPUSharedData::NormalMapFromDisplacement + BilateralFilter,
 * CPUBilateralFilter2D.cpp:15-246; GPUOCLLayer's kernels GPUOCLData.cpp:549-640): the RGBA8 height map of a <displacement type="height_bump"> becomes
 * the RGBA8 normal map the shading's BumpMapping reads (xy = 0.5 + 0.5 n, z = n.z, w = height).  bump_amt = 0.5 x the XML amount, smooth_lvl = 10 x the
 * XML smooth level (RenderDriverRTE_AuxTextures.cpp:11-31); smooth_lvl >= 1 adds the 11 x 11 bilateral filter.  rgba_in / rgba_out: w*h*4 bytes in
 * host memory.  No layer handle: the driver calls it while it packs materials.  device_ms_out (may be null): device time of the kernels. */
/* The two multi-scattering energy tables of the globals header, baked on the device: ggx4096 = EngineGlobals::m_essGgx2017Table (u16 [64 roughness][64 dot(N,V)]),
 * transp262144 = m_essTranspTable (u16 [64 ior][64][64]) (hydra_drv/cfetch.h:77-79).  The reference copies offline-baked data into the header when a
 * layer is constructed (hydra_drv/IHWLayer.h:101, getGgxTable / getTranspTable of bakeBrdfEnergy/MSTables*.cpp, made by bakeBrdfEnergy/bakeBrdf.cpp);
 * this is the same integrand over the same cells integrated with a fixed point set (csrc/hydra_bake.hip).  No layer handle; baked once per process. */
int hydra_hip_bake_energy_tables(int device, uint16_t* ggx4096, uint16_t* transp262144, float* device_ms_out);
const char* hydra_hip_bake_last_error(void);
int hydra_hip_normal_map_from_displacement(int device, int w, int h, const uint8_t* rgba_in, float bump_amt, int inv_height, float smooth_lvl, uint8_t* rgba_out, float* device_ms_out);
const char* hydra_hip_image_last_error(void);
/* ---- IntegratorMMLT (row f3; hydra_drv/CPUExp_Integrators_MMLT.cpp): multiplexed MLT over the simplified bidirectional sampler --------------
 * The reference runs 8 chains (one per OpenMP thread, :583-585) of width*height mutations per pass; here every chain is a GPU thread and a
 * pass advances all of them together: mutate (MutatePrimarySpace :93-144), F (:146-315) through the traversal kernels, accept / reject with
 * the two expected-value contributions (:379-447, atomic adds into one float4 image).  The clock()-driven stirring of the generators
 * (:97-103, :362-368) is not reproduced: a run is a function of (scene, chains, seed).
 * mmlt_begin: DoPassEstimateAvgBrightness (:463-520, estimate_passes x chains samples per path length; 0 = 4 passes), then for every chain
 *   its path length d ~ average brightness (:348-356), a fresh sample (InitialSamplePS :51-57) and F of it.  first_bounce / max_depth 0 =
 *   varsI[HRT_MMLT_FIRST_BOUNCE] clamped to 2..3 (:481-483) / varsI[HRT_TRACE_DEPTH]; shorter paths are the direct-light pass's
 *   (DoPassDirectLight :522-546 = the path tracer of this layer at trace depth first_bounce - 1).
 * mmlt_pass: `mutations` steps of every chain.  mmlt_get_image: kScale x indirect image (GetImageHDR :616-635, EstimateScaleCoeff :548-552),
 *   info8 = average brightness, kScale, acceptance rate, mutations so far, chains, first bounce, max depth, 0.
 * mmlt_get_state (test hook): chain planes [11][chains] (y, colour, pixel, two generators, accepted), d per chain, current x vectors as rows of
 *   12 + 10 * max_depth floats, average brightness per path length [max_depth + 1]; any pointer may be null. */
int hydra_hip_mmlt_begin(hydra_hip_handle h, int chains, int seed, int first_bounce, int max_depth, int estimate_passes);
int hydra_hip_mmlt_pass(hydra_hip_handle h, int mutations);
int hydra_hip_mmlt_get_image(hydra_hip_handle h, float* image4, int width, int height, float* info8);   /* image4 (may be null) = width*height float4, must be the frame of mmlt_begin */
/* the indirect image restarts from zero while the chains go on (GPUOCLLayer::ClearAccumulatedColor, GPUOCLLayer.cpp:1288-1297, leaves the MLT state
 * alone): what a contribution to a shared accumulation image (IHWLayer::ContribToExternalImageAccumulator :201) needs after it has taken the image */
int hydra_hip_mmlt_reset_image(hydra_hip_handle h);
int hydra_hip_mmlt_get_state(hydra_hip_handle h, float* chains, int32_t* depth, float* xrows, float* avg_b);
int hydra_hip_mmlt_end(hydra_hip_handle h);
/* IntegratorSBDPT::DoPass (hydra_drv/CPUExp_Integrators_SBDPT.cpp:11-216) on the buffers and generators of the MMLT run: `passes` x chains
 * samples, each with a path length drawn uniformly from 2..max_depth (:21), a fresh primary-sample vector, F (the reference's SBDPT keeps its own
 * copies of the sub-path / connection / MIS code of MMLT), and a splat weighted by the selector's (d + 1)(max_depth - 1) (:24).  Differs from the
 * reference in where the split and the pixel come from (x[MMLT_DIM_SPLIT] and the lens dimensions instead of two rndInt draws, :22, :40-41).
 * sbdpt_get_image: splats x width*height / samples (paths of 2..max_depth segments; directly visible emitters are not part of this pass). */
int hydra_hip_sbdpt_pass(hydra_hip_handle h, int passes);
int hydra_hip_sbdpt_get_image(hydra_hip_handle h, float* image4, int width, int height, double* samples);
/* IHWLayer::EvalGBuffer (hydra_drv/IHWLayer.h:136; GPUOCLLayer::EvalGBuffer, GPUOCLLayerOther.cpp:694-870), following the CPU restatement
 * IntegratorCommon::gbufferEval / gbufferSample (hydra_drv/CPUExp_GBuffer.cpp:15-113): per pixel 64 Hammersley-placed primary rays, one
 * surface sample each (depth, normal, diffuse colour, material / object / instance id, texture coordinate), the sample most similar to all
 * others wins (gbuffDiff, cglobals.h:2193-2205) and carries the share of samples like it as coverage.  data1 / data2 = width*height float4
 * each, the two layers the reference writes into the shared accumulation image (packGBuffer1 / packGBuffer2, cglobals.h:2098-2145);
 * inst_remap (may be null) = a_instIdByInstId, applied to the instance id (:846-853).  raw14 (may be null; tests) = per pixel depth, normal xyz,
 * rgba, matId (int bits), coverage, texCoord xy, objId, instId (int bits).  Alpha is 0 on every hit, as in the CPU form.
 * Needs the camera in the globals header (SetCamMatrices + PrepareEngineGlobals); width x height must be the layer's frame and the header's
 * HRT_WIDTH_F x HRT_HEIGHT_F (GetHDRImage's size check, CPUExpLayer.cpp:133-147, made loud). */
int hydra_hip_eval_gbuffer(hydra_hip_handle h, float* data1, float* data2, int width, int height, const int32_t* inst_remap, int inst_remap_size, float* raw14);
/* IntegratorMMLT::F (hydra_drv/CPUExp_Integrators_MMLT.cpp:146-315; sub-paths :637-929, connections :931-1047 + cbidir.h:190-477): the
 * contribution of n primary-sample vectors.  xvec = n rows of `stride` floats laid out as the reference's PSSampleV (cglobals.h:102-128:
 * lens 0..3, light 4..10, split 11, then 10 floats per bounce, light part first), depth[i] = d (path length in segments, 1..16),
 * stride >= 12 + 10 * d.  out8 per vector = colour xyz (MIS-weighted), pixel x, y, split s, MIS weight, contribFunc(colour).
 * Runs the traversal kernels of the path tracer between per-chain stage kernels (2n rays per level, n + n connection rays). */
int hydra_hip_stage_mmlt_f(hydra_hip_handle h, int n, const int32_t* depth, const float* xvec, int stride, float* out8);
/* R1  RandomGenInit + rndFloat4_Pseudo (crandom.h:20-63): for each seed the first `draws` float4 outputs */
int hydra_hip_stage_random(hydra_hip_handle h, int n, const int32_t* seeds, int draws, float* out4, uint32_t* state2);

/* traversal replay for the roofline number: rays are uploaded once, then the closest-hit kernel is
 * launched `iters` times between two hipEvents on the context stream; returns average ms per launch. */
int hydra_hip_bench_trace(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                          int iters, int shadow, float* avg_ms);

/* the persistent COUNTING traversal kernels (k_trace_dyn<*, true>: what bench.py prices the roofline bytes with) on n
 * caller-provided rays: closest hit (t_far == NULL; IntegratorCommon::rayTrace, Common.cpp:122-154) or any-hit shadow rays
 * (t_far[n]; Common.cpp:156-180, early-out form ctrace.h:1065-1294).  totals6 = rays traced, quads visited, instance quads
 * entered, leaves visited, triangles tested, out-of-range fetches (see hydra_hip_get_traversal_oob). */
int hydra_hip_stage_trace_totals(hydra_hip_handle h, int n, const float* ray_pos4, const float* ray_dir4,
                                 const float* t_far, uint64_t* totals6);

#ifdef __cplusplus
}
#endif
#endif
