// hip_layer.cpp -- HipHWLayer: the IHWLayer subclass that forwards to the C-ABI of libhydra_hip.so.
//
// Counterpart of GPUOCLLayer (hydra_drv/GPUOCLLayer.h:28 derives from CPUSharedData the same way) for the PT path.
// C-ABI error codes are turned into RUN_TIME_ERROR exceptions, which is how the reference reports fatal layer errors
// (hydra_drv/globals_sys.h:56-66, caught in hydra_app/main.cpp:331-338).
#include "hw_layer.h"
#include "../../include/hydra_hip.h"
#include <cstring>

namespace hydra_host {

class HipHWLayer : public SharedDataLayer {
public:
  HipHWLayer(int w, int h, int a_flags, int a_deviceId) : SharedDataLayer(w, h, a_flags), m_h(nullptr), m_deviceId(a_deviceId) {
    const int rc = hydra_hip_create(w, h, a_flags, a_deviceId, &m_h);
    if (rc != HYDRA_HIP_OK) RunTimeError(std::string("CreateHipImpl: ") + hydra_hip_last_error(nullptr));
    hydra_hip_device_name(m_h, m_devName, sizeof(m_devName));
    // the reference's layers fill the header's energy tables from baked data in their constructor (IHWLayer.h:101); this one bakes them
    // on its device (once per process, milliseconds)
    std::vector<uint16_t> ess(4096 + 262144);
    if (hydra_hip_bake_energy_tables(a_deviceId, ess.data(), ess.data() + 4096, nullptr) != HYDRA_HIP_OK)
      RunTimeError(std::string("CreateHipImpl: ") + hydra_hip_bake_last_error());
    SetEnergyTables(ess.data(), ess.data() + 4096);
  }
  ~HipHWLayer() override { if (m_h) hydra_hip_destroy(m_h); }

  bool HasDevice() const override { return true; }
  bool StoreCPUData() const override { return false; }
  const char* GetDeviceName(int* pOCLVer = nullptr) const override { if (pOCLVer) *pOCLVer = 0; return m_devName; }

  void ResizeScreen(int w, int h, int a_flags) override {
    SharedDataLayer::ResizeScreen(w, h, a_flags);
    check(hydra_hip_resize(m_h, w, h), "ResizeScreen");
  }

  void PrepareEngineGlobals() override {
    SharedDataLayer::PrepareEngineGlobals();
    if (m_cdataPrepared.empty()) RunTimeError("HipHWLayer: EngineGlobals were not prepared");
    if (m_tablesUploaded)   // per-Draw refresh: camera matrices, vars, flags, table offsets (first 1268 words)
      check(hydra_hip_update_globals_header(m_h, m_cdataPrepared.data(), HG_TABLES_READY + 1), "PrepareEngineGlobals");
  }

  void PrepareEngineTables() override {
    SharedDataLayer::PrepareEngineTables();
    static const char* names[HYDRA_STORAGE_KINDS] = {"textures", "textures_aux", "geom", "materials", "pdfs"};
    for (int k = 0; k < HYDRA_STORAGE_KINDS; k++) {
      IMemoryStorage* st = FindStorage(names[k]);
      if (st == nullptr) RunTimeError(std::string("HipHWLayer::PrepareEngineTables: no storage ") + names[k]);
      check(hydra_hip_upload_storage(m_h, k, st->GetBegin(), st->GetSize()), names[k]);
    }
    check(hydra_hip_upload_globals(m_h, m_cdataPrepared.data(), m_cdataPrepared.size()), "upload_globals");
    m_tablesUploaded = true;
  }

  void SetAllBVH4(const ConvertionResult& cr, IBVHBuilder2* a_builder, int a_flags) override {
    SharedDataLayer::SetAllBVH4(cr, a_builder, a_flags);   // host copy for debugging / CPU cross-checks
    for (int i = 0; i < cr.treesNum; i++)
      check(hydra_hip_upload_bvh(m_h, i, cr.pBVH[i], cr.nodesNum[i], cr.pTriangleData[i], cr.trif4Num[i],
                                 cr.pTriangleAlpha[i], cr.triAfNum[i], m_bvhTrees[i].haveInst ? 1 : 0), "SetAllBVH4");
    check(hydra_hip_set_bvh_trees_num(m_h, cr.treesNum), "SetAllBVH4");
  }

  void SetAllInstMatrices(const float4x4* a_matrices, int32_t n) override {
    SharedDataLayer::SetAllInstMatrices(a_matrices, n);
    upload_instances();
  }
  void SetAllInstLightInstId(const int32_t* ids, int32_t n) override {
    SharedDataLayer::SetAllInstLightInstId(ids, n);
    upload_instances();
  }
  void SetAllRemapLists(const int* a_allLists, const int2* a_table, int a_allSize, int a_tableSize) override {
    SharedDataLayer::SetAllRemapLists(a_allLists, a_table, a_allSize, a_tableSize);
    upload_remap();
  }
  void SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) override {
    SharedDataLayer::SetAllInstIdToRemapId(a_allInstId, a_instNum);
    upload_remap();
  }

  void InitPathTracing(int seed, std::vector<int32_t>* = nullptr) override { m_mmltSeed = seed; MLT_Free(); check(hydra_hip_init_path_tracing(m_h, seed), "InitPathTracing"); }
  // a user-requested clear restarts everything, the Markov chains included (RenderDriverRTE calls this with InitPathTracing when the scene or
  // the camera changed, RenderDriverRTE.cpp:1738,1776); the per-pass contribution to a shared image clears the sums only (below)
  void ClearAccumulatedColor() override { MLT_Free(); check(hydra_hip_clear_accumulated_color(m_h), "ClearAccumulatedColor"); }
  // HRT_ENABLE_MMLT (cglobals.h:419, set by RenderDriverRTE::UpdateSettings for method_secondary = "mmlt", RenderDriverRTE.cpp:196-202):
  // the pass is the reference layer's DL_Pass + MMLT_Pass (GPUOCLLayer.cpp:1368-1375): the path tracer limited to the paths
  // shorter than HRT_MMLT_FIRST_BOUNCE for the direct part, then NUM_MMLT_PASS (= 32, GPUOCLLayer.h:681) mutations of every chain.
  // Chains: the init flags' GPU_MMLT_THREADS_* (GPUOCLLayer.cpp:859-870), 524 288 without one.
  void BeginTracingPass() override {
    if ((m_vars.m_flags & HF_ENABLE_MMLT) == 0) { check(hydra_hip_trace_pass(m_h, GetRaysPerPixel()), "BeginTracingPass"); return; }
    int first = m_vars.m_varsI[HV_I_MMLT_FIRST_BOUNCE];
    first = first > 3 ? 3 : (first < 2 ? 2 : first);
    const int maxDepth = m_vars.m_varsI[HV_I_TRACE_DEPTH];
    if (!m_tablesUploaded || m_cdataPrepared.empty()) RunTimeError("HipHWLayer::BeginTracingPass(MMLT): EngineGlobals were not prepared");
    {   // direct part: paths of fewer than `first` segments
      std::vector<int32_t> hdr(m_cdataPrepared.begin(), m_cdataPrepared.begin() + HG_TABLES_READY + 1);
      hdr[HG_VARS_I + HV_I_TRACE_DEPTH] = first - 1;
      check(hydra_hip_update_globals_header(m_h, hdr.data(), HG_TABLES_READY + 1), "BeginTracingPass(MMLT direct)");
      check(hydra_hip_trace_pass(m_h, GetRaysPerPixel()), "BeginTracingPass(MMLT direct)");
      check(hydra_hip_update_globals_header(m_h, m_cdataPrepared.data(), HG_TABLES_READY + 1), "BeginTracingPass(MMLT direct)");
    }
    if (!m_mmltRunning) {
      int chains = 524288;
      if (m_initFlags & 65536) chains = 262144; else if (m_initFlags & 65536 * 2) chains = 131072; else if (m_initFlags & 65536 * 4) chains = 65536; else if (m_initFlags & 65536 * 8) chains = 16384;
      check(hydra_hip_mmlt_begin(m_h, chains, m_mmltSeed, first, maxDepth, 0), "BeginTracingPass(MMLT begin)");
      m_mmltRunning = true;
    }
    check(hydra_hip_mmlt_pass(m_h, 32), "BeginTracingPass(MMLT)");
  }
  bool   MLT_IsAllocated() const override { return m_mmltRunning; }
  size_t MLT_Alloc(int, int, int) override { return 0; }   // the run allocates with its first pass, when the chain count and path lengths are known
  void   MLT_Free() override { if (m_mmltRunning) { hydra_hip_mmlt_end(m_h); m_mmltRunning = false; } }
  // with a shared accumulation image attached, every pass ends by adding its samples to it (the reference's layers do
  // this inside their per-pass contribution, GPUOCLLayerOther.cpp:259-283)
  void EndTracingPass() override { if (m_pExternalImage != nullptr) ContribToExternalImageAccumulator(m_pExternalImage); }
  // IHWLayer.h:205 with RECOMPILE_PROCTEX_FROM_STRING (IHWLayer.h:341): the argument is the program text RenderDriverRTE::EndTexturesUpdate assembled
  // from shaders/texproc.cl and the scene's data/proctex_*.c; GPUOCLLayer rebuilds its OpenCL program from it (GPUOCLLayer.cpp:788-810), this layer cuts
  // the scene's functions and the generated calls out of it and builds them for gfx950 (hydra_hip_proctex_compile).  A text that does not compile throws with the compiler's log.
  void RecompileProcTexShaders(const std::string& a_shaderText) override { check(hydra_hip_proctex_compile(m_h, a_shaderText.c_str(), a_shaderText.size()), "RecompileProcTexShaders"); }
  void FinishAll() override { check(hydra_hip_finish(m_h), "FinishAll"); }
  void SetRaysPerPixel(int a_num) override { m_spp = a_num > 0 ? a_num : 1; }
  int  GetRaysPerPixel() const override { return m_spp; }

  void ResetPerfCounters() override { hydra_hip_reset_perf_counters(m_h); }
  HydraRaysStat GetRaysStat() override {
    HydraRaysStat st;
    memset(&st, 0, sizeof(st));
    hydra_hip_get_rays_stat(m_h, &st);
    return st;
  }
  // size mismatch: silently return, as CPUExpLayer does (hydra_drv/CPUExpLayer.cpp:133-147)
  // with MMLT running: direct (path tracer's mean) + kScale x indirect (IntegratorMMLT::GetImageHDR, CPUExp_Integrators_MMLT.cpp:616-635)
  void GetHDRImage(float4* data, int width, int height) const override {
    hydra_hip_get_hdr_image(m_h, &data->x, width, height);
    if (!m_mmltRunning || width != m_width || height != m_height) return;
    std::vector<float> ind(size_t(width) * height * 4);
    float info[8];
    if (hydra_hip_mmlt_get_image(m_h, ind.data(), width, height, info) != HYDRA_HIP_OK) return;
    for (size_t i = 0; i < size_t(width) * height; i++) { data[i].x += ind[4 * i]; data[i].y += ind[4 * i + 1]; data[i].z += ind[4 * i + 2]; }
  }
  void GetLDRImage(uint32_t* data, int width, int height) const override { hydra_hip_get_ldr_image(m_h, data, width, height); }
  float GetSPP() const override { return hydra_hip_get_spp(m_h); }

  size_t GetAvaliableMemoryAmount(bool allMem = false) override {
    size_t f = 0, t = 0;
    hydra_hip_available_memory(m_h, &f, &t);
    return allMem ? t : f;
  }
  // IHWLayer::SetExternalImageAccumulator (:199) keeps the pointer (base class); the contribution itself:
  // GPUOCLLayer::ContribToExternalImageAccumulator (GPUOCLLayerOther.cpp:365-429): internal sums += into the shared image
  // under its lock, spp and the receive counter advance, the internal accumulator restarts.
  // With MMLT running the reference's layer splats its mutations into the same screen buffer as the direct-light pass (MMLT_Pass,
  // GPUOCLLayerAdvanced.cpp:395-493), so a contribution carries both parts, and ClearAccumulatedColor (GPUOCLLayer.cpp:1288-1297) zeroes
  // that buffer without touching the chains.  Here the indirect part is an image of its own: the contribution adds spp x (kScale x indirect
  // image of the mutations since the last contribution) next to the direct sums -- shared / shared spp is then the spp-weighted mean of
  // direct + indirect estimates, what GetHDRImage returns for one interval -- and restarts that image; the chains go on.
  void ContribToExternalImageAccumulator(IHRSharedAccumImage* a_pImage) override {
    if (a_pImage == nullptr) return;
    const float spp = hydra_hip_get_spp(m_h);
    if (spp <= 0.0f) return;
    m_sums.resize(size_t(m_width) * m_height * 4);
    check(hydra_hip_get_accumulator(m_h, m_sums.data(), m_width, m_height), "ContribToExternalImageAccumulator");
    if (m_mmltRunning) {
      m_indirect.resize(m_sums.size());
      float info[8];
      check(hydra_hip_mmlt_get_image(m_h, m_indirect.data(), m_width, m_height, info), "ContribToExternalImageAccumulator(MMLT)");
    }
    if (!a_pImage->Lock(100)) return;                        // busy: the samples stay in the internal accumulator for the next call
    HRSharedBufferHeader* hdr = a_pImage->Header();
    if (hdr->width != m_width || hdr->height != m_height || hdr->channels != 4) { a_pImage->Unlock(); RunTimeError("HipHWLayer::ContribToExternalImageAccumulator: shared image does not match the frame"); }
    float* out = a_pImage->ImageData(0);
    for (size_t i = 0; i < m_sums.size(); i++) out[i] += m_sums[i];
    if (m_mmltRunning) for (size_t i = 0; i < m_sums.size(); i++) if ((i & 3) != 3) out[i] += spp * m_indirect[i];
    hdr->counterRcv++;
    hdr->spp += spp;
    a_pImage->Unlock();
    check(hydra_hip_clear_accumulated_color(m_h), "ContribToExternalImageAccumulator");   // the sums only: not the virtual ClearAccumulatedColor, which ends the MMLT run
    if (m_mmltRunning) check(hydra_hip_mmlt_reset_image(m_h), "ContribToExternalImageAccumulator(MMLT)");
    m_sppContrib += spp;
  }
  float GetSPPContrib() const override { return m_sppContrib; }
  // IHWLayer::NormalMapFromDisplacement (:197): the height map of a height_bump material -> its normal map, on the device
  std::vector<uchar4> NormalMapFromDisplacement(int w, int h, const uchar4* a_data, float bumpAmt, bool invHeight, float smoothLvl) override {
    std::vector<uchar4> res(size_t(w) * size_t(h));
    const int rc = hydra_hip_normal_map_from_displacement(m_deviceId, w, h, reinterpret_cast<const uint8_t*>(a_data), bumpAmt, invHeight ? 1 : 0, smoothLvl,
                                                          reinterpret_cast<uint8_t*>(res.data()), nullptr);
    if (rc != HYDRA_HIP_OK) RunTimeError(std::string("HipHWLayer::NormalMapFromDisplacement: ") + hydra_hip_image_last_error());
    return res;
  }
  // GPUOCLLayer::EvalGBuffer (GPUOCLLayerOther.cpp:694-870): the hand-shake over Header()->gbufferIsEmpty under the image lock, the two
  // layers chosen by the image's depth (:725-741), instance ids mapped through a_instIdByInstId (:846-853); the records themselves come
  // from hydra_hip_eval_gbuffer (IntegratorCommon::gbufferEval, CPUExp_GBuffer.cpp:15-113)
  void EvalGBuffer(IHRSharedAccumImage* a_pAccumImage, const std::vector<int32_t>& a_instIdByInstId) override {
    if (a_pAccumImage == nullptr) return;
    if (a_pAccumImage->Header()->gbufferIsEmpty != 1) return;
    bool locked = false;
    for (int i = 0; i < 20 && !locked; i++) locked = a_pAccumImage->Lock(100);
    if (!locked) return;
    HRSharedBufferHeader* hdr = a_pAccumImage->Header();
    if (hdr->gbufferIsEmpty != 1) { a_pAccumImage->Unlock(); return; }   // another process has computed it meanwhile
    float* data1 = nullptr, *data2 = nullptr;
    if (hdr->depth == 4) { data1 = a_pAccumImage->ImageData(2); data2 = a_pAccumImage->ImageData(3); }
    else if (hdr->depth == 3) { data1 = a_pAccumImage->ImageData(1); data2 = a_pAccumImage->ImageData(2); }
    if (data1 == nullptr || data2 == nullptr || hdr->width != m_width || hdr->height != m_height) {
      a_pAccumImage->Unlock();
      RunTimeError("HipHWLayer::EvalGBuffer: the shared image has no G-buffer layers (depth 3 or 4) of the frame's size");
    }
    const int rc = hydra_hip_eval_gbuffer(m_h, data1, data2, m_width, m_height, a_instIdByInstId.empty() ? nullptr : a_instIdByInstId.data(), int(a_instIdByInstId.size()), nullptr);
    if (rc != HYDRA_HIP_OK) { a_pAccumImage->Unlock(); check(rc, "EvalGBuffer"); }
    hdr->gbufferIsEmpty = 0;
    a_pAccumImage->Unlock();
  }
  // multi-GPU tile partition: a caller-owned DEVICE accumulator (float4 sums), exchanged over RCCL by the caller or by
  // hydra_hip_comm_* (include/hydra_hip.h); not part of IHWLayer
  void SetExternalDeviceAccumulator(void* a_devFloat4, size_t a_bytes) { check(hydra_hip_set_external_accumulator(m_h, a_devFloat4, a_bytes), "SetExternalDeviceAccumulator"); }
  hydra_hip_handle Handle() const { return m_h; }

private:
  hydra_hip_handle m_h;
  int m_deviceId = 0;
  char m_devName[256] = {0};
  bool m_tablesUploaded = false;
  bool m_mmltRunning = false;
  int m_mmltSeed = 777;
  int m_spp = 1;
  float m_sppContrib = 0.0f;
  std::vector<float> m_sums, m_indirect;

  void check(int rc, const char* where) const {
    if (rc != HYDRA_HIP_OK) RunTimeError(std::string("HipHWLayer::") + where + ": " + hydra_hip_last_error(m_h));
  }
  void upload_instances() {
    const size_t n = m_instMatrices.size() / 16;
    if (n == 0) return;
    std::vector<int32_t> lid = m_instLightInstId;
    lid.resize(n, -1);   // lights may be absent: every instance is "not a light"
    check(hydra_hip_upload_instances(m_h, m_instMatrices.data(), lid.data(), int(n)), "upload_instances");
  }
  void upload_remap() {
    check(hydra_hip_upload_remap_lists(m_h, m_remapLists.data(), int(m_remapLists.size()), m_remapTable.data(), int(m_remapTable.size() / 2),
                                       m_remapInst.data(), int(m_remapInst.size())), "upload_remap");
  }
};

IHWLayer* CreateHipImpl(int w, int h, int a_flags, int a_deviceId) { return new HipHWLayer(w, h, a_flags, a_deviceId); }

void* HipLayerHandle(IHWLayer* layer) {
  HipHWLayer* p = dynamic_cast<HipHWLayer*>(layer);
  return p ? static_cast<void*>(p->Handle()) : nullptr;
}

}  // namespace hydra_host
