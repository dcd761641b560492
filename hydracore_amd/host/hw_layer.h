// hw_layer.h -- host-side mirror of the reference's hardware-layer interface for the PT hot path.
//
// Same method names, arity, argument meaning and error behaviour as hydra_drv/IHWLayer.h:97-246 so that a maintainer can
// paste HipHWLayer into the reference tree, swap the include of this header for IHWLayer.h, and register CreateHipImpl
// next to CreateOclImpl/CreateCPUExpImpl (IHWLayer.h:256-257, RenderDriverRTE.cpp:85-88); see INTEGRATION.md.
// tests/test_boundary_drift.py parses the reference header and diffs the two method sets.  HydraAPI / pugixml types the
// original signatures mention are not available in this image; they appear here under their own names as opaque or
// minimal declarations (XmlNodeHandle for pugi::xml_node, IBVHBuilder2, HRRenderDeviceInfoListElem forward-declared,
// IHRSharedAccumImage with the members the reference's layers call) -- the only differences the drift test whitelists.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <unordered_map>
#include <stdexcept>

#include "../../include/hydra_layouts.h"
#include "math_util.h"

namespace hydra_host {

// reference: RUN_TIME_ERROR (hydra_drv/globals_sys.h:56-66) throws std::runtime_error
[[noreturn]] inline void RunTimeError(const std::string& msg) { throw std::runtime_error(msg); }

// hydra_drv/IHWLayer.h:23-45 (spelling kept)
typedef int32_t EngineGlobals;   // the reference's struct (cfetch.h:21-81) is addressed as int words here: HG_* offsets in include/hydra_layouts.h

struct AllRenderVarialbes {
  AllRenderVarialbes() : m_flags(0) {
    for (int i = 0; i < 64; i++) { m_varsI[i] = 0; m_varsF[i] = 0.0f; }
  }
  int          m_varsI[64];
  float        m_varsF[64];
  unsigned int m_flags;
};

enum CLEAR_FLAGS { CLEAR_MATERIALS = 1, CLEAR_GEOMETRY = 2, CLEAR_LIGHTS = 4, CLEAR_TEXTURES = 8, CLEAR_CUSTOM_DATA = 16, CLEAR_ALL = 31 };

// creation flags, hydra_drv/IHWLayer.h:322-339 (subset)
enum { GPU_RT_NOWINDOW = 1, GPU_RT_CPU_FRAMEBUFFER = 8192 };

// hydra_drv/IBVHBuilderAPI.h:5-33
#define MAXBVHTREES 4
struct ConvertionResult {
  ConvertionResult() : treesNum(0) {
    for (int i = 0; i < MAXBVHTREES; i++) {
      bvhType[i] = nullptr; pBVH[i] = nullptr; pTriangleData[i] = nullptr; pTriangleAlpha[i] = nullptr;
      nodesNum[i] = 0; trif4Num[i] = 0; triAfNum[i] = 0;
    }
  }
  const char*         bvhType[MAXBVHTREES];        // "object" => instanced two-level tree
  const HydraBVHNode* pBVH[MAXBVHTREES];
  const float*        pTriangleData[MAXBVHTREES];  // float4 units
  const uint32_t*     pTriangleAlpha[MAXBVHTREES]; // uint2 units
  int nodesNum[MAXBVHTREES];
  int trif4Num[MAXBVHTREES];
  int triAfNum[MAXBVHTREES];
  int treesNum;
};

// hydra_drv/IMemoryStorage.h:16-49 + MemoryStorageCPU.cpp:9-127: linear arena of 16-byte blocks with an
// id -> block-offset table; offsets are reported in int4 units, -1 = absent.
class IMemoryStorage {
public:
  explicit IMemoryStorage(const char* name) : m_name(name), maxId(0) {}
  virtual ~IMemoryStorage() {}

  virtual void   Clear() { m_data.clear(); objects.clear(); maxId = 0; }
  virtual size_t Reserve(uint64_t bytes) { m_data.reserve(bytes); return m_data.capacity(); }
  virtual const void*  GetBegin() const { return m_data.data(); }
  virtual size_t GetSize() const { return m_data.size(); }
  virtual size_t GetCapacity() const { return m_data.capacity(); }
  virtual int    GetAlignSizeInBytes() const { return 16; }
  virtual int    GetMaxObjectId() const { return maxId; }

  virtual int32_t Update(int32_t id, const void* a_data, uint64_t a_sizeInBytes);
  virtual void    UpdatePartial(int32_t id, const void* a_data, uint64_t a_offsetInBytes, uint64_t a_sizeInBytes);
  virtual std::vector<int32_t> GetTable() const;
  const std::string& Name() const { return m_name; }

protected:
  struct LChunk { int begin, endMax, endCur; };
  std::string m_name;
  std::vector<char> m_data;
  std::unordered_map<int, LChunk> objects;
  int maxId;
  LChunk AppendToTheEnd(const void* a_data, uint64_t a_sizeInBytes);
};

// ---- HydraAPI-side types the interface mentions (absent from this image; in the reference tree their headers take over)
struct XmlNodeHandle { const void* node = nullptr; }; // stands for pugi::xml_node: passed by value, opaque to the layer
class IBVHBuilder2;                                   // hydra_drv/IBVHBuilderAPI.h:35-68; the HIP layer consumes the converted layout only
struct HRRenderDeviceInfoListElem;                    // HydraAPI device list element (ListDevices)
typedef void (*RTE_PROGRESSBAR_CALLBACK)(const wchar_t* message, float a_progress);   // hydra_drv/IHWLayer.h:20
typedef float PlainLight;                             // cfetch.h:6-13: a light is 128 floats; SetAllPODLights takes number of LIGHTS
// Shared accumulation image: the members the reference's layers use (GPUOCLLayerOther.cpp:259-283 implicit contribution,
// :365-429 ContribToExternalImageAccumulator): Lock(ms) / Unlock around `+=` into ImageData(0), Header()->spp and counterRcv.
struct HRSharedBufferHeader { int width, height, depth, channels; float spp; int counterRcv, counterSnd; int gbufferIsEmpty; };   // gbufferIsEmpty: EvalGBuffer's hand-shake (GPUOCLLayerOther.cpp:699-716, 846; RenderDriverRTE.h:332)
struct IHRSharedAccumImage {
  virtual ~IHRSharedAccumImage() {}
  virtual bool   Lock(int a_miliseconds) = 0;
  virtual void   Unlock() = 0;
  virtual float* ImageData(int layerNum) = 0;
  virtual HRSharedBufferHeader* Header() = 0;
};

// ------------------------------------------------------------------------------------------------
// IHWLayer: the boundary class (hydra_drv/IHWLayer.h:97-246).  Base-class bodies assemble the globals
// blob exactly as IHWLayerDataAssembler.cpp:66-452 does.
class IHWLayer {
public:
  IHWLayer();
  virtual ~IHWLayer();

  virtual void Clear(CLEAR_FLAGS a_flags) = 0;

  virtual IMemoryStorage* CreateMemStorage(uint64_t a_maxSizeInBytes, const char* a_name) = 0;
  virtual void ResizeTablesForEngineGlobals(int32_t a_geomNum, int32_t a_imgNum, int32_t a_matNum, int32_t a_lightNum);

  virtual void PrepareEngineGlobals();
  virtual void PrepareEngineTables();

  virtual void SetCamMatrices(float mProjInverse[16], float mWorldViewInverse[16], float mProj[16], float mWorldView[16],
                              float a_aspect, float a_fovX, float3 a_lookAt);
  virtual void SetCamNode(XmlNodeHandle a_camNode) { m_camNode = a_camNode; }
  virtual void SetSettingsNode(XmlNodeHandle a_node) { m_settingsNode = a_node; }

  virtual void SetAllBVH4(const ConvertionResult& a_convertedBVH, IBVHBuilder2* a_inBuilderAPI, int a_flags) = 0;
  virtual void SetAllInstMatrices(const float4x4* a_matrices, int32_t a_matrixNum) = 0;
  virtual void SetAllInstLightInstId(const int32_t* a_lightInstIds, int32_t a_instNum) = 0;
  virtual void SetAllPODLights(PlainLight* a_lights2, size_t a_number);

  virtual void SetAllLightsSelectTable(const float* a_table, int32_t a_tableSize, bool a_fwd = false);
  virtual void SetAllRemapLists(const int* a_allLists, const int2* a_table, int a_allSize, int a_tableSize) {}
  virtual void SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) {}

  virtual void SetAllFlagsAndVars(const AllRenderVarialbes& a_vars);
  virtual AllRenderVarialbes GetAllFlagsAndVars() const;

  virtual void BeginTracingPass() = 0;
  virtual void EndTracingPass() = 0;
  virtual void EvalGBuffer(IHRSharedAccumImage* a_pAccumImage, const std::vector<int32_t>& a_instIdByInstId) {}
  virtual void FinishAll() {}

  virtual void InitPathTracing(int seed, std::vector<int32_t>* pInstRemapTable = nullptr) = 0;
  virtual void ClearAccumulatedColor() = 0;
  virtual void CPUPluginFinish() {}

  virtual void ResetPerfCounters() = 0;
  virtual void ResizeScreen(int w, int h, int a_flags) { m_width = w; m_height = h; }

  virtual void GetLDRImage(uint32_t* data, int width, int height) const = 0;
  virtual void GetHDRImage(float4* data, int width, int height) const = 0;

  virtual size_t GetAvaliableMemoryAmount(bool allMem = false) = 0;
  virtual size_t GetMaxBufferSizeInBytes() { return GetAvaliableMemoryAmount(); }

  virtual HydraRaysStat GetRaysStat() = 0;
  virtual int32_t GetRayBuffSize() const { return 0; }
  virtual const char* GetDeviceName(int* pOCLVer = nullptr) const { return "host"; }
  virtual const HRRenderDeviceInfoListElem* ListDevices() const { return nullptr; }

  virtual void SetRaysPerPixel(int a_num) {}
  virtual int  GetRaysPerPixel() const { return 1; }

  virtual void SetNamedBuffer(const char* a_name, void* a_data, size_t a_size) {}
  virtual void CallNamedFunc(const char* a_name, const char* a_args) {}
  virtual void RenderFullScreenBuffer(const char* a_dataName, float4* a_data, int width, int height, int a_spp) {
    std::vector<ushort2> pixels(size_t(m_width) * m_height);
    for (int y = 0; y < height; y++)
      for (int x = 0; x < m_width; x++) { pixels[size_t(y) * m_width + x].x = (unsigned short)x; pixels[size_t(y) * m_width + x].y = (unsigned short)y; }
    renderSubPixelData(a_dataName, pixels, a_spp, a_data, nullptr);
  }

  virtual bool ImplementPhotonMapping() const { return false; }
  virtual bool StoreCPUData() const { return false; }
  virtual bool   MLT_IsAllocated() const { return true; }
  virtual size_t MLT_Alloc(int a_width, int a_height, int a_maxBounce) { return 0; }
  virtual void   MLT_Free() {}
  virtual void   SetProgressBarCallback(RTE_PROGRESSBAR_CALLBACK a_pFunc) { m_progressBar = a_pFunc; }

  virtual std::vector<uchar4> NormalMapFromDisplacement(int w, int h, const uchar4* a_data, float bumpAmt, bool invHeight, float smoothLvl) { return std::vector<uchar4>(); }

  virtual void SetExternalImageAccumulator(IHRSharedAccumImage* a_pImage) { m_pExternalImage = a_pImage; }   ///< implicit contribution after every pass
  virtual void ContribToExternalImageAccumulator(IHRSharedAccumImage* a_pImage) {}                            ///< explicit contribution

  virtual EngineGlobals* GetEngineGlobals() { return m_cdataPrepared.data(); }   // EngineGlobals = the int blob [header | tables | lights], include/hydra_layouts.h
  size_t GetEngineGlobalsSizeInWords() const { return m_cdataPrepared.size(); }

  virtual void RecompileProcTexShaders(const std::string& a_shaderPath) {}

  virtual float GetSPP() const { return 0.0f; }
  virtual float GetSPPDone() const { return GetSPP(); }
  virtual float GetSPPContrib() const { return GetSPP(); }

  IMemoryStorage* FindStorage(const char* name) const {
    auto p = m_allMemStorages.find(name);
    return p == m_allMemStorages.end() ? nullptr : p->second;
  }

  // The two multi-scattering energy tables of the header (EngineGlobals::m_essGgx2017Table u16[64 x 64], m_essTranspTable u16[64^3],
  // cfetch.h:77-79).  The reference's layers copy baked data in when they are constructed (IHWLayer.h:101: InitEngineGlobals with
  // getGgxTable() / getTranspTable() of bakeBrdfEnergy/); this build's device layer bakes its own on the GPU (hydra_hip_bake_energy_tables)
  // and the host-blob layer reads a file of such a bake when HYDRA_AMD_ENERGY_TABLES names one.  Not virtual: not part of the boundary.
  void SetEnergyTables(const uint16_t* a_ggx4096, const uint16_t* a_transp262144);
  bool HaveEnergyTables() const { return m_haveEnergyTables; }

protected:
  virtual void renderSubPixelData(const char* a_dataName, const std::vector<ushort2>& a_pixels, int spp, float4* a_pixValues, float4* a_subPixValues) {}

  int m_width, m_height;
  XmlNodeHandle m_camNode;
  XmlNodeHandle m_settingsNode;
  AllRenderVarialbes m_vars;
  std::vector<int32_t> m_globsBuffHeader;   // EngineGlobals header, HG_HEADER_WORDS words
  RTE_PROGRESSBAR_CALLBACK m_progressBar;
  IHRSharedAccumImage* m_pExternalImage;
  std::vector<int32_t> m_cdataPrepared;     // [header | tables | lights]
  std::unordered_map<std::string, IMemoryStorage*> m_allMemStorages;
  std::vector<float> m_lightSelectTableRev, m_lightSelectTableFwd;
  bool m_haveEnergyTables = false;
};

size_t CalcConstGlobDataOffsets(int32_t* pGlobalsHeader);

// ------------------------------------------------------------------------------------------------
// SharedDataLayer: host-resident copy of everything the kernels read; counterpart of CPUSharedData
// (hydra_drv/IHWLayer.h:266-319).  Usable on its own (no device) to hand buffers to the CPU oracle.
class SharedDataLayer : public IHWLayer {
public:
  SharedDataLayer(int w, int h, int a_flags);
  ~SharedDataLayer() override;

  void Clear(CLEAR_FLAGS a_flags) override;
  IMemoryStorage* CreateMemStorage(uint64_t a_maxSizeInBytes, const char* a_name) override;

  void SetAllBVH4(const ConvertionResult& a_convertedBVH, IBVHBuilder2* a_inBuilderAPI, int a_flags) override;
  void SetAllInstMatrices(const float4x4* a_matrices, int32_t a_matrixNum) override;
  void SetAllInstLightInstId(const int32_t* a_lightInstIds, int32_t a_instNum) override;
  void SetAllRemapLists(const int* a_allLists, const int2* a_table, int a_allSize, int a_tableSize) override;
  void SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) override;

  void BeginTracingPass() override { RunTimeError("SharedDataLayer: no compute device behind this layer"); }
  void EndTracingPass() override {}
  void InitPathTracing(int seed, std::vector<int32_t>* = nullptr) override {}
  void ClearAccumulatedColor() override {}
  void ResetPerfCounters() override {}
  void GetLDRImage(uint32_t*, int, int) const override {}
  void GetHDRImage(float4*, int, int) const override {}
  size_t GetAvaliableMemoryAmount(bool = false) override { return size_t(8) << 30; }
  HydraRaysStat GetRaysStat() override { return HydraRaysStat(); }
  bool StoreCPUData() const override { return true; }
  virtual bool HasDevice() const { return false; }   // true when a compute device executes BeginTracingPass

  struct TreeCopy {
    std::vector<HydraBVHNode> m_bvh;
    std::vector<float> m_tris;        // float4 units * 4
    std::vector<uint32_t> m_atbl;
    bool haveInst = false;
  };
  int m_bvhTreesNum = 0;
  TreeCopy m_bvhTrees[MAXBVHTREES];
  std::vector<float> m_instMatrices;       // 16 floats per instance, columns (world -> object)
  std::vector<int32_t> m_instLightInstId;
  std::vector<int32_t> m_remapLists, m_remapTable, m_remapInst;

protected:
  int m_initFlags;   // storages are owned by the driver, as in the reference (IHWLayerDataAssembler.cpp:66-78)
};

IHWLayer* CreateHostBlobImpl(int w, int h, int a_flags);
// the new factory next to CreateOclImpl / CreateCPUExpImpl (IHWLayer.h:256-257)
IHWLayer* CreateHipImpl(int w, int h, int a_flags, int a_deviceId);

}  // namespace hydra_host
