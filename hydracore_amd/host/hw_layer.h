// hw_layer.h -- host-side mirror of the reference's hardware-layer interface for the PT hot path.
//
// Same method names, argument meaning and error behaviour as hydra_drv/IHWLayer.h:97-246 so that a
// maintainer can paste HipHWLayer into the reference tree, swap the include of this header for
// IHWLayer.h, and register CreateHipImpl next to CreateOclImpl/CreateCPUExpImpl
// (IHWLayer.h:256-257, RenderDriverRTE.cpp:85-88); see INTEGRATION.md.  HydraAPI/pugixml types that the
// original signatures mention (pugi::xml_node, IHRSharedAccumImage) are not available in this image
// and are replaced by opaque pointers.
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
  virtual void SetCamNode(const void* a_camNode) { m_camNode = a_camNode; }
  virtual void SetSettingsNode(const void* a_node) { m_settingsNode = a_node; }

  virtual void SetAllBVH4(const ConvertionResult& a_convertedBVH, void* a_inBuilderAPI, int a_flags) = 0;
  virtual void SetAllInstMatrices(const float4x4* a_matrices, int32_t a_matrixNum) = 0;
  virtual void SetAllInstLightInstId(const int32_t* a_lightInstIds, int32_t a_instNum) = 0;
  virtual void SetAllPODLights(const float* a_lights128, size_t a_number);

  virtual void SetAllLightsSelectTable(const float* a_table, int32_t a_tableSize, bool a_fwd = false);
  virtual void SetAllRemapLists(const int* a_allLists, const int* a_tableInt2, int a_allSize, int a_tableSize) {}
  virtual void SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) {}

  virtual void SetAllFlagsAndVars(const AllRenderVarialbes& a_vars);
  virtual AllRenderVarialbes GetAllFlagsAndVars() const;

  virtual void BeginTracingPass() = 0;
  virtual void EndTracingPass() = 0;
  virtual void FinishAll() {}

  virtual void InitPathTracing(int seed, std::vector<int32_t>* pInstRemapTable = nullptr) = 0;
  virtual void ClearAccumulatedColor() = 0;

  virtual void ResetPerfCounters() = 0;
  virtual void ResizeScreen(int w, int h, int a_flags) { m_width = w; m_height = h; }

  virtual void GetLDRImage(uint32_t* data, int width, int height) const = 0;
  virtual void GetHDRImage(float* data4, int width, int height) const = 0;

  virtual size_t GetAvaliableMemoryAmount(bool allMem = false) = 0;
  virtual size_t GetMaxBufferSizeInBytes() { return GetAvaliableMemoryAmount(); }

  virtual HydraRaysStat GetRaysStat() = 0;
  virtual int32_t GetRayBuffSize() const { return 0; }
  virtual const char* GetDeviceName(int* pOCLVer = nullptr) const { return "host"; }

  virtual void SetRaysPerPixel(int a_num) {}
  virtual int  GetRaysPerPixel() const { return 1; }

  virtual void SetNamedBuffer(const char* a_name, void* a_data, size_t a_size) {}
  virtual void CallNamedFunc(const char* a_name, const char* a_args) {}

  virtual bool StoreCPUData() const { return false; }
  virtual bool   MLT_IsAllocated() const { return true; }
  virtual size_t MLT_Alloc(int a_width, int a_height, int a_maxBounce) { return 0; }
  virtual void   MLT_Free() {}

  // external accumulator: device pointer to float4 sums in this build (reference: IHRSharedAccumImage*)
  virtual void SetExternalImageAccumulator(void* a_pImage, size_t a_bytes) { m_pExternalImage = a_pImage; }
  virtual void ContribToExternalImageAccumulator(void* a_pImage) {}

  virtual const int32_t* GetEngineGlobals() const { return m_cdataPrepared.data(); }
  virtual size_t GetEngineGlobalsSizeInWords() const { return m_cdataPrepared.size(); }

  virtual float GetSPP() const { return 0.0f; }
  virtual float GetSPPDone() const { return GetSPP(); }
  virtual float GetSPPContrib() const { return GetSPP(); }

  IMemoryStorage* FindStorage(const char* name) const {
    auto p = m_allMemStorages.find(name);
    return p == m_allMemStorages.end() ? nullptr : p->second;
  }

protected:
  int m_width, m_height;
  const void* m_camNode;
  const void* m_settingsNode;
  AllRenderVarialbes m_vars;
  std::vector<int32_t> m_globsBuffHeader;   // EngineGlobals header, HG_HEADER_WORDS words
  void* m_pExternalImage;
  std::vector<int32_t> m_cdataPrepared;     // [header | tables | lights]
  std::unordered_map<std::string, IMemoryStorage*> m_allMemStorages;
  std::vector<float> m_lightSelectTableRev, m_lightSelectTableFwd;
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

  void SetAllBVH4(const ConvertionResult& a_convertedBVH, void* a_inBuilderAPI, int a_flags) override;
  void SetAllInstMatrices(const float4x4* a_matrices, int32_t a_matrixNum) override;
  void SetAllInstLightInstId(const int32_t* a_lightInstIds, int32_t a_instNum) override;
  void SetAllRemapLists(const int* a_allLists, const int* a_tableInt2, int a_allSize, int a_tableSize) override;
  void SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) override;

  void BeginTracingPass() override { RunTimeError("SharedDataLayer: no compute device behind this layer"); }
  void EndTracingPass() override {}
  void InitPathTracing(int seed, std::vector<int32_t>* = nullptr) override {}
  void ClearAccumulatedColor() override {}
  void ResetPerfCounters() override {}
  void GetLDRImage(uint32_t*, int, int) const override {}
  void GetHDRImage(float*, int, int) const override {}
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
