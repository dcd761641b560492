// render_driver_lite.h -- minimal caller of the IHWLayer boundary: reads a HydraAPI scene library
// (statex_*.xml + data/chunk_*.vsgf|image4ub) and issues the same IHWLayer calls, in the same order and with the
// same buffer contents, as RenderDriverRTE does when HydraAPI commits a scene (call stack: SURVEY.md 3.1;
// hydra_drv/RenderDriverRTE.cpp:604-745 AllocAll, :753-836 UpdateImage, :1059-1159 UpdateMesh, :1171-1289 UpdateCamera,
// :160-396 UpdateSettings, :1953-2080 InstanceMeshes/InstanceLights, :1396-1550 EndScene, :1687-1936 Draw).
//
// HydraAPI itself is not in this image, so this class also plays HydraAPI's part of walking the XML (row f1,
// "scene front end").  Supported subset: hydra_material {lambert diffuse, phong/mirror reflectivity, emission},
// area lights (rect/disk), uvn camera with optional DOF.  Anything else is counted in UnsupportedFeatures().
#pragma once
#include <string>
#include <vector>
#include <memory>
#include <map>
#include <set>
#include "hw_layer.h"
#include "bvh4_builder.h"
#include "xml_mini.h"

namespace hydra_host {

class RenderDriverLite {
public:
  // a_layer is owned by the driver (reference: RenderDriverRTE deletes m_pHWLayer, RenderDriverRTE.cpp:406-410)
  RenderDriverLite(IHWLayer* a_layer, int w, int h);
  ~RenderDriverLite();

  // plays HydraAPI: open the library and commit scene 0 / camera 0 / render settings 0.
  // width/height <= 0 keep the values of <render_settings>.  trace_depth < 0 keeps the XML value.
  void LoadSceneLibrary(const std::string& a_libPath, int a_width, int a_height, int a_traceDepth, int a_enableDof);

  void Draw();                         // one pass: SetCamMatrices, PrepareEngineGlobals, (InitPathTracing), BeginTracingPass
  void GetFrameBufferHDR(float* rgba, int w, int h) { m_pHWLayer->GetHDRImage(reinterpret_cast<float4*>(rgba), w, h); }

  IHWLayer* Layer() { return m_pHWLayer; }
  int Width() const { return m_width; }
  int Height() const { return m_height; }
  int UnsupportedFeatures() const { return m_unsupported; }
  const std::string& Log() const { return m_log; }
  const std::string& ProcTexProgram() const { return m_procTexProgram; }   // what RecompileProcTexShaders was handed (empty: no procedural textures)
  BVH4Builder& Builder() { return m_bvh; }
  void SetSeed(int s) { m_seed = s; m_ptInitDone = false; }

  // individual driver entry points (same names as the IHRRenderDriver overrides in RenderDriverRTE.h:60-134)
  void AllocAll(int imgNum, int matNum, int lightNum, int meshNum);
  bool UpdateImage(int32_t a_texId, int32_t w, int32_t h, int32_t bpp, int32_t chan, const void* a_data);
  bool UpdateMaterial(int32_t a_matId, const XmlNode* a_materialNode);
  int32_t AuxNormalMapFromHeight(int32_t a_texId, int32_t a_matId, float bumpAmt, float smoothLvl);   // ... of a height map, through IHWLayer::NormalMapFromDisplacement (GetAuxNormalMapFromDisaplacement, :77-160)
  int32_t AuxNormalMapFor(int32_t a_texId, int32_t a_matId);   // aux-arena copy of a normal map (GetCachedAuxNormalMatId, RenderDriverRTE_AuxTextures.cpp:46-77)
  bool UpdateLight(int32_t a_lightId, const XmlNode* a_lightNode);
  bool UpdateSkyLight(int32_t a_lightId, const XmlNode* a_lightNode);
  void LuminanceImageOf(int32_t texId, std::vector<float>& lum, int& lw, int& lh);
  int32_t PutPdfTable2D(const std::vector<float>& lum, int lw, int lh);
  std::pair<int32_t, int32_t> AddIesTexTable(const std::string& loc);
  std::map<std::string, std::pair<int32_t, int32_t>> m_iesCache;
  std::string m_libPath;
  bool UpdateDeltaLight(int32_t a_lightId, const XmlNode* a_lightNode);
  bool UpdateMesh(int32_t a_meshId, int vertNum, int triNum, const float* pos4f, const float* norm4f, const float* tan4f,
                  const float* texcoord2f, const int* indices, const int* triMatIndices);
  bool UpdateCamera(const XmlNode* a_camNode);
  bool UpdateSettings(const XmlNode* a_settingsNode);
  void BeginScene();
  void InstanceMeshes(int32_t a_mesh_id, const float* a_matrices, int32_t a_instNum, const int* a_lightInstId,
                      const int* a_remapId, const int* a_realInstId);
  void InstanceLights(int32_t a_lightId, const float* a_matrix, const XmlNode** a_lightNodes, int32_t a_instNum, int32_t a_lightGroupId);
  void EndScene();

private:
  IHWLayer* m_pHWLayer;
  int m_width, m_height;
  int m_unsupported = 0;
  std::string m_log;
  int m_seed = 777;                    // the value the reference test harness forces (hydra_app/main_app_console.cpp:142-143)
  bool m_ptInitDone = false;
  int m_forceDof = -1;

  IMemoryStorage *m_pTexStorage = nullptr, *m_pTexStorageAux = nullptr, *m_pGeomStorage = nullptr,
                 *m_pMaterialStorage = nullptr, *m_pPdfStorage = nullptr;
  BVH4Builder m_bvh;
  BVH4Builder m_bvhAlpha;              // second tree for the instances of alpha-tested meshes when the scene asks for one (<split_alpha_tree>)
  bool m_splitAlphaTree = false;
  struct Opacity { int32_t texId; float sampler[12]; bool smooth, skipShadow; };
  std::map<int, Opacity> m_matOpacity; // materials with an <opacity> node (PlainMaterialConverter.cpp:1429-1445)
  std::map<int32_t, int32_t> m_auxNormalMaps;   // texture id -> aux texture id of its copy in the aux arena (m_texturesProcessedNM, RenderDriverRTE_AuxTextures.cpp:50-73)
  int32_t m_auxImageNumber = 0;
  std::vector<uint32_t> m_alphaTable[2];   // uint2 per float4 of the triangle list + the opacity samplers, per tree
  bool MeshHasOpacity(int32_t a_meshId) const;
  void CreateAlphaTestTable(ConvertionResult& cr);

  struct Camera { float fov = 45.0f, nearPlane = 0.1f, farPlane = 1000.0f; float3 pos{0, 0, 0}, lookAt{0, 0, -1}, up{0, 1, 0};
                  bool useMatrices = false; float4x4 mProj, mWorldView; } m_camera;   // useMatrices: a "two_matrices" camera (RenderDriverRTE.cpp:1178-1201)

  struct LightProto { std::vector<float> plain; bool isDisk = false, isSky = false, isDelta = false, isSphere = false, isMesh = false, isCylinder = false, hasIes = false; int kind = 0;
                      std::vector<float> meshPos; std::vector<int32_t> meshInd; };   // meshPos / meshInd: MeshLight::tempPos / tempInd   // kind: 0 point, 1 spot, 2 directional   // un-instanced PlainLight (128 floats)
  std::map<int, LightProto> m_lights;
  std::map<std::string, int32_t> m_auxHeightMaps;   // m_texturesProcessedNM: texture id + bump parameters -> aux id
  bool m_sceneHaveSkyPortals = false;
  // the back-plate (RenderDriverRTE.h:128-131): set by a shadow_catcher's <back> (PlainMaterialConverter.cpp:1642-1674), then overwritten by every instanced sky light with
  // ITS <back> -- or with "none" when it has no such node (RenderDriverRTE.cpp:2072-2078); EndScene hands it to the layer's variables (:1487-1492)
  int32_t m_shadowMatteBackTexId = int32_t(HYDRA_INVALID_TEXTURE); int m_shadowMatteBackMode = 0; float m_shadowMatteBackGamma = 2.2f; float3 m_shadowMatteBackColor{1, 1, 1};
  struct SkyBack { int32_t texId = int32_t(HYDRA_INVALID_TEXTURE); int mode = 0; float gamma = 2.2f; float3 color{1, 1, 1}; };
  std::map<int, SkyBack> m_skyBack;   // ILight::tmpSkyLightBack*, AbstractMaterial.h:152-155
  // procedural textures (RenderDriverRTE_ProcTex.cpp): what <texture type="proc"> declares -- the generated call, the functions' text, the return width
  struct ProcTex { std::string call, code; int retT = 4; };
  std::map<int32_t, std::shared_ptr<void>> m_materialTrees;   // RenderDriverRTE::m_materialUpdated: the converted tree (MatTree, render_driver_lite.cpp) of every material, for hydra_blend
  std::map<int32_t, const XmlNode*> m_materialNodes;          // RenderDriverRTE::m_materialNodes: valid while LoadSceneLibrary's document lives (the blends are converted inside it)
  std::map<int32_t, ProcTex> m_procTextures;
  std::string m_procTexProgram;
  std::set<int32_t> m_procTexMissing;                                     // declared, but the code file is not in the library
  bool UpdateImageProc(int32_t a_texId, const XmlNode* a_texNode);        // RenderDriverRTE::UpdateImageProc :576-629
  std::string ProcTexProgramText() const;                                  // the text EndTexturesUpdate hands to IHWLayer::RecompileProcTexShaders :485-574
  void AppendProcTexTail(const XmlNode* a_materialNode, int32_t a_matId, std::vector<float>& mdata);   // UpdateMaterial :875-899 + PutAbstractMaterialToStorage :1865-1874
  std::vector<int32_t> m_lightIdByInst;                                   // light id of every instanced record
  std::vector<float> m_lightsInstanced;                                   // 128 floats per light instance

  std::vector<float4x4> m_instMatricesInv;
  std::vector<int32_t> m_instLightInstId, m_meshIdByInstId, m_meshRemapListId;

  void Unsupported(const std::string& what);
  std::vector<float> CalcLightPickProbTable(bool a_fwd);
};

}  // namespace hydra_host
