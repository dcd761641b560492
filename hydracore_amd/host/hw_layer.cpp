// hw_layer.cpp -- arena, globals-blob assembly and the host-resident SharedDataLayer.
// Follows hydra_drv/MemoryStorageCPU.cpp:9-127 and hydra_drv/IHWLayerDataAssembler.cpp:66-452.
#include "hw_layer.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cassert>
#include <cmath>

namespace hydra_host {

// ------------------------------------------------------------------------------------------ arena
static inline size_t SizeInBlocks(uint64_t bytes, int block) { return (bytes % block == 0) ? bytes / block : bytes / block + 1; }

IMemoryStorage::LChunk IMemoryStorage::AppendToTheEnd(const void* a_data, uint64_t a_sizeInBytes) {
  const int bpb = GetAlignSizeInBytes();
  const size_t blocks = SizeInBlocks(a_sizeInBytes, bpb);
  const size_t begin = m_data.size();
  m_data.resize(begin + blocks * bpb, 0);
  if (a_data != nullptr) memcpy(m_data.data() + begin, a_data, a_sizeInBytes);
  LChunk c;
  c.begin = int(begin / bpb);
  c.endCur = c.endMax = int(m_data.size() / bpb);
  return c;
}

int32_t IMemoryStorage::Update(int32_t id, const void* a_data, uint64_t a_sizeInBytes) {
  if (id > maxId) maxId = id;
  const int bpb = GetAlignSizeInBytes();
  const size_t blocks = SizeInBlocks(a_sizeInBytes, bpb);
  auto p = objects.find(id);
  if (p != objects.end() && size_t(p->second.begin) + blocks <= size_t(p->second.endMax)) {  // update in place
    if (a_data != nullptr) memcpy(m_data.data() + size_t(p->second.begin) * bpb, a_data, a_sizeInBytes);
    p->second.endCur = p->second.begin + int(blocks);
    return p->second.begin;
  }
  LChunk c = AppendToTheEnd(a_data, a_sizeInBytes);
  objects[id] = c;
  return c.begin;
}

void IMemoryStorage::UpdatePartial(int32_t id, const void* a_data, uint64_t a_offsetInBytes, uint64_t a_sizeInBytes) {
  auto p = objects.find(id);
  if (p == objects.end()) return;
  const int bpb = GetAlignSizeInBytes();
  const LChunk c = p->second;
  if (a_offsetInBytes + a_sizeInBytes > size_t(c.endMax) * bpb) return;
  if (c.begin == -1) return;
  memcpy(m_data.data() + size_t(c.begin) * bpb + a_offsetInBytes, a_data, a_sizeInBytes);
}

std::vector<int32_t> IMemoryStorage::GetTable() const {
  const int mult = GetAlignSizeInBytes() / 16;
  std::vector<int32_t> res(maxId + 1, -1);
  for (auto& o : objects) res[o.first] = o.second.begin * mult;
  return res;
}

// ------------------------------------------------------------------------------------------ IHWLayer base
static inline size_t roundBlocks(size_t elems, int per) {
  if (elems < size_t(per)) return size_t(per);  // reference quirk (cglobals.h:625-631): even 0 takes one block
  return ((elems % per == 0) ? elems / per : elems / per + 1) * per;
}

size_t CalcConstGlobDataOffsets(int32_t* g) {
  const int A = 16;
  size_t cur = roundBlocks(HG_HEADER_WORDS, A);
  g[HG_MAT_TABLE_OFFS] = int(cur);    cur += roundBlocks(g[HG_MAT_TABLE_SIZE], A);
  g[HG_GEOM_TABLE_OFFS] = int(cur);   cur += roundBlocks(g[HG_GEOM_TABLE_SIZE], A);
  g[HG_TEX_TABLE_OFFS] = int(cur);    cur += roundBlocks(g[HG_TEX_TABLE_SIZE], A);
  g[HG_TEXAUX_TABLE_OFFS] = int(cur); cur += roundBlocks(g[HG_TEXAUX_TABLE_SIZE], A);
  g[HG_PDF_TABLE_OFFS] = int(cur);    cur += roundBlocks(g[HG_PDF_TABLE_SIZE], A);
  g[HG_LSEL_REV_OFFS] = int(cur);     cur += roundBlocks(g[HG_LSEL_REV_SIZE], A);
  g[HG_LSEL_FWD_OFFS] = int(cur);     cur += roundBlocks(g[HG_LSEL_FWD_SIZE], A);
  g[HG_FLOAT_ARRAYS_OFFS] = int(cur); cur += roundBlocks(g[HG_FLOAT_ARRAYS_SIZE], A);
  g[HG_LIGHTS_OFFS] = int(cur);       cur += roundBlocks(g[HG_LIGHTS_SIZE], A);
  return cur;
}

IHWLayer::IHWLayer() : m_width(0), m_height(0), m_progressBar(nullptr), m_pExternalImage(nullptr) {
  // InitEngineGlobals (cfetch.h:83-93): zero, rmQMC = -1, tables-ready flag.  The GGX / transparency energy tables stay zero until a
  // subclass supplies them (SetEnergyTables): the device layer bakes them, the host-blob layer may read a file.
  m_globsBuffHeader.assign(HG_HEADER_WORDS, 0);
  for (int i = 0; i < 16; i++) m_globsBuffHeader[HG_RM_QMC + i] = -1;
  m_globsBuffHeader[HG_TABLES_READY] = 1;
}

IHWLayer::~IHWLayer() { m_allMemStorages.clear(); }  // storages are deleted by the driver

void IHWLayer::SetEnergyTables(const uint16_t* a_ggx4096, const uint16_t* a_transp262144) {
  memcpy(&m_globsBuffHeader[HG_ESS_GGX_TABLE], a_ggx4096, 64 * 64 * sizeof(uint16_t));
  memcpy(&m_globsBuffHeader[HG_ESS_TRANSP_TABLE], a_transp262144, 64 * 64 * 64 * sizeof(uint16_t));
  m_haveEnergyTables = true;
}

void IHWLayer::SetAllFlagsAndVars(const AllRenderVarialbes& a_vars) {
  m_globsBuffHeader[HG_FLAGS] = int32_t(a_vars.m_flags);
  m_vars = a_vars;
  memcpy(&m_globsBuffHeader[HG_VARS_I], m_vars.m_varsI, sizeof(int) * 64);
  memcpy(&m_globsBuffHeader[HG_VARS_F], m_vars.m_varsF, sizeof(float) * 64);
}

AllRenderVarialbes IHWLayer::GetAllFlagsAndVars() const { return m_vars; }

void IHWLayer::SetCamMatrices(float mProjInverse[16], float mWorldViewInverse[16], float mProj[16], float mWorldView[16],
                              float a_aspectX, float a_fovX, float3 a_lookAt) {
  memcpy(&m_globsBuffHeader[HG_MPROJ_INV], mProjInverse, 64);
  memcpy(&m_globsBuffHeader[HG_MWORLDVIEW_INV], mWorldViewInverse, 64);
  memcpy(&m_globsBuffHeader[HG_MPROJ], mProj, 64);
  memcpy(&m_globsBuffHeader[HG_MWORLDVIEW], mWorldView, 64);

  const float w = float(m_width), h = float(m_height);
  AllRenderVarialbes vars = this->GetAllFlagsAndVars();
  vars.m_varsF[HV_F_FOV_X] = a_fovX;
  vars.m_varsF[HV_F_FOV_Y] = a_fovX / a_aspectX;
  vars.m_varsF[HV_F_WIDTH_F] = w;
  vars.m_varsF[HV_F_HEIGHT_F] = h;

  float4x4 mT;
  memcpy(mT.c, mWorldViewInverse, 64);
  const float3 p1 = mul_point(mT, float3(0, 0, 0)), p2 = mul_point(mT, float3(0, 0, -1)), p3 = mul_point(mT, float3(0, 1, 0));
  const float3 fwd = normalize(p2 - p1), up = normalize(p3 - p1);
  float* gf = reinterpret_cast<float*>(m_globsBuffHeader.data());
  gf[HG_CAM_FORWARD + 0] = fwd.x; gf[HG_CAM_FORWARD + 1] = fwd.y; gf[HG_CAM_FORWARD + 2] = fwd.z;
  gf[HG_CAM_UP + 0] = up.x; gf[HG_CAM_UP + 1] = up.y; gf[HG_CAM_UP + 2] = up.z;
  gf[HG_CAM_LOOKAT + 0] = a_lookAt.x; gf[HG_CAM_LOOKAT + 1] = a_lookAt.y; gf[HG_CAM_LOOKAT + 2] = a_lookAt.z;
  gf[HG_IMAGE_PLANE_DIST] = w / (2.f * tanf(0.5f * a_fovX));
  this->SetAllFlagsAndVars(vars);
}

void IHWLayer::ResizeTablesForEngineGlobals(int32_t a_geomNum, int32_t a_imgNum, int32_t a_matNum, int32_t a_lightNum) {
  m_globsBuffHeader[HG_TEX_TABLE_SIZE] = a_imgNum;
  m_globsBuffHeader[HG_TEXAUX_TABLE_SIZE] = a_imgNum;
  m_globsBuffHeader[HG_MAT_TABLE_SIZE] = a_matNum;
  m_globsBuffHeader[HG_PDF_TABLE_SIZE] = a_lightNum;
  m_globsBuffHeader[HG_GEOM_TABLE_SIZE] = a_geomNum;
  m_globsBuffHeader[HG_LIGHTS_SIZE] = HL_FLOATS * a_lightNum;
}

void IHWLayer::SetAllLightsSelectTable(const float* a_table, int32_t a_tableSize, bool a_fwd) {
  auto& dst = a_fwd ? m_lightSelectTableFwd : m_lightSelectTableRev;
  m_globsBuffHeader[a_fwd ? HG_LSEL_FWD_SIZE : HG_LSEL_REV_SIZE] = a_tableSize;
  dst.assign(a_table, a_table + (a_tableSize > 0 ? a_tableSize : 0));
}

void IHWLayer::PrepareEngineGlobals() {
  const size_t total = CalcConstGlobDataOffsets(m_globsBuffHeader.data());
  if (m_cdataPrepared.size() < total) m_cdataPrepared.resize(total, 0);
  if (m_cdataPrepared.empty()) return;
  memcpy(m_cdataPrepared.data(), m_globsBuffHeader.data(), sizeof(int32_t) * HG_HEADER_WORDS);
}

void IHWLayer::PrepareEngineTables() {
  int32_t* pbuff = m_cdataPrepared.data();
  struct { const char* name; int offs, size; } tabs[5] = {
      {"textures", HG_TEX_TABLE_OFFS, HG_TEX_TABLE_SIZE},   {"textures_aux", HG_TEXAUX_TABLE_OFFS, HG_TEXAUX_TABLE_SIZE},
      {"geom", HG_GEOM_TABLE_OFFS, HG_GEOM_TABLE_SIZE},     {"materials", HG_MAT_TABLE_OFFS, HG_MAT_TABLE_SIZE},
      {"pdfs", HG_PDF_TABLE_OFFS, HG_PDF_TABLE_SIZE}};
  for (auto& t : tabs) {
    IMemoryStorage* st = FindStorage(t.name);
    if (st == nullptr) RunTimeError(std::string("PrepareEngineTables: memory storage not found: ") + t.name);
    auto table = st->GetTable();
    if (int(table.size()) > m_globsBuffHeader[t.size])
      RunTimeError(std::string("PrepareEngineTables: table too large for ") + t.name);
    if (!table.empty()) memcpy(pbuff + m_globsBuffHeader[t.offs], table.data(), sizeof(int32_t) * table.size());
  }
  if (!m_lightSelectTableRev.empty()) {
    memcpy(pbuff + m_globsBuffHeader[HG_LSEL_REV_OFFS], m_lightSelectTableRev.data(), 4 * m_lightSelectTableRev.size());
    memcpy(pbuff + m_globsBuffHeader[HG_LSEL_FWD_OFFS], m_lightSelectTableFwd.data(), 4 * m_lightSelectTableFwd.size());
  }
}

void IHWLayer::SetAllPODLights(PlainLight* a_lights, size_t a_number) {
  m_globsBuffHeader[HG_LIGHTS_SIZE] = int(a_number) * HL_FLOATS;  // reference stores bytes here first, then words (Assembler.cpp:392,446)
  PrepareEngineGlobals();
  if (a_number > 0)
    memcpy(m_cdataPrepared.data() + m_globsBuffHeader[HG_LIGHTS_OFFS], a_lights, sizeof(float) * HL_FLOATS * a_number);

  int skyLightOffset = -1;
  for (size_t i = 0; i < a_number; i++) {
    int32_t type;
    memcpy(&type, a_lights + i * HL_FLOATS + HL_TYPE, 4);
    if (type == HLT_SKY_DOME) { skyLightOffset = int(i); break; }
  }
  // suns: directional lights with a soft shadow, which a path that leaves through a sky portal may look into (lightGetIntensity, clight.h:1670-1690).
  // sic: every slot finds the same first such light again, so a scene with one has MAX_SUN_NUM copies of it (Assembler.cpp:422-443)
  int sunNumber = 0;
  for (int sunId = 0; sunId < 8 /*MAX_SUN_NUM, cfetch.h:18*/; sunId++) {
    int sunCurrOffset = -1;
    for (size_t i = 0; i < a_number; i++) {
      int32_t type;
      memcpy(&type, a_lights + i * HL_FLOATS + HL_TYPE, 4);
      if (type == HLT_DIRECT && a_lights[i * HL_FLOATS + HL_DIRECT_SSOFTNESS] > 1e-6f) { sunCurrOffset = int(i); break; }
    }
    if (sunCurrOffset < 0) break;
    memcpy(m_globsBuffHeader.data() + HG_SUNS + sunId * HL_FLOATS, a_lights + size_t(sunCurrOffset) * HL_FLOATS, sizeof(float) * HL_FLOATS);
    sunNumber++;
  }
  m_globsBuffHeader[HG_SKY_LIGHT_ID] = skyLightOffset;
  m_globsBuffHeader[HG_LIGHTS_NUM] = int(a_number);
  m_globsBuffHeader[HG_SUN_NUMBER] = sunNumber;
}

// ------------------------------------------------------------------------------------------ SharedDataLayer
SharedDataLayer::SharedDataLayer(int w, int h, int a_flags) : m_initFlags(a_flags) {
  m_width = w;
  m_height = h;
}
SharedDataLayer::~SharedDataLayer() {}

void SharedDataLayer::Clear(CLEAR_FLAGS) {
  for (auto& t : m_bvhTrees) t = TreeCopy();
  m_bvhTreesNum = 0;
  m_instMatrices.clear();
  m_instLightInstId.clear();
}

IMemoryStorage* SharedDataLayer::CreateMemStorage(uint64_t a_maxSizeInBytes, const char* a_name) {
  IMemoryStorage* st = new IMemoryStorage(a_name);
  (void)a_maxSizeInBytes;
  m_allMemStorages[a_name] = st;
  return st;
}

void SharedDataLayer::SetAllBVH4(const ConvertionResult& cr, IBVHBuilder2* a_inBuilderAPI, int) {
  if (cr.treesNum <= 0 || cr.treesNum > MAXBVHTREES) RunTimeError("SetAllBVH4: converted layout with 1..4 trees is required by the HIP layer");
  for (int i = 0; i < cr.treesNum; i++) {
    if (cr.pBVH[i] == nullptr || cr.pTriangleData[i] == nullptr) RunTimeError("SetAllBVH4: null tree data");
    m_bvhTrees[i].m_bvh.assign(cr.pBVH[i], cr.pBVH[i] + cr.nodesNum[i]);
    m_bvhTrees[i].m_tris.assign(cr.pTriangleData[i], cr.pTriangleData[i] + size_t(cr.trif4Num[i]) * 4);
    if (cr.pTriangleAlpha[i] != nullptr)
      m_bvhTrees[i].m_atbl.assign(cr.pTriangleAlpha[i], cr.pTriangleAlpha[i] + size_t(cr.triAfNum[i]) * 2);
    else
      m_bvhTrees[i].m_atbl.clear();
    m_bvhTrees[i].haveInst = (std::string(cr.bvhType[i] ? cr.bvhType[i] : "") == "object");
  }
  for (int i = cr.treesNum; i < MAXBVHTREES; i++) m_bvhTrees[i] = TreeCopy();
  m_bvhTreesNum = cr.treesNum;
  if (FindStorage("geom") == nullptr) RunTimeError("SharedDataLayer::SetAllBVH4: memory storage for 'geom' not found");
}

void SharedDataLayer::SetAllInstMatrices(const float4x4* a_matrices, int32_t n) {
  m_instMatrices.resize(size_t(n) * 16);
  if (n > 0) memcpy(m_instMatrices.data(), a_matrices, size_t(n) * 64);   // copied: the driver may free its array
}
void SharedDataLayer::SetAllInstLightInstId(const int32_t* ids, int32_t n) {
  m_instLightInstId.assign(ids, ids + (n > 0 ? n : 0));
}
void SharedDataLayer::SetAllRemapLists(const int* a_allLists, const int2* a_table, int a_allSize, int a_tableSize) {
  const int* a_tableInt2 = reinterpret_cast<const int*>(a_table);
  m_remapLists.assign(a_allLists, a_allLists + a_allSize);
  m_remapTable.assign(a_tableInt2, a_tableInt2 + size_t(a_tableSize) * 2);
}
void SharedDataLayer::SetAllInstIdToRemapId(const int* a_allInstId, int a_instNum) {
  m_remapInst.assign(a_allInstId, a_allInstId + a_instNum);
}

// the layer without a device (buffers for the CPU oracle): the energy tables come from a file of a device bake when the environment
// names one -- 4 096 + 262 144 little-endian u16, GGX table first (tests/conftest.py writes it from tests/golden/energy_tables.npz)
IHWLayer* CreateHostBlobImpl(int w, int h, int a_flags) {
  SharedDataLayer* layer = new SharedDataLayer(w, h, a_flags);
  if (const char* path = getenv("HYDRA_AMD_ENERGY_TABLES")) {
    std::vector<uint16_t> t(4096 + 262144);
    if (FILE* f = fopen(path, "rb")) {
      const size_t got = fread(t.data(), sizeof(uint16_t), t.size(), f);
      fclose(f);
      if (got == t.size()) layer->SetEnergyTables(t.data(), t.data() + 4096);
    }
  }
  return layer;
}

}  // namespace hydra_host
