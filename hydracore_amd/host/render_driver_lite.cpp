// render_driver_lite.cpp -- scene-library reader + RenderDriverRTE-compatible packing (see header for citations).
#include "render_driver_lite.h"
#include <fstream>
#include <sstream>
#include <functional>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <set>

namespace hydra_host {

namespace {

typedef std::vector<float> PlainMaterialVec;   // n * 192 floats

inline void put_i(float* data, int idx, int32_t v) { memcpy(data + idx, &v, 4); }
inline int32_t get_i(const float* data, int idx) { int32_t v; memcpy(&v, data + idx, 4); return v; }

struct Sampler {               // SWTexSampler, hydra_drv/cfetch.h:108-131
  int32_t flags = 0; float gamma = 2.2f; int32_t texId = int32_t(HYDRA_INVALID_TEXTURE); int32_t dummy2 = 0;
  float row0[4] = {1, 0, 0, 0}, row1[4] = {0, 1, 0, 0};
};
inline void put_sampler_raw(float* data, int offset, const Sampler& s) {
  put_i(data, offset + HS_FLAGS, s.flags);
  data[offset + HS_GAMMA] = s.gamma;
  put_i(data, offset + HS_TEXID, s.texId);
  put_i(data, offset + HS_DUMMY, s.dummy2);
  memcpy(data + offset + HS_ROW0, s.row0, 16);
  memcpy(data + offset + HS_ROW1, s.row1, 16);
}
// IMaterial::PutSamplerAt, hydra_drv/AbstractMaterial.h:83-92
inline void put_sampler_at(float* data, int32_t texId, const Sampler& s, int texSlot, int matrixSlot, int offset) {
  const int32_t samplerOffset = (texId == int32_t(HYDRA_INVALID_TEXTURE)) ? int32_t(HYDRA_INVALID_TEXTURE) : offset / 4;
  put_i(data, texSlot, texId);
  put_i(data, matrixSlot, samplerOffset);
  put_sampler_raw(data, offset, s);
}
// IMaterial::IMaterial, AbstractMaterial.h:32-68
void init_material_node(float* d) {
  memset(d, 0, sizeof(float) * HM_NODE_FLOATS);
  const int32_t INV = int32_t(HYDRA_INVALID_TEXTURE);
  put_i(d, HM_EMISSIVE_LIGHTID, -1);
  Sampler dummy;
  put_sampler_at(d, INV, dummy, HM_NORMAL_TEX, HM_NORMAL_TEX_MATRIX, HM_NORMAL_SAMPLER);
  put_sampler_at(d, INV, dummy, HM_OPACITY_TEX, HM_OPACITY_TEX_MATRIX, HM_OPACITY_SAMPLER);
  put_sampler_at(d, INV, dummy, HM_EMISSIVE_TEXID, HM_EMISSIVE_TEXMATRIXID, HM_EMISSIVE_SAMPLER);
  for (int i = 0; i < 16; i++) put_i(d, HM_PROC_TEX_IDS + i, INV);
  put_i(d, HM_AO_TYPE, 0); put_i(d, HM_AO_TEX_ID, INV); put_i(d, HM_AO_TEXMATRIX_ID, INV); d[HM_AO_LENGTH] = 0.0f;
  put_i(d, HM_AO_TYPE2, 0); put_i(d, HM_AO_TEX_ID2, INV); put_i(d, HM_AO_TEXMATRIX_ID2, INV); d[HM_AO_LENGTH2] = 0.0f;
}

bool parse_floats(const std::string& s, float* out, int n) {
  std::istringstream in(s);
  for (int i = 0; i < n; i++) {
    std::string tok;
    if (!(in >> tok)) return false;
    out[i] = strtof(tok.c_str(), nullptr);   // tolerates "0.25f"
  }
  return true;
}
// HydraXMLHelpers::ReadValue3f / ReadValue1f (HydraAPI, absent): value in attribute "val" or in the node text
float3 read_value3f(const XmlNode* n) {
  float v[3] = {0, 0, 0};
  if (n == nullptr) return float3(0, 0, 0);
  std::string s = n->has_attr("val") ? std::string(n->attr("val")) : n->text;
  if (!parse_floats(s, v, 3)) {
    float one = 0.0f;
    if (parse_floats(s, &one, 1)) v[0] = v[1] = v[2] = one;
  }
  return float3(v[0], v[1], v[2]);
}
float read_value1f(const XmlNode* n) {
  if (n == nullptr) return 0.0f;
  float v = 0.0f;
  parse_floats(n->has_attr("val") ? std::string(n->attr("val")) : n->text, &v, 1);
  return v;
}

// SamplerNode / SamplerFromTexref, hydra_drv/PlainMaterialConverter.cpp:886-951
const XmlNode* sampler_node(const XmlNode* a) {
  const XmlNode* inside = xchild(xchild(a, "color"), "texture");
  return inside ? inside : xchild(a, "texture");
}
Sampler sampler_from_texref(const XmlNode* a, bool allowAlphaToRGB = false) {
  Sampler res;
  res.texId = a->attr_int("id");
  res.flags = 0;
  res.gamma = a->has_attr("input_gamma") ? a->attr_float("input_gamma") : 2.2f;
  if (a->has_attr("matrix")) {
    float m[16];
    if (parse_floats(a->attr("matrix"), m, 16)) { memcpy(res.row0, m, 16); memcpy(res.row1, m + 4, 16); }
  }
  const std::string modeU = a->attr("addressing_mode_u"), modeV = a->attr("addressing_mode_v"), modeS = a->attr("filter");
  const std::string alphaSrc = a->attr("input_alpha"), channel = a->attr("channel");
  if (modeU == "clamp") res.flags |= HTEX_CLAMP_U;
  if (modeV == "clamp") res.flags |= HTEX_CLAMP_V;
  if (modeS == "point" || modeS == "nearest") res.flags |= HTEX_POINT_SAM;
  if (allowAlphaToRGB && alphaSrc == "alpha") res.flags |= HTEX_ALPHASRC_W;
  if (channel == "1") res.flags |= HTEX_COORD_SECOND;
  else if (channel == "camera_mapped") res.flags |= HTEX_COORD_CAM_PROJ;
  return res;
}

struct MatTree {                      // RAYTR::IMaterial tree, flattened by flatten()
  float plain[HM_NODE_FLOATS];
  std::shared_ptr<MatTree> c1, c2;    // blend children
  bool isBlend = false;
};
typedef std::shared_ptr<MatTree> MatPtr;

MatPtr new_node() { auto p = std::make_shared<MatTree>(); init_material_node(p->plain); return p; }

// LambertMaterial, PlainMaterialConverter.cpp:96-136
MatPtr make_lambert(float3 color, int32_t texId, const Sampler& s);
// TranslucentMaterial, PlainMaterialConverter.cpp:182-213: the lambert node's slots with class TRANSLUCENT
MatPtr make_translucent(float3 color, int32_t texId, const Sampler& s) {
  MatPtr m = make_lambert(color, texId, s);
  put_i(m->plain, HM_TYPE, HMT_TRANSLUCENT);
  put_i(m->plain, HM_FLAGS, HMF_HAS_DIFFUSE);
  return m;
}
MatPtr make_lambert(float3 color, int32_t texId, const Sampler& s) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  put_sampler_at(d, texId, s, HM_TEXID, HM_TEXMATRIXID, HM_LAMBERT_SAMPLER);
  put_i(d, HM_TYPE, HMT_LAMBERT);
  put_i(d, HM_FLAGS, HMF_HAS_DIFFUSE);
  return p;
}
// OrenNayarMaterial, PlainMaterialConverter.cpp:137-170
MatPtr make_orennayar(float3 color, float roughness, int32_t texId, const Sampler& s) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  const float sigma = roughness * (3.14159265358979323846f / 2.0f), sigma2 = sigma * sigma;
  d[HM_ORENNAYAR_ROUGHNESS] = roughness;
  d[HM_ORENNAYAR_A] = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
  d[HM_ORENNAYAR_B] = 0.45f * sigma2 / (sigma2 + 0.09f);
  put_sampler_at(d, texId, s, HM_TEXID, HM_TEXMATRIXID, HM_ORENNAYAR_SAMPLER);
  put_i(d, HM_TYPE, HMT_OREN_NAYAR);
  put_i(d, HM_FLAGS, HMF_HAS_DIFFUSE);
  return p;
}
// PhongMaterial, PlainMaterialConverter.cpp:414-460
MatPtr make_phong(float3 color, int32_t texId, const Sampler& sc, float cosPower, int32_t glossTexId, const Sampler& sg, float gloss) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  d[HM_PHONG_COSPOWER] = cosPower;
  d[HM_PHONG_GLOSINESS] = gloss;
  put_sampler_at(d, texId, sc, HM_TEXID, HM_TEXMATRIXID, HM_PHONG_SAMPLER0);
  put_sampler_at(d, glossTexId, sg, HM_PHONG_GLOSS_TEXID, HM_PHONG_GLOSS_TEXMATRIXID, HM_PHONG_SAMPLER1);
  put_i(d, HM_TYPE, HMT_PHONG);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS);
  return p;
}
// BlinnTorranceSrappowMaterial, PlainMaterialConverter.cpp:462-498: the phong offsets plus the anisotropy value at 19, which the shading never reads
MatPtr make_blinn(float3 color, int32_t texId, const Sampler& sc, float cosPower, int32_t glossTexId, const Sampler& sg, float gloss, float aniso) {
  MatPtr p = make_phong(color, texId, sc, cosPower, glossTexId, sg, gloss);
  p->plain[HM_BLINN_ANISOTROPY] = aniso;
  put_i(p->plain, HM_TYPE, HMT_BLINN);
  return p;
}
// BeckmannMaterial / TRGGXMaterial, PlainMaterialConverter.cpp:500-631: one layout (cmaterial.h:1531-1556); the TRGGX constructor does not
// store the texture ids of its anisotropy and rotation samplers (only their sampler slots), which the shading never reads
MatPtr make_aniso(bool trggx, float3 color, const Sampler& sc, float cosPower, const Sampler& sg, float gloss, const Sampler& sa, float aniso, const Sampler& sr, float rot, bool flipAxis) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  d[HM_PHONG_COSPOWER] = cosPower;
  d[HM_PHONG_GLOSINESS] = gloss;
  d[HM_BECKMANN_ANISOTROPY] = aniso;
  d[HM_BECKMANN_ANISO_ROT] = rot;
  put_sampler_at(d, sc.texId, sc, HM_TEXID, HM_TEXMATRIXID, HM_PHONG_SAMPLER0);
  put_sampler_at(d, sg.texId, sg, HM_PHONG_GLOSS_TEXID, HM_PHONG_GLOSS_TEXMATRIXID, HM_PHONG_SAMPLER1);
  put_sampler_at(d, sa.texId, sa, HM_BECKMANN_ANISO_TEXID, HM_BECKMANN_ANISO_TEXMATRIXID, HM_BECKMANN_SAMPLER2);
  put_sampler_at(d, sr.texId, sr, HM_BECKMANN_ROT_TEXID, HM_BECKMANN_ROT_TEXMATRIXID, HM_BECKMANN_SAMPLER3);
  if (trggx) { put_i(d, HM_BECKMANN_ANISO_TEXID, 0); put_i(d, HM_BECKMANN_ROT_TEXID, 0); }
  put_i(d, HM_TYPE, trggx ? HMT_TRGGX : HMT_BECKMANN);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS | (flipAxis ? HMF_FLIP_TANGENT : 0));
  return p;
}
// GGXMaterial, PlainMaterialConverter.cpp:635-680 (the anisotropy arguments of the constructor are never stored)
MatPtr make_ggx(float3 color, int32_t texId, const Sampler& sc, float cosPower, int32_t glossTexId, const Sampler& sg, float gloss, float ior) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  d[HM_GGX_COSPOWER] = cosPower;
  d[HM_GGX_GLOSINESS] = gloss;
  d[HM_GGX_FRESNEL_IOR] = ior;
  put_sampler_at(d, texId, sc, HM_TEXID, HM_TEXMATRIXID, HM_GGX_SAMPLER0);
  put_sampler_at(d, glossTexId, sg, HM_GGX_GLOSS_TEXID, HM_GGX_GLOSS_TEXMATRIXID, HM_GGX_SAMPLER1);
  put_i(d, HM_TYPE, HMT_GGX);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS);
  return p;
}
// MirrorMaterial, PlainMaterialConverter.cpp:219-247
MatPtr make_mirror(float3 color, int32_t texId, const Sampler& s) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  put_sampler_at(d, texId, s, HM_TEXID, HM_TEXMATRIXID, HM_MIRROR_SAMPLER);
  put_i(d, HM_TYPE, HMT_MIRROR);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS);
  return p;
}
// ThinGlassMaterial, PlainMaterialConverter.cpp:254-300
MatPtr make_thinglass(float3 color, int32_t texId, const Sampler& sc, float cosPower, float gloss, int32_t glossTexId, const Sampler& sg) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  d[HM_THINGLASS_COS_POWER] = cosPower;
  d[HM_THINGLASS_GLOSINESS] = gloss;
  put_sampler_at(d, texId, sc, HM_TEXID, HM_TEXMATRIXID, HM_THINGLASS_SAMPLER0);
  put_sampler_at(d, glossTexId, sg, HM_THINGLASS_GLOSS_TEXID, HM_THINGLASS_GLOSS_TEXMATRIXID, HM_THINGLASS_SAMPLER1);
  put_i(d, HM_TYPE, HMT_THIN_GLASS);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS | HMF_HAS_TRANSPARENCY);
  return p;
}
// GlassMaterial, PlainMaterialConverter.cpp:363-410
MatPtr make_glass(float3 color, int32_t texId, const Sampler& sc, float ior, float3 fogColor, float fogMult, float cosPower, float gloss,
                  int32_t glossTexId, const Sampler& sg) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_COLOR] = color.x; d[HM_COLOR + 1] = color.y; d[HM_COLOR + 2] = color.z;
  d[HM_GLASS_FOG_COLOR] = fogColor.x; d[HM_GLASS_FOG_COLOR + 1] = fogColor.y; d[HM_GLASS_FOG_COLOR + 2] = fogColor.z;
  d[HM_GLASS_FOG_MULT] = fogMult;
  d[HM_GLASS_IOR] = ior;
  d[HM_GLASS_COS_POWER] = cosPower;
  d[HM_GLASS_GLOSINESS] = gloss;
  put_sampler_at(d, texId, sc, HM_TEXID, HM_TEXMATRIXID, HM_GLASS_SAMPLER0);
  put_sampler_at(d, glossTexId, sg, HM_GLASS_GLOSS_TEXID, HM_GLASS_GLOSS_TEXMATRIXID, HM_GLASS_SAMPLER1);
  put_i(d, HM_TYPE, HMT_GLASS);
  put_i(d, HM_FLAGS, HMF_CAST_CAUSTICS | HMF_HAS_TRANSPARENCY);
  return p;
}
// EmissiveMaterial, PlainMaterialConverter.cpp:31-70
MatPtr make_emissive(float3 color, int32_t texId, const Sampler& s, int32_t lightId) {
  MatPtr p = new_node();
  float* d = p->plain;
  d[HM_EMISSIVE_COLOR] = color.x; d[HM_EMISSIVE_COLOR + 1] = color.y; d[HM_EMISSIVE_COLOR + 2] = color.z;
  put_sampler_at(d, texId, s, HM_EMISSIVE_TEXID, HM_EMISSIVE_TEXMATRIXID, HM_EMISSIVE_SAMPLER);
  put_i(d, HM_EMISSIVE_LIGHTID, lightId);
  put_i(d, HM_TYPE, HMT_EMISSIVE);
  return p;
}
// BlendMaskMaterial, PlainMaterialConverter.cpp:750-791.  Field aliasing reproduced byte for byte:
// BLEND_TYPE written first, then the 12-word sampler lands on words 20..31, then FALOFF_OFFSET/SIZE overwrite 19/20.
MatPtr make_blend(MatPtr m1, MatPtr m2, float3 alpha, int32_t alphaTex, const Sampler& s, bool isFresnel, bool vrayLike, int extrusion, float ior) {
  MatPtr p = new_node();
  float* d = p->plain;
  p->isBlend = true; p->c1 = m1; p->c2 = m2;
  d[HM_COLOR] = alpha.x; d[HM_COLOR + 1] = alpha.y; d[HM_COLOR + 2] = alpha.z;
  d[HM_BLEND_FRESNEL_IOR] = isFresnel ? ior : 1.0f;
  put_i(d, HM_BLEND_TYPE, isFresnel ? 1 : 3);
  put_sampler_at(d, alphaTex, s, HM_TEXID, HM_TEXMATRIXID, HM_BLEND_SAMPLER);
  put_i(d, HM_TYPE, HMT_BLEND_MASK);
  put_i(d, HM_BLEND_FALOFF_OFFSET, -1);
  put_i(d, HM_BLEND_FALOFF_SIZE, 0);
  int32_t flags = isFresnel ? HBF_FRESNEL : 0;
  if (vrayLike && !isFresnel) flags |= HBF_REFLECTION_WEIGHT_IS_ONE;
  flags |= extrusion;
  put_i(d, HM_BLEND_FLAGS, flags);
  return p;
}
// BlendMaskMaterial::ConvertToPlainMaterial, PlainMaterialConverter.cpp:793-811: [blend | subtree1 | subtree2], relative offsets
PlainMaterialVec flatten(const MatPtr& m) {
  PlainMaterialVec res(m->plain, m->plain + HM_NODE_FLOATS);
  if (!m->isBlend) return res;
  PlainMaterialVec d1 = flatten(m->c1), d2 = flatten(m->c2);
  put_i(res.data(), HM_BLEND_MAT1, int32_t(res.size() / HM_NODE_FLOATS));
  res.insert(res.end(), d1.begin(), d1.end());
  put_i(res.data(), HM_BLEND_MAT2, int32_t(res.size() / HM_NODE_FLOATS));
  res.insert(res.end(), d2.begin(), d2.end());
  return res;
}

int read_extrusion(const XmlNode* n) {   // ReadExtrusionType, PlainMaterialConverter.cpp:1201-1216
  if (n == nullptr) return HBF_EXTRUSION_STRONG;
  const std::string e = xattr(xchild(n, "extrusion"), "val");
  if (e == "luminance") return HBF_EXTRUSION_LUMINANCE;
  if (e == "colored") return 0;
  return HBF_EXTRUSION_STRONG;
}
float read_fresnel_ior(const XmlNode* n) {  // ReadFresnelIOR, :1218-1226
  if (n == nullptr) return 1.5f;
  if (n->child("fresnel_ior")) return n->child("fresnel_ior")->attr_float("val");
  return n->child("fresnel_IOR") ? n->child("fresnel_IOR")->attr_float("val") : 0.0f;
}

bool read_file(const std::string& path, std::vector<char>& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  out.resize(size_t(f.tellg()));
  f.seekg(0);
  f.read(out.data(), out.size());
  return true;
}

// PushDownNormalMaps, PlainMaterialConverter.cpp:1240-1282: ids, sampler and the invert flags of a normal map go to every leaf of the blend tree
void push_down_normal_map(MatTree* node, int32_t auxId, int32_t samplerOffset, int flags, const Sampler& sm) {
  const int mask = HMF_INVERT_NMAP_X | HMF_INVERT_NMAP_Y | HMF_INVERT_SWAP_NMAP_XY | HMF_INVERT_HEIGHT;
  if (node->isBlend) {
    if (!node->c1 || !node->c2) return;
    for (MatTree* child : {node->c1.get(), node->c2.get()}) {
      const int32_t own = get_i(child->plain, HM_NORMAL_TEX);
      if (uint32_t(own) != HYDRA_INVALID_TEXTURE) push_down_normal_map(child, own, get_i(child->plain, HM_NORMAL_TEX_MATRIX), get_i(child->plain, HM_FLAGS) & mask, sm);
      else push_down_normal_map(child, auxId, samplerOffset, flags, sm);
    }
  } else if (uint32_t(auxId) != HYDRA_INVALID_TEXTURE) {
    put_i(node->plain, HM_NORMAL_TEX, auxId);
    put_i(node->plain, HM_NORMAL_TEX_MATRIX, samplerOffset);
    put_i(node->plain, HM_FLAGS, (get_i(node->plain, HM_FLAGS) & ~mask) | flags);
  }
  put_sampler_at(node->plain, auxId, sm, HM_NORMAL_TEX, HM_NORMAL_TEX_MATRIX, HM_NORMAL_SAMPLER);   // IMaterial::SetNormalSampler
}

}  // namespace

// ================================================================================================
RenderDriverLite::RenderDriverLite(IHWLayer* a_layer, int w, int h) : m_pHWLayer(a_layer), m_width(w), m_height(h) {}

RenderDriverLite::~RenderDriverLite() {
  delete m_pHWLayer;
  delete m_pTexStorage; delete m_pTexStorageAux; delete m_pGeomStorage; delete m_pMaterialStorage; delete m_pPdfStorage;
}

void RenderDriverLite::Unsupported(const std::string& what) {
  m_unsupported++;
  m_log += "[unsupported] " + what + "\n";
}

void RenderDriverLite::AllocAll(int imgNum, int matNum, int lightNum, int meshNum) {
  m_pTexStorage = m_pHWLayer->CreateMemStorage(0, "textures");
  m_pTexStorageAux = m_pHWLayer->CreateMemStorage(0, "textures_aux");
  m_pGeomStorage = m_pHWLayer->CreateMemStorage(0, "geom");
  m_pMaterialStorage = m_pHWLayer->CreateMemStorage(0, "materials");
  m_pPdfStorage = m_pHWLayer->CreateMemStorage(0, "pdfs");
  m_pHWLayer->ResizeTablesForEngineGlobals(meshNum, imgNum, matNum, lightNum);

  // white diffuse dummy in the last material slot, first in the arena (RenderDriverRTE.cpp:718-727)
  PlainMaterialVec white = flatten(make_lambert(float3(1, 1, 1), int32_t(HYDRA_INVALID_TEXTURE), Sampler()));
  const int32_t whiteOffs = m_pMaterialStorage->Update(matNum - 1, white.data(), white.size() * sizeof(float));
  auto vars = m_pHWLayer->GetAllFlagsAndVars();
  vars.m_varsI[8 /*HRT_WHITE_DIFFUSE_OFFSET*/] = whiteOffs;
  m_pHWLayer->SetAllFlagsAndVars(vars);
}

bool RenderDriverLite::UpdateImage(int32_t a_texId, int32_t w, int32_t h, int32_t bpp, int32_t chan, const void* a_data) {
  if (a_data == nullptr) return false;
  int32_t header[4] = {w, h, chan, bpp};   // SWTextureHeader, cfetch.h:96-105
  const size_t inBytes = size_t(w) * size_t(h) * size_t(bpp);
  const size_t headerSize = 16;
  const size_t total = ((inBytes + 15) / 16) * 16 + headerSize;
  if (m_pTexStorage->Update(a_texId, nullptr, total) == -1) return false;
  m_pTexStorage->UpdatePartial(a_texId, header, 0, 16);
  m_pTexStorage->UpdatePartial(a_texId, a_data, headerSize, inBytes);
  return true;
}

// the aux copy of a normal map: one per texture id (the reference keys on texture id + bump parameters, which normal_bump does not have);
// UpdateImageAux, RenderDriverRTE_AuxTextures.cpp:195-216; ids count up from 0 (AuxNormalTexPerMaterial, PlainMaterialConverter.cpp:1883-1902)
// GetCachedAuxNormalMatId + GetAuxNormalMapFromDisaplacement for height maps (RenderDriverRTE_AuxTextures.cpp:46-160): the layer turns the 8-bit RGBA
// height texture into a normal map (IHWLayer::NormalMapFromDisplacement), cached per (texture, amount, smoothing)
int32_t RenderDriverLite::AuxNormalMapFromHeight(int32_t texId, int32_t a_matId, float bumpAmt, float smoothLvl) {
  const std::string key = std::to_string(texId) + " " + std::to_string(bumpAmt) + " " + std::to_string(smoothLvl);
  auto found = m_auxHeightMaps.find(key);
  if (found != m_auxHeightMaps.end()) return found->second;
  int32_t auxId = int32_t(HYDRA_INVALID_TEXTURE);
  const std::vector<int32_t> table = m_pTexStorage->GetTable();
  if (texId >= 0 && size_t(texId) < table.size() && table[size_t(texId)] >= 0) {
    const int32_t* header = reinterpret_cast<const int32_t*>(m_pTexStorage->GetBegin()) + size_t(table[size_t(texId)]) * 4;
    if (header[3] != 4) Unsupported("height map " + std::to_string(texId) + " is not an 8-bit RGBA texture (material " + std::to_string(a_matId) + ")");
    else {
      const std::vector<uchar4> normals = m_pHWLayer->NormalMapFromDisplacement(header[0], header[1], reinterpret_cast<const uchar4*>(header + 4), bumpAmt, true /* PLAIN_MATERIAL_INVERT_HEIGHT is set by then, PlainMaterialConverter.cpp:1372 */, smoothLvl);
      if (normals.size() != size_t(header[0]) * size_t(header[1]))
        Unsupported("height_bump (material " + std::to_string(a_matId) + "): this layer has no NormalMapFromDisplacement (a device is needed)");
      else {
        auxId = m_auxImageNumber++;
        const size_t inBytes = normals.size() * 4, headerSize = 16, total = ((inBytes + 15) / 16) * 16 + headerSize;
        const int32_t auxHeader[4] = {header[0], header[1], 4, 4};
        m_pTexStorageAux->Update(auxId, nullptr, total);
        m_pTexStorageAux->UpdatePartial(auxId, auxHeader, 0, 16);
        m_pTexStorageAux->UpdatePartial(auxId, normals.data(), headerSize, inBytes);
      }
    }
  }
  m_auxHeightMaps[key] = auxId;
  return auxId;
}

int32_t RenderDriverLite::AuxNormalMapFor(int32_t texId, int32_t a_matId) {
  auto found = m_auxNormalMaps.find(texId);
  if (found != m_auxNormalMaps.end()) return found->second;
  int32_t auxId = int32_t(HYDRA_INVALID_TEXTURE);
  const std::vector<int32_t> table = m_pTexStorage->GetTable();
  if (texId >= 0 && size_t(texId) < table.size() && table[size_t(texId)] >= 0) {
    const int32_t* header = reinterpret_cast<const int32_t*>(m_pTexStorage->GetBegin()) + size_t(table[size_t(texId)]) * 4;
    if (header[3] != 4) Unsupported("normal map " + std::to_string(texId) + " is not an 8-bit RGBA texture (material " + std::to_string(a_matId) + ")");
    else {
      auxId = m_auxImageNumber++;
      const size_t inBytes = size_t(header[0]) * size_t(header[1]) * 4, headerSize = 16, total = ((inBytes + 15) / 16) * 16 + headerSize;
      const int32_t auxHeader[4] = {header[0], header[1], 4, 4};   // SWTextureHeader: width, height, depth = 4, bpp
      m_pTexStorageAux->Update(auxId, nullptr, total);
      m_pTexStorageAux->UpdatePartial(auxId, auxHeader, 0, 16);
      m_pTexStorageAux->UpdatePartial(auxId, header + 4, headerSize, inBytes);
    }
  }
  m_auxNormalMaps[texId] = auxId;
  return auxId;
}

// CreateFromHydraMaterialXmlNode + CreateMaterialFromXmlNode, PlainMaterialConverter.cpp:1502-1738
bool RenderDriverLite::UpdateMaterial(int32_t a_matId, const XmlNode* a_node) {
  const std::string mtype = a_node->attr("type");
  if (mtype == "shadow_catcher") {   // ShadowMatteMaterial, PlainMaterialConverter.cpp:77-99, 1638-1660: a bare node of class SHADOW_MATTE; no bump, opacity or emission (:1710).
    // The CPU integrator hands its sampler a zero shadow value (PT_Loop.cpp:240), so the surface passes rays on with zero throughput; the
    // back-plate texture of <back> belongs to the OpenCL layer's environmentColorExtended and is not read here
    MatPtr pMatte = new_node();
    put_i(pMatte->plain, HM_TYPE, HMT_SHADOW_MATTE);
    if (const XmlNode* back = a_node->child("back")) {   // PlainMaterialConverter.cpp:1642-1668.  Always camera-projected: the reference compares the mode string by pointer (:1655)
      m_shadowMatteBackTexId = xchild(back, "texture") ? xchild(back, "texture")->attr_int("id") : 0;
      m_shadowMatteBackColor = float3(1, 1, 1);
      if (back->has_attr("multcolor")) { float v[3] = {1, 1, 1}; parse_floats(back->attr("multcolor"), v, 3); m_shadowMatteBackColor = float3(v[0], v[1], v[2]); }
      if (m_shadowMatteBackTexId == 0) m_shadowMatteBackTexId = int32_t(HYDRA_INVALID_TEXTURE);
      m_shadowMatteBackMode = 0;
      if (xhas(xchild(back, "texture"), "input_gamma")) m_shadowMatteBackGamma = xchild(back, "texture")->attr_float("input_gamma");
      if (back->attr_int("reflection") == 1 || back->attr_int("fix_black_triangles") == 1)
        Unsupported("shadow_catcher <back reflection / fix_black_triangles> (material " + std::to_string(a_matId) + "): camera-mapped reflections of the OpenCL layer's catcher");
    } else { m_shadowMatteBackTexId = int32_t(HYDRA_INVALID_TEXTURE); m_shadowMatteBackColor = float3(1, 1, 1); }
    if (length(read_value3f(xchild(a_node->child("emission"), "color"))) > 1e-4f)   // PLAIN_MATERIAL_EMISSIVE_SHADOW_CATCHER, PlainMaterialConverter.cpp:1676-1699: read by the OpenCL layer's NextBounce only
      Unsupported("emissive shadow_catcher (material " + std::to_string(a_matId) + ")");
    PlainMaterialVec mdata = flatten(pMatte);
    m_pMaterialStorage->Update(a_matId, mdata.data(), mdata.size() * sizeof(float));
    return true;
  }
  if (mtype == "sky_portal_mtl") {   // SkyPortalMaterial + CreateSkyPortalMaterial, PlainMaterialConverter.cpp:304-350, 1604-1614: a thin glass no shadow ray stops at;
    // the emission block of CreateMaterialFromXmlNode (:1716) skips materials with this flag, ReadBumpAndOpacity finds nothing to read on one
    const XmlNode* em = a_node->child("emission");
    const float mult = xchild(em, "multiplier") ? xchild(em, "multiplier")->attr_float("val") : 1.0f;
    float c3[3] = {0, 0, 0};
    if (xchild(em, "color")) parse_floats(xchild(em, "color")->attr("val"), c3, 3);
    MatPtr pPortal = make_thinglass(float3(c3[0], c3[1], c3[2]) * mult, int32_t(HYDRA_INVALID_TEXTURE), Sampler(), 1000000.0f, 1.0f, int32_t(HYDRA_INVALID_TEXTURE), Sampler());
    put_i(pPortal->plain, HM_FLAGS, HMF_HAS_TRANSPARENCY | HMF_SKIP_SHADOW | HMF_SKIP_SKY_PORTAL | ((a_node->attr_int("visible") == 1) ? 0 : HMF_INVIS_LIGHT));
    Opacity o;
    o.texId = int32_t(HYDRA_INVALID_TEXTURE);
    float raw[12];
    put_sampler_raw(raw, 0, Sampler());
    memcpy(o.sampler, raw, sizeof(raw));
    o.smooth = false;
    o.skipShadow = true;
    m_matOpacity[a_matId] = o;
    PlainMaterialVec mdata = flatten(pPortal);
    m_pMaterialStorage->Update(a_matId, mdata.data(), mdata.size() * sizeof(float));
    return true;
  }
  const bool isBlendOfTwo = (mtype == "hydra_blend");   // two materials of the library under a mask (CreateBlendDefferedProxyFromXmlNode, PlainMaterialConverter.cpp:1457-1500)
  if (mtype != "hydra_material" && !isBlendOfTwo) { Unsupported("material type '" + mtype + "' (id " + std::to_string(a_matId) + ")"); }

  const XmlNode* emission = a_node->child("emission");
  const XmlNode* diffuse = a_node->child("diffuse");
  const XmlNode* reflect = a_node->child("reflectivity");
  const XmlNode* transpar = a_node->child("transparency");
  const XmlNode* sss = a_node->child("translucency");
  const float3 colorE = read_value3f(xchild(emission, "color"));
  float3 colorD = read_value3f(xchild(diffuse, "color"));
  const float3 colorS = read_value3f(xchild(reflect, "color"));
  const float3 colorT = read_value3f(xchild(transpar, "color"));
  const float3 colorSSS = read_value3f(xchild(sss, "color"));
  if (const XmlNode* displ = a_node->child("displacement")) {
    const std::string btype = displ->attr("type");
    if (btype != "normal_bump" && btype != "height_bump") Unsupported("displacement type '" + btype + "' (material " + std::to_string(a_matId) + "): normal_bump and height_bump are built");
  }
  m_matOpacity.erase(a_matId);
  if (const XmlNode* op = a_node->child("opacity")) {   // PlainMaterialConverter.cpp:1429-1445: alpha-tested in the traversal, not a BxDF
    Opacity o;
    Sampler sm;
    o.texId = int32_t(HYDRA_INVALID_TEXTURE);
    if (const XmlNode* tx = xchild(op, "texture")) { sm = sampler_from_texref(tx, true); o.texId = sm.texId; }
    float raw[12];
    put_sampler_raw(raw, 0, sm);
    memcpy(o.sampler, raw, sizeof(raw));
    o.smooth = (op->attr_int("smooth") == 1);
    o.skipShadow = (xchild(op, "skip_shadow") && xchild(op, "skip_shadow")->attr_int("val") == 1) || (op->attr_int("skip_shadow") == 1);
    if (o.smooth) Unsupported("smooth opacity (material " + std::to_string(a_matId) + "): stochastic alpha is the OpenCL layer's BVH4InstTraverseAlphaS, the CPU path tests against 0.5");
    m_matOpacity[a_matId] = o;
  }
  if (length(colorD) <= 1e-5f) colorD = colorSSS;

  const bool haveFresnelRefl = (xchild(reflect, "fresnel") && xchild(reflect, "fresnel")->attr_int("val") == 1);
  const float fresnelIOR = read_fresnel_ior(reflect);
  const int reflExtrusion = read_extrusion(reflect);

  // DiffuseMaterialFromHydraMtl :980-1001
  MatPtr pMaterialD;
  {
    Sampler s; int32_t texId = int32_t(HYDRA_INVALID_TEXTURE);
    if (sampler_node(diffuse)) { s = sampler_from_texref(sampler_node(diffuse)); texId = s.texId; }
    if (std::string(xattr(diffuse, "brdf_type")) == "orennayar")
      pMaterialD = make_orennayar(read_value3f(xchild(diffuse, "color")), read_value1f(xchild(diffuse, "roughness")), texId, s);
    else
      pMaterialD = make_lambert(read_value3f(xchild(diffuse, "color")), texId, s);
    // DiffuseAndTranslucentBlendMaterialFromHydraMtl :1024-1059: translucency alone replaces the diffuse node, both blend by the translucency colour
    // (plain mask, strong extrusion); TranslucentMaterialFromHydraMtl :1003-1022 halves the colour
    if (length(colorSSS) > 1e-5f) {
      Sampler st; int32_t ttexId = int32_t(HYDRA_INVALID_TEXTURE);
      if (sampler_node(sss)) { st = sampler_from_texref(sampler_node(sss)); ttexId = st.texId; }
      MatPtr pTrans = make_translucent(colorSSS * 0.5f, ttexId, st);
      if (length(read_value3f(xchild(diffuse, "color"))) > 1e-5f) pMaterialD = make_blend(pTrans, pMaterialD, colorSSS, ttexId, st, false, true, HBF_EXTRUSION_STRONG, 1.5f);
      else pMaterialD = pTrans;
    }
  }
  // ReflectiveMaterialFromHydraMtl :1061-1149
  MatPtr pMaterialS;
  int32_t texReflId = int32_t(HYDRA_INVALID_TEXTURE);
  Sampler samplRefl;
  {
    const XmlNode* gloss = xchild(reflect, "glossiness");
    const float glossVal = read_value1f(gloss);
    Sampler sg; int32_t texGloss = int32_t(HYDRA_INVALID_TEXTURE);
    if (sampler_node(reflect)) { samplRefl = sampler_from_texref(sampler_node(reflect)); texReflId = samplRefl.texId; }
    if (sampler_node(gloss)) { sg = sampler_from_texref(sampler_node(gloss)); texGloss = sg.texId; }
    const std::string brdf = xattr(reflect, "brdf_type");
    if (texGloss == int32_t(HYDRA_INVALID_TEXTURE) && glossVal >= 0.995f)
      pMaterialS = make_mirror(colorS, texReflId, samplRefl);
    else if (brdf == "torranse_sparrow") {   // sic, PlainMaterialConverter.cpp:1130
      const XmlNode* an = xchild(reflect, "anisotropy");
      pMaterialS = make_blinn(colorS, texReflId, samplRefl, 0.0f, texGloss, sg, glossVal, an ? read_value1f(an) : 0.0f);
    }
    else if (brdf == "beckmann" || brdf == "trggx" || brdf == "TRGGX") {   // PlainMaterialConverter.cpp:1068-1142
      const XmlNode* an = xchild(reflect, "anisotropy");
      Sampler sa, sr;                    // DummySampler: texId = INVALID_TEXTURE
      if (an && sampler_node(an)) sa = sampler_from_texref(sampler_node(an));
      if (an && xchild(an, "texture_rot")) sr = sampler_from_texref(xchild(an, "texture_rot"));
      pMaterialS = make_aniso(brdf != "beckmann", colorS, samplRefl, 0.0f, sg, glossVal, sa, an ? read_value1f(an) : 0.0f, sr,
                              an ? an->attr_float("rot") : 0.0f, an ? an->attr_int("flip_axis") == 1 : false);
    }
    else if (brdf == "ggx" || brdf == "GGX")
      pMaterialS = make_ggx(colorS, texReflId, samplRefl, 0.0f, texGloss, sg, glossVal, fresnelIOR);
    else {
      if (length(colorS) > 1e-5f && brdf != "phong" && brdf != "")
        Unsupported("reflectivity brdf_type '" + brdf + "' (material " + std::to_string(a_matId) + "), packed as phong");
      pMaterialS = make_phong(colorS, texReflId, samplRefl, 0.0f, texGloss, sg, glossVal);
    }
    const XmlNode* efix = xchild(reflect, "energy_fix");
    if (!efix) efix = xchild(reflect, "multiscatter_fix");
    if (!efix) efix = xchild(reflect, "multiscatter");
    if (efix && efix->attr_int("val") == 1) {
      put_i(pMaterialS->plain, HM_FLAGS, get_i(pMaterialS->plain, HM_FLAGS) | HMF_ENERGY_FIX);
      // on a GGX node the flag makes the shading read EngineGlobals::m_essGgx2017Table, which the layer supplies (IHWLayer::SetEnergyTables:
      // baked on the device; a layer without a device has it only when it was given a bake)
      if (get_i(pMaterialS->plain, HM_TYPE) == HMT_GGX && length(colorS) > 1e-5f && !m_pHWLayer->HaveEnergyTables())
        Unsupported("GGX multi-scattering (material " + std::to_string(a_matId) + "): this layer has no energy tables (no device, and HYDRA_AMD_ENERGY_TABLES names no bake)");
    }
  }
  // TransparentMaterialFromHydraMtl :1151-1199.  The fog (Beer) term the glass node carries is stored as the reference
  // stores it; IntegratorMISPTLoop2 never reads it (no materialLeafGetFog call on that path).
  MatPtr pMaterialT;
  int32_t texTranspId = int32_t(HYDRA_INVALID_TEXTURE);
  Sampler samplTransp;
  {
    const XmlNode* gloss = xchild(transpar, "glossiness");
    const float3 fogColor = read_value3f(xchild(transpar, "fog_color"));
    const float fogMult = read_value1f(xchild(transpar, "fog_multiplier"));
    const float glossVal = read_value1f(gloss), iorVal = read_value1f(xchild(transpar, "ior"));
    const bool thinWall = (xchild(transpar, "thin_walled") && xchild(transpar, "thin_walled")->attr_int("val") == 1);
    Sampler sg; int32_t texGloss = int32_t(HYDRA_INVALID_TEXTURE);
    if (sampler_node(transpar)) { samplTransp = sampler_from_texref(sampler_node(transpar)); texTranspId = samplTransp.texId; }
    if (sampler_node(gloss)) { sg = sampler_from_texref(sampler_node(gloss)); texGloss = sg.texId; }
    if (fabsf(iorVal) < 1e-4f || thinWall) pMaterialT = make_thinglass(colorT, texTranspId, samplTransp, 0.0f, glossVal, texGloss, sg);
    else pMaterialT = make_glass(colorT, texTranspId, samplTransp, iorVal, fogColor, fogMult, 0.0f, glossVal, texGloss, sg);
  }
  // EmissiveMaterialFromHydraMtl :953-978
  MatPtr pMaterialE;
  {
    float mult = 1.0f;
    if (xhas(xchild(emission, "multiplier"), "val")) mult = xchild(emission, "multiplier")->attr_float("val");
    Sampler s; int32_t texId = int32_t(HYDRA_INVALID_TEXTURE);
    if (sampler_node(emission)) { s = sampler_from_texref(sampler_node(emission)); texId = s.texId; }
    pMaterialE = make_emissive(colorE * mult, texId, s, a_node->attr_int("light_id"));
    if (xchild(emission, "cast_gi") && xchild(emission, "cast_gi")->attr_int("val") == 0)
      put_i(pMaterialE->plain, HM_FLAGS, get_i(pMaterialE->plain, HM_FLAGS) | HMF_FORBID_EMISSIVE_GI);
  }

  MatPtr pResult;
  const bool haveT = length(colorT) > 1e-5f, haveS = length(colorS) > 1e-5f, haveD = length(colorD) > 1e-5f;
  auto add_flags = [](const MatPtr& m, int f) { put_i(m->plain, HM_FLAGS, get_i(m->plain, HM_FLAGS) | f); };
  if (isBlendOfTwo) {
    // The reference defers these to EndMaterialUpdate and resolves node_top / node_bottom against the materials updated so far, in id order (:1787-1842); LoadSceneLibrary
    // calls this function for the blends last and in id order.  Mask: a plain texture value, or Fresnel of the blend's own IOR; the colour factor is white; the
    // extrusion is read from the MATERIAL node (ReadExtrusionType(a_node), :1481), the IOR from its <blend> child.
    const XmlNode* blend = a_node->child("blend");
    const bool fresnelBlend = std::string(xattr(blend, "type")) == "fresnel_blend";
    Sampler sm;
    int32_t maskTex = int32_t(HYDRA_INVALID_TEXTURE);
    if (const XmlNode* tx = xchild(xchild(blend, "mask"), "texture")) { sm = sampler_from_texref(tx); maskTex = sm.texId; }
    auto sub = [&](const char* attr) -> MatPtr {
      const auto it = a_node->has_attr(attr) ? m_materialTrees.find(a_node->attr_int(attr)) : m_materialTrees.end();
      if (it == m_materialTrees.end() || !it->second) {
        Unsupported("hydra_blend " + std::to_string(a_matId) + ": '" + attr + "' names no material of the library that was converted before it");
        return make_lambert(float3(1, 1, 1), int32_t(HYDRA_INVALID_TEXTURE), Sampler());
      }
      return std::static_pointer_cast<MatTree>(it->second);
    };
    pResult = make_blend(sub("node_top"), sub("node_bottom"), float3(1, 1, 1), maskTex, sm, fresnelBlend, false, read_extrusion(a_node), read_fresnel_ior(blend));
  } else if (haveT && haveS && haveD) {          // :1541-1553
    MatPtr pST = make_blend(pMaterialS, pMaterialT, colorS, texReflId, samplRefl, haveFresnelRefl, true, reflExtrusion, fresnelIOR);
    pResult = make_blend(pST, pMaterialD, colorT, texTranspId, samplTransp, false, true, reflExtrusion, fresnelIOR);
    add_flags(pST, HMF_HAS_TRANSPARENCY | HMF_CAN_SAMPLE_REFL_ONLY);
    add_flags(pResult, HMF_HAS_TRANSPARENCY);
  } else if (haveT && haveS) {            // :1554-1563
    pResult = make_blend(pMaterialS, pMaterialT, colorS, texReflId, samplRefl, haveFresnelRefl, true, reflExtrusion, fresnelIOR);
    add_flags(pResult, HMF_HAS_TRANSPARENCY | HMF_CAN_SAMPLE_REFL_ONLY);
  } else if ((haveD && haveS) || (haveS && haveFresnelRefl))
    pResult = make_blend(pMaterialS, pMaterialD, colorS, texReflId, samplRefl, haveFresnelRefl, true, reflExtrusion, fresnelIOR);
  else if (haveD && haveT) {              // :1570-1582: plain mask, strong extrusion, the glass IOR in the (unused) fresnel slot
    pResult = make_blend(pMaterialT, pMaterialD, colorT, texTranspId, samplTransp, false, true, HBF_EXTRUSION_STRONG, read_value1f(xchild(transpar, "ior")));
    add_flags(pResult, HMF_HAS_TRANSPARENCY);
  } else if (haveT) pResult = pMaterialT;
  else if (length(colorS) > 1e-5f) pResult = pMaterialS;
  else if (length(colorD) > 1e-5f) pResult = pMaterialD;
  else if (length(colorE) > 1e-5f) pResult = pMaterialE;
  else pResult = pMaterialD;

  // emission header copied into the root when the material is not emissive-only (:1716-1728)
  if (length(colorE) > 1e-5f && pResult != pMaterialE) {
    const bool visible = (a_node->attr_int("visible") == 1);
    if (!visible && a_node->has_attr("light_id")) {   // an invisible light: a clear thin glass that shadow rays pass (:1719-1724)
      pResult = make_thinglass(float3(1, 1, 1), int32_t(HYDRA_INVALID_TEXTURE), Sampler(), 1e6f, 1.0f, int32_t(HYDRA_INVALID_TEXTURE), Sampler());
      add_flags(pResult, HMF_INVIS_LIGHT);
      Opacity o;
      auto had = m_matOpacity.find(a_matId);
      if (had != m_matOpacity.end()) o = had->second;
      else { o.texId = int32_t(HYDRA_INVALID_TEXTURE); float raw[12]; put_sampler_raw(raw, 0, Sampler()); memcpy(o.sampler, raw, sizeof(raw)); o.smooth = false; }
      o.skipShadow = true;
      m_matOpacity[a_matId] = o;
    }
    float* dst = pResult->plain;
    const float* src = pMaterialE->plain;
    memcpy(dst + HM_EMISSIVE_COLOR, src + HM_EMISSIVE_COLOR, 12);
    memcpy(dst + HM_EMISSIVE_TEXID, src + HM_EMISSIVE_TEXID, 12);
    memcpy(dst + HM_EMISSIVE_SAMPLER, src + HM_EMISSIVE_SAMPLER, 48);
  }
  {   // HaveAnyNodeWithBTDF :1306-1326, :1729-1730
    std::function<bool(const MatTree*)> anyBtdf = [&](const MatTree* n) { return n && (n->isBlend ? (anyBtdf(n->c1.get()) || anyBtdf(n->c2.get())) : get_i(n->plain, HM_TYPE) == HMT_TRANSLUCENT); };
    if (anyBtdf(pResult.get())) put_i(pResult->plain, HM_FLAGS, get_i(pResult->plain, HM_FLAGS) | HMF_HAVE_BTDF);
  }
  // PopUpTransparencyAndCaustics :1284-1304 (one level)
  if (pResult->isBlend) {
    const int mask = HMF_CAST_CAUSTICS | HMF_HAS_TRANSPARENCY;
    const int f = (get_i(pResult->c1->plain, HM_FLAGS) & mask) | (get_i(pResult->c2->plain, HM_FLAGS) & mask);
    put_i(pResult->plain, HM_FLAGS, get_i(pResult->plain, HM_FLAGS) | f);
  }
  // RenderDriverRTE::ReadBumpAndOpacity, the normal-map half (PlainMaterialConverter.cpp:1340-1424, 1447-1455), <displacement type="normal_bump">
  if (const XmlNode* displ = a_node->child("displacement")) {
    const XmlNode* hm = displ->child("height_map");
    const XmlNode* hTex = hm ? hm->child("texture") : nullptr;
    if (std::string(displ->attr("type")) == "height_bump" && hTex != nullptr) {   // PlainMaterialConverter.cpp:1357-1378, BumpAmtAndLvl RenderDriverRTE_AuxTextures.cpp:11-31
      Sampler sm = sampler_from_texref(hTex);
      sm.gamma = 1.0f;
      const float bumpAmt = hm->has_attr("amount") ? 0.5f * hm->attr_float("amount") : 0.0f;
      const float smoothLvl = hm->has_attr("smooth") ? 10.0f * hm->attr_float("smooth") : (hm->has_attr("smooth_lvl") ? 10.0f * hm->attr_float("smooth_lvl") : 0.0f);
      const int32_t auxId = AuxNormalMapFromHeight(sm.texId, a_matId, bumpAmt, smoothLvl);
      int flags = HMF_INVERT_HEIGHT;
      const XmlNode* invert = hm->child("invert");
      if (invert && invert->attr_int("x") == 1) flags |= HMF_INVERT_NMAP_X;
      if (invert && invert->attr_int("y") == 1) flags |= HMF_INVERT_NMAP_Y;
      if (invert && invert->attr_int("swap_xy") == 1) flags |= HMF_INVERT_SWAP_NMAP_XY;
      put_i(pResult->plain, HM_FLAGS, get_i(pResult->plain, HM_FLAGS) | flags);
      push_down_normal_map(pResult.get(), auxId, HM_NORMAL_SAMPLER / 4, flags, sm);
    }
    const XmlNode* nm = displ->child("normal_map");
    const XmlNode* texNode = nm ? nm->child("texture") : nullptr;
    if (std::string(displ->attr("type")) == "normal_bump" && texNode != nullptr) {
      Sampler sm = sampler_from_texref(texNode);
      if (!texNode->has_attr("input_gamma")) sm.gamma = 1.0f;
      // a procedural normal map keeps its texture id in the slot (PlainMaterialConverter.cpp:1396-1399): the value comes from the path's list, not from the aux arena
      const int32_t auxId = m_procTextures.count(sm.texId) ? sm.texId : AuxNormalMapFor(sm.texId, a_matId);
      int flags = 0;
      const XmlNode* invert = nm->child("invert");
      if (invert && invert->attr_int("x") == 1) flags |= HMF_INVERT_NMAP_X;
      if (invert && invert->attr_int("y") == 1) flags |= HMF_INVERT_NMAP_Y;
      if (invert && invert->attr_int("swap_xy") == 1) flags |= HMF_INVERT_SWAP_NMAP_XY;
      put_i(pResult->plain, HM_FLAGS, get_i(pResult->plain, HM_FLAGS) | flags);
      push_down_normal_map(pResult.get(), auxId, HM_NORMAL_SAMPLER / 4, flags, sm);
    }
  }
  m_materialTrees[a_matId] = pResult;   // m_materialUpdated: what a later hydra_blend composes
  m_materialNodes[a_matId] = a_node;
  // PutAbstractMaterialToStorage :1848-1881
  PlainMaterialVec mdata = flatten(pResult);
  AppendProcTexTail(a_node, a_matId, mdata);
  m_pMaterialStorage->Update(a_matId, mdata.data(), mdata.size() * sizeof(float));
  return true;
}

// ---- procedural textures.  The scene library declares them as <texture type="proc"> with the text of their functions in a data/proctex_*.c file and one
// generated call; a material binds one with <texture type="texref_proc" id=...> and its arguments.  The reference splices the functions and the calls into
// shaders/texproc.cl and has the layer rebuild that program (RenderDriverRTE_ProcTex.cpp); the material head lists the ids and is followed by an (id, offset)
// table and the argument words the calls read as `stack[...]`.
static const char* const kProcTexTailTag = "_PROCTEXTAILTAG_";
static std::string replace_all(std::string s, const std::string& what, const std::string& with) {
  for (size_t at = s.find(what); at != std::string::npos; at = s.find(what, at + with.size())) s.replace(at, what.size(), with);
  return s;
}
bool RenderDriverLite::UpdateImageProc(int32_t a_texId, const XmlNode* a_texNode) {
  const XmlNode* code = a_texNode->child("code");
  const XmlNode* gen = xchild(code, "generated");
  if (!code || !gen || !gen->child("call")) { Unsupported("procedural texture " + std::to_string(a_texId) + " without <code><generated><call>"); return false; }
  if (const XmlNode* ao = a_texNode->child("ao"))   // ReadAOFromNode :272-302: ambient-occlusion rays feed readAttr_AO; this layer does not trace them
    if (std::string(ao->attr("hemisphere")) != "") Unsupported("procedural texture " + std::to_string(a_texId) + " asks for ambient occlusion (<ao>)");
  ProcTex pt;
  pt.retT = (std::string(xattr(gen->child("return"), "type")) == "float4") ? 4 : 1;
  pt.call = replace_all(gen->child("call")->text, kProcTexTailTag, "in_texStorage1, in_globals, hr_viewVectorHack");
  std::vector<char> d;
  if (!read_file(m_libPath + "/" + code->attr("loc"), d)) {   // the reference goes on without the functions (UpdateImageProc :603-615): fatal only once a material binds the texture
    m_log += "procedural texture " + std::to_string(a_texId) + ": code file '" + code->attr("loc") + "' is missing from the scene library\n";
    m_procTexMissing.insert(a_texId);
    return false;
  }
  pt.code = replace_all(std::string(d.begin(), d.end()), kProcTexTailTag, " __global const float4* restrict in_texStorage1, __global const EngineGlobals* restrict in_globals, const float3 hr_viewVectorHack");
  m_procTextures[a_texId] = pt;
  return true;
}
std::string RenderDriverLite::ProcTexProgramText() const {
  // the two regions of shaders/texproc.cl by their marker lines; every region ends at this layer's own end mark (include/hydra_hip.h, hydra_hip_proctex_compile)
  std::string t = "// procedural textures of " + m_libPath + "\n//#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:\n\n";
  for (const auto& pt : m_procTextures) t += pt.second.code + "\n";
  t += "//#HK_END_OF_PROCEDURAL_TEXTURES\n//#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:\n\n    int counter = 0;\n";
  for (const auto& pt : m_procTextures) {
    const std::string id = std::to_string(pt.first);
    t += "    if(materialHeadHaveTargetProcTex(pHitMaterial," + id + ") && counter < MAXPROCTEX)\n    {\n";
    t += "      __global const float* stack = fdata + findArgDataOffsetInTable(" + id + ", table);\n";
    t += "      ptl.fdata4[counter] = to_float3(" + pt.second.call + ");\n";
    t += "      ptl.id_f4 [counter] = " + id + ";\n      counter++;\n    }\n\n";
  }
  t += "    ptl.currMaxProcTex = counter;\n//#HK_END_OF_PROCEDURAL_TEXTURES_EVAL\n";
  return t;
}
void RenderDriverLite::AppendProcTexTail(const XmlNode* a_materialNode, int32_t a_matId, std::vector<float>& mdata) {
  if (m_procTextures.empty() && m_procTexMissing.empty()) return;
  // FindAllProcTextures :57-87: the bound ids in document order; ReadAllProcTexArgsFromMaterialNode :90-193: the <arg> words of every texref_proc node
  std::vector<int32_t> ids;
  std::map<int32_t, std::vector<float>> args;
  std::function<void(const XmlNode*)> scan = [&](const XmlNode* n) {
    if (n->name == "material" && std::string(n->attr("type")) == "hydra_blend")   // the blended materials' bindings count for the blend (:63-75, 99-110): top, then bottom
      for (const char* a : {"node_top", "node_bottom"}) {
        const auto sub = m_materialNodes.find(n->attr_int(a));
        if (sub != m_materialNodes.end()) scan(sub->second);
      }
    if (n->name == "texture" && n->has_attr("id") && m_procTextures.count(n->attr_int("id"))) {
      const int32_t id = n->attr_int("id");
      ids.push_back(id);
      if (std::string(n->attr("type")) == "texref_proc") {
        std::vector<float> datav;
        for (const XmlNode* arg : n->children_named("arg")) {
          const std::string type = arg->attr("type");
          const int size = arg->attr_int("size");
          std::istringstream in(arg->attr("val"));
          int comps = 0;
          if (type == "sampler2D" || type == "int") { for (int i = 0; i < size; i++) { int x = 0; in >> x; float f; memcpy(&f, &x, 4); datav.push_back(f); } }
          else if (type == "unsigned") { for (int i = 0; i < size; i++) { unsigned x = 0; in >> x; float f; memcpy(&f, &x, 4); datav.push_back(f); } }
          else if (type == "float") comps = 1;
          else if (type == "float2") comps = 2;
          else if (type == "float3") comps = 3;
          else if (type == "float4") comps = 4;
          for (int i = 0; i < comps * size; i++) { float x = 0; in >> x; datav.push_back(x); }
        }
        args[id] = datav;
      }
    }
    for (const auto& ch : n->children) scan(ch.get());
  };
  if (!m_procTexMissing.empty()) {
    std::function<void(const XmlNode*)> scanMissing = [&](const XmlNode* n) {
      if (n->name == "texture" && n->has_attr("id") && m_procTexMissing.count(n->attr_int("id")))
        Unsupported("material " + std::to_string(a_matId) + " binds procedural texture " + std::to_string(n->attr_int("id")) + " whose code file is missing from the scene library");
      for (const auto& ch : n->children) scanMissing(ch.get());
    };
    scanMissing(a_materialNode);
  }
  scan(a_materialNode);
  if (ids.empty()) return;
  if (mdata.size() < size_t(HM_NODE_FLOATS)) return;
  float* head = mdata.data();
  put_i(head, HM_FLAGS, get_i(head, HM_FLAGS) | HMF_HAVE_PROC_TEXTURES);
  // MakePTListFromTupleArray :255-277 + PutProcTexturesIdListToMaterialHead (cglobals.h:2732-2739): at most 16, the rest of the slots invalid
  if (ids.size() > 16) { Unsupported("material " + std::to_string(a_matId) + " binds more than 16 procedural textures"); ids.resize(16); }
  for (int i = 0; i < 16; i++) put_i(head, HM_PROC_TEX_IDS + i, i < int(ids.size()) ? ids[size_t(i)] : int32_t(HYDRA_INVALID_TEXTURE));
  // PutTexParamsToMaterialWithDamnTable :196-252: one node of (id, offset) pairs over ALL procedural textures of the scene (offset -1: not bound here), its last
  // word their number; then the argument words in pages of one node
  std::vector<float> table(HM_NODE_FLOATS, 0.0f), data;
  int counter = 0;
  for (const auto& pt : m_procTextures) {
    if (counter >= HM_NODE_FLOATS / 2) break;
    const auto p = args.find(pt.first);
    put_i(table.data(), counter * 2 + 0, pt.first);
    put_i(table.data(), counter * 2 + 1, p != args.end() ? int32_t(data.size()) : -1);
    if (p != args.end()) data.insert(data.end(), p->second.begin(), p->second.end());
    counter++;
  }
  put_i(table.data(), HM_NODE_FLOATS - 1, counter);
  data.resize((data.size() / HM_NODE_FLOATS + 1) * HM_NODE_FLOATS, 0.0f);
  const int32_t tableOffset = int32_t(mdata.size());   // in floats from the head (PROC_TEX_TABLE_OFFSET = oldSize * PLAIN_MATERIAL_DATA_SIZE)
  mdata.insert(mdata.end(), table.begin(), table.end());
  mdata.insert(mdata.end(), data.begin(), data.end());
  put_i(mdata.data(), HM_PROC_TEX_TABLE, tableOffset);
}

// ---- IES photometric webs.  The reference reads the file through Ian Ashdown's IESNA.C (hydra_drv/utils/ies_parser/IESNA.H, IESRender.cpp:19-36), which is not
// part of the reference tree; what follows reads the published IESNA LM-63 (1986/1991/1995) layout directly: label / keyword lines up to "TILT=", an optional
// TILT=INCLUDE block, then free-format numbers -- lamps, lumens, candela multiplier, number of vertical and horizontal angles, photometric type, units,
// width, length, height; ballast factor, ballast-lamp factor, input watts; the vertical angles; the horizontal angles; per horizontal angle the candela
// values over the vertical angles (pcandela[horz][vert] in IESNA.C, raw as in the file).  Parity of the reader is unpinned: the reference tree holds no .ies file.
struct IesData { std::vector<float> vert, horz; std::vector<std::vector<float>> candela; };
static bool read_ies_file(const std::string& path, IesData& out) {
  std::vector<char> raw;
  if (!read_file(path, raw)) return false;
  const std::string text(raw.begin(), raw.end());
  size_t at = text.find("TILT=");
  if (at == std::string::npos) return false;
  const size_t eol = text.find('\n', at);
  const std::string tilt = text.substr(at + 5, (eol == std::string::npos ? text.size() : eol) - at - 5);
  std::vector<double> num;
  {
    std::string rest = (eol == std::string::npos) ? std::string() : text.substr(eol + 1);
    for (char& ch : rest) if (ch == ',') ch = ' ';
    std::istringstream is(rest);
    double v;
    while (is >> v) num.push_back(v);
  }
  size_t k = 0;
  if (tilt.find("INCLUDE") != std::string::npos) {   // lamp-to-luminaire geometry, number of pairs, angles, multiplying factors: not used by the renderer
    if (num.size() < 2) return false;
    const size_t pairs = size_t(num[1]);
    k = 2 + 2 * pairs;
  }
  if (num.size() < k + 13) return false;
  const int nV = int(num[k + 3]), nH = int(num[k + 4]);
  k += 13;
  if (nV <= 0 || nH <= 0 || num.size() < k + size_t(nV) + size_t(nH) + size_t(nV) * size_t(nH)) return false;
  out.vert.assign(num.begin() + k, num.begin() + k + nV); k += size_t(nV);
  out.horz.assign(num.begin() + k, num.begin() + k + nH); k += size_t(nH);
  out.candela.assign(size_t(nH), std::vector<float>(size_t(nV)));
  for (int h = 0; h < nH; h++)
    for (int v = 0; v < nV; v++) out.candela[size_t(h)][size_t(v)] = float(num[k++]);
  return true;
}
// CreateSphericalTextureFromIES, hydra_drv/IESRender.cpp:29-200: the web as a lat-long image (x: phi 0..360, y: theta 0..180), mirrored into the quadrants the file leaves out
static std::vector<float> spherical_texture_from_ies(const IesData& ies, int& w, int& h) {
  const float PI = 3.14159265358979323846f, INV_PI_F = 1.0f / PI, D2R = PI / 180.0f;
  const int nV = int(ies.vert.size()), nH = int(ies.horz.size());
  const float verticalStart = ies.vert[0], verticalEnd = ies.vert[size_t(nV) - 1];
  const float horizontStart = ies.horz[0];
  float horizontEnd = ies.horz[size_t(nH) - 1];
  const float eps = 1e-5f;
  if (fabsf(verticalStart) < eps && fabsf(verticalEnd - 90.0f) < eps) h = nV * 2;
  else if (fabsf(verticalStart - 90.0f) < eps && fabsf(verticalEnd - 180.0f) < eps) h = nV * 2;
  else h = nV;
  enum { REFLECT4, REFLECT2, REFLECT0 } reflectType = REFLECT0;
  if (fabsf(horizontStart) < eps && fabsf(horizontEnd) < eps) w = 1;
  else if (fabsf(horizontStart) < eps && fabsf(horizontEnd - 90.0f) < eps) { w = nH * 4; reflectType = REFLECT4; }
  else if (fabsf(horizontStart) < eps && fabsf(horizontEnd - 180.0f) < eps) { w = nH * 4; reflectType = REFLECT4; }   // sic: treated like the quadrant case (:79-83)
  else if (fabsf(horizontStart - 90.0f) < eps && fabsf(horizontEnd - 180.0f) < eps) { w = nH * 2; reflectType = REFLECT2; }
  else w = nH;
  if (horizontEnd > 180.0f && horizontEnd < 360.0f) horizontEnd = 360.0f;
  std::vector<float> res(size_t(w) * size_t(h), 0.0f);
  const float stepTheta = (verticalEnd - verticalStart) / float(nV);
  float stepPhi = (horizontEnd - horizontStart) / float(nH);
  if (fabsf(stepPhi) < eps) stepPhi = 360.0f;
  auto row = [&](float thetaGrad) { int iY = int(((D2R * thetaGrad) * INV_PI_F) * float(h) + 0.5f); return iY >= h ? h - 1 : iY; };
  auto col = [&](float phiGrad) { int iX = int(((D2R * phiGrad) * 0.5f * INV_PI_F) * float(w) + 0.5f); return iX >= w ? w - 1 : iX; };
  int thetaIndex = 0;
  for (float thetaGrad = verticalStart; thetaIndex < nV; thetaGrad += stepTheta, thetaIndex++) {
    const int iY = row(thetaGrad);
    int phiIndex = 0;
    for (float phiGrad = horizontStart; phiIndex < nH; phiGrad += stepPhi, phiIndex++) res[size_t(iY) * w + col(phiGrad)] = ies.candela[size_t(phiIndex)][size_t(thetaIndex)];
  }
  if (reflectType == REFLECT4) {
    for (float thetaGrad = verticalStart; thetaGrad <= verticalEnd; thetaGrad += stepTheta) {
      const int iY = row(thetaGrad);
      for (float phiGrad = 0.0f; phiGrad <= 90.0f; phiGrad += stepPhi) {
        const int iX1 = col(phiGrad), iX2 = col(180.0f - phiGrad - stepPhi), iX3 = col(180.0f + phiGrad), iX4 = col(360.0f - phiGrad - stepPhi);
        res[size_t(iY) * w + iX2] = res[size_t(iY) * w + iX1];
        res[size_t(iY) * w + iX3] = res[size_t(iY) * w + iX1];
        res[size_t(iY) * w + iX4] = res[size_t(iY) * w + iX1];
      }
    }
  } else if (reflectType == REFLECT2) {
    for (float thetaGrad = verticalStart; thetaGrad <= verticalEnd; thetaGrad += stepTheta) {
      const int iY = row(thetaGrad);
      for (float phiGrad = 0.0f; phiGrad <= 180.0f; phiGrad += stepPhi) res[size_t(iY) * w + col(360.0f - phiGrad - stepPhi)] = res[size_t(iY) * w + col(phiGrad)];
    }
  }
  return res;
}
// HDRImageLite::gaussBlur(2, 1.5) on one channel (RenderDriverRTE_PdfTables.cpp:32-111, 151-176): rows, then columns, windows clipped at the border, weights re-normalised (+ 1e-5)
static void gauss_blur_1ch(std::vector<float>& lum, int w, int h) {
  float gk[5], gsum = 0.0f;
  { const float sg = 2.0f * 1.5f * 1.5f; for (int x = -2; x <= 2; x++) { const float r = sqrtf(float(x * x)); gk[x + 2] = expf(-r / sg) / (3.141592654f * sg); gsum += gk[x + 2]; } for (float& v : gk) v /= gsum; }
  std::vector<float> tmp(lum.size());
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float c = 0.0f, sw = 0.0f;
      for (int q = std::max(x - 2, 0); q <= std::min(x + 2, w - 1); q++) { c += lum[size_t(y) * w + q] * gk[q + 2 - x]; sw += gk[q + 2 - x]; }
      tmp[size_t(y) * w + x] = c / (sw + 1e-5f);
    }
  if (h == 1) { lum = tmp; return; }
  for (int x = 0; x < w; x++)
    for (int y = 0; y < h; y++) {
      float c = 0.0f, sw = 0.0f;
      for (int q = std::max(y - 2, 0); q <= std::min(y + 2, h - 1); q++) { c += tmp[size_t(q) * w + x] * gk[q + 2 - y]; sw += gk[q + 2 - y]; }
      lum[size_t(y) * w + x] = c / (sw + 1e-5f);
    }
}
// AddIesTexTableToStorage, RenderDriverRTE_PdfTables.cpp:385-478: {w, h, 1, 4} + the web scaled to a maximum of 1 (+ one spare float) as the image, {w, h, 1, 4} + prefix
// sums of the blurred web + 0.05 x mean as the sampling table; both in the pdf arena, cached per file.  Returns {-1, -1} for a file that cannot be used.
std::pair<int32_t, int32_t> RenderDriverLite::AddIesTexTable(const std::string& loc) {
  auto p = m_iesCache.find(loc);
  if (p != m_iesCache.end()) return p->second;
  IesData ies;
  if (!read_ies_file(m_libPath + "/" + loc, ies) || ies.vert.empty()) { m_log += "oldies::IE_ReadFile error: " + loc + "\n"; return {-1, -1}; }
  int w = 0, h = 0;
  std::vector<float> tex = spherical_texture_from_ies(ies, w, h);
  const int32_t iesTexId = m_pPdfStorage->GetMaxObjectId() + 1;
  float maxVal = 0.0f;
  for (float v : tex) maxVal = fmaxf(maxVal, v);
  if (tex.size() == 1 || maxVal == 0.0f) { m_log += "[ERROR]: broken IES file (maxVal = 0.0): " + loc + "\n"; return {-1, -1}; }
  const float invMax = 1.0f / maxVal;
  for (float& v : tex) v = invMax * v;
  std::vector<float> data2(tex.size() + 5, 0.0f);
  put_i(data2.data(), 0, w); put_i(data2.data(), 1, h); put_i(data2.data(), 2, 1); put_i(data2.data(), 3, 4);
  double avgVal = 0.0;
  for (size_t i = 0; i < tex.size(); i++) { avgVal += double(tex[i]); data2[i + 4] = tex[i]; }
  avgVal /= double(tex.size());
  m_pPdfStorage->Update(iesTexId, data2.data(), data2.size() * sizeof(float));
  gauss_blur_1ch(tex, w, h);
  for (float& v : tex) v = v + 0.05f * float(avgVal);          // no pixel with zero pdf
  std::vector<float> data3(4 + tex.size() + 1);
  put_i(data3.data(), 0, w); put_i(data3.data(), 1, h); put_i(data3.data(), 2, 1); put_i(data3.data(), 3, 4);
  float acc = 0.0f;
  for (size_t i = 0; i < tex.size(); i++) { data3[4 + i] = acc; acc += tex[i]; }
  data3[4 + tex.size()] = acc;
  const int32_t iesPdfId = m_pPdfStorage->GetMaxObjectId() + 1;
  m_pPdfStorage->Update(iesPdfId, data3.data(), data3.size() * sizeof(float));
  m_iesCache[loc] = {iesTexId, iesPdfId};
  return {iesTexId, iesPdfId};
}
// the IES frame of a light record: iesMatrix (optionally turned 90 degrees about Y first, ROTATE_IES_90_DEG, PlainLightConverter.cpp:14, 170-174, 645-649) in the light-matrix slot
static void put_ies_matrix(float* d, const XmlNode* iesNode, bool rotate90) {
  float4x4 iesMatrix;
  if (iesNode && iesNode->has_attr("matrix")) {
    float m[16];
    if (parse_floats(iesNode->attr("matrix"), m, 16)) iesMatrix = float4x4::from_row_major(m);
  }
  if (rotate90) {
    float4x4 mrot;                                            // LiteMath::rotate4x4Y(90 degrees): (0,2) = +sin, (2,0) = -sin (HydraAPI's LiteMath is not part of the reference tree)
    mrot.at(0, 0) = 0.0f; mrot.at(0, 2) = 1.0f; mrot.at(2, 0) = -1.0f; mrot.at(2, 2) = 0.0f;
    iesMatrix = mul(mrot, iesMatrix);
  }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) d[HL_IES_LIGHT_MATRIX + r * 3 + c] = iesMatrix.at(r, c);
}
// ILight::TransformIESMatrix, PlainLightConverter.cpp:113-124
static void transform_ies_matrix(const float* proto, const float4x4& M, float* copy) {
  float4x4 mrot = M;
  mrot.c[3][0] = 0.0f; mrot.c[3][1] = 0.0f; mrot.c[3][2] = 0.0f; mrot.c[3][3] = 1.0f;
  float4x4 sub;                                               // transpose(GetSubMatrix3x3(m_plain, IES_LIGHT_MATRIX_E00))
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) sub.at(c, r) = proto[HL_IES_LIGHT_MATRIX + r * 3 + c];
  const float4x4 ies = mul(mrot, sub), inv = inverse4x4(ies);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) { copy[HL_IES_LIGHT_MATRIX + r * 3 + c] = ies.at(c, r); copy[HL_IES_INV_MATRIX + r * 3 + c] = inv.at(c, r); }   // PutSubMatrix3x3Transp
}

// DirectLight (:500-566, CreateDirectLightFromXmlNode :840-855), SpotLight (:568-626, CreatePointSpotLightFromXmlNode :894-906)
// and PointLight without IES (:628-700); OLD_PHOTOMETRIC_SCALE is 1 (:15)
bool RenderDriverLite::UpdateDeltaLight(int32_t a_lightId, const XmlNode* a_node) {
  const std::string ltype = a_node->attr("type"), distr = a_node->attr("distribution");
  const XmlNode* inten = a_node->child("intensity");
  const float3 color = read_value3f(xchild(inten, "color")) * read_value1f(xchild(inten, "multiplier"));
  const float DEG2RAD = 3.14159265358979323846f / 180.f;
  LightProto lp;
  lp.plain.assign(HL_FLOATS, 0.0f);
  float* d = lp.plain.data();
  d[HL_PROB_MULT] = 1.0f;
  d[HL_COLOR] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
  put_i(d, HL_FLAGS, 0);
  if (ltype == "directional" || distr == "directional") {
    const XmlNode* size = a_node->child("size");
    const float r1 = size ? size->attr_float("inner_radius") : 0.0f, r2 = size ? size->attr_float("outer_radius") : 0.0f;
    const float soft = read_value1f(xchild(a_node, "shadow_softness"));
    float angle = read_value1f(xchild(a_node, "angle_radius"));
    if (!a_node->child("angle_radius")) angle = 0.25f * soft;
    d[HL_NORM + 1] = -1.0f;
    d[HL_DIRECT_RADIUS1] = r1; d[HL_DIRECT_RADIUS2] = r2;
    const float alpha = DEG2RAD * angle;
    d[HL_DIRECT_SSOFTNESS] = angle / 0.25f;
    d[HL_DIRECT_ALPHA_TAN] = tanf(alpha);
    d[HL_DIRECT_ALPHA_COS] = cosf(alpha);
    d[HL_SURFACE_AREA] = 3.14159265358979323846f * r2 * r2;
    put_i(d, HL_TYPE, HLT_DIRECT);
    lp.kind = 2;
  } else if (distr == "spot") {
    const float angle2 = read_value1f(xchild(a_node, "falloff_angle")), angle1 = read_value1f(xchild(a_node, "falloff_angle2"));
    d[HL_NORM + 1] = -1.0f;
    d[HL_POINT_SPOT_COS1] = cosf(0.5f * DEG2RAD * angle1);
    d[HL_POINT_SPOT_COS2] = cosf(0.5f * DEG2RAD * angle2);
    d[HL_SURFACE_AREA] = 1e-10f;
    put_i(d, HL_TYPE, HLT_POINT_SPOT);
    lp.kind = 1;
  } else {
    d[HL_SURFACE_AREA] = 1e-10f;
    put_i(d, HL_TYPE, HLT_POINT_OMNI);
    put_i(d, HL_IES_SPHERE_TEX_ID, int32_t(HYDRA_INVALID_TEXTURE));
    put_i(d, HL_IES_SPHERE_PDF_ID, int32_t(HYDRA_INVALID_TEXTURE));
    const XmlNode* iesNode = a_node->child("ies");             // PointLight with a photometric web, PlainLightConverter.cpp:640-697
    if (iesNode && iesNode->has_attr("loc") && distr == "ies") {
      const auto ids = AddIesTexTable(iesNode->attr("loc"));
      if (ids.first >= 0 && ids.second >= 0) {
        put_i(d, HL_IES_SPHERE_TEX_ID, ids.first);
        put_i(d, HL_IES_SPHERE_PDF_ID, ids.second);
        put_i(d, HL_FLAGS, HLF_HAS_IES);
        // the frame of the web: <ies matrix> turned 90 degrees about Y, else a <honio matrix> as it is, else identity (:640-656)
        const XmlNode* honio = a_node->child("honio");
        if (iesNode->has_attr("matrix")) put_ies_matrix(d, iesNode, true);
        else if (honio && honio->has_attr("matrix")) put_ies_matrix(d, honio, false);
        else put_ies_matrix(d, nullptr, false);
        lp.hasIes = true;
      }
    }
    lp.kind = 0;
  }
  lp.isDelta = true;
  m_lights[a_lightId] = lp;
  return true;
}

// The luminance image a light's 2-D sampling table is made of (RenderDriverRTE::UpdatePdfTablesForLight, RenderDriverRTE_PdfTables.cpp:512-533): sky domes
// and cylinder lights share it
void RenderDriverLite::LuminanceImageOf(int32_t texId, std::vector<float>& lum, int& lw, int& lh) {
  const std::vector<int32_t> table = m_pTexStorage->GetTable();
  if (texId < 0 || texId >= int32_t(table.size()) || table[texId] < 0) RunTimeError("UpdateLight: light texture " + std::to_string(texId) + " is not loaded");
  const int32_t* hdr = reinterpret_cast<const int32_t*>(static_cast<const char*>(m_pTexStorage->GetBegin()) + size_t(table[texId]) * 16);
  int w = hdr[0], h = hdr[1];
  const int bpp = hdr[3];
  if (bpp == 16) {
    std::vector<float> px(reinterpret_cast<const float*>(hdr + 4), reinterpret_cast<const float*>(hdr + 4) + size_t(w) * h * 4);
    for (int pass = 0; pass < 4; pass++) {                       // resizeToHalfSizef4, :186-215
      if (w <= 2048 && h <= 2048) continue;                      // MAX_ENV_LIGHT_PDF_SIZE, RenderDriverRTE.h:23
      const int nw = std::max(w / 2, 1), nh = std::max(h / 2, 1);
      std::vector<float> half(size_t(nw) * nh * 4);
      for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++)
          for (int ch = 0; ch < 4; ch++) {
            const int o1 = (2 * y + 0) * 2 * nw, o2 = (2 * y + 1) * 2 * nw;
            half[(size_t(y) * nw + x) * 4 + ch] = 0.25f * (((px[size_t(o1 + 2 * x) * 4 + ch] + px[size_t(o1 + 2 * x + 1) * 4 + ch]) + px[size_t(o2 + 2 * x) * 4 + ch]) + px[size_t(o2 + 2 * x + 1) * 4 + ch]);
          }
      px.swap(half); w = nw; h = nh;
    }
    lum.assign(size_t(w) * h, 0.0f);
    float avg = 0.0f;
    for (size_t i = 0; i < lum.size(); i++) { lum[i] = std::max(px[i * 4], std::max(px[i * 4 + 1], px[i * 4 + 2])); avg += lum[i]; }
    avg /= float(lum.size());
    avg = std::max(avg, 1.0f);
    gauss_blur_1ch(lum, w, h);                                    // HDRImageLite::gaussBlur(2, 1.5), one channel
    for (float& v : lum) v += 0.05f * avg;                        // no pixel with zero pdf
    lw = w; lh = h;
  }
  else if (bpp != 4) Unsupported("light texture of " + std::to_string(bpp) + " bytes per texel");
  else {
    std::vector<uint8_t> px(reinterpret_cast<const uint8_t*>(hdr + 4), reinterpret_cast<const uint8_t*>(hdr + 4) + size_t(w) * h * 4);
    for (int pass = 0; pass < 4; pass++) {                       // resizeToHalfSizeUB4, :268-310
      if (w <= 256 && h <= 256) continue;
      const int nw = std::max(w / 2, 1), nh = std::max(h / 2, 1);
      std::vector<uint8_t> half(size_t(nw) * nh * 4);
      for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++)
          for (int ch = 0; ch < 4; ch++) {
            const int o1 = (2 * y + 0) * 2 * nw, o2 = (2 * y + 1) * 2 * nw;
            const int sum = px[size_t(o1 + 2 * x) * 4 + ch] + px[size_t(o1 + 2 * x + 1) * 4 + ch] + px[size_t(o2 + 2 * x) * 4 + ch] + px[size_t(o2 + 2 * x + 1) * 4 + ch];
            half[(size_t(y) * nw + x) * 4 + ch] = uint8_t(std::min(sum >> 2, 255));
          }
      px.swap(half); w = nw; h = nh;
    }
    lum.assign(size_t(w) * h, 0.0f);
    float avg = 0.0f;
    for (size_t i = 0; i < lum.size(); i++) {
      const float r = px[i * 4] * (1.0f / 255.0f), g = px[i * 4 + 1] * (1.0f / 255.0f), b = px[i * 4 + 2] * (1.0f / 255.0f);
      lum[i] = std::max(r, std::max(g, b));
      avg += lum[i];
    }
    avg /= float(lum.size());
    avg = std::max(avg, 1.0f);
    for (float& v : lum) v += 0.1f * avg;                         // no pixel with zero pdf
    lw = w; lh = h;
  }
}
// header {w, h, 1, n + 1} + prefix sums of the luminance image + a trailing 1.0 (:546-566); returns the id the table got
int32_t RenderDriverLite::PutPdfTable2D(const std::vector<float>& lum, int lw, int lh) {
  const int32_t tabId = m_pPdfStorage->GetMaxObjectId() + 1;
  const size_t n = lum.size() + 1;                                // PrefixSumm: n + 1 entries
  std::vector<float> data(4 + n + 1);
  put_i(data.data(), 0, lw); put_i(data.data(), 1, lh); put_i(data.data(), 2, 1); put_i(data.data(), 3, int32_t(n + 1));
  float acc = 0.0f;
  for (size_t i = 0; i < lum.size(); i++) { data[4 + i] = acc; acc += lum[i]; }
  data[4 + lum.size()] = acc;
  data[4 + n] = 1.0f;
  m_pPdfStorage->Update(tabId, data.data(), data.size() * sizeof(float));
  return tabId;
}

// SkyDomeLight, hydra_drv/PlainLightConverter.cpp:909-1051 + RenderDriverRTE::UpdatePdfTablesForLight
// (RenderDriverRTE_PdfTables.cpp:479-570).  A sky without texture gets the reference's 2x2 uniform luminance image as its
// sampling table; an 8-bit lat-long texture gets the table of LuminanceFromUchar4Image (:312-356: halve until <= 256,
// max(r,g,b)/255, + 0.1 * max(mean, 1)); a float (.image4f, e.g. an .hdr environment) texture that of LuminanceFromFloat4Image (:227-266:
// halve until <= 2048, max(r,g,b), HDRImageLite::gaussBlur(2, 1.5) (:32-111, 151-176), + 0.05 * max(mean, 1)).  The texture's sampler matrix
// goes into the sampler's rows and, inverted, into the light record (sampling maps table cells back through it).  <perez turbidity sun_id> switches the colour seen by
// camera and bounce rays to the Perez model (:920-923, 1019-1028); the sun it names is copied in at EndScene.
bool RenderDriverLite::UpdateSkyLight(int32_t a_lightId, const XmlNode* a_node) {
  const XmlNode* perez = a_node->child("perez");
  const XmlNode* inten = a_node->child("intensity");
  const XmlNode* colorNode = xchild(inten, "color");
  const XmlNode* texNode = colorNode ? colorNode->child("texture") : nullptr;
  int32_t skyTexId = int32_t(HYDRA_INVALID_TEXTURE);
  float skyGamma = 1.0f;
  float4x4 samplerMat;          // identity; row-major as HydraXMLHelpers::ReadMatrix4x4 hands it over
  if (texNode) {
    if (texNode->has_attr("matrix")) {
      float m[16];
      if (!parse_floats(texNode->attr("matrix"), m, 16)) RunTimeError("UpdateLight: sky light texture matrix needs 16 numbers");
      samplerMat = float4x4::from_row_major(m);
    }
    if (texNode->has_attr("input_gamma")) skyGamma = texNode->attr_float("input_gamma");
    if (texNode->has_attr("id")) skyTexId = texNode->attr_int("id");
  }
  float3 color = read_value3f(colorNode);
  color = color * read_value1f(xchild(inten, "multiplier"));   // HydraXMLHelpers::ReadLightIntensity

  LightProto lp;
  lp.plain.assign(HL_FLOATS, 0.0f);
  float* d = lp.plain.data();
  d[HL_PROB_MULT] = 1.0f;
  d[HL_COLOR] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
  float* sam0 = d + HL_SKY_SAMPLER0;                       // SWTexSampler: flags, gamma, texId, dummy, row0, row1
  put_i(sam0, HS_FLAGS, 0); sam0[HS_GAMMA] = skyGamma; put_i(sam0, HS_TEXID, skyTexId); put_i(sam0, HS_DUMMY, 0);
  for (int k = 0; k < 4; k++) { sam0[HS_ROW0 + k] = samplerMat.at(0, k); sam0[HS_ROW1 + k] = samplerMat.at(1, k); }   // :986-989
  memcpy(d + HL_SKY_SAMPLER1, sam0, 12 * sizeof(float));
  put_i(d, HL_COLOR_TEX, skyTexId);                                                 // used to build the pdf table
  put_i(d, HL_COLOR_TEX_MATRIX, texNode ? 0 : int32_t(HYDRA_INVALID_TEXTURE));     // 0 = "have sampler and texture" (:976-977)
  {
    const float4x4 inv = inverse4x4(samplerMat);           // the reference writes both inverses to slot 0 (:947-948, 991-992)
    memcpy(d + HL_SKY_INV_MATRIX0, inv.data(), 64);
  }
  d[HL_SKY_COLOR_AUX] = color.x; d[HL_SKY_COLOR_AUX + 1] = color.y; d[HL_SKY_COLOR_AUX + 2] = color.z;
  put_i(d, HL_SKY_COLOR_TEX_AUX, skyTexId);
  put_i(d, HL_SKY_COLOR_TEX_MATRIX_AUX, texNode ? 0 : int32_t(HYDRA_INVALID_TEXTURE));
  put_i(d, HL_SKY_AUX_TEX_MATRIX_INV, int32_t(HYDRA_INVALID_TEXTURE));
  d[HL_SKY_SUN_DIR] = 0.0f; d[HL_SKY_SUN_DIR + 1] = -1.0f; d[HL_SKY_SUN_DIR + 2] = 0.0f;
  d[HL_SKY_TURBIDITY] = perez ? perez->attr_float("turbidity") : 0.0f;
  d[HL_SKY_SUN_COLOR] = 1.0f; d[HL_SKY_SUN_COLOR + 1] = 1.0f; d[HL_SKY_SUN_COLOR + 2] = 1.0f;
  put_i(d, HL_SKY_SUN_DIR_ID, (perez && perez->has_attr("sun_id")) ? perez->attr_int("sun_id") : -1);
  put_i(d, HL_TYPE, HLT_SKY_DOME);
  put_i(d, HL_FLAGS, perez ? HLF_SKY_USE_PEREZ : 0);

  // luminance image the directions are importance-sampled from
  int lw = 2, lh = 2;
  std::vector<float> lum(4, 0.25f);
  if (skyTexId != int32_t(HYDRA_INVALID_TEXTURE)) LuminanceImageOf(skyTexId, lum, lw, lh);
  for (int t = 0; t < 2; t++) put_i(d, HL_SKY_PDF_TABLE0 + t, PutPdfTable2D(lum, lw, lh));   // pdf tables 0 and 1
  lp.isDisk = false;
  lp.isSky = true;
  m_lights[a_lightId] = lp;
  SkyBack sb;   // RenderDriverRTE::UpdateLight :946-967
  if (const XmlNode* back = a_node->child("back")) {
    sb.texId = xchild(back, "texture") ? xchild(back, "texture")->attr_int("id") : 0;
    if (xhas(xchild(back, "texture"), "input_gamma")) sb.gamma = xchild(back, "texture")->attr_float("input_gamma");
    if (back->has_attr("multcolor")) { float v[3] = {1, 1, 1}; parse_floats(back->attr("multcolor"), v, 3); sb.color = float3(v[0], v[1], v[2]); }
    sb.mode = (std::string(back->attr("mode")) == "spherical") ? 1 : 0;
  }
  m_skyBack[a_lightId] = sb;
  return true;
}

// AreaDiffuseLight, hydra_drv/PlainLightConverter.cpp:130-253
bool RenderDriverLite::UpdateLight(int32_t a_lightId, const XmlNode* a_node) {
  const std::string ltype = a_node->attr("type"), lshape = a_node->attr("shape"), distr = a_node->attr("distribution");
  if (ltype == "sky") return UpdateSkyLight(a_lightId, a_node);
  if (ltype == "directional" || distr == "directional" || lshape == "point") return UpdateDeltaLight(a_lightId, a_node);   // factory order of PlainLightConverter.cpp:1070-1105
  if (ltype == "area" && lshape == "mesh") {   // MeshLight, PlainLightConverter.cpp:724-835; RenderDriverRTE::UpdateLight :925-938 hands it the light's mesh
    const int32_t meshId = a_node->attr_int("mesh_id", -1);
    const std::vector<int32_t> gtable = m_pGeomStorage->GetTable();
    if (meshId < 0 || size_t(meshId) >= gtable.size() || gtable[size_t(meshId)] < 0) { Unsupported("mesh light " + std::to_string(a_lightId) + ": its mesh " + std::to_string(meshId) + " is not in the geometry storage yet"); return false; }
    const char* blob = reinterpret_cast<const char*>(m_pGeomStorage->GetBegin()) + size_t(gtable[size_t(meshId)]) * 16;
    const HydraPlainMesh* pm = reinterpret_cast<const HydraPlainMesh*>(blob);
    const float* vpos = reinterpret_cast<const float*>(blob + size_t(pm->vPosOffset) * 16);
    const int32_t* indices = reinterpret_cast<const int32_t*>(blob + size_t(pm->vIndicesOffset) * 16);
    LightProto mp;
    mp.plain.assign(HL_FLOATS, 0.0f);
    float* d = mp.plain.data();
    d[HL_PROB_MULT] = 1.0f;
    const XmlNode* inten = a_node->child("intensity");
    const float3 color = read_value3f(xchild(inten, "color")) * read_value1f(xchild(inten, "multiplier"));
    d[HL_COLOR + 0] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
    // CalcTrianglePickProbTable + PrefixSumm, RenderDriverRTE_PdfTables.cpp:650-673, :359-370
    const int triNum = pm->tIndicesNum / 3;
    std::vector<float> table(size_t(triNum) + 1);
    double surfaceAreaTotal = 0.0;
    float accum = 0.0f;
    for (int t = 0; t < triNum; t++) {
      const int iA = indices[3 * t], iB = indices[3 * t + 1], iC = indices[3 * t + 2];
      const float3 A(vpos[iA * 4], vpos[iA * 4 + 1], vpos[iA * 4 + 2]), B(vpos[iB * 4], vpos[iB * 4 + 1], vpos[iB * 4 + 2]), C(vpos[iC * 4], vpos[iC * 4 + 1], vpos[iC * 4 + 2]);
      const float triSA = 0.5f * length(cross(B - A, C - A));
      table[size_t(t)] = accum;
      accum += triSA;
      surfaceAreaTotal += double(triSA);
    }
    table[size_t(triNum)] = accum;
    const int32_t meshVerId = m_pPdfStorage->GetMaxObjectId() + 1, meshPdfId = m_pPdfStorage->GetMaxObjectId() + 2;
    m_pPdfStorage->Update(meshVerId, blob, pm->totalBytesNum);
    m_pPdfStorage->Update(meshPdfId, table.data(), table.size() * sizeof(float));
    put_i(d, HL_MESH_MESH_ID, meshVerId);
    put_i(d, HL_MESH_TABLE_ID, meshPdfId);
    put_i(d, HL_MESH_TRI_NUM, triNum);
    d[HL_SURFACE_AREA] = float(surfaceAreaTotal);
    put_i(d, HL_TYPE, HLT_MESH);
    put_i(d, HL_FLAGS, 0);
    put_i(d, HL_IES_SPHERE_TEX_ID, int32_t(HYDRA_INVALID_TEXTURE));
    put_i(d, HL_IES_SPHERE_PDF_ID, int32_t(HYDRA_INVALID_TEXTURE));
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(d + HL_MESH_MATRIX, ident, 36);
    {   // the colour texture, CreateMeshLightFromXmlNode (PlainLightConverter.cpp:837-...): looked up at the mesh's own texture coordinates
      Sampler sc; int32_t texIdColor = int32_t(HYDRA_INVALID_TEXTURE);
      if (const XmlNode* tx = sampler_node(xchild(inten, "color"))) { sc = sampler_from_texref(tx); texIdColor = sc.texId; }
      put_sampler_at(d, texIdColor, sc, HL_MESH_TEX_ID, HL_MESH_TEXMATRIX_ID, HL_MESH_TEX_SAMPLER);
    }
    mp.isMesh = true;
    mp.meshPos.assign(vpos, vpos + size_t(pm->vPosNum) * 4);
    mp.meshInd.assign(indices, indices + pm->tIndicesNum);
    m_lights[a_lightId] = mp;
    return true;
  }
  if (ltype == "area" && lshape == "cylinder") {   // CylinderLight, PlainLightConverter.cpp:354-443, CreateCylinderLightFromXmlNode :867-893
    LightProto cp;
    cp.plain.assign(HL_FLOATS, 0.0f);
    float* d = cp.plain.data();
    d[HL_PROB_MULT] = 1.0f;
    const XmlNode* size = a_node->child("size");
    const float radius = size ? size->attr_float("radius") : 0.0f, height = size ? size->attr_float("height") : 0.0f, angle = size ? size->attr_float("angle") : 0.0f;
    const float zMin = -0.5f * height, zMax = +0.5f * height, phiMax = (3.14159265358979323846f / 180.f) * angle;
    const XmlNode* inten = a_node->child("intensity");
    const float3 color = read_value3f(xchild(inten, "color")) * read_value1f(xchild(inten, "multiplier"));
    d[HL_COLOR + 0] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
    d[HL_CYL_RADIUS] = radius; d[HL_CYL_ZMIN] = zMin; d[HL_CYL_ZMAX] = zMax; d[HL_CYL_PHIMAX] = phiMax;
    d[HL_SURFACE_AREA] = (zMax - zMin) * radius * phiMax;
    Sampler sc; int32_t texIdColor = int32_t(HYDRA_INVALID_TEXTURE);
    if (const XmlNode* tx = sampler_node(xchild(inten, "color"))) { sc = sampler_from_texref(tx); texIdColor = sc.texId; }
    sc.row0[0] = 1; sc.row0[1] = 0; sc.row0[2] = 0; sc.row0[3] = 0;   // no texture matrices on cylinder lights (:888-889)
    sc.row1[0] = 0; sc.row1[1] = 1; sc.row1[2] = 0; sc.row1[3] = 0;
    put_sampler_at(d, texIdColor, sc, HL_CYL_TEX_ID, HL_CYL_TEXMATRIX_ID, HL_CYL_TEX_SAMPLER);
    put_i(d, HL_TYPE, HLT_CYLINDER);
    put_i(d, HL_FLAGS, 0);
    // the (z, phi) sampling table, RenderDriverRTE::UpdateLight :940-941 -> UpdatePdfTablesForLight: the luminance image of the colour texture, 2 x 2 uniform without one
    int lw = 2, lh = 2;
    std::vector<float> lum(4, 0.25f);
    if (texIdColor != int32_t(HYDRA_INVALID_TEXTURE)) LuminanceImageOf(texIdColor, lum, lw, lh);
    put_i(d, HL_CYL_PDF_TABLE_ID, PutPdfTable2D(lum, lw, lh));
    cp.isCylinder = true;
    m_lights[a_lightId] = cp;
    return true;
  }
  if (ltype == "area" && lshape == "sphere") {   // SphereLight, PlainLightConverter.cpp:445-496, CreateSphereLightFromXmlNode :858-865
    LightProto sp;
    sp.plain.assign(HL_FLOATS, 0.0f);
    float* d = sp.plain.data();
    d[HL_PROB_MULT] = 1.0f;
    const XmlNode* size = a_node->child("size");
    const float radius = size ? size->attr_float("radius") : 0.0f;
    const XmlNode* inten = a_node->child("intensity");
    const float3 color = read_value3f(xchild(inten, "color")) * read_value1f(xchild(inten, "multiplier"));
    d[HL_COLOR + 0] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
    d[HL_SPHERE_RADIUS] = radius;
    d[HL_SURFACE_AREA] = 4.0f * 3.1415926535f * radius * radius;
    put_i(d, HL_TYPE, HLT_SPHERE);
    put_i(d, HL_FLAGS, 0);
    sp.isSphere = true;
    m_lights[a_lightId] = sp;
    return true;
  }
  if (ltype != "area" || (lshape != "rect" && lshape != "disk")) { Unsupported("light type '" + ltype + "/" + lshape + "'"); }
  const bool isSkyPortal = xchild(a_node, "sky_portal") && xchild(a_node, "sky_portal")->attr_int("val") == 1;   // :197-204; OLD_PHOTOMETRIC_SCALE is 1

  LightProto lp;
  lp.plain.assign(HL_FLOATS, 0.0f);
  float* d = lp.plain.data();
  d[HL_PROB_MULT] = 1.0f;                                   // ILight::ILight, AbstractMaterial.h:130-134
  const bool isSpot = (distr == "spot"), isDisk = (lshape == "disk");
  const XmlNode* size = a_node->child("size");
  float sx = size ? size->attr_float("half_length") : 0.0f;  // ReadRectLightSize (HydraAPI): x = half_length, y = half_width,
  float sy = size ? size->attr_float("half_width") : 0.0f;   // pinned by the light meshes of test_42/test_224 (x extent = half_length)
  if (isDisk) sx = size ? size->attr_float("radius") : 0.0f;
  const XmlNode* inten = a_node->child("intensity");
  float3 color = read_value3f(xchild(inten, "color"));
  const float mult = read_value1f(xchild(inten, "multiplier"));
  color = color * mult;
  float area = 4.0f * sx * sy;
  if (isDisk) area = 3.14159265358979323846f * sx * sx;
  const float angle1 = read_value1f(xchild(a_node, "falloff_angle")), angle2 = read_value1f(xchild(a_node, "falloff_angle2"));
  const float DEG2RAD = 3.14159265358979323846f / 180.f;

  d[HL_POS + 0] = 0; d[HL_POS + 1] = 0; d[HL_POS + 2] = 0;
  d[HL_NORM + 0] = 0; d[HL_NORM + 1] = -1; d[HL_NORM + 2] = 0;
  d[HL_COLOR + 0] = color.x; d[HL_COLOR + 1] = color.y; d[HL_COLOR + 2] = color.z;
  put_i(d, HL_COLOR_TEX, int32_t(HYDRA_INVALID_TEXTURE));
  put_i(d, HL_COLOR_TEX_MATRIX, int32_t(HYDRA_INVALID_TEXTURE));
  d[HL_SURFACE_AREA] = area;
  d[HL_AREA_SIZE_X] = sx; d[HL_AREA_SIZE_Y] = sy;
  const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(d + HL_AREA_MATRIX, ident, 36);
  d[HL_AREA_SPOT_COS1] = cosf(0.5f * DEG2RAD * angle2);
  d[HL_AREA_SPOT_COS2] = cosf(0.5f * DEG2RAD * angle1);
  put_i(d, HL_AREA_IS_DISK, int(isDisk));
  put_i(d, HL_AREA_SPOT_DISTR, int(isSpot));
  put_i(d, HL_AREA_SKY_SOURCE, isSkyPortal ? xchild(a_node, "sky_portal")->attr_int("source_id") : 0);
  put_i(d, HL_TYPE, HLT_AREA);
  put_i(d, HL_FLAGS, isSkyPortal ? HLF_SKY_PORTAL : 0);
  if (isSkyPortal) { put_i(d, HL_AREA_SKYPORTAL_BTEX, int32_t(HYDRA_INVALID_TEXTURE)); put_i(d, HL_AREA_SKYPORTAL_BTEX_MATRIX, int32_t(HYDRA_INVALID_TEXTURE)); }
  put_i(d, HL_IES_SPHERE_TEX_ID, int32_t(HYDRA_INVALID_TEXTURE));
  put_i(d, HL_IES_SPHERE_PDF_ID, int32_t(HYDRA_INVALID_TEXTURE));
  // IES matrix: the <ies matrix> (identity without one) turned 90 degrees about Y (ROTATE_IES_90_DEG, PlainLightConverter.cpp:14, 162-174); read only with a web
  const XmlNode* iesNode = a_node->child("ies");
  put_ies_matrix(d, (iesNode && iesNode->has_attr("matrix")) ? iesNode : nullptr, true);
  if (iesNode && iesNode->has_attr("loc") && distr == "ies") {   // :176-187, 259-266
    const auto ids = AddIesTexTable(iesNode->attr("loc"));
    if (ids.first >= 0 && ids.second >= 0) {
      put_i(d, HL_FLAGS, get_i(d, HL_FLAGS) | HLF_HAS_IES | ((iesNode->attr_int("point_area") == 1) ? HLF_IES_POINT_AREA : 0));
      put_i(d, HL_IES_SPHERE_TEX_ID, ids.first);
      put_i(d, HL_IES_SPHERE_PDF_ID, ids.second);
    }
  }
  lp.hasIes = true;                                           // AreaDiffuseLight::Transform turns the IES frame with or without a web (:304)
  lp.isDisk = isDisk;
  m_lights[a_lightId] = lp;
  return true;
}

// CalcAuxShadowRaysOffsets + UpdateMesh, RenderDriverRTE.cpp:990-1159
bool RenderDriverLite::UpdateMesh(int32_t a_meshId, int vertNum, int triNum, const float* pos4f, const float* norm4f, const float* tan4f,
                                  const float* texcoord2f, const int* indices, const int* triMatIndices) {
  auto rb = [](size_t b) { return ((b + 15) / 16) * 16; };
  const size_t headerSize = rb(sizeof(HydraPlainMesh));
  const size_t vertPosOffset = headerSize, vertPosSize = rb(16 * size_t(vertNum));
  const size_t vertNormOffset = vertPosOffset + vertPosSize, vertNormSize = vertPosSize;
  const size_t vertTexcOffset = vertNormOffset + vertNormSize, vertTexcSize = 0;
  const size_t vertTangOffset = vertTexcOffset + vertTexcSize, vertTangSize = vertPosSize;
  const size_t triIndOffset = vertTangOffset + vertTangSize, triIndSize = rb(size_t(triNum) * 3 * 4);
  const size_t triMIndOffset = triIndOffset + triIndSize, triMIndSize = rb(size_t(triNum) * 4);
  const size_t triSOffOffset = triMIndOffset + triMIndSize, triSOffSize = rb(size_t(triNum) * 4);
  const size_t totalByteSize = triSOffOffset + triSOffSize;

  std::vector<float> posAndTx(pos4f, pos4f + size_t(vertNum) * 4), normAndTy(norm4f, norm4f + size_t(vertNum) * 4);
  for (int i = 0; i < vertNum; i++) { posAndTx[i * 4 + 3] = texcoord2f[2 * i]; normAndTy[i * 4 + 3] = texcoord2f[2 * i + 1]; }

  // per-polygon auxiliary shadow offsets
  float3 bmn(1e30f, 1e30f, 1e30f), bmx(-1e30f, -1e30f, -1e30f);
  auto P = [&](int i) { return float3(pos4f[i * 4], pos4f[i * 4 + 1], pos4f[i * 4 + 2]); };
  auto N = [&](int i) { return float3(norm4f[i * 4], norm4f[i * 4 + 1], norm4f[i * 4 + 2]); };
  for (int t = 0; t < triNum; t++)
    for (int k = 0; k < 3; k++) { const float3 v = P(indices[t * 3 + k]); bmn = vmin(bmn, v); bmx = vmax(bmx, v); }
  const float3 ext = bmx - bmn;
  const float meshMaxShadowOffset = 0.00025f * fmaxf(ext.x, fmaxf(ext.y, ext.z));
  auto normalDiff = [](float3 n1, float3 n2) { if (dot(n1, n2) < 0) n2 = n2 * (-1.0f); return length(n1 - n2); };
  std::vector<float> shadowOffsets(triNum);
  for (int t = 0; t < triNum; t++) {
    const int iA = indices[t * 3], iB = indices[t * 3 + 1], iC = indices[t * 3 + 2];
    const float3 A = P(iA), B = P(iB), C = P(iC);
    const float3 crpd = cross(A - B, A - C);
    const float3 fN = normalize(crpd);
    const float nd = normalDiff(fN, N(iA)) + normalDiff(fN, N(iB)) + normalDiff(fN, N(iC));
    shadowOffsets[t] = (nd > 0.001f) ? fminf(0.05f * sqrtf(length(crpd * 0.5f)), meshMaxShadowOffset) : 0.0f;
  }

  if (m_pGeomStorage->Update(a_meshId, nullptr, totalByteSize) == -1) return false;
  HydraPlainMesh header;
  memset(&header, 0, sizeof(header));
  header.vPosOffset = int(vertPosOffset / 16); header.vNormOffset = int(vertNormOffset / 16);
  header.vTexCoordOffset = int(vertTexcOffset / 16); header.vTangentOffset = int(vertTangOffset / 16);
  header.vIndicesOffset = int(triIndOffset / 16); header.mIndicesOffset = int(triMIndOffset / 16);
  header.polyShadowOffset = int(triSOffOffset / 16);
  header.vPosNum = header.vNormNum = header.vTexCoordNum = header.vTangentNum = vertNum;
  header.tIndicesNum = triNum * 3; header.mIndicesNum = triNum;
  header.totalBytesNum = uint32_t(totalByteSize);
  m_pGeomStorage->UpdatePartial(a_meshId, &header, 0, sizeof(header));
  m_pGeomStorage->UpdatePartial(a_meshId, posAndTx.data(), vertPosOffset, size_t(vertNum) * 16);
  m_pGeomStorage->UpdatePartial(a_meshId, normAndTy.data(), vertNormOffset, size_t(vertNum) * 16);
  m_pGeomStorage->UpdatePartial(a_meshId, tan4f, vertTangOffset, size_t(vertNum) * 16);
  m_pGeomStorage->UpdatePartial(a_meshId, indices, triIndOffset, size_t(triNum) * 12);
  m_pGeomStorage->UpdatePartial(a_meshId, triMatIndices, triMIndOffset, size_t(triNum) * 4);
  m_pGeomStorage->UpdatePartial(a_meshId, shadowOffsets.data(), triSOffOffset, size_t(triNum) * 4);
  return true;
}

bool RenderDriverLite::UpdateCamera(const XmlNode* cam) {
  if (cam == nullptr) return true;
  m_camera.useMatrices = false;
  if (std::string(cam->attr("type")) == "two_matrices") {   // RenderDriverRTE.cpp:1178-1201: both matrices come as text, 16 numbers in row-major order
    float mWorldView[16], mProj[16];
    if (!parse_floats(xtext(cam->child("mWorldView")), mWorldView, 16) || !parse_floats(xtext(cam->child("mProj")), mProj, 16))
      RunTimeError("UpdateCamera: a two_matrices camera needs <mWorldView> and <mProj> with 16 numbers each");
    m_camera.mProj = float4x4::from_row_major(mProj);
    m_camera.mWorldView = float4x4::from_row_major(mWorldView);
    m_camera.useMatrices = true;
    m_camera.fov = 2.0f * atanf(1.0f / m_camera.mProj.at(1, 1)) * (180.f / 3.14159265358979323846f);   // the field of view back from the projection
  }
  float v[3];
  if (!m_camera.useMatrices) {
    if (cam->child("fov") && !cam->child("fov")->text.empty()) m_camera.fov = strtof(cam->child("fov")->text.c_str(), nullptr);
    if (cam->child("nearClipPlane")) m_camera.nearPlane = strtof(cam->child("nearClipPlane")->text.c_str(), nullptr);
    if (cam->child("farClipPlane")) m_camera.farPlane = strtof(cam->child("farClipPlane")->text.c_str(), nullptr);
    if (cam->child("position") && parse_floats(cam->child("position")->text, v, 3)) m_camera.pos = float3(v[0], v[1], v[2]);
    if (cam->child("look_at") && parse_floats(cam->child("look_at")->text, v, 3)) m_camera.lookAt = float3(v[0], v[1], v[2]);
    if (cam->child("up") && parse_floats(cam->child("up")->text, v, 3)) m_camera.up = float3(v[0], v[1], v[2]);
  }

  auto vars = m_pHWLayer->GetAllFlagsAndVars();
  vars.m_varsF[HV_F_CAM_FOV] = (3.14159265358979323846f / 180.f) * m_camera.fov;
  if (cam->child("dof_lens_radius") && !cam->child("dof_lens_radius")->text.empty())
    vars.m_varsF[HV_F_DOF_LENS_RADIUS] = strtof(cam->child("dof_lens_radius")->text.c_str(), nullptr);
  vars.m_varsF[HV_F_DOF_FOCAL_PLANE_DIST] = length(m_camera.pos - m_camera.lookAt);
  int hasDof = -1;
  if (cam->child("enable_dof") && !cam->child("enable_dof")->text.empty()) hasDof = atoi(cam->child("enable_dof")->text.c_str());
  if (m_forceDof >= 0) hasDof = m_forceDof;
  if (hasDof >= 0) {
    if (hasDof > 0) vars.m_varsI[HV_I_ENABLE_DOF] = hasDof;
    else { vars.m_varsI[HV_I_ENABLE_DOF] = 0; vars.m_varsF[HV_F_DOF_LENS_RADIUS] = 0.0f; }
  }
  if (cam->child("tiltRotX") || cam->child("tiltRotY")) {  // swapped on purpose, RenderDriverRTE.cpp:1254-1258
    vars.m_varsF[HV_F_TILT_ROT_Y] = strtof(xtext(cam->child("tiltRotX")).c_str(), nullptr);
    vars.m_varsF[HV_F_TILT_ROT_X] = strtof(xtext(cam->child("tiltRotY")).c_str(), nullptr);
  } else { vars.m_varsF[HV_F_TILT_ROT_Y] = 0.0f; vars.m_varsF[HV_F_TILT_ROT_X] = 0.0f; }
  vars.m_varsI[21 /*HRT_SAMPLES_PER_PASS*/] = cam->has_attr("integrator_iters") ? cam->attr_int("integrator_iters") : 1;
  m_pHWLayer->SetAllFlagsAndVars(vars);
  return true;
}

bool RenderDriverLite::UpdateSettings(const XmlNode* st) {
  auto vars = m_pHWLayer->GetAllFlagsAndVars();
  vars.m_varsI[47 /*HRT_FBUF_CHANNELS*/] = 4;
  vars.m_flags |= (HF_USE_MIS | HF_COMPUTE_SHADOWS);
  const std::string mc = xtext(xchild(st, "method_caustic"));
  if (mc == "none" || mc == "disabled") vars.m_flags &= ~unsigned(HF_ENABLE_PT_CAUSTICS);
  else vars.m_flags |= HF_ENABLE_PT_CAUSTICS;
  vars.m_varsI[HV_I_TRACE_DEPTH] = 6;
  vars.m_varsI[HV_I_DIFFUSE_TRACE_DEPTH] = 3;
  vars.m_varsI[29 /*HRT_ENABLE_PATH_REGENERATE*/] = 1;
  vars.m_varsF[HV_F_IMAGE_GAMMA] = 2.2f;
  vars.m_varsF[HV_F_TEXINPUT_GAMMA] = 2.2f;
  vars.m_varsF[15 /*HRT_PATH_TRACE_ERROR*/] = 0.025f;
  vars.m_varsF[3 /*HRT_TRACE_PROCEEDINGS_TRESHOLD*/] = 1e-8f;
  vars.m_varsF[16 /*HRT_PATH_TRACE_CLAMPING*/] = 1e6f;
  vars.m_varsI[33 /*HRT_MMLT_BURN_ITERS*/] = 1024;
  vars.m_varsI[HV_I_MMLT_FIRST_BOUNCE] = 3;   // RenderDriverRTE.cpp:263
  {   // RenderDriverRTE.cpp:196-202
    const std::string ms = xtext(xchild(st, "method_secondary"));
    if (ms == "mmlt" || ms == "MMLT" || ms == "mlt") vars.m_flags |= HF_ENABLE_MMLT; else vars.m_flags &= ~unsigned(HF_ENABLE_MMLT);
  }
  if (xchild(st, "outgamma")) vars.m_varsF[HV_F_IMAGE_GAMMA] = strtof(xtext(xchild(st, "outgamma")).c_str(), nullptr);
  if (xchild(st, "trace_depth")) vars.m_varsI[HV_I_TRACE_DEPTH] = atoi(xtext(xchild(st, "trace_depth")).c_str()) + 1;
  if (xchild(st, "diff_trace_depth")) vars.m_varsI[HV_I_DIFFUSE_TRACE_DEPTH] = atoi(xtext(xchild(st, "diff_trace_depth")).c_str()) + 1;
  m_splitAlphaTree = xchild(st, "split_alpha_tree") && atoi(xtext(xchild(st, "split_alpha_tree")).c_str()) == 1;
  m_pHWLayer->SetAllFlagsAndVars(vars);
  m_pHWLayer->ResizeScreen(m_width, m_height, 0);
  return true;
}

void RenderDriverLite::BeginScene() {
  m_instMatricesInv.clear(); m_instLightInstId.clear(); m_meshIdByInstId.clear(); m_meshRemapListId.clear();
  m_lightsInstanced.clear();
  m_lightIdByInst.clear();
  m_sceneHaveSkyPortals = false;
  m_bvh.ClearScene();
  m_bvhAlpha.ClearScene();
  const int32_t dummyList[2] = {0, 0};
  m_pHWLayer->SetAllRemapLists(dummyList, reinterpret_cast<const int2*>(dummyList), 0, 0);
}

void RenderDriverLite::InstanceMeshes(int32_t a_mesh_id, const float* a_matrices, int32_t a_instNum, const int* a_lightInstId,
                                      const int* a_remapId, const int* a_realInstId) {
  const auto table = m_pGeomStorage->GetTable();
  if (a_mesh_id >= int(table.size()) || table[a_mesh_id] < 0) { m_log += "InstanceMeshes: bad mesh id\n"; return; }
  const char* base = static_cast<const char*>(m_pGeomStorage->GetBegin()) + size_t(table[a_mesh_id]) * 16;
  const HydraPlainMesh* hdr = reinterpret_cast<const HydraPlainMesh*>(base);
  BVH4Builder::InstanceInputData in;
  in.vert4f = reinterpret_cast<const float*>(base + size_t(hdr->vPosOffset) * 16);
  in.indices = reinterpret_cast<const int*>(base + size_t(hdr->vIndicesOffset) * 16);
  in.numVert = hdr->vPosNum;
  in.numIndices = hdr->tIndicesNum;
  in.meshId = a_mesh_id;
  in.matrices = a_matrices;
  in.numInst = a_instNum;
  // Embree hands the reference up to four trees (bvh_access_dll2.cpp:547-600); this builder makes one, or -- when the scene's
  // settings ask for it -- a second one for the instances of alpha-tested meshes, so that the multi-tree walk has a producer
  if (m_splitAlphaTree && MeshHasOpacity(a_mesh_id)) m_bvhAlpha.InstanceTriangleMeshes(in, 1, int(m_meshIdByInstId.size()));
  else m_bvh.InstanceTriangleMeshes(in, 0, int(m_meshIdByInstId.size()));
  for (int i = 0; i < a_instNum; i++) {
    float4x4 m;
    memcpy(m.c, a_matrices + 16 * i, 64);
    m_instMatricesInv.push_back(inverse4x4(m));
    m_instLightInstId.push_back(a_lightInstId[i]);
    m_meshIdByInstId.push_back(a_mesh_id);
    m_meshRemapListId.push_back(a_remapId[i]);
    (void)a_realInstId;
  }
}

// AreaDiffuseLight::Transform, PlainLightConverter.cpp:281-352 + RenderDriverRTE::InstanceLights :2012-2080
void RenderDriverLite::InstanceLights(int32_t a_lightId, const float* a_matrix, const XmlNode** a_lightNodes, int32_t a_instNum, int32_t a_lightGroupId) {
  auto it = m_lights.find(a_lightId);
  if (it == m_lights.end()) { m_log += "InstanceLights: bad light id\n"; return; }
  struct IdsOnExit {   // m_lightIdByLightInstId of the reference: one entry per record this call appends
    RenderDriverLite* self; int32_t id;
    ~IdsOnExit() { self->m_lightIdByInst.resize(self->m_lightsInstanced.size() / HL_FLOATS, id); }
  } idsOnExit{this, a_lightId};
  if (it->second.isSky) {   // :2072-2078: an instanced sky light decides about the back-plate, with or without a <back> node of its own
    const SkyBack sb = m_skyBack.count(a_lightId) ? m_skyBack[a_lightId] : SkyBack();
    m_shadowMatteBackTexId = sb.texId; m_shadowMatteBackGamma = sb.gamma; m_shadowMatteBackMode = sb.mode; m_shadowMatteBackColor = sb.color;
  }
  for (int i = 0; i < a_instNum; i++) {
    float4x4 M;
    memcpy(M.c, a_matrix + 16 * i, 64);
    std::vector<float> copy = it->second.plain;
    float* d = copy.data();
    if (it->second.isDelta) {                     // position by the matrix; spot/direct also rotate their normal (:534-566, 594-626, 700-715)
      const float3 lp0 = mul_point(M, float3(d[HL_POS], d[HL_POS + 1], d[HL_POS + 2]));
      d[HL_POS] = lp0.x; d[HL_POS + 1] = lp0.y; d[HL_POS + 2] = lp0.z;
      if (it->second.kind != 0) {
        const float3 ln0 = mul_vec(M, float3(d[HL_NORM], d[HL_NORM + 1], d[HL_NORM + 2]));
        d[HL_NORM] = ln0.x; d[HL_NORM + 1] = ln0.y; d[HL_NORM + 2] = ln0.z;
      }
      if (it->second.hasIes) transform_ies_matrix(it->second.plain.data(), M, d);   // PointLight::Transform :700-715 (a light without a web keeps its zeros: the reference inverts a zero matrix there)
    } else if (it->second.isSky) {                // SkyDomeLight::Transform returns the light unchanged (:1024-1027)
    } else if (it->second.isMesh) {               // MeshLight::Transform, PlainLightConverter.cpp:795-829: position, the 3x3 part, the area of the transformed triangles
      const float3 mp0 = mul_point(M, float3(d[HL_POS], d[HL_POS + 1], d[HL_POS + 2]));
      d[HL_POS] = mp0.x; d[HL_POS + 1] = mp0.y; d[HL_POS + 2] = mp0.z;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) d[HL_MESH_MATRIX + r * 3 + c] = M.at(r, c);
      double totalSA = 0.0;
      const std::vector<float>& P = it->second.meshPos;
      const std::vector<int32_t>& I = it->second.meshInd;
      for (size_t k = 0; k + 2 < I.size(); k += 3) {
        const float3 A = mul_point(M, float3(P[size_t(I[k]) * 4], P[size_t(I[k]) * 4 + 1], P[size_t(I[k]) * 4 + 2]));
        const float3 B = mul_point(M, float3(P[size_t(I[k + 1]) * 4], P[size_t(I[k + 1]) * 4 + 1], P[size_t(I[k + 1]) * 4 + 2]));
        const float3 C = mul_point(M, float3(P[size_t(I[k + 2]) * 4], P[size_t(I[k + 2]) * 4 + 1], P[size_t(I[k + 2]) * 4 + 2]));
        totalSA += double(0.5f * length(cross(B - A, C - A)));
      }
      d[HL_SURFACE_AREA] = float(totalSA);
    } else if (it->second.isSphere) {             // SphereLight::Transform, PlainLightConverter.cpp:466-491
      const float3 sp0 = mul_point(M, float3(d[HL_POS], d[HL_POS + 1], d[HL_POS + 2]));
      d[HL_POS] = sp0.x; d[HL_POS + 1] = sp0.y; d[HL_POS + 2] = sp0.z;
      const float radius = d[HL_SPHERE_RADIUS] * length(mul_vec(M, normalize(float3(1, 1, 1))));
      d[HL_SPHERE_RADIUS] = radius;
      d[HL_SURFACE_AREA] = 4.0f * 3.1415926535f * radius * radius;
    } else if (it->second.isCylinder) {           // CylinderLight::Transform, PlainLightConverter.cpp:382-413: the radius field stays, the area takes the scale twice
      const float3 cp0 = mul_point(M, float3(d[HL_POS], d[HL_POS + 1], d[HL_POS + 2]));
      d[HL_POS] = cp0.x; d[HL_POS + 1] = cp0.y; d[HL_POS + 2] = cp0.z;
      const float mult = length(mul_vec(M, normalize(float3(1, 1, 1))));
      const float newRadius = d[HL_CYL_RADIUS] * mult;
      d[HL_SURFACE_AREA] = (d[HL_CYL_ZMAX] - d[HL_CYL_ZMIN]) * mult * newRadius * d[HL_CYL_PHIMAX];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) d[HL_CYL_MATRIX + r * 3 + c] = M.at(r, c);
    } else {                                      // AreaDiffuseLight::Transform, :281-352
      const float3 lpos = mul_point(M, float3(d[HL_POS], d[HL_POS + 1], d[HL_POS + 2]));
      d[HL_POS] = lpos.x; d[HL_POS + 1] = lpos.y; d[HL_POS + 2] = lpos.z;
      const float3 ln = mul_vec(M, float3(d[HL_NORM], d[HL_NORM + 1], d[HL_NORM + 2]));
      d[HL_NORM] = ln.x; d[HL_NORM + 1] = ln.y; d[HL_NORM + 2] = ln.z;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) d[HL_AREA_MATRIX + r * 3 + c] = M.at(r, c);
      if (it->second.hasIes) transform_ies_matrix(it->second.plain.data(), M, d);
      if (it->second.isDisk) {
        const float3 vert = mul_vec(M, normalize(float3(1, 1, 1)));
        const float radius = d[HL_AREA_SIZE_X] * length(vert);
        d[HL_SURFACE_AREA] = 3.1415926535f * radius * radius;
      } else {
        const float sx = d[HL_AREA_SIZE_X], sy = d[HL_AREA_SIZE_Y];
        const float3 v0 = mul_point(M, float3(-sx, 0, -sy)), v1 = mul_point(M, float3(-sx, 0, sy)), v2 = mul_point(M, float3(sx, 0, sy));
        d[HL_SURFACE_AREA] = length(v1 - v0) * length(v1 - v2);
      }
    }
    // the instance node's own attributes, every light type alike (RenderDriverRTE.cpp:2032-2066)
    bool doNotSampleMe = false;
    if (a_lightNodes && a_lightNodes[i]) {
      const XmlNode* n = a_lightNodes[i];
      if (n->has_attr("color_mult")) {
        float cm[3] = {1, 1, 1};
        parse_floats(n->attr("color_mult"), cm, 3);
        d[HL_COLOR] *= cm[0]; d[HL_COLOR + 1] *= cm[1]; d[HL_COLOR + 2] *= cm[2];
      }
      if (n->has_attr("do_not_sample_me")) { const std::string v = n->attr("do_not_sample_me"); doNotSampleMe = !v.empty() && (v[0] == '1' || v[0] == 't' || v[0] == 'T' || v[0] == 'y' || v[0] == 'Y'); }   // pugi as_bool
      if (n->has_attr("prob_mult")) d[HL_PROB_MULT] = n->attr_float("prob_mult");
    }
    put_i(d, HL_GROUP_ID, a_lightGroupId);
    d[HL_PICK_PROB_REV] = 1.0f;
    d[HL_PICK_PROB_FWD] = 1.0f;
    if (doNotSampleMe) put_i(d, HL_FLAGS, get_i(d, HL_FLAGS) | HLF_DO_NOT_SAMPLE_ME);
    m_lightsInstanced.insert(m_lightsInstanced.end(), copy.begin(), copy.end());
  }
  if (get_i(it->second.plain.data(), HL_FLAGS) & HLF_SKY_PORTAL) m_sceneHaveSkyPortals = true;   // :2069-2070
}

// RenderDriverRTE_PdfTables.cpp:575-647
std::vector<float> RenderDriverLite::CalcLightPickProbTable(bool a_fwd) {
  const size_t n = m_lightsInstanced.size() / HL_FLOATS;
  std::vector<float> pick(n);
  std::map<int, int> groups;
  std::set<int> disableSky;
  int noGroups = 0;
  for (size_t i = 0; i < n; i++) {
    const int g = get_i(&m_lightsInstanced[i * HL_FLOATS], HL_GROUP_ID);
    if (g == -1) noGroups++; else groups[g]++;
    // sic: the reference collects light ids here and looks them up by position in the instanced array below (:598-602, 631-635)
    if (get_i(&m_lightsInstanced[i * HL_FLOATS], HL_FLAGS) & HLF_SKY_PORTAL) disableSky.insert(get_i(&m_lightsInstanced[i * HL_FLOATS], HL_AREA_SKY_SOURCE));
  }
  noGroups += int(groups.size());
  const float pickGroupProb = 1.0f / float(noGroups);
  for (size_t i = 0; i < n; i++) {
    float* d = &m_lightsInstanced[i * HL_FLOATS];
    const int g = get_i(d, HL_GROUP_ID);
    float pp = (g == -1) ? pickGroupProb : pickGroupProb / float(groups[g]);
    if (get_i(d, HL_FLAGS) & HLF_DO_NOT_SAMPLE_ME) pp = 0.0f;
    if (get_i(d, HL_TYPE) == HLT_SKY_DOME && (a_fwd || disableSky.count(int(i)))) pp = 0.0f;
    if (length(float3(d[HL_COLOR], d[HL_COLOR + 1], d[HL_COLOR + 2])) < 0.01f) pp = 0.0f;
    if (d[HL_PROB_MULT] > 0.0f) pp *= d[HL_PROB_MULT];
    d[a_fwd ? HL_PICK_PROB_FWD : HL_PICK_PROB_REV] = pp;
    pick[i] = pp;
  }
  return pick;
}

// RenderDriverRTE::MeshHaveOpacity, RenderDriverRTE_AlphaTestTable.cpp:41-63
bool RenderDriverLite::MeshHasOpacity(int32_t a_meshId) const {
  const auto table = m_pGeomStorage->GetTable();
  if (a_meshId < 0 || a_meshId >= int(table.size()) || table[a_meshId] < 0) return false;
  const char* base = static_cast<const char*>(m_pGeomStorage->GetBegin()) + size_t(table[a_meshId]) * 16;
  const HydraPlainMesh* hdr = reinterpret_cast<const HydraPlainMesh*>(base);
  const int* mind = reinterpret_cast<const int*>(base + size_t(hdr->mIndicesOffset) * 16);
  for (int i = 0; i < hdr->tIndicesNum / 3; i++) {
    auto p = m_matOpacity.find(mind[i]);
    if (p != m_matOpacity.end() && p->second.texId != int32_t(HYDRA_INVALID_TEXTURE)) return true;
  }
  return false;
}
// CompressTexCoord16 + WrapVal, RenderDriverRTE_AlphaTestTable.cpp:25-39, cglobals.h:3014-3022
static uint32_t compress_tex_coord16(float x, float y) {
  auto wrap = [](float v) { return v > 1.0f ? v - float(int(v)) : (v < -1.0f ? float(int(v)) - v : v); };
  auto cl = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
  const float tx = cl(0.5f * wrap(x) + 0.5f), ty = cl(0.5f * wrap(y) + 0.5f);
  return (uint32_t(ty * 65535.0f) << 16) | uint32_t(tx * 65535.0f);
}
// RenderDriverRTE::CreateAlphaTestTable, RenderDriverRTE_AlphaTestTable.cpp:65-224: per tree one uint2 for every float4 of the
// triangle list -- triangle float4 0: {position of the opacity sampler in this table | INVALID_TEXTURE, uv of A}, 1: {smooth opacity
// 0/1, uv of B}, 2: {skip shadow 0/1, uv of C}, header float4: {-1, -1} -- followed by the samplers (6 uint2 each)
void RenderDriverLite::CreateAlphaTestTable(ConvertionResult& cr) {
  int maxSamplers = 0;
  for (const auto& kv : m_matOpacity) if (kv.second.texId != int32_t(HYDRA_INVALID_TEXTURE) || kv.second.skipShadow) maxSamplers++;
  if (maxSamplers == 0) return;
  const int INV = int32_t(HYDRA_INVALID_TEXTURE);
  const auto geomTable = m_pGeomStorage->GetTable();
  const char* geomBase = static_cast<const char*>(m_pGeomStorage->GetBegin());
  std::vector<Opacity> samplers;                 // one list for all trees, as in the reference
  std::map<int, int> samplerOf;
  for (int tree = 0; tree < cr.treesNum && tree < 2; tree++) {
    const int numPrims = cr.trif4Num[tree];
    std::vector<uint32_t>& out = m_alphaTable[tree];
    out.assign(size_t(numPrims + maxSamplers * 6) * 2, 0u);
    bool haveOpacity = false;
    const int32_t* i4 = reinterpret_cast<const int32_t*>(cr.pTriangleData[tree]);
    for (int off = 0; off < numPrims;) {
      if (i4[off * 4 + 2] == -1 && i4[off * 4 + 3] == -1) { out[size_t(off) * 2] = 0xFFFFFFFFu; out[size_t(off) * 2 + 1] = 0xFFFFFFFFu; off++; continue; }
      const int primId = i4[off * 4 + 3], geomId = i4[(off + 1) * 4 + 3];
      const char* base = geomBase + size_t(geomTable[geomId]) * 16;
      const HydraPlainMesh* hdr = reinterpret_cast<const HydraPlainMesh*>(base);
      const int* vind = reinterpret_cast<const int*>(base + size_t(hdr->vIndicesOffset) * 16);
      const float* vpos = reinterpret_cast<const float*>(base + size_t(hdr->vPosOffset) * 16);
      const float* vnrm = reinterpret_cast<const float*>(base + size_t(hdr->vNormOffset) * 16);
      const int* mind = reinterpret_cast<const int*>(base + size_t(hdr->mIndicesOffset) * 16);
      auto p = m_matOpacity.find(mind[primId]);
      const bool tested = (p != m_matOpacity.end()) && (p->second.texId != INV || p->second.skipShadow);
      for (int k = 0; k < 3; k++) { out[size_t(off + k) * 2] = uint32_t(INV); out[size_t(off + k) * 2 + 1] = 0xFFFFFFFFu; }
      if (tested) {
        haveOpacity = true;
        int rel;
        auto q = samplerOf.find(mind[primId]);
        if (q == samplerOf.end()) { rel = int(samplers.size()); samplerOf[mind[primId]] = rel; samplers.push_back(p->second); } else rel = q->second;
        out[size_t(off) * 2] = (p->second.texId != INV) ? uint32_t(numPrims + rel * 6) : uint32_t(INV);
        out[size_t(off + 1) * 2] = p->second.smooth ? 1u : 0u;
        out[size_t(off + 2) * 2] = p->second.skipShadow ? 1u : 0u;
        for (int k = 0; k < 3; k++) {
          const int v = vind[primId * 3 + k];
          out[size_t(off + k) * 2 + 1] = compress_tex_coord16(vpos[v * 4 + 3], vnrm[v * 4 + 3]);   // u rides in pos.w, v in norm.w
        }
      }
      off += 3;
    }
    for (size_t i = 0; i < samplers.size(); i++) memcpy(&out[size_t(numPrims + int(i) * 6) * 2], samplers[i].sampler, 48);
    if (haveOpacity) { cr.pTriangleAlpha[tree] = out.data(); cr.triAfNum[tree] = int(out.size() / 2); }
    else { cr.pTriangleAlpha[tree] = nullptr; cr.triAfNum[tree] = 0; }
  }
}

void RenderDriverLite::EndScene() {
  // HYDRA_GPU_BVH=<device>: the mesh trees are built on that GPU (LBVH, hydracore_amd/csrc/hydra_bvh.hip) instead of by the host's binned-SAH build
  if (const char* e = getenv("HYDRA_GPU_BVH")) { m_bvh.gpuBuildDevice = atoi(e); m_bvhAlpha.gpuBuildDevice = atoi(e); }
  m_bvh.CommitScene();
  if (m_bvh.gpuBuildDevice >= 0) m_log += "BVH: mesh trees built on GPU " + std::to_string(m_bvh.gpuBuildDevice) + " in " + std::to_string(m_bvh.statGpuBuildMs) + " ms of device time\n";
  {
    ConvertionResult cr = m_bvh.ConvertMap();
    if (m_splitAlphaTree && m_bvhAlpha.HasInstances()) {
      m_bvhAlpha.CommitScene();
      const ConvertionResult cr2 = m_bvhAlpha.ConvertMap();
      cr.bvhType[1] = cr2.bvhType[0]; cr.pBVH[1] = cr2.pBVH[0]; cr.pTriangleData[1] = cr2.pTriangleData[0];
      cr.nodesNum[1] = cr2.nodesNum[0]; cr.trif4Num[1] = cr2.trif4Num[0];
      cr.treesNum = 2;
    }
    CreateAlphaTestTable(cr);
    m_pHWLayer->SetAllBVH4(cr, nullptr, 0);
    m_bvh.ConvertUnmap();
    m_bvhAlpha.ConvertUnmap();
  }
  float bmin[3], bmax[3];
  m_bvh.GetBounds(bmin, bmax);
  if (m_splitAlphaTree && m_bvhAlpha.HasInstances()) {
    float b2[3], t2[3];
    m_bvhAlpha.GetBounds(b2, t2);
    for (int a = 0; a < 3; a++) { bmin[a] = std::min(bmin[a], b2[a]); bmax[a] = std::max(bmax[a], t2[a]); }
  }
  const float3 half = 0.5f * (float3(bmax[0], bmax[1], bmax[2]) - float3(bmin[0], bmin[1], bmin[2]));
  const float3 center = 0.5f * (float3(bmax[0], bmax[1], bmax[2]) + float3(bmin[0], bmin[1], bmin[2]));

  if (m_instMatricesInv.empty()) RunTimeError("RenderDriverRTE::EndScene, no instances in the scene!");
  m_pHWLayer->SetAllInstMatrices(m_instMatricesInv.data(), int32_t(m_instMatricesInv.size()));
  m_pHWLayer->SetAllInstIdToRemapId(m_meshRemapListId.data(), int32_t(m_meshRemapListId.size()));

  auto vars = m_pHWLayer->GetAllFlagsAndVars();
  vars.m_varsF[HV_F_BSPHERE_CENTER_X] = center.x;
  vars.m_varsF[HV_F_BSPHERE_CENTER_X + 1] = center.y;
  vars.m_varsF[HV_F_BSPHERE_CENTER_X + 2] = center.z;
  vars.m_varsF[HV_F_BSPHERE_RADIUS] = length(half);
  vars.m_varsI[HV_I_SHADOW_MATTE_BACK] = m_shadowMatteBackTexId;            // RenderDriverRTE.cpp:1487-1492
  vars.m_varsF[HV_F_BACK_TEXINPUT_GAMMA] = m_shadowMatteBackGamma;
  vars.m_varsI[HV_I_SHADOW_MATTE_BACK_MODE] = m_shadowMatteBackMode;
  vars.m_varsF[HV_F_SHADOW_MATTE_BACK_COLOR_X] = m_shadowMatteBackColor.x;
  vars.m_varsF[HV_F_SHADOW_MATTE_BACK_COLOR_X + 1] = m_shadowMatteBackColor.y;
  vars.m_varsF[HV_F_SHADOW_MATTE_BACK_COLOR_X + 2] = m_shadowMatteBackColor.z;
  m_pHWLayer->SetAllFlagsAndVars(vars);

  size_t nl = m_lightsInstanced.size() / HL_FLOATS;
  // a Perez sky takes direction and colour of its sun from the first instance of light `sun_id`
  // (RenderDriverRTE::BuildSkyPortalsDependencyDummyInstances, RenderDriverRTE.cpp:1603-1647)
  for (size_t i = 0; i < nl; i++) {
    float* sky = &m_lightsInstanced[i * HL_FLOATS];
    if (get_i(sky, HL_TYPE) != HLT_SKY_DOME || get_i(sky, HL_SKY_SUN_DIR_ID) == -1) continue;
    for (size_t j = 0; j < nl && j < m_lightIdByInst.size(); j++)
      if (m_lightIdByInst[j] == get_i(sky, HL_SKY_SUN_DIR_ID)) {
        const float* sun = &m_lightsInstanced[j * HL_FLOATS];
        for (int k = 0; k < 3; k++) { sky[HL_SKY_SUN_DIR + k] = sun[HL_NORM + k]; sky[HL_SKY_SUN_COLOR + k] = sun[HL_COLOR + k]; }
        break;
      }
  }
  {   // (1) of the same function: the layer is told whether portals exist
    auto v26 = m_pHWLayer->GetAllFlagsAndVars();
    v26.m_varsI[26 /*HRT_HRT_SCENE_HAVE_PORTALS*/] = m_sceneHaveSkyPortals ? 1 : 0;
    m_pHWLayer->SetAllFlagsAndVars(v26);
  }
  if (m_sceneHaveSkyPortals) {   // (3) sky lights no instance names get a record all the same, (4) every portal learns how far away its sky's record is (:1653-1684)
    std::map<int32_t, int32_t> alreadyInstanced;
    for (size_t i = 0; i < nl && i < m_lightIdByInst.size(); i++)
      if (get_i(&m_lightsInstanced[i * HL_FLOATS], HL_TYPE) == HLT_SKY_DOME) alreadyInstanced[m_lightIdByInst[i]] = int32_t(i);
    for (const auto& kv : m_lights) {
      if (!kv.second.isSky || alreadyInstanced.count(kv.first)) continue;
      m_lightsInstanced.insert(m_lightsInstanced.end(), kv.second.plain.begin(), kv.second.plain.end());
      m_lightIdByInst.push_back(kv.first);
      alreadyInstanced[kv.first] = int32_t(m_lightsInstanced.size() / HL_FLOATS - 1);
    }
    nl = m_lightsInstanced.size() / HL_FLOATS;
    for (size_t i = 0; i < nl; i++) {
      float* d = &m_lightsInstanced[i * HL_FLOATS];
      if (!(get_i(d, HL_FLAGS) & HLF_SKY_PORTAL)) continue;
      const auto sky = alreadyInstanced.find(get_i(d, HL_AREA_SKY_SOURCE));
      if (sky == alreadyInstanced.end()) RunTimeError("EndScene: sky portal names light " + std::to_string(get_i(d, HL_AREA_SKY_SOURCE)) + ", which is not a sky light");   // the reference's map would hand out record 0 here
      put_i(d, HL_AREA_SKY_OFFSET, sky->second - int32_t(i));
    }
    m_sceneHaveSkyPortals = false;
  }
  if (nl > 0) {
    m_pHWLayer->SetAllInstLightInstId(m_instLightInstId.data(), int32_t(m_instLightInstId.size()));
    const std::vector<float> rev = CalcLightPickProbTable(false), fwd = CalcLightPickProbTable(true);
    auto prefix = [](const std::vector<float>& v) {
      std::vector<float> acc(v.size() + 1);
      float a = 0.0f;
      for (size_t i = 0; i < v.size(); i++) { acc[i] = a; a += v[i]; }
      acc[v.size()] = a;
      return acc;
    };
    const std::vector<float> tRev = prefix(rev), tFwd = prefix(fwd);
    const float nRev = 1.0f / tRev.back(), nFwd = tFwd.back() > 0.0f ? 1.0f / tFwd.back() : 0.0f;   // a lone sky light has no forward sampler
    for (size_t i = 0; i < nl; i++) {
      m_lightsInstanced[i * HL_FLOATS + HL_PICK_PROB_FWD] *= nFwd;
      m_lightsInstanced[i * HL_FLOATS + HL_PICK_PROB_REV] *= nRev;
    }
    m_pHWLayer->SetAllLightsSelectTable(tRev.data(), int32_t(tRev.size()), false);
    m_pHWLayer->SetAllLightsSelectTable(tFwd.data(), int32_t(tFwd.size()), true);
    m_pHWLayer->SetAllPODLights(m_lightsInstanced.data(), nl);
  } else {
    m_log += "WARNING: RenderDriverRTE::EndScene(), no lights!\n";
    std::vector<int32_t> none(m_instLightInstId.size(), -1);
    m_pHWLayer->SetAllInstLightInstId(none.data(), int32_t(none.size()));
    m_pHWLayer->SetAllLightsSelectTable(nullptr, 0, false);
    m_pHWLayer->SetAllLightsSelectTable(nullptr, 0, true);
    m_pHWLayer->SetAllPODLights(nullptr, 0);
  }
  m_pHWLayer->PrepareEngineTables();
}

void RenderDriverLite::Draw() {
  const float aspect = float(m_width) / float(m_height);
  // RenderDriverRTE::CalcCameraMatrices, RenderDriverRTE.cpp:1301-1324
  float4x4 proj = m_camera.useMatrices ? m_camera.mProj : perspective_matrix(m_camera.fov, aspect, m_camera.nearPlane, m_camera.farPlane);
  float4x4 worldView = m_camera.useMatrices ? m_camera.mWorldView : look_at(m_camera.pos, m_camera.lookAt, m_camera.up);
  float4x4 projInv = inverse4x4(proj), mvInv = inverse4x4(worldView);
  m_pHWLayer->SetCamMatrices(&projInv.c[0][0], &mvInv.c[0][0], &proj.c[0][0], &worldView.c[0][0], aspect,
                             (3.14159265358979323846f / 180.f) * m_camera.fov, m_camera.lookAt);
  m_pHWLayer->PrepareEngineGlobals();
  SharedDataLayer* shared = dynamic_cast<SharedDataLayer*>(m_pHWLayer);
  if (shared != nullptr && !shared->HasDevice()) return;   // host-blob layer: buffers only, nothing to trace
  if (!m_ptInitDone) { m_pHWLayer->InitPathTracing(m_seed); m_ptInitDone = true; }
  m_pHWLayer->BeginTracingPass();
  m_pHWLayer->EndTracingPass();
}

// ------------------------------------------------------------------------------------------------ HydraAPI's part
void RenderDriverLite::LoadSceneLibrary(const std::string& libPath, int a_width, int a_height, int a_traceDepth, int a_enableDof) {
  std::vector<char> xmlData;
  m_libPath = libPath;
  std::string xmlPath = libPath + "/statex_00001.xml";
  if (!read_file(xmlPath, xmlData)) RunTimeError("LoadSceneLibrary: can't open " + xmlPath);
  const std::string src(xmlData.begin(), xmlData.end());
  XmlParser parser(src);
  std::unique_ptr<XmlNode> doc = parser.parse_document();
  m_forceDof = a_enableDof;

  const XmlNode* texLib = doc->child("textures_lib");
  const XmlNode* matLib = doc->child("materials_lib");
  const XmlNode* lgtLib = doc->child("lights_lib");
  const XmlNode* camLib = doc->child("cam_lib");
  const XmlNode* geoLib = doc->child("geometry_lib");
  const XmlNode* rndLib = doc->child("render_lib");
  const XmlNode* scnLib = doc->child("scenes");
  if (!matLib || !geoLib || !scnLib) RunTimeError("LoadSceneLibrary: incomplete scene library " + xmlPath);

  auto maxId = [](const XmlNode* lib, const char* name) {
    int m = -1;
    if (lib) for (auto* n : lib->children_named(name)) m = std::max(m, n->attr_int("id"));
    return m;
  };
  const int imgNum = maxId(texLib, "texture") + 1, matNum = maxId(matLib, "material") + 2;
  const int lightNum = std::max(maxId(lgtLib, "light") + 1, 1), meshNum = maxId(geoLib, "mesh") + 1;

  const XmlNode* settings = rndLib ? rndLib->child("render_settings") : nullptr;
  if (a_width <= 0 && settings && settings->child("width")) a_width = atoi(settings->child("width")->text.c_str());
  if (a_height <= 0 && settings && settings->child("height")) a_height = atoi(settings->child("height")->text.c_str());
  if (a_width > 0) m_width = a_width;
  if (a_height > 0) m_height = a_height;

  const XmlNode* scene = scnLib->child("scene");
  if (!scene) RunTimeError("LoadSceneLibrary: no <scene>");
  const auto lightInstNodes = scene->children_named("instance_light");
  const int lightInstNum = int(lightInstNodes.size());

  AllocAll(imgNum, matNum, std::max(lightNum, lightInstNum) + 4, meshNum);   // + room in the pdf-table table for the two tables of a sky light

  if (texLib)
    for (auto* t : texLib->children_named("texture")) {
      if (std::string(t->attr("type")) == "proc") { UpdateImageProc(t->attr_int("id"), t); continue; }
      if (!t->has_attr("loc")) continue;                       // delayed-load textures without data (dl="1")
      std::vector<char> d;
      if (!read_file(libPath + "/" + t->attr("loc"), d) || d.size() < 8) { m_log += std::string("missing texture chunk ") + t->attr("loc") + "\n"; continue; }
      int32_t wh[2];
      memcpy(wh, d.data(), 8);
      const size_t bytes = d.size() - 8;
      const int bpp = int(bytes / (size_t(wh[0]) * size_t(wh[1])));
      if (bpp != 4 && bpp != 16) { Unsupported("texture bpp " + std::to_string(bpp)); continue; }
      UpdateImage(t->attr_int("id"), wh[0], wh[1], bpp, 4, d.data() + 8);
    }
  // EndTexturesUpdate :485-574: the layer builds the scene's procedural textures (nothing to do without any: the layer keeps no program)
  m_procTexProgram = m_procTextures.empty() ? std::string() : ProcTexProgramText();
  m_pHWLayer->RecompileProcTexShaders(m_procTexProgram);
  {   // BeginMaterialUpdate ... EndMaterialUpdate (:1749-1843): everything but the blends of two materials first, then those in id order
    std::map<int, const XmlNode*> blends;
    for (auto* m : matLib->children_named("material")) {
      if (std::string(m->attr("type")) == "hydra_blend") blends[m->attr_int("id")] = m;
      else UpdateMaterial(m->attr_int("id"), m);
    }
    for (const auto& bl : blends) UpdateMaterial(bl.first, bl.second);
    m_materialNodes.clear();   // they point into this function's document
  }
  std::set<int> loadedMeshes;
  for (auto* me : geoLib->children_named("mesh")) {
    std::vector<char> d;
    if (!me->has_attr("loc") || !read_file(libPath + "/" + me->attr("loc"), d)) { m_log += std::string("missing mesh chunk for mesh ") + me->attr("id") + "\n"; continue; }
    const int vertNum = me->attr_int("vertNum"), triNum = me->attr_int("triNum");
    auto arr = [&](const char* name) -> const char* {
      const XmlNode* a = me->child(name);
      if (!a) RunTimeError(std::string("mesh without <") + name + ">");
      const size_t off = size_t(atoll(a->attr("offset"))), sz = size_t(atoll(a->attr("bytesize")));
      if (off + sz > d.size()) RunTimeError("mesh chunk is shorter than its XML description");
      return d.data() + off;
    };
    UpdateMesh(me->attr_int("id"), vertNum, triNum, (const float*)arr("positions"), (const float*)arr("normals"),
               (const float*)arr("tangents"), (const float*)arr("texcoords"), (const int*)arr("indices"), (const int*)arr("matindices"));
    loadedMeshes.insert(me->attr_int("id"));
  }
  if (lgtLib)   // after the meshes: a mesh light copies its mesh out of the geometry storage (RenderDriverRTE::UpdateLight :925-938)
    for (auto* l : lgtLib->children_named("light")) UpdateLight(l->attr_int("id"), l);

  UpdateCamera(camLib ? camLib->child("camera") : nullptr);
  UpdateSettings(settings);
  if (a_traceDepth >= 0) {
    auto vars = m_pHWLayer->GetAllFlagsAndVars();
    vars.m_varsI[HV_I_TRACE_DEPTH] = a_traceDepth + 1;
    m_pHWLayer->SetAllFlagsAndVars(vars);
  }

  BeginScene();
  // lights first so that linst_id of mesh instances indexes m_lightsInstanced
  {
    std::vector<const XmlNode*> sorted(lightInstNodes.begin(), lightInstNodes.end());
    std::sort(sorted.begin(), sorted.end(), [](const XmlNode* a, const XmlNode* b) { return a->attr_int("id") < b->attr_int("id"); });
    for (auto* li : sorted) {
      float m[16];
      if (!parse_floats(li->attr("matrix"), m, 16)) RunTimeError("instance_light without matrix");
      const float4x4 M = float4x4::from_row_major(m);
      const XmlNode* nodes[1] = {li};
      InstanceLights(li->attr_int("light_id"), M.data(), nodes, 1, li->attr_int("lgroup_id", -1));
    }
  }
  // mesh instances grouped by mesh id (HydraAPI hands the driver one InstanceMeshes call per mesh)
  std::map<int, std::vector<const XmlNode*>> byMesh;
  for (auto* in : scene->children_named("instance")) byMesh[in->attr_int("mesh_id")].push_back(in);
  for (auto& kv : byMesh) {
    if (!loadedMeshes.count(kv.first)) { m_log += "skipping instances of missing mesh " + std::to_string(kv.first) + "\n"; continue; }
    std::vector<float> mats; std::vector<int> lid, rid, real;
    for (auto* in : kv.second) {
      float m[16];
      if (!parse_floats(in->attr("matrix"), m, 16)) RunTimeError("instance without matrix");
      const float4x4 M = float4x4::from_row_major(m);
      mats.insert(mats.end(), M.data(), M.data() + 16);
      lid.push_back(in->has_attr("linst_id") ? in->attr_int("linst_id") : -1);
      rid.push_back(in->attr_int("rmap_id", -1));
      real.push_back(in->attr_int("id"));
    }
    InstanceMeshes(kv.first, mats.data(), int(kv.second.size()), lid.data(), rid.data(), real.data());
  }
  EndScene();
}

}  // namespace hydra_host
