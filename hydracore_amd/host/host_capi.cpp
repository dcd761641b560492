// host_capi.cpp -- small C surface over the C++ host layer, for the Python test/bench harness (ctypes).
// Not part of the drop-in boundary: the boundary is include/hydra_hip.h; this only exposes the scene front end and the
// buffers it packs so that tests can hand the SAME bytes to the HIP layer and to the CPU oracle.
#include "render_driver_lite.h"
#include <cstring>
#include <cstdio>
#include <mutex>
#include <chrono>

namespace hydra_host { void* HipLayerHandle(IHWLayer* layer); }
using namespace hydra_host;

struct HostScene {
  RenderDriverLite* drv = nullptr;
  std::string err;
};

// an in-process IHRSharedAccumImage over caller memory: what HydraAPI's shared-memory image is to the reference's layers
// (Lock with a time-out / Unlock around `+=`; header with spp and the receive counter).  Used by the harness to drive
// IHWLayer::SetExternalImageAccumulator / ContribToExternalImageAccumulator.
struct LocalAccumImage : public IHRSharedAccumImage {
  LocalAccumImage(float* data, int w, int h, int depth = 1) : m_data(data) { m_hdr = HRSharedBufferHeader{w, h, depth, 4, 0.0f, 0, 0, 1}; }
  bool Lock(int ms) override { return m_mutex.try_lock_for(std::chrono::milliseconds(ms)); }
  void Unlock() override { m_mutex.unlock(); }
  float* ImageData(int layer) override { return (layer >= 0 && layer < m_hdr.depth) ? m_data + size_t(layer) * size_t(m_hdr.width) * size_t(m_hdr.height) * 4 : nullptr; }
  HRSharedBufferHeader* Header() override { return &m_hdr; }
  float* m_data;
  HRSharedBufferHeader m_hdr;
  std::timed_mutex m_mutex;
};

static void set_err(char* err, int n, const std::string& s) {
  if (err && n > 0) { strncpy(err, s.c_str(), size_t(n - 1)); err[n - 1] = 0; }
}

extern "C" {

// use_hip = 0: host-blob layer (no device); 1: HipHWLayer on device_id
void* hydra_host_open_scene(const char* lib_path, int width, int height, int trace_depth, int enable_dof,
                            int use_hip, int device_id, int seed, char* err, int err_len) {
  HostScene* s = new HostScene();
  try {
    const int w0 = width > 0 ? width : 512, h0 = height > 0 ? height : 512;
    IHWLayer* layer = use_hip ? CreateHipImpl(w0, h0, 0, device_id) : CreateHostBlobImpl(w0, h0, 0);
    s->drv = new RenderDriverLite(layer, w0, h0);
    s->drv->SetSeed(seed);
    s->drv->LoadSceneLibrary(lib_path, width, height, trace_depth, enable_dof);
    if (!use_hip) s->drv->Draw();   // assembles camera matrices + globals header; nothing is traced
    return s;
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    delete s->drv;
    delete s;
    return nullptr;
  }
}

void hydra_host_close_scene(void* p) {
  HostScene* s = static_cast<HostScene*>(p);
  if (!s) return;
  delete s->drv;
  delete s;
}

int hydra_host_width(void* p) { return static_cast<HostScene*>(p)->drv->Width(); }
int hydra_host_height(void* p) { return static_cast<HostScene*>(p)->drv->Height(); }
int hydra_host_unsupported(void* p) { return static_cast<HostScene*>(p)->drv->UnsupportedFeatures(); }
const char* hydra_host_log(void* p) { return static_cast<HostScene*>(p)->drv->Log().c_str(); }
const char* hydra_host_proctex_program(void* p) { return static_cast<HostScene*>(p)->drv->ProcTexProgram().c_str(); }
const char* hydra_host_last_error(void* p) { return static_cast<HostScene*>(p)->err.c_str(); }

// what: 0 globals blob, 1 textures, 2 textures_aux, 3 geom, 4 materials, 5 pdfs, 6 bvh nodes (tree 0), 7 triangle float4 list
// (tree 0), 8 inverse instance matrices, 9 instLightInstId, 10 remap lists, 11 remap table, 12 inst->remap id
int hydra_host_get_buffer(void* p, int what, const void** ptr, size_t* bytes) {
  HostScene* s = static_cast<HostScene*>(p);
  SharedDataLayer* L = dynamic_cast<SharedDataLayer*>(s->drv->Layer());
  if (!L) return -1;
  static const char* names[5] = {"textures", "textures_aux", "geom", "materials", "pdfs"};
  *ptr = nullptr; *bytes = 0;
  if (what == 0) { *ptr = L->GetEngineGlobals(); *bytes = L->GetEngineGlobalsSizeInWords() * 4; }
  else if (what >= 1 && what <= 5) {
    IMemoryStorage* st = L->FindStorage(names[what - 1]);
    if (!st) return -1;
    *ptr = st->GetBegin(); *bytes = st->GetSize();
  }
  else if (what == 6) { *ptr = L->m_bvhTrees[0].m_bvh.data(); *bytes = L->m_bvhTrees[0].m_bvh.size() * sizeof(HydraBVHNode); }
  else if (what == 7) { *ptr = L->m_bvhTrees[0].m_tris.data(); *bytes = L->m_bvhTrees[0].m_tris.size() * 4; }
  else if (what == 8) { *ptr = L->m_instMatrices.data(); *bytes = L->m_instMatrices.size() * 4; }
  else if (what == 9) { *ptr = L->m_instLightInstId.data(); *bytes = L->m_instLightInstId.size() * 4; }
  else if (what == 10) { *ptr = L->m_remapLists.data(); *bytes = L->m_remapLists.size() * 4; }
  else if (what == 11) { *ptr = L->m_remapTable.data(); *bytes = L->m_remapTable.size() * 4; }
  else if (what == 12) { *ptr = L->m_remapInst.data(); *bytes = L->m_remapInst.size() * 4; }
  else if (what == 13) { *ptr = L->m_bvhTrees[0].m_atbl.data(); *bytes = L->m_bvhTrees[0].m_atbl.size() * 4; }     // alpha table of tree 0 (uint2 pairs)
  else if (what == 14) { *ptr = L->m_bvhTrees[1].m_bvh.data(); *bytes = L->m_bvhTrees[1].m_bvh.size() * sizeof(HydraBVHNode); }
  else if (what == 15) { *ptr = L->m_bvhTrees[1].m_tris.data(); *bytes = L->m_bvhTrees[1].m_tris.size() * 4; }
  else if (what == 16) { *ptr = L->m_bvhTrees[1].m_atbl.data(); *bytes = L->m_bvhTrees[1].m_atbl.size() * 4; }
  else return -1;
  return 0;
}

int hydra_host_have_inst(void* p) {
  SharedDataLayer* L = dynamic_cast<SharedDataLayer*>(static_cast<HostScene*>(p)->drv->Layer());
  return (L && L->m_bvhTrees[0].haveInst) ? 1 : 0;
}

int hydra_host_trees_num(void* p) {
  SharedDataLayer* L = dynamic_cast<SharedDataLayer*>(static_cast<HostScene*>(p)->drv->Layer());
  return L ? L->m_bvhTreesNum : 0;
}
int hydra_host_have_inst_tree(void* p, int tree) {
  SharedDataLayer* L = dynamic_cast<SharedDataLayer*>(static_cast<HostScene*>(p)->drv->Layer());
  return (L && tree >= 0 && tree < MAXBVHTREES && L->m_bvhTrees[tree].haveInst) ? 1 : 0;
}

// the hydra_hip_handle behind a HipHWLayer (NULL for the host-blob layer)
void* hydra_host_hip_handle(void* p) { return HipLayerHandle(static_cast<HostScene*>(p)->drv->Layer()); }

// RenderDriverRTE::Draw, `passes` times with `spp` samples per pixel each
// "mmlt" sets HRT_ENABLE_MMLT in the layer's flags, anything else clears it (what RenderDriverRTE::UpdateSettings does for
// <method_secondary>, RenderDriverRTE.cpp:196-202); the accumulated image restarts
int hydra_host_set_render_method(void* p, const char* method) {
  HostScene* s = static_cast<HostScene*>(p);
  try {
    IHWLayer* L = s->drv->Layer();
    auto vars = L->GetAllFlagsAndVars();
    const std::string m = method ? method : "";
    if (m == "mmlt" || m == "MMLT" || m == "mlt") vars.m_flags |= HF_ENABLE_MMLT; else vars.m_flags &= ~unsigned(HF_ENABLE_MMLT);
    L->SetAllFlagsAndVars(vars);
    L->ClearAccumulatedColor();
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

int hydra_host_draw(void* p, int passes, int spp) {
  HostScene* s = static_cast<HostScene*>(p);
  try {
    s->drv->Layer()->SetRaysPerPixel(spp);
    for (int i = 0; i < passes; i++) s->drv->Draw();
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

int hydra_host_get_hdr(void* p, float* rgba, int w, int h) {
  HostScene* s = static_cast<HostScene*>(p);
  try {
    s->drv->Layer()->FinishAll();
    s->drv->GetFrameBufferHDR(rgba, w, h);
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

// shared accumulation image over `rgba` (w*h*4 floats, caller-owned, zero it first).  attach = 1: SetExternalImageAccumulator
// (every later Draw contributes at the end of its pass); attach = 0: one explicit ContribToExternalImageAccumulator call.
// Returns an image handle; hydra_host_shared_image_stat reads its header; hydra_host_shared_image_close detaches and frees.
void* hydra_host_shared_image_open(void* p, float* rgba, int w, int h, int attach) {
  HostScene* s = static_cast<HostScene*>(p);
  LocalAccumImage* img = new LocalAccumImage(rgba, w, h);
  try {
    if (attach) s->drv->Layer()->SetExternalImageAccumulator(img);
    else { s->drv->Layer()->FinishAll(); s->drv->Layer()->ContribToExternalImageAccumulator(img); }
    return img;
  } catch (const std::exception& e) {
    s->err = e.what();
    delete img;
    return nullptr;
  }
}
// IHWLayer::EvalGBuffer into an in-process shared image of `depth` float4 layers over `layers` (depth * w * h * 4 floats): depth 3 -> the G-buffer
// goes to layers 1 and 2, depth 4 -> layers 2 and 3 (GPUOCLLayerOther.cpp:725-741).  state_io: in = Header()->gbufferIsEmpty before the call, out = after.
int hydra_host_eval_gbuffer(void* p, float* layers, int w, int h, int depth, const int32_t* inst_remap, int remap_size, int* state_io) {
  HostScene* s = static_cast<HostScene*>(p);
  if (!s || !layers || !state_io) return -1;
  LocalAccumImage img(layers, w, h, depth);
  img.m_hdr.gbufferIsEmpty = *state_io;
  try {
    s->drv->Layer()->EvalGBuffer(&img, std::vector<int32_t>(inst_remap, inst_remap + (inst_remap ? remap_size : 0)));
    *state_io = img.m_hdr.gbufferIsEmpty;
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}
int hydra_host_shared_image_stat(void* image, float* spp, int* counterRcv) {
  LocalAccumImage* img = static_cast<LocalAccumImage*>(image);
  if (!img) return -1;
  *spp = img->m_hdr.spp; *counterRcv = img->m_hdr.counterRcv;
  return 0;
}
void hydra_host_shared_image_close(void* p, void* image) {
  HostScene* s = static_cast<HostScene*>(p);
  if (s) s->drv->Layer()->SetExternalImageAccumulator(nullptr);
  delete static_cast<LocalAccumImage*>(image);
}

float hydra_host_get_spp(void* p) { return static_cast<HostScene*>(p)->drv->Layer()->GetSPP(); }

int hydra_host_bvh_stats(void* p, size_t* quads, size_t* leaves, size_t* tris) {
  BVH4Builder& b = static_cast<HostScene*>(p)->drv->Builder();
  *quads = b.statInnerQuads; *leaves = b.statLeaves; *tris = b.statTriangles;
  return 0;
}

}  // extern "C"
