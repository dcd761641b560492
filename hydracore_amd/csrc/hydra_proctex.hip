// hydra_proctex.hip -- procedural textures, host half: builds the scene's own texture functions into a gfx950 kernel at scene load.
//
// Reference: RenderDriverRTE splices the scene library's data/proctex_*.c and one generated call per texture into shaders/texproc.cl and hands
// the text to IHWLayer::RecompileProcTexShaders (RenderDriverRTE_ProcTex.cpp:446-629, IHWLayer.h:205); GPUOCLLayer rebuilds its OpenCL program
// from it and sizes the per-ray result buffer (GPUOCLLayer.cpp:788-810).  Here the two spliced regions are cut out of that same text, placed
// into the frame of hk_proctex_rt.h and compiled with hiprtc against this library's own device headers (embedded as text at build time:
// hk_proctex_amalgam.inc), so the run-time kernel shares evalSurface, the texture fetch and every layout constant with the built-in kernels.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>
#include "hk_kernels.h"

static const char* const kAmalgam =
#include "hk_proctex_amalgam.inc"
    ;

struct HkProcTexProgram {
  hipModule_t module = nullptr;
  hipFunction_t kPaths = nullptr, kPoints = nullptr;
  std::string log;
};

namespace {
// the lines of `text` after the first line containing `begin`, up to the first later line containing one of `ends`
bool cut_region(const std::string& text, const char* begin, const std::vector<const char*>& ends, std::string& out) {
  const size_t b = text.find(begin);
  if (b == std::string::npos) return false;
  size_t from = text.find('\n', b);
  if (from == std::string::npos) return false;
  from++;
  size_t to = std::string::npos;
  for (const char* e : ends) {
    const size_t at = text.find(e, from);
    if (at != std::string::npos && at < to) to = at;
  }
  if (to == std::string::npos) return false;
  to = text.rfind('\n', to);                    // whole lines only
  if (to == std::string::npos || to < from) to = from;
  out = text.substr(from, to - from);
  return true;
}
bool replace_marker(std::string& frame, const char* marker, const std::string& with) {
  const size_t at = frame.find(marker);
  if (at == std::string::npos) return false;
  frame.replace(at, strlen(marker), with);
  return true;
}
}   // namespace

// the text -> a gfx950 code object (no device needed: hiprtc compiles for the named architecture)
bool hk_proctex_compile(const char* source, size_t len, std::vector<char>& code, std::string& log, std::string& err) {
  const std::string text(source, len);
  std::string user, eval;
  // the markers are the reference's (shaders/texproc.cl:69, 173); the ends are what follows them in its file, or this layer's own end marks
  if (!cut_region(text, "#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:", {"#HK_END_OF_PROCEDURAL_TEXTURES", "const int findArgDataOffsetInTable"}, user)) {
    err = "proctex_compile: the text has no '#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:' region (expected the layout of shaders/texproc.cl after RenderDriverRTE's splice)";
    return false;
  }
  if (!cut_region(text, "#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:", {"#HK_END_OF_PROCEDURAL_TEXTURES_EVAL", "// BREAK SHADER CACHE AT:", "// (5) take what we need"}, eval)) {
    err = "proctex_compile: the text has no '#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:' region";
    return false;
  }
  std::string program = kAmalgam;
  // the scene's functions carry no __device__: compile them for the device as they are
  if (!replace_marker(program, "//#HK_PROCTEX_USER_CODE", "#pragma clang force_cuda_host_device begin\n#line 1 \"procedural_textures.c\"\n" + user + "\n#pragma clang force_cuda_host_device end\n") ||
      !replace_marker(program, "//#HK_PROCTEX_EVAL_CODE", "#line 1 \"procedural_texture_calls.c\"\n" + eval + "\n")) {
    err = "proctex_compile: internal error: the embedded frame lost its markers";
    return false;
  }
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, program.c_str(), "hydra_proctex.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { err = "proctex_compile: hiprtcCreateProgram failed"; return false; }
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value", "-Wno-pragma-once-outside-header"};
  if (getenv("HYDRA_HIP_PROCTEX_RESOURCES") != nullptr) opts.push_back("-Rpass-analysis=kernel-resource-usage");   // registers / spills / occupancy of k_proctex into the build log
  const hiprtcResult rc = hiprtcCompileProgram(prog, int(opts.size()), opts.data());
  size_t logSize = 0;
  log.clear();
  if (hiprtcGetProgramLogSize(prog, &logSize) == HIPRTC_SUCCESS && logSize > 1) { log.resize(logSize); (void)hiprtcGetProgramLog(prog, &log[0]); }
  if (rc != HIPRTC_SUCCESS) {
    err = std::string("proctex_compile: the procedural textures do not compile (") + hiprtcGetErrorString(rc) + "):\n" + log;
    (void)hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t codeSize = 0;
  if (hiprtcGetCodeSize(prog, &codeSize) != HIPRTC_SUCCESS || codeSize == 0) { err = "proctex_compile: no code object"; (void)hiprtcDestroyProgram(&prog); return false; }
  code.resize(codeSize);
  const bool got = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
  (void)hiprtcDestroyProgram(&prog);
  if (!got) { err = "proctex_compile: hiprtcGetCode failed"; return false; }
  return true;
}

HkProcTexProgram* hk_proctex_build(const char* source, size_t len, std::string& err) {
  std::vector<char> code;
  std::string log;
  if (!hk_proctex_compile(source, len, code, log, err)) return nullptr;
  HkProcTexProgram* p = new HkProcTexProgram();
  p->log = log;
  if (hipModuleLoadData(&p->module, code.data()) != hipSuccess) { err = "proctex_compile: hipModuleLoadData failed (no HIP device?)"; delete p; return nullptr; }
  if (hipModuleGetFunction(&p->kPaths, p->module, "k_proctex") != hipSuccess || hipModuleGetFunction(&p->kPoints, p->module, "k_proctex_points") != hipSuccess) {
    err = "proctex_compile: the compiled module has no k_proctex";
    (void)hipModuleUnload(p->module);
    delete p;
    return nullptr;
  }
  return p;
}

void hk_proctex_free(HkProcTexProgram* p) {
  if (p == nullptr) return;
  if (p->module) (void)hipModuleUnload(p->module);
  delete p;
}

hipError_t hk_proctex_launch(HkProcTexProgram* p, int grid, hipStream_t stream, const SceneDev& s, const SegQ& q, const float4* pos4, const float4* dir4, const HydraLiteHit* hits,
                             int* ids, uint2* vals, int stride, int maxNum) {
  SceneDev sv = s; SegQ qv = q;
  void* args[] = {&sv, &qv, &pos4, &dir4, &hits, &ids, &vals, &stride, &maxNum};
  return hipModuleLaunchKernel(p->kPaths, unsigned(grid), 1, 1, 256, 1, 1, 0, stream, args, nullptr);
}
hipError_t hk_proctex_launch_points(HkProcTexProgram* p, hipStream_t stream, const SceneDev& s, int n, const float4* pos4, const float4* dir4, const HydraLiteHit* hits,
                                    int* ids, uint2* vals, int stride, int maxNum) {
  SceneDev sv = s;
  void* args[] = {&sv, &n, &pos4, &dir4, &hits, &ids, &vals, &stride, &maxNum};
  return hipModuleLaunchKernel(p->kPoints, unsigned((n + 255) / 256), 1, 1, 256, 1, 1, 0, stream, args, nullptr);
}
