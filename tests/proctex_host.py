"""The scene's procedural texture functions built for the HOST, to feed the CPU oracle (test infrastructure).

The oracle restates the consumer of the per-ray lists (readProcTex in sample2DExt); the functions themselves are user code of the scene library.  For whole-frame
comparisons they are compiled here with clang for x86 inside the same frame the device build uses (hydracore_amd/csrc/hk_proctex_rt.h under HK_HOST_EMU, like
tests/emu/): the program text the front end produced is cut at the reference's markers exactly as hydra_proctex.hip does it, and the result exports one C entry the
oracle calls per hit (orc_set_proctex_eval)."""
import ctypes as C
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hydracore_amd", "csrc")
OUT = os.path.join(ROOT, "build", "proctex_host")


def _cut(text, begin, ends):
    b = text.index(begin)
    start = text.index("\n", b) + 1
    stop = min(text.index(e, start) for e in ends if e in text[start:])
    return text[start:text.rindex("\n", 0, stop)]


def build(program_text):
    """-> path of a shared object exporting proctex_eval(user, surf19, head, view3, count*, ids16*, vals48*); user = {globals*, texStorage*}"""
    user = _cut(program_text, "#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:", ["#HK_END_OF_PROCEDURAL_TEXTURES", "const int findArgDataOffsetInTable"])
    calls = _cut(program_text, "#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:", ["#HK_END_OF_PROCEDURAL_TEXTURES_EVAL", "// BREAK SHADER CACHE AT:", "// (5) take what we need"])
    frame = open(os.path.join(CSRC, "hk_proctex_rt.h")).read()
    frame = frame.replace("//#HK_PROCTEX_USER_CODE", user).replace("//#HK_PROCTEX_EVAL_CODE", calls)
    src = """#define HK_HOST_EMU 1
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime.h>
#include "%(csrc)s/hk_common.h"
#include "%(csrc)s/hk_trace.h"
#include "%(csrc)s/hk_shading.h"
%(frame)s
struct ProcTexHostUser { const int* globals; const int* texStorage; };
extern "C" void proctex_eval(void* user, const float* surf19, const float* head, const float* view3, int* count, int* ids16, float* vals48) {
  const ProcTexHostUser* u = static_cast<const ProcTexHostUser*>(user);
  SceneDev s = {};
  s.globals = u->globals; s.hdr = u->globals; s.texStorage = reinterpret_cast<const int4*>(u->texStorage);
  s.texTable = u->globals + u->globals[HG_TEX_TABLE_OFFS]; s.srgbLut = nullptr; s.ptlSlot = -1;
  hk_user::SurfaceInfo si;
  si.wp = hk_user::make_float3(surf19[0], surf19[1], surf19[2]);   si.lp = hk_user::make_float3(surf19[3], surf19[4], surf19[5]);
  si.n = hk_user::make_float3(surf19[6], surf19[7], surf19[8]);    si.tg = hk_user::make_float3(surf19[9], surf19[10], surf19[11]);
  si.bn = hk_user::make_float3(surf19[12], surf19[13], surf19[14]); si.tc0 = hk_user::make_float2(surf19[15], surf19[16]);
  si.ao = surf19[17]; si.ao2 = surf19[18];
  hk_user::ProcTextureList ptl;
  ptl.currMaxProcTex = 0;
  hk_user::evalAll(head, &si, hk_user::make_float3(view3[0], view3[1], view3[2]), &s, ptl);
  *count = ptl.currMaxProcTex;
  for (int k = 0; k < ptl.currMaxProcTex && k < 16; k++) { ids16[k] = ptl.id_f4[k]; vals48[3 * k] = ptl.fdata4[k].x; vals48[3 * k + 1] = ptl.fdata4[k].y; vals48[3 * k + 2] = ptl.fdata4[k].z; }
}
""" % dict(csrc=CSRC, frame=frame)
    os.makedirs(OUT, exist_ok=True)
    tag = hashlib.sha1(src.encode()).hexdigest()[:16]
    so, cpp = os.path.join(OUT, "proctex_%s.so" % tag), os.path.join(OUT, "proctex_%s.cpp" % tag)
    if not os.path.exists(so):
        with open(cpp, "w") as f:
            f.write(src)
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-ffp-contract=off", "-w", "-I/opt/rocm/include", "-fPIC", "-shared", cpp, "-o", so])
    return so


class HostUser(C.Structure):
    _fields_ = [("globals", C.c_void_p), ("texStorage", C.c_void_p)]


def attach(orc, program_text, b):
    """build the scene's functions for the host and hand them to the oracle: every hit on a material with procedural textures gets its list from them"""
    lib = C.CDLL(build(program_text))
    user = HostUser(b["globals"].ctypes.data, b["textures"].ctypes.data)
    orc._proctex_keep = (lib, user, b["globals"], b["textures"])
    orc.lib.orc_set_proctex_eval.argtypes = [C.c_void_p, C.c_void_p]
    orc.lib.orc_set_proctex_eval.restype = None
    orc.lib.orc_set_proctex_eval(C.cast(lib.proctex_eval, C.c_void_p), C.cast(C.pointer(user), C.c_void_p))


def detach(orc):
    orc.lib.orc_set_proctex_eval(None, None)
