#!/usr/bin/env python3
"""splice_texproc.py <scene library> <out.cl> -- the REFERENCE's procedural-texture program for one scene library, put together the way its render driver does it at
scene load (RenderDriverRTE::BeginTexturesUpdate / UpdateImageProc / EndTexturesUpdate, hydra_drv/RenderDriverRTE_ProcTex.cpp:446-629): the reference's own
shaders/texproc.cl, read where it lies under /root/reference, with the library's data/proctex_*.c after its '#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:' line and one generated
call per <texture type="proc"> after '#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:'.  Test infrastructure (oracle/): the output holds reference text, so it goes to a
scratch path (oracle/build_ref.sh texproc compiles it into oracle/_ref/ and nothing else keeps it)."""
import os
import re
import sys
import xml.etree.ElementTree as ET

REF = os.environ.get("REF", "/root/reference/hydra_drv")
TAIL_DECL = " __global const float4* restrict in_texStorage1, __global const EngineGlobals* restrict in_globals, const float3 hr_viewVectorHack"   # UpdateImageProc :612-613
TAIL_CALL = "in_texStorage1, in_globals, hr_viewVectorHack"                                                                                       # :588


def main(lib, out):
    text = open(os.path.join(lib, "statex_00001.xml")).read()
    root = ET.fromstring("<root>" + re.sub(r"<\?xml[^>]*\?>", "", text) + "</root>")
    procs = {}
    for t in root.find("textures_lib").findall("texture"):
        if t.get("type") != "proc":
            continue
        code = t.find("code")
        procs[int(t.get("id"))] = (code.find("generated").find("call").text.replace("_PROCTEXTAILTAG_", TAIL_CALL),
                                   open(os.path.join(lib, code.get("loc"))).read().replace("_PROCTEXTAILTAG_", TAIL_DECL))
    src = open(os.path.join(REF, "shaders", "texproc.cl")).read().split("\n")
    res, i = [], 0
    while i < len(src):                                   # BeginTexturesUpdate: the file up to the first marker
        res.append(src[i])
        i += 1
        if "#PUT_YOUR_PROCEDURAL_TEXTURES_HERE:" in src[i - 1]:
            break
    res.append("")
    for tid in sorted(procs):                             # UpdateImageProc: the functions of every procedural texture
        res.extend(procs[tid][1].split("\n"))
    while i < len(src):                                   # EndTexturesUpdate: the rest, with the calls after the second marker
        res.append(src[i])
        if "#PUT_YOUR_PROCEDURAL_TEXTURES_EVAL_HERE:" in src[i]:
            res += ["", "    int counter = 0; "]
            for tid in sorted(procs):
                res += ["    if(materialHeadHaveTargetProcTex(pHitMaterial,%d) && counter < MAXPROCTEX)" % tid, "    {",
                        "      __global const float* stack = fdata + findArgDataOffsetInTable(%d, table);" % tid,
                        "      ptl.fdata4[counter] = to_float3(%s);" % procs[tid][0],
                        "      ptl.id_f4 [counter] = %d;" % tid, "      counter++;", "    }", ""]
            res.append("    ptl.currMaxProcTex = counter;")
        i += 1
    with open(out, "w") as f:
        f.write("\n".join(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
