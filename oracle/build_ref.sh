#!/bin/bash
# build_ref.sh -- compile the REFERENCE's own kernel code for gfx950 into oracle/_ref/ (git-ignored, shipped to the GPU
# box as built binaries; the reference sources stay in /root/reference and never enter the repo).
#
#   oracle/_ref/ref_driver.hsaco : oracle/ref_driver.cl (our kernel entry points) + the reference inline functions of
#                                  hydra_drv/c*.h, compiled through their OpenCL branch (-D OCL_COMPILER)
#   oracle/_ref/trace.hsaco      : the reference's unmodified shaders/trace.cl (BVH4TraversalInstKernel, ComputeHit, ...)
#   oracle/_ref/material.hsaco, light.hsaco, mlt.hsaco, screen.hsaco : the reference's unmodified stage kernels of its own wavefront
#                                  layer (HitEnvOrLightKernel, Shade, NextBounce; LightSample; the MMLT kernels; ray generation / clear)
#
# Toolchain: the image's clang (ROCm LLVM) + ROCm device bitcode libraries; nothing is stubbed.  Options follow the
# reference's GetOCLShaderCompilerOptions (hydra_drv/GPUOCLLayer.cpp:877-898) except the three arithmetic relaxations
# (-cl-mad-enable, -cl-no-signed-zeros, -cl-denorms-are-zero) and with -ffp-contract=off, because the parity target is
# the CPU path (CPUExp_Integrators_PT), which is compiled without FMA contraction.
set -euo pipefail
REF=${REF:-/root/reference/hydra_drv}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
if [ ! -d "$REF/shaders" ]; then echo "build_ref.sh: $REF not present (GPU box): using prebuilt $OUT" ; exit 0; fi
mkdir -p "$OUT"
# -force-attribute ... optnone: ROCm 7.2's clang crashes (InstCombine, visitPHINode) while optimising the reference's
# MapSamplesToDisc (cglobals.h:1609-1652); the function is compiled unoptimised instead, its source is untouched.
FLAGS="-x cl -Xclang -finclude-default-header -cl-std=CL1.2 -target amdgcn-amd-amdhsa -mcpu=gfx950 -O2 \
 -mllvm -force-attribute=MapSamplesToDisc:noinline -mllvm -force-attribute=MapSamplesToDisc:optnone \
 -cl-single-precision-constant -cl-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -D OCL_COMPILER -D SHADOW_TRACE_COLORED_SHADOWS -D ENABLE_OPACITY_TEX -D ENABLE_BLINN \
 -I $REF -I $REF/shaders -w"
if [ "${1:-}" = texproc ]; then
  # build_ref.sh texproc <scene library> <name>: the reference's procedural-texture program of one scene library (shaders/texproc.cl + the library's functions, put
  # together by oracle/splice_texproc.py as RenderDriverRTE does at scene load) -> oracle/_ref/texproc_<name>.hsaco.  The spliced text lives in a scratch directory only.
  # -Wno-error=incompatible-pointer-types: texproc.cl:27 hands a float4 pointer to a parameter declared int4*, which this clang rejects by default (a diagnostic
  # level, the source is untouched).
  TMPD=$(mktemp -d)
  python3 "$HERE/splice_texproc.py" "$2" "$TMPD/texproc_generated.cl"
  $CLANG $FLAGS -Wno-error=incompatible-pointer-types "$TMPD/texproc_generated.cl" -o "$OUT/texproc_$3.hsaco"
  rm -rf "$TMPD"
  ls -la "$OUT/texproc_$3.hsaco"
  exit 0
fi
$CLANG $FLAGS "$HERE/ref_driver.cl" -o "$OUT/ref_driver.hsaco"
$CLANG $FLAGS "$REF/shaders/trace.cl" -o "$OUT/trace.hsaco"
for k in material light mlt screen; do $CLANG $FLAGS "$REF/shaders/$k.cl" -o "$OUT/$k.hsaco"; done
ls -la "$OUT"
