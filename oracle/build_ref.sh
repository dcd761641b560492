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
$CLANG $FLAGS "$HERE/ref_driver.cl" -o "$OUT/ref_driver.hsaco"
$CLANG $FLAGS "$REF/shaders/trace.cl" -o "$OUT/trace.hsaco"
for k in material light mlt screen; do $CLANG $FLAGS "$REF/shaders/$k.cl" -o "$OUT/$k.hsaco"; done
ls -la "$OUT"
