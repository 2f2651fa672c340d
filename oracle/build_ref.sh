#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (oracle).  Builds oracle/_ref/libref_rtxadvect.so: the
# reference's OWN hot-path device functions compiled for the host CPU, from the
# sources where they lie under $CPF_REFERENCE (default /root/reference).
#
# Nothing from the reference is copied into this repository: the function
# bodies are spliced (by line range) into a scratch file
# in a temp dir that is deleted afterwards; only the .so lands in oracle/_ref/
# (git-ignored).  The only things this recipe supplies are the nvcc language
# built-ins g++ does not have (__device__/__global__ keywords, double4,
# threadIdx/blockIdx) -- no stand-ins for any header or library of the
# algorithm: cuda/HostTetMesh.h and the OWL vec headers are #included in place.
# cuRAND (Brownian kernel) and OptiX (initial BVH query) are NOT buildable here
# and are deliberately left out (DESIGN.md "parity unpinned" items).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${CPF_REFERENCE:-/root/reference}"
RTX="$REF/third_party/RTXAdvect"
OUT="$HERE/_ref"
if [ ! -d "$RTX" ]; then
  echo "build_ref.sh: reference tree not found at $REF (fine on the GPU box: prebuilt .so is used)" >&2
  exit 3
fi
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT

# sanity: the pinned line ranges must start where we expect (reference @ v1)
expect() { # file line needle
  sed -n "${2}p" "$1" | grep -q -- "$3" || { echo "build_ref.sh: $1:$2 does not contain '$3' -- reference changed?" >&2; exit 4; }
}
expect "$RTX/cuda/DeviceTetMesh.cuh" 82  "double det(const vec3d A"
expect "$RTX/cuda/DeviceTetMesh.cuh" 108 "tetBaryCoord"
expect "$RTX/cuda/DeviceTetMesh.cuh" 193 "triNorm"
expect "$RTX/query/ConvexQuery.cu"   32  "traceIntet"
expect "$RTX/query/ConvexQuery.cu"   136 "particleLocator"
expect "$RTX/query/ConvexQuery.cu"   239 "reflectInTet"
expect "$RTX/query/ConvexQuery.cu"   321 "convexReflector"
expect "$RTX/cuda/particles.cu"      78  "initParticlesKernel"
expect "$RTX/cuda/particles.cu"      245 "particleAdvectKernel(Particle"
expect "$RTX/cuda/particles.cu"      317 "particleAdvectKernelTetVel"
expect "$RTX/cuda/particles.cu"      660 "particleMoveKernel"
expect "$RTX/query/RTQuery.cu"       35  "baryTetSearch"
expect "$RTX/query/RTQuery.cu"       109 "RTreflection"
expect "$RTX/query/RTQuery.cu"       221 "baryQueryDisp"
expect "$RTX/cuda/utils.cpp"         144 "writeParticles2VTU"
expect "$RTX/cuda/utils.cpp"         172 "Output particle into VTU file"
expect "$RTX/cuda/utils.cpp"         217 "h_ctetIDs(numParticles)"
expect "$RTX/cuda/utils.cpp"         282 "fclose(fp)"
expect "$RTX/cuda/utils.cpp"         7   "void addToTrajectories"
expect "$RTX/cuda/utils.cpp"         20  "for (int i = 0; i < numParticles; i++)"
expect "$RTX/cuda/utils.cpp"         30  "void saveTrajectories"
expect "$RTX/cuda/utils.cpp"         49  "void writeStreamline2VTK"
expect "$RTX/cuda/utils.cpp"         96  "void writeParticles2OBJ"

{
  sed -n '82,104p;108,156p;193,199p' "$RTX/cuda/DeviceTetMesh.cuh"   # det, tetBaryCoord, triNorm
  sed -n '30,33p'                     "$RTX/query/RTQuery.cu"          # SearchInfo
  sed -n '32,131p;135,216p'           "$RTX/query/ConvexQuery.cu"      # traceIntet, particleLocator
  sed -n '239,317p;320,436p'          "$RTX/query/ConvexQuery.cu"      # reflectInTet, convexReflector
  sed -n '78,97p'                     "$RTX/cuda/particles.cu"         # initParticlesKernel
  sed -n '244,313p'                   "$RTX/cuda/particles.cu"         # particleAdvectKernel ("VertexVelocity")
  sed -n '316,373p'                   "$RTX/cuda/particles.cu"         # particleAdvectKernelTetVel
  sed -n '659,704p'                   "$RTX/cuda/particles.cu"         # particleMoveKernel (disp)
  sed -n '35,186p;189,248p'           "$RTX/query/RTQuery.cu"          # bary search, RT reflection, baryQuery(Disp)
} > "$TMP/ref_extract.inc"
# the body of writeParticles2VTU (host code: the on-disk format).  Its three cudaMemcpy device-to-host copies
# (:151-170 and :218-221) are left out -- the driver passes host arrays -- everything that formats is the
# reference's own text.
# the trajectory collection and its two writers (host code): the sampling loop of addToTrajectories (:20-27, i.e. without its
# cudaMemcpy -- the driver passes a host array) and saveTrajectories / writeStreamline2VTK whole (:30-47, :49-94)
sed -n '20,27p'  "$RTX/cuda/utils.cpp" > "$TMP/ref_traj_add.inc"
sed -n '30,47p;49,94p' "$RTX/cuda/utils.cpp" > "$TMP/ref_traj_writers.inc"
sed -n '172,217p' "$RTX/cuda/utils.cpp" > "$TMP/ref_vtu_head.inc"
sed -n '222,282p' "$RTX/cuda/utils.cpp" > "$TMP/ref_vtu_tail.inc"

g++ -std=c++14 -O2 -fPIC -shared -fopenmp -ffp-contract=off -w \
    -I"$RTX" -I"$RTX/owl/owl/include" -I"$TMP" \
    "$HERE/ref_driver.cpp" -o "$OUT/libref_rtxadvect.so"
echo "built $OUT/libref_rtxadvect.so"

# The same splices once more, CONTRACTING: nvcc fuses a*b+c into one fma by default (--fmad=true) and the reference's CMake
# does not switch that off, so the binary its authors ran rounds differently from the strict build above.  g++'s
# -ffp-contract=fast -mfma fuses in the same spirit (every multiply whose only use is an add, also across statements after
# inlining) but not necessarily at the same places as nvcc: this is A contracting build of the reference's own functions,
# used to MEASURE how far contraction moves particles (tests/test_fma_sensitivity.py, tests/golden/*_fma.npz), not a
# bit-replica of the CUDA binary.
g++ -std=c++14 -O2 -fPIC -shared -fopenmp -ffp-contract=fast -mfma -w \
    -I"$RTX" -I"$RTX/owl/owl/include" -I"$TMP" \
    "$HERE/ref_driver.cpp" -o "$OUT/libref_rtxadvect_fma.so"
echo "built $OUT/libref_rtxadvect_fma.so"
