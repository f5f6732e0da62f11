#!/usr/bin/env bash
# Builds the REFERENCE's own kernels for gfx950, from the sources where they lie under /root/reference, into oracle/_ref/.
# Test infrastructure only (the checker of tests/test_gpu_reference_kernels.py); nothing of the product links or loads it.
#
#   oracle/_ref/libref_pn2.so      the seven *_cuda.cu files of utils/mm3d_pn2/ops/*/src (kernels + their
#                                  *_kernel_launcher functions; called through their C++ symbols with ctypes)
#   oracle/_ref/ref_emd.so         utils/metrics/EMD/emd_cuda.cu + emd.cpp            (the reference's pybind module)
#   oracle/_ref/ref_chamfer_3D.so  utils/metrics/CD/chamfer3D/chamfer3D.cu + chamfer_cuda.cpp
#
# How: the image's own tools only.  Each .cu goes through /opt/rocm/bin/hipify-perl (CUDA runtime names -> HIP runtime
# names; a text filter shipped with ROCm) into a temporary directory that is removed on exit, and is compiled by hipcc with
# its DEFAULT floating-point contraction (the counterpart of nvcc's default -fmad=true); the two .cpp files are compiled
# where they lie.  No header, library or tool is written or stood in for.  One edit is made in transit and only one:
# emd_cuda.cu defines its own `atomicMax(float*, float)` (a CAS loop, emd_cuda.cu:10-20), a name the HIP runtime headers
# also define -- the reference's function and its one call site (emd_cuda.cu:176) are renamed ref_atomicMax so that the
# REFERENCE's loop is the one that runs.  The PN2 .cpp shims (THC/THC.h, removed from PyTorch) are not needed: tests call
# the launchers those shims call.
# Only .so files are kept; translated sources never enter the repository or the GPU box.
set -euo pipefail
REF=${MVP_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF/utils/mm3d_pn2/ops" ] || { echo "no reference tree at $REF: keeping whatever is in $OUT"; exit 0; }
HIPCC=/opt/rocm/bin/hipcc
HIPIFY=/opt/rocm/bin/hipify-perl
ARCH=${MVP_ARCH:-gfx950}
CONTRACT=${MVP_REF_CONTRACT:-}          # e.g. -ffp-contract=off for the sensitivity build (suffix via MVP_REF_SUFFIX)
SUFFIX=${MVP_REF_SUFFIX:-}
T=$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')
PYINC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
INC="-I$T/include -I$T/include/torch/csrc/api/include -I$PYINC -I/opt/rocm/include"
DEFS="-D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DHIPBLAS_V2 -DCUDA_HAS_FP16=1 -D__HIP_NO_HALF_OPERATORS__=1 -D__HIP_NO_HALF_CONVERSIONS__=1"
LIBS="-L$T/lib -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip -ltorch_python -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$T/lib"
mkdir -p "$OUT"
stamp=$OUT/.built_from$SUFFIX
want=$(cd "$REF" && cat utils/mm3d_pn2/ops/*/src/*_cuda.cu utils/metrics/EMD/emd_cuda.cu utils/metrics/EMD/emd.cpp \
        utils/metrics/CD/chamfer3D/chamfer3D.cu utils/metrics/CD/chamfer3D/chamfer_cuda.cpp "$HERE/build_ref_gpu.sh" | md5sum | cut -d' ' -f1)
if [ -f "$stamp" ] && [ "$(cat "$stamp")" = "$want$CONTRACT" ] && [ -f "$OUT/libref_pn2$SUFFIX.so" ] && [ -f "$OUT/ref_emd$SUFFIX.so" ] && [ -f "$OUT/ref_chamfer_3D$SUFFIX.so" ]; then
  echo "oracle/_ref is current"; exit 0
fi
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT

# (every translation + compilation below is independent of the others: run them side by side, link at the end)
pids=()
# --- PointNet++ kernels: no CUDA header is included by these files; hipcc needs nothing but the renamed runtime calls
objs=()
for f in furthest_point_sample/src/furthest_point_sample_cuda.cu knn/src/knn_cuda.cu ball_query/src/ball_query_cuda.cu \
         interpolate/src/three_nn_cuda.cu interpolate/src/three_interpolate_cuda.cu gather_points/src/gather_points_cuda.cu \
         group_points/src/group_points_cuda.cu; do
  n=$(basename "$f" .cu)
  "$HIPIFY" "$REF/utils/mm3d_pn2/ops/$f" > "$TMP/$n.hip" 2>/dev/null
  "$HIPCC" --offload-arch=$ARCH -O2 -fPIC -w $CONTRACT -c "$TMP/$n.hip" -o "$TMP/$n.o" & pids+=($!)
  objs+=("$TMP/$n.o")
done

# --- EMD and Chamfer: kernels + the reference's own pybind shims (at::Tensor in, raw pointers out)
"$HIPIFY" "$REF/utils/metrics/EMD/emd_cuda.cu" 2>/dev/null | sed 's/\batomicMax\b/ref_atomicMax/g' > "$TMP/emd_cuda.hip"
"$HIPCC" --offload-arch=$ARCH $INC $DEFS -std=c++17 -O2 -fPIC -w $CONTRACT -c "$TMP/emd_cuda.hip" -o "$TMP/emd_cuda.o" & pids+=($!)
g++ $INC $DEFS -std=c++17 -O2 -fPIC -w -DTORCH_EXTENSION_NAME=ref_emd$SUFFIX -DTORCH_API_INCLUDE_EXTENSION_H \
    -c "$REF/utils/metrics/EMD/emd.cpp" -o "$TMP/emd_bind.o" & pids+=($!)
"$HIPIFY" "$REF/utils/metrics/CD/chamfer3D/chamfer3D.cu" > "$TMP/chamfer3D.hip" 2>/dev/null
"$HIPCC" --offload-arch=$ARCH $INC $DEFS -std=c++17 -O2 -fPIC -w $CONTRACT -c "$TMP/chamfer3D.hip" -o "$TMP/chamfer3D.o" & pids+=($!)
g++ $INC $DEFS -std=c++17 -O2 -fPIC -w -DTORCH_EXTENSION_NAME=ref_chamfer_3D$SUFFIX -DTORCH_API_INCLUDE_EXTENSION_H \
    -c "$REF/utils/metrics/CD/chamfer3D/chamfer_cuda.cpp" -o "$TMP/chamfer_bind.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done

"$HIPCC" -shared -fPIC "${objs[@]}" -L/opt/rocm/lib -lamdhip64 -o "$OUT/libref_pn2$SUFFIX.so"
"$HIPCC" -shared -fPIC "$TMP/emd_cuda.o" "$TMP/emd_bind.o" $LIBS -o "$OUT/ref_emd$SUFFIX.so"
"$HIPCC" -shared -fPIC "$TMP/chamfer3D.o" "$TMP/chamfer_bind.o" $LIBS -o "$OUT/ref_chamfer_3D$SUFFIX.so"

echo "$want$CONTRACT" > "$stamp"
ls -la "$OUT"
