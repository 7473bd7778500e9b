#!/bin/sh
# Build libdbir_hip.so (gfx950) in-tree. Usage: sh diffbir_amd/csrc/build.sh   (DBIR_DIAG=1 adds the GEMM diagnostics)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
if [ -n "$DBIR_DIAG" ]; then FLAGS="$FLAGS -DDBIR_DIAG"; rm -f build/gemm_glds.o build/gemm_ph.o; fi
mkdir -p build
for f in api gemm gemm_glds gemm_ph attention norm elementwise swin; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ ../../include/dbir.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
  fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC build/api.o build/gemm.o build/gemm_glds.o build/gemm_ph.o build/attention.o build/norm.o build/elementwise.o build/swin.o -o ../libdbir_hip.so
echo "built $(pwd)/../libdbir_hip.so"
