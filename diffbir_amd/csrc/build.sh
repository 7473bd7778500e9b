#!/bin/sh
# Build libdbir_hip.so (gfx950) in-tree. Usage: sh diffbir_amd/csrc/build.sh   (DBIR_DIAG=1 adds the GEMM diagnostics)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
if [ -n "$DBIR_DIAG" ]; then FLAGS="$FLAGS -DDBIR_DIAG"; fi
SRCS="api gemm gemm_glds gemm_halo gemm_pers gemm_8p gemm_rs attention norm elementwise swin clip xformer xformer2 plan"
mkdir -p build
# objects built with a different flag set (e.g. a DBIR_DIAG build) must not be linked into this one
if [ "$(cat build/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f build/*.o; echo "$FLAGS" > build/.flags; fi
PIDS=""
for f in $SRCS; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_epilogue.h -nt build/$f.o ] || [ ../../include/dbir.h -nt build/$f.o ] || { [ $f = plan ] && [ plan_dispatch.inc -nt build/$f.o ]; }; then
    rm -f build/$f.o
    if [ $f = xformer2 ]; then STD="-std=c++20"; else STD=""; fi   # (xformer2.hip: templated lambdas)
    $HIPCC $FLAGS $STD -c $f.hip -o build/$f.o &
    PIDS="$PIDS $!"
  fi
done
for p in $PIDS; do wait $p || { echo "build.sh: a hipcc job failed" >&2; exit 1; }; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o ../libdbir_hip.so
echo "built $(pwd)/../libdbir_hip.so"
