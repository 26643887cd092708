#!/bin/bash
# Host code of the library under AddressSanitizer (CPU only: GPU ASan is not available on this pool).
# Builds an instrumented copy in /tmp and runs the tests that need no GPU against it.
set -e
OUT=/tmp/asan_build; mkdir -p $OUT/obj $OUT/lib
for f in device.cpp filters.cpp logging.cpp pointcloud.cpp stubs.cpp synthetic.cpp kernels_basic.hip kernels_sor.hip kernels_voxel.hip; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -DCWIPC_VERSION=amd-gfx950-asan \
    -Iinclude -Icwipc_util_amd/csrc -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -x hip -c cwipc_util_amd/csrc/$f -o $OUT/obj/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -o $OUT/lib/libcwipc_util.so $OUT/obj/*.o
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 CWIPC_LIBRARY_DIR=$OUT/lib python -m pytest tests/test_boundary.py tests/test_multigpu_gloo.py -q 2>&1 | tee $OUT/log.txt | tail -3
echo "AddressSanitizer reports: $(grep -c AddressSanitizer $OUT/log.txt || true)"
