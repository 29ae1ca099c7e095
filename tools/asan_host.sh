#!/bin/bash
# The host side of the library under AddressSanitizer + UBSan, on the CPU (no GPU: the step controller against its Python twin with the
# oracle as planner, and the MATLAB marshalling).  step_controller.cpp and matlab_marshal.cpp are rebuilt instrumented and linked with the
# regular objects of `make -C p-dmpc_amd/csrc`; the tests load that library through PDMPC_LIB.
set -e
cd "$(dirname "$0")/.."
make -C p-dmpc_amd/csrc > /dev/null
mkdir -p build/asan
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude"
for f in step_controller matlab_marshal; do $CLANG $FLAGS -c -o build/asan/$f.o p-dmpc_amd/csrc/$f.cpp; done
OBJ=$(ls build/obj/*.hip.o build/obj/api.cpp.o build/obj/group.cpp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o build/asan/libpdmpc_hip_asan.so $OBJ build/asan/step_controller.o build/asan/matlab_marshal.o -ldl
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT PDMPC_LIB=$PWD/build/asan/libpdmpc_hip_asan.so \
    python -m pytest tests/test_native_controller.py tests/test_oracle_producers.py tests/test_matlab_marshal.py -x -q "$@"
