#!/bin/bash
# AddressSanitizer + UBSan run of the CPU tier: the oracle and the test-only host build of the product's
# __host__ __device__ math and LM drivers (GPU sanitizers are not available on the pool).  Usage: tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/cba_asan
mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -Wl,-Bsymbolic"
(cd tests/cpu_backend && g++ $FLAGS -o $OUT/libhostmath.so hostmath_capi.cpp backend_cpu.cpp handeye_cpu.cpp planarpose_cpu.cpp homography_cpu.cpp semidlt_cpu.cpp seed_cpu.cpp)
(cd oracle && g++ $FLAGS -o $OUT/liboracle.so oracle_capi.cpp)
CBA_TEST_LIBDIR=$OUT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
python -m pytest tests/test_host_logic.py tests/test_oracle_kat.py tests/test_multirank_gloo.py -x -q -m "not gpu" -k "not two_ranks_match" -p no:cacheprovider
