#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on the pool):
#   * the product's host layer (terra_amd/csrc/scene_host.cpp, tree_build.cpp, multi_gpu.cpp: scene API, both tree builders and the 4-wide binary16 conversion, upload bookkeeping)
#     compiled with g++ against host stand-ins for this repo's own kernel launchers (stub_launchers.cpp);
#   * the oracle.
# Drives them over Cornell / textured / sphere / 97k-hall / random soups / empty and 1-triangle scenes, all tree modes.
# Usage: tools/sanitize/run.sh      (build container; outputs under terra_amd/build_asan/, git-ignored)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/terra_amd/build_asan; mkdir -p $OUT
g++ -std=c++17 -O1 -g -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/terra_amd/csrc -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -w \
    -shared -o $OUT/libterra_host_asan.so $ROOT/terra_amd/csrc/scene_host.cpp $ROOT/terra_amd/csrc/tree_build.cpp $ROOT/terra_amd/csrc/multi_gpu.cpp $ROOT/tools/sanitize/stub_launchers.cpp -ldl -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
gcc -std=gnu11 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include -shared -o $OUT/liboracle_asan.so $ROOT/oracle/terra_oracle.c -lm
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
python3 $ROOT/tools/sanitize/drive_host.py $OUT/libterra_host_asan.so 2>&1 | grep -v "no HIP device" | tail -5
python3 $ROOT/tools/sanitize/drive_oracle.py $OUT/liboracle_asan.so 2>&1 | tail -5
# ThreadSanitizer over the same host layer (the fast tree's host builder is threaded)
unset LD_PRELOAD ASAN_OPTIONS
g++ -std=c++17 -O1 -g -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/terra_amd/csrc -fsanitize=thread -fno-omit-frame-pointer -ffp-contract=off -w \
    -shared -o $OUT/libterra_host_tsan.so $ROOT/terra_amd/csrc/scene_host.cpp $ROOT/terra_amd/csrc/tree_build.cpp $ROOT/terra_amd/csrc/multi_gpu.cpp $ROOT/tools/sanitize/stub_launchers.cpp -ldl -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0" python3 $ROOT/tools/sanitize/drive_host.py $OUT/libterra_host_tsan.so 2>&1 | grep -v "no HIP device" > $OUT/tsan.log || true
echo "tsan warnings: $(grep -c 'WARNING: ThreadSanitizer' $OUT/tsan.log)"
