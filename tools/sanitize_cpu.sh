#!/bin/bash
# Host code of the boundary under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU box (SURVEY.md section 5:
# "host tests under ASan/UBSan"; GPU sanitizers are not available on this pool).
#   1. libeleven_hip_asan.so: every translation unit of csrc/ with -fsanitize=address,undefined for the HOST side
#      (-fno-gpu-sanitize keeps the gfx950 code objects as they are), objects in csrc/build_asan/;
#   2. the CPU test-suite (-m "not gpu": ABI validation, BVH builder, OOM path, CDF search, collective entry points, OBJ
#      ingest, host server sessions ...) against that library, with the ASan runtime preloaded into python;
#   3. the host server and the OBJ dump tool built with g++ -fsanitize=address,undefined and driven by the same tests.
# Usage: bash tools/sanitize_cpu.sh        (from the repo root; ~2 minutes)
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
CLANG_RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
ASAN_SO=$CLANG_RT/libclang_rt.asan-x86_64.so
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
make -C elevenrender_amd/csrc -j4 BUILD=build_asan OUT=../libeleven_hip_asan.so EXTRA="$SAN -fno-gpu-sanitize -shared-libsan" LDEXTRA="$SAN -shared-libsan"
# native drivers against the sanitized library, themselves sanitized (g++'s own runtimes)
for t in abi_smoke host_cornell; do
  /opt/rocm/lib/llvm/bin/clang++ $SAN -shared-libsan -O1 -std=c++17 tests/native/$t.cpp -o tests/native/${t}_asan -L elevenrender_amd -l:libeleven_hip_asan.so -Wl,-rpath,"$ROOT/elevenrender_amd" -Wl,-rpath,"$CLANG_RT"
done
/opt/rocm/lib/llvm/bin/clang++ $SAN -shared-libsan -O1 -std=c++17 tests/native/obj_dump.cpp -o tests/native/obj_dump -Wl,-rpath,"$CLANG_RT"
/opt/rocm/lib/llvm/bin/clang++ $SAN -shared-libsan -O1 -std=c++17 -pthread elevenrender_amd/host/eleven_server.cpp -o elevenrender_amd/host/eleven_server \
    -L elevenrender_amd -l:libeleven_hip_asan.so -Wl,-rpath,"$ROOT/elevenrender_amd" -Wl,-rpath,"$CLANG_RT"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export ELEVEN_HIP_LIB=$ROOT/elevenrender_amd/libeleven_hip_asan.so
set +e
LD_PRELOAD=$ASAN_SO python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/sanitize_cpu.log
rc=${PIPESTATUS[0]}
for t in abi_smoke host_cornell; do tests/native/${t}_asan > /tmp/${t}_asan.log 2>&1; echo "$t (no GPU: must fail loudly, not crash): rc=$? $(tail -n 1 /tmp/${t}_asan.log)"; done
grep -c "ERROR: AddressSanitizer\|runtime error:" /tmp/sanitize_cpu.log /tmp/abi_smoke_asan.log /tmp/host_cornell_asan.log
# leave the tree as the normal build expects it
rm -f tests/native/abi_smoke_asan tests/native/host_cornell_asan tests/native/obj_dump elevenrender_amd/host/eleven_server
exit $rc
