#!/bin/bash
# CPU sanitizer runs of the oracle and of the plugin layer's host side (never on the GPU box: GPU sanitizers are not available
# on the pool, and nothing here touches a device).  From the repo root:   bash tests/run_sanitizers.sh [out-dir]
#   1. oracle/strsim_oracle.c under ASan + UBSan: every reference vector and the README table (tests/test_oracle_golden.py)
#   2. csrc/polars_plugin.cpp (test-hooks build: packers, validity builder, thread pool, input ownership) under ASan + UBSan and
#      under TSan, driven by tests/cpu_harness/plugin_sanitize_driver.cpp: several caller threads, each call fanning out over the
#      packing pool
#   3. the Python-driven packer tests (tests/test_plugin_packing_cpu.py) and the ABI ownership test against the ASan build
# Logs: <out-dir>/sanitize_*.txt (default profiles/; committed as profiles/r6_sanitize_*.txt).  Exit code 0 = every run clean.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=${1:-profiles}; mkdir -p "$OUT"
B=${TMPDIR:-/tmp}/strsim_sanitize; rm -rf "$B"; mkdir -p "$B"
PKG=polars-strsim_amd; LIBDIR=$ROOT/$PKG/polars_strsim
make -s -C $PKG || exit 1
RT=$(dirname "$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)")
FAIL=0
COMMON="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -Iinclude -I$PKG/csrc -DSTRSIM_TEST_HOOKS -fno-gpu-sanitize"

# ---- 1. the oracle
gcc -O1 -g -ffp-contract=off -fPIC -std=c11 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared -pthread \
    -o $B/libstrsim_oracle.so oracle/strsim_oracle.c || FAIL=1
( export STRSIM_ORACLE_LIB=$B/libstrsim_oracle.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  python -m pytest tests/test_oracle_golden.py -x -q ) > "$OUT/sanitize_oracle_asan_ubsan.txt" 2>&1 || FAIL=1
tail -2 "$OUT/sanitize_oracle_asan_ubsan.txt"

# ---- 2. the plugin layer's host side, C++ driver
for san in address,undefined thread; do
  tag=$([ $san = thread ] && echo tsan || echo asan_ubsan)
  hipcc $COMMON -fsanitize=$san -fno-sanitize-recover=undefined -shared -x hip $PKG/csrc/polars_plugin.cpp -o $B/libhooks_$tag.so \
        -L$LIBDIR -lpolars_strsim_amd -Wl,-rpath,$LIBDIR 2> $B/build_$tag.log || { cat $B/build_$tag.log; FAIL=1; }
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fno-omit-frame-pointer -Iinclude -fsanitize=$san -fno-sanitize-recover=undefined tests/cpu_harness/plugin_sanitize_driver.cpp \
        -o $B/driver_$tag $B/libhooks_$tag.so -L$LIBDIR -lpolars_strsim_amd -Wl,-rpath,$B -Wl,-rpath,$LIBDIR -lpthread 2>> $B/build_$tag.log || { cat $B/build_$tag.log; FAIL=1; }
  ( export ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 LSAN_OPTIONS=suppressions=$ROOT/tests/cpu_harness/lsan.supp
    $B/driver_$tag 6 50 ) > "$OUT/sanitize_plugin_$tag.txt" 2>&1 || FAIL=1
  grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" "$OUT/sanitize_plugin_$tag.txt" | sed "s/^/$tag reports: /"
  tail -1 "$OUT/sanitize_plugin_$tag.txt"
done

# ---- 3. the Python-driven packer tests against the ASan + UBSan build (python itself is not instrumented: the runtime is preloaded)
( export STRSIM_TESTHOOKS_LIB=$B/libhooks_asan_ubsan.so LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
  python -m pytest tests/test_plugin_packing_cpu.py -x -q ) > "$OUT/sanitize_plugin_pytest_asan_ubsan.txt" 2>&1 || FAIL=1
tail -2 "$OUT/sanitize_plugin_pytest_asan_ubsan.txt"
[ $FAIL = 0 ] && echo "sanitizers: all clean" || echo "sanitizers: FAILURES (see $OUT/sanitize_*.txt)"
exit $FAIL
