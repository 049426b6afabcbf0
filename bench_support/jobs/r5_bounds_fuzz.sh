#!/bin/bash
# The bounds-checked LAB build (ab_builds/libbounds.so = make EXTRA="-DSTRSIM_LAB -DSTRSIM_BOUNDS", bench_support/build_variants.sh)
# under the differential fuzzer with seeds the suite does not use -- every global and LDS address the kernels form is compared with
# the extents of its launch (csrc/strsim_bounds.h) and every row with the oracle -- then the thin-ABI parity suite and the plugin ABI
# on the same build.  bash bench_support/jobs/r5_bounds_fuzz.sh [seconds per seed = 75] -> gpurun_out/r5_bounds_fuzz.txt
SECS=${1:-75}
export STRSIM_AMD_LIB=$(pwd)/ab_builds/libbounds.so
OUT=gpurun_out/r5_bounds_fuzz.txt; mkdir -p gpurun_out; : > $OUT
for seed in 5004 7001 7002 7003 7004 7005 7006 7007 $(date +%s); do
  python tests/fuzz_gpu.py $SECS $seed 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT
done
python -m pytest tests/test_gpu_parity.py tests/test_plugin_abi_gpu.py tests/test_plugin_configs_gpu.py tests/test_gpu_hypothesis.py -m gpu -q 2>&1 | tail -3 | tee -a $OUT
python - <<'PY' 2>&1 | tee -a $OUT
import sys
sys.path[:0] = ["polars-strsim_amd", "tests"]
import ctypes as C
import strsim_amd as S
for unit in ("kernels", "codec"):
    f = getattr(S.lib(), "strsim_debug_bounds_" + unit)
    f.restype = C.c_int; f.argtypes = [C.c_void_p]
print("lab library:", S._lib.LIB_PATH)
PY
