#!/bin/bash
# Host-side AddressSanitizer + UBSan pass over the C-ABI library (build container, no GPU needed: GPU ASan is not available on this
# pool, so the DEVICE code is built uninstrumented with -fno-gpu-sanitize).  Builds cf-nerf_amd/libvar_asan.so (W = 256 kernels only, to
# keep the build short) and runs the CPU ABI tests against it: every host path that does not need a device - configuration
# validation, parameter layout, operand packing plan, weight-gradient tile / block / split plans, the debug accessors, the error
# paths (NULL / bad arguments) - under the sanitizers.  Any report fails the script.
#   tools/asan_host.sh
set -euo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/cf-nerf_amd/build/asan"; mkdir -p "$O"
RT="$(dirname "$(hipcc --offload-arch=gfx950 -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)")"
[ -f "$RT/libclang_rt.asan-x86_64.so" ] || RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-result -Wno-unused-value -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
for f in cfnerf_fwd cfnerf_bwd cfnerf_tail cfnerf_abi; do
  hipcc $F "-DCFN_FOR_EACH_WIDTH(X)=X(256)" -c "$R/cf-nerf_amd/csrc/$f.hip" -o "$O/$f.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script="$R/cf-nerf_amd/csrc/cfnerf_exports.map" -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$R/cf-nerf_amd/libvar_asan.so" "$O"/cfnerf_fwd.o "$O"/cfnerf_bwd.o "$O"/cfnerf_tail.o "$O"/cfnerf_abi.o
# the host planners (operand packing, weight-gradient plan) are header-only code the tests reach through the test-hooks library: same flags
hipcc $F "-DCFN_FOR_EACH_WIDTH(X)=X(256)" -I"$R/cf-nerf_amd/csrc" -I"$R/include" -shared -shared-libsan "$R/tests/csrc/cfnerf_testhooks.hip" -o "$O/libcfnerf_testhooks_asan.so"
cd "$R"
LOG="$O/asan_run.log"
CFNERF_TESTHOOKS_LIB="$O/libcfnerf_testhooks_asan.so" CFNERF_LIB="$R/cf-nerf_amd/libvar_asan.so" LD_LIBRARY_PATH="$RT:${LD_LIBRARY_PATH:-}" LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" \
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_abi_cpu.py -q -p no:cacheprovider -s 2>&1 | tee "$LOG" | tail -3
if grep -q "runtime error\|AddressSanitizer" "$LOG"; then echo "SANITIZER REPORTS:"; grep -n "runtime error\|AddressSanitizer" "$LOG" | head; exit 1; fi
grep -q " passed" "$LOG" && ! grep -q " failed" "$LOG"
echo "host ASan + UBSan: clean"
