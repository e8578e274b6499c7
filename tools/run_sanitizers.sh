#!/bin/bash
# CPU sanitizer job ("race detection / sanitizers"): AddressSanitizer + UBSan builds of
#   (1) the host side of libgpmpc_hip.so (same sources, sanitizers on the host compilation) driven by tools/abi_argcheck.cpp,
#   (2) the plain-C checker oracle/cport, under which the CPU test-suite parts that use it are run.
# Never on a GPU (GPU AddressSanitizer is not available on the pool).  Log: profiles/<round>/sanitizer_cpu.log
# Exit status: non-zero when a build step fails, the sanitized binary reports anything, or the sanitized tests fail.
set -u
set -o pipefail
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r04}
LOG=profiles/$ROUND/sanitizer_cpu.log
mkdir -p profiles/$ROUND
FAIL=0
{
echo "# tools/run_sanitizers.sh  $(date -u +%Y-%m-%dT%H:%MZ)  HEAD $(git rev-parse --short HEAD 2>/dev/null)"
echo "== (1) host side of the library, -fsanitize=address,undefined (-Xarch_host: host code of csrc/*.hip only)"
rm -f /tmp/abi_argcheck gaussian_process_mpc_amd/csrc/libgpmpc_hip_host_asan.so          # never run a stale binary / library
if ! make -C gaussian_process_mpc_amd/csrc -f Makefile.san -j8 asan-host > /tmp/asan_host_build.log 2>&1; then
    echo "BUILD FAILED (library)"; tail -20 /tmp/asan_host_build.log; FAIL=1
elif ! /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -shared-libsan tools/abi_argcheck.cpp \
        -Lgaussian_process_mpc_amd/csrc -lgpmpc_hip_host_asan -L/opt/rocm/lib -lamdhip64 -Wl,--allow-shlib-undefined \
        -Wl,-rpath,$PWD/gaussian_process_mpc_amd/csrc -o /tmp/abi_argcheck > /tmp/asan_link.log 2>&1; then
    echo "BUILD FAILED (abi_argcheck)"; tail -5 /tmp/asan_link.log; FAIL=1
else
    LD_LIBRARY_PATH=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)):/opt/rocm/lib:${LD_LIBRARY_PATH:-} \
        ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 /tmp/abi_argcheck > /tmp/abi_argcheck.log 2>&1
    RC=$?                                                  # the sanitized binary's own status (not a pipe's last stage)
    tail -40 /tmp/abi_argcheck.log
    echo "exit status: $RC"
    [ $RC -ne 0 ] && FAIL=1
fi
echo "== (2) oracle/cport under -fsanitize=address,undefined: CPU tests that drive the C checker"
rm -f oracle/cport/libgpmpc_cpu_asan.so
if ! make -C oracle/cport -f Makefile.san asan > /tmp/asan_cport_build.log 2>&1; then
    echo "BUILD FAILED (cport)"; tail -20 /tmp/asan_cport_build.log; FAIL=1
else
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
        GPMPC_CPORT_LIB=$PWD/oracle/cport/libgpmpc_cpu_asan.so OMP_NUM_THREADS=4 \
        python -m pytest tests/test_oracle_golden.py -q -m "not gpu" -k "cport or yardstick or g10" -p no:cacheprovider > /tmp/asan_pytest.log 2>&1
    RC=$?
    tail -15 /tmp/asan_pytest.log
    echo "pytest exit status: $RC"
    [ $RC -ne 0 ] && FAIL=1
fi
echo "== overall: $([ $FAIL -eq 0 ] && echo OK || echo FAILED)"
} 2>&1 | tee $LOG
grep -q "^== overall: OK" $LOG
