#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU build of the DEVICE SOURCE (tests/native/hipemu: the kernels and host drivers of
# openmvg_amd/csrc compiled for the host, the threads of a workgroup as fibers) - GPU sanitizers are not available on this pool.
#   bash tools/sanitize_cpu.sh [pytest selection ...]        default: the emulation test files
# Reports go to /tmp/mvgx_san.<pid> (ASAN_OPTIONS=log_path: pytest captures fd 2); the script prints their number and the first lines.
set -u
cd "$(dirname "$0")/.."
export MVGX_EMU_SANITIZE=1
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
rm -f /tmp/mvgx_san.*
SEL=${*:-tests/test_matching_emu_cpu.py tests/test_ba_emu_cpu.py tests/test_ba_update.py tests/test_hamming_cpu.py tests/test_l2f_cpu.py tests/test_l2u8_cpu.py tests/test_geofilter_cpu.py tests/test_guided_matching.py tests/test_cascade.py}
python -c "from tests import _emu; _emu.build(); _emu.build_match()" || exit 1
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:log_path=/tmp/mvgx_san UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/mvgx_san \
  python -m pytest $SEL -q -m "not gpu" -k "not adapter" -n ${SAN_JOBS:-6} -p no:cacheprovider 2>&1 | tail -5   # (the adapter tests dlopen the reference with RTLD_DEEPBIND, which the sanitizer runtime refuses)
n=$(ls /tmp/mvgx_san.* 2>/dev/null | wc -l)
echo "sanitizer report files: $n"
[ "$n" -gt 0 ] && grep -h "ERROR\|runtime error" /tmp/mvgx_san.* | sort | uniq -c | sort -rn | head -40
exit 0
