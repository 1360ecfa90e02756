#!/bin/bash
# CPU sanitizer run (SURVEY §5 "race detection / sanitizers"; GPU ASan / XNACK are not available on this pool, so this is the
# CPU build only): the oracle and the C-ABI shim rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle
# asan` -> oracle/_build_asan/) and the CPU tests that drive them — known-answer tests, the ABI scenario through the brl_*
# symbols, header <-> binding checks, the two-rank gloo tests — run in a python that preloads the sanitizer runtimes.
#   usage: bash scripts/cpu_sanitize.sh [out_file]       (default: profiles/r05/r05_cpu_sanitize.txt)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05/r05_cpu_sanitize.txt}
mkdir -p "$(dirname "$OUT")"
make -C oracle --no-print-directory asan || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
TESTS="tests/test_oracle_kat.py tests/test_capi_cpu.py tests/test_abi_either_library.py tests/test_dist_gloo.py"
{
  echo "# scripts/cpu_sanitize.sh — $(gcc --version | head -1)"
  echo "# flags: $(make -C oracle --no-print-directory -pn asan 2>/dev/null | grep '^ASAN_FLAGS' | head -1)"
  echo "# LD_PRELOAD=$ASAN:$UBSAN BRL_ORACLE_BUILD=asan python -m pytest $TESTS -q -m 'not gpu'"
  echo "# detect_leaks=0: CPython itself never frees its interned objects; every other ASan check and all of UBSan are on,"
  echo "# halt_on_error=1 / -fno-sanitize-recover: the first report aborts the run (a clean log == no report)"
} > "$OUT"
# (verify_asan_link_order=0: the runtime is preloaded into an uninstrumented python, which is the supported way to sanitize a dlopen'ed library)
LD_PRELOAD="$ASAN:$UBSAN" BRL_ORACLE_BUILD=asan \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  timeout -k 10 1500 python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider 2>&1 | tee -a "$OUT" | tail -5
rc=${PIPESTATUS[0]}
# proof that the instrumented objects were the ones loaded
LD_PRELOAD="$ASAN:$UBSAN" BRL_ORACLE_BUILD=asan ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python -c "
from oracle import Oracle
from oracle.binding import shim_path
import ctypes
Oracle(); ctypes.CDLL(shim_path())
print('# loaded:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'liboracle' in l or 'libasan' in l or 'libubsan' in l}))" | tee -a "$OUT"
echo "# exit code $rc; sanitizer reports in this log: $(grep -c -E 'ERROR: AddressSanitizer|runtime error:' "$OUT")" | tee -a "$OUT"
exit $rc
