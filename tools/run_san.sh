#!/bin/bash
# The CPU test suite on AddressSanitizer + UndefinedBehaviorSanitizer builds of everything native it exercises: the oracle
# (oracle/Makefile SAN=1) and the three host builds of the product's per-path code (tests/host_harness/Makefile `san`: both
# kernel cores and the tracer, mega + wavefront + reparam).  The reference's counterpart: MI_SANITIZE_ADDRESS,
# CMakeLists.txt:34-35, 245-268.  GPU sanitizers are not available on this pool; the device code is the same headers.
#   tools/run_san.sh [pytest args]      (default: the whole `-m "not gpu"` suite; ~3x slower than the plain run)
set -euo pipefail
cd "$(dirname "$0")/.."
make -C oracle -s SAN=1
make -C tests/host_harness -s san
export EPSM_SAN=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1"        # python itself leaks by design; any report aborts the run
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
if [ $# -eq 0 ]; then set -- tests -x -q -m "not gpu" -p no:cacheprovider; fi
exec python -m pytest "$@"
