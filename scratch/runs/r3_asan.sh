#!/bin/bash
# Host-side parsers and writers under AddressSanitizer + UBSan (CPU only):
#   make asan -> libepihip_host_asan.so (bam_pack.cpp, report_writer.cpp, host_common.cpp; pinned allocation -> malloc)
#   tests/test_preprocess_bam.py (fixtures x options, windows, malformed records, threads), the writer tests of
#   tests/test_host_api.py, and the shim-core host test (tests/cpp/test_shim_core.cpp cpu) compiled with the same flags.
# usage: scratch/runs/r3_asan.sh [outfile]
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-$R/profiles/r03_asan_host.txt}
cd $R
make -C epialleler_amd/csrc asan > /dev/null || exit 1
ASAN_SO=$(g++ -print-file-name=libasan.so)
UBSAN_SO=$(g++ -print-file-name=libubsan.so)
export EPIHIP_LIB=$R/epialleler_amd/csrc/libepihip_host_asan.so EPIHIP_HOST_ONLY=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
echo "# $(date -u +%F) host-side sanitizer run: g++ $(g++ -dumpversion), -fsanitize=address,undefined -fno-sanitize-recover=undefined"
echo "## pytest (LD_PRELOAD=libasan): tests/test_preprocess_bam.py + writer tests"
LD_PRELOAD="$ASAN_SO $UBSAN_SO" python -m pytest tests/test_preprocess_bam.py -q -p no:cacheprovider 2>&1 | tail -4
LD_PRELOAD="$ASAN_SO $UBSAN_SO" python -m pytest tests/test_host_api.py tests/test_long_read.py -q -k "write_report or producer" -m "not gpu" -p no:cacheprovider 2>&1 | tail -4
echo "## shim core (cpu mode) compiled with the same flags"
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -DEPI_SHIM_CPU_ONLY -Wall -I include -I epialleler_amd/r \
    tests/cpp/test_shim_core.cpp -o /tmp/test_shim_core_asan -L epialleler_amd/csrc -lepihip_host_asan -Wl,-rpath,$R/epialleler_amd/csrc -lpthread && \
  /tmp/test_shim_core_asan cpu tests/golden/bam/capture.bam
echo "shim core exit code: $?"
} > $OUT 2>&1
cat $OUT
