#!/bin/bash
# The library's HOST code (pf_pack.cpp, pf_input.cpp, pf_gzip.cpp: packer, native reader, one-pass scanner, gzip writer) built
# with AddressSanitizer + UBSan by g++ and linked with the unchanged HIP objects into /tmp/pf_asan/libpanfeed_hip.so; the CPU
# test suite and the reader fuzz then run against THAT library (no GPU needed: the sanitizers are not available for device
# code on this pool, and nothing here launches a kernel).  Usage: bash tools/asan_host.sh [fuzz cases]
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/pf_asan
CASES=${1:-300}
mkdir -p $OUT
for f in pf_pack pf_input pf_gzip; do
    g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC \
        -I$REPO/include -c $REPO/panfeed_amd/csrc/$f.cpp -o $OUT/$f.o
done
for f in pf_api pf_rowfilter; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $REPO/panfeed_amd/csrc/$f.hip -o $OUT/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libpanfeed_hip.so $OUT/pf_api.o $OUT/pf_rowfilter.o $OUT/pf_pack.o \
    $OUT/pf_input.o $OUT/pf_gzip.o -lz -Wl,--allow-shlib-undefined
cat > $OUT/run.py <<PY
import runpy, sys
sys.path.insert(0, "$REPO")
if __name__ == "__main__":
    import panfeed_amd._lib as l
    l.LIB_PATH = "$OUT/libpanfeed_hip.so"
    if sys.argv[1] == "pytest":
        import pytest
        sys.exit(pytest.main(sys.argv[2:]))
    sys.argv = sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name="__main__")
PY
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1
cd $REPO
# (tests/test_distributed_cpu.py spawns its ranks: they would load the ordinary library)
python $OUT/run.py pytest tests -q -m "not gpu" -p no:cacheprovider --ignore tests/test_distributed_cpu.py
python $OUT/run.py tests/fuzz_reader.py $CASES
