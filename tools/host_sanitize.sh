#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the library (CPU only: GPU sanitizers are not available
# on this pool, and none of this needs a GPU).  The four translation units are compiled --offload-host-only (seconds: no
# device code is generated), their fat-binary symbols are satisfied by empty stand-ins, and tests/host_sanitize_main.cpp
# drives everything reachable without a device: validation, plan shape, kernel selection, the JIT cache incl. a real hipcc
# child process.   usage: tools/host_sanitize.sh [build dir]      exit code 0 = no finding
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/tmp/cpf_host_sanitize}
rm -rf "$out"; mkdir -p "$out/lib/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FL="--offload-arch=gfx950 --offload-host-only -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -Wno-unused-function"
$HIPCC $FL -c -o "$out/cpflow_amd.o" "$root/cpflow_amd/csrc/cpflow_amd.hip" &
$HIPCC $FL -c -o "$out/cpf_dense_aot.o" "$root/cpflow_amd/csrc/cpf_dense_aot.hip" &
$HIPCC $FL -c -o "$out/cpf_zyz_aot.o" "$root/cpflow_amd/csrc/cpf_zyz_aot.hip" &
$HIPCC $FL -c -o "$out/cpf_zyz3_aot.o" "$root/cpflow_amd/csrc/cpf_zyz3_aot.hip" &
wait
# the host objects reference their (absent) device code objects by a hashed symbol: define each as an empty blob
: > "$out/fatbin_stubs.c"
for sym in $(nm "$out/cpflow_amd.o" "$out/cpf_dense_aot.o" "$out/cpf_zyz_aot.o" "$out/cpf_zyz3_aot.o" | awk '$1 == "U" && $2 ~ /^__hip_fatbin_/ {print $2}' | sort -u); do
  echo "const char $sym[256] = {0};" >> "$out/fatbin_stubs.c"
done
gcc -c -fPIC -o "$out/fatbin_stubs.o" "$out/fatbin_stubs.c"
$HIPCC --offload-arch=gfx950 -fsanitize=address,undefined -shared -fPIC -o "$out/lib/libcpflow_amd.so" "$out/cpflow_amd.o" "$out/cpf_dense_aot.o" "$out/cpf_zyz_aot.o" "$out/cpf_zyz3_aot.o" "$out/fatbin_stubs.o" -lhiprtc
cp "$root"/cpflow_amd/csrc/*.h "$out/lib/csrc/"        # the JIT reads the kernel headers next to the library
$HIPCC -x c++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -o "$out/driver" "$root/tests/host_sanitize_main.cpp" -L"$out/lib" -lcpflow_amd -Wl,-rpath,"$out/lib"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$out/driver" "$out/cache"
