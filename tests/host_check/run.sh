#!/bin/sh
# Host side of libsccbf under AddressSanitizer and UndefinedBehaviorSanitizer: every C ABI entry
# point on the calls that return before a launch (empty batches, bad arguments).  Needs no GPU; the
# device code is compiled as usual, only the host pass is instrumented.  About two minutes.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
out=${TMPDIR:-/tmp}/sccbf_host_check
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops \
    -mllvm -amdgpu-kernarg-preload-count=16 \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=all \
    -I"$root/include" -o "$out" "$root/sccav_cbf_amd/csrc/sccbf.hip" "$here/driver.cpp"
ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 "$out"
