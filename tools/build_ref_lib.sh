#!/bin/bash
# Build bokego_amd/libbokego_amd_old.so from the kernel/engine sources of a git ref (default HEAD), for same-box
# A/B runs with tools/ab_bench.sh (box-to-box variation of the MI355X pool is larger than most kernel changes).
set -e
REF=${1:-HEAD}; ROOT=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
for f in bk_kernels.hip bk_kernels_f16.hip bk_encode.hip bk_engine.cpp bk_internal.h; do git -C "$ROOT" show "$REF:bokego_amd/csrc/$f" > "$T/$f"; done
git -C "$ROOT" show "$REF:bokego_amd/csrc/bk_encode_dev.h" > "$T/bk_encode_dev.h" 2>/dev/null || rm -f "$T/bk_encode_dev.h"   # (since round 6)
sed -i "s|\.\./\.\./include/|$ROOT/include/|" "$T/bk_engine.cpp"
(cd "$T" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared bk_kernels.hip bk_kernels_f16.hip bk_encode.hip bk_engine.cpp -o "$ROOT/bokego_amd/libbokego_amd_old.so")
rm -rf "$T"; ls -la "$ROOT/bokego_amd/libbokego_amd_old.so"
