#!/bin/bash
# Build libaudiocodecs_amd.so for gfx950 (MI355X) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libaudiocodecs_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -I"$here/../../include" "$@" -o "$out" "$here/ac_api.hip"
echo "built $out"
