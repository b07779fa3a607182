#!/bin/bash
# build the library in-tree, then run a command on an MI355X box: tools/g.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python kosmos-x_amd/build.py > /tmp/kx_build.log 2>&1 || { grep -E "error" -A3 /tmp/kx_build.log | head -40; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
