#!/bin/bash
# Build every native piece here (hipcc cross-compiles gfx950), then hand the command to gpurun: the GPU box runs the
# snapshot's prebuilt .so files, so a stale library would silently test old code.   usage: tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
make -s -j4 -C bloomfiltertrie_amd/csrc all
make -s -C oracle all >/dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
