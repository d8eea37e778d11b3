#!/bin/bash
# Build everything, then run a command on an MI355X box through gpurun.
# usage: tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
make -s -j4 -C calibration_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
make -s -C oracle
make -s -C tests/cpu_backend 2>&1 | grep -E " error " && exit 1
exec /usr/local/graft/bin/gpurun "$@"
