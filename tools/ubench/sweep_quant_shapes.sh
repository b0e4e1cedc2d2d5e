#!/bin/bash
# Workgroup shape of the quantizers (threads x 16-byte loads per thread), every encoder x input type, one tensor per
# launch.  Build the binaries first on the CPU box (make -C tools/ubench shapes); run on the GPU box from the repo root.
set -e
cd "$(dirname "$0")/../.."
for s in ${SHAPES:-64x1 64x2 64x4 128x1 128x2 128x4 256x1 256x2 256x4 512x1 512x2}; do
  echo "== shape $s"
  timeout -k 10 200 tools/ubench/ubench_q$s ${1:-1024} ${2:-7} "quant,single" | grep -v "^variant"
done
