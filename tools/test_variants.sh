#!/bin/bash
# Runs the GPU parity tests of the line / ORB extractors once per environment switch that selects an alternative kernel
# (kept for A/B measurements).  usage on the GPU box: bash tools/test_variants.sh
set -e
for v in "PSLFE_LSD_GROW=lds" "PSLFE_LSD_GROW=serial" "PSLFE_LSD_SCALE_SIMPLE=1"; do
  echo "== $v"
  env $v timeout -k 10 400 python -m pytest tests/test_line_gpu.py -x -q -k "lsd_segments or large_regions or full_line" | tail -1
done
for v in "PSLFE_FAST_V1=1" "PSLFE_PYR_SIMPLE=1" "PSLFE_NO_XCD=1" "PSLFE_OVERLAP=1"; do
  echo "== $v"
  env $v timeout -k 10 300 python -m pytest tests/test_orb_gpu.py -x -q | tail -1
done
