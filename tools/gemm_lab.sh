#!/bin/bash
# lab sweep: which part of the GEMM main loop costs the time (NR_GEMM_LAB bits: 1 no gload, 2 no lstore/barrier, 4 no epilogue)
for lab in ${LABS:-0 1 2 4 7}; do
  echo "LAB=$lab"
  NR_GEMM_LAB=$lab timeout -k 10 100 python tools/gemm_bench.py 2>/dev/null | cut -c1-260 || exit 1
done
