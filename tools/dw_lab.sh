#!/bin/bash
for tall in 1 0; do for wgs in 1536 2048 3072; do
 echo "TALL=$tall WGS=$wgs"
 NR_DW_TALL=$tall NR_DW_WGS=$wgs timeout -k 10 100 python tools/gemm_bench.py 2>/dev/null | grep -v 768 | sed 's/.*| dW/dW/' | cut -c1-24 || exit 1
done; done
