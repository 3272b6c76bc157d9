#!/bin/bash
# row tiles of the packed pyramid kernel: parity tests per setting, then timings (tools/time_pyr_knob.py) with and without the XCD-aware order
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rt in 2 4; do for x in 0 1; do
  echo "== tests row_tiles=$rt xcd=$x"; VPP_TUNE="pyr.row_tiles=$rt,pyr.xcd=$x" timeout 600 python -m pytest tests/test_gpu_algos.py tests/test_gpu_edges.py -m gpu -x -q -k "pyramid or pyramids" 2>&1 | tail -2
done; done
for x in 0 1; do echo "== xcd=$x"; VPP_TUNE="pyr.xcd=$x" timeout 300 python tools/time_pyr_knob.py pyr.row_tiles 0 2 4 2>&1 | grep x; done
