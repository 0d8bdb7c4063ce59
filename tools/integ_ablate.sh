#!/bin/bash
# timing-only ablation of the fused IntegrationNetwork forward (results are wrong with a knob set): tools/integ_ablate.sh [BM]
BM=${1:-128}
for d in 0 1 2 4 8 16 32 3 19 23 63 59; do
  echo "dbg=$d: $(DIST_AMD_INTEG_BM=$BM DIST_AMD_INTEG_DBG=$d python tools/bench_integ.py 2>/dev/null | tr '\n' ' ')"
done
