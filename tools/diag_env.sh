#!/bin/bash
# Run a command against the DIAGNOSTICS build of the library (retired kernels and LR_* environment overrides compiled in):
#   python -m luciddreamer_amd.build --diagnostics          # once: luciddreamer_amd/lib_diag/liblucid_raster.so
#   tools/diag_env.sh python tools/ab_bench.py --knob bwd_red --values 1,0 ...
# The product library (luciddreamer_amd/lib/) is what everything else loads; it contains none of these.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export LR_LIB_DIR="$ROOT/luciddreamer_amd/lib_diag"
export LD_LIBRARY_PATH="$LR_LIB_DIR${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
exec "$@"
