#!/bin/bash
# Who waits in the ring of the wave-specialised GEMM (csrc/gemm_h2w.hip)?  Rebuild with poll counters, time one shape with the kernel switched on.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
touch xpoint_amd/csrc/gemm_h2w.hip
XP_EXTRA_HIPCC_FLAGS="-DXP_H2W_DBG=1" python3 -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
XP_H2W=1 GB_H2=1 GB_ONLY=${GB_ONLY:-12} timeout 120 python3 -c "
import runpy, sys
sys.argv=['gemm_bench.py']
runpy.run_path('tools/gemm_bench.py')
from xpoint_amd import _lib as L
print('err', L.load().xp_gemm_h2w_error())
" 2>&1 | grep -E "^M|polls|err"
touch xpoint_amd/csrc/gemm_h2w.hip; python3 -m xpoint_amd.build > /dev/null 2>&1
