#!/bin/bash
# stage-removal timings of the matcher's Gram kernels (XP_MT_DBG builds; results of those builds are wrong by design)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for k in "$@"; do
  touch $R/xpoint_amd/csrc/match.hip
  XP_EXTRA_HIPCC_FLAGS="-DXP_MT_DBG=$k" python3 -m xpoint_amd.build > /dev/null 2>&1
  echo "== XP_MT_DBG=$k"
  bash $R/tools/match_prof.sh mdbg$k 4060 8192 8 | grep -E "gram|per call"
done
touch $R/xpoint_amd/csrc/match.hip; python3 -m xpoint_amd.build > /dev/null 2>&1
