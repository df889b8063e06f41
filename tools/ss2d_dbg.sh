#!/bin/bash
# stage removal in the chunked SS2D passes (results wrong on purpose): per-variant libraries, timed with tools/ss2d_bench.py
for d in ${SS2D_DBG_SET:-0 1 2 3}; do
  touch xpoint_amd/csrc/ss2d.hip
  XP_EXTRA_HIPCC_FLAGS="-DXP_SS2D_DBG=$d" python -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
  cp xpoint_amd/libxpoint_hip.so xpoint_amd/libxp_ss2d_dbg$d.so
done
touch xpoint_amd/csrc/ss2d.hip; python -m xpoint_amd.build > /dev/null 2>&1
