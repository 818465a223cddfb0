#!/bin/bash
# the -m gpu suite under every fallback / option path (each knob is read once per process)
for env in "PAYNE_NO_PREP=1" "PAYNE_OUT_TILE=0" "PAYNE_OUT_TILE=9" "PAYNE_OUT_TILE=7" "PAYNE_OUT_TILE=6" "PAYNE_DMA_WIDE=0" "PAYNE_POST_FULL=1" "PAYNE_HIDDEN_KERNEL=0" "PAYNE_TW_GLOBAL=1" "PAYNE_POST_GENERIC=1" "PAYNE_BIG_TILED=0"; do
  r=$(env $env timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)
  echo "$env : $r"
done
