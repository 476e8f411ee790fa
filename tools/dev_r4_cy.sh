#!/bin/bash
# dev (GPU box): build the PS_HVAR variants of gemm_bf16.hip and time the C @ Y product with each
cd "$(dirname "$0")/.."
for v in 0 1 2 3 4 5 6 7 8 9; do tools/ab_build.sh hv$v gemm_bf16.hip -DPS_HVAR=$v > /dev/null 2>&1 & done; wait
tools/ab_build.sh hs4v6 gemm_bf16.hip -DPS_HSETS=4 -DPS_HVAR=6 > /dev/null 2>&1 &
tools/ab_build.sh hs4v9 gemm_bf16.hip -DPS_HSETS=4 -DPS_HVAR=9 > /dev/null 2>&1 &
wait
python tools/dev_r4_cy.py 2>&1 | grep TB
for v in 0 1 2 3 4 5 6 7 8 9; do PS_AB_LIB=.ab/hv$v/libprecondition_amd.so python tools/dev_r4_cy.py 2>&1 | grep TB; done
for v in hs4v6 hs4v9; do PS_AB_LIB=.ab/$v/libprecondition_amd.so python tools/dev_r4_cy.py 2>&1 | grep TB; done
