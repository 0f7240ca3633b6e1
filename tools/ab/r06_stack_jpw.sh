#!/bin/bash
# Round 6 A/B (profiling build): jobs per wavefront of the fused frame-stack draw (CRL_GRAY_JPW; 0 = all of an env's jobs), kernel us.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for j in 0 1 2 3 4; do CRL_LIB_VARIANT=abl CRL_GRAY_JPW=$j python tools/stack_time.py f32 100; done; done
for j in 0 1 2 3 5; do CRL_LIB_VARIANT=abl CRL_GRAY_JPW=$j python tools/stack_time.py u8 100; done
