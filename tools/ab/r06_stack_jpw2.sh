#!/bin/bash
# Round 6 A/B, five alternating pairs: all of an env's jobs in one wavefront vs two / three jobs per wavefront, float32 stack (profiling build)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4 5; do for j in 0 2 3; do CRL_LIB_VARIANT=abl CRL_GRAY_JPW=$j python tools/stack_time.py f32 150 2>&1 | grep -v amdgpu; done; done
