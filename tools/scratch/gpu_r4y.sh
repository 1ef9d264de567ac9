#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
QP_PROFILE=1 QP_AT=8 AA_HIP_OPTIONS=qp_quad_lazy=1 timeout -k 10 300 python3 tools/qp_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm" | tail -20
QP_PROFILE=1 QP_AT=30 AA_HIP_OPTIONS=qp_quad_lazy=1 timeout -k 10 300 python3 tools/qp_profile.py 2>&1 | grep "lazy\|weights" | tail -8
