#!/bin/bash
# builds kernel variants next to the default library for A/B timing on the GPU box: tools/ab_build.sh name "extra flags"
set -e
cd "$(dirname "$0")/../dql_multirotor_landing_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $2 dql_hip.hip -o "libdql_hip_$1.so"
