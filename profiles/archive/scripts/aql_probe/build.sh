#!/bin/bash
# builds the probe next to its sources (GPU box or build container): ticket_kernel.hsaco (gfx950 code object) + aql_probe
set -e
cd $(dirname $0)
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O2 -o ticket_kernel.hsaco ticket_kernel.hip
/opt/rocm/bin/hipcc -O2 -std=c++17 -o aql_probe aql_probe.cpp -L/opt/rocm/lib -lhsa-runtime64
