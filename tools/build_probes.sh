#!/bin/sh
# Builds the stand-alone probes used for the rocprofv3 --pmc passes and the MFMA rate check.  gemm_bench drives the
# library's own contraction launchers, so it links against the built libcrm_hip.so (python -m cellregmap_amd.build).
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -Lcellregmap_amd -lcrm_hip \
      -Wl,-rpath,'$ORIGIN/../cellregmap_amd' -o tools/gemm_bench
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_clock.hip -o tools/mfma_clock
