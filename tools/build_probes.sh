#!/bin/sh
# Builds the stand-alone probes used for the rocprofv3 --pmc passes and the MFMA rate check.
set -e
cd "$(dirname "$0")/.."
CS=cellregmap_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip $CS/gemm_tn.hip $CS/gemm_tn_glds.hip $CS/api_core.hip \
      $CS/davies.hip $CS/nullfit.hip $CS/nullfit_wide.hip -o tools/gemm_bench
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_clock.hip -o tools/mfma_clock
