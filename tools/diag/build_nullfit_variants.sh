#!/bin/bash
# Diagnostic builds of the library with the round-5 forms of the null fit's wave sum and / or logarithms
# (cellregmap_amd/csrc/nullfit.hip: CRM_NF_BUTTERFLY_SUM, CRM_NF_SERIAL_LOGS) -> tools/_r05/libcrm_hip_<tag>.so;
# tools/diag/compare_builds.py holds them against round 5's build (CRM_OTHER_LIB / CRM_THIS_LIB).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/_r05
objs=$(ls cellregmap_amd/_build/*.o | grep -v "/nullfit.o")
build() {  # tag, flags...
  tag=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c cellregmap_amd/csrc/nullfit.hip -o /tmp/nullfit_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_r05/libcrm_hip_$tag.so $objs /tmp/nullfit_$tag.o -ldl
  echo "built tools/_r05/libcrm_hip_$tag.so"
}
build old_sum_old_logs -DCRM_NF_BUTTERFLY_SUM -DCRM_NF_SERIAL_LOGS &
build old_sum -DCRM_NF_BUTTERFLY_SUM &
build old_logs -DCRM_NF_SERIAL_LOGS &
wait
