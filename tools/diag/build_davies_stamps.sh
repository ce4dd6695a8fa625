#!/bin/bash
# Diagnostic build of the library whose eig_davies_kernel (csrc/davies.hip: CRM_DAVIES_STAMPS) writes its phase durations
# over the eigenvalues it returns -> tools/_r05/libcrm_hip_davies_stamps.so; read by tools/diag/davies_phases.py --stamps.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/_r05
objs=$(ls cellregmap_amd/_build/*.o | grep -v "/davies.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DCRM_DAVIES_STAMPS -c cellregmap_amd/csrc/davies.hip -o /tmp/davies_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_r05/libcrm_hip_davies_stamps.so $objs /tmp/davies_stamps.o -ldl
echo "built tools/_r05/libcrm_hip_davies_stamps.so"
