#!/bin/bash
# diagnostic kernels of tools/*.py (not product code): built by hand, the .so files travel to the GPU box with the snapshot
set -e
cd "$(dirname "$0")"
for f in lds_poison bperm_check gate_check spin stamp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $f.hip -o lib$f.so
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 graph_replay.hip -o graph_replay      # standalone program
