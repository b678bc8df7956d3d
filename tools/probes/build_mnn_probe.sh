#!/bin/bash
# builds tools/probes/mnn_probe (gfx950); the binary travels to the GPU box with the repo snapshot
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value"
/opt/rocm/bin/hipcc $F -fno-honor-nans -c mnn_probe_gemm.hip -o /tmp/mnn_probe_gemm.o
/opt/rocm/bin/hipcc $F -c mnn_probe.hip -o /tmp/mnn_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/mnn_probe.o /tmp/mnn_probe_gemm.o -o mnn_probe
# the persistent many-pairs GEMM (mnn_gemm_seg.hip.h) against the one-tile-per-workgroup kernel
/opt/rocm/bin/hipcc $F -fno-honor-nans -c mnn_seg_probe_gemm.hip -o /tmp/mnn_seg_probe_gemm.o
/opt/rocm/bin/hipcc $F -c mnn_seg_probe.hip -o /tmp/mnn_seg_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/mnn_seg_probe.o /tmp/mnn_seg_probe_gemm.o -o mnn_seg_probe
# the floor of a match finished inside the GEMM launch (mnn_tail_probe.hip): the real GEMM + the skeleton of a last-arriver tail
/opt/rocm/bin/hipcc $F -fno-honor-nans mnn_tail_probe.hip -o mnn_tail_probe
