#!/bin/bash
# where do kernel arguments live and what does reading them cost at the start of each launch?  (round 4)
#   HIP_FORCE_DEV_KERNARG = 0 / 1 (host-coherent vs device memory for the kernarg ring) on the single-frame path and the one-pair match GEMM,
#   and the same binaries built with -mllvm -amdgpu-kernarg-preload-count=16 (the first 16 dwords arrive in SGPRs with the wave: scalar
#   arguments only, a by-value struct such as ConvArgs is not preloaded).  -> gpurun_out/kernarg_ab.log
#   `tools/kernarg_ab.sh build` (in the build container, before the gpurun call) makes the two preload builds: tools/probes/mnn_probe_pl, tools/ab/v_preload.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O
if [ "$1" = build ]; then
  PL="-mllvm -amdgpu-kernarg-preload-count=16"; F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value"
  ( cd tools/probes && /opt/rocm/bin/hipcc $F -fno-honor-nans $PL -c mnn_probe_gemm.hip -o /tmp/mnn_probe_gemm_pl.o && /opt/rocm/bin/hipcc $F $PL -c mnn_probe.hip -o /tmp/mnn_probe_pl.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/mnn_probe_pl.o /tmp/mnn_probe_gemm_pl.o -o mnn_probe_pl )
  make -C xfeatslam_amd/csrc variant NAME=v_preload DEFS="$PL" | tail -1
  exit 0
fi
{
for env in default 0 1; do
  if [ $env = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$env; fi
  echo "== HIP_FORCE_DEV_KERNARG=$env"
  timeout 100 python tools/b1_loop.py
  XFEAT_HIP_LIB=$PWD/tools/ab/v_preload.so timeout 100 python tools/b1_loop.py | sed 's/$/   [kernels_misc + kernels_conv built with kernarg preload]/'
  timeout 100 tools/probes/mnn_probe 200 | grep -A1 "gemm pipe1 stg1  " | head -2
  timeout 100 tools/probes/mnn_probe_pl 200 | grep -A1 "gemm pipe1 stg1  " | head -2 | sed 's/$/   [preload]/'
done
} 2>&1 | tee $O/kernarg_ab.log
