#!/bin/bash
# gpurun_out/ of tools/r4_round.sh + tools/r4_serial.sh -> profiles/r04_* (run in the repository, no GPU needed)
cd "$(dirname "$0")/.."; O=gpurun_out
python tools/summarize_profiles.py r04
for B in 256 64; do
  rm -rf $O/pmck1 $O/pmck2; cp -r $O/pmck1_B$B $O/pmck1; cp -r $O/pmck2_B$B $O/pmck2
  python tools/pmc_kernels.py r04 $([ $B = 64 ] && echo _B64) > profiles/r04_pmc_kernels_B$B.txt 2>&1
done
rm -rf $O/pmck1 $O/pmck2
python tools/roofline_table.py r04 prof_serial > /dev/null
python tools/roofline_table.py r04 prof_serial64 _B64 > /dev/null
cp $O/bench_default.json profiles/r04_bench_default.json; cp $O/pytest_gpu.log profiles/r04_pytest_gpu.log
{ echo "# tools/probes/mnn_seg_probe 200 on MI355X (round 4): the persistent many-pairs GEMM k_mnn_gemm_seg against k_mnn_gemm_img -- correctness on every shape, A/B, phase stamps"; cat $O/mnn_seg_probe.log; echo; echo "# tools/probes/mnn_probe 200 (the round-2 probe of the one-pair path, same box)"; cat $O/mnn_probe.log; } > profiles/r04_mnn_probe.log
{ echo "# tools/probes/pipe_probe on MI355X (round 4): f32 MFMA vs VALU on one SIMD (clock64 ticks): same wave, two waves, s_setprio, yielding, dependent chains"; cat $O/pipe_probe.log; } > profiles/r04_pipe_probe.log
cp $O/gemm_b2b.md profiles/r04_gemm_b2b.md; cp $O/b1_modes.log profiles/r04_b1_modes.log
cp $O/host_batch_probe.log profiles/r04_host_batch_probe.log; cp $O/queue_view.txt profiles/r04_queue_view.txt
ls profiles | grep r04
