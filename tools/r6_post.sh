#!/bin/bash
# gpurun_out/ of tools/r6_round.sh + tools/r6_round.sh -> profiles/r06_* (run in the repository, no GPU needed)
cd "$(dirname "$0")/.."; O=gpurun_out
python tools/summarize_profiles.py r06
for B in 256 64; do
  rm -rf $O/pmck1 $O/pmck2; cp -r $O/pmck1_B$B $O/pmck1; cp -r $O/pmck2_B$B $O/pmck2
  python tools/pmc_kernels.py r06 $([ $B = 64 ] && echo _B64) > profiles/r06_pmc_kernels_B$B.txt 2>&1
done
rm -rf $O/pmck1 $O/pmck2
python tools/roofline_table.py r06 prof_serial > /dev/null
python tools/roofline_table.py r06 prof_serial64 _B64 > /dev/null
cp $O/bench_default.json profiles/r06_bench_default.json; cp $O/pytest_gpu.log profiles/r06_pytest_gpu.log
{ echo "# tools/probes/mnn_seg_probe 200 on MI355X (round 6): the persistent many-pairs GEMM k_mnn_gemm_seg against k_mnn_gemm_img -- correctness on every shape, A/B, phase stamps"; cat $O/mnn_seg_probe.log; echo; echo "# tools/probes/mnn_probe 200 (the round-2 probe of the one-pair path, same box)"; cat $O/mnn_probe.log; } > profiles/r06_mnn_probe.log
{ echo "# tools/probes/pipe_probe on MI355X (round 6): f32 MFMA vs VALU on one SIMD (clock64 ticks): same wave, two waves, s_setprio, yielding, dependent chains"; cat $O/pipe_probe.log; } > profiles/r06_pipe_probe.log
cp $O/gemm_b2b.md profiles/r06_gemm_b2b.md; cp $O/b1_modes.log profiles/r06_b1_modes.log
cp $O/host_batch_probe.log profiles/r06_host_batch_probe.log; cp $O/queue_view.txt profiles/r06_queue_view.txt
{ echo "# tests/test_gpu_campaign.py + tests/test_gpu_hazard.py with -s on MI355X (round 6): every (weight family, image family, size) case with its near-tie audit, the hazard probe built with the box's compiler"; cat $O/campaign_gpu.log; } > profiles/r06_campaign_gpu.log
{ echo "# tools/probes/dist_probe + tools/dist_probe.py on MI355X (round 6): k_dist_mfma with phases switched off one by one (timing only), then the product kernel back to back on device-resident rows"; cat $O/dist_probe.log; } > profiles/r06_dist_probe.log
{ echo "# tools/probes/mnn_tail_probe 300 on MI355X (round 6): the floor of a 4096 x 4096 match finished INSIDE the GEMM launch (last arriver per d1 panel): the real k_mnn_gemm_img + the tail's mandatory traffic and exchanges, no arithmetic; three levels of cache maintenance"; cat $O/mnn_tail_probe.log; } > profiles/r06_mnn_tail_probe.log
python tools/isa_mix.py --pmc profiles/r06_pmc_kernels.json --table profiles/r06_roofline_table.md > profiles/r06_isa_mix.md
ls profiles | grep r06
