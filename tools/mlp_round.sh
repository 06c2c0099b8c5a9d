#!/bin/bash
# The MLP section's evidence on one box: bash tools/mlp_round.sh <tag>  -> gpurun_out/<tag>_mlp_*.txt
# request-shape micro-benchmark, section time, phases of the chain and of the weight-gradient launch, per-kernel durations under
# rocprofv3, PMC counters (one pass per group), the DeepFM step through the trainer and through fmx_deepfm_stream
tag=${1:-run}
out=gpurun_out
mkdir -p $out
timeout -k 10 120 tools/micro/l2_shapes > $out/${tag}_l2_shapes.txt 2>&1
timeout -k 10 120 python tools/mlp_section_times.py 2>&1 | grep -v amdgpu > $out/${tag}_mlp_section_times.txt
timeout -k 10 120 python tools/mlp_chain_stamps.py 2>&1 | grep -v amdgpu > $out/${tag}_mlp_chain_stamps.txt
timeout -k 10 120 python tools/mlp_wgrad_stamps.py 2>&1 | grep -v amdgpu > $out/${tag}_mlp_wgrad_stamps.txt
bash tools/mlp_kernel_trace.sh ${tag}_kt > $out/${tag}_mlp_kernel_trace.txt 2>&1
bash tools/mlp_kernel_pmc.sh ${tag} > $out/${tag}_mlp_kernel_pmc.txt 2>&1
python tools/deepfm_host_time.py 2>&1 | grep steps > $out/${tag}_deepfm_steps.txt
bash tools/deepfm_trace.sh ${tag}_dt >> $out/${tag}_deepfm_steps.txt 2>&1
echo "--- the native loop (fmx_deepfm_stream) ---" >> $out/${tag}_deepfm_steps.txt
bash tools/deepfm_trace.sh ${tag}_ds stream >> $out/${tag}_deepfm_steps.txt 2>&1
tail -3 $out/${tag}_mlp_section_times.txt $out/${tag}_deepfm_steps.txt
