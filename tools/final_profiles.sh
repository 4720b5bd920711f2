tag=round5
bash tools/profile.sh $tag > gpurun_out/profile_$tag.log 2>&1
bash tools/profile_b3.sh $tag > gpurun_out/profile_b3_$tag.log 2>&1
bash tools/profile_cfg5.sh $tag > gpurun_out/profile_cfg5_$tag.log 2>&1
head -8 gpurun_out/${tag}_kernel_trace_summary.txt | cut -c1-120
