# round 4: rocprofv3 evidence for every single-GPU shape (kernel trace + stats, FETCH / WRITE, SQ counters)
cd $GRAFT_REPO_ROOT
for C in C2 C3 C4; do
  ROUND=r04 CONFIG=$C bash scripts/gpu_profile.sh > gpurun_out/prof_r04_$C.log 2>&1
  tail -30 gpurun_out/prof_r04_$C.log
done
