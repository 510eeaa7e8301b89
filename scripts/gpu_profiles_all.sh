# rocprofv3 evidence of a round for every single-GPU BASELINE shape: kernel trace + stats, FETCH / WRITE, SQ counters
# (scripts/gpu_profile.sh per configuration).  Copy gpurun_out/prof_<ROUND>_<C>/<ROUND>_<C>_* into profiles/ afterwards.
# usage (on the GPU box, through gpurun):  ROUND=r05 bash scripts/gpu_profiles_all.sh [C2 C3 C4]
cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
CONFIGS=${@:-C2 C3 C4}
for C in $CONFIGS; do
  ROUND=$ROUND CONFIG=$C bash scripts/gpu_profile.sh > gpurun_out/prof_${ROUND}_$C.log 2>&1
  tail -30 gpurun_out/prof_${ROUND}_$C.log
done
