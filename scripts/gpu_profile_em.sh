# The round's profile set: scripts/gpu_profile.sh for C2 (rocprofv3 kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes, SQ counters of
# the trials GEMM), then the EM's kernels: kernel trace + stats of scripts/fit_groups_probe.py and one EM iteration kernel by kernel
# (scripts/em_iter_trace.py) for the row form (G = 40, 36, 12) and the moment form (G = 1).
# usage (GPU box, through gpurun): bash scripts/gpu_profile_em.sh ; copy the summaries into profiles/ afterwards
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ROUND=r06 CONFIG=C2 bash scripts/gpu_profile.sh > gpurun_out/prof_r06_C2.log 2>&1
tail -40 gpurun_out/prof_r06_C2.log
# the EM's kernels (row form at G = 36, moment form at G = 1): kernel trace + stats of the fit probe
rm -rf gpurun_out/prof_r06_em; mkdir -p gpurun_out/prof_r06_em
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06_em -o em -- python $GRAFT_REPO_ROOT/scripts/fit_groups_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r06_em/probe.txt 2>&1 )
f=$(find gpurun_out/prof_r06_em -name "*kernel_trace.csv" | head -1)
( echo "one EM iteration (the 5th of the last fit), kernel by kernel: start offset us, duration us"; echo "G = 40 (n_k in [1, 200]):"; python scripts/em_iter_trace.py $f 1; echo "G = 36 (n_k in [5, 60]):"; python scripts/em_iter_trace.py $f 4; echo "G = 12:"; python scripts/em_iter_trace.py $f 7; echo "G = 1 (moment form):"; python scripts/em_iter_trace.py $f 10 ) > gpurun_out/prof_r06_em/r06_em_iteration_kernels.txt 2>&1
cat gpurun_out/prof_r06_em/r06_em_iteration_kernels.txt
find gpurun_out/prof_r06_em -name "*.csv" -size +2M -delete
