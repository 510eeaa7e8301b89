cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2s
mkdir -p $OUT
V=${VARIANTS:-0,20}
timeout 300 python scripts/gemm_sweep.py $V 100000 200 6 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_d200.log
timeout 300 python scripts/gemm_sweep.py $V 65536 512 4 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_d512.log
for tv in ${TLV:-31}; do TL_VARIANT=$tv timeout 300 python scripts/gemm_timeline.py 32768 200 2>&1 | grep -v amdgpu.ids > $OUT/timeline_$tv.log; tail -22 $OUT/timeline_$tv.log; done
