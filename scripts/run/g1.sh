cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -15
python scripts/fit_groups_probe.py 2>&1 | tail -6
PLDA_EM_VARIANT=3 python scripts/fit_groups_probe.py 2>&1 | tail -6
