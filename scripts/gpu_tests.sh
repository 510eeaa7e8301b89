set -x
cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40
