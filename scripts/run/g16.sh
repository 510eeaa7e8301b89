cd $GRAFT_REPO_ROOT
python scripts/eig_probe.py 224:0 240:0 240:2 256:0 256:2 288:0 320:0 384:0 512:0 2>&1 | grep -v amdgpu.ids
