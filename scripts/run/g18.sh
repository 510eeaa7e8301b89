cd $GRAFT_REPO_ROOT
for c in 192 256 384 512; do echo "chunk cus $c"; PLDA_EM_CHUNK_CUS=$c python scripts/fit_groups_probe.py 2>&1 | tail -3; done
