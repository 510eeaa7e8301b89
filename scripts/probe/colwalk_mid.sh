cd $GRAFT_REPO_ROOT
for ARGS in "--rows 200000 --dim 200" "--rows 150000 --dim 512" "--rows 140000 --dim 256"; do
 for V in 0 48 0 48; do
  echo "== $ARGS variant $V"
  PLDA_GEMM_VARIANT=$V python bench.py $ARGS --speakers 1000 --steps 5 --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ms/step %.3f  kernel frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
 done
done
