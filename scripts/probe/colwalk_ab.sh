# A/B of the bt4 tile-queue order (score.hip: bt4_schedule): PLDA_GEMM_VARIANT 0 = product, 48 = row walk
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LOG=gpurun_out/colwalk_ab.log; rm -f $LOG
for C in ${CONFIGS:-C2 C3 C4}; do
 for V in ${VARIANTS:-0 48 0 48}; do
  echo "== $C variant $V" >> $LOG
  PLDA_GEMM_VARIANT=$V python bench.py --config $C --steps 8 --warmup 2 --no-cpu --no-extra >> $LOG 2>&1
 done
done
python - <<'PY'
import json
for l in open("gpurun_out/colwalk_ab.log"):
    if l.startswith("=="): print(l.strip())
    elif l.startswith("{"):
        d=json.loads(l); print("  ms/step %.3f  kernel frac %.4f  achieved %.1f" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"]))
PY
