set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/r4/bench_c2.json 2> gpurun_out/r4/bench_c2.err; tail -c 1500 gpurun_out/r4/bench_c2.json; tail -5 gpurun_out/r4/bench_c2.err
PLDA_GEMM_VARIANT=30 timeout 900 python bench.py --steps 20 --warmup 2 --no-cpu --no-extra > gpurun_out/r4/bench_c2_bt2.json 2> gpurun_out/r4/bench_c2_bt2.err; python -c "
import json;j=json.load(open('gpurun_out/r4/bench_c2_bt2.json'));print('bt2:',j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['kernel'])
j=json.load(open('gpurun_out/r4/bench_c2.json'));print('bt4:',j['value'],j['ms_per_step'],j['roofline']['frac'],j['roofline']['kernel'],j['oracle_check'],j['fit']['stages'])"
