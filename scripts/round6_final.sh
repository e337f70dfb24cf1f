#!/bin/bash
# usage (GPU box, repo root): scripts/round6_final.sh TAG -- every measurement of the round in one call: the default bench (the driver's command),
# rocprofv3 kernel stats of the same command (shortened), the PMC traffic passes, the emulated scaling model (chunks 2 and 4), the K3 probes,
# the MLP3w8 timeline, teacher-training kernel stats.  Results under gpurun_out/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_${TAG}.log 2> gpurun_out/bench_${TAG}.err
cp gpurun_out/bench_detail.json gpurun_out/bench_${TAG}.json 2>/dev/null
cp gpurun_out/bench_detail_xl.json gpurun_out/bench_${TAG}_xl.json 2>/dev/null
cp gpurun_out/bench_detail_arxiv.json gpurun_out/bench_${TAG}_arxiv.json 2>/dev/null
tail -1 gpurun_out/bench_${TAG}.log > gpurun_out/bench_${TAG}_line.json
scripts/profile_round.sh ${TAG} ogbn-products > gpurun_out/profile_${TAG}.log 2>&1
scripts/pmc_round.sh ${TAG} > gpurun_out/pmc_${TAG}.log 2>&1
for c in 2 4; do
  python bench.py --emulate 2,4,8 --steps 3 --chunks $c --detail-file gpurun_out/scale_model_${TAG}_c$c.json > gpurun_out/emu_${TAG}_c$c.log 2>&1
done
scripts/student_w8_timeline.sh ${TAG} > /dev/null 2>&1
scripts/train_sage_timeline.sh ${TAG} > /dev/null 2>&1
tail -c 3500 gpurun_out/bench_${TAG}_line.json
echo
for c in 2 4; do tail -1 gpurun_out/emu_${TAG}_c$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks $c', {k:{N:round(w['max_kernel_ms'],3) for N,w in v.items() if N!='one_gpu_ms'} for k,v in d['scale_model'].items()}, d['verified'])"; done
cat gpurun_out/student_w8_${TAG}.txt | head -20
head -3 gpurun_out/train_sage_timeline_${TAG}.txt; tail -2 gpurun_out/train_sage_timeline_${TAG}.txt
