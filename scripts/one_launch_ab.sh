#!/bin/bash
# Round 6: the fused layer of the sharded forward as ONE launch over chunk-ordered tiles with completion signals (GLNN_ONE_LAUNCH=1, default)
# against one launch per chunk (=0): every rank of N = 8 emulated on one GPU, chunks 2 and 4; per form the max over ranks of the kernel ms.
export TMPDIR=/tmp
for ol in 0 1; do for c in 2 4; do
  GLNN_ONE_LAUNCH=$ol timeout 900 python bench.py --emulate 8 --steps 3 --chunks $c --detail-file gpurun_out/one_launch_${ol}_c$c.json > gpurun_out/one_launch_${ol}_c$c.log 2>&1
  tail -1 gpurun_out/one_launch_${ol}_c$c.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,v in d['scale_model'].items():
    w=v['worlds']; e=w[-1] if isinstance(w,list) else w[sorted(w)[-1]]
    ks={}
    for r in e['ranks']:
        for n,t in r['kernels'].items(): ks[n]=max(ks.get(n,0),t)
    print('one_launch $ol chunks $c %-20s max kernel ms %.3f' % (k, e['max_kernel_ms']), {n:round(t,3) for n,t in ks.items()}, e['verified'])
"
done; done
