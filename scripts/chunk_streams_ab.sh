#!/bin/bash
# usage (GPU box, repo root): scripts/chunk_streams_ab.sh -> emulated N = 8 ranks, producer chunks on 1 vs 2 streams, 2 / 4 chunks: max over ranks of wall - fill (ms)
for cs in 1 2; do for c in 2 4; do
  GLNN_CHUNK_STREAMS=$cs python bench.py --emulate 8 --chunks $c --emulate-forms narrow,wide,mixed --detail-file gpurun_out/emu_cs${cs}_c$c.json > /dev/null 2>&1
  python - $cs $c <<'PY'
import json, sys
cs, c = sys.argv[1:]
d = json.load(open(f"gpurun_out/emu_cs{cs}_c{c}.json"))
for form, o in d["scale_model"].items():
    w = o["worlds"]["8"]
    print(f"streams {cs} chunks {c} {form:22s} max kernel ms {w['max_kernel_ms']:.3f}   max (wall - fill) ms {max(r['wall_ms'] - r['fill_ms'] for r in w['ranks']):.3f}   max wall {max(r['wall_ms'] for r in w['ranks']):.3f}  verified {w['verified']}")
PY
done; done
