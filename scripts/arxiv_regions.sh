for i in 1 2 3; do python bench.py --workload arxiv --steps 200 --warmup 5 --student-steps-per-step 10 --no-cpu-baseline 2>/dev/null | grep '^DETAIL ' | tail -1 | cut -c8- | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), [round(x,3) for x in d['timed_regions_ms_per_step']], [(x['d'], round(x['avg_ms'],3)) for x in d['roofline']['all_aggregation_launches']])"; done
