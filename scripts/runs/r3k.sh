cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
(time timeout 1500 python bench.py) > $O/bench_full.log 2> $O/bench_full.err
tail -5 $O/bench_full.err
python - $O/bench_full.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l)
        print(d['ms_per_step'], d['roofline'])
        print(json.dumps(d['cpu_baseline']))
        ex=d['extras']
        for k,v in ex.items():
            print(k, json.dumps(v)[:900])
PY
