for mp in "1,0,0,0,0" "0,1,0,0,0" "0,0,1,0,0" "0,0,0,1,0"; do
python bench.py --workload mc --steps 5 --no-cpu-baseline --mode-probs $mp 2>/dev/null > /tmp/e.json
python -c "import json; d=json.load(open('/tmp/e.json')); print('$mp', {k:(v['avg_ms'], v['GBps']) for k,v in d['kernels'].items()})"
done
