python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','ms_per_step','dtype','parity_vs_oracle')})
print(d['roofline']); print(d['cpu_baseline']); print(d.get('sustained'))"
