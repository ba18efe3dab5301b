cd $GRAFT_REPO_ROOT
python bench.py --gpus 8 --debug-single-device --mode batch --tois 64 --batch-n 20000 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pr=d['config']['per_rank']
for k,v in pr.items(): print(k, ['%.4f'%x for x in v])
print('ms_per_step', d['ms_per_step'])
"
python bench.py --gpus 1 --mode batch --tois 64 --batch-n 20000 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pr=d['config']['per_rank']
for k,v in pr.items(): print(k, ['%.4f'%x for x in v])
print('ms_per_step', d['ms_per_step'])
"
