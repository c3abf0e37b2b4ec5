cd $GRAFT_REPO_ROOT
cp zkp_subnet_amd/libkzg_mi355x.so /tmp/_orig.so
for r in 1 2 3; do for v in A B; do cp zkp_subnet_amd/ab/$v.so zkp_subnet_amd/libkzg_mi355x.so
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['pipelined']['ms_per_step'],4), d['stages_ms']['accumulate'])"
done; done
cp /tmp/_orig.so zkp_subnet_amd/libkzg_mi355x.so
