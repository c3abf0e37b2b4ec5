cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in A B; do
KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['pipelined']['ms_per_step'],4), d['stages_ms']['accumulate'])"
done; done
