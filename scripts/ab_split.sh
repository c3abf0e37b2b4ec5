#!/bin/bash
# Same-box A/B of the bucket-range split of the accumulate (VERDICT r5 task 5; pipeline.hip, -DKZG_EXP_ACC_SPLIT):
#   A = the shipped library, B = zkp_subnet_amd/ab/split.so  (KZG_BUILD_TAG=split KZG_EXTRA_HIPCC_FLAGS=-DKZG_EXP_ACC_SPLIT
#   python -m zkp_subnet_amd.build).  First the MSM parity tests on B (a fast answer that is wrong is no answer), then three
#   interleaved rounds of the driver-style headline.   gpurun -- 'bash scripts/ab_split.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/r06_ab_split.log}
{
if [ -z "$SKIP_PARITY" ]; then
echo "== parity of the split build: tests/test_gpu_msm.py"
KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/split.so python -m pytest tests/test_gpu_msm.py -x -q 2>&1 | tail -3
fi
echo "== one request at a time, python bench.py --steps 20 --warmup 5 (no extras; ms per 2^20-point MSM; stage table)"
for r in 1 2 3; do
  for v in A B; do
    LIB=$PWD/zkp_subnet_amd/libkzg_mi355x.so; [ $v = B ] && LIB=$PWD/zkp_subnet_amd/ab/split.so
    echo -n "round $r $v: "
    KZG_MI355X_LIB=$LIB python bench.py --steps 20 --warmup 5 --no-kzg-rows --no-cpu-baseline --no-adversarial --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
st=d.get('stages_ms',{})
print('ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  stages %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], (d.get('pipelined') or {}).get('ms_per_step',0), {k: round(v,4) for k,v in st.items() if v}))"
  done
done
} 2>&1 | tee $OUT
