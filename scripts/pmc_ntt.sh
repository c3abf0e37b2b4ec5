cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_ntt
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_ntt/$tag -o p --output-format csv -- python3 bench.py --workload kzg22 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_ntt/$tag.json 2> gpurun_out/pmc_ntt/$tag.err
  tail -2 gpurun_out/pmc_ntt/$tag.err
done
python3 - <<'PY'
import csv,glob
acc={}
for path in glob.glob("gpurun_out/pmc_ntt/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if 'ntt_pass' not in r["Kernel_Name"]: continue
        k=r["Counter_Name"]; d=acc.setdefault(k,{}); d[r["Dispatch_Id"]]=d.get(r["Dispatch_Id"],0.0)+float(r["Counter_Value"])
for k,d in sorted(acc.items()): print(f"{k:26s} n={len(d):3d} avg={sum(d.values())/len(d):.4g}")
PY
