"""Counts the v_mad_u64_u32 (and all VALU) instructions on the common path of ONE mixed point addition in
k_msm_accumulate, from the gfx950 ISA hipcc emits for csrc/msm_accumulate.hip (no GPU needed).  The hot loop's body is
[loop header .. the `g1_madd_checked` exit label]; the rare doubling branch (equal x coordinates: g1_dbl_aff inlined,
the block between the second-level `s_cbranch_execz` pair) is excluded.  Writes profiles/isa_counts.json.

    python scripts/count_mads.py            # rewrite profiles/isa_counts.json
    python scripts/count_mads.py --check    # exit 1 if the committed file no longer matches the current source
(bench.py quotes `mad_issue.wave_mads_per_launch` from that file; tests/test_host_logic.py runs --check so it cannot go stale)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "zkp_subnet_amd", "csrc", "msm_accumulate.hip")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-c", src, "-o",
                    os.path.join(td, "msm.o"), "-save-temps=obj"], check=True, cwd=td, capture_output=True)
    asm = open(os.path.join(td, "msm_accumulate-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r"^_Z16k_msm_accumulate.*?s_endpgm", asm, re.S | re.M)
lines = m.group(0).splitlines()
# the loop: from the line after the exit label of the previous iteration's madd to that label again
hdr = next(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l)
end = max(i for i, l in enumerate(lines) if l.startswith(".LBB") and "Flow" in l and i > hdr and
          any("s_cbranch_execz " + l.split(":")[0] in x for x in lines[hdr:i]))
body = lines[hdr:]
# rare branch = the longest run between an `s_cbranch_execz .LBBx` and its target label that contains > 2000 mads
mads = [i for i, l in enumerate(body) if "v_mad_u64_u32" in l]
best = (0, 0)
for i, l in enumerate(body):
    mm = re.match(r"\s*s_cbranch_execz (\.LBB\d+_\d+)", l)
    if not mm:
        continue
    tgt = next((j for j in range(i, len(body)) if body[j].startswith(mm.group(1) + ":")), None)
    if tgt is None:
        continue
    inside = sum(1 for k in mads if i < k < tgt)
    total_after = sum(1 for k in mads if k > i)
    # the doubling branch: a skipped region holding ~2.4k mads that is NOT the whole rest of the loop
    if 2000 < inside < 3000 and inside > best[1] - best[0] and total_after - inside > 1500:
        best = (i, tgt)
rare = sum(1 for k in mads if best[0] < k < best[1])
common = len(mads) - rare
valu = sum(1 for i, l in enumerate(body) if re.match(r"\s*v_", l) and not (best[0] < i < best[1]))
out = {"kernel": "k_msm_accumulate", "mads_per_mixed_add": common, "mads_in_rare_doubling_branch": rare,
       "valu_per_loop_iteration_static": valu,
       "source": "hipcc --offload-arch=gfx950 -O3 -save-temps of zkp_subnet_amd/csrc/msm_accumulate.hip (scripts/count_mads.py)"}
path = os.path.join(ROOT, "profiles", "isa_counts.json")
if "--check" in sys.argv:
    have = json.load(open(path))
    keys = ("kernel", "mads_per_mixed_add", "mads_in_rare_doubling_branch")
    diff = {k: (have.get(k), out[k]) for k in keys if have.get(k) != out[k]}
    print(json.dumps({"committed": have, "current": out, "stale": diff}))
    sys.exit(1 if diff else 0)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
