"""Copies measured JSON lines from gpurun_out/ into profiles/ -- and REFUSES any file that does not identify itself as a
measurement of THIS tree (VERDICT r4 task 4: an "r04" evidence file turned out to be round 3's, byte for byte).

    python scripts/evidence_keep.py <src.json> <profiles/rNN_name.json> [more pairs ...]

A bench line carries `identity` = {lib_version, git_head, bench_py_sha16, source_sha16, utc} (bench.py identity()); the
file is kept only if identity.source_sha16 equals bench.source_sha16() of the tree it is copied into (the GPU boxes have no
.git, so the content hash of the source set is the identity that always exists; git_head is checked too when both sides
know it).  A kept file is also checked against every other JSON under profiles/: two evidence files with equal bytes are an
error.  Exit code 1 on any refusal; nothing is copied for a refused pair."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16  # noqa: E402


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip().startswith("{")]
    if not lines:
        raise ValueError("no JSON line")
    return lines[-1], json.loads(lines[-1])


def tree_head():
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20)
        return r.stdout.strip() if r.returncode == 0 else None
    except OSError:
        return None


def main(argv):
    if len(argv) < 2 or len(argv) % 2:
        print(__doc__)
        return 2
    want, head, bad = source_sha16(ROOT), tree_head(), 0
    existing = {}
    for dirpath, _, files in os.walk(os.path.join(ROOT, "profiles")):
        for f in files:
            if f.endswith(".json"):
                p = os.path.join(dirpath, f)
                existing.setdefault(hashlib.md5(open(p, "rb").read()).hexdigest(), []).append(os.path.relpath(p, ROOT))
    for src, dst in zip(argv[0::2], argv[1::2]):
        try:
            line, rec = last_json_line(src)
        except (OSError, ValueError) as e:
            print(f"REFUSED {src}: {e}")
            bad += 1
            continue
        ident = rec.get("identity") or {}
        if ident.get("source_sha16") != want:
            print(f"REFUSED {src}: measured on source set {ident.get('source_sha16')}, this tree is {want} "
                  f"(git_head of the line: {ident.get('git_head')})")
            bad += 1
            continue
        lh = (ident.get("git_head") or "").replace("+dirty", "")
        if head and lh and lh != head:
            print(f"REFUSED {src}: measured at git {lh[:12]}, the tree is at {head[:12]}")
            bad += 1
            continue
        digest = hashlib.md5((line + "\n").encode()).hexdigest()
        twins = [p for p in existing.get(digest, []) if os.path.abspath(os.path.join(ROOT, p)) != os.path.abspath(dst)]
        if twins:
            print(f"REFUSED {src}: byte-identical to {twins}")
            bad += 1
            continue
        with open(dst, "w") as f:
            f.write(line + "\n")
        print(f"kept {os.path.relpath(dst, ROOT)}  (source {want}, git {lh[:12] or 'n/a'}, {ident.get('utc')})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
