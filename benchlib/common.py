"""benchlib.common -- constants and host-side helpers of bench.py: the synthetic inputs BASELINE.md prescribes, what the box looks
like (CPU model, usable cores, the STREAM-style copy peak, the sysfs clock), and the identity a line is stamped with."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy peak
HBM_COPY_GBS = 6290.0
# v_mad_u64_u32 (the only wide integer multiply): 2.30 ns per wave-instruction per SIMD measured with 2 and 4 waves per
# SIMD (scripts/ubench/valu_rates.hip -> profiles/ubench_valu_rates.txt) => 1024 SIMDs / 2.30 ns = 445 G mads/s.
# For scale: the guide's full-rate figure for simple VALU (wave64 v_fma_f32 in 2 cycles on a SIMD-32 once >= 2 waves
# share the SIMD) is 1024 x 2.4 GHz / 2 = 1228.8 G wave-instructions/s; integer multiplies do not issue at that rate.
MAD_NS = 2.30                  # ns per v_mad_u64_u32 wave-instruction per SIMD at >= 2 waves/SIMD (4.5 ns for a lone wave)
SIMDS = 1024
VALU_FULL_RATE_GINST_S = SIMDS * 2.4 / 2
TAU = 0x2F6C7A1D3B5E9F80412D6A7C93E1B5F7086A4D2C1E9B3F5A7D6C8E0F1A2B3C4D % R_MOD



def uniform_fr(n, seed):
    """n scalars uniform in [0, r): seeded PCG64 stream, 255-bit candidates, rejection of values >= r."""
    import numpy as np

    rng = np.random.default_rng(seed)
    r_words = np.array([(R_MOD >> (64 * (3 - i))) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    out = []
    have = 0
    while have < n:
        m = int((n - have) * 1.15) + 64
        raw = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        raw[:, 0] &= 0x7F
        w = raw.view(">u8").astype(np.uint64)
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for i in range(4):
            lt |= eq & (w[:, i] < r_words[i])
            eq &= w[:, i] == r_words[i]
        keep = raw[lt]
        out.append(keep)
        have += len(keep)
    return np.concatenate(out)[:n].tobytes()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def measured_copy_peak_gbs(torch):
    """STREAM-style device copy (1 GiB read + 1 GiB written per pass) on torch's stream: the achievable-HBM yardstick
    SURVEY 8d asks for beside the 8 TB/s spec figure."""
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return gbs


def sysfs_sclk_mhz(device):
    """Current shader clock from the amdgpu sysfs node (the `*` line of pp_dpm_sclk), when the container exposes it."""
    import glob

    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        with open(cards[device]) as f:
            for line in f:
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("M")[0])
    except (OSError, IndexError, ValueError):
        pass
    return None


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def thread_counts(user, usable=None):
    """CPU-baseline thread counts: 1, the box's share for one GPU (16) and every core this process may actually use
    (`usable`: the affinity mask cut down to the cgroup's CPU quota -- a box that SHOWS 256 cores but grants 16 has 16)."""
    if user:
        return sorted({1, user})
    n = usable or host_cores()
    return sorted({1, min(16, n), n})


def pctl(xs, q):
    s = sorted(xs)
    return s[min(len(s) - 1, int(q * len(s)))]


def errmsg(e):
    return f"{type(e).__name__}: {e}"[:400]


def flush_c_stdio():
    """RCCL writes its banner through C stdio, which is flushed at exit: push it out so that a JSON line printed next is
    the LAST line of stdout."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def visible_gpus():
    """Devices a rank could be given, counted in a throw-away child (`torch.cuda.device_count()` honours
    HIP_/ROCR_/CUDA_VISIBLE_DEVICES and does not create a HIP context); None when the count cannot be taken."""
    import subprocess

    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return None


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SOURCE_GLOBS = ("bench.py", "benchlib/*.py", "__graft_entry__.py", "include/*.h", "oracle/*.py", "oracle/*.c", "oracle/Makefile",
                "zkp_subnet_amd/*.py", "zkp_subnet_amd/csrc/*")


def source_sha16(root=ROOT):
    """Content identity of the tree a line was measured on, computable where there is no .git (the GPU boxes receive a
    snapshot without it): sha256 over (relative path, sha256 of the file) of every product / oracle / bench source."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for pat in SOURCE_GLOBS:
        for path in sorted(glob.glob(os.path.join(root, pat))):
            if os.path.isfile(path) and not path.endswith((".so", ".o", ".pyc")):
                with open(path, "rb") as f:
                    h.update(os.path.relpath(path, root).encode() + b"\0" + hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def identity():
    """Who measured this line: library version, git head (when the tree has a .git: a gpurun snapshot has none, and a
    side file would go stale -- the content hash below is the identity that always exists), sha256[:16] of bench.py and of
    the whole source set.  scripts/evidence_keep.py refuses a file whose source_sha16 is not
    the tree's; no two rounds' evidence files can be byte-identical."""
    import hashlib
    import subprocess

    from zkp_subnet_amd import _native

    head = None
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20)
        if r.returncode == 0:
            head = r.stdout.strip()
            d = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                               text=True, timeout=20)
            if d.returncode == 0 and d.stdout.strip():
                head += "+dirty"
    except (OSError, subprocess.TimeoutExpired):
        pass
    with open(os.path.join(ROOT, "bench.py"), "rb") as f:
        bsha = hashlib.sha256(f.read()).hexdigest()[:16]
    return {"lib_version": _native.load().kzg_version().decode(), "git_head": head, "bench_py_sha16": bsha,
            "source_sha16": source_sha16(), "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
