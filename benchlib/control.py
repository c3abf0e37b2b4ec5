"""benchlib.control -- the control plane of a multi-rank bench launch: fault injection for the tests, barriers / MAX over ranks /
phase agreement through the rendezvous store (`Ctl`), the choice of the data-path collective with its preflight
(`make_collective`), and the launcher-less `python bench.py --gpus N` (`self_launch`).  torch.distributed is the launcher and
the control plane only; the data path is the library's own RCCL collective."""
import json
import os
import sys
import time

from .common import *  # noqa: F401,F403
from .common import errmsg, free_port, visible_gpus

class Fault(Exception):
    """BENCH_FAULT=<phase>:<rank> -- an injected failure (tests of the multi-rank error paths)."""


def inject(phase, rank):
    spec = os.environ.get("BENCH_FAULT", "")
    for item in spec.split(","):
        if item and item.split(":")[0] == phase and int(item.split(":")[1]) == rank:
            raise Fault(f"injected fault in {phase} on rank {rank} (BENCH_FAULT)")


class Ctl:
    """Control plane of a multi-rank launch.  Collectives (barrier, MAX, byte gathers) run on the default process group
    -- gloo with a timeout unless BENCH_BACKEND says otherwise; phase STATUS travels through the rendezvous store (set /
    wait / get with a timeout), never through a collective: a rank that failed its phase cannot be waited for in one."""

    def __init__(self, torch, dist, rank, world, active, timeout_s):
        self.torch, self.dist, self.rank, self.world, self.active, self.timeout_s = torch, dist, rank, world, active, timeout_s
        self.poisoned = None         # set once a collective of the group may have been left half-done
        self.store = None
        if active:
            try:
                from torch.distributed.distributed_c10d import _get_default_store

                self.store = _get_default_store()
            except Exception:        # noqa: BLE001 -- older / newer torch: agree() then uses an object gather
                self.store = None
        self.cpu = not active or dist.get_backend() != "nccl"

    def _dev(self):
        return "cpu" if self.cpu else "cuda"

    def barrier(self):
        if self.active:
            self.dist.barrier()
        if self.torch.cuda.is_available():      # (always, in a measurement; the CPU tests of this class have no device)
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.active:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_bytes(self, b):
        if not self.active:
            return [bytes(b)]
        src = self.torch.frombuffer(bytearray(b), dtype=self.torch.uint8).to(self._dev())
        out = self.torch.empty(self.world * len(b), dtype=self.torch.uint8, device=self._dev())
        self.dist.all_gather_into_tensor(out, src)
        raw = out.cpu().numpy().tobytes()
        return [raw[i * len(b):(i + 1) * len(b)] for i in range(self.world)]

    def agree(self, phase, ok, msg=""):
        """Every rank reports (ok, msg) for `phase`; returns (all ok, {rank: msg of the failed ones}).  A rank that does not
        report within the timeout counts as failed."""
        if not self.active:
            return ok, ({} if ok else {self.rank: msg})
        import datetime

        if self.store is None:
            objs = [None] * self.world
            self.dist.all_gather_object(objs, (bool(ok), str(msg)[:400]))
            bad = {r: m for r, (k, m) in enumerate(objs) if not k}
            return not bad, bad
        self.store.set(f"bench/{phase}/{self.rank}", json.dumps([bool(ok), str(msg)[:400]]))
        bad = {}
        for r in range(self.world):
            key = f"bench/{phase}/{r}"
            try:
                self.store.wait([key], datetime.timedelta(seconds=self.timeout_s))
                k, m = json.loads(self.store.get(key).decode())
                if not k:
                    bad[r] = m
            except Exception as e:   # noqa: BLE001 -- no status from that rank: it is gone or stuck
                bad[r] = f"no status within {self.timeout_s} s ({type(e).__name__})"
        return not bad, bad


def make_collective(args, ctl, eng, torch, dist, rank, world):
    """The data-path collective of the SRS-sharded MSM, decided ONCE per launch, before any table is built.
    Preferred: the library's own (kzg_comm_init + a checked all_gather; both under a watchdog).  If ANY rank fails that
    preflight, every rank uses the process group's all_gather instead (`DeviceGather`: gloo moves the device tensors
    through the host; with BENCH_BACKEND=nccl it is torch's RCCL group) and the line records why.
    Returns (gather object with .msm(slot, n, offset), description dict)."""
    from zkp_subnet_amd.distributed import DeviceGather, LibraryGather

    want = os.environ.get("BENCH_COLLECTIVE", "library")
    backend = dist.get_backend()
    info = {"preferred": want, "process_group_backend": backend}
    if want == "library":
        g, err = None, ""
        try:
            inject("comm_init", rank)
            t0 = time.time()
            g = LibraryGather(eng, timeout_ms=int(os.environ.get("BENCH_COMM_TIMEOUT_MS", "120000")),
                              init_timeout_s=float(os.environ.get("BENCH_COMM_INIT_TIMEOUT_S", "120")))
            eng.comm_selftest()
            info["comm_init_s"] = round(time.time() - t0, 2)
        except BaseException as e:       # noqa: BLE001 -- including a TimeoutError of the init watchdog
            err = errmsg(e)
        ok, bad = ctl.agree("collective_preflight", not err, err)
        if ok:
            ci = eng.comm_info()
            info.update({"collective": "library: ncclAllGather of 192 B per rank on the lane's own stream (kzg_msm_sharded)",
                         "rccl_version": ci["rccl_version"], "rccl_binding": "dlopen(librccl.so.1) inside libkzg_mi355x.so"})
            return g, info
        info["library_preflight_failed"] = {str(r): m for r, m in sorted(bad.items())}
        if g is not None and not err:
            try:
                g.close()                # healthy here, unusable elsewhere: drop it
            except Exception:            # noqa: BLE001
                pass
        first = next(iter(sorted(bad.items())))
        why = f"library RCCL preflight failed on rank {first[0]}: {first[1]}"
        if backend == "nccl":
            info.update({"collective": f"torch.distributed all_gather over torch's RCCL group ({why})",
                         "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())})
        else:
            info.update({"collective": f"{backend} fallback ({why})", "rccl_version": f"none ({backend} fallback: {why})"})
        # the library bounds the rendezvous itself (kzg_comm_init_bounded): the engine stays usable after a timeout, but a
        # helper thread may still sit inside RCCL's bootstrap -- this process then leaves through os._exit after its line
        left = bool(err) and "did not all join" in err
        return DeviceGather(eng), dict(info, comm_init_helper_left_behind=left)
    if backend == "nccl":
        info.update({"collective": "torch.distributed all_gather over torch's RCCL group, chained through streams (A/B form)",
                     "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())})
    else:
        info.update({"collective": f"torch.distributed all_gather over {backend} (self-test form)",
                     "rccl_version": f"none ({backend} self-test)"})
    return DeviceGather(eng), info


def self_launch(n):
    """The N > 1 launch line of the bench contract (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same args>`) run as a child process under a watchdog; returns its exit
    code.  The ranks' stdout is relayed line by line with the JSON lines held back so that the LAST one ends the output;
    their stderr passes through and its tail is repeated when the launch fails or is killed."""
    import collections
    import subprocess
    import threading

    one_gpu = os.environ.get("BENCH_ONE_GPU") == "1"       # self-test: every rank on device 0 (gloo)
    have = 1 if one_gpu else visible_gpus()
    if not one_gpu and have is not None and n > have:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible to this process "
              "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES respected)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
    limit = float(os.environ.get("BENCH_WATCHDOG_S", "1500"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True, bufsize=1)
    tail = collections.deque(maxlen=60)

    def relay_err():
        for line in proc.stderr:
            tail.append(line)
            sys.stderr.write(line)

    state = {"last_json": None}

    def relay_out():
        for line in proc.stdout:
            if line.lstrip().startswith("{") and '"metric"' in line:
                state["last_json"] = line                  # held back: printed after everything else the ranks wrote
            else:
                sys.stdout.write(line)

    threads = [threading.Thread(target=relay_err, daemon=True), threading.Thread(target=relay_out, daemon=True)]
    for t in threads:
        t.start()
    killed = False
    try:
        try:
            rc = proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            killed = True
            print(f"bench.py: the launch did not finish within {limit:.0f} s (BENCH_WATCHDOG_S): terminating it",
                  file=sys.stderr)
            proc.terminate()                               # the exact child we started, never a pattern
            try:
                rc = proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                proc.kill()
                rc = proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise
    for t in threads:
        t.join(timeout=10)
    last_json = state["last_json"]
    if rc != 0 or killed:
        print(f"bench.py: the launch ended with code {rc}" + (" (killed by the watchdog)" if killed else "")
              + "; last lines of the ranks' stderr:", file=sys.stderr)
        sys.stderr.write("".join(list(tail)[-25:]))
    if last_json is not None:
        sys.stdout.write(last_json if last_json.endswith("\n") else last_json + "\n")
    sys.stdout.flush()
    if killed:
        return rc if rc not in (0, None) else 124
    if rc == 0 and last_json is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 3
    return rc
