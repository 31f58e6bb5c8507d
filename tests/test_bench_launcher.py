"""bench.py's own rank launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver starts
the scaling runs) must start N rank processes before any GPU call, relay rank 0's ONE JSON line and fail loudly -- not at
an assert in the parent -- when the ranks cannot run.  CPU-only: the rank plumbing runs on gloo (--selftest-launcher);
the real workload is refused in every child with 'needs an MI355X'."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""   # also on a GPU box this test stays off the card
    return env


def test_parent_spawns_ranks_and_relays_rank0_json():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=_clean_env(), capture_output=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out == {"selftest": "launcher", "n_gpus": 2, "max_t": 2.0, "sum": 300.0}   # world size from the process group


def test_ranks_fail_loudly_without_a_gpu_not_at_a_parent_assert():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, timeout=300)
    err = p.stderr.decode()
    assert p.returncode != 0
    assert err.count("needs an MI355X") >= 1 and "AssertionError" not in err
    assert p.stdout.decode().strip() == ""


def test_world_size_mismatch_is_reported():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr.decode()


def _session_alive(pgid):
    """Any RUNNING process left in the process group a rank led?  (Orphans that were killed stay behind as zombies where the
    container's pid 1 does not reap; a zombie holds nothing.)"""
    for pid in os.listdir("/proc"):
        if not pid.isdigit():
            continue
        try:
            fields = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()
        except OSError:
            continue
        if int(fields[2]) == pgid and fields[0] != "Z":   # after "comm)": state, ppid, pgrp
            return True
    return False


def _hung_ranks(tmp_path, extra_env):
    import time
    tag = str(tmp_path / "rank")
    env = dict(_clean_env(), PIME_SELFTEST_HANG=tag, **extra_env)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE)
    t_end = time.time() + 120
    while time.time() < t_end and not all(os.path.exists(f"{tag}.{r}") for r in (0, 1)):
        time.sleep(0.1)
    pids = [int(open(f"{tag}.{r}").read()) for r in (0, 1)]
    assert all(_session_alive(pid) for pid in pids)
    return p, pids


def test_sigterm_to_the_parent_leaves_no_rank_behind(tmp_path):
    """a SIGTERMed parent (driver time-limit kill) must take its ranks -- and their children -- with it."""
    import signal
    import time
    p, pids = _hung_ranks(tmp_path, {})
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    time.sleep(0.2)
    assert not any(_session_alive(pid) for pid in pids), "a rank (or a child of one) survived the parent"


def test_deadline_stops_hung_ranks(tmp_path):
    import time
    p, pids = _hung_ranks(tmp_path, {"PIME_BENCH_DEADLINE_S": "3"})
    assert p.wait(timeout=60) == 124
    assert b"deadline reached" in p.stderr.read()
    time.sleep(0.2)
    assert not any(_session_alive(pid) for pid in pids)


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver starts it, rehearsed on a one-GPU box: PIME_BENCH_REHEARSE=1 puts both
    ranks on cuda:0 with gloo collectives (RCCL refuses two ranks on one device).  Exercises the launcher, the sharded lanes,
    the weight broadcast, the per-step flat-gradient all-reduce between the gradient launches and Adam, the max-over-ranks
    timing and rank 0's JSON line -- everything of the N > 1 path but RCCL itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PIME_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 1e6 and abs(out["value"] - 2 * 819200 / (out["ms_per_step"] * 1e-3)) < 1e-3 * out["value"]
