"""The hand-written one-shot all-reduce (csrc/allreduce.hip) between TWO PROCESSES that share cuda:0: each rank
exports its inbox region (hipIpcGetMemHandle), maps its peer's (hipIpcOpenMemHandle), and every call is one kernel launch per
rank -- push into the peer's inbox, flag, wait, sum in rank order.  Checked bit for bit against gloo's all-reduce of the same
vectors (11 calls: eager on both parities and replayed from a captured HIP graph), and that every rank ends with the SAME bits.
What this cannot show on a one-GPU box: visibility of the pushed rows across DEVICES over xGMI (fine-grained memory); hence the
path is opt-in (PIME_ONESHOT_ALLREDUCE=1) and RCCL stays the default."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n", [67460, 1000])     # the pH nets' flat gradient (270 KB); a size that is no multiple of anything
def test_two_processes_on_one_gpu_bit_equal_to_gloo(tmp_path, n):
    world, port = 2, _free_port()
    out = str(tmp_path / "ar")
    procs = []
    for r in range(world):   # fresh children: nothing has touched the GPU before they start
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oneshot_worker.py"), out, str(n)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n---\n".join(logs)
    res = [torch.load(f"{out}.{r}.pt", weights_only=True) for r in range(world)]
    for r in range(world):
        assert res[r]["status"] == 0, f"rank {r}: a peer did not arrive (status {res[r]['status']})"
        assert len(res[r]["got"]) == 11
        for k, (g, w) in enumerate(zip(res[r]["got"], res[r]["want"])):
            assert torch.equal(g, w), f"rank {r}, call {k}: one-shot all-reduce differs from gloo (max {float((g - w).abs().max())})"
    for k in range(11):
        assert torch.equal(res[0]["got"][k], res[1]["got"][k]), "the ranks must end with identical bits"


def _run_dp(tmp_path, tag, oneshot):
    world, port = 2, _free_port()
    out = str(tmp_path / tag)
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k != "PIME_ONESHOT_ALLREDUCE"}
        env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if oneshot:
            env["PIME_ONESHOT_ALLREDUCE"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oneshot_dp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n---\n".join(logs)
    return [torch.load(f"{out}.{r}.pt", weights_only=True) for r in range(world)]


def test_data_parallel_update_with_the_oneshot_allreduce(tmp_path):
    """Two data-parallel ranks (each its own lane slice, both on cuda:0) run three rollouts + updates of the bench's agent with
    the per-optimizer-step flat-gradient all-reduce done by the one-shot kernel, captured INSIDE the step graphs: the replicas
    end bit-identical, and bit-equal to the same run with gloo's all-reduce (two ranks: a + b is one rounding either way)."""
    fast = _run_dp(tmp_path, "oneshot", True)
    ref = _run_dp(tmp_path, "gloo", False)
    assert all(r["oneshot"] and r["status"] == 0 and r["in_graph"] for r in fast), [(r["oneshot"], r["status"], r["in_graph"]) for r in fast]
    assert not any(r["oneshot"] for r in ref)
    assert torch.equal(fast[0]["flat"], fast[1]["flat"]), "replicas diverged under the one-shot all-reduce"
    assert torch.equal(fast[0]["flat"], ref[0]["flat"]), "one-shot and gloo all-reduce give different weights"
    assert torch.isfinite(fast[0]["flat"]).all()


def test_a_peer_that_never_arrives_is_fatal(tmp_path):
    """a one-shot all-reduce whose peer does not show up used to leave the gradient silently un-averaged (the kernel gives
    up after ~2 s and only sets a status word nobody read).  DataParallel.check(), which update_net calls where it synchronises
    anyway, now raises on that status: the rank exits non-zero and says why."""
    world, port = 2, _free_port()
    out = str(tmp_path / "absent")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oneshot_worker.py"), out, "4096", "absent"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode()[-3000:])
    assert procs[1].returncode == 0, logs[1]
    assert procs[0].returncode != 0, "rank 0 must fail loudly when its peer never pushes:\n" + logs[0]
    assert "one-shot all-reduce failed" in logs[0] and "replicas have diverged" in logs[0], logs[0]
