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
