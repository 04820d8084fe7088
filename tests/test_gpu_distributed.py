"""
The distributed decomposition (localmd_decomposition(distributed=True), one process per rank) against the single-rank
result, run as fresh child processes: 2 ranks over gloo sharing the one GPU of the test box (the RCCL branch of the
collectives needs one GPU per rank and is exercised by `bench.py --gpus N` on a multi-GPU node).  Seven configurations
(scripts/dist_check.py): R <= frames, R > frames (row-sharded Cholesky route with the halo exchange of the right
matrix and, since round 3, the frames x frames products of the last stage sharded by frame columns), a frame subset with pixel
weights in C order, several temporal windows, denoiser hooks, rank_prune, and the generic-width tile path (max_components 80, every
component kept: virtual tiles of 64 component rows in the sharded global stage).
Tile ranks, CSR structure and the statistics images must equal the single-rank ones; floating-point results agree to
fp32 summation order.
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_matches_single_rank(gpu_ctx, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1", PYTHONFAULTHANDLER="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "dist_check.py")]
    # own process group + a hard limit: a rank that dies or a stuck collective must fail this test, not stall the suite
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        import signal

        os.killpg(proc.pid, signal.SIGKILL)
        stdout, _ = proc.communicate()
        pytest.fail("distributed run did not finish within 300 s:\n" + stdout[-4000:])
    print(stdout[-6000:])
    assert proc.returncode == 0, stdout[-3000:]
    assert stdout.count("ok=True") == 7 * world and "ok=False" not in stdout
