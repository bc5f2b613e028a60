"""the hand-written reduce-scatter / all-gather kernels behind nvit_amd.xgmi.XgmiAllReduce, exercised by
2 and 4 fresh processes that share the box's one MI355X (IPC-mapped symmetric buffers and uncached flag blocks; on a
multi-GPU node the same peer reads and flag stores travel over xGMI).  Phases are separated by device-side flags only
(no host barrier inside a collective): ranks arrive skewed, and in the multi-slot case every reduce-scatter of four
regions is enqueued before any all-gather, on two streams.  Sums must equal the rank-ordered reference and be
bit-identical on every rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(world, tmp_path, *extra):
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = tmp_path / f"x_{rank}.json"
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "xgmi_worker.py"), str(out), *extra], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    if any(p.returncode != 0 for p in procs):
        raise AssertionError("\n".join(f"--- rank {i} rc={p.returncode}\n{o[-1500:]}" for i, (p, o) in enumerate(zip(procs, logs))))
    return [json.load(open(out)) for out in outs]


def test_xgmi_timeout_is_fatal_on_every_rank_and_fails_closed(tmp_path):
    """A rank that arrives later than the bound (1 s here; 1800 s by default): the waiting rank gives up and poisons every
    rank's error word, the late rank's kernels find it, both raise from the pinned host word within the call, no buffer
    holds a partially reduced / gathered result, a second call returns at once and raises again, nothing hangs."""
    res = _spawn(2, tmp_path, "late")
    for r in res:
        assert r["raised"] and "timed out" in r["message"], r
        assert r["buffer_untouched"], r
        assert r["second_raised"] and r["second_seconds"] < 1.0, r
        assert r["check_error_raised"], r
        assert r["seconds"] < 10.0, r


@pytest.mark.parametrize("world", [2, 4])
def test_xgmi_all_reduce_shared_device(world, tmp_path):
    for r in _spawn(world, tmp_path):
        assert len(r["cases"]) == 4 and r["shared_device"]
        assert r["multi_slot"]["max_err"] < 1e-5 and r["multi_slot"]["bit_identical_across_ranks"], r["multi_slot"]
        for c in r["cases"]:
            assert c["max_err"] < 1e-5, c            # (the reference sums in the same order; fp32 rounding of the 1/world scale)
            assert c["bit_identical_across_ranks"] and c["padding_zero"], c
