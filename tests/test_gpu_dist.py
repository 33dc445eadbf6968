"""N > 1 launch path with REAL kernels: two ranks (one process each, both on this box's single GPU, `gloo` for the record
gather -- ISAC_DIST_BACKEND is bench.py's test hook; on a multi-GPU node the same code runs one rank per GPU over RCCL)
process BASELINE configs[2]'s 7 cells, cell c -> rank c mod 2 (networkSimulation.m:44-60: one worker per cell, no
inter-cell coupling).  The gathered per-cell records must equal a single-rank run of the same 7 cells field for field."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_seven_cells_match_single_rank():
    common = ["--cells", "7", "--steps", "1", "--warmup", "0", "--prime-ms", "0", "--no-cpu-baseline", "--inflight", "2", "--slots", "4", "--ants", "16"]
    one = _run([sys.executable, "bench.py", "--gpus", "1", *common])
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "2", *common], env={"ISAC_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["scaling"] == two["scaling"] == "strong"
    assert [c["cell"] for c in one["cells"]] == list(range(7)) == [c["cell"] for c in two["cells"]]
    assert one["cells"] == two["cells"]                              # every field of every cell's record, bit for bit
    assert sum(c["nRng"] is not None for c in one["cells"]) >= 5     # the cells actually detect their targets
    assert two["value"] > 0 and two["steps"] == 1


def test_gpus_flag_spawns_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment must launch two ranks on its own."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ISAC_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--prime-ms", "0", "--no-cpu-baseline", "--slots", "4",
                        "--ants", "16", "--inflight", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and [c["cell"] for c in res["cells"]] == [0, 1]
