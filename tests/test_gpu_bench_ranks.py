"""The real multi-rank flow of bench.py (shard -> broadcast identity -> per-rank engine -> chunked gather -> max-over-ranks timing)
with two ranks, each with its own engine, sharing the one GPU of the test box (collectives on gloo / host): a rank-local
cs_create, device-index or gather-order bug shows up here.  The fixed-size job (BASELINE configs[3] shape: --frames) must give
the same bytes for every frame as the single-process run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(cmd, tmp):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_two_rank_fixed_size_job_equals_single_process(tmp_path):
    common = ["--frames", "18", "--batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    crc1, crc2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    one = _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", crc1], tmp_path)
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", *common, "--dump-crc", crc2], tmp_path)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for line in (one, two):
        assert line["scaling"] == "strong" and line["config"]["frames_total"] == 18 and line["unit"] == "frames/s"
        assert line["roofline"]["frac"] > 0 and line["value"] > 0
    a, b = json.load(open(crc1)), json.load(open(crc2))
    assert len(a) == 18 and a == b                      # ranks own frames [0, 9) and [9, 18): same bytes, same order
    assert len(set(a)) == 16 and a[16:] == a[:2]        # the input pool holds 4 x batch = 16 distinct frames, cycled
