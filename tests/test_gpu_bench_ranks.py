"""The real multi-rank flow of bench.py (shard -> broadcast identity -> per-rank engine -> chunked gather -> max-over-ranks timing)
with several ranks, each with its own engine, sharing the one GPU of the test box (collectives on gloo / host): a rank-local
cs_create, device-index or gather-order bug shows up here.  The fixed-size job (BASELINE configs[3] shape: --frames) must give
the same bytes for every frame as the single-process run.  bench.py is started exactly as the driver starts it at N = 1
(`python bench.py --gpus N ...`): it spawns its own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # ONE JSON line, from rank 0
    assert r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-600:]      # ... and the last one (librccl's banner comes before it)
    return json.loads(lines[0])


def test_two_rank_fixed_size_job_equals_single_process(tmp_path):
    common = ["--frames", "18", "--batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    crc1, crc2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    one = _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", crc1])
    two = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", *common, "--dump-crc", crc2])      # self-spawned ranks
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["ranks_seen"] == 1 and two["ranks_seen"] == 2 and len(two["devices"]) == 2
    for line in (one, two):
        assert line["scaling"] == "strong" and line["config"]["frames_total"] == 18 and line["unit"] == "frames/s"
        assert line["roofline"]["frac"] > 0 and line["value"] > 0
    a, b = json.load(open(crc1)), json.load(open(crc2))
    assert len(a) == 18 and a == b                      # ranks own frames [0, 9) and [9, 18): same bytes, same order
    assert len(set(a)) == 16 and a[16:] == a[:2]        # the input pool holds 4 x batch = 16 distinct frames, cycled


def test_two_identities_are_indexed_by_global_frame(tmp_path):
    """with several identities the identity of a frame follows its GLOBAL index, so a fixed-size job gives the same bytes
    on any rank count."""
    common = ["--frames", "10", "--batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--identities", "3"]
    crc1, crc2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", crc1])
    _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", *common, "--dump-crc", crc2])
    a, b = json.load(open(crc1)), json.load(open(crc2))
    assert len(a) == 10 and a == b


def test_streams_on_rank_pairs_equal_streams_hosted_on_one_rank(tmp_path):
    """BASELINE configs[4] placement: 2 streams on 4 ranks (stream s -> ranks {2s, 2s+1}, per-stream sub-communicator
    gather, identity slot 0 on each rank) against the same 2 streams hosted on ONE rank (frames interleaved in one launch, one
    identity slot per stream: the per-sample modulated convolution of adaptive_modulate.py:157-167).  Same bytes per stream frame,
    in frame order; the two streams differ (own identity, own video)."""
    common = ["--streams", "2", "--frames", "6", "--batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    c1, c4 = str(tmp_path / "one"), str(tmp_path / "four")
    one = _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", c1])
    four = _run([sys.executable, "bench.py", "--gpus", "4", "--backend", "gloo", *common, "--dump-crc", c4])
    assert four["ranks_seen"] == 4 and [s["ranks"] for s in four["streams"]] == [[0, 1], [2, 3]]
    assert [s["ranks"] for s in one["streams"]] == [[0], [0]]
    assert four["config"]["frames_total"] == 12 and all(s["value"] > 0 for s in four["streams"])
    for s in range(2):
        a, b = json.load(open(f"{c1}.s{s}")), json.load(open(f"{c4}.s{s}"))
        assert len(a) == 6 and a == b, s
    assert json.load(open(f"{c1}.s0")) != json.load(open(f"{c1}.s1"))


def test_rccl_world_of_one_runs_every_collective_and_equals_the_plain_run(tmp_path):
    """the nccl (= RCCL) branch of bench.py on the one GPU a box has.  --force-dist forms the communicator with
    device_id, broadcasts the identity, builds ChunkedFrameGather on the DEVICE and pushes the real out_u8 chunks through asynchronous
    dist.gather inside the timed region (ragged: 18 frames in chunks of 4 -> the padded last chunk and the torch.cat path; exact:
    16 frames -> the view path).  Same bytes per frame as the run without a process group."""
    for frames in (18, 16):
        common = ["--frames", str(frames), "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
        crc1, crc2 = str(tmp_path / f"plain{frames}.json"), str(tmp_path / f"rccl{frames}.json")
        one = _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", crc1])
        two = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", *common, "--dump-crc", crc2])
        assert one["ranks_seen"] == 1 and one["backend"] is None
        assert two["ranks_seen"] == 1 and two["backend"] == "nccl" and "gather" in two["collectives"]
        a, b = json.load(open(crc1)), json.load(open(crc2))
        assert len(a) == frames and a == b


def test_rccl_world_of_one_weak_scaling_line_and_stream_subcommunicators(tmp_path):
    """The default (weak-scaling) line with its fixed-size job through RCCL at N = 1, and --streams 2 with dist.new_group sub-communicators
    (one per stream, formed and barriered on the device) against the same streams without a process group."""
    line = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert line["backend"] == "nccl" and line["scaling"] == "weak" and line["fixed_job"]["frames"] == 300 and line["value"] > 0
    common = ["--streams", "2", "--frames", "6", "--batch", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    c1, c2 = str(tmp_path / "plain"), str(tmp_path / "rccl")
    _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", c1])
    two = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", *common, "--dump-crc", c2])
    assert two["backend"] == "nccl"
    for s in range(2):
        assert json.load(open(f"{c1}.s{s}")) == json.load(open(f"{c2}.s{s}"))


def test_configs3_job_shape_through_rccl_equals_the_plain_run(tmp_path):
    """the BASELINE configs[3] job at its real size on the one GPU a box has - 1200 frames in launches of 64 (19 chunks, the
    last one ragged), the 943 MB receive buffer on the device, every chunk pushed through an asynchronous RCCL gather inside the timed region -
    CRC-equal, frame by frame, to the run without a process group.  The two frames/s figures (with and without the gather on the leader) are
    printed for profiles/."""
    common = ["--frames", "1200", "--batch", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    crc1, crc2 = str(tmp_path / "plain.json"), str(tmp_path / "rccl.json")
    one = _run([sys.executable, "bench.py", "--gpus", "1", *common, "--dump-crc", crc1])
    two = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", *common, "--dump-crc", crc2])
    assert two["backend"] == "nccl" and "gather" in two["collectives"] and two["config"]["frames_total"] == 1200
    a, b = json.load(open(crc1)), json.load(open(crc2))
    assert len(a) == 1200 and a == b
    print(f"configs[3] shape on 1 GPU: {one['value']:.1f} frames/s plain, {two['value']:.1f} frames/s with the chunked RCCL gather into the "
          f"device receive buffer ({two['value'] / one['value']:.4f})")
