"""N > 1 path of bench.py on CPU: two gloo processes launched exactly like the driver launches the
GPU bench (torch.distributed.run, one rank per device, 127.0.0.1 rendezvous).  The hot path does
not shard (replicas only), so what is covered is the replica plumbing: rendezvous,
barrier, max-over-ranks step time, whole-job aggregate, single JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_replicas_gloo():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1", "--dist-selftest"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout             # only rank 0 prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 10 and r["scaling"] == "weak"
    # rank 1 needs 2 ms per token -> job time is the max; 2 replicas x 10 tokens / 20 ms
    assert abs(r["ms_per_step"] - 2.0) < 1e-6
    assert abs(r["value"] - 1000.0) < 1e-3


def test_single_process_selftest():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--dist-selftest"],
                         capture_output=True, text=True, timeout=120, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 1 and abs(r["value"] - 1000.0) < 1e-3
