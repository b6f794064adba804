"""bench.py's own rank launcher (`python bench.py --gpus N` without a torchrun environment), on CPU: it must start N fresh rank
processes with a one-node torchrun environment on 127.0.0.1 BEFORE touching a GPU and relay rank 0's single JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=300)


def test_bench_spawns_its_ranks_and_relays_one_line():
    r = _run(["--gpus", "2", "--probe-launch", "--config", "C4"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ok"] is True
    # the label is derived from what ran: C4's per-GPU share on 2 GPUs is not "C4" (that is 8 GPUs), and never "C2"
    assert d["config"]["workload"].startswith("custom (C4's per-GPU share on 2 GPU)")
    assert "bs=32/GPU x 2 GPU" in d["config"]["workload"]


def test_workload_labels_follow_the_arguments():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.config_name(bench.parse([]), 1) == "C2"
    assert bench.config_name(bench.parse(["--batch", "32"]), 1).startswith("custom")          # C4's share, not C2
    assert bench.config_name(bench.parse(["--config", "C4"]), 8) == "C4"
    assert bench.config_name(bench.parse(["--config", "C3"]), 1) == "C3"
    assert bench.config_name(bench.parse(["--sam", "vit_h", "--with-msqp", "--batch", "32", "--seg-tokens", "14"]), 1) == "C3"
    assert bench.config_name(bench.parse(["--config", "C5", "--dtype", "fp8"]), 8) == "C5"
    assert bench.config_name(bench.parse(["--config", "C5"]), 1) == "C5 geometry with bf16 GEMMs"
    assert bench.config_name(bench.parse(["--sam", "vit_l"]), 1) == "custom"
    a = bench.parse(["--config", "C3"])
    assert (a.batch, a.sam, a.seg_tokens, a.with_msqp) == (32, "vit_h", 14, True)
