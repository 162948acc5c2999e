"""bench.py's contract, rehearsed on the CPU: launched the way the multi-GPU run launches it
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W``),
rank 0 prints exactly one JSON line with the agreed fields, the value is the aggregate over replicas
and every replica served part of the load.  Tiny model, gloo."""

import json
import os

import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--device", "cpu", "--model", "tiny-llama", "--users-per-gpu", "3", "--warmup-users-per-gpu", "1",
         "--sys-len", "40", "--history-len", "200", "--answer-len", "8", "--max-model-len", "2048",
         "--max-num-batched-tokens", "512", "--reference-users-per-gpu", "2",
         "--reference-rounds", "2", "--reference-offload-gb", "0.05", "--open-loop-seconds", "4",
         "--open-loop-qps-per-gpu", "2.0"]
FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config")


def _one_json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_under_torchrun():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29733", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1"] + SMALL
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = _one_json_line(p.stdout)
    for f in FIELDS:
        assert f in res, f
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["higher_is_better"] is True and res["scaling"] == "weak" and res["unit"] == "tokens/s"
    assert res["config"]["global_batch"] == 6 and res["requests"] == 12 and res["failed_requests"] == 0
    assert res["value"] > 0 and abs(res["ms_per_step"] - res["elapsed_s"] / 2 * 1e3) < 1.0
    assert len(res["per_replica"]) == 2 and all(r["rows"] > 0 for r in res["per_replica"])
    # the second phase (the reference's own configuration) rides along and used the host tier
    ref = res["reference_config"]
    assert "error" not in ref, ref
    assert ref["failed_requests"] == 0 and ref["requests"] == 2 * 2 * 2 and ref["output_tok_per_s"] > 0
    assert ref["kv_tiers"]["host_stores_pages"] > 0
    # ... and ends with one point of the reference's open-loop procedure
    ol = ref["open_loop"]
    assert "error" not in ol, ol
    assert ol["requests"] > 0 and ol["mean_ttft_s"] > 0 and ol["offered_qps"] == pytest.approx(2 * 2.0)
    assert sum(r.get("host_loads_pages", 0) for r in ref["per_replica"]) > 0
    assert "Failed to scrape" not in p.stderr  # engines outlive the router at teardown


def test_single_process_defaults_to_one_gpu():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--base-port", "19860"] + SMALL, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = _one_json_line(p.stdout)
    assert res["n_gpus"] == 1 and res["config"]["global_batch"] == 3 and res["failed_requests"] == 0


def test_a_replica_that_cannot_start_ends_the_run_on_every_rank():
    """One rank fails while bringing its engine up (injected): the others must not wait for it at
    a barrier.  The run ends quickly on all ranks, with the failing rank named."""
    import time

    env = dict(os.environ, OMP_NUM_THREADS="1", VPSA_BENCH_FAIL_START_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--base-port", "19900"] + SMALL
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert time.time() - t0 < 120
    assert "engine start failed on rank(s) [1]" in p.stderr, p.stderr[-2000:]
    assert not p.stdout.strip()  # no half-measured JSON line


def _layout_line(args, timeout=600):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--device", "cpu", "--model", "tiny-llama",
                        "--steps", "2", "--warmup", "1", "--users-per-gpu", "2", "--warmup-users-per-gpu", "1",
                        "--sys-len", "64", "--history-len", "300", "--answer-len", "8", "--max-model-len", "2048",
                        "--num-gpu-blocks", "64"] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_pd_layout_runs_end_to_end_on_the_cpu():
    """``bench.py --layout pd:1x2``: one prefill and two decode replicas as processes of their own,
    the router in disaggregated_prefill mode, the multi-round workload; every request goes prefill
    pool -> decode pool, the pages travel (TCP between CPU replicas) chunk by chunk, and the decode
    pool finds every pushed prefix resident."""
    r = _layout_line(["--layout", "pd:1x2", "--base-port", "18300"])
    assert r["failed_requests"] == 0 and r["requests"] == 8 and r["value"] > 0
    assert r["config"]["parallelism"].startswith("pd:1x2")
    assert r["requests_routed_prefill_then_decode"] >= 8
    pre = r["prefill_pool"][0]
    assert pre["pushed_pages"] > 0 and pre["failed_pushes"] == 0 and pre["push_jobs"] >= 2 * 8
    assert sum(d["prefix_hit_tokens"] for d in r["decode_pool"]) > 0
    # follow-up turns return to the decode replica that holds the conversation: nothing left to push
    assert {"tcp"} <= set(r["kv_pushed_gb_by_transport"]) <= {"tcp", "resident"}


def test_multimodel_layout_runs_end_to_end_on_the_cpu():
    """``bench.py --layout multimodel``: six dense replicas + a two-rank tensor-parallel MoE replica
    (torch.distributed.run, gloo), the cache server and the router with kvaware routing hosting the
    KV controller; both models answer through one router and every conversation stays on the
    replica that holds its KV."""
    r = _layout_line(["--layout", "multimodel", "--base-port", "18400"], timeout=900)
    assert r["tiny-llama"]["failed_requests"] == 0 and r["tiny-mixtral"]["failed_requests"] == 0
    assert r["tiny-llama"]["requests"] == 24 and r["tiny-mixtral"]["requests"] == 4
    assert r["conversations_kept_on_one_replica"] == r["conversations"] == 14
    assert any(e["model"] == "tiny-mixtral" and e["rows"] > 0 for e in r["per_replica"])
    assert r["cache_server"]["pages_known"] > 0


def test_a_failing_load_generator_in_the_second_phase_does_not_strand_the_other_ranks():
    """Rank 0 loses its load generators in the reference-configuration phase (injected).  The
    other ranks wait at that phase's barriers: rank 0 must still pass them, so the run ends at
    once on every rank, with the headline result intact and the failure recorded in its place."""
    import time

    env = dict(os.environ, OMP_NUM_THREADS="1", VPSA_BENCH_FAIL_CLIENTS="20000000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29743", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--base-port", "19940"] + SMALL
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    assert time.time() - t0 < 300
    res = _one_json_line(p.stdout)
    assert res["value"] > 0 and res["failed_requests"] == 0
    assert "injected load generator failure" in res["reference_config"]["error"]
