"""The whole multi-model stack as processes on the CPU (tiny models): cache server, router with
KV-aware routing and its controller, two dense replicas and one MoE replica
(benchmarks/config5_multi_model.py --small).  The reference only stubs its kvaware e2e check
(/root/reference/tests/e2e/test-routing.py:471-482)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kvaware_routing_and_shared_cache_across_real_processes(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "config5_multi_model.py"), "--small",
                        "--base-port", "19740"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["pass"] and res["models"] == ["tiny-llama", "tiny-mixtral"]
    assert res["llama_repeats_on_holder"] == 3 and res["mixtral_repeats_on_holder"] == 3
    assert res["remote_hit_tokens_on_other_replica"] > 0 and res["moved_to"] != res["llama_first_replica"]
    assert res["llama_cached_tokens_on_repeat"] > 0


def test_disaggregated_prefill_with_pools_of_replicas_across_real_processes():
    """Two prefill and three decode replicas behind the router in disaggregated_prefill mode
    (benchmarks/pd_pools.py --small): every request visits one replica of each pool, both pools are
    spread over, the decode replicas find the pushed prompt resident (pages travel by TCP here)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "pd_pools.py"), "--small", "--prefill", "2",
                        "--decode", "3", "--requests", "6", "--concurrency", "3", "--base-port", "19780"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["pass"] and res["requests_routed_prefill_then_decode"] == 6
    assert len(res["prefill_pool_use"]) == 2 and len(res["decode_pool_use"]) == 3
    assert all(v["by_mode"] == {"tcp": v["pushes"]} for v in res["push_by_prefill_replica"].values())
    assert res["first_answer_equals_self_computed"]
