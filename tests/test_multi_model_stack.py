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
