"""Disaggregated prefill across real processes on the GPU: one prefill and two decode replicas of a
tiny model behind the router (benchmarks/pd_pools.py).  By default the decode replicas export their
KV pools and pages are written where they will live (``ipc-direct``); with the export switched off
they go through hipIpc into a landing window with fewer slots than a prompt has pages, so every
push goes in rounds and the two concurrent producers' requests contend for the slots."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("export_pool,mode", [("1", "ipc-direct"), ("0", "ipc")])
def test_pd_pools_push_by_hipipc_between_processes(export_pool, mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VPSA_PD_LANDING_MIB="1", VPSA_PD_EXPORT_POOL=export_pool)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "pd_pools.py"), "--model", "tiny-llama-gqa",
                        "--prefill", "1", "--decode", "2", "--requests", "4", "--concurrency", "2",
                        "--prompt-tokens", "700", "--pages", "96", "--base-port", "19920", "--request-timeout", "60"],
                       capture_output=True, text=True, timeout=110, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["pass"] and res["requests_routed_prefill_then_decode"] == 4
    push = res["push_by_prefill_replica"]["prefill-0"]
    assert push["by_mode"] == {mode: 4} and push["pages"] == 40
    assert res["cached_tokens_min"] == 640 and res["first_answer_equals_self_computed"]
