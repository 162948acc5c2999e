import sys, pytest
sys.path.insert(0, "/root/repo")
from vllm_production_stack_amd import ops
v = int(sys.argv[1])
ops.set_prefill_variant(v)
sys.exit(pytest.main(["-x", "-q", "tests/test_kernels_gpu.py", "-k", "prefill", "-m", "gpu", "--timeout", "120", "--timeout-method", "thread"]))
