"""Randomised scheduler test on the CPU: chunked prefill x prefix cache x preemption x abort x
host-tier reload, interleaved at random.

Every seed draws an engine configuration (page pool of a few pages, token budget, chunked prefill
and prefix caching on or off, host KV tier on or off) and a script of events over the step loop:
requests arrive at random steps (prompts share random-length prefixes with earlier prompts and with
earlier answers, so prefix hits, partial pages and reloads from the host tier all occur), some are
aborted at a random later step, and the pool is small enough that running sequences are preempted
and recomputed.  Invariants:

* every request that was not aborted generates exactly the tokens of its SOLO greedy generation on
  a fresh engine with a large pool (no sharing, no preemption, no tiers);
* when the engine has drained, no page holds a reference and every page is free or cached;
* the prefix-cache counters never report more hits than queries.
"""

import random

import pytest

from vllm_production_stack_amd.engine.config import EngineConfig, KVOffloadConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.request import SamplingParams

N_SEEDS = 200


class _LateEvent:
    """Completion of a host-tier copy that reports done only after a few polls, as a copy queued
    behind others on the GPU's copy stream does: the pages it reads stay pinned over several
    steps, across the finish, preemption or abort of the request that owned them."""

    def __init__(self, polls: int):
        self.left = polls

    def query(self) -> bool:
        self.left -= 1
        return self.left <= 0

    def wait(self) -> None:
        self.left = 0


def _delay_host_copies(eng, rng: random.Random) -> None:
    host = eng.connector.host
    store = host.store

    def late_store(items, compute_event=None):
        fresh = [h for h, _ in items if h not in host.index]
        store(items, compute_event)  # the CPU tier copies at once
        return _LateEvent(rng.randint(1, 6)) if fresh else None

    host.store = late_store
_solo_engine = None
_solo_cache: dict = {}


def _solo(prompt: list[int], max_tokens: int) -> list[int]:
    """Greedy generation of one prompt alone (large pool, prefix caching off), memoised."""
    global _solo_engine
    key = (tuple(prompt), max_tokens)
    if key not in _solo_cache:
        if _solo_engine is None:
            _solo_engine = LLMEngine(EngineConfig(
                model="tiny-llama", device="cpu", dtype="float32", max_model_len=1024, max_num_seqs=4,
                max_num_batched_tokens=1024, num_gpu_blocks_override=64, enable_prefix_caching=False))
        sp = SamplingParams(max_tokens=max_tokens, temperature=0, ignore_eos=True)
        _solo_cache[key] = _solo_engine.generate([prompt], sp)[0]
    return _solo_cache[key]


def _script(rng: random.Random):
    """(arrival step, prompt, max_tokens, abort step or None) per request."""
    vocab = lambda n: [rng.randrange(16, 500) for _ in range(n)]  # noqa: E731
    reqs = []
    pool_of_prefixes = [vocab(rng.choice((64, 128, 200)))]
    for i in range(rng.randint(4, 10)):
        kind = rng.random()
        if kind < 0.5 and pool_of_prefixes:  # follow-up turn / shared system prompt
            base = rng.choice(pool_of_prefixes)
            prompt = base[: rng.randint(1, len(base))] + vocab(rng.randint(1, 70))
        else:
            prompt = vocab(rng.choice((1, 5, 63, 64, 65, 127, 128, 130, 250)))
        prompt = prompt[:400]
        pool_of_prefixes.append(prompt)
        arrive = rng.randint(0, 40)
        abort = arrive + rng.randint(0, 10) if rng.random() < 0.15 else None
        # long answers cross page boundaries: the pool runs dry in the middle of decoding and the
        # scheduler has to preempt; short ones keep most requests cheap
        max_tokens = rng.randint(40, 90) if rng.random() < 0.3 else rng.randint(1, 9)
        reqs.append((arrive, prompt, max_tokens, abort))
    return reqs


@pytest.mark.parametrize("chunk", range(0, N_SEEDS, 20))
def test_random_interleavings_match_solo_generation(chunk):
    for seed in range(chunk, chunk + 20):
        rng = random.Random(seed)
        pages = rng.choice((8, 9, 10, 12, 16))
        budget = rng.choice((32, 64, 100, 256, 512))
        chunked = rng.random() < 0.8
        prefix = rng.random() < 0.8
        host = rng.random() < 0.5
        cfg = EngineConfig(
            model="tiny-llama", device="cpu", dtype="float32", max_model_len=1024,
            max_num_seqs=rng.choice((2, 4, 8)), max_num_batched_tokens=budget,
            num_gpu_blocks_override=pages, enable_chunked_prefill=chunked, enable_prefix_caching=prefix,
            kv_offload=KVOffloadConfig(local_cpu=True, max_local_cpu_gb=0.004) if host else KVOffloadConfig(), port=0,
            kv_layer_groups=1 + seed % 2)  # whole pages / one pool per layer
        eng = LLMEngine(cfg)
        late = host and rng.random() < 0.7
        if late:
            _delay_host_copies(eng, rng)
        script = _script(rng)
        # a prompt that can never fit the pool (with its answer) would be refused: keep it feasible
        script = [(a, p[: (pages - 3) * 64 - 16], m, ab) for a, p, m, ab in script]
        live: dict[int, object] = {}
        aborted: set[int] = set()
        step = 0
        tag = f"seed {seed} pages {pages} budget {budget} chunked {chunked} prefix {prefix} host {host} late {late}"
        while step < 1500:
            for i, (arrive, prompt, max_tokens, _ab) in enumerate(script):
                if arrive == step:
                    sp = SamplingParams(max_tokens=max_tokens, temperature=0, ignore_eos=True)
                    live[i] = eng.add_request(prompt, sp, request_id=f"s{seed}-r{i}")
            for i, (_a, _p, _m, ab) in enumerate(script):
                if ab == step and i in live and i not in aborted:
                    eng.abort(f"s{seed}-r{i}")
                    aborted.add(i)
            if eng.has_work():
                eng.step()
            elif step > max(a for a, *_ in script):
                break
            step += 1
        assert not eng.has_work(), f"{tag}: did not drain in 1500 steps"
        for i, (_a, prompt, max_tokens, _ab) in enumerate(script):
            if i in aborted:
                continue
            assert live[i].output_ids == _solo(prompt, max_tokens), f"{tag}: request {i} (prompt {len(prompt)} tokens)"
        if eng.connector is not None:
            for _ in range(50):  # copies to the host tier still hold references until polled
                eng.connector.poll()
                if not any(eng.bm.ref):
                    break
        assert not any(eng.bm.ref), f"{tag}: pages still referenced after the drain: {[r for r in eng.bm.ref if r]}"
        assert eng.bm.hits <= eng.bm.queries, tag
        assert eng.bm.num_free() == pages or prefix or host, tag
        assert len(set(eng.bm.free_ids)) == len(eng.bm.free_ids), f"{tag}: a page is on the free list twice"
        assert not set(eng.bm.free_ids) & set(eng.bm.evictable), f"{tag}: a page is both free and cached"
        eng.shutdown()
