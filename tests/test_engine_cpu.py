"""Engine core on CPU (PyTorch op definitions): the paged / chunked / prefix-cached /
preempting execution must generate exactly what a dense full-recompute forward generates."""

import copy

import pytest
import torch

from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.request import SamplingParams
from vllm_production_stack_amd.ops import reference as ref


def dense_next_token(model, ids):
    """Full-recompute greedy next token with the engine's weights (no cache, no paging)."""
    cfg = model.cfg
    x = model.embed[torch.tensor(ids)].float()
    t = len(ids)
    half = cfg.head_dim // 2
    cs = model.cos_sin[:t]
    cos, sin = cs[:, None, :half], cs[:, None, half:]
    mask = torch.triu(torch.ones(t, t, dtype=torch.bool), 1)
    for lw in model.layers:
        h = ref.rms_norm(x, lw.ln1.float(), cfg.norm_eps)
        qkv = (h @ lw.wqkv.float().t()).view(t, -1, cfg.head_dim)
        q, k, v = qkv[:, :cfg.num_heads], qkv[:, cfg.num_heads:cfg.num_heads + cfg.num_kv_heads], \
            qkv[:, cfg.num_heads + cfg.num_kv_heads:]

        def rot(z):
            a, b = z[..., :half], z[..., half:]
            return torch.cat([a * cos - b * sin, b * cos + a * sin], -1)

        q, k = rot(q), rot(k)
        g = cfg.num_heads // cfg.num_kv_heads
        k = k.repeat_interleave(g, 1)
        v = v.repeat_interleave(g, 1)
        sc = torch.einsum("qhd,khd->hqk", q, k) / cfg.head_dim ** 0.5
        sc = sc.masked_fill(mask, float("-inf"))
        o = torch.einsum("hqk,khd->qhd", sc.softmax(-1), v).reshape(t, -1)
        x = x + o @ lw.wo.float().t()
        h = ref.rms_norm(x, lw.ln2.float(), cfg.norm_eps)
        if cfg.num_experts:
            p = (h @ lw.router.float().t()).softmax(-1)
            w, idx = p.topk(cfg.num_experts_per_tok, -1)
            w = w / w.sum(-1, keepdim=True)
            y = torch.zeros_like(x)
            for e in range(cfg.num_experts):
                gu = h @ lw.w13[e].float().t()
                a = ref.silu_and_mul(gu) @ lw.w2[e].float().t()
                y += a * (w * (idx == e)).sum(-1, keepdim=True)
            x = x + y
        else:
            gu = h @ lw.w_gate_up.float().t()
            x = x + ref.silu_and_mul(gu) @ lw.w_down.float().t()
    h = ref.rms_norm(x[-1:], model.norm.float(), cfg.norm_eps)
    return int((h @ model.lm_head.float().t()).argmax(-1))


def make_engine(model="tiny-llama", **kw):
    cfg = EngineConfig(model=model, device="cpu", dtype="float32", max_model_len=1024,
                       max_num_seqs=8, max_num_batched_tokens=kw.pop("budget", 256),
                       num_gpu_blocks_override=kw.pop("pages", 64), **kw)
    return LLMEngine(cfg)


@pytest.mark.parametrize("model", ["tiny-llama", "tiny-llama-gqa", "tiny-mixtral"])
def test_greedy_matches_dense(model):
    eng = make_engine(model)
    g = torch.Generator().manual_seed(0)
    prompt = torch.randint(16, 500, (150,), generator=g).tolist()
    out = eng.generate([prompt], SamplingParams(max_tokens=6, temperature=0, ignore_eos=True))[0]
    ids = list(prompt)
    for tok in out:
        assert dense_next_token(eng.runner.model, ids) == tok
        ids.append(tok)


def test_chunked_prefill_equals_whole_prompt():
    g = torch.Generator().manual_seed(1)
    prompt = torch.randint(16, 500, (300,), generator=g).tolist()
    sp = SamplingParams(max_tokens=5, temperature=0, ignore_eos=True)
    a = make_engine(budget=64).generate([prompt], copy.deepcopy(sp))[0]
    b = make_engine(budget=1024).generate([prompt], copy.deepcopy(sp))[0]
    c = make_engine(budget=64, enable_chunked_prefill=False).generate([prompt], copy.deepcopy(sp))[0]
    assert a == b == c


def test_prefix_cache_hit_same_tokens_and_counted():
    eng = make_engine()
    g = torch.Generator().manual_seed(2)
    base = torch.randint(16, 500, (200,), generator=g).tolist()
    sp = SamplingParams(max_tokens=4, temperature=0, ignore_eos=True)
    first = eng.generate([base], copy.deepcopy(sp))[0]
    assert eng.bm.hits == 0
    # second turn: previous prompt + answer + new question
    turn2 = base + first + [20, 21, 22]
    r = eng.add_request(turn2, copy.deepcopy(sp))
    while eng.has_work():
        eng.step()
    assert r.num_cached_tokens == 192  # three full 64-token pages of shared history
    assert eng.bm.hits == 192
    fresh = make_engine().generate([turn2], copy.deepcopy(sp))[0]
    assert r.output_ids == fresh


def test_batched_requests_are_independent():
    eng = make_engine()
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(16, 500, (n,), generator=g).tolist() for n in (5, 70, 131, 64)]
    sp = SamplingParams(max_tokens=8, temperature=0, ignore_eos=True)
    together = eng.generate(prompts, copy.deepcopy(sp))
    for p, t in zip(prompts, together):
        alone = make_engine().generate([p], copy.deepcopy(sp))[0]
        assert alone == t


def test_preemption_recomputes_correctly():
    g = torch.Generator().manual_seed(4)
    prompts = [torch.randint(16, 500, (100,), generator=g).tolist() for _ in range(4)]
    sp = SamplingParams(max_tokens=40, temperature=0, ignore_eos=True)
    small = make_engine(pages=7, enable_prefix_caching=False)
    got = small.generate(prompts, copy.deepcopy(sp))
    assert small.scheduler.num_preemptions > 0
    big = make_engine(pages=64).generate(prompts, copy.deepcopy(sp))
    assert got == big


def test_opt_model_runs_on_cpu():
    eng = make_engine("tiny-opt")
    out = eng.generate([[5, 6, 7, 8, 9]], SamplingParams(max_tokens=5, temperature=0, ignore_eos=True))
    assert len(out[0]) == 5


def test_stop_conditions_and_sampling_seed():
    eng = make_engine()
    sp = SamplingParams(max_tokens=10, temperature=0.8, top_p=0.9, top_k=50, seed=7, ignore_eos=True)
    a = eng.generate([[17, 18, 19]], copy.deepcopy(sp))[0]
    b = eng.generate([[17, 18, 19]], copy.deepcopy(sp))[0]
    assert a == b and len(a) == 10
    stop_tok = a[2]
    sp2 = copy.deepcopy(sp)
    sp2.stop_token_ids = [stop_tok]
    c = eng.generate([[17, 18, 19]], sp2)[0]
    assert c == a[: a.index(stop_tok) + 1]


def test_context_limit_is_enforced():
    eng = make_engine()
    with pytest.raises(ValueError):
        eng.add_request(list(range(16, 16 + 1024)), SamplingParams())


def test_kernel_launches_move_the_calling_thread_to_the_bound_gpu(monkeypatch):
    """The HIP current device is per thread and starts at 0: a launch from a fresh thread of a
    process bound to GPU 3 must select GPU 3 first (streams and kernel lookups follow it)."""
    import threading

    import torch

    from vllm_production_stack_amd import ops

    state = threading.local()
    calls = []
    monkeypatch.setattr(torch.cuda, "current_device", lambda: getattr(state, "dev", 0))

    def set_device(d):
        state.dev = int(d)
        calls.append((threading.current_thread().name, int(d)))

    monkeypatch.setattr(torch.cuda, "set_device", set_device)
    ops.bind_device(3)
    try:
        ops.ensure_device()
        ops.ensure_device()  # already there: no second switch
        t = threading.Thread(target=ops.ensure_device, name="fresh")
        t.start()
        t.join()
    finally:
        ops.bind_device(None)
    assert calls == [(threading.current_thread().name, 3), ("fresh", 3)]


def test_latency_aware_prefill_admission_splits_a_burst_of_follow_up_turns():
    """With a prefill cost model, a burst of short prompts is spread over several steps (each
    under the step target) and a single long prompt is still taken whole; without it the token
    budget alone decides."""
    from vllm_production_stack_amd.engine.block_manager import BlockManager
    from vllm_production_stack_amd.engine.request import Request, SamplingParams
    from vllm_production_stack_amd.engine.scheduler import Scheduler

    def burst(sched, n, length):
        for i in range(n):
            sched.add(Request(f"r{i}", [1000 * (i + 1) + j for j in range(length)], SamplingParams(max_tokens=4)))

    plain = Scheduler(BlockManager(512, True), 64, 8192, 4096)
    burst(plain, 12, 100)
    assert len(plain.schedule().prefills) == 12

    timed = Scheduler(BlockManager(512, True), 64, 8192, 4096)
    timed.prefill_cost_ms = lambda n, start: n * 0.1  # 10 ms per 100-token prompt
    timed.prefill_step_ms = 35.0
    burst(timed, 12, 100)
    sizes = []
    for _ in range(6):
        out = timed.schedule()
        sizes.append(len(out.prefills))
        for s in out.prefills + out.decodes:
            timed.commit(s)
            s.req.output_ids.append(1)
    assert sizes[:4] == [3, 3, 3, 3] and sum(sizes) == 12

    alone = Scheduler(BlockManager(512, True), 64, 8192, 4096)
    alone.prefill_cost_ms = lambda n, start: n * 0.1
    alone.prefill_step_ms = 35.0
    burst(alone, 1, 3000)  # 300 ms estimated: still scheduled (the first prefill always is)
    out = alone.schedule()
    assert [s.n for s in out.prefills] == [3000]


def test_kv_transfer_reservations_expire_when_the_producer_never_commits():
    """A prefill replica that dies between /kv/reserve and /kv/commit must not keep the decode
    replica's pages: the reservation is dropped after its TTL, and a late commit is refused."""
    import time

    from vllm_production_stack_amd.parallel.kv_transfer import KVReceiver

    eng = make_engine(pages=16)
    recv = KVReceiver(eng)
    recv.reserve_ttl = 0.05
    pool = eng.runner.pool
    page_bytes = pool.stride(0) * pool.element_size()
    free0 = eng.bm.num_free()
    res = recv.reserve([11, 12, 13], page_bytes)
    assert len(res["pages"]) == 3 and eng.bm.num_free() == free0 - 3
    time.sleep(0.1)
    again = recv.reserve([21], page_bytes)  # any later reserve sweeps
    assert eng.bm.num_free() == free0 - 1 and not any(h in recv.reserved for h in (11, 12, 13))
    with pytest.raises(ValueError):
        recv.commit([11, 12, 13], res["pages"])
    recv.commit([21], again["pages"])
    assert eng.bm.contains(21) and not recv.reserved and not recv._reserved_at


def test_ipc_safe_size_avoids_the_sizes_hipipc_cannot_map():
    from vllm_production_stack_amd.parallel.kv_transfer import ipc_safe_size

    G = 1 << 30
    for want in (G // 2, G, 4 * G, 4 * G + G // 2, 5 * G, 8 * G, 9 * G, 12 * G, 100 * G):
        assert ipc_safe_size(want) == want  # measured to map
    for want in (2 * G, 2 * G + G // 2, 3 * G, 6 * G, 7 * G, 10 * G, 14 * G, 4 * G - 1):
        got = ipc_safe_size(want)
        assert got <= want and want - got <= 2 * G + (2 << 20) and (got & 0xFFFFFFFF) < 0x80000000 and got % 4096 == 0
    assert ipc_safe_size(2 * G) == 2 * G - (2 << 20)


def test_prefill_split_planner_covers_every_key_tile_once():
    """The planner of key-range splits (short work lists in front of long contexts): pieces of a
    tile are contiguous, cover [0, key tiles) exactly, are at least 8 tiles long (the kernel needs
    3 in the last one), record numbers are consecutive, caps of fixed-size graph buffers hold, and
    nothing is cut when the list already fills the chip or the contexts are short."""
    import random

    from vllm_production_stack_amd import ops

    rng = random.Random(3)
    cut_some = False
    for _ in range(300):
        n_seqs = rng.choice((rng.randint(1, 8), rng.randint(9, 48)))
        q_lens = [rng.randint(1, 200) for _ in range(n_seqs)]
        shared = rng.choice((None, 21000))  # equal histories: the part-filled last round is what gets cut
        ctx = [q + (shared or rng.choice((0, 100, 3000, 21000, 60000))) for q in q_lens]
        n_q, n_kv = rng.choice(((32, 8), (8, 2), (64, 8), (16, 4), (28, 4), (12, 4)))
        seqs, q0s = [], []
        for s in range(n_seqs):
            for q0 in range((q_lens[s] - 1) // 64 * 64, -1, -64):
                seqs.append(s)
                q0s.append(q0)
        caps = rng.choice((None, (len(seqs) + 72, 72), (len(seqs) + 4, 4)))
        r = ops.plan_prefill_splits(seqs, q0s, q_lens, ctx, n_q, n_kv, 64,
                                    max_entries=caps and caps[0], max_records=caps and caps[1])
        if r is None:
            continue
        cut_some = True
        seqs2, q0s2, split, groups, n_rec = r
        assert len(split) == 3 * len(seqs2) == 3 * len(q0s2)
        if caps:
            assert len(seqs2) <= caps[0] and n_rec <= caps[1]
        by_tile: dict = {}
        for i, (s, q0) in enumerate(zip(seqs2, q0s2)):
            by_tile.setdefault((s, q0), []).append(split[3 * i:3 * i + 3])
        assert sorted(by_tile) == sorted(zip(seqs, q0s))
        recs = []
        for (s, q0), pieces in by_tile.items():
            kt = (ctx[s] - q_lens[s] + min(q0 + 63, q_lens[s] - 1)) // 64 + 1
            assert pieces[0][0] == 0 and pieces[-1][1] == kt
            for a, b in zip(pieces, pieces[1:]):
                assert a[1] == b[0]
            if len(pieces) == 1:
                assert pieces[0][2] == -1
            else:
                assert all(p[1] - p[0] >= 8 for p in pieces)
                recs += [p[2] for p in pieces]
                g = [groups[j:j + 4] for j in range(0, len(groups), 4) if groups[j:j + 2] == [s, q0]]
                assert g == [[s, q0, pieces[0][2], len(pieces)]]
        assert sorted(recs) == list(range(n_rec))
    assert cut_some
    # lists that fill the chip in whole rounds, and short contexts, stay as they are
    for n in (32, 64):
        assert ops.plan_prefill_splits(list(range(n)), [0] * n, [30] * n, [21000] * n, 32, 8, 64) is None
    # 40 x 8 workgroups would run as one full round and a quarter-filled second
    cut = ops.plan_prefill_splits(list(range(40)), [0] * 40, [30] * 40, [21000] * 40, 32, 8, 64)
    assert cut is not None and len(cut[0]) * 8 % 256 <= 8 * 8
    assert ops.plan_prefill_splits([0], [0], [8], [300], 32, 8, 64) is None
    assert ops.plan_prefill_splits([0], [0], [8], [21000], 12, 4, 64) is not None  # ratio 3: one padded head group
    assert ops.plan_prefill_splits([0], [0], [8], [21000], 8, 4, 64) is None  # ratio 2: no 64-row kernel


def test_a_pool_cut_into_layer_groups_generates_the_same_tokens():
    """``kv_layer_groups``: each group of layers keeps its K/V in a pool of its own; what attention
    reads and writes is the same, and replicas whose pages leave the pool as a whole keep one pool."""
    from vllm_production_stack_amd import ops
    g = torch.Generator().manual_seed(5)
    ps = [torch.randint(16, 500, (n,), generator=g).tolist() for n in (150, 70, 300)]
    ps.append(ps[0][:128] + ps[1][:40])  # shares two pages with the first prompt
    sp = SamplingParams(max_tokens=8, temperature=0, ignore_eos=True)
    want = make_engine("tiny-llama-gqa", budget=96).generate(ps, copy.deepcopy(sp))
    eng = make_engine("tiny-llama-gqa", budget=96, kv_layer_groups=3)  # the model has three layers
    pool = eng.runner.pool
    assert isinstance(pool, ops.LayerGroupedPool) and len(pool.groups) == 3
    assert pool.shape == (64, 3, 2, eng.runner.n_kv_local, 64 * eng.runner.kv_head_dim)
    assert eng.generate(ps, copy.deepcopy(sp)) == want
    page = pool.page(3)
    assert page.shape == pool.shape[1:] and page.abs().sum() > 0
    pool.write_page(5, page)
    assert torch.equal(pool.page(5), page)
    assert isinstance(make_engine("tiny-llama-gqa", kv_layer_groups=2).runner.pool, torch.Tensor)  # 3 layers
    # a replica that exchanges pages with other replicas keeps whole pages: the request is ignored there
    from vllm_production_stack_amd.engine.config import KVTransferConfig

    pd = make_engine("tiny-llama-gqa", kv_layer_groups=3, kv_transfer=KVTransferConfig(kv_role="kv_consumer"))
    assert isinstance(pd.runner.pool, torch.Tensor)
    pd.shutdown()
