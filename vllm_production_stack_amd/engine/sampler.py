"""Token sampling.  Greedy rows use the argmax kernel.  On the GPU all stochastic rows of a step
(temperature / top-k / top-p) are drawn by one launch of the batched sampling kernel
(``csrc/kernels/sampling.hip``: key-threshold bisection instead of a sort, no host round trip
per row); rows that ask for penalties or logprobs, and CPU tensors, take the per-row PyTorch
definitions below (on the GPU penalties are applied to the logits rows in place first, so those
rows join the batched kernels too)."""

from __future__ import annotations

import os

import torch

from .. import ops
from .request import Request


class Sampler:
    def __init__(self, device: torch.device, seed: int = 0):
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self._req_gens: dict[str, torch.Generator] = {}

    def forget(self, request_id: str) -> None:
        self._req_gens.pop(request_id, None)

    def _gen_for(self, req: Request) -> torch.Generator:
        if req.params.seed is None:
            return self.gen
        g = self._req_gens.get(req.request_id)
        if g is None:
            g = torch.Generator(device=self.device)
            g.manual_seed(req.params.seed)
            self._req_gens[req.request_id] = g
        return g

    _penalized = False  # this step's penalties are already in the logits (GPU path)

    def _needs_penalty(self, p) -> bool:
        return (not self._penalized) and (p.presence_penalty != 0 or p.frequency_penalty != 0
                                          or p.repetition_penalty != 1.0)

    def sample(self, logits: torch.Tensor, reqs: list[Request],
               greedy_ids: torch.Tensor | None = None) -> tuple[list[int], list[dict | None]]:
        """logits [n, vocab] (row i belongs to reqs[i]) -> (token ids, optional logprob info)."""
        n = len(reqs)
        self._penalized = False
        all_plain_greedy = all(r.params.greedy and not self._needs_penalty(r.params)
                               and r.params.logprobs is None for r in reqs)
        if all_plain_greedy:
            ids = greedy_ids if greedy_ids is not None else ops.argmax(logits[:n])
            return ids[:n].tolist(), [None] * n
        out: list[int] = []
        infos: list[dict | None] = []
        lp_rows: list[int] = []
        # On the GPU, penalties are written into the logits rows themselves (the buffer is
        # rewritten by the next step anyway); after that every row is an ordinary row for the
        # batched kernels and nothing needs the per-row path.
        self._penalized = False
        pen_rows = [i for i, r in enumerate(reqs) if self._needs_penalty(r.params)]
        if pen_rows and ops.sample_supported(logits):
            idx = torch.tensor(pen_rows, device=logits.device)
            work = logits.index_select(0, idx).float()
            for j, i in enumerate(pen_rows):
                work[j] = self._penalize(work[j], reqs[i])
            logits.index_copy_(0, idx, work.to(logits.dtype))
            greedy_ids = None  # computed before the penalties
            self._penalized = True
        greedy_dev = (greedy_ids if greedy_ids is not None else ops.argmax(logits[:n]))[:n]
        batched = self._sample_batched(logits, reqs)  # launched before the greedy ids are read back
        greedy_all = greedy_dev.tolist()
        for i, r in enumerate(reqs):
            p = r.params
            if p.greedy and not self._needs_penalty(p) and p.logprobs is None:
                out.append(greedy_all[i])
                infos.append(None)
                continue
            if i in batched and p.logprobs is None:
                out.append(batched[i])
                infos.append(None)
                continue
            if not self._needs_penalty(p) and (p.greedy or i in batched):
                # token already known; its logprobs come from the batched pass below
                out.append(greedy_all[i] if p.greedy else batched[i])
                infos.append(None)
                lp_rows.append(i)
                continue
            row = logits[i].float()
            if self._needs_penalty(p):
                row = self._penalize(row, r)
            if p.greedy:
                tok = int(torch.argmax(row).item())
                probs_src = row
            else:
                row = row / max(p.temperature, 1e-5)
                if p.top_k and p.top_k > 0:
                    kth = torch.topk(row, min(p.top_k, row.numel())).values[-1]
                    row = row.masked_fill(row < kth, float("-inf"))
                if p.top_p < 1.0:
                    srt, idx = torch.sort(row, descending=True)
                    cum = torch.softmax(srt, -1).cumsum(-1)
                    drop = cum - torch.softmax(srt, -1) > p.top_p
                    srt = srt.masked_fill(drop, float("-inf"))
                    row = torch.full_like(row, float("-inf")).scatter_(0, idx, srt)
                probs = torch.softmax(row, -1)
                tok = int(torch.multinomial(probs, 1, generator=self._gen_for(r)).item())
                probs_src = row
            out.append(tok)
            if p.logprobs is not None:
                lp = torch.log_softmax(logits[i].float(), -1)  # the model's distribution, as in the batched pass
                k = max(1, min(int(p.logprobs), 20))
                tv, ti = torch.topk(lp, k)
                infos.append({"token_logprob": float(lp[tok]), "top_ids": ti.tolist(),
                              "top_logprobs": tv.tolist()})
            else:
                infos.append(None)
        if lp_rows:
            for i, info in zip(lp_rows, self._logprobs_batched(logits, lp_rows, [out[i] for i in lp_rows],
                                                               [reqs[i].params.logprobs for i in lp_rows])):
                infos[i] = info
        return out, infos

    @staticmethod
    def _logprobs_batched(logits: torch.Tensor, rows: list[int], toks: list[int], ks: list[int]) -> list[dict]:
        """Log-probabilities of the model's distribution (log-softmax of the unmodified logits)
        for the chosen token and the top-k alternatives of every listed row: one pass, one
        read-back."""
        dev = logits.device
        idx = torch.tensor(rows, device=dev)
        lp = torch.log_softmax(logits.index_select(0, idx).float(), -1)
        k = max(1, min(20, max(int(x) for x in ks)))
        tv, ti = torch.topk(lp, k, dim=-1)
        chosen = lp.gather(1, torch.tensor(toks, device=dev).view(-1, 1)).view(-1)
        tv, ti, chosen = tv.tolist(), ti.tolist(), chosen.tolist()
        out = []
        for j, want in enumerate(ks):
            n = max(1, min(20, int(want)))
            out.append({"token_logprob": chosen[j], "top_ids": ti[j][:n], "top_logprobs": tv[j][:n]})
        return out

    def _sample_batched(self, logits: torch.Tensor, reqs: list[Request]) -> dict[int, int]:
        """Rows that only need temperature / top-k / top-p, drawn in one kernel launch.
        -> {row index: token}.  The uniform numbers come from the request's own generator when
        it has a seed (reproducible whatever else is in the batch), else from the engine's."""
        if not ops.sample_supported(logits) or os.environ.get("VPSA_BATCHED_SAMPLING", "1") == "0":
            return {}
        idx = [i for i, r in enumerate(reqs)
               if not r.params.greedy and not self._needs_penalty(r.params)]
        if not idx:
            return {}
        params = torch.tensor([[reqs[i].params.temperature, reqs[i].params.top_p,
                                float(reqs[i].params.top_k or 0)] for i in idx], dtype=torch.float32)
        uniform = torch.empty(len(idx), dtype=torch.float32, device=logits.device)
        unseeded = [j for j, i in enumerate(idx) if reqs[i].params.seed is None]
        if unseeded:
            u = torch.rand(len(unseeded), generator=self.gen, device=logits.device)
            if len(unseeded) == len(idx):
                uniform = u
            else:
                uniform[torch.tensor(unseeded, device=logits.device)] = u
        for j, i in enumerate(idx):
            if reqs[i].params.seed is not None:
                uniform[j:j + 1] = torch.rand(1, generator=self._gen_for(reqs[i]), device=logits.device)
        rows = torch.tensor(idx, dtype=torch.int32).to(logits.device, non_blocking=True)
        toks = ops.sample(logits, params.to(logits.device, non_blocking=True), uniform, rows)
        return dict(zip(idx, toks.tolist()))

    @staticmethod
    def _penalize(row: torch.Tensor, r: Request) -> torch.Tensor:
        p = r.params
        row = row.clone()
        if r.output_ids:
            out_ids = torch.tensor(r.output_ids, device=row.device)
            counts = torch.bincount(out_ids, minlength=row.numel()).float()
            row -= p.frequency_penalty * counts
            row -= p.presence_penalty * (counts > 0).float()
        if p.repetition_penalty != 1.0:
            seen = torch.tensor(sorted(set(r.all_ids)), device=row.device)
            v = row[seen]
            row[seen] = torch.where(v > 0, v / p.repetition_penalty, v * p.repetition_penalty)
        return row
