"""Token sampling: greedy rows use the argmax kernel; stochastic rows apply temperature,
top-k, top-p and (optionally) repetition/presence/frequency penalties in fp32."""

from __future__ import annotations

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

    @staticmethod
    def _needs_penalty(p) -> bool:
        return p.presence_penalty != 0 or p.frequency_penalty != 0 or p.repetition_penalty != 1.0

    def sample(self, logits: torch.Tensor, reqs: list[Request],
               greedy_ids: torch.Tensor | None = None) -> tuple[list[int], list[dict | None]]:
        """logits [n, vocab] (row i belongs to reqs[i]) -> (token ids, optional logprob info)."""
        n = len(reqs)
        all_plain_greedy = all(r.params.greedy and not self._needs_penalty(r.params)
                               and r.params.logprobs is None for r in reqs)
        if all_plain_greedy:
            ids = greedy_ids if greedy_ids is not None else ops.argmax(logits[:n])
            return ids[:n].tolist(), [None] * n
        out: list[int] = []
        infos: list[dict | None] = []
        greedy_all = (greedy_ids if greedy_ids is not None else ops.argmax(logits[:n]))[:n].tolist()
        for i, r in enumerate(reqs):
            p = r.params
            if p.greedy and not self._needs_penalty(p) and p.logprobs is None:
                out.append(greedy_all[i])
                infos.append(None)
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
                lp = torch.log_softmax(probs_src, -1)
                k = max(1, min(int(p.logprobs), 20))
                tv, ti = torch.topk(lp, k)
                infos.append({"token_logprob": float(lp[tok]), "top_ids": ti.tolist(),
                              "top_logprobs": tv.tolist()})
            else:
                infos.append(None)
        return out, infos

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
