"""Device-side description of one engine step (a mixed batch).

Token rows are ordered [decode rows | prefill rows]: the first ``num_decode`` rows are
one-token continuations (served by the decode kernel, HIP-graph capturable), the rest
are prompt chunks of ``num_prefill_seqs`` sequences (served by the prefill kernel).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class BatchMeta:
    positions: torch.Tensor  # int32 [T]
    slots: torch.Tensor  # int64 [T]  page*64 + offset, -1 = do not cache
    num_decode: int = 0
    # decode part
    dec_block_tables: torch.Tensor | None = None  # int32 [num_decode, max_pages]
    dec_ctx_lens: torch.Tensor | None = None  # int32 [num_decode]  (includes the new token)
    dec_num_splits: int = 1
    dec_ws_o: torch.Tensor | None = None
    dec_ws_ml: torch.Tensor | None = None
    # prefill part
    num_prefill_seqs: int = 0
    pre_block_tables: torch.Tensor | None = None  # int32 [num_prefill_seqs, max_pages]
    pre_cu_q: torch.Tensor | None = None  # int32 [num_prefill_seqs + 1], rows relative to prefill part
    pre_ctx_lens: torch.Tensor | None = None  # int32 [num_prefill_seqs]
    pre_tile_seq: torch.Tensor | None = None
    pre_tile_q0: torch.Tensor | None = None
    pre_tile_rows: int | None = None  # tile height the work list was built for
    # key-range splits of a short work list (ops.plan_prefill_splits): int32 [entries, 3], the fp32
    # workspace of the partial records, int32 [groups, 4]; None = every entry walks all its keys
    pre_split: torch.Tensor | None = None
    pre_split_ws: torch.Tensor | None = None
    pre_split_groups: torch.Tensor | None = None
    # the batch is padded to a captured graph's fixed size: rows past the real ones belong to no
    # sequence (slot -1, no tile covers them), so the attention output is zeroed first and what
    # flows on from those rows stays finite
    pre_padded: bool = False
    # rows whose logits are needed (last row of every sequence that samples this step)
    logits_rows: torch.Tensor | None = None  # int64 [num_out]
    # per-row LoRA slot (-1 = base model), only when adapters are active
    lora_slots: torch.Tensor | None = None
