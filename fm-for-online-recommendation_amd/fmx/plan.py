"""OwnerPlan: which rank owns which rows, and where they sit in the forward tree (the multi-GPU field-owner mode).

The forward pass adds a sample's rows in a fixed tree (include/fmx.h): position j = pass * SLOTS + lane group, the lane group
adds its passes in order, a butterfly adds the lane groups.  The tree is cut into NB blocks of SL = SLOTS / NB consecutive
lane groups (NB = the smallest power of two >= world); rank g owns consecutive blocks.  What sits at a position is a PIECE: a
consecutive row range of one index column (SURVEY.md section 8(e) "row-sharded table"; reference workloads: the 39 Criteo
columns of main_experiment.py:56-58, the Frappe columns of utils/data_manager.py:83-136).  Large columns are cut into several
pieces, and the pieces are dealt over the blocks by cost, so that

  * every rank owns rows for any number of columns and any world size up to SLOTS (a column of >= world rows can be cut),
  * HBM footprint (rows) and gather / update work (expected occurrences: a piece of a column sees its share of every batch)
    are each balanced over the ranks,
  * G ranks stay BIT-IDENTICAL to one table that places the same pieces at the same positions (table_whole()): the additions
    and their order are those of that table's forward pass.  With world == 1 the plan is the ordinary table (column f at
    position f), so the one-GPU results are the old ones.
"""
import math

import numpy as np
import torch

from .table import FlatTable, padded_k

MAX_PASSES = 4          # k_fm_forward_part is built for up to four fields per lane group


class OwnerPlan:
    def __init__(self, feature_sizes, k, world, global_batch=4096, split_factor=0.6):
        self.sizes = [int(s) for s in feature_sizes]
        self.k, self.kp, self.world = int(k), padded_k(k), int(world)
        self.slots = 64 // (self.kp // 4)
        if self.world < 1 or self.world > self.slots:
            raise ValueError(f"the forward tree has {self.slots} lane groups at k = {k}: 1 .. {self.slots} owners, not {world}")
        nb = 1
        while nb < self.world:
            nb *= 2
        self.nb, self.sl = nb, self.slots // nb
        # rank g owns the consecutive blocks [first[g], first[g] + count[g]): the nb - world extra blocks go to the first ranks
        extra = nb - self.world
        self.block_count = [2 if g < extra else 1 for g in range(self.world)]
        self.block_first = [sum(self.block_count[:g]) for g in range(self.world)]
        self.block_owner = [g for g in range(self.world) for _ in range(self.block_count[g])]
        self.pieces = self._cut(split_factor)                       # [(col, base, rows)]
        self.block_pieces = self._deal(global_batch)                # per block: piece ids in local position order
        per_block = max(len(b) for b in self.block_pieces)
        self.np = max(1, -(-per_block // self.sl))
        assert self.np <= MAX_PASSES

    # ---- cutting the columns ----
    def _cut(self, split_factor):
        sizes, R = self.sizes, sum(self.sizes)
        cap = self.slots * MAX_PASSES                              # positions of the tree
        if len(sizes) > cap:
            raise ValueError(f"{len(sizes)} columns exceed the {cap} positions of the forward tree at k = {self.k}")
        if self.world == 1:
            return [(c, 0, r) for c, r in enumerate(sizes)]
        if R < self.world:
            raise ValueError(f"{R} rows cannot be dealt over {self.world} owners")
        target = split_factor * R / self.nb                         # rows a piece should not exceed
        n_cut = [max(1, math.ceil(r / target)) for r in sizes]
        while sum(n_cut) > cap:                                     # too many pieces: undo the finest cuts
            c = max(range(len(sizes)), key=lambda c: (n_cut[c] > 1, -sizes[c] / n_cut[c]))
            if n_cut[c] == 1:
                break
            n_cut[c] -= 1
        while sum(n_cut) < self.world:                              # fewer pieces than owners: cut the largest piece again
            c = max(range(len(sizes)), key=lambda c: sizes[c] / n_cut[c] if n_cut[c] < sizes[c] else -1)
            n_cut[c] += 1
        pieces = []
        for c, (r, n) in enumerate(zip(sizes, n_cut)):
            lo = 0
            for j in range(n):
                rows = r // n + (1 if j < r % n else 0)
                pieces.append((c, lo, rows))
                lo += rows
        return pieces

    # ---- dealing the pieces over the blocks ----
    def _deal(self, global_batch):
        nb = self.nb
        if self.world == 1:
            return [list(range(len(self.pieces)))]
        # no block takes more passes than the fullest one must: a further pass is a further gather slot for EVERY owner
        cap = self.sl * max(1, -(-len(self.pieces) // (nb * self.sl)))
        R, F = sum(self.sizes), len(self.sizes)
        # cost of a piece: rows (HBM footprint) and its share of its column's occurrences (gather and update work: every sample
        # has one index per column, a piece of a column sees the samples whose index falls into it).  The hand-off chains of
        # the small hot columns are as long wherever their rows live, so they are not a load to balance; their occurrences are.
        def cost(p):
            c, _, rows = self.pieces[p]
            return np.array([rows / R, rows / self.sizes[c] / F])
        costs = [cost(p) for p in range(len(self.pieces))]
        tot = np.maximum(np.sum(costs, axis=0), 1e-30)
        costs = [c / tot * nb for c in costs]                       # a perfectly balanced block carries 1.0 of each component
        order = sorted(range(len(self.pieces)), key=lambda p: -float(costs[p].max()))
        load = [np.zeros(2) for _ in range(nb)]
        blocks = [[] for _ in range(nb)]
        # ranks with two blocks: their blocks share the rank's budget
        rank_load = [np.zeros(2) for _ in range(self.world)]
        for i, p in enumerate(order):
            empty_ranks = [g for g in range(self.world) if not any(blocks[b] for b in range(nb) if self.block_owner[b] == g)]
            left = len(order) - i
            best, best_key = None, None
            for b in range(nb):
                if len(blocks[b]) >= cap:
                    continue
                g = self.block_owner[b]
                if len(empty_ranks) >= left and g not in empty_ranks:
                    continue                                        # the remaining pieces are needed to give every rank a row
                after = rank_load[g] + costs[p]
                key = (float(after.max()), float(after.sum()), len(blocks[b]), b)
                if best_key is None or key < best_key:
                    best, best_key = b, key
            if best is None:
                raise ValueError("the pieces do not fit the positions of the forward tree")
            blocks[best].append(p)
            load[best] += costs[p]
            rank_load[self.block_owner[best]] += costs[p]
        for b in blocks:
            b.sort()                                                # a fixed local order: by column, then by row range
        self.rank_load = rank_load                                  # per rank: (rows, occurrences) in units of a block's even share
        return blocks

    # ---- tables ----
    def _fields_of_block(self, b):
        """The block's np * sl local fields in position order (pass-major): (col, base, rows), holes as empty fields."""
        ids = self.block_pieces[b] + [None] * (self.np * self.sl - len(self.block_pieces[b]))
        return [self.pieces[i] if i is not None else (0, 0, 0) for i in ids]

    def owner_fields(self, g):
        """Rank g's local fields [(col, base, rows)]: local field (lb * np + p) * sl + s sits at position
        p * slots + (block_first[g] + lb) * sl + s of the whole tree."""
        out = []
        for lb in range(self.block_count[g]):
            out += self._fields_of_block(self.block_first[g] + lb)
        return out

    def whole_fields(self):
        """Every position of the tree in order (position j = pass * slots + lane group): (col, base, rows)."""
        out = [(0, 0, 0)] * (self.np * self.slots)
        for b in range(self.nb):
            for l, pc in enumerate(self._fields_of_block(b)):
                p, s = divmod(l, self.sl)
                out[p * self.slots + b * self.sl + s] = pc
        return out

    def _table(self, fields, layout, device, ftrl):
        if self.world == 1:                                          # the ordinary table: column f is field f
            return FlatTable(self.sizes, self.k, layout=layout, device=device, ftrl=ftrl)
        return FlatTable([r for _, _, r in fields], self.k, layout=layout, device=device, ftrl=ftrl,
                         field_cols=[c for c, _, _ in fields], field_base=[b for _, b, _ in fields], n_cols=len(self.sizes))

    def table_for_owner(self, g, layout="weights", device=None, ftrl=None):
        t = self._table(self.owner_fields(g), layout, device, ftrl)
        t.plan_fields = self.owner_fields(g) if self.world > 1 else [(c, 0, r) for c, r in enumerate(self.sizes)]
        return t

    def table_whole(self, layout="weights", device=None, ftrl=None):
        """ONE table with every piece at its position: what G owners are bit-identical to."""
        t = self._table(self.whole_fields(), layout, device, ftrl)
        t.plan_fields = self.whole_fields() if self.world > 1 else [(c, 0, r) for c, r in enumerate(self.sizes)]
        return t

    def rows_per_owner(self):
        return [sum(r for _, _, r in self.owner_fields(g)) for g in range(self.world)]

    def describe(self):
        rows = self.rows_per_owner()
        return (f"{len(self.sizes)} columns -> {len(self.pieces)} pieces in {self.nb} blocks of {self.sl} lane groups x {self.np} passes; "
                f"rows per owner {rows} (max/min {max(rows) / max(1, min(rows)):.2f})")


class WholeOwnerPlan:
    """The same pieces at the same positions held by ONE rank that owns every block: the plan a single process runs the owner
    step with when it must give the bits of `plan.world` ranks (tests; a one-GPU replay of a multi-GPU run)."""

    def __init__(self, plan):
        self.inner, self.world, self.sizes, self.k, self.kp = plan, 1, plan.sizes, plan.k, plan.kp
        self.nb, self.sl, self.np, self.slots = plan.nb, plan.sl, plan.np, plan.slots
        self.block_count, self.block_first = [plan.nb], [0]

    def owner_fields(self, g):
        return sum((self.inner._fields_of_block(b) for b in range(self.nb)), [])

    def table_for_owner(self, g, layout="weights", device=None, ftrl=None):
        fields = self.owner_fields(0)
        if self.inner.world == 1:
            t = FlatTable(self.sizes, self.k, layout=layout, device=device, ftrl=ftrl)
            t.plan_fields = [(c, 0, r) for c, r in enumerate(self.sizes)]
            return t
        t = FlatTable([r for _, _, r in fields], self.k, layout=layout, device=device, ftrl=ftrl,
                      field_cols=[c for c, _, _ in fields], field_base=[b for _, b, _ in fields], n_cols=len(self.sizes))
        t.plan_fields = fields
        return t


def load_columns(table, first_list, second_list):
    """Fill a plan table from per-COLUMN reference weights (first_list[c] [rows_c, 1], second_list[c] [rows_c, k]): every field
    takes its row range of its column (FlatTable.load_reference wants one tensor per field)."""
    fields = table.plan_fields
    first = [torch.as_tensor(first_list[c], dtype=torch.float32).reshape(-1, 1)[b:b + r] for c, b, r in fields]
    second = [torch.as_tensor(second_list[c], dtype=torch.float32).reshape(-1, table.k)[b:b + r] for c, b, r in fields]
    table.load_reference(first, second)


def export_columns(table, sizes):
    """-> {(col, base, rows): rows tensor [rows, row_stride] (CPU)} for the non-empty fields of a plan table."""
    out, rows = {}, table.rows.detach().cpu()
    for f, (c, b, r) in enumerate(table.plan_fields):
        if r:
            lo = int(table.offsets_host[f])
            out[(c, b, r)] = rows[lo:lo + r]
    return out
