"""Online DeepFM / NFM mini-batch step on device tensors, data-parallel over the GPUs of one node
(BASELINE.json configs[3]: bi-interaction + 3 x 256 relu MLP, SGD, RCCL all-reduce of the dense gradients).

Per step and rank (B local samples, G ranks, everything scaled by 1 / (G B) so that the result is the G B-sample step):
  1. forward of the tables on the local slice (k_fm_forward): S, bi, FM logit
  2. the MLP on bi (fmx_mlp_section: fp32 MFMA GEMMs for forward, loss and backward; a PyTorch autograd form of the same
     section is kept as the tests' comparison and for networks with unequal layer widths)
     -> loss, dL/dlogit, dL/dbi, local MLP gradients
  3. ONE fused all-reduce (sum) of the flattened MLP gradients            -- the dense exchange
  4. all-gather of the low-rank factors of the row gradients (idx, S, dz, dL/dbi)  -- 288 B per sample, no rows move
  5. identical sort + row-reduced table update on every replica (k_sort_occ, k_fm_update, k_fm_fixup), SGD on the MLP
Replicas stay bit-identical in the tables (deterministic kernels) and equal in the MLP up to the all-reduce's own
summation order.  The compute behind the table is a small backend interface (HIP in the product; tests inject an
oracle-backed one to exercise this logic on CPU with gloo).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F


class HipDeepBackend:
    def __init__(self, engine, hyper, rule):
        self.e, self.hyper, self.rule = engine, hyper, rule

    def forward(self, idx):
        """-> (S [B,kp], bi [B,kp], sfirst [B], logit_fm [B]) views valid until the next forward."""
        B = self.e.forward(self.hyper, idx, None, want_first=False, want_bi=True)
        e = self.e
        return e.S[:B], e.bi[:B], e.sfirst[:B], e.logit[:B]

    def bias(self):
        return self.e.table.bias_weight()

    def mlp_section(self, flat, gflat, k, hidden, n_layers, loss, bi, base, y, inv_b, lr_apply):
        return self.e.mlp_section(flat, gflat, k, hidden, n_layers, loss, bi, base, y, bi.shape[0], inv_b, lr_apply)

    def start_sort(self, idx_g):
        """The occurrence sort only needs the (gathered) indices: it runs on a side stream beside the forward pass and
        the MLP section."""
        e = self.e
        e._ensure(idx_g.shape[0])
        cur = torch.cuda.current_stream(e.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=e.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            e.sort(idx_g)
        self._sorted = object()          # a handle for THIS sort: update() takes it back (not the tensor's address, which can be recycled)
        return self._sorted

    def update(self, idx_g, S_g, dz_g, gbi_g, fm_term, inv_b, sorted_handle=None):
        e = self.e
        GB = idx_g.shape[0]
        e._ensure(GB)
        if sorted_handle is not None and sorted_handle is getattr(self, "_sorted", None):
            torch.cuda.current_stream(e.device).wait_stream(self._side)
        else:
            e.sort(idx_g)
        self._sorted = None
        e.update(self.hyper, self.rule, GB, None, dz_g, dz_g if fm_term else None, gbi_g, inv_b=inv_b, with_loss=False, S=S_g)


class DeepFMTrainer:
    def __init__(self, backend, hidden_layers, k, kp, mlp_lr, fm_term=True, loss="logits", group=None, use_graph=False,
                 native_mlp=True):
        """hidden_layers: list of nn.Linear on the device (k -> H -> ... -> H); the network's logit contribution is the
        sum of the last activation (reference deepfm_adam.py:82-88).  fm_term=False gives NFM (nfm_adam.py:78-88).
        native_mlp: run the MLP section through fmx_mlp_section (needs a backend that has it and equal layer widths);
        otherwise PyTorch autograd, optionally replayed as a graph (use_graph)."""
        self.backend, self.layers, self.k, self.kp = backend, list(hidden_layers), k, kp
        self.mlp_lr, self.fm_term, self.loss, self.group = mlp_lr, fm_term, loss, group
        self.use_graph = use_graph      # replay the PyTorch MLP section as a captured graph (static batch size)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for layer in self.layers for p in (layer.weight, layer.bias)]
        self._bufs = {}
        widths = {layer.out_features for layer in self.layers} | {layer.in_features for layer in self.layers[1:]}
        self.native = bool(native_mlp and hasattr(backend, "mlp_section") and len(widths) == 1 and
                           self.layers[0].in_features == k and len(self.layers) <= 8 and self.params[0].is_cuda)
        if self.native:   # one flat buffer (W_l then b_l per layer) that the nn.Linear parameters become views of
            self.hidden = self.layers[0].out_features
            self.flat = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
            self.gflat = torch.zeros_like(self.flat)
            off = 0
            for p in self.params:
                n = p.numel()
                p.data = self.flat[off:off + n].view_as(p)
                off += n

    def prepare_stream(self, idx_pool, y_pool, loss_out=None, stream=None):
        """-> run(n_steps): the steps of step() over a device-resident pool (step s: batch s mod n_pool) issued from ONE foreign call
        (fmx_deepfm_stream) -- through step() the trainer is bound by its host side (84 us of calls per step for 67 us of
        kernels at configs[3]).  One rank, the native MLP section; DeepFM or NFM (fm_term=False: weights-layout tables); identical results."""
        if not (self.native and self.world == 1 and hasattr(self.backend, "e")):
            raise ValueError("prepare_stream: one rank and the native MLP section only; use step()")
        be = self.backend
        if not self.fm_term and be.e.table.layout != "weights":
            raise ValueError("prepare_stream: NFM (fm_term=False) needs tables in the weights layout; use step()")
        return be.e.prepare_deepfm_stream(be.hyper, be.rule, self.loss, self.flat, self.gflat, self.k, self.hidden, len(self.layers),
                                          self.mlp_lr, idx_pool, y_pool, loss_out=loss_out, stream=stream, fm_term=self.fm_term)

    def _gathered(self, name, local):
        if self.world == 1:
            return local
        shape = (self.world * local.shape[0],) + tuple(local.shape[1:])
        key = (name, shape, local.dtype, local.device)
        out = self._bufs.get(key)
        if out is None:
            out = self._bufs[key] = torch.empty(shape, dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.group) == "gloo":     # rehearsal on a shared GPU only
            host = torch.empty(shape, dtype=local.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    # ---- the MLP section (forward, loss, backward) -- optionally captured once into a graph and replayed: at 3 x 256 the
    #      ~40 small PyTorch launches cost more host time than the GEMMs take on the GPU ----
    def _mlp_section(self, bi_in, base_in, y_in, inv_b):
        bi_leaf = bi_in.detach().clone().requires_grad_(True)
        base = base_in.detach().clone().requires_grad_(True)
        for p in self.params:
            p.grad = None
        x = bi_leaf
        for layer in self.layers:
            x = F.relu(layer(x))
        out = base + x.sum(1)
        z = torch.sigmoid(out) if self.loss == "sigmoid" else out
        loss = F.binary_cross_entropy_with_logits(z, y_in, reduction="sum") * inv_b
        loss.backward()
        flat = torch.cat([p.grad.reshape(-1) for p in self.params])
        gbi = bi_leaf.grad
        if self.k != self.kp:
            gbi = F.pad(gbi, (0, self.kp - self.k))
        return loss.detach(), base.grad.contiguous(), gbi.contiguous(), flat

    def _mlp_section_graphed(self, bi_in, base_in, y_in, inv_b):
        key = (bi_in.shape[0], inv_b)
        if getattr(self, "_graph_key", None) != key:
            B = bi_in.shape[0]
            dev = bi_in.device
            self._g_in = (torch.empty((B, self.k), device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                       # warm-up outside the capture (allocator, rocBLAS handles)
                for src, dst in zip((bi_in, base_in, y_in), self._g_in):
                    dst.copy_(src)
                for _ in range(2):
                    self._mlp_section(*self._g_in, inv_b)
            torch.cuda.current_stream(dev).wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._g_out = self._mlp_section(*self._g_in, inv_b)
            self._graph_key = key
        for src, dst in zip((bi_in, base_in, y_in), self._g_in):
            dst.copy_(src)
        self._graph.replay()
        return self._g_out

    def step(self, idx_local, y_local):
        """One exact data-parallel step; returns this rank's share of the global mean loss (sum over ranks = the loss)."""
        B = idx_local.shape[0]
        inv_b = 1.0 / (B * self.world)
        idx_g = self._gathered("idx", idx_local)
        handle = self.backend.start_sort(idx_g) if (hasattr(self.backend, "start_sort") and idx_g.is_cuda) else None
        S, bi, sfirst, logit_fm = self.backend.forward(idx_local)
        base_in = logit_fm if self.fm_term else sfirst + self.backend.bias()
        applied = False
        if self.native:
            applied = self.world == 1      # single rank: SGD on the MLP inside the reduction kernel
            loss, dz, gbi = self.backend.mlp_section(self.flat, self.gflat, self.k, self.hidden, len(self.layers), self.loss,
                                                     bi, base_in.contiguous(), y_local, inv_b,
                                                     self.mlp_lr if applied else 0.0)
            loss, flat = loss[0], self.gflat
        else:
            section = self._mlp_section_graphed if (self.use_graph and bi.is_cuda) else self._mlp_section
            loss, dz, gbi, flat = section(bi[:, :self.k], base_in, y_local, inv_b)
        # ---- the dense exchange: one bucket ----
        if self.world > 1:
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":
                host = flat.cpu()
                dist.all_reduce(host, group=self.group)
                flat = host.to(flat.device)
            else:
                dist.all_reduce(flat, group=self.group)
        # ---- the sparse exchange: low-rank factors only ----
        S_g = self._gathered("S", S)
        dz_g = self._gathered("dz", dz)
        gbi_g = self._gathered("gbi", gbi)
        if handle is not None:
            self.backend.update(idx_g, S_g.contiguous(), dz_g.contiguous(), gbi_g, self.fm_term, inv_b, sorted_handle=handle)
        else:
            self.backend.update(idx_g, S_g.contiguous(), dz_g.contiguous(), gbi_g, self.fm_term, inv_b)
        if self.native:
            if not applied:
                self.flat.sub_(flat, alpha=self.mlp_lr)
            return loss
        with torch.no_grad():
            off = 0
            for p in self.params:
                n = p.numel()
                p -= self.mlp_lr * flat[off:off + n].view_as(p)
                off += n
        return loss


class OwnerDeepFMTrainer:
    """Online DeepFM / NFM on FIELD OWNERS (fmx.owner, fmx.plan): the tables and their update work shard over the ranks, the MLP is
    replicated.  Per step and rank (B local samples, G ranks; BASELINE.json configs[3]):

        (ahead of time)  all-gather idx, sort the owned pieces' occurrences                       prefetch(): 1 collective, off the critical path
        partial forward  the owned blocks' sub-trees for every sample of the global batch
        all-to-all       the records of this rank's samples                                         collective 1 of the critical path
        finish           S, bi, first-order sum, FM logit of the B local samples
        MLP section      forward, loss, backward on bi (fmx_mlp_section, fp32 MFMA) -> dlogit, dL/dbi, local MLP gradients
        all-reduce       the flattened MLP gradients (3 x 256: 544 KB), ONE bucket, on a side stream:   beside the table update
                         it does not touch the tables -- then SGD on the replicated MLP
        all-gather       ONE record per sample: S | dlogit | dL/dbi (144 B at k = 16)               collective 2 of the critical path
        update           the owned rows (every row by its owner only)

    The dense exchange is what north_star names (reference deepfm_adam.py:38-44,82-88: nn.Linear layers, replicated); no embedding
    row crosses xGMI.  The tables end bit-identical on every rank to the one-table step while the MLP parameters agree (they differ
    by the all-reduce's summation order from the second step on: 1e-5, tests/test_dp_gpu.py)."""

    def __init__(self, backend, hidden_layers, k, mlp_lr, fm_term=True, loss="logits", group=None):
        be = self.backend = backend
        self.layers, self.k, self.kp = list(hidden_layers), k, be.kp
        self.mlp_lr, self.fm_term, self.loss, self.group = mlp_lr, fm_term, loss, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.params = [p for layer in self.layers for p in (layer.weight, layer.bias)]
        widths = {layer.out_features for layer in self.layers} | {layer.in_features for layer in self.layers[1:]}
        if not (len(widths) == 1 and self.layers[0].in_features == k and len(self.layers) <= 8):
            raise ValueError("OwnerDeepFMTrainer runs the MLP through fmx_mlp_section: equal hidden widths, at most 8 layers")
        self.hidden = self.layers[0].out_features
        self.flat = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        self.gflat = torch.zeros_like(self.flat)
        off = 0
        for p in self.params:
            n = p.numel()
            p.data = self.flat[off:off + n].view_as(p)
            off += n
        from .owner import FieldOwnerFM
        self.fo = FieldOwnerFM(be, group)            # its exchanges (all-gather, all-to-all) and its prefetch slots
        self._side, self._ar_done = None, None

    def prefetch(self, idx_next):
        return self.fo.prefetch(idx_next)

    def step(self, idx_local, y_local, token=None):
        """One exact step on the global batch; returns this rank's share of the global mean loss (sum over ranks = the loss)."""
        be, fo, B = self.backend, self.fo, idx_local.shape[0]
        dev = idx_local.device
        inv_b = 1.0 / (B * self.world)
        gpu = idx_local.is_cuda                       # (the tests drive the exchange logic on CPU with an oracle-backed backend)
        cur = torch.cuda.current_stream(dev) if gpu else None
        kw = {"stream": cur} if gpu else {}
        pref = fo._pref.pop(id(token), None) if token is not None else None
        if pref is not None:
            _, idx_all, slot = pref
            cur.wait_event(fo._pf[1][slot])
        else:
            idx_all, slot = fo._all_gather("idx", idx_local), None
        parts = be.partial_forward(idx_all, B, **kw)
        mine = fo._all_to_all("parts", parts, B)
        rec, bi, sfirst, logit_fm = be.finish_bi(mine, **kw)
        base_in = logit_fm if self.fm_term else sfirst + be.bias_weight()
        if self._ar_done is not None:                 # the previous step's SGD on the MLP (side stream) comes first
            cur.wait_event(self._ar_done)
        alone = self.world == 1
        loss, dz, gbi = be.mlp_section(self.flat, self.gflat, self.k, self.hidden, len(self.layers), self.loss, bi,
                                       base_in.contiguous(), y_local, B, inv_b, self.mlp_lr if alone else 0.0)
        if not alone and gpu:                         # the dense exchange: one bucket, beside the table update
            if self._side is None:
                self._side, self._ar_done = torch.cuda.Stream(device=dev), torch.cuda.Event()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                if self._gloo:
                    host = self.gflat.cpu()
                    dist.all_reduce(host, group=self.group)
                    self.gflat.copy_(host)
                else:
                    dist.all_reduce(self.gflat, group=self.group)
                self.flat.sub_(self.gflat, alpha=self.mlp_lr)
            self._ar_done.record(self._side)
        elif not alone:
            dist.all_reduce(self.gflat, group=self.group)
            self.flat.sub_(self.gflat, alpha=self.mlp_lr)
        kp = self.kp
        rec[:, kp] = dz                               # the record: S (written by the finish) | dlogit | dL/dbi
        rec[:, kp + 4:] = gbi
        rec_all = fo._all_gather("deep_rec", rec)
        be.update_deep(idx_all, rec_all, self.fm_term, inv_b, slot, **kw) if slot is not None else be.update_deep(idx_all, rec_all, self.fm_term, inv_b, **kw)
        if slot is not None:
            fo._pf[2][slot].record(cur)
        return loss[0]

    def finish(self):
        """Order the caller's stream behind the last step's MLP update (call before reading the MLP parameters)."""
        if self._ar_done is not None:
            torch.cuda.current_stream(self.backend.device).wait_event(self._ar_done)
