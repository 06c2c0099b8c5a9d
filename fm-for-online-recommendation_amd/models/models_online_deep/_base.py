"""Shared machinery of the five embedding-table online models (hot path A).

The reference implements each class as ~39 x 2 nn.Embedding gathers, Python sums, autograd into dense table-sized
gradients and a fresh torch.optim.Adam over every parameter per call (reference models/models_online_deep/*.py).
Here the tables live in one fmx.FlatTable in HBM and every table-touching step is three gfx950 kernels
(sort occurrences -> gather + bi-interaction forward -> row-reduced backward with the fused per-row update);
PyTorch is left with the tiny dense MLP on top of the bi-interaction vector and with plumbing.

Reference behaviours kept on purpose (SURVEY.md sections 2a, 3.3, 7 "hard parts"; all are results-bearing):
  * a fresh Adam per call == p -= lr * g / (|g| + 1e-8) on coordinates with a non-zero gradient ('signadam' rule);
  * the loss is BCE-with-logits of sigmoid(logit) in some (class, method) pairs ("double sigmoid");
  * DeepFM evaluates second_order twice, so its table gradient is the FM-term gradient plus the MLP-input gradient;
  * the ONN classes' fit() trains only the hidden layers and alpha (Hedge); their predict() applies a second sigmoid;
  * `bias` and `n` appear in state_dict(); the oracle is the reference's CPU behaviour, where `bias` IS trained.
There is no CPU path: constructing a model without a ROCm GPU raises.
"""
from time import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import fmx


class _FieldView:
    """Stands where the reference has an nn.Embedding: `.weight` is a strided view into the flat table."""

    def __init__(self, table, f, second):
        self._table, self._f, self._second = table, f, second

    @property
    def weight(self):
        return self._table.field_V(self._f) if self._second else self._table.field_w(self._f)


class OnlineFMBase(nn.Module):
    _name = "FMAdam"
    _has_mlp = False          # DeepFM / NFM: relu MLP on the bi-interaction vector
    _fm_term_in_forward = True  # False for NFM: forward() has no sum_d bi_d term
    _onn = False
    _loss_update_embedding = "logits"
    _loss_fit = "sigmoid"

    def __init__(self, feature_sizes, embedding_size=4, num_hidden_layers=0, neuron_per_hidden_layer=0, batch_size=1,
                 num_classes=1, b=0.99, n=0.01, s=0.2, use_cuda=True, update_rule="signadam", ftrl=None):
        super().__init__()
        if not (use_cuda and torch.cuda.is_available()):
            raise RuntimeError(f"{self._name}: this build runs the hot path in gfx950 kernels only -- it needs use_cuda=True "
                               "and a ROCm GPU (there is no CPU or PyTorch fallback; use the reference for CPU runs)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.field_size = len(feature_sizes)
        self.feature_sizes = feature_sizes
        self.embedding_size = embedding_size
        self.num_hidden_layers = num_hidden_layers
        self.neuron_per_hidden_layer = neuron_per_hidden_layer
        self.batch_size = batch_size
        self.num_classes = num_classes
        self.dtype = torch.long
        self.update_rule = update_rule
        if update_rule not in ("signadam", "sgd", "ftrl"):
            raise ValueError(update_rule)

        # ---- parameter initialisation in the reference's RNG order (reference fm_adam.py:26-32,
        #      deepfm_onn.py:30-46), so that the same torch seed gives the same model ----
        if self._onn:
            bias0 = torch.rand(1)
            self.b = torch.nn.Parameter(torch.tensor(b))
            self.s = torch.nn.Parameter(torch.tensor(s), requires_grad=False)
        else:
            bias0 = torch.tensor(b)
        self.n = torch.nn.Parameter(torch.tensor(n), requires_grad=False)
        first = [nn.Embedding(fs, 1).weight.data for fs in feature_sizes]
        second = [nn.Embedding(fs, embedding_size).weight.data for fs in feature_sizes]
        self._bias_shape = tuple(bias0.shape)

        self._ftrl = dict(alpha=0.05, beta=1.0, l1=0.0, l2=0.0)
        if ftrl:
            self._ftrl.update(ftrl)
        self._table = fmx.FlatTable(feature_sizes, embedding_size, layout="ftrl" if update_rule == "ftrl" else "weights",
                                    device=self.device, ftrl=self._ftrl)
        self._load_weights(first, second, bias0)
        del first, second
        self._engine = fmx.FMEngine(self._table, max_batch=max(int(batch_size), 64))
        self._hyper = fmx.Hyper(lr=float(n), eps=1e-8, **self._ftrl)

        layers = []
        if self._has_mlp:
            layers.append(nn.Linear(embedding_size, neuron_per_hidden_layer))
            for _ in range(num_hidden_layers - 1):
                layers.append(nn.Linear(neuron_per_hidden_layer, neuron_per_hidden_layer))
            self.hidden_layers = nn.ModuleList(layers).to(self.device)
            # one flat buffer behind every hidden parameter (W_0, b_0, W_1, b_1, ...): the fused MLP kernel updates it in
            # place and the nn.Linear modules (state_dict, the PyTorch path for large shapes) see the same memory
            flat = torch.cat([p.detach().reshape(-1) for layer in self.hidden_layers for p in (layer.weight, layer.bias)])
            self._mlp_flat = flat.contiguous()
            off = 0
            for layer in self.hidden_layers:
                for p in (layer.weight, layer.bias):
                    p.data = self._mlp_flat[off:off + p.numel()].view(p.shape)
                    off += p.numel()
        if self._onn:
            # a plain device tensor (on a GPU the reference's Parameter(...).to(device) is one too)
            self.alpha = torch.full((num_hidden_layers,), 1 / (num_hidden_layers + 1), dtype=torch.float32,
                                    device=self.device)
        self.first_order_embeddings = [_FieldView(self._table, f, False) for f in range(self.field_size)]
        self.second_order_embeddings = [_FieldView(self._table, f, True) for f in range(self.field_size)]

    # ------------------------------------------------------------------------------------------------------
    # table <-> reference-shaped weights
    # ------------------------------------------------------------------------------------------------------
    def _load_weights(self, first, second, bias):
        self._table.load_reference(first, second)
        self._table.set_bias_weight(float(torch.as_tensor(bias).reshape(-1)[0]))

    def _export_weights(self):
        first, second = self._table.export_reference()
        return first, second, self._table.bias_weight().detach().cpu()

    @property
    def bias(self):
        return self._table.bias_weight().reshape(self._bias_shape)

    def state_dict(self, *args, **kwargs):
        """The reference's keys and shapes: first_order_embeddings.{i}.weight [size_i,1],
        second_order_embeddings.{i}.weight [size_i,k], hidden_layers.{j}.weight/.bias, bias, n (+ b, s, alpha)."""
        first, second, bias = self._export_weights()
        sd = {"bias": bias.reshape(self._bias_shape).clone()}
        if self._onn:
            sd["b"] = self.b.detach().clone()
        sd["n"] = self.n.detach().clone()
        if self._onn:
            sd["s"] = self.s.detach().clone()
        for i in range(self.field_size):
            sd[f"first_order_embeddings.{i}.weight"] = first[i]
        for i in range(self.field_size):
            sd[f"second_order_embeddings.{i}.weight"] = second[i]
        if self._has_mlp:
            for j, layer in enumerate(self.hidden_layers):
                sd[f"hidden_layers.{j}.weight"] = layer.weight.detach().cpu().clone()
                sd[f"hidden_layers.{j}.bias"] = layer.bias.detach().cpu().clone()
        if self._onn:
            sd["alpha"] = self.alpha.detach().cpu().clone()
        return sd

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v) for k, v in state_dict.items()}
        want = set(self.state_dict().keys())
        if strict and set(sd.keys()) != want:
            raise RuntimeError(f"state_dict keys differ: missing {sorted(want - set(sd))}, unexpected {sorted(set(sd) - want)}")
        first = [sd[f"first_order_embeddings.{i}.weight"].float().cpu() for i in range(self.field_size)]
        second = [sd[f"second_order_embeddings.{i}.weight"].float().cpu() for i in range(self.field_size)]
        self._load_weights(first, second, sd["bias"].float().cpu())
        with torch.no_grad():
            self.n.copy_(sd["n"].float().cpu())
            self._hyper = fmx.Hyper(lr=float(self.n), eps=1e-8, **self._ftrl)
            if self._has_mlp:
                for j, layer in enumerate(self.hidden_layers):
                    layer.weight.copy_(sd[f"hidden_layers.{j}.weight"].float().to(self.device))
                    layer.bias.copy_(sd[f"hidden_layers.{j}.bias"].float().to(self.device))
            if self._onn:
                self.b.copy_(sd["b"].float().cpu())
                self.s.copy_(sd["s"].float().cpu())
                self.alpha = sd["alpha"].float().clone().to(self.device)

    # ---- the optimizer state of update_rule='ftrl' (an extension: the reference has no such rule).  state_dict() keeps
    #      the reference's keys, i.e. the derived weights only; a resumed FTRL run also needs every coordinate's (z, n),
    #      or all per-coordinate learning rates restart from n = 0 ----
    def ftrl_state_dict(self):
        """{'zV' [R,k], 'nV' [R,k], 'zw' [R], 'nw' [R], 'bias_zn' [2], 'V_cached' [R,k], 'w_cached' [R]} on the CPU (rows flat
        over the fields, as in the table); None for the other rules."""
        if self.update_rule != "ftrl":
            return None
        zV, nV, zw, nw = self._table.export_ftrl_state()
        # V_cached / w_cached: the derived weights exactly as the last update stored them (the kernels derive them with
        # 1-ulp v_rcp_f32 / v_sqrt_f32; re-deriving them on the host would differ in the last bit and the resumed run with it)
        return {"zV": zV, "nV": nV, "zw": zw, "nw": nw, "bias_zn": self._table.bias.detach().cpu().clone(),
                "V_cached": self._table.V.detach().cpu().clone(), "w_cached": self._table.w.detach().cpu().clone()}

    def load_ftrl_state_dict(self, st):
        """Restore (z, n) bit for bit; V, w and the bias weight are re-derived from it exactly as an update does."""
        if self.update_rule != "ftrl":
            raise ValueError("load_ftrl_state_dict: this model does not use update_rule='ftrl'")
        self._table.load_ftrl_state(st["zV"], st["nV"], st["zw"], st["nw"])
        self._table.bias.copy_(torch.as_tensor(st["bias_zn"], dtype=torch.float32).to(self.device))
        if st.get("V_cached") is not None:
            self._table.V.copy_(torch.as_tensor(st["V_cached"], dtype=torch.float32).to(self.device))
            self._table.w.copy_(torch.as_tensor(st["w_cached"], dtype=torch.float32).to(self.device))

    # pickle support (reference main_experiment.py:160-162 pickles the whole model): tensors go through the CPU
    def __getstate__(self):
        return {"ctor": dict(feature_sizes=list(self.feature_sizes), embedding_size=self.embedding_size,
                             num_hidden_layers=self.num_hidden_layers,
                             neuron_per_hidden_layer=self.neuron_per_hidden_layer, batch_size=self.batch_size,
                             num_classes=self.num_classes, update_rule=self.update_rule, ftrl=dict(self._ftrl)),
                "cls": self._name, "state_dict": {k: v.cpu() for k, v in self.state_dict().items()},
                "ftrl_state": self.ftrl_state_dict()}

    def __setstate__(self, state):
        ctor = state["ctor"]
        OnlineFMBase.__init__(self, ctor["feature_sizes"], embedding_size=ctor["embedding_size"],
                              num_hidden_layers=ctor["num_hidden_layers"],
                              neuron_per_hidden_layer=ctor["neuron_per_hidden_layer"], batch_size=ctor["batch_size"],
                              num_classes=ctor["num_classes"], update_rule=ctor["update_rule"], ftrl=ctor["ftrl"])
        self.load_state_dict(state["state_dict"])
        if state.get("ftrl_state") is not None:       # after the weights: (z, n) is the state, V / w / bias follow from it
            self.load_ftrl_state_dict(state["ftrl_state"])

    # ------------------------------------------------------------------------------------------------------
    # forward pieces (reference deepfm_adam.py:46-89)
    # ------------------------------------------------------------------------------------------------------
    def _inputs(self, Xi, Xv, Y=None):
        """-> (idx int32 [B, F], xv fp32 [B, F] or None when every value is 1, y fp32 [B] or None), all on the device.
        Nested lists (the reference's convention, fm_adam.py:35-36) are converted and range-checked on the host.  Arrays
        and tensors take the fast path: no list conversion; a CUDA tensor is used where it lies (int32 contiguous: as is),
        Xv may be None (all ones); the range check is the kernels' (strict_index_check: the flag is read after the step,
        one small device-to-host copy; set it to False to leave the stream asynchronous and call check_index_flag())."""
        if torch.is_tensor(Xi) or isinstance(Xi, np.ndarray):
            return self._inputs_fast(Xi, Xv, Y)
        idx, xv = fmx.normalize_inputs(Xi, Xv, self.field_size, self.feature_sizes)
        return self._engine.to_device(idx, xv, Y)

    strict_index_check = True

    def _inputs_fast(self, Xi, Xv, Y):
        dev, F = self.device, self.field_size

        def to_dev(a, dtype):
            if a is None:
                return None
            t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
            if not t.is_cuda:
                if t.dtype != dtype:
                    t = t.to(dtype)
                t = t.contiguous()
                t = (t.pin_memory() if not t.is_pinned() else t).to(dev, non_blocking=True)
            elif t.dtype != dtype:
                t = t.to(dtype)
            return t.contiguous()
        # the reference's shapes only: [B, F], [B, F, 1] (Xi), or one sample [F] -- an array of another width whose size
        # happens to divide by F must not be re-cut into wrong rows
        shp = tuple(Xi.shape)
        if not ((len(shp) == 2 and shp[1] == F) or (len(shp) == 3 and shp[1:] == (F, 1)) or shp == (F,) or shp == (F, 1)):
            raise ValueError(f"Xi of shape {shp}: expected [B, {F}] ([B, {F}, 1], or one sample [{F}])")
        if (torch.is_tensor(Xi) and Xi.dtype == torch.int64) or (isinstance(Xi, np.ndarray) and Xi.dtype.itemsize > 4):
            # an index beyond int32 would wrap in the cast below and could land on a valid row: checked before the cast
            big = bool((Xi >= 2 ** 31).any()) if torch.is_tensor(Xi) else bool((Xi >= 2 ** 31).any())
            if big:
                raise IndexError("index out of range in self")
        idx_d = to_dev(Xi, torch.int32).reshape(-1, F)
        xv_d = to_dev(Xv, torch.float32)
        if xv_d is not None:
            xv_d = xv_d.reshape(-1, F)
            if xv_d.shape != idx_d.shape:
                raise ValueError(f"Xi {tuple(idx_d.shape)} and Xv {tuple(xv_d.shape)} disagree")
        y_d = to_dev(Y, torch.float32)
        if y_d is not None:
            y_d = y_d.reshape(-1)
        self._fast_inputs_pending = True
        return idx_d, xv_d, y_d

    def check_index_flag(self):
        """Raise IndexError if a kernel met an index outside its field since the last check (array / tensor inputs are
        range-checked on the device; nested lists on the host, before anything is launched)."""
        self._fast_inputs_pending = False
        self._engine.check_error_flag()

    def _after_step(self):
        if getattr(self, "_fast_inputs_pending", False) and self.strict_index_check:
            self.check_index_flag()

    def _fm_forward(self, Xi, Xv):
        idx_d, xv_d, _ = self._inputs(Xi, Xv)
        B = self._engine.forward(self._hyper, idx_d, xv_d)
        self._after_step()
        return B

    def first_order(self, Xi, Xv):
        B = self._fm_forward(Xi, Xv)
        return self._engine.first[:B].clone()

    def second_order(self, Xi, Xv):
        B = self._fm_forward(Xi, Xv)
        return self._engine.bi[:B, :self.embedding_size].clone()

    def forward_fm(self, Xi, Xv):
        B = self._fm_forward(Xi, Xv)
        return self._engine.logit[:B].clone().reshape(self._logit_shape(B))

    def _logit_shape(self, B):
        return (B,)

    def _mlp(self, x):
        acts = []
        for layer in self.hidden_layers:
            x = F.relu(layer(x))
            acts.append(x)
        return acts

    def _base_logit(self, B):
        e = self._engine
        if self._fm_term_in_forward:
            return e.logit[:B]
        return e.sfirst[:B] + self.bias.reshape(-1)[0]

    def forward(self, Xi, Xv):
        B = self._fm_forward(Xi, Xv)
        e = self._engine
        with torch.no_grad():
            if not self._has_mlp:
                return e.logit[:B].clone()
            if e.mlp_fits(B, self.embedding_size, self.neuron_per_hidden_layer, self.num_hidden_layers, "forward"):
                out, layers = e.mlp_forward(self._mlp_flat, self.embedding_size, self.neuron_per_hidden_layer,
                                            self.num_hidden_layers, self._base_logit(B).contiguous(), B, self._onn)
                return (layers[-1], layers) if self._onn else out
            if getattr(self, "native_mlp", True):     # mini-batch sizes: the forward GEMM chain (fmx_mlp_forward_batch)
                out, layers = e.mlp_forward_batch(self._mlp_flat, self.embedding_size, self.neuron_per_hidden_layer,
                                                  self.num_hidden_layers, e.bi[:B], self._base_logit(B).contiguous(), B,
                                                  self._onn)
                return (layers[-1], layers) if self._onn else out
            acts = self._mlp(e.bi[:B, :self.embedding_size])
            base = self._base_logit(B)
            if not self._onn:
                return base + acts[-1].sum(1)
            layers = torch.stack([torch.sigmoid(base + a.sum(1)) for a in acts])
            return layers[-1], layers

    # ------------------------------------------------------------------------------------------------------
    # training steps
    # ------------------------------------------------------------------------------------------------------
    def _fm_step(self, Xi, Xv, Y, loss_kind):
        idx_d, xv_d, y_d = self._inputs(Xi, Xv, Y)
        if y_d.numel() != idx_d.shape[0]:
            raise ValueError(f"Target size ({y_d.numel()}) must be the same as input size ({idx_d.shape[0]})")
        self._engine.step(self._hyper, self.update_rule, loss_kind, idx_d, xv_d, y_d)
        out = self._engine.loss_out[0].clone()
        self._after_step()
        return out

    def update_embedding(self, Xi, Xv, Y):
        """One mini-batch step on forward_fm (reference fm_adam.py:56-69); returns the loss tensor."""
        self.train()
        return self._fm_step(Xi, Xv, Y, self._loss_update_embedding)

    def fit(self, Xi, Xv, Y):
        self._fit(Xi, Xv, Y)
        self._after_step()

    def _fit(self, Xi, Xv, Y):
        self.train()
        if not self._has_mlp:
            self._fm_step(Xi, Xv, Y, self._loss_fit)
            return
        if self._onn:
            self._fit_hedge(Xi, Xv, Y)
            return
        # DeepFM / NFM (reference deepfm_adam.py:106-119, nfm_adam.py:105-118): tables through the kernels,
        # the MLP through autograd, every parameter updated by the fresh-Adam rule
        idx_d, xv_d, y_d = self._inputs(Xi, Xv, Y)
        e, k = self._engine, self.embedding_size
        e.sort(idx_d)
        B = e.forward(self._hyper, idx_d, xv_d)
        if self.update_rule != "ftrl" and e.mlp_fits(B, k, self.neuron_per_hidden_layer, self.num_hidden_layers, "fit"):
            # one launch: MLP forward, loss, backward, fresh-Adam update of the hidden layers; then the table update
            dz, gbi = e.mlp_fit(self._mlp_flat, k, self.neuron_per_hidden_layer, self.num_hidden_layers, self._hyper,
                                self.update_rule, self._loss_fit, self._base_logit(B).contiguous(), y_d, B)
            e.update(self._hyper, self.update_rule, B, xv_d, dz, dz if self._fm_term_in_forward else None, gbi,
                     with_loss=False)
            return
        if self.update_rule != "ftrl" and getattr(self, "native_mlp", True):
            # mini-batch sizes: the MLP section as fp32 MFMA GEMMs (fmx_mlp_section); the hidden layers then take the
            # closed form of the reference's fresh-Adam first step, p -= lr * g / (|g| + 1e-8) (or plain SGD)
            if getattr(self, "_mlp_gflat", None) is None:
                self._mlp_gflat = torch.zeros_like(self._mlp_flat)
            _, dz, gbi = e.mlp_section(self._mlp_flat, self._mlp_gflat, k, self.neuron_per_hidden_layer,
                                       self.num_hidden_layers, self._loss_fit, e.bi[:B],
                                       self._base_logit(B).contiguous(), y_d, B, 1.0 / B)
            e.update(self._hyper, self.update_rule, B, xv_d, dz, dz if self._fm_term_in_forward else None, gbi,
                     with_loss=False)
            g, lr = self._mlp_gflat, float(self.n)
            with torch.no_grad():
                if self.update_rule == "sgd":
                    self._mlp_flat.sub_(g, alpha=lr)
                else:
                    self._mlp_flat.sub_(lr * g / (g.abs() + 1e-8))
            return
        bi = e.bi[:B, :k].detach().clone().requires_grad_(True)
        base = self._base_logit(B).detach().clone().requires_grad_(True)
        for p in self.hidden_layers.parameters():
            p.grad = None
        out = base + self._mlp(bi)[-1].sum(1)
        if self._loss_fit == "sigmoid":
            loss = F.binary_cross_entropy_with_logits(torch.sigmoid(out), y_d)
        else:
            loss = F.binary_cross_entropy_with_logits(out, y_d)
        loss.backward()
        dz = base.grad.contiguous()
        gbi = bi.grad
        if k != self._table.kp:
            gbi = F.pad(gbi, (0, self._table.kp - k))
        gbi = gbi.contiguous()
        e.update(self._hyper, self.update_rule, B, xv_d, dz, dz if self._fm_term_in_forward else None, gbi,
                 with_loss=False)
        # hidden layers: literally the reference's optimizer (a new Adam, first step)
        torch.optim.Adam(self.hidden_layers.parameters(), lr=float(self.n)).step()

    def _fit_hedge(self, Xi, Xv, Y):
        """Hedge backprop (reference deepfm_onn.py:109-154): hidden layers and alpha only.  Since
        d(loss_i)/d(layer j) = 0 for j > i, the reference's alpha-weighted accumulation over L backward passes is the
        gradient of sum_i alpha_i * loss_i, taken here in ONE backward pass over the MLP."""
        idx_d, xv_d, y_d = self._inputs(Xi, Xv, Y)
        B = idx_d.shape[0]
        if B != self.batch_size or y_d.numel() != self.batch_size:
            raise RuntimeError(f"shape '[{self.batch_size}]' is invalid for input of size {B}")
        e, k = self._engine, self.embedding_size
        e.forward(self._hyper, idx_d, xv_d, want_first=False)
        if e.mlp_fits(B, k, self.neuron_per_hidden_layer, self.num_hidden_layers, "hedge"):
            e.mlp_hedge_fit(self._mlp_flat, k, self.neuron_per_hidden_layer, self.num_hidden_layers, float(self.n),
                            float(self.b.detach()), float(self.s.detach()), self.alpha, self._base_logit(B).contiguous(), y_d, B)
            return
        if getattr(self, "native_mlp", True):    # mini-batch sizes: the same step on the MFMA GEMMs (fmx_mlp_hedge_section)
            if getattr(self, "_mlp_gflat", None) is None:
                self._mlp_gflat = torch.zeros_like(self._mlp_flat)
            e.mlp_hedge_section(self._mlp_flat, self._mlp_gflat, k, self.neuron_per_hidden_layer, self.num_hidden_layers,
                                float(self.n), float(self.b.detach()), float(self.s.detach()), self.alpha, e.bi[:B],
                                self._base_logit(B).contiguous(), y_d, B)
            return
        base = self._base_logit(B).detach()
        for p in self.hidden_layers.parameters():
            p.grad = None
        acts = self._mlp(e.bi[:B, :k].detach())
        losses = torch.stack([F.binary_cross_entropy(torch.sigmoid(base + a.sum(1)), y_d) for a in acts])
        alpha = self.alpha.detach()
        (alpha * losses).sum().backward()
        with torch.no_grad():
            n = float(self.n)
            for layer in self.hidden_layers:
                layer.weight -= n * layer.weight.grad
                layer.bias -= n * layer.bias.grad
            bdev = self.b.detach().to(self.device)
            a = alpha * torch.pow(bdev, losses.detach())
            a = torch.maximum(a, (self.s.detach() / self.num_hidden_layers).to(self.device))
            self.alpha = a / a.sum()

    # ------------------------------------------------------------------------------------------------------
    # inference / online protocol (reference fm_adam.py:84-119)
    # ------------------------------------------------------------------------------------------------------
    def predict(self, Xi, Xv):
        self.eval()
        out = self.forward(Xi, Xv)
        if isinstance(out, tuple):
            out = out[0]
        pred = torch.sigmoid(out).cpu()
        return pred.data.numpy() > 0.5

    def _device_loop_ok(self):
        """Can run_experiment's predict-then-fit loop run on the device for this model?"""
        if not getattr(self, "device_online_loop", True):
            return False
        e, k = self._engine, self.embedding_size
        if not self._has_mlp:
            return e.online_run_fits(self.field_size, self._table.kp)
        H, L = self.neuron_per_hidden_layer, self.num_hidden_layers
        if self._onn:
            return self.batch_size == 1 and e.mlp_fits(1, k, H, L, "hedge")
        return self.update_rule != "ftrl" and e.mlp_fits(1, k, H, L, "fit")

    def _run_experiment_on_device(self, data_Xi, data_Xv, data_Y):
        """The whole predict-then-fit loop on the device, same arithmetic as predict() + fit() per sample: pure FM as one
        wavefront walking the stream (fmx_fm_online_run); the MLP classes as the per-sample launches queued back to back
        without host synchronisation (fmx_online_run_mlp).  The confusion matrix and its checkpoints are then counted on
        the host from the per-sample predictions, in the reference's order."""
        start = time()
        idx_d, xv_d, y_d = self._inputs(data_Xi, data_Xv, data_Y)
        self.train()
        e = self._engine
        if not self._has_mlp:
            pred, _ = e.online_run(self._hyper, self.update_rule, self._loss_fit, idx_d, xv_d, y_d)
            pred = pred.cpu().numpy().astype(bool)
        else:
            out = e.online_run_mlp(self._hyper, self.update_rule, self._loss_fit, self._mlp_flat, self.embedding_size,
                                   self.neuron_per_hidden_layer, self.num_hidden_layers, self._onn, self._fm_term_in_forward,
                                   float(self.b.detach()) if self._onn else 0.0, float(self.s.detach()) if self._onn else 0.0,
                                   self.alpha if self._onn else None, idx_d, xv_d, y_d)
            pred = (torch.sigmoid(out) > 0.5).cpu().numpy()          # predict(): sigmoid of what forward() returns
        e.check_error_flag()
        y = np.asarray(data_Y).reshape(-1)
        accuracy, roc = [], []
        n = len(y)
        pos, hit = y == 1, pred == (y == 1)      # `pred == data_Y[i]` in the reference: a bool against a 0/1 label
        tp, fn = np.cumsum(pos & hit), np.cumsum(pos & ~hit)
        tn, fp = np.cumsum(~pos & hit), np.cumsum(~pos & ~hit)
        for i in sorted(set(range(0, n, 1000)) | {n - 1}):
            roc.append({"tpr": tp[i] / (tp[i] + fn[i] + 1e-16), "fpr": fp[i] / (fp[i] + tn[i] + 1e-16)})
            accuracy.append((tp[i] + tn[i]) / (i + 1) * 100)
        cm = {"tp": int(tp[-1]), "fp": int(fp[-1]), "tn": int(tn[-1]), "fn": int(fn[-1])}
        return time() - start, float(accuracy[-1]), {k: float(v) for k, v in roc[-1].items()}, cm

    def run_experiment(self, data_Xi, data_Xv, data_Y):
        data_size = len(data_Y)
        if data_size > 0 and self._device_loop_ok():
            return self._run_experiment_on_device(data_Xi, data_Xv, data_Y)
        confusion_matrix = {"tp": 0, "fp": 0, "tn": 0, "fn": 0}
        accuracy, roc = [], []
        start = time()
        for i in range(data_size):
            pred = self.predict(data_Xi[i], data_Xv[i])
            self.fit([data_Xi[i]], [data_Xv[i]], [data_Y[i]])
            hit = bool(pred == data_Y[i])
            if data_Y[i] == 1:
                confusion_matrix["tp" if hit else "fn"] += 1
            else:
                confusion_matrix["tn" if hit else "fp"] += 1
            if i % 1000 == 0 or i == data_size - 1:
                tpr = confusion_matrix["tp"] / (confusion_matrix["tp"] + confusion_matrix["fn"] + 1e-16)
                fpr = confusion_matrix["fp"] / (confusion_matrix["fp"] + confusion_matrix["tn"] + 1e-16)
                roc.append({"tpr": tpr, "fpr": fpr})
                accuracy.append((confusion_matrix["tp"] + confusion_matrix["tn"]) / (i + 1) * 100)
        time_elapsed = time() - start
        return time_elapsed, accuracy[-1], roc[-1], confusion_matrix

    def __str__(self):
        s = f"{self._name}-Feature_Sizes{self.feature_sizes}-Embedding_Sizes{self.embedding_size}-"
        if self._has_mlp:
            s += f"Num_Hidden_Layers{self.num_hidden_layers}-Neuron_Per_Hidden_Layer{self.neuron_per_hidden_layer}-"
        s += f"Num_Classes{self.num_classes}"
        if self._onn:
            s += f"-N{self.n}"
        return s
