"""TEST-ONLY subclass of dib_amd.SetTransformerDIB whose device steps - forward / loss_and_backward / _loss_only / adam_step -
run on the float64 CPU oracle (oracle/set_transformer_oracle.py), so the product's HOST logic - train_step's data-parallel
protocol (neighbourhood sharding, global-token noise keys, inv_global_batch, gradient + statistics all-reduce), fit's
schedules - runs without a GPU.  Lives in tests/; the product package never imports it and has no hook for it: the
subclass overrides the five methods the product class itself defines."""
import numpy as np
import torch

import dib_oracle as orc
import set_transformer_oracle as sto


class OracleSetTransformerBackend:
    """the five device steps on the oracle; `m` is the model object whose flat buffers they read and write"""

    def __init__(self, spec: sto.SetTransformerSpec):
        self.spec = spec

    def _params(self, m, requires_grad=False):
        flat = m.params
        out = {}
        for name, shp in m.shapes.items():
            n = int(np.prod(shp))
            t = flat[m.offsets[name]: m.offsets[name] + n].view(*shp).clone()
            out[name] = t.requires_grad_(requires_grad)
        return out

    def forward(self, m, batch_inp, step, deterministic, row0, embs_reparam):
        x = torch.as_tensor(np.asarray(batch_inp.cpu() if isinstance(batch_inp, torch.Tensor) else batch_inp), dtype=torch.float64)
        B, P, _ = x.shape
        step = m._step if step is None else int(step)
        E = self.spec.bottleneck_dimension
        rows = (int(row0) + np.arange(B * P)).astype(np.uint32)                  # GLOBAL token index keys the noise
        eps = orc.philox_normal_all(m.noise_seed, step, rows, 1, E)[:, 0, :].reshape(B, P, E)
        if deterministic:
            eps = np.zeros_like(eps)
        p = self._params(m, requires_grad=True)
        out = sto.forward(self.spec, p, x, eps)
        m.last = dict(plan=None, step=step, row0=int(row0), B=B, P=P, kl=out["kl"].detach().reshape(1), _p=p, _out=out)
        return out["pred"].detach()

    def _rows(self, m, is_loci):
        out = m.last["_out"]
        y = torch.as_tensor(np.asarray(is_loci.cpu() if isinstance(is_loci, torch.Tensor) else is_loci),
                            dtype=torch.float64).reshape(out["pred"].shape)
        z = out["pred"]
        bce_rows = torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs()))
        kl_rows = (0.5 * (out["mu"] ** 2 + torch.exp(out["logvar"]) - out["logvar"] - 1.0)).sum(dim=(-1, -2))
        return bce_rows, kl_rows, y

    def loss_and_backward(self, m, is_loci, inv_global_batch):
        inv = 1.0 / m.last["B"] if inv_global_batch is None else float(inv_global_batch)
        bce_rows, kl_rows, y = self._rows(m, is_loci)
        loss = (bce_rows.sum() + float(m.beta_dev.item()) * kl_rows.sum()) * inv
        p = m.last["_p"]
        grads = torch.autograd.grad(loss, list(p.values()))
        m.grads.zero_()
        for (name, shp), g in zip(m.shapes.items(), grads):
            n = int(np.prod(shp))
            m.grads[m.offsets[name]: m.offsets[name] + n] = g.reshape(-1)
        m.last["bce"] = (bce_rows.sum() * inv).detach().reshape(1)
        m.last["correct"] = ((m.last["_out"]["pred"] > 0).double() == y).sum().reshape(1).double()

    def loss_only(self, m, is_loci, inv_global_batch):
        inv = 1.0 / m.last["B"] if inv_global_batch is None else float(inv_global_batch)
        bce_rows, _, y = self._rows(m, is_loci)
        m.last["bce"] = (bce_rows.sum() * inv).detach().reshape(1)
        m.last["correct"] = ((m.last["_out"]["pred"] > 0).double() == y).sum().reshape(1).double()

    def adam_step(self, m, beta_1, beta_2, epsilon):
        """Keras Adam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps)."""
        m.t_dev += 1
        t = int(m.t_dev.item())
        lr_t = float(m.lr_dev.item()) * np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
        m.adam_m += (1.0 - beta_1) * (m.grads - m.adam_m)
        m.adam_v += (1.0 - beta_2) * (m.grads ** 2 - m.adam_v)
        m.params -= lr_t * m.adam_m / (torch.sqrt(m.adam_v) + epsilon)


def make_model(spec: sto.SetTransformerSpec, **kw):
    import dib_amd
    backend = OracleSetTransformerBackend(spec)

    class OracleSetTransformerDIB(dib_amd.SetTransformerDIB):
        state_dtype = torch.float64   # the CPU checker keeps its state in float64

        def _acquire_device(self, device):
            self.lib, self.device = None, torch.device("cpu")

        def forward(self, batch_inp, step=None, deterministic=False, row0=0, embs_reparam=None, _step_from_device=False,
                    for_backward=True, _skip_head=False):
            return backend.forward(self, batch_inp, step, deterministic, row0, embs_reparam)

        def loss_and_backward(self, is_loci, inv_global_batch=None, reduce=True):
            return backend.loss_and_backward(self, is_loci, inv_global_batch)

        def _loss_only(self, is_loci, inv_global_batch=None):
            return backend.loss_only(self, is_loci, inv_global_batch)

        def adam_step(self, beta_1=0.9, beta_2=0.999, epsilon=1e-7, fused_reduce=False):
            return backend.adam_step(self, beta_1, beta_2, epsilon)

    kw.setdefault("use_graphs", False)
    return OracleSetTransformerDIB(spec.particle_feature_dimensions, spec.number_positional_encoding_frequencies,
                                   spec.particle_encoder_arch_spec, spec.bottleneck_dimension, spec.key_dim,
                                   spec.number_heads_per_mha, spec.number_attention_blocks, spec.ff_arch_per_block,
                                   spec.final_processing_arch, spec.output_dimensionality, spec.logvar_initialization,
                                   spec.layer_norm_epsilon, **kw)
