"""BigGAN-deep generator (reference model/biggan_generator.py: BigGANBatchNorm :100-150, GenBlock
:153-203, SelfAttn :58-97, Generator :206-256, BigGAN :259-304; config model/utils/biggan_config.py)
on the HIP kernels.  Same class names, forward signatures and state_dict keys (602 keys for
deep-256: `embeddings.weight`, `generator.gen_z.{bias,weight_orig,weight_u,weight_v}`,
`generator.layers.{n}.bn_{k}.{running_means,running_vars,scale.weight_*,offset.weight_*}`, ...).

Conditional batch norm + ReLU are fused into the prologue of the following convolution, the
nearest upsample into its read, bias and the residual add into its epilogue; self-attention is
a dedicated kernel.  Spectral normalisation keeps torch.nn.utils.spectral_norm's semantics
(`weight_orig / sigma`, one power iteration per forward in train mode - SURVEY Q2) as a few
small matvecs on the parameters; the effective weight is what gets packed for the conv kernel.
Only 3 of the 128 output channels of `conv_to_rgb` are ever used (:251-253): 16 are
computed.
"""
import copy
import json
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .stylegan2_generator import _dt


class BigGANConfig(object):
    """model/utils/biggan_config.py:11-71"""

    def __init__(self, output_dim=128, z_dim=128, class_embed_dim=128, channel_width=128, num_classes=1000,
                 layers=((False, 16, 16), (True, 16, 16), (False, 16, 16), (True, 16, 8), (False, 8, 8), (True, 8, 4),
                         (False, 4, 4), (True, 4, 2), (False, 2, 2), (True, 2, 1)),
                 attention_layer_position=8, eps=1e-4, n_stats=51):
        self.output_dim, self.z_dim, self.class_embed_dim, self.channel_width = output_dim, z_dim, class_embed_dim, channel_width
        self.num_classes, self.layers, self.attention_layer_position, self.eps, self.n_stats = \
            num_classes, [tuple(l) for l in layers], attention_layer_position, eps, n_stats

    @classmethod
    def from_dict(cls, json_object):
        config = BigGANConfig()
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)


class _SN(nn.Module):
    """Parameter holder with the state_dict layout of torch.nn.utils.spectral_norm(module)."""

    def __init__(self, weight_shape, bias, eps):
        super().__init__()
        if bias:
            self.bias = nn.Parameter(torch.zeros(weight_shape[0]))
        else:
            self.bias = None
        self.weight_orig = nn.Parameter(torch.randn(*weight_shape) * 0.05)
        out = weight_shape[0]
        self.register_buffer("weight_u", F.normalize(torch.randn(out), dim=0, eps=eps))
        self.register_buffer("weight_v", F.normalize(torch.randn(int(torch.tensor(weight_shape[1:]).prod())), dim=0, eps=eps))
        self.eps = eps

    @torch.no_grad()
    def effective_weight(self, training, ctx=None):
        """`ctx` (a dict) receives sigma and copies of u, v: what the gradient w.r.t. weight_orig needs (sn_weight_grad)."""
        prep = self.__dict__.get("_prep")
        if prep is not None:                       # computed by sn_prepare() for the whole module in five grouped launches
            self.__dict__["_prep"] = None
            w_eff, sigma, us, vs = prep
            if ctx is not None:
                ctx.update(sigma=sigma, u=us, v=vs)
            return w_eff
        w = self.weight_orig.detach()
        wm = w.reshape(w.shape[0], -1)
        u, v = self.weight_u, self.weight_v
        if training:                               # one power iteration, buffers updated in place
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=self.eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=self.eps))
        sigma = torch.dot(u, torch.mv(wm, v))
        if ctx is not None:
            ctx.update(sigma=sigma, u=u.clone(), v=v.clone())
        return w / sigma


def sn_prepare(module, training):
    """Spectral normalisation of EVERY _SN weight under `module` for the coming forward in five grouped launches
    (dge_sn_group): in train mode one power iteration per weight, written into the weight_u / weight_v buffers exactly as
    the per-weight path does.  Each _SN then hands out its W_eff (and sigma, u, v for the backward) once."""
    import numpy as np
    sns = module.__dict__.get("_sn_list")
    if sns is None:
        skip = module.__dict__.get("_sn_skip", ())          # sub-modules the forward never reaches keep their u / v untouched
        sns = [m for n, m in module.named_modules() if isinstance(m, _SN) and not any(n.startswith(p) for p in skip)]
        # conditional-BN linears first (their W_eff rows then form ONE [sum C, cond_dim] matrix: sn_cbn_linears), then the rest
        bnl = {id(l) for bn in module.modules() if isinstance(bn, BigGANBatchNorm) and bn.conditional for l in (bn.scale, bn.offset)}
        sns = [m for m in sns if id(m) in bnl] + [m for m in sns if id(m) not in bnl]
        module.__dict__["_sn_list"] = sns
        module.__dict__["_sn_nbn"] = sum(1 for m in sns if id(m) in bnl)
    if not sns:
        return
    dev = sns[0].weight_orig.device
    eps = sns[0].eps
    assert all(m.eps == eps for m in sns)
    key = tuple((m.weight_orig.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr()) for m in sns)
    st = module.__dict__.get("_sn_static")
    if st is None or st["key"] != key:
        n = len(sns)
        Os = np.array([m.weight_orig.shape[0] for m in sns], dtype=np.int64)
        Ks = np.array([m.weight_orig[0].numel() for m in sns], dtype=np.int64)
        dt = np.dtype([("W", "u8"), ("u", "u8"), ("v", "u8"), ("t", "u8"), ("s", "u8"), ("weff", "u8"), ("usnap", "u8"), ("vsnap", "u8"),
                       ("O", "i4"), ("K", "i4")])
        assert dt.itemsize == ops.lib().dge_sn_entry_size()
        tab = np.zeros(n, dtype=dt)
        tab["W"] = [m.weight_orig.data_ptr() for m in sns]
        tab["u"] = [m.weight_u.data_ptr() for m in sns]
        tab["v"] = [m.weight_v.data_ptr() for m in sns]
        tab["O"], tab["K"] = Os, Ks
        cum = lambda a: np.concatenate([[0], np.cumsum(a)[:-1]])
        st = dict(key=key, tab=tab, Os=Os, Ks=Ks, offO=cum(Os), offK=cum(Ks), offW=cum(Os * Ks), totO=int(Os.sum()), totK=int(Ks.sum()),
                  totW=int((Os * Ks).sum()), maxO=int(Os.max()), maxK=int(Ks.max()))
        module.__dict__["_sn_static"] = st
    n = len(sns)
    f32 = dict(dtype=torch.float32, device=dev)
    t_all = torch.zeros(st["totK"], **f32) if training else torch.empty(1, **f32)
    s_all, weff = torch.empty(st["totO"], **f32), torch.empty(st["totW"], **f32)
    usnap, vsnap, sigma = torch.empty(st["totO"], **f32), torch.empty(st["totK"], **f32), torch.empty(n, **f32)
    tab = st["tab"].copy()
    tab["t"] = t_all.data_ptr() + 4 * (st["offK"] if training else 0)
    tab["s"] = s_all.data_ptr() + 4 * st["offO"]
    tab["weff"] = weff.data_ptr() + 4 * st["offW"]
    tab["usnap"] = usnap.data_ptr() + 4 * st["offO"]
    tab["vsnap"] = vsnap.data_ptr() + 4 * st["offK"]
    # (the table repeats from step to step once the allocator has settled: uploaded only when it changes, through a pinned buffer -
    #  a pageable host-to-device copy waits for the stream to drain, once per generator / encoder forward here)
    raw = tab.view(np.uint8)
    hit = module.__dict__.get("_sn_tab_dev")
    if hit is not None and hit[0].shape == raw.shape and np.array_equal(hit[0], raw):
        tab_dev = hit[1]
    else:
        pin = torch.from_numpy(raw.copy()).pin_memory()
        tab_dev = pin.to(dev, non_blocking=True)
        module.__dict__["_sn_tab_dev"] = (raw.copy(), tab_dev, pin)
    ops.check(ops.lib().dge_sn_group(ops._p(tab_dev), n, st["maxO"], st["maxK"], ops._p(sigma), float(eps), 1 if training else 0,
                                     ops._stream()), "dge_sn_group")
    for i, m in enumerate(sns):
        O, K, oo, ok, ow = int(st["Os"][i]), int(st["Ks"][i]), int(st["offO"][i]), int(st["offK"][i]), int(st["offW"][i])
        m.__dict__["_prep"] = (weff[ow:ow + O * K].view(m.weight_orig.shape), sigma[i], usnap[oo:oo + O], vsnap[ok:ok + K])
    module.__dict__["_sn_weff"] = weff


def sn_cbn_linears(module, cond):
    """scale / offset of EVERY conditional batch norm under `module` (:141-144: two snlinears of the condition vector each) as
    one dense launch over the concatenated W_eff rows; each BigGANBatchNorm then slices its columns.  Call after sn_prepare."""
    st, sns, nbn = module.__dict__.get("_sn_static"), module.__dict__.get("_sn_list"), module.__dict__.get("_sn_nbn", 0)
    if not nbn:
        return
    K = int(st["Ks"][0])
    if any(int(k) != K for k in st["Ks"][:nbn]) or cond.shape[1] != K:
        return
    rows = int(st["offO"][nbn - 1] + st["Os"][nbn - 1])
    out = ops.linear(cond, module.__dict__["_sn_weff"][:rows * K].view(rows, K))          # [B, sum C]
    for i in range(nbn):
        o0 = int(st["offO"][i])
        sns[i].__dict__["_lin"] = out[:, o0:o0 + int(st["Os"][i])]


def sn_weight_grad(g_w_eff, weight_orig, sn_ctx):
    """Gradient w.r.t. `weight_orig` of W_eff = W / sigma, sigma = u^T W v with u, v held constant (the semantics of
    torch.nn.utils.spectral_norm's autograd): (g - <g, W> / sigma * u v^T) / sigma.  sigma, u, v are the forward's;
    `weight_orig` is the LIVE parameter: the division's backward reads the tensor autograd saved, which LREQAdam has
    meanwhile updated through .data when the second backward of a step runs.  Parameter-sized glue."""
    gm = g_w_eff.reshape(g_w_eff.shape[0], -1)
    dot = (gm * weight_orig.detach().reshape(gm.shape)).sum() / sn_ctx["sigma"]
    return ((gm - dot * torch.outer(sn_ctx["u"], sn_ctx["v"])) / sn_ctx["sigma"]).reshape(g_w_eff.shape)


class BigGANBatchNorm(nn.Module):
    def __init__(self, num_features, condition_vector_dim=None, n_stats=51, eps=1e-4, conditional=True):
        super().__init__()
        self.num_features, self.eps, self.conditional = num_features, eps, conditional
        self.register_buffer("running_means", torch.zeros(n_stats, num_features))
        self.register_buffer("running_vars", torch.ones(n_stats, num_features))
        self.step_size = 1.0 / (n_stats - 1)
        if conditional:
            self.scale = _SN((num_features, condition_vector_dim), False, eps)
            self.offset = _SN((num_features, condition_vector_dim), False, eps)
        else:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))

    def affine(self, truncation, cond, training, ctx=None):
        """-> per-(b,c) affine (a, b).  `ctx` (a dict) receives what the backward w.r.t. the condition vector needs."""
        # `truncation` may be the float32 tensor the training script builds (E_align_s2.py:148): tensor / float is then a
        # float32 division (0.4f / 0.02 == 20.0 exactly, whereas float(0.4f) / 0.02 == 20.0000003 would select the next row)
        coef, start_idx = math.modf(float(truncation / self.step_size))
        start_idx = int(start_idx)
        # the interpolated statistics (and 1 / sqrt(var + eps) for the backward) depend on the truncation and on the buffers only:
        # computed once per (truncation row, buffer version) instead of 2 - 8 tensor launches per norm and pass (127 norms per step)
        skey = (start_idx, coef, self.running_means._version, self.running_vars._version, self.running_means.data_ptr(), self.running_vars.data_ptr())
        hit = self.__dict__.get("_stat_cache")
        if hit is None or hit[0] != skey:
            if coef != 0.0:
                mean = self.running_means[start_idx] * coef + self.running_means[start_idx + 1] * (1 - coef)
                var = self.running_vars[start_idx] * coef + self.running_vars[start_idx + 1] * (1 - coef)
            else:
                mean, var = self.running_means[start_idx], self.running_vars[start_idx]
            hit = (skey, mean, var, torch.rsqrt(var + self.eps))
            self.__dict__["_stat_cache"] = hit
        _, mean, var, rstd_c = hit
        if self.conditional:
            sn_sc, sn_of = ({}, {}) if ctx is not None else (None, None)
            wsc, wof = self.scale.effective_weight(training, sn_sc).contiguous(), self.offset.effective_weight(training, sn_of).contiguous()
            sc, of = self.scale.__dict__.pop("_lin", None), self.offset.__dict__.pop("_lin", None)     # sn_cbn_linears
            if sc is None or of is None:
                sc, of = ops.linear(cond, wsc), ops.linear(cond, wof)
            if ctx is not None:
                ctx.update(wsc=wsc, wof=wof, mean=mean, rstd=rstd_c, sn_sc=sn_sc, sn_of=sn_of)
        else:
            sc = (self.weight.detach() - 1.0).reshape(1, -1).contiguous()
            of = self.bias.detach().reshape(1, -1).contiguous()
        return ops.cbn_affine(sc, of, mean, var, self.eps)


def _cbn_cond_grad(ctx, st, g_cond):
    """st [B,C,2] = (dL/da, dL/db) of the affine a = (1+scale)*rstd, b = offset - mean*a; accumulates dL/dcond into g_cond
    through the two spectral-norm linears scale = cond @ Wsc^T, offset = cond @ Wof^T (:141-144)."""
    g_a, g_b = st[:, :, 0], st[:, :, 1]
    g_scale = ((g_a - g_b * ctx["mean"]) * ctx["rstd"]).contiguous()          # [B,C] glue on the per-(b,c) sums
    ops.linear_t(g_scale, ctx["wsc"], g_cond, accumulate=True)
    ops.linear_t(g_b.contiguous(), ctx["wof"], g_cond, accumulate=True)


import os as _os
_NO_PACK_GROUP = bool(_os.environ.get("DGE_NO_PACK_GROUP"))


def _pack_all(module, sns, dt, dgrad):
    """Packed copies of the W_eff of every conv `_SN` in `sns` for the coming pass in ONE launch per layout (forward; data gradient
    when the pass is saved for a backward) instead of one dge_pack_conv_weight launch per conv and pass (round 5 profile of --mtype 4:
    214 pack launches per step).  The W_eff are the ones sn_prepare() parked on the modules; the packed tensors are allocated on first
    use and refreshed in place afterwards (same stream: a pass reads them before the next pass rewrites them).  Returns
    {id(sn): {mode: packed}}; modules whose W_eff is not parked (per-weight path) are left to pack_conv_weight.

    The data-gradient copies are SHARED buffers, not snapshots: a second saved pass rewrites them (the power iteration moves W_eff on
    every train-mode pass).  Every saved pass therefore takes a stamp (`out["_stamp"]`, kept in its records next to `wpd`): a backward
    uses `wpd` only while its stamp is still the live one (`_wpd_live`) and otherwise packs the record's own `w` again."""
    out, entries = {}, {ops.PACK_FWD: [], ops.PACK_DGRAD: []}
    if _NO_PACK_GROUP:
        return out
    cache = module.__dict__.setdefault("_pk_cache", {})
    if dgrad:
        cell = module.__dict__.setdefault("_pk_stamp", [0])
        cell[0] += 1
        out["_stamp"] = (cell, cell[0])
    for sn in sns:
        prep = sn.__dict__.get("_prep")
        if prep is None or prep[0].dim() != 4:
            continue
        w = prep[0]
        if not w.is_contiguous():
            continue
        for mode in ((ops.PACK_FWD, ops.PACK_DGRAD) if dgrad else (ops.PACK_FWD,)):
            key = (id(sn), mode, dt, w.device)
            pk = cache.get(key)
            if pk is None:
                pk = cache[key] = ops.pack_conv_weight(w, mode, dt, 1.0)        # first pass: allocates (and packs)
            else:
                entries[mode].append((w, mode, dt, 1.0, pk))
            out.setdefault(id(sn), {})[mode] = pk
    for mode, ent in entries.items():
        if ent:
            module.__dict__["_pk_scratch_%d" % mode] = ops.pack_conv_weights_multi(ent, module.__dict__.get("_pk_scratch_%d" % mode))
    return out


def _wpd_live(wpd, stamp):
    """The shared data-gradient copy `wpd` of a saved pass, or None when a later saved pass has rewritten it (see _pack_all)."""
    if wpd is None or stamp is None or stamp[0][0] != stamp[1]:
        return None
    return wpd


class GenBlock(nn.Module):
    def __init__(self, in_size, out_size, condition_vector_dim, reduction_factor=4, up_sample=False, n_stats=51, eps=1e-12):
        super().__init__()
        self.up_sample, self.drop_channels = up_sample, in_size != out_size
        self.in_size, self.out_size = in_size, out_size
        mid = in_size // reduction_factor
        self.mid = mid
        self.bn_0 = BigGANBatchNorm(in_size, condition_vector_dim, n_stats, eps, True)
        self.conv_0 = _SN((mid, in_size, 1, 1), True, eps)
        self.bn_1 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_1 = _SN((mid, mid, 3, 3), True, eps)
        self.bn_2 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_2 = _SN((mid, mid, 3, 3), True, eps)
        self.bn_3 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_3 = _SN((out_size, mid, 1, 1), True, eps)

    def run(self, x, cond, truncation, dt, training, saved=None, packed=None):
        recs = []

        def conv(sn, bn, inp, k, cout, **kw):
            ctx = {} if saved is not None else None
            a, b = bn.affine(truncation, cond, training, ctx)
            w = sn.effective_weight(training).contiguous()
            pk = (packed or {}).get(id(sn), {})
            wp = pk.get(ops.PACK_FWD)
            if wp is None:
                wp = ops.pack_conv_weight(w, ops.PACK_FWD, dt, 1.0)
            if saved is not None:
                recs.append(dict(inp=inp, a=a, b=b, w=w, k=k, ctx=ctx, wpd=pk.get(ops.PACK_DGRAD), stamp=(packed or {}).get("_stamp")))
            return ops.conv2d(inp, wp, cout, k, in_scale=a, in_shift=b, in_relu=True, bias=sn.bias.detach(), **kw)
        t = conv(self.conv_0, self.bn_0, x, 1, self.mid)
        t = conv(self.conv_1, self.bn_1, t, 3, self.mid, in_up2=self.up_sample)
        t = conv(self.conv_2, self.bn_2, t, 3, self.mid)
        skip = ops.slice_up(x, self.out_size, self.up_sample) if (self.drop_channels or self.up_sample) else x
        y = conv(self.conv_3, self.bn_3, t, 1, self.out_size, addend=skip, add_scale=1.0)
        if saved is not None:
            saved.append(("block", self, recs))
        return y

    def backward(self, recs, g_out, g_cond, dt):
        """Data gradient of the block: returns dL/dx, accumulates dL/dcond (through the four conditional batch norms)."""
        g = g_out
        for idx in (3, 2, 1, 0):
            r = recs[idx]
            cin = r["inp"].shape[3]
            wpd = _wpd_live(r.get("wpd"), r.get("stamp"))
            gu = ops.conv2d(g, wpd if wpd is not None else ops.pack_conv_weight(r["w"], ops.PACK_DGRAD, dt, 1.0), cin, r["k"])
            if idx == 1 and self.up_sample:
                gu, _ = ops.nearest_up2_bwd(gu)
            g, st = ops.affine_relu_bwd(gu, r["inp"], r["a"], r["b"])
            _cbn_cond_grad(r["ctx"], st, g_cond)
        return ops.slice_up_bwd(g_out, g, self.up_sample)          # skip path: channel drop + nearest upsample adjoint, added


class SelfAttn(nn.Module):
    def __init__(self, in_channels, eps=1e-12):
        super().__init__()
        self.in_channels = in_channels
        self.snconv1x1_theta = _SN((in_channels // 8, in_channels, 1, 1), False, eps)
        self.snconv1x1_phi = _SN((in_channels // 8, in_channels, 1, 1), False, eps)
        self.snconv1x1_g = _SN((in_channels // 2, in_channels, 1, 1), False, eps)
        self.snconv1x1_o_conv = _SN((in_channels, in_channels // 2, 1, 1), False, eps)
        self.gamma = nn.Parameter(torch.zeros(1))

    def run(self, x, dt, training, saved=None, packed=None):
        B, H, W, Cc = x.shape
        sns = (self.snconv1x1_theta, self.snconv1x1_phi, self.snconv1x1_g, self.snconv1x1_o_conv)
        ws = [sn.effective_weight(training).contiguous() for sn in sns]
        pks = [(packed or {}).get(id(sn), {}) for sn in sns]
        fwd = {id(w): pk.get(ops.PACK_FWD) for w, pk in zip(ws, pks)}
        pk = lambda w: fwd[id(w)] if fwd.get(id(w)) is not None else ops.pack_conv_weight(w, ops.PACK_FWD, dt, 1.0)
        theta = ops.conv2d(x, pk(ws[0]), Cc // 8, 1)
        phi_pre = ops.conv2d(x, pk(ws[1]), Cc // 8, 1)
        phi = ops.maxpool2(phi_pre)
        g_pre = ops.conv2d(x, pk(ws[2]), Cc // 2, 1)
        g = ops.maxpool2(g_pre)
        # softmax(theta phi^T) g as per-sample GEMMs on the MFMA conv kernel: the sample's keys / values are the "weights"
        # of a 1x1 convolution over its queries (N x M scores, row softmax in place, then P V)
        M, D, DV = H * W // 4, Cc // 8, Cc // 2
        pkm = lambda m: ops.pack_conv_weight(m.float().contiguous().view(m.shape[0], m.shape[1], 1, 1), ops.PACK_FWD, dt, 1.0)
        o = torch.empty((B, H, W, DV), dtype=x.dtype, device=x.device)
        probs = []
        import os
        if os.environ.get("DGE_ATTN_KERNEL") == "1":       # one-wavefront-per-query kernel (kept for cross-checks)
            o = ops.attention(theta.view(B, H * W, D), phi.view(B, M, D), g.view(B, M, DV)).view(B, H, W, DV)
            probs = None
        for b in range(B if probs is not None else 0):
            P = ops.softmax_rows_(ops.conv2d(theta[b:b + 1], pkm(phi[b].reshape(M, D)), M, 1))          # [1,H,W,M]
            ops.conv2d(P, pkm(g[b].reshape(M, DV).t()), DV, 1, out=o[b:b + 1])
            probs.append(P)
        gam = self.gamma.detach().reshape(1, 1).expand(B, Cc).contiguous()
        if saved is not None:
            saved.append(("attn", self, dict(theta=theta, phi_pre=phi_pre, phi=phi, g_pre=g_pre, g=g, ws=ws, probs=probs,
                                             wpd=[pk_.get(ops.PACK_DGRAD) for pk_ in pks], stamp=(packed or {}).get("_stamp"))))
        return ops.conv2d(o, pk(ws[3]), Cc, 1, out_scale=gam, addend=x, add_scale=1.0)

    def backward(self, rec, g_out, dt):
        """out = x + gamma * conv_o(softmax(theta phi^T) g)  (:75-97).  The softmax-attention backward runs as per-sample
        GEMMs on the MFMA conv kernels (1x1 convs / weight-gradient kernels with the sample's own K, V as the "weights");
        the forward's probability matrices are re-used."""
        theta, phi, gv = rec["theta"], rec["phi"], rec["g"]
        B, H, W, D = theta.shape
        M, DV, Cc = phi.shape[1] * phi.shape[2], gv.shape[3], 8 * D
        wth, wph, wg, wo = rec["ws"]
        dg = {id(w): _wpd_live(p_, rec.get("stamp")) for w, p_ in zip(rec["ws"], rec.get("wpd") or [None] * 4)}
        pkd = lambda w: dg[id(w)] if dg.get(id(w)) is not None else ops.pack_conv_weight(w, ops.PACK_DGRAD, dt, 1.0)
        pkf = lambda m: ops.pack_conv_weight(m.float().contiguous().view(m.shape[0], m.shape[1], 1, 1), ops.PACK_FWD, dt, 1.0)
        gam = self.gamma.detach().reshape(1, 1).expand(B, DV).contiguous()
        g_o = ops.conv2d(g_out, pkd(wo), DV, 1, out_scale=gam)                     # [B,H,W,DV]
        g_q = torch.empty_like(theta)
        g_k = torch.empty((B, M, D), dtype=torch.float32, device=theta.device)
        g_v = torch.empty((B, M, DV), dtype=torch.float32, device=theta.device)
        for b in range(B):
            Qb, Kb, Vb, gOb = theta[b:b + 1], phi[b].reshape(M, D), gv[b].reshape(M, DV), g_o[b:b + 1]
            P = rec["probs"][b] if rec["probs"] is not None else ops.softmax_rows_(ops.conv2d(Qb, pkf(Kb), M, 1))   # [1,H,W,M]
            gvb = ops.zeros((M, DV, 1, 1), theta.device)
            ops.conv_wgrad(P, gOb, gvb)                                           # gV = P^T gO
            gS = ops.softmax_rows_bwd_(P, ops.conv2d(gOb, pkf(Vb), M, 1))         # gP = gO V^T -> gS
            ops.conv2d(gS, pkf(Kb.t()), D, 1, out=g_q[b:b + 1])                   # gQ = gS K
            gkb = ops.zeros((M, D, 1, 1), theta.device)
            ops.conv_wgrad(gS, Qb, gkb)                                           # gK = gS^T Q
            g_k[b].copy_(gkb.view(M, D)); g_v[b].copy_(gvb.view(M, DV))
        g_phi_pre = ops.maxpool2_bwd(g_k.view(B, H // 2, W // 2, D).to(theta.dtype), rec["phi_pre"])
        g_g_pre = ops.maxpool2_bwd(g_v.view(B, H // 2, W // 2, DV).to(theta.dtype), rec["g_pre"])
        gx = ops.conv2d(g_q, pkd(wth), Cc, 1, addend=g_out, add_scale=1.0)
        gx = ops.conv2d(g_phi_pre, pkd(wph), Cc, 1, addend=gx, add_scale=1.0)
        return ops.conv2d(g_g_pre, pkd(wg), Cc, 1, addend=gx, add_scale=1.0)


class Generator(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        ch = config.channel_width
        cdim = config.z_dim * 2
        self.gen_z = _SN((4 * 4 * 16 * ch, cdim), True, config.eps)
        layers = []
        for i, layer in enumerate(config.layers):
            if i == config.attention_layer_position:
                layers.append(SelfAttn(ch * layer[1], eps=config.eps))
            layers.append(GenBlock(ch * layer[1], ch * layer[2], cdim, up_sample=layer[0], n_stats=config.n_stats, eps=config.eps))
        self.layers = nn.ModuleList(layers)
        self.bn = BigGANBatchNorm(ch, n_stats=config.n_stats, eps=config.eps, conditional=False)
        self.conv_to_rgb = _SN((ch, ch, 3, 3), True, config.eps)

    def forward(self, cond_vector, truncation, compute_dtype="bf16", saved=None):
        dt = _dt(compute_dtype)
        training = self.training
        B = cond_vector.shape[0]
        ch = self.config.channel_width
        sn_prepare(self, training)
        sn_cbn_linears(self, cond_vector)
        convs = self.__dict__.get("_conv_sns")
        if convs is None:
            convs = []
            for layer in self.layers:
                convs += [layer.conv_0, layer.conv_1, layer.conv_2, layer.conv_3] if isinstance(layer, GenBlock) else \
                    [layer.snconv1x1_theta, layer.snconv1x1_phi, layer.snconv1x1_g, layer.snconv1x1_o_conv]
            self.__dict__["_conv_sns"] = convs
        packed = _pack_all(self, convs, dt, saved is not None)
        wz = self.gen_z.effective_weight(training).contiguous()
        z = ops.linear(cond_vector, wz, self.gen_z.bias.detach())   # [B, 4*4*16ch] == NHWC
        x = ops.nchw_to_nhwc(z.view(B, 4 * 4 * 16 * ch, 1, 1), B, dt).view(B, 4, 4, 16 * ch)
        layers = [] if saved is not None else None
        for layer in self.layers:
            x = layer.run(x, cond_vector, truncation, dt, training, layers, packed) if isinstance(layer, GenBlock) else \
                layer.run(x, dt, training, layers, packed)
        a, b = self.bn.affine(truncation, None, training)
        a, b = a.expand(B, -1).contiguous(), b.expand(B, -1).contiguous()
        w = self.conv_to_rgb.effective_weight(training)[:16].contiguous()            # only channels 0..2 are used (:253)
        y = ops.conv2d(x, ops.pack_conv_weight(w, ops.PACK_FWD, dt, 1.0), 16, 3, in_scale=a, in_shift=b, in_relu=True,
                       bias=self.conv_to_rgb.bias.detach()[:16].contiguous())
        img = ops.rgb_tanh(y)
        if saved is not None:
            saved.update(wz=wz, layers=layers, x_last=x, a=a, b=b, w_rgb=w, img=img, dt=dt)
        return img

    def backward(self, saved, g_img):
        """dL/dcond_vector [B, 2*z_dim] for a gradient g_img on the image (hand-written data gradient; parameters frozen)."""
        dt = saved["dt"]
        x = saved["x_last"]
        B, ch = x.shape[0], x.shape[3]
        g_cond = torch.zeros((B, saved["wz"].shape[1]), dtype=torch.float32, device=x.device)
        g_y = ops.rgb_tanh_bwd(g_img.float(), saved["img"], 16, dt)
        g_u = ops.conv2d(g_y, ops.pack_conv_weight(saved["w_rgb"], ops.PACK_DGRAD, dt, 1.0), ch, 3)
        g, _ = ops.affine_relu_bwd(g_u, x, saved["a"], saved["b"])
        for kind, layer, rec in reversed(saved["layers"]):
            g = layer.backward(rec, g, g_cond, dt) if kind == "block" else layer.backward(rec, g, dt)
        g_flat = ops.nhwc_to_nchw(g.reshape(B, 1, 1, -1)).view(B, -1)               # f32, same (y, x, c) order as gen_z's output
        ops.linear_t(g_flat, saved["wz"], g_cond, accumulate=True)
        return g_cond


class _BigGANFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z, embed, truncation):
        need = ctx.needs_input_grad[1]
        cond_vector = torch.cat((z.detach().float(), embed), dim=1).contiguous()
        saved = {} if need else None
        img = G.generator(cond_vector, truncation, G.compute_dtype, saved)
        ctx.G, ctx.saved_acts, ctx.zdim = G, saved, z.shape[1]
        ctx.mark_non_differentiable(cond_vector)
        return img, cond_vector

    @staticmethod
    def backward(ctx, g_img, _g_cond):
        if ctx.saved_acts is None:
            raise RuntimeError("BigGAN forward ran without saved activations")
        g_cond = ctx.G.generator.backward(ctx.saved_acts, g_img.contiguous())
        return None, g_cond[:, :ctx.zdim].contiguous(), None, None


class BigGAN(nn.Module):
    def __init__(self, config, compute_dtype="bf16"):
        super().__init__()
        _dt(compute_dtype)
        self.config, self.compute_dtype = config, compute_dtype
        self.embeddings = nn.Linear(config.num_classes, config.z_dim, bias=False)
        self.generator = Generator(config)

    def forward(self, z, class_label, truncation):
        truncation = truncation.detach().cpu() if torch.is_tensor(truncation) else float(truncation)
        assert 0 < float(truncation) <= 1
        with torch.no_grad():
            embed = ops.linear(class_label.float().contiguous(), self.embeddings.weight.detach())
        return _BigGANFunction.apply(self, z, embed, truncation)          # differentiable w.r.t. z (E_align_s2.py:162)
