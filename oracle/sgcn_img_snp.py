"""CPU restatement of SGCN_GCN_IMGSNP + its losses + one train step (TEST INFRASTRUCTURE ONLY).

Follows /root/reference:
  kernel/sgcn_img_snp.py:133-151   cal_probability          -> edge_and_region_masks
  kernel/sgcn_img_snp.py:153-181   loss_probability         -> loss_probability
  kernel/sgcn_img_snp.py:183-196   consist_loss             -> consist_loss
  util/image_cluster.py:15-31      rbf_kernel_torch         -> (inside consist_loss)
  kernel/sgcn_img_snp.py:198-205   OrthogonalConstraint     -> orthogonal_constraint
  kernel/sgcn_img_snp.py:207-307   forward (default branch: isCrossAtten, both modalities,
                                   isuseProb4Regr; the isImageOnly and isSNPsOnly heads) -> model_forward
  kernel/train_eval_sgcn_img_snps.py:511-548  train()       -> train_step
  sgcn_hyperparameters.py:18-23    lamda_*                  -> HP

Functional over a flat state_dict with the reference's key names (``conv1.lin.weight``,
``convs.0.bias``, ``multihead_attn.in_proj_weight``, ``prob``, ``go_network.t.0`` ...).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import go_network as G
from .pyg_ops import gcn_conv, global_pools, to_dense_batch

HP = SimpleNamespace(lamda_x_l1=0.1, lamda_e_l1=0.1, lamda_x_ent=0.1, lamda_e_ent=0.1,
                     lamda_mi=1, lamda_ce=1)                      # sgcn_hyperparameters.py:18-23
DEFAULT_LAMBDA = [0.0, 1.0, 0.5, 1.5e-6, 0.1, 0.0]                # main.py:73-78,204


def edge_and_region_masks(sd, x, edge_index, edge_weight, rois, snps=None):
    """cal_probability :133-151."""
    n, h0 = x.shape
    xm = (x.view(n // rois, rois, h0) * sd["prob"]).reshape(n, h0)
    pair = torch.cat([xm[edge_index[0]], xm[edge_index[1]]], dim=-1)
    e = torch.sigmoid(pair @ sd["prob_bias"]).view(-1)
    out = [xm, edge_weight * e, e]
    if snps is not None:
        out.append(snps * torch.sigmoid(sd["snps_prob"]))
    return out


def _bin_entropy_and_l1(p, eps):
    n = p.numel()
    l1 = p.abs().sum() / n
    ent = -(p * torch.log(p + eps) + (1 - p) * torch.log((1 - p) + eps)).sum() / n
    return l1, ent


def loss_probability(sd, x, edge_index, edge_weight, rois, hp=HP, eps=1e-6):
    """:153-181 — L1 + binary-entropy regularisers on sigma(prob), the edge mask and sigma(snps_prob)."""
    _, _, e = edge_and_region_masks(sd, x, edge_index, edge_weight, rois)
    f_l1, f_ent = _bin_entropy_and_l1(torch.sigmoid(sd["prob"]), eps)
    e_l1, e_ent = _bin_entropy_and_l1(e, eps)
    s_l1, s_ent = _bin_entropy_and_l1(torch.sigmoid(sd["snps_prob"]), eps)
    l1 = hp.lamda_x_l1 * f_l1 + hp.lamda_e_l1 * e_l1 + hp.lamda_x_l1 * s_l1
    ent = hp.lamda_x_ent * f_ent + hp.lamda_e_ent * e_ent + hp.lamda_x_ent * s_ent
    return l1 + ent


def consist_loss(s, tsne, rbf_gamma, soft=True):
    """:183-196 with rbf_kernel_torch (util/image_cluster.py:15-31): tr(s^T (D-W) s) / B^2."""
    b = s.shape[0]
    if b == 0:
        return 0
    if soft and tsne is not None:
        w = torch.exp(-rbf_gamma * torch.cdist(tsne, tsne, p=2) ** 2)
    else:
        w = torch.ones(b, b, dtype=s.dtype)
    lap = torch.eye(b, dtype=s.dtype) * w.sum(dim=1) - w
    return torch.trace(s.t() @ lap @ s) / (b * b)


def orthogonal_constraint(w):
    """:198-205 — rows L2-normalised, ||W^T W - I||_F^2 / B^2 with W^T W of size (R*D)^2."""
    wn = w / w.norm(dim=1)[:, None]
    gram = wn.t() @ wn
    pen = torch.norm(gram - torch.eye(wn.shape[1], dtype=w.dtype)) ** 2
    return pen / (wn.shape[0] * wn.shape[0])


def _mha(sd, q_in, kv_in, heads=2):
    """torch.nn.MultiheadAttention(D, 2, batch_first=True) forward, weights discarded (:46,240)."""
    d = q_in.shape[-1]
    w, b = sd["multihead_attn.in_proj_weight"], sd["multihead_attn.in_proj_bias"]
    q = q_in @ w[:d].t() + b[:d]
    k = kv_in @ w[d:2 * d].t() + b[d:2 * d]
    v = kv_in @ w[2 * d:].t() + b[2 * d:]
    bsz, lq, lk, hd = q.shape[0], q.shape[1], k.shape[1], d // heads
    q = q.view(bsz, lq, heads, hd).transpose(1, 2)
    k = k.view(bsz, lk, heads, hd).transpose(1, 2)
    v = v.view(bsz, lk, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(bsz, lq, d)
    return o @ sd["multihead_attn.out_proj.weight"].t() + sd["multihead_attn.out_proj.bias"]


def model_forward(sd, cfg, go_idx, data, is_explain=False, training=False, dropout=True,
                  faithful=False):
    """SGCN_GCN_IMGSNP.forward :207-307.

    cfg: SimpleNamespace(num_layers, rois[, image_only, snps_only, use_prob4regr]) ; data: object with x, edge_index, edge_attr,
    batch, snps_feat.  Returns the reference 6-tuple.
    """
    x, ei, batch, ew, snps = data.x, data.edge_index, data.batch, data.edge_attr, data.snps_feat
    rois = cfg.rois
    if is_explain:
        xm, ewm, _, snpsm = edge_and_region_masks(sd, x, ei, ew, rois, snps)
    else:
        xm, ewm, snpsm = x, ew, snps

    hs = [torch.relu(gcn_conv(xm, ei, ewm, sd["conv1.lin.weight"], sd["conv1.bias"]))]
    i = 0
    while f"convs.{i}.lin.weight" in sd:
        hs.append(torch.relu(gcn_conv(hs[-1], ei, ewm, sd[f"convs.{i}.lin.weight"], sd[f"convs.{i}.bias"])))
        i += 1
    xcat = torch.cat(hs, dim=1)
    dense, _ = to_dense_batch(xcat, batch, float(xcat.min()) - 1)            # :225-226
    bsz = dense.shape[0]
    img_out = dense.reshape(bsz, -1)
    graph_pool = getattr(cfg, "graph_pool", False)
    if graph_pool:                                                            # :230-235
        img_out = global_pools(xcat, batch)

    latent, x_hat, atten_out = G.go_forward(sd, go_idx, snpsm, training, dropout, faithful,
                                            prefix="go_network.")
    image_only, snps_only = getattr(cfg, "image_only", False), getattr(cfg, "snps_only", False)
    if image_only:                                                            # :257-276
        out_z = img_out
        out_lin = out_z
    elif snps_only:                                                           # :277-285
        out_z = latent
        out_lin = torch.cat([snpsm, latent], dim=-1)
    else:                                                                     # :239-242,286-288
        out_cross = torch.relu(_mha(sd, dense, atten_out))
        if graph_pool:                                                        # :246-252
            out_cross = global_pools(out_cross.reshape(-1, out_cross.shape[2]), batch)
        else:
            out_cross = out_cross.reshape(bsz, -1)
        out_z = (img_out + out_cross) / 2
        out_lin = torch.cat([out_z, latent], dim=-1)
    lin_f = torch.relu(out_lin @ sd["lin1.weight"].t() + sd["lin1.bias"])
    h = F.dropout(lin_f, 0.5, True) if (training and dropout) else lin_f
    logits = h @ sd["lin2.weight"].t() + sd["lin2.bias"]
    if getattr(cfg, "use_prob4regr", True) and not snps_only:
        xd, _ = to_dense_batch(data.x, batch, float(data.x.min()) - 1)        # :293-297 (isuseProb4Regr)
        img_feat = (xd * sd["prob"]).reshape(bsz, -1)
        feat = torch.cat([out_lin, img_feat], dim=-1)
    else:
        feat = out_lin
    r = torch.relu(feat @ sd["lin1_regr.weight"].t() + sd["lin1_regr.bias"])
    r = F.dropout(r, 0.3, True) if (training and dropout) else r
    reg = r @ sd["lin2_regr.weight"].t() + sd["lin2_regr.bias"]
    return F.log_softmax(logits, dim=-1), x_hat, out_z, out_lin, lin_f, reg


def train_losses(sd, cfg, go_idx, data, lam=None, dropout=True, faithful=False):
    """The loss of train() :521-543 (model in training mode).  Returns (loss, dict of terms)."""
    lam = DEFAULT_LAMBDA if lam is None else lam
    y = data.y.view(-1)
    o1 = model_forward(sd, cfg, go_idx, data, False, True, dropout, faithful)
    o2 = model_forward(sd, cfg, go_idx, data, True, True, dropout, faithful)
    clin = data.clini_score.view(-1)
    t = {}
    t["ce"] = lam[0] * F.nll_loss(o1[0], y)
    t["mi"] = lam[0] * F.nll_loss(o2[0], y)
    t["reg"] = lam[1] * (F.mse_loss(o1[5].view(-1), clin) + F.mse_loss(o2[5].view(-1), clin)) / 2
    t["prob"] = lam[2] * loss_probability(sd, data.x, data.edge_index, data.edge_attr, cfg.rois)
    t["recon"] = lam[3] * (((o1[1] - data.snps_feat) ** 2).sum() + ((o2[1] - data.snps_feat) ** 2).sum()) / 2
    t["cluster"] = lam[4] * (consist_loss(o1[2], data.tsne_fdim, cfg.rbf_gamma)
                             + consist_loss(o2[2], data.tsne_fdim, cfg.rbf_gamma)) / 2
    t["orth"] = lam[5] * orthogonal_constraint(o1[2])
    if lam[0] == 0:                                                           # :540-542
        t["ce"] = 0.0
        t["mi"] = 0.0
    loss = HP.lamda_ce * t["ce"] + HP.lamda_mi * t["mi"] + t["reg"] + t["prob"] + t["recon"] \
        + t["cluster"] + t["orth"]
    return loss, t, (o1, o2)


def trainable_keys(sd):
    return [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]


def make_leaf_state(sd, dtype=None):
    """Detach-clone a state_dict; floating parameters become autograd leaves."""
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if dtype is not None and v.dtype.is_floating_point:
            v = v.to(dtype)
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


def train_step(sd, cfg, go_idx, data, lr=1e-3, lam=None, dropout=True, faithful=False, opt=None):
    """One iteration of train() :515-547: zero_grad, 2 forwards, losses, backward, Adam(wd=0)."""
    keys = trainable_keys(sd)
    if opt is None:
        opt = torch.optim.Adam([sd[k] for k in keys], lr=lr)
    opt.zero_grad()
    data.x.requires_grad_(True)                                               # :210
    loss, terms, outs = train_losses(sd, cfg, go_idx, data, lam, dropout, faithful)
    loss.backward()
    opt.step()
    return loss.detach(), terms, opt


def sgcn_param_shapes(num_layers, hidden, rois=90, h0=3, l_dim=32, num_classes=3, num_regr=3,
                      hidden_linear=64, image_only=False, snps_only=False, cross_atten=True, use_prob4regr=True,
                      graph_pool=False):
    """Top-level parameter shapes of SGCN_GCN_IMGSNP (:34-101) for the head selected by the flags."""
    d = num_layers * hidden
    shp = {"prob": (rois, h0), "prob_bias": (2 * h0, 1), "edge_prob": (rois, rois), "snps_prob": (1, 54),
           "conv1.bias": (hidden,), "conv1.lin.weight": (hidden, h0)}
    for i in range(num_layers - 1):
        shp[f"convs.{i}.bias"] = (hidden,)
        shp[f"convs.{i}.lin.weight"] = (hidden, hidden)
    if cross_atten:
        shp["multihead_attn.in_proj_weight"] = (3 * d, d)
        shp["multihead_attn.in_proj_bias"] = (3 * d,)
        shp["multihead_attn.out_proj.weight"] = (d, d)
        shp["multihead_attn.out_proj.bias"] = (d,)
    lin_in = rois * d if image_only else (l_dim + 54 if snps_only else rois * d + l_dim)
    reg_in = lin_in + (rois * h0 if (use_prob4regr and not snps_only) else 0)
    if graph_pool:                                                            # :51-54
        lin_in = reg_in = 3 * d + l_dim
    shp["lin1.weight"] = (hidden_linear, lin_in)
    shp["lin1.bias"] = (hidden_linear,)
    shp["lin1_regr.weight"] = (hidden_linear, reg_in)
    shp["lin1_regr.bias"] = (hidden_linear,)
    shp["lin2.weight"] = (num_classes, hidden_linear)
    shp["lin2.bias"] = (num_classes,)
    shp["lin2_regr.weight"] = (num_regr, hidden_linear)
    shp["lin2_regr.bias"] = (num_regr,)
    return shp
