"""CPU restatement of the image-only sibling ``SGCN_GCN`` and its train loss (TEST INFRASTRUCTURE ONLY).

Follows /root/reference:
  kernel/sgcn.py:321-332        cal_probability    -> (oracle.sgcn_img_snp.edge_and_region_masks: same formula)
  kernel/sgcn.py:334-358        loss_probability   -> loss_probability  (node L1 term divided by rois, no SNP term)
  kernel/sgcn.py:360-388        forward            -> model_forward
  kernel/train_eval_sgcn.py:303-308  train() loss  -> train_losses

Pinned by tests/golden/sgcn_only.npz (the reference class executed with oracle.pyg_ops standing in for the absent
PyG GCNConv / to_dense_batch: those two stay "parity unpinned", see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from .pyg_ops import gcn_conv, to_dense_batch
from .sgcn_img_snp import HP, edge_and_region_masks


def loss_probability(sd, x, edge_index, edge_weight, rois, hp=HP, eps=1e-6):
    _, _, e = edge_and_region_masks(sd, x, edge_index, edge_weight, rois)
    p = torch.sigmoid(sd["prob"])
    n, d = p.shape
    f_l1 = p.abs().sum(dim=-1).sum() / n
    f_ent = -(p * torch.log(p + eps) + (1 - p) * torch.log((1 - p) + eps)).sum() / (n * d)
    m = e.shape[0]
    e_l1 = e.abs().sum() / m
    e_ent = -(e * torch.log(e + eps) + (1 - e) * torch.log((1 - e) + eps)).sum() / m
    return hp.lamda_x_l1 * f_l1 + hp.lamda_e_l1 * e_l1 + hp.lamda_x_ent * f_ent + hp.lamda_e_ent * e_ent


def model_forward(sd, rois, data, is_explain=False, training=False, dropout=True):
    x, ei, batch, ew = data.x, data.edge_index, data.batch, data.edge_attr
    if is_explain:
        xm, ewm, _ = edge_and_region_masks(sd, x, ei, ew, rois)
    else:
        xm, ewm = x, ew
    hs = [torch.relu(gcn_conv(xm, ei, ewm, sd["conv1.lin.weight"], sd["conv1.bias"]))]
    i = 0
    while f"convs.{i}.lin.weight" in sd:
        hs.append(torch.relu(gcn_conv(hs[-1], ei, ewm, sd[f"convs.{i}.lin.weight"], sd[f"convs.{i}.bias"])))
        i += 1
    xcat = torch.cat(hs, dim=1)
    dense, _ = to_dense_batch(xcat, batch, float(xcat.min()) - 1)
    z = dense.reshape(dense.shape[0], -1)
    h = torch.relu(z @ sd["lin1.weight"].t() + sd["lin1.bias"])
    h = F.dropout(h, 0.5, True) if (training and dropout) else h
    return F.log_softmax(h @ sd["lin2.weight"].t() + sd["lin2.bias"], dim=-1)


def train_losses(sd, rois, data, hp=HP, training=True, dropout=True):
    y = data.y.view(-1)
    out = model_forward(sd, rois, data, False, training, dropout)
    out_p = model_forward(sd, rois, data, True, training, dropout)
    t = {"ce": F.nll_loss(out, y), "mi": F.nll_loss(out_p, y),
         "prob": loss_probability(sd, data.x, data.edge_index, data.edge_attr, rois, hp)}
    return hp.lamda_ce * t["ce"] + t["prob"] + hp.lamda_mi * t["mi"], t, (out, out_p)


def param_shapes(num_layers, hidden, rois=90, h0=3, num_classes=2, hidden_linear=64):
    shp = {"prob": (rois, h0), "prob_bias": (2 * h0, 1), "edge_prob": (rois, rois),
           "conv1.bias": (hidden,), "conv1.lin.weight": (hidden, h0)}
    for i in range(num_layers - 1):
        shp[f"convs.{i}.bias"] = (hidden,)
        shp[f"convs.{i}.lin.weight"] = (hidden, hidden)
    shp.update({"lin1.weight": (hidden_linear, rois * num_layers * hidden), "lin1.bias": (hidden_linear,),
                "lin2.weight": (num_classes, hidden_linear), "lin2.bias": (num_classes,)})
    return shp
