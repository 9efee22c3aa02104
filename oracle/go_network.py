"""CPU restatement of the GO-hierarchical attention network (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/kernel/go_model.py:
  index sets      :42-74  (+ store_ind :161-168)      -> go_index_sets
  parameters      :78-157                              -> go_param_shapes / init_go_state
  forward         :205-287 (helpers :170-201)          -> go_forward

Functional over a flat state_dict whose keys equal the reference module's
(``t.0``, ``w_inc.0.weight``, ``G_B.1.bias``, ``conc_for_attention.1.running_mean`` ...).
``faithful=True`` keeps the reference's operator sequence (torch.sparse.mm per sample, one
sparse matrix per sample and layer) and is what bench.py times as the CPU baseline;
``faithful=False`` computes the same numbers with batched index_add (fast, used by tests).
"""
import torch
import torch.nn.functional as F

from .pyg_ops import scatter_sum_dim1

N_SNPS = 54


def _coo_rc(sp):
    """(row, col) int64 of the structural non-zeros of a 2-D sparse COO tensor, row-major."""
    sp = sp.coalesce()
    idx = sp.indices()
    keep = sp.values() != 0            # to_dense().to_sparse() round trip drops stored zeros
    return idx[0][keep].cpu(), idx[1][keep].cpu()


def go_index_sets(A_g, A, pool, n_l=2):
    """Per-layer edge lists of the hierarchy (go_model.py:42-88).

    A   : sparse [N,N], row aggregates from col (the trainer passes adj.T, train_eval_sgcn_img_snps.py:69)
    A_g : sparse [N,54] GO-node x SNP membership
    pool: [p0, p1, ...] level sizes, deepest level first
    Encoder layer i works on nodes >= off_i = sum(pool[:i]) and uses the block A[off_i:, off_i:]
    re-based to 0 (:51-61; the reference re-slices the previous block, which compounds to the
    same cumulative offset).  Decoder layer i uses A^T[sum(pool[:n_l-i-1]):, sum(pool[:n_l-i]):] (:68-74).
    """
    pool = [int(p) for p in pool]
    n = int(A.shape[0])
    r, c = _coo_rc(A)
    enc = []
    for i in range(n_l):
        off = sum(pool[:i])
        m = (r >= off) & (c >= off)
        enc.append((r[m] - off, c[m] - off, n - off))
    rt, ct = _coo_rc(A.t())
    dec = []
    for i in range(n_l):
        ro, co = sum(pool[:n_l - i - 1]), sum(pool[:n_l - i])
        m = (rt >= ro) & (ct >= co)
        dec.append((rt[m] - ro, ct[m] - co, n - ro, n - co))
    gn, gs = _coo_rc_keep_all(A_g)
    dn, dsn = _coo_rc_keep_all(A_g.t())     # rows = SNP, cols = GO node
    return dict(n=n, pool=pool, n_l=n_l, enc=enc, dec=dec, gene=(gn, gs), gene_t=(dn, dsn),
                n_top=n - sum(pool[:n_l]))


def _coo_rc_keep_all(sp):
    sp = sp.coalesce()
    idx = sp.indices()
    return idx[0].cpu(), idx[1].cpu()


def go_param_shapes(idx, in_f=2, f_dim=(5, 5), l_dim=32, d_att=5):
    """name -> shape for every parameter and buffer (go_model.py:78-157), in module order."""
    n, pool, n_l, n_top = idx["n"], idx["pool"], idx["n_l"], idx["n_top"]
    fd = [in_f] + list(f_dim)
    nnzg = idx["gene"][0].numel()
    shp = {}
    for c in range(in_f):
        shp[f"t.{c}"] = (nnzg,)
    shp["t_D.0"] = (nnzg,)
    for i in range(n_l):
        shp[f"w_inc.{i}.weight"] = (fd[i + 1], fd[i])
    for i in range(n_l):
        shp[f"w_s_loop.{i}.weight"] = (fd[i + 1], fd[i])
    for i in range(n_l):
        shp[f"w_att_s.{i}.weight"] = (1, fd[i + 1])
    for i in range(n_l):
        shp[f"G_B.{i}.weight"] = (sum(pool[i:]),)
        shp[f"G_B.{i}.bias"] = (sum(pool[i:]),)
    for i in range(n_l):
        shp[f"w_att_in.{i}.weight"] = (1, 2 * fd[i + 1])
    for j, i in enumerate(range(n_l, 0, -1)):
        shp[f"w_out.{j}.weight"] = (fd[i - 1], fd[i])
    for j, i in enumerate(range(n_l, 0, -1)):
        shp[f"w_s_loop_out.{j}.weight"] = (fd[i - 1], fd[i])
    for j, i in enumerate(range(n_l - 1, -1, -1)):
        shp[f"G_B_D.{j}.weight"] = (sum(pool[i:]),)
        shp[f"G_B_D.{j}.bias"] = (sum(pool[i:]),)
    shp["conc_for_attention.0.weight"] = (d_att, fd[-1])
    _bn(shp, "conc_for_attention.1", n_top)
    shp["conc.weight"] = (1, fd[-1])
    _bn(shp, "B.0", n_top)
    shp["conc_D.weight"] = (1, fd[0])
    _bn(shp, "B_D.0", n)
    shp["latent.0.weight"] = (32, n_top)
    _bn(shp, "latent.1", 32)
    shp["latent.4.weight"] = (l_dim, 32)
    _bn(shp, "latent.5", l_dim)
    _bn(shp, "classification.0", l_dim + N_SNPS)
    shp["classification.3.weight"] = (16, l_dim + N_SNPS)
    shp["classification.6.weight"] = (1, 16)
    shp["classification.6.bias"] = (1,)
    return shp


def _bn(shp, name, c):
    shp[name + ".weight"] = (c,)
    shp[name + ".bias"] = (c,)
    shp[name + ".running_mean"] = (c,)
    shp[name + ".running_var"] = (c,)
    shp[name + ".num_batches_tracked"] = ()


def _batch_norm(sd, name, x, training):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training, 0.1, 1e-5)


def _node_dropout(x, p, training, enabled):
    """nn.Dropout2d on a 3-D [B,N,f] tensor: zeroes whole nodes (dim-1 slices) per sample."""
    if not (training and enabled):
        return x
    keep = torch.bernoulli(torch.full((x.shape[0], x.shape[1], 1), 1.0 - p, dtype=x.dtype))
    return x * keep / (1.0 - p)


def _dropout(x, p, training, enabled):
    return F.dropout(x, p, True) if (training and enabled) else x


def _row_normalise(row, v, n_rows):
    """attention_adj (go_model.py:173-180): v_e / sum of v over the edges sharing e's row."""
    z = torch.zeros(*v.shape[:-1], n_rows, dtype=v.dtype).index_add(v.dim() - 1, row, v)
    return v / z.index_select(v.dim() - 1, row)


def go_forward(sd, idx, snps, training=False, dropout=True, faithful=False, prefix=""):
    """go_model.py:205-287.  snps [B,54] -> (latent [B,l], x_D [B,54], atten_out [B,Ntop,d_att])."""
    g = lambda k: sd[prefix + k]   # noqa: E731
    sdp = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    n, pool, n_l = idx["n"], idx["pool"], idx["n_l"]
    bsz = snps.shape[0]
    gn, gs = idx["gene"]

    # gene encoding :208-215 -- x[b,node,c] = sum_snp t_c[(node,snp)] * snps[b,snp]
    chans = []
    c = 0
    while prefix + f"t.{c}" in sd:
        if faithful:
            w = torch.sparse_coo_tensor(torch.stack([gn, gs]), g(f"t.{c}"), (n, N_SNPS))
            chans.append(torch.sparse.mm(w, snps.t()).t())
        else:
            chans.append(torch.zeros(bsz, n, dtype=snps.dtype).index_add(1, gn, snps[:, gs] * g(f"t.{c}")))
        c += 1
    x = torch.stack(chans, dim=2)

    # encoder layers :219-251
    for j in range(n_l):
        row, col, nj = idx["enc"][j]
        x_in = x @ g(f"w_inc.{j}.weight").t()
        x_s = x @ g(f"w_s_loop.{j}.weight").t()
        pair = torch.cat([x_in[:, row, :], x_in[:, col, :]], dim=2)
        v = torch.exp(torch.tanh(pair @ g(f"w_att_in.{j}.weight").t())).squeeze(2)      # [B,nnz]
        v_s = torch.sigmoid(x_s @ g(f"w_att_s.{j}.weight").t())                          # [B,nj,1]
        if faithful:
            out = torch.zeros(bsz, nj, x_in.shape[2], dtype=x.dtype)
            ii = torch.stack([row, col])
            outs = []
            for k in range(bsz):
                tot = torch.sparse.sum(torch.sparse_coo_tensor(ii, v[k], (nj, nj)), dim=1)
                # values() of the row sums are indexed by rank among non-empty rows (store_ind :161-168)
                rank = torch.unique_consecutive(row, return_inverse=True)[1]
                a_hat = torch.sparse_coo_tensor(ii, v[k] / tot.values()[rank], (nj, nj))
                outs.append(torch.sparse.mm(a_hat, x_in[k]) + x_s[k] * v_s[k])
            out = torch.stack(outs)
        else:
            alpha = _row_normalise(row, v, nj)
            out = torch.zeros(bsz, nj, x_in.shape[2], dtype=x.dtype).index_add(
                1, row, alpha.unsqueeze(2) * x_in[:, col, :]) + x_s * v_s
        out = F.layer_norm(out.permute(0, 2, 1), (nj,), g(f"G_B.{j}.weight"), g(f"G_B.{j}.bias"),
                           1e-5).permute(0, 2, 1)
        out = _node_dropout(torch.relu(out), 0.4, training, dropout)
        x = out[:, pool[j]:, :]

    # read-outs :254-255
    att = x @ g("conc_for_attention.0.weight").t()
    atten_out = torch.relu(_batch_norm(sdp, "conc_for_attention.1", att, training))
    inp = (x @ g("conc.weight").t()).squeeze(2)
    inp_out = _dropout(torch.relu(_batch_norm(sdp, "B.0", inp, training)), 0.5, training, dropout)

    # decoder layers :258-275 (mean aggregation back down the hierarchy)
    for j in range(n_l):
        row, col, n_rows, n_cols = idx["dec"][j]
        x_out = x @ g(f"w_out.{j}.weight").t()
        x_s_out = x @ g(f"w_s_loop_out.{j}.weight").t()
        v_out = _row_normalise(row, torch.ones(row.numel(), dtype=x.dtype), n_rows)
        agg = scatter_sum_dim1(v_out.view(1, -1, 1) * x_out[:, col, :], row, n_rows)
        self_term = torch.zeros_like(agg)
        self_term[:, pool[n_l - j - 1]:, :] = x_s_out
        y = agg + self_term
        y = F.layer_norm(y.permute(0, 2, 1), (n_rows,), g(f"G_B_D.{j}.weight"), g(f"G_B_D.{j}.bias"),
                         1e-5).permute(0, 2, 1)
        x = _node_dropout(torch.relu(y), 0.4, training, dropout)

    # gene decoding :278-282
    out_d = (x @ g("conc_D.weight").t()).squeeze(2)
    out_d = _dropout(torch.relu(_batch_norm(sdp, "B_D.0", out_d, training)), 0.5, training, dropout)
    dsn, dn = idx["gene_t"]                      # rows = SNP, cols = GO node
    if faithful:
        w_d = torch.sparse_coo_tensor(torch.stack([dsn, dn]), g("t_D.0"), (N_SNPS, n))
        x_d = torch.sparse.mm(w_d, out_d.t()).t()
    else:
        x_d = torch.zeros(bsz, N_SNPS, dtype=x.dtype).index_add(1, dsn, out_d[:, dn] * g("t_D.0"))

    # latent projection :138-146,285
    h = inp_out.view(bsz, -1) @ g("latent.0.weight").t()
    h = _dropout(torch.relu(_batch_norm(sdp, "latent.1", h, training)), 0.5, training, dropout)
    h = h @ g("latent.4.weight").t()
    latent = torch.relu(_batch_norm(sdp, "latent.5", h, training))
    return latent, x_d, atten_out


def init_go_state(idx, l_dim=32, d_att=5, seed=0, dtype=torch.float32):
    """Random state_dict with the reference's init distributions (go_model.py:80-88 N(1,0.1) for
    t/t_D; nn.Linear default kaiming-uniform(a=sqrt5); norm layers ones/zeros)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in go_param_shapes(idx, l_dim=l_dim, d_att=d_att).items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(s, dtype=dtype)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(s, dtype=dtype)
        elif k.startswith("t.") or k.startswith("t_D."):
            sd[k] = (1.0 + 0.1 * torch.randn(s, generator=gen)).to(dtype)
        elif len(s) == 1 and k.endswith(".weight"):
            sd[k] = torch.ones(s, dtype=dtype)
        elif k.endswith(".bias") and len(s) == 1 and not k.startswith("classification.6"):
            sd[k] = torch.zeros(s, dtype=dtype)
        else:
            fan_in = s[-1] if len(s) > 1 else 16
            bound = 1.0 / fan_in ** 0.5
            sd[k] = ((torch.rand(s, generator=gen) * 2 - 1) * bound).to(dtype)
    return sd
