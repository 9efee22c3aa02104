"""GPU parity, op by op: every libigcn entry point (called through the C ABI via igcn_amd.ops) against the
CPU oracle / a plain PyTorch fp64 restatement of the same operator on identical seeded inputs.
Tolerance: scale-relative 1e-4 (north_star: 1e-4 fp32); index work (graph plan) is bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import assert_matches

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib, ops as o
    _lib.load()          # raises if libigcn.so is missing: no fallback
    return o


def _rand_graph(rng, n, e, loops=True, multi_loops=False, isolated=True):
    src = rng.integers(0, n, e)
    dst = rng.integers(0, n, e)
    if not loops:
        dst = np.where(dst == src, (dst + 1) % n, dst)
    if isolated and n > 3:
        src = np.where(src == n - 1, 0, src)
        dst = np.where(dst == n - 1, 1, dst)
    # at most one stored loop per node unless asked for (several stored loops on a node make the
    # reference's index_put gradient undefined)
    seen, keep = set(), np.ones(src.size, dtype=bool)
    for k in range(src.size):
        if src[k] == dst[k]:
            if int(src[k]) in seen or (multi_loops and src[k] == 2):
                keep[k] = False
            seen.add(int(src[k]))
    src, dst = src[keep], dst[keep]
    if multi_loops:
        src = np.concatenate([src, [2, 2, 2]])
        dst = np.concatenate([dst, [2, 2, 2]])
    ei = torch.from_numpy(np.vstack([src, dst])).long()
    w = torch.from_numpy(rng.random(ei.shape[1]) + 0.05).float()
    return ei, w


# ------------------------------------------------------------------------------------------------ plan
@pytest.mark.parametrize("n,e,seed", [(7, 0, 0), (90, 270, 1), (1, 5, 2), (1000, 20000, 3), (23040, 69120, 4),
                                       (4096, 300000, 5), (2, 9000, 6), (1025, 4097, 7), (1 << 21, 50000, 8),
                                       (1500000, 700000, 9)])     # 1, 2 and 3 LSD passes of the hand-written sort
def test_graph_plan_bit_exact(ops, n, e, seed):
    rng = np.random.default_rng(seed)
    ei = torch.from_numpy(rng.integers(0, n, (2, e))).long()
    plan = ops.GraphPlan(ei.cuda(), n)
    torch.cuda.synchronize()
    src, dst = ei[0].numpy(), ei[1].numpy()
    for key, ptr_t, perm_t in ((dst, plan.tgt_ptr, plan.tgt_perm), (src, plan.src_ptr, plan.src_perm)):
        perm = np.argsort(key, kind="stable").astype(np.int32)
        ptr_ref = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=n))]).astype(np.int32)
        assert np.array_equal(ptr_t.cpu().numpy(), ptr_ref)
        if e:
            assert np.array_equal(perm_t.cpu().numpy()[:e], perm)
    if e:
        assert np.array_equal(plan.src32.cpu().numpy()[:e], src.astype(np.int32))
        assert np.array_equal(plan.dst32.cpu().numpy()[:e], dst.astype(np.int32))
        # int64 round trip of the permutation: gathering edge_index through it is exactly sorted edge_index
        back = ei[:, plan.tgt_perm.cpu().long()[:e]]
        assert torch.equal(back[1], torch.sort(ei[1], stable=True)[0])
    loop = np.full(n, -1, dtype=np.int32)
    for k in range(e):
        if src[k] == dst[k]:
            loop[src[k]] = k
    assert np.array_equal(plan.loop_edge.cpu().numpy(), loop)


# ------------------------------------------------------------------------------------------------ masks
@pytest.mark.parametrize("bsz,rois,h0,seed,deg", [(3, 10, 3, 0, 6), (8, 90, 3, 1, 6), (2, 17, 5, 2, 6),
                                                  (3, 70, 3, 3, 40)])      # deg >= 16: wave-per-node backward
def test_edge_mask_fwd_bwd(ops, bsz, rois, h0, seed, deg):
    from oracle import sgcn_img_snp as OS
    rng = np.random.default_rng(seed)
    n = bsz * rois
    ei, ew = _rand_graph(rng, n, deg * n)
    x = torch.from_numpy(rng.random((n, h0))).float()
    sd = {"prob": torch.from_numpy(rng.standard_normal((rois, h0))).float(),
          "prob_bias": torch.from_numpy(rng.standard_normal((2 * h0, 1))).float()}
    cots = [torch.from_numpy(rng.standard_normal(s)).float() for s in ((n, h0), (ei.shape[1],), (ei.shape[1],))]
    # oracle (fp64 autograd)
    ref_in = [t.double().requires_grad_(True) for t in (x, sd["prob"], sd["prob_bias"])]
    xm, ewm, e = OS.edge_and_region_masks({"prob": ref_in[1], "prob_bias": ref_in[2]}, ref_in[0], ei, ew.double(),
                                          rois)
    g_ref = torch.autograd.grad(sum((o * c.double()).sum() for o, c in zip((xm, ewm, e), cots)), ref_in)
    # HIP
    dev_in = [t.cuda().requires_grad_(True) for t in (x, sd["prob"], sd["prob_bias"])]
    plan = ops.GraphPlan(ei.cuda(), n)
    xm_g, ewm_g, e_g = ops.EdgeMask.apply(dev_in[0], dev_in[1], dev_in[2], ew.cuda(), plan, rois)
    g = torch.autograd.grad(sum((o * c.cuda()).sum() for o, c in zip((xm_g, ewm_g, e_g), cots)), dev_in)
    for got, want, nm in zip((xm_g, ewm_g, e_g), (xm, ewm, e), ("xm", "ewm", "e")):
        assert_matches(got, want.detach().numpy(), TOL, nm)
    for got, want, nm in zip(g, g_ref, ("dx", "dprob", "dprob_bias")):
        assert_matches(got, want.numpy(), TOL, nm)


# ------------------------------------------------------------------------------------------------ GCN layer
@pytest.mark.parametrize("n,e,fin,fout,loops,multi,seed", [
    (12, 40, 3, 16, True, True, 0), (90, 270, 3, 16, False, False, 1), (200, 3000, 16, 16, True, False, 2),
    (64, 500, 5, 7, True, True, 3), (33, 0, 4, 4, False, False, 4), (512, 20000, 16, 32, True, False, 5)])
def test_gcn_layer_fwd_bwd(ops, n, e, fin, fout, loops, multi, seed):
    from oracle import pyg_ops
    rng = np.random.default_rng(seed)
    ei, ew = _rand_graph(rng, n, e, loops=loops, multi_loops=multi) if e else (
        torch.zeros(2, 0, dtype=torch.long), torch.zeros(0))
    x = torch.from_numpy(rng.standard_normal((n, fin))).float()
    w = torch.from_numpy(rng.standard_normal((fout, fin)) / np.sqrt(fin)).float()
    b = torch.from_numpy(rng.standard_normal(fout)).float()
    cot = torch.from_numpy(rng.standard_normal((n, fout))).float()
    ref_in = [t.double().requires_grad_(True) for t in (x, ew, w, b)]
    out_ref = torch.relu(pyg_ops.gcn_conv(ref_in[0], ei, ref_in[1], ref_in[2], ref_in[3]))
    g_ref = torch.autograd.grad((out_ref * cot.double()).sum(), ref_in, allow_unused=True)
    dev_in = [t.cuda().requires_grad_(True) for t in (x, ew, w, b)]
    plan = ops.GraphPlan(ei.cuda(), n)
    coef = ops.GcnNorm.apply(dev_in[1], plan)
    h = ops.linear(dev_in[0], dev_in[2])
    out = ops.GcnPropagate.apply(h, coef[0], coef[1], dev_in[3], plan, True, coef[2], coef[3])
    g = torch.autograd.grad((out * cot.cuda()).sum(), dev_in, allow_unused=True)
    assert_matches(out, out_ref.detach().numpy(), TOL, "out")
    for got, want, nm in zip(g, g_ref, ("dx", "dew", "dW", "db")):
        if want is None or (nm == "dew" and (e == 0 or multi)):
            continue      # several stored loops on one node: index_put's duplicate-index gradient is undefined
        assert_matches(got, want.numpy(), TOL, nm, floor=1e-6)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("m,n,k", [(23040, 16, 3), (23040, 16, 16), (256, 64, 2912), (256, 64, 3182), (256, 3, 64),
                                   (5, 7, 9), (1000, 33, 130), (64, 64, 4096),
                                   # short-K streaming kernel (M >= 4096, N, K multiples of 16 up to 64), ragged M
                                   (8197, 64, 32), (5003, 32, 64), (4100, 48, 48), (20480, 64, 64)])
def test_gemm_nt_nn_tn(ops, m, n, k):
    rng = np.random.default_rng(m + n + k)
    a = torch.from_numpy(rng.standard_normal((m, k))).float()
    b = torch.from_numpy(rng.standard_normal((n, k))).float()
    bias = torch.from_numpy(rng.standard_normal(n)).float()
    want = a.double() @ b.double().t() + bias.double()
    assert_matches(ops.gemm_nt(a.cuda(), b.cuda(), bias.cuda(), 0), want.numpy(), TOL, "NT")
    assert_matches(ops.gemm_nt(a.cuda(), b.cuda(), bias.cuda(), 1), torch.relu(want).numpy(), TOL, "NT+relu")
    dy = torch.from_numpy(rng.standard_normal((m, n))).float()
    assert_matches(ops.gemm_nn(dy.cuda(), b.cuda()), (dy.double() @ b.double()).numpy(), TOL, "NN")
    assert_matches(ops.gemm_tn(dy.cuda(), a.cuda()), (dy.double().t() @ a.double()).numpy(), TOL, "TN")


@pytest.mark.parametrize("m,n,k", [(23040, 16, 3), (32768, 16, 16), (256, 64, 2912), (64, 64, 16416), (83200, 64, 32),
                                   (100, 37, 50), (1000, 32, 64)])
def test_gemm_bf16_operands_fp32_accumulate(ops, m, n, k):
    """igcn_gemm_bf16 (BASELINE configs[4]): exact w.r.t. the fp64 product of the bf16-ROUNDED operands up to fp32
    accumulation order — and within the bf16 rounding bound of the unrounded product."""
    rng = np.random.default_rng(m + n + k)
    a = torch.from_numpy(rng.standard_normal((m, k))).float()
    b = torch.from_numpy(rng.standard_normal((n, k))).float()
    bias = torch.from_numpy(rng.standard_normal(n)).float()
    r = lambda t: t.bfloat16().double()                                       # noqa: E731  (round-to-nearest-even)
    want = r(a) @ r(b).t() + bias.double()
    got = ops.gemm_nt(a.cuda(), b.cuda(), bias.cuda(), 0, bf16=True)
    assert_matches(got, want.numpy(), 2e-5, "NT bf16")
    exact = a.double() @ b.double().t() + bias.double()
    assert_matches(got, exact.numpy(), 2.0 ** -7, "NT vs unrounded")
    assert_matches(ops.gemm_nt(a.cuda(), b.cuda(), bias.cuda(), 1, bf16=True), torch.relu(want).numpy(), 2e-5, "relu")
    dy = torch.from_numpy(rng.standard_normal((m, n))).float()
    assert_matches(ops.gemm_nn(dy.cuda(), b.cuda(), bf16=True), (r(dy) @ r(b)).numpy(), 2e-5, "NN bf16")
    assert_matches(ops.gemm_tn(dy.cuda(), a.cuda(), bf16=True), (r(dy).t() @ r(a)).numpy(), 2e-5, "TN bf16")


@pytest.mark.parametrize("rows,fin,fout,relu,bias", [(512, 64, 3, False, True), (512, 2912, 64, True, True),
                                                     (46080, 16, 32, False, True), (1000, 7, 12, True, True),
                                                     (300, 5, 64, True, False), (70000, 32, 64, False, True),
                                                     # narrow outputs (<= 4): the VALU kernels igcn_small_linear_*
                                                     (46080, 16, 3, False, True), (517, 64, 1, False, True),
                                                     (130, 256, 4, False, False), (9, 4, 2, False, True)])
def test_linear_fwd_bwd(ops, rows, fin, fout, relu, bias):
    """ops.linear = GEMM with fused bias/ReLU epilogue; backward = igcn_bias_grad (ReLU mask + bias gradient in one
    pass) + two GEMMs, against torch in fp64."""
    rng = np.random.default_rng(rows + fout)
    x = torch.from_numpy(rng.standard_normal((rows, fin))).float()
    w = torch.from_numpy(rng.standard_normal((fout, fin)) / np.sqrt(fin)).float()
    b = torch.from_numpy(rng.standard_normal(fout)).float() if bias else None
    cot = torch.from_numpy(rng.standard_normal((rows, fout))).float()
    ref_in = [t.double().requires_grad_(True) for t in (x, w)] + ([b.double().requires_grad_(True)] if bias else [])
    y_ref = torch.nn.functional.linear(ref_in[0], ref_in[1], ref_in[2] if bias else None)
    if relu:
        y_ref = torch.relu(y_ref)
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    dev = [t.cuda().requires_grad_(True) for t in (x, w)] + ([b.cuda().requires_grad_(True)] if bias else [])
    y = ops.linear(dev[0], dev[1], dev[2] if bias else None, relu=relu)
    g = torch.autograd.grad((y * cot.cuda()).sum(), dev)
    assert_matches(y, y_ref.detach().numpy(), TOL, "y")
    for got, want, nm in zip(g, g_ref, ("dx", "dW", "db")):
        assert_matches(got, want.numpy(), TOL, nm)


@pytest.mark.parametrize("bsz,g,w,l,rois,use_prob,used", [(6, 2, 96, 32, 8, True, (1, 1, 1)), (5, 1, 40, 6, 4, True, (0, 1, 1)),
                                                          (4, 2, 64, 32, 8, False, (1, 1, 0)), (256, 2, 2880, 32, 90, True, (1, 0, 1))])
def test_head_inputs(ops, bsz, g, w, l, rois, use_prob, used):
    """igcn_head_inputs_* (out_z, out_lin, feat in one launch) against the composite of kernel/sgcn_img_snp.py:284-297;
    ``used`` selects which outputs feed the loss (unused ones reach the backward as None)."""
    rng = np.random.default_rng(bsz + w)
    mk = lambda *shape: torch.from_numpy(rng.standard_normal(shape)).float()      # noqa: E731
    img, cross, latent, x, prob = mk(g * bsz, w), mk(g * bsz, w), mk(g * bsz, l), mk(bsz * rois, 3), mk(rois, 3)
    cots = [mk(g * bsz, w), mk(g * bsz, w + l), mk(g * bsz, w + l + (3 * rois if use_prob else 0))]

    def composite(img, cross, latent, x, prob):
        out_z = (img + cross) / 2
        out_lin = torch.cat((out_z, latent), -1)
        feat = out_lin
        if use_prob:
            img_feat = (x.view(bsz, rois, -1) * prob).reshape(bsz, -1)
            feat = torch.cat((out_lin, img_feat.repeat(g, 1)), -1)
        return out_z, out_lin, feat

    def total(outs, cots):
        return sum((o * c).sum() for o, c, u in zip(outs, cots, used) if u)

    ref_in = [t.double().requires_grad_(True) for t in (img, cross, latent, x, prob)]
    ref_out = composite(*ref_in)
    g_ref = torch.autograd.grad(total(ref_out, [c.double() for c in cots]), ref_in, allow_unused=True)
    dev = [t.cuda().requires_grad_(True) for t in (img, cross, latent, x, prob)]
    if use_prob:
        outs = ops.HeadInputs.apply(dev[0], dev[1], dev[2], dev[3].view(bsz, -1), dev[4].view(-1), bsz)
    else:
        oz, ol, _ = ops.HeadInputs.apply(dev[0], dev[1], dev[2], None, None, bsz)
        outs = (oz, ol, ol)
    for got, want, nm in zip(outs, ref_out, ("out_z", "out_lin", "feat")):
        assert_matches(got, want.detach().numpy(), TOL, nm)
    gd = torch.autograd.grad(total(outs, [c.cuda() for c in cots]), dev, allow_unused=True)
    for got, want, nm in zip(gd, g_ref, ("d_img", "d_cross", "d_latent", "dx", "dprob")):
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, nm
        else:
            assert_matches(got, want.numpy(), TOL, nm)


@pytest.mark.parametrize("bsz,g,rois,d,l,use_prob", [(8, 2, 90, 32, 64, True), (5, 1, 12, 16, 10, False),
                                                      (3, 2, 7, 2, 4, True), (16, 2, 90, 64, 2, False)])
def test_relu_owed_layer_and_head_inputs(ops, bsz, g, rois, d, l, use_prob):
    """relu(out_proj(.)) whose ReLU backward and bias gradient ride in the head-input backward (ops.LinearReluOwed +
    HeadInputs(cross_bias=...), igcn_head_inputs_bwd_relu) against the composite of kernel/sgcn_img_snp.py:241-242,
    :284-297 in fp64: outputs and the gradients of every input, weight and bias."""
    rng = np.random.default_rng(bsz * 7 + d)
    mk = lambda *shape: torch.from_numpy(rng.standard_normal(shape)).float()      # noqa: E731
    w = rois * d
    o, wt, bias = mk(g * bsz * rois, d), mk(d, d) * 0.3, mk(d) * 0.2
    img, latent, x, prob = mk(g * bsz, w), mk(g * bsz, l), mk(bsz * rois, 2), mk(rois, 2)
    cots = [mk(g * bsz, w), mk(g * bsz, w + l), mk(g * bsz, w + l + (2 * rois if use_prob else 0))]

    def composite(o, wt, bias, img, latent, x, prob):
        cross = torch.relu(o @ wt.t() + bias).reshape(g * bsz, -1)
        out_z = (img + cross) / 2
        out_lin = torch.cat((out_z, latent), -1)
        feat = out_lin
        if use_prob:
            feat = torch.cat((out_lin, (x.view(bsz, rois, -1) * prob).reshape(bsz, -1).repeat(g, 1)), -1)
        return out_z, out_lin, feat

    ref_in = [t.double().requires_grad_(True) for t in (o, wt, bias, img, latent, x, prob)]
    ref_out = composite(*ref_in)
    g_ref = torch.autograd.grad(sum((a * c.double()).sum() for a, c in zip(ref_out, cots)), ref_in, allow_unused=True)
    dev = [t.cuda().requires_grad_(True) for t in (o, wt, bias, img, latent, x, prob)]
    assert ops.relu_owed_supported(d, w)
    cross = ops.LinearReluOwed.apply(dev[0], dev[1], dev[2], False).reshape(g * bsz, -1)
    if use_prob:
        outs = ops.HeadInputs.apply(dev[3], cross, dev[4], dev[5].view(bsz, -1), dev[6].view(-1), bsz, dev[2])
    else:
        oz, ol, _ = ops.HeadInputs.apply(dev[3], cross, dev[4], None, None, bsz, dev[2])
        outs = (oz, ol, ol)
    for got, want, nm in zip(outs, ref_out, ("out_z", "out_lin", "feat")):
        assert_matches(got, want.detach().numpy(), TOL, nm)
    gd = torch.autograd.grad(sum((a * c.cuda()).sum() for a, c in zip(outs, cots)), dev, allow_unused=True)
    for got, want, nm in zip(gd, g_ref, ("d_o", "d_weight", "d_bias", "d_img", "d_latent", "dx", "dprob")):
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, nm
        else:
            assert_matches(got, want.numpy(), TOL, nm)
    # ... and with the layer itself inside the head-input launch (ops.OutProjHeadInputs: igcn_outproj_head_inputs_fwd)
    if ops.outproj_head_inputs_supported(d, w):
        dev3 = [t.cuda().requires_grad_(True) for t in (o, wt, bias, img, latent, x, prob)]
        o3 = ops.OutProjHeadInputs.apply(dev3[0].view(g * bsz, -1), dev3[1], dev3[2], dev3[3], dev3[4],
                                         dev3[5].view(bsz, -1) if use_prob else None,
                                         dev3[6].view(-1) if use_prob else None, bsz, False)
        cross3 = o3[3]
        o3 = o3[:3] if use_prob else (o3[0], o3[1], o3[1])
        for got, want, nm in zip(o3, ref_out, ("out_z", "out_lin", "feat")):
            assert_matches(got, want.detach().numpy(), TOL, "fused " + nm)
        assert_matches(cross3, torch.relu(ref_in[0] @ ref_in[1].t() + ref_in[2]).reshape(g * bsz, -1).detach().numpy(), TOL,
                       "fused cross")
        g3 = torch.autograd.grad(sum((a * c.cuda()).sum() for a, c in zip(o3, cots)), dev3, allow_unused=True)
        for got, want, nm in zip(g3, g_ref, ("d_o", "d_weight", "d_bias", "d_img", "d_latent", "dx", "dprob")):
            if want is None:
                assert got is None or float(got.abs().max()) == 0.0, nm
            else:
                assert_matches(got, want.numpy(), TOL, "fused " + nm)
    # the same through the deferred reductions of a captured step (the bias sums join the flush), bit for bit
    with ops.deferred_reductions():
        cross = ops.LinearReluOwed.apply(dev[0], dev[1], dev[2], False).reshape(g * bsz, -1)
        o2 = ops.HeadInputs.apply(dev[3], cross, dev[4], dev[5].view(bsz, -1) if use_prob else None,
                                  dev[6].view(-1) if use_prob else None, bsz, dev[2])
        o2 = o2 if use_prob else (o2[0], o2[1], o2[1])
        g2 = torch.autograd.grad(sum((a * c.cuda()).sum() for a, c in zip(o2, cots)), dev, allow_unused=True)
    torch.cuda.synchronize()
    for a, b_, nm in zip(gd, g2, ("d_o", "d_weight", "d_bias")):
        assert torch.equal(a, b_), nm


@pytest.mark.parametrize("b,lq,lk,d", [(6, 90, 40, 32), (3, 7, 5, 16), (5, 90, 130, 48)])
def test_in_proj_packed_projection(ops, b, lq, lk, d):
    """ops.InProj = the packed q / key|value projection of nn.MultiheadAttention (cross-attention) with the parameters
    taken whole: outputs and all four gradients against F.linear on the slices in fp64; under deferred reductions the
    parameters are leaves of the op and the gradients are bit-identical to the immediate ones."""
    rng = np.random.default_rng(b * 100 + d)
    mk = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))     # noqa: E731
    query, memory, w, bias = mk(b, lq, d), mk(b, lk, d), mk(3 * d, d), mk(3 * d)
    gq, gkv = mk(b, lq, d), mk(b, lk, 2 * d)
    ref = [t.double().requires_grad_(True) for t in (query, memory, w, bias)]
    rq = torch.nn.functional.linear(ref[0], ref[2][:d], ref[3][:d])
    rkv = torch.nn.functional.linear(ref[1], ref[2][d:], ref[3][d:])
    rg = torch.autograd.grad([rq, rkv], ref, [gq.double(), gkv.double()])
    grads = []
    for defer in (False, True):
        dev = [t.cuda().requires_grad_(True) for t in (query, memory, w, bias)]
        q, kv = ops.InProj.apply(*dev)
        if defer:
            with ops.deferred_reductions():
                g = torch.autograd.grad([q, kv], dev, [gq.cuda(), gkv.cuda()])
        else:
            g = torch.autograd.grad([q, kv], dev, [gq.cuda(), gkv.cuda()])
        torch.cuda.synchronize()
        assert_matches(q, rq.detach().numpy(), TOL, "q")
        assert_matches(kv, rkv.detach().numpy(), TOL, "kv")
        for got, want, nm in zip(g, rg, ("dquery", "dmemory", "dW", "dbias")):
            assert_matches(got, want.numpy(), TOL, nm)
        grads.append([t.cpu() for t in g])
    for a, c in zip(*grads):
        assert torch.equal(a, c)


@pytest.mark.parametrize("b,lq,lk,d,h", [(5, 90, 40, 32, 2), (3, 17, 70, 32, 4), (4, 90, 130, 48, 2)])
def test_projected_attention_matches_multihead_attention(ops, b, lq, lk, d, h):
    """ops.ProjectedAttention (packed in-projection + attention core as one autograd node) against
    nn.MultiheadAttention's in-projection + scaled-dot-product attention in fp64: output and all four gradients —
    including the closed forms d b_k = 0 and d b_v = sum d o, which the reference reaches through autograd."""
    rng = np.random.default_rng(b + lq + lk)
    mk = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))     # noqa: E731
    query, memory, w, bias, cot = mk(b, lq, d), mk(b, lk, d), mk(3 * d, d) * 0.3, mk(3 * d), mk(b, lq, d)
    ref = [t.double().requires_grad_(True) for t in (query, memory, w, bias)]
    q = torch.nn.functional.linear(ref[0], ref[2][:d], ref[3][:d]).view(b, lq, h, d // h).transpose(1, 2)
    k = torch.nn.functional.linear(ref[1], ref[2][d:2 * d], ref[3][d:2 * d]).view(b, lk, h, d // h).transpose(1, 2)
    v = torch.nn.functional.linear(ref[1], ref[2][2 * d:], ref[3][2 * d:]).view(b, lk, h, d // h).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(2, 3) / (d // h) ** 0.5, dim=-1)
    o_ref = (att @ v).transpose(1, 2).reshape(b, lq, d)
    g_ref = torch.autograd.grad((o_ref * cot.double()).sum(), ref)
    assert float(g_ref[3][d:2 * d].abs().max()) < 1e-12 * max(1.0, float(g_ref[3].abs().max()))   # d b_k: noise around 0
    dev = [t.cuda().requires_grad_(True) for t in (query, memory, w, bias)]
    o = ops.ProjectedAttention.apply(*dev, h)
    g = torch.autograd.grad((o * cot.cuda()).sum(), dev)
    assert_matches(o, o_ref.detach().numpy(), TOL, "o")
    for got, want, nm in zip(g, g_ref, ("dquery", "dmemory", "dW", "dbias")):
        assert_matches(got, want.numpy(), TOL, nm)
    assert float(g[3][d:2 * d].abs().max()) == 0.0


@pytest.mark.parametrize("m,n,k", [(64, 32, 32), (1000, 64, 32), (70001, 64, 32), (4613, 32, 32), (64, 48, 48),
                                   (70001, 96, 48), (4613, 48, 48), (1000, 96, 48)])
def test_proj_bwd_one_pass(ops, m, n, k):
    """igcn_proj_bwd: dX = G W and dW = G^T X of a bias-free projection from ONE pass over G (K = 32 or 48, N = K or 2 K;
    ragged last row block, more row blocks than workgroups) against fp64."""
    from igcn_amd._lib import call, load, ptr, stream_ptr
    rng = np.random.default_rng(m + n)
    mk = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))     # noqa: E731
    g, x, w = mk(m, n), mk(m, k), mk(n, k)
    lib = load()
    assert lib.igcn_proj_bwd_supported(m, n, k) and not lib.igcn_proj_bwd_supported(m, n, 16)
    assert not lib.igcn_proj_bwd_supported(m, 3 * k, k) and not lib.igcn_proj_bwd_supported(m, 64, 48)
    gd, xd, wd = g.cuda(), x.cuda(), w.cuda()
    dx = torch.empty(m, k, device="cuda")
    dw = torch.empty(n, k, device="cuda")
    scr = torch.empty(int(lib.igcn_proj_bwd_scratch_floats(m, n)), device="cuda")
    call("igcn_proj_bwd", m, n, k, ptr(gd), ptr(xd), ptr(wd), ptr(dx), ptr(dw), ptr(scr), stream_ptr())
    assert_matches(dx, (g.double() @ w.double()).numpy(), TOL, "dX")
    assert_matches(dw, (g.double().t() @ x.double()).numpy(), TOL, "dW")


@pytest.mark.parametrize("kd", [32, 48])
@pytest.mark.parametrize("m1,m2", [(23040, 102400), (90, 400), (1, 1), (64 * 2100 + 7, 130)])
def test_proj_fwd_streaming_pair(ops, m1, m2, kd):
    """igcn_proj_fwd_pair: y1 = x1 W1^T + b1 (32 columns) and y2 = x2 W2^T + b2 (64 columns) in one streaming launch
    (K = 32): the bench shape, single tiles, ragged last tiles, more tiles than workgroups; small-integer operands are
    reproduced EXACTLY (fp32 products, k-ordered accumulation), random ones match fp64; and the second projection may be
    absent."""
    from igcn_amd._lib import call, load, ptr, stream_ptr
    rng = np.random.default_rng(m1 + m2)
    lib = load()
    assert int(lib.igcn_proj_fwd_blocks(64 * 5000)) < 5000        # a workgroup then walks several tiles
    for exact in (True, False):
        mk = (lambda *sh: torch.from_numpy(rng.integers(-8, 9, sh).astype(np.float32))) if exact else \
            (lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32)))
        x1, w1, b1, x2, w2, b2 = mk(m1, kd), mk(kd, kd), mk(kd), mk(m2, kd), mk(2 * kd, kd), mk(2 * kd)
        dev = [t.cuda() for t in (x1, w1, b1, x2, w2, b2)]
        y1 = torch.full((m1, kd), float("nan"), device="cuda")
        y2 = torch.full((m2, 2 * kd), float("nan"), device="cuda")
        call("igcn_proj_fwd_pair", m1, kd, ptr(dev[0]), ptr(dev[1]), ptr(dev[2]), ptr(y1), m2, 2 * kd, ptr(dev[3]), ptr(dev[4]),
             ptr(dev[5]), ptr(y2), kd, stream_ptr())
        r1 = x1.double() @ w1.double().t() + b1.double()
        r2 = x2.double() @ w2.double().t() + b2.double()
        if exact:
            assert torch.equal(y1.cpu().double(), r1) and torch.equal(y2.cpu().double(), r2)
        else:
            assert_matches(y1, r1.numpy(), TOL, "y1")
            assert_matches(y2, r2.numpy(), TOL, "y2")
    y1.fill_(float("nan"))
    call("igcn_proj_fwd_pair", m1, kd, ptr(dev[0]), ptr(dev[1]), None, ptr(y1), 0, 0, None, None, None, None, kd, stream_ptr())
    assert_matches(y1, (x1.double() @ w1.double().t()).numpy(), TOL, "y1 alone, no bias")


def test_sparse_map_reads_channel_vectors_at_a_stride(ops, monkeypatch):
    """ops.SparseMap with per-channel value vectors that are views at a constant stride of one buffer (how FlatAdam lays
    the ParameterList of go_model.py:67 out): the kernels read them in place (igcn_spmm_{fwd,bwd}_strided) — same bits
    as the stacked [C, nnz] tensor, forward and every gradient, and no torch.stack."""
    monkeypatch.setenv("IGCN_SPARSE_MAPS", "1")
    a_g, a, pool_dim, idx = _hier((300, 120, 60, 19, 1), 1)
    agc = a_g.coalesce()
    gi = agc.indices()
    csr = ops.Csr(gi[0], gi[1], a.shape[0], 54, "cuda")
    nnz = csr.nnz
    stride = (nnz + 15) // 16 * 16
    rng = np.random.default_rng(0)
    flat = torch.from_numpy(rng.standard_normal(2 * stride + 7).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.standard_normal((24, 54)).astype(np.float32)).cuda()
    cot = torch.from_numpy(rng.standard_normal((24, 2, a.shape[0])).astype(np.float32)).cuda()

    def run(strided):
        xs = x.clone().requires_grad_(True)
        if strided:
            vals = [flat[c * stride:c * stride + nnz].detach().requires_grad_(True) for c in range(2)]
            # leaves that ALIAS the flat buffer at the stride: what FlatAdam's `p.data = flat[...]` produces
            assert ops.SparseMap._row_stride(vals, nnz) == stride
        else:                                                # the stacked form (any two vectors with ascending
            monkeypatch.setattr(ops.SparseMap, "_row_stride", staticmethod(lambda vals, nnz: None))   # addresses would otherwise be read in place too)
            vals = [flat[c * stride:c * stride + nnz].clone().requires_grad_(True) for c in range(2)]
        y = ops.SparseMap.apply(xs, csr, *vals)
        return (y,) + torch.autograd.grad((y * cot).sum(), [xs] + vals)

    calls = []
    real = torch.stack
    monkeypatch.setattr(torch, "stack", lambda *a_, **k: (calls.append(1), real(*a_, **k))[1])
    got = run(True)
    assert not calls
    want = run(False)
    assert calls
    for g, w_, nm in zip(got, want, ("y", "dx", "dval0", "dval1")):
        assert torch.equal(g, w_), nm


def test_gemm_is_exact_fp32_fma_chain(ops):
    """MFMA f32 = k-ordered fmaf chain: small-integer operands must be reproduced exactly."""
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.integers(-8, 9, (130, 70)).astype(np.float32))
    b = torch.from_numpy(rng.integers(-8, 9, (45, 70)).astype(np.float32))   # asymmetric B catches a swapped C map
    got = ops.gemm_nt(a.cuda(), b.cuda())
    assert torch.equal(got.cpu(), a @ b.t())


def test_final_reductions_deferred_and_immediate_give_the_same_bits(ops):
    """Every form of the partial-row sums (plan.hip rr_form: in-order thread per column, 16-byte wide, wave per column,
    16 x 16 tiles, four row groups) — launched on its own and as one entry of the deferred launch (k_multi_reduce) —
    against an fp64 sum, and bit-identical between the two ways."""
    import ctypes
    from igcn_amd import _lib
    lib = _lib.load()
    fn = lib.igcn_debug_reduce_rows_final
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    # (rows, n, ld): few rows narrow / wide (16-byte) / wide unaligned; tall with < 16 columns; tall tiles; tall and wide
    # below and above 256 rows; rows = 0 (writes zeros); a padded leading dimension
    shapes = [(8, 64, 64), (16, 4096, 4096), (4, 1030, 1031), (448, 5, 5), (3008, 2, 2), (800, 160, 160),
              (40, 2400, 2400), (128, 8067, 8067), (300, 5000, 5000), (512, 1208, 2416), (0, 100, 100), (33, 17, 20)]
    torch.manual_seed(3)
    st = torch.cuda.current_stream().cuda_stream
    bufs = [(torch.randn(max(r, 1), ld, device="cuda"), torch.full((n,), 7.0, device="cuda"),
             torch.full((n,), 9.0, device="cuda")) for r, n, ld in shapes]
    for (p, o1, _), (r, n, ld) in zip(bufs, shapes):                    # immediate
        assert fn(p.data_ptr(), r, ld, n, o1.data_ptr(), st) == 0
    lib.igcn_reduce_defer(st, 1)
    try:
        for (p, _, o2), (r, n, ld) in zip(bufs, shapes):                # queued, then ONE launch
            assert fn(p.data_ptr(), r, ld, n, o2.data_ptr(), st) == 0
        assert lib.igcn_reduce_pending() == len(shapes)
        assert lib.igcn_reduce_flush(st) == 0
    finally:
        lib.igcn_reduce_defer(st, 0)
    torch.cuda.synchronize()
    for (p, o1, o2), (r, n, ld) in zip(bufs, shapes):
        want = p[:r, :n].double().sum(0).cpu().numpy() if r else np.zeros(n)
        assert_matches(o1, want, 2e-6, f"immediate {r} x {n}", floor=float(max(r, 1)) ** 0.5)
        assert torch.equal(o1, o2), f"deferred != immediate at {r} x {n} (ld {ld})"


def test_deferred_reductions_are_kept_per_stream(ops):
    """Two streams in defer mode at once (two trainers of one process): each flush performs its OWN stream's entries
    only, a stream that does not defer reduces at once, and the Python-side state (kept buffers, queued passes) is per
    stream as well."""
    import ctypes
    from igcn_amd import _lib
    lib = _lib.load()
    fn = lib.igcn_debug_reduce_rows_final
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    s1, s2, s3 = (torch.cuda.Stream() for _ in range(3))
    torch.manual_seed(5)
    p = [torch.randn(40, 300, device="cuda") for _ in range(3)]
    o = [torch.full((300,), 7.0, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for st in (s1, s2):
        assert lib.igcn_reduce_defer(st.cuda_stream, 1) == 0
    try:
        for k, st in enumerate((s1, s2, s3)):
            assert fn(p[k].data_ptr(), 40, 300, 300, o[k].data_ptr(), st.cuda_stream) == 0
        assert lib.igcn_reduce_pending() == 2                           # s3 does not defer: reduced at once
        s3.synchronize()
        assert torch.allclose(o[2], p[2].sum(0), atol=1e-4)
        assert lib.igcn_reduce_flush(s1.cuda_stream) == 0               # s1's entry only
        torch.cuda.synchronize()
        assert torch.allclose(o[0], p[0].sum(0), atol=1e-4) and float(o[1][0]) == 7.0
        assert lib.igcn_reduce_pending() == 1
        assert lib.igcn_reduce_flush(s2.cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.allclose(o[1], p[1].sum(0), atol=1e-4) and lib.igcn_reduce_pending() == 0
    finally:
        for st in (s1, s2):
            lib.igcn_reduce_defer(st.cuda_stream, 0)
            lib.igcn_reduce_flush(st.cuda_stream)
    with torch.cuda.stream(s1):
        with ops.deferred_reductions():
            assert ops._DEFER["on"]
            with torch.cuda.stream(s2):
                assert not ops._DEFER["on"]                             # another stream: its own state
        assert not ops._DEFER["on"]


# ------------------------------------------------------------------------------------------------ GO ops
def _hier(pool, seed):
    from igcn_amd import synth
    from oracle import go_network as OG
    go_snps, adj, pool_dim = synth.go_hierarchy(pool, seed=seed)
    a_g, a = synth.go_sparse_inputs(go_snps, adj)
    return a_g, a, pool_dim, OG.go_index_sets(a_g, a, pool, 2)


@pytest.mark.parametrize("bsz,pool,fin,seed", [(4, (20, 10, 6, 3, 1), 2, 0), (16, (300, 120, 60, 19, 1), 2, 1),
                                               (5, (40, 20, 9, 2, 1), 5, 2),
                                               # the bench hierarchy: 3000 / 1200 nodes = 3 / 2 passes of the
                                               # LDS-resident backward's 1024 threads, and the 99-reader hub
                                               (3, (1800, 800, 300, 99, 1), 2, 3), (3, (1800, 800, 300, 99, 1), 5, 4),
                                               (2, (1801, 800, 300, 99, 1), 2, 5)])
def test_go_attention_layer(ops, bsz, pool, fin, seed):
    _, _, _, idx = _hier(pool, seed)
    layer = 0 if fin == 2 else 1
    row, col, nj = idx["enc"][layer]
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((bsz, nj, fin))).float()
    par = [torch.from_numpy(rng.standard_normal(s) * 0.5).float() for s in ((5, fin), (5, fin), (1, 10), (1, 5))]
    cot = torch.from_numpy(rng.standard_normal((bsz, nj, 5))).float()

    def ref(xd, w_inc, w_s, a_in, a_s):
        x_in, x_s = xd @ w_inc.t(), xd @ w_s.t()
        v = torch.exp(torch.tanh(torch.cat([x_in[:, row], x_in[:, col]], 2) @ a_in.t())).squeeze(2)
        z = torch.zeros(bsz, nj, dtype=xd.dtype).index_add(1, row, v)
        alpha = v / z[:, row]
        return torch.zeros(bsz, nj, 5, dtype=xd.dtype).index_add(1, row, alpha.unsqueeze(2) * x_in[:, col]) \
            + x_s * torch.sigmoid(x_s @ a_s.t())

    ref_in = [t.double().requires_grad_(True) for t in [x] + par]
    y_ref = ref(*ref_in)
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    csr = ops.Csr(row, col, nj, nj, "cuda")
    dev = [x.transpose(1, 2).contiguous().cuda().requires_grad_(True)] + [p.cuda().requires_grad_(True) for p in par]
    y = ops.GoAttention.apply(dev[0], dev[1], dev[2], dev[3].view(-1), dev[4].view(-1), csr)
    g = torch.autograd.grad((y * cot.transpose(1, 2).contiguous().cuda()).sum(), dev, retain_graph=True)
    assert_matches(y.transpose(1, 2), y_ref.detach().numpy(), TOL, "y")
    assert_matches(g[0].transpose(1, 2), g_ref[0].numpy(), TOL, "dx")
    for got, want, nm in zip(g[1:], g_ref[1:], ("dW_inc", "dW_s", "da_in", "da_s")):
        assert_matches(got, want.numpy(), TOL, nm)
    # the balanced thread -> node map of the column walks (igcn_go_attn_walk_order) only moves work between waves: in
    # plain node order the per-node results are the same bits, the parameter sums the same up to summation order
    wo = csr.walk_order(fin, 5)
    assert wo is not None and int((wo >= 0).sum()) == nj
    ident = torch.arange(wo.numel(), dtype=torch.int32, device="cuda")
    ident[nj:] = -1
    csr._walk_order[(fin, 5)] = ident
    g2 = torch.autograd.grad((y * cot.transpose(1, 2).contiguous().cuda()).sum(), dev)
    assert torch.equal(g2[0], g[0])
    for a, c2, nm in zip(g2[1:], g[1:], ("dW_inc", "dW_s", "da_in", "da_s")):
        assert_matches(a, c2.cpu().numpy(), 1e-5, nm)


@pytest.mark.parametrize("bsz,f,n,pool,with_keep", [(3, 5, 40, 20, False), (8, 5, 3000, 1800, True),
                                                     (4, 2, 257, 0, True), (2, 5, 1200, 800, False),
                                                     # rows beyond 4096 nodes: the 1024-thread form of the 16-byte
                                                     # kernels (configs[4]: 10 000 GO nodes), and beyond 16384: scalar
                                                     (2, 5, 10000, 6000, True), (2, 2, 4100, 0, False),
                                                     (1, 2, 16400, 400, False),
                                                     # many rows (the train step's launches have 1024-2560)
                                                     (205, 5, 400, 200, True), (512, 2, 1200, 0, False)])
def test_nodes_layernorm(ops, bsz, f, n, pool, with_keep):
    rng = np.random.default_rng(n)
    y = torch.from_numpy(rng.standard_normal((bsz, f, n)) * 2 + 0.3).float()
    gamma = torch.from_numpy(1 + 0.2 * rng.standard_normal(n)).float()
    beta = torch.from_numpy(0.1 * rng.standard_normal(n)).float()
    keep = torch.from_numpy((rng.random((bsz, n)) > 0.4) / 0.6).float() if with_keep else None
    cot = torch.from_numpy(rng.standard_normal((bsz, f, n - pool))).float()
    ref_in = [t.double().requires_grad_(True) for t in (y, gamma, beta)]
    z_ref = torch.relu(torch.nn.functional.layer_norm(ref_in[0], (n,), ref_in[1], ref_in[2], 1e-5))
    if keep is not None:
        z_ref = z_ref * keep.double().unsqueeze(1)
    z_ref = z_ref[:, :, pool:]
    g_ref = torch.autograd.grad((z_ref * cot.double()).sum(), ref_in)
    dev = [t.cuda().requires_grad_(True) for t in (y, gamma, beta)]
    z = ops.NodesLayerNorm.apply(dev[0], dev[1], dev[2], keep.cuda() if keep is not None else None, pool, 1e-5)
    g = torch.autograd.grad((z * cot.cuda()).sum(), dev)
    assert_matches(z, z_ref.detach().numpy(), TOL, "z")
    for got, want, nm in zip(g, g_ref, ("dy", "dgamma", "dbeta")):
        assert_matches(got, want.numpy(), TOL, nm)


@pytest.mark.parametrize("bsz,pool,layer,seed", [(4, (20, 10, 6, 3, 1), 0, 0), (4, (20, 10, 6, 3, 1), 1, 1),
                                                 (9, (300, 120, 60, 19, 1), 0, 2), (9, (300, 120, 60, 19, 1), 1, 3),
                                                 (3, (1800, 800, 300, 99, 1), 0, 4), (3, (1800, 800, 300, 99, 1), 1, 5),
                                                 (2, (1801, 800, 300, 99, 1), 1, 6),
                                                 # >= 128 samples: the one-workgroup-per-sample backward (below that the
                                                 # thread-per-node kernel runs, unless the LayerNorm backward rides along)
                                                 (130, (300, 120, 60, 19, 1), 0, 7), (130, (300, 120, 60, 19, 1), 1, 8)])
def test_go_decoder_layer(ops, bsz, pool, layer, seed):
    _, _, _, idx = _hier(pool, seed)
    row, col, n_rows, n_cols = idx["dec"][layer]
    fin, fout = 5, (5 if layer == 0 else 2)
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((bsz, n_cols, fin))).float()
    par = [torch.from_numpy(rng.standard_normal((fout, fin)) * 0.5).float() for _ in range(2)]
    cot = torch.from_numpy(rng.standard_normal((bsz, n_rows, fout))).float()

    def ref(xd, w_out, w_sout):
        deg = torch.zeros(n_rows, dtype=xd.dtype).index_add(0, row, torch.ones(row.numel(), dtype=xd.dtype))
        agg = torch.zeros(bsz, n_rows, fout, dtype=xd.dtype).index_add(
            1, row, (xd @ w_out.t())[:, col] / deg[row].view(1, -1, 1))
        pad = torch.zeros_like(agg)
        pad[:, n_rows - n_cols:] = xd @ w_sout.t()
        return agg + pad

    ref_in = [t.double().requires_grad_(True) for t in [x] + par]
    y_ref = ref(*ref_in)
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    csr = ops.Csr(row, col, n_rows, n_cols, "cuda")
    dev = [x.transpose(1, 2).contiguous().cuda().requires_grad_(True)] + [p.cuda().requires_grad_(True) for p in par]
    y = ops.GoDecode.apply(dev[0], dev[1], dev[2], csr)
    g = torch.autograd.grad((y * cot.transpose(1, 2).contiguous().cuda()).sum(), dev)
    assert_matches(y.transpose(1, 2), y_ref.detach().numpy(), TOL, "y")
    assert_matches(g[0].transpose(1, 2), g_ref[0].numpy(), TOL, "dx")
    assert_matches(g[1], g_ref[1].numpy(), TOL, "dW_out")
    assert_matches(g[2], g_ref[2].numpy(), TOL, "dW_sout")


def _ln_block_ref(y, gamma, beta, keep, pool):
    """fp64 dropout(relu(LayerNorm_over_nodes(y)))[:, pool:] for y [B, N, f] (go_model.py:246-251)."""
    z = torch.relu(torch.nn.functional.layer_norm(y.transpose(1, 2), (y.shape[1],), gamma, beta, 1e-5))
    if keep is not None:
        z = z * keep.double().unsqueeze(1)
    return z[:, :, pool:]


@pytest.mark.parametrize("bsz,pool,layer,with_keep,seed", [
    (4, (20, 10, 6, 3, 1), 0, True, 0), (5, (40, 20, 8, 3, 1), 1, False, 1), (16, (300, 120, 60, 19, 1), 0, True, 2),
    # the bench hierarchy: 3000 nodes on 1024 threads (layer 0), 1200 on 512 (layer 1)
    (3, (1800, 800, 300, 99, 1), 0, True, 3), (3, (1800, 800, 300, 99, 1), 1, True, 4),
    # 3001 nodes: no 16-byte rows, the op falls back to the two backward passes
    (2, (1801, 800, 300, 99, 1), 0, True, 5)])
def test_go_attention_with_layernorm_backward_in_one_launch(ops, monkeypatch, bsz, pool, layer, with_keep, seed):
    """ops.GoAttentionLN: encoder layer + LayerNorm block.  Its backward forms the LayerNorm's input gradient inside the
    LDS-resident attention backward (igcn_go_attn_ln_bwd) — against fp64 autograd of the composed reference, and
    against the op's own two-pass path (IGCN_NO_LN_FUSED)."""
    from igcn_amd import _lib
    _, _, _, idx = _hier(pool, seed)
    fin = 2 if layer == 0 else 5
    row, col, nj = idx["enc"][layer]
    drop = pool[layer]
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((bsz, nj, fin))).float()
    par = [torch.from_numpy(rng.standard_normal(s) * 0.5).float() for s in ((5, fin), (5, fin), (1, 10), (1, 5))]
    gamma = torch.from_numpy(1 + 0.2 * rng.standard_normal(nj)).float()
    beta = torch.from_numpy(0.1 * rng.standard_normal(nj)).float()
    keep = torch.from_numpy((rng.random((bsz, nj)) > 0.3) / 0.7).float() if with_keep else None
    cot = torch.from_numpy(rng.standard_normal((bsz, 5, nj - drop))).float()

    def ref(xd, w_inc, w_s, a_in, a_s, g_, b_):
        x_in, x_s = xd @ w_inc.t(), xd @ w_s.t()
        v = torch.exp(torch.tanh(torch.cat([x_in[:, row], x_in[:, col]], 2) @ a_in.t())).squeeze(2)
        z = torch.zeros(bsz, nj, dtype=xd.dtype).index_add(1, row, v)
        alpha = v / z[:, row]
        y = torch.zeros(bsz, nj, 5, dtype=xd.dtype).index_add(1, row, alpha.unsqueeze(2) * x_in[:, col]) \
            + x_s * torch.sigmoid(x_s @ a_s.t())
        return _ln_block_ref(y, g_, b_, keep, drop)

    ref_in = [t.double().requires_grad_(True) for t in [x] + par + [gamma, beta]]
    z_ref = ref(*ref_in)
    g_ref = torch.autograd.grad((z_ref * cot.double()).sum(), ref_in)
    csr = ops.Csr(row, col, nj, nj, "cuda")
    dev = [x.transpose(1, 2).contiguous().cuda().requires_grad_(True)] + \
          [p.cuda().requires_grad_(True) for p in par + [gamma, beta]]
    kd = keep.cuda() if keep is not None else None
    fused = bool(_lib.load().igcn_go_attn_ln_fused_ok(nj, fin, 5, drop))
    assert fused == (nj % 4 == 0 and drop % 4 == 0)
    z = ops.GoAttentionLN.apply(dev[0], dev[1], dev[2], dev[3], dev[4], csr, dev[5], dev[6], kd, drop, 1e-5)
    g = torch.autograd.grad((z * cot.cuda()).sum(), dev, retain_graph=True)
    names = ("dx", "dW_inc", "dW_s", "da_in", "da_s", "dgamma", "dbeta")
    assert_matches(z, z_ref.detach().numpy(), TOL, "z")
    assert_matches(g[0].transpose(1, 2), g_ref[0].numpy(), TOL, "dx")
    for got, want, nm in zip(g[1:], g_ref[1:], names[1:]):
        assert_matches(got, want.numpy(), TOL, nm)
    monkeypatch.setenv("IGCN_NO_LN_FUSED", "1")
    g2 = torch.autograd.grad((z * cot.cuda()).sum(), dev, retain_graph=True)
    for a, c2, nm in zip(g2, g, names):
        assert_matches(a, c2.cpu().numpy(), 2e-5, nm + " (two passes)")
    monkeypatch.delenv("IGCN_NO_LN_FUSED")
    # three consumers of z (fan = 3): their gradients are added while the backward loads them — the same bits as one
    # consumer holding their (in-order) sum
    z3 = ops.GoAttentionLN.apply(dev[0], dev[1], dev[2], dev[3], dev[4], csr, dev[5], dev[6], kd, drop, 1e-5, 3)
    assert isinstance(z3, tuple) and len(z3) == 3 and all(t.data_ptr() == z3[0].data_ptr() for t in z3)
    cots = [torch.from_numpy(rng.standard_normal(tuple(cot.shape))).float().cuda() for _ in range(3)]
    g3 = torch.autograd.grad(sum((t * c).sum() for t, c in zip(z3, cots)), dev, retain_graph=True)
    g1 = torch.autograd.grad((z * ((cots[0] + cots[1]) + cots[2])).sum(), dev, retain_graph=True)
    for a, c2, nm in zip(g3, g1, names):
        assert torch.equal(a, c2), nm + " (three consumers)"
    g2c = torch.autograd.grad((z3[0] * cots[0]).sum() + (z3[2] * cots[2]).sum(), dev)      # one consumer unused
    g1c = torch.autograd.grad((z * (cots[0] + cots[2])).sum(), dev)
    for a, c2, nm in zip(g2c, g1c, names):
        assert torch.equal(a, c2), nm + " (two of three consumers)"


def test_go_layer_with_layernorm_entry_points_refuse_what_they_do_not_cover(ops):
    """igcn_go_attn_ln_bwd / igcn_go_decode_ln_bwd run only where igcn_go_*_ln_fused_ok says so: other sizes, a missing
    operand or an unaligned one come back as an error (IgcnError), never as a silent different path."""
    from igcn_amd import _lib
    from igcn_amd._lib import call, ptr, stream_ptr
    lib = _lib.load()
    _, _, _, idx = _hier((20, 10, 6, 3, 1), 0)
    row, col, nj = idx["enc"][0]
    csr = ops.Csr(row, col, nj, nj, "cuda")
    b, fin, fout, pool = 2, 2, 5, 20
    f32 = dict(dtype=torch.float32, device="cuda")
    x, y = torch.randn(b, fin, nj, **f32), torch.randn(b, fout, nj, **f32)
    w = [torch.randn(fout, fin, **f32), torch.randn(fout, fin, **f32), torch.randn(2 * fout, **f32), torch.randn(fout, **f32)]
    gamma, beta = torch.ones(nj, **f32), torch.zeros(nj, **f32)
    mean, rstd = torch.zeros(b * fout, **f32), torch.ones(b * fout, **f32)
    dz = torch.randn(b, fout, nj - pool, **f32)
    dx, dpar, dgb = torch.empty_like(x), torch.empty(2 * fout * fin + 3 * fout, **f32), torch.empty(2, nj, **f32)
    scratch = torch.empty(int(lib.igcn_go_attn_bwd_scratch_floats(b, nj, fin, fout)), **f32)
    part = torch.empty(int(lib.igcn_go_ln_part_floats(b, nj)), **f32)

    def run(n=nj, pool_=pool, y_=y, dz_=dz, dz3=None):
        call("igcn_go_attn_ln_bwd", b, n, fin, fout, ptr(csr.row_ptr), ptr(csr.col), ptr(csr.t_ptr), ptr(csr.t_row),
             ptr(csr.walk_order(fin, fout)), ptr(x), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[3]), pool_, ptr(y_), ptr(gamma),
             ptr(beta), None, ptr(mean), ptr(rstd), ptr(dz_), None, ptr(dz3), ptr(dx), ptr(dpar), ptr(dgb), ptr(scratch),
             ptr(part), stream_ptr())

    assert lib.igcn_go_attn_ln_fused_ok(nj, fin, fout, pool) == 1
    run()                                                               # the covered case goes through
    torch.cuda.synchronize()
    assert lib.igcn_go_attn_ln_fused_ok(nj, fin, fout, pool + 2) == 0   # pooled prefix not a multiple of four nodes
    with pytest.raises(_lib.IgcnError):
        run(pool_=pool + 2)
    with pytest.raises(_lib.IgcnError):
        run(dz3=dz)                                                     # a third consumer without a second
    with pytest.raises(_lib.IgcnError):
        run(dz_=dz.view(-1)[1:])                                        # 4-byte aligned only
    assert lib.igcn_go_decode_ln_fused_ok(10, 21, 5, 5) == 0            # 21 output nodes: no 16-byte rows


@pytest.mark.parametrize("bsz,pool,layer,with_keep,seed", [
    (4, (20, 10, 6, 3, 1), 0, True, 0), (4, (20, 12, 8, 3, 1), 1, False, 1), (9, (300, 120, 60, 19, 1), 0, True, 2),
    (9, (300, 120, 60, 19, 1), 1, False, 3), (3, (1800, 800, 300, 99, 1), 0, True, 4),
    (3, (1800, 800, 300, 99, 1), 1, True, 5), (2, (1801, 800, 300, 99, 1), 1, True, 6)])
def test_go_decoder_with_layernorm_backward_in_one_launch(ops, monkeypatch, bsz, pool, layer, with_keep, seed):
    """ops.GoDecodeLN: decoder layer + LayerNorm block (no pooling), see the encoder test above."""
    from igcn_amd import _lib
    _, _, _, idx = _hier(pool, seed)
    row, col, n_rows, n_cols = idx["dec"][layer]
    fin, fout = 5, (5 if layer == 0 else 2)
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((bsz, n_cols, fin))).float()
    par = [torch.from_numpy(rng.standard_normal((fout, fin)) * 0.5).float() for _ in range(2)]
    gamma = torch.from_numpy(1 + 0.2 * rng.standard_normal(n_rows)).float()
    beta = torch.from_numpy(0.1 * rng.standard_normal(n_rows)).float()
    keep = torch.from_numpy((rng.random((bsz, n_rows)) > 0.3) / 0.7).float() if with_keep else None
    cot = torch.from_numpy(rng.standard_normal((bsz, fout, n_rows))).float()

    def ref(xd, w_out, w_sout, g_, b_):
        deg = torch.zeros(n_rows, dtype=xd.dtype).index_add(0, row, torch.ones(row.numel(), dtype=xd.dtype))
        agg = torch.zeros(bsz, n_rows, fout, dtype=xd.dtype).index_add(
            1, row, (xd @ w_out.t())[:, col] / deg[row].view(1, -1, 1))
        pad = torch.zeros_like(agg)
        pad[:, n_rows - n_cols:] = xd @ w_sout.t()
        return _ln_block_ref(agg + pad, g_, b_, keep, 0)

    ref_in = [t.double().requires_grad_(True) for t in [x] + par + [gamma, beta]]
    z_ref = ref(*ref_in)
    g_ref = torch.autograd.grad((z_ref * cot.double()).sum(), ref_in)
    csr = ops.Csr(row, col, n_rows, n_cols, "cuda")
    dev = [x.transpose(1, 2).contiguous().cuda().requires_grad_(True)] + \
          [p.cuda().requires_grad_(True) for p in par + [gamma, beta]]
    kd = keep.cuda() if keep is not None else None
    fused = bool(_lib.load().igcn_go_decode_ln_fused_ok(n_cols, n_rows, fin, fout))
    assert fused == (n_rows % 4 == 0)
    z = ops.GoDecodeLN.apply(dev[0], dev[1], dev[2], csr, dev[3], dev[4], kd, 1e-5)
    g = torch.autograd.grad((z * cot.cuda()).sum(), dev, retain_graph=True)
    names = ("dx", "dW_out", "dW_sout", "dgamma", "dbeta")
    assert_matches(z, z_ref.detach().numpy(), TOL, "z")
    assert_matches(g[0].transpose(1, 2), g_ref[0].numpy(), TOL, "dx")
    for got, want, nm in zip(g[1:], g_ref[1:], names[1:]):
        assert_matches(got, want.numpy(), TOL, nm)
    monkeypatch.setenv("IGCN_NO_LN_FUSED", "1")
    g2 = torch.autograd.grad((z * cot.cuda()).sum(), dev)
    for a, c2, nm in zip(g2, g, names):
        assert_matches(a, c2.cpu().numpy(), 2e-5, nm + " (two passes)")


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("bsz,pool,seed", [(4, (20, 10, 6, 3, 1), 0), (32, (300, 120, 60, 19, 1), 1),
                                           (37, (1800, 800, 300, 99, 1), 2)])
def test_sparse_map_encode_decode(ops, monkeypatch, bsz, pool, seed, dense):
    """The learnable sparse SNP <-> GO maps in both orientations — encode (C = 2, rows = GO nodes with a few SNPs each,
    plus the root's 54) and decode (C = 1, rows = SNPs with ~5 % of the nodes each) — on the LDS-tiled CSR kernels
    (default) and as dense image + MFMA GEMMs (IGCN_DENSE_MAPS): outputs and both gradients against fp64.  Batch
    sizes off the sample-tile sizes (8 / 4) exercise the partial tiles."""
    monkeypatch.setattr(ops.SparseMap, "DENSE_LIMIT", (1 << 24) if dense else 0)
    a_g, _, _, idx = _hier(pool, seed)
    n = idx["n"]
    gn, gs = idx["gene"]
    rng = np.random.default_rng(seed)
    # ---- encode: y[b, c, node] = sum over the node's SNPs
    snps = torch.from_numpy(rng.random((bsz, 54))).float()
    val = torch.from_numpy(1 + 0.1 * rng.standard_normal((2, gn.numel()))).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, 2, n))).float()
    ref_in = [snps.double().requires_grad_(True), val.double().requires_grad_(True)]
    y_ref = torch.stack([torch.zeros(bsz, n, dtype=torch.float64).index_add(1, gn, ref_in[0][:, gs] * ref_in[1][c])
                         for c in range(2)], dim=1)
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    csr = ops.Csr(gn, gs, n, 54, "cuda")
    dev = [snps.cuda().requires_grad_(True), val.cuda().requires_grad_(True)]
    y = ops.SparseMap.apply(dev[0], csr, dev[1])
    g = torch.autograd.grad((y * cot.cuda()).sum(), dev)
    assert_matches(y, y_ref.detach().numpy(), TOL, "y")
    assert_matches(g[0], g_ref[0].numpy(), TOL, "dsnps")
    assert_matches(g[1], g_ref[1].numpy(), TOL, "dval")
    # ---- decode: y[b, 0, snp] = sum over the SNP's GO nodes (the transposed structure, row-major again)
    order = torch.argsort(gs * n + gn)
    rs, cn = gs[order], gn[order]
    xg = torch.from_numpy(rng.standard_normal((bsz, n))).float()
    vald = torch.from_numpy(1 + 0.1 * rng.standard_normal((1, rs.numel()))).float()
    cotd = torch.from_numpy(rng.standard_normal((bsz, 1, 54))).float()
    ref_in = [xg.double().requires_grad_(True), vald.double().requires_grad_(True)]
    yd_ref = torch.zeros(bsz, 54, dtype=torch.float64).index_add(1, rs, ref_in[0][:, cn] * ref_in[1][0]).unsqueeze(1)
    gd_ref = torch.autograd.grad((yd_ref * cotd.double()).sum(), ref_in)
    csrd = ops.Csr(rs, cn, 54, n, "cuda")
    dev = [xg.cuda().requires_grad_(True), vald.cuda().requires_grad_(True)]
    yd = ops.SparseMap.apply(dev[0], csrd, dev[1])
    gd = torch.autograd.grad((yd * cotd.cuda()).sum(), dev)
    assert_matches(yd, yd_ref.detach().numpy(), TOL, "decode y")
    assert_matches(gd[0], gd_ref[0].numpy(), TOL, "decode dx")
    assert_matches(gd[1], gd_ref[1].numpy(), TOL, "decode dval")


def test_adam_matches_torch(ops):
    from igcn_amd.train import FlatAdam
    rng = np.random.default_rng(0)
    ps = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(s)).float().cuda()) for s in ((7, 5), (33,), (2, 3, 4))]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt, opt_ref = FlatAdam(ps, lr=1e-3), torch.optim.Adam(ref, lr=1e-3)
    for it in range(5):
        opt.zero_grad()
        for p, r in zip(ps, ref):
            g = torch.from_numpy(rng.standard_normal(tuple(p.shape))).float()
            p.grad = g.cuda()
            r.grad = g.clone()
        opt.step()
        opt_ref.step()
    for p, r in zip(ps, ref):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ read-outs / losses
@pytest.mark.parametrize("bsz,f,n,d,training,groups", [
    (6, 5, 11, 5, True, 1), (32, 5, 400, 32, True, 1), (32, 5, 400, 1, True, 1), (16, 2, 3000, 1, True, 1),
    (8, 5, 70, 8, False, 1), (5, 5, 33, 12, True, 1), (64, 5, 400, 32, True, 2), (12, 2, 130, 1, True, 2),
    (10, 5, 50, 1, False, 2),
    # row-coalesced kernels (D = 16 / 32): node counts that leave shadow lanes, eval mode, the bench shape
    (9, 5, 77, 32, True, 1), (6, 5, 77, 32, False, 1), (8, 5, 70, 16, True, 2), (512, 5, 400, 32, True, 2)])
def test_node_linear_bn(ops, bsz, f, n, d, training, groups):
    rng = np.random.default_rng(n + d)
    x = torch.from_numpy(rng.standard_normal((bsz, f, n)) + 0.5).float()
    w = torch.from_numpy(rng.standard_normal((d, f)) * 0.6).float()
    gamma = torch.from_numpy(1 + 0.2 * rng.standard_normal(n)).float()
    beta = torch.from_numpy(0.2 * rng.standard_normal(n)).float()
    rm0 = torch.from_numpy(0.1 * rng.standard_normal(n)).float()
    rv0 = torch.from_numpy(1 + 0.3 * rng.random(n)).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, n, d))).float()
    ref_in = [t.double().requires_grad_(True) for t in (x, w, gamma, beta)]
    rm, rv = rm0.double().clone(), rv0.double().clone()
    pre = ref_in[0].transpose(1, 2) @ ref_in[1].t()                       # [B,N,D]
    bg = bsz // groups                                                    # groups == successive module calls
    out_ref = torch.cat([torch.relu(torch.nn.functional.batch_norm(
        pre[g * bg:(g + 1) * bg], rm, rv, ref_in[2], ref_in[3], training, 0.1, 1e-5)) for g in range(groups)])
    g_ref = torch.autograd.grad((out_ref * cot.double()).sum(), ref_in)
    dev = [t.cuda().requires_grad_(True) for t in (x, w, gamma, beta)]
    rmg, rvg = rm0.cuda(), rv0.cuda()
    out = ops.NodeLinearBN.apply(dev[0], dev[1], dev[2], dev[3], rmg, rvg, training, 0.1, 1e-5, groups)
    g = torch.autograd.grad((out * cot.cuda()).sum(), dev)
    assert_matches(out, out_ref.detach().numpy(), TOL, "out")
    for got, want, nm in zip(g, g_ref, ("dx", "dW", "dgamma", "dbeta")):
        assert_matches(got, want.numpy(), 3e-4, nm, floor=1e-6)
    assert_matches(rmg, rm.numpy(), TOL, "running_mean")
    assert_matches(rvg, rv.numpy(), TOL, "running_var")


@pytest.mark.parametrize("bsz,c,training,relu,groups", [(32, 32, True, True, 1), (512, 32, True, True, 2),
                                                         (10, 7, True, False, 2), (9, 5, False, True, 1),
                                                      # general kernels: three groups / 1100 samples per group
                                                      (96, 8, True, True, 3), (2200, 4, True, False, 2)])
def test_batchnorm1d_grouped(ops, bsz, c, training, relu, groups):
    rng = np.random.default_rng(bsz + c)
    x = torch.from_numpy(rng.standard_normal((bsz, c)) * 1.5 + 0.2).float()
    gamma = torch.from_numpy(1 + 0.2 * rng.standard_normal(c)).float()
    beta = torch.from_numpy(0.2 * rng.standard_normal(c)).float()
    rm0, rv0 = torch.from_numpy(0.1 * rng.standard_normal(c)).float(), torch.from_numpy(1 + rng.random(c)).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, c))).float()
    ref_in = [t.double().requires_grad_(True) for t in (x, gamma, beta)]
    rm, rv = rm0.double().clone(), rv0.double().clone()
    bg = bsz // groups
    parts = [torch.nn.functional.batch_norm(ref_in[0][g * bg:(g + 1) * bg], rm, rv, ref_in[1], ref_in[2], training,
                                            0.1, 1e-5) for g in range(groups)]
    y_ref = torch.cat(parts)
    y_ref = torch.relu(y_ref) if relu else y_ref
    g_ref = torch.autograd.grad((y_ref * cot.double()).sum(), ref_in)
    dev = [t.cuda().requires_grad_(True) for t in (x, gamma, beta)]
    rmg, rvg = rm0.cuda(), rv0.cuda()
    y = ops.BatchNorm1dGrouped.apply(dev[0], dev[1], dev[2], rmg, rvg, training, 0.1, 1e-5, relu, groups)
    g = torch.autograd.grad((y * cot.cuda()).sum(), dev)
    assert_matches(y, y_ref.detach().numpy(), TOL, "y")
    for got, want, nm in zip(g, g_ref, ("dx", "dgamma", "dbeta")):
        assert_matches(got, want.numpy(), 3e-4, nm, floor=1e-6)
    assert_matches(rmg, rm.numpy(), TOL, "running_mean")
    assert_matches(rvg, rv.numpy(), TOL, "running_var")


def test_plan_replicate_equals_plan_of_the_doubled_graph(ops):
    rng = np.random.default_rng(0)
    n, e = 90 * 4, 270 * 4
    ei = torch.from_numpy(rng.integers(0, n, (2, e))).long().cuda()
    rep = ops.GraphPlan(ei, n).replicate(2)
    full = ops.GraphPlan(torch.cat([ei, ei + n], dim=1), 2 * n)
    for name in ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge"):
        assert torch.equal(getattr(rep, name), getattr(full, name)), name


def test_segmented_rebuild_refills_the_replica_plan(ops):
    """A per-graph (LDS) plan that has a 2-copy replica refills it inside rebuild() — no igcn_graph_plan_replicate launch
    behind the build: after a rebuild on a permuted batch the replica equals the replica derived from a fresh plan."""
    from igcn_amd import synth
    from igcn_amd.data import Batch
    batch = Batch.from_data_list(synth.brain_graph_list(8, seed=3, rois=90, tsne_dim=90)).to("cuda")
    plan = ops.plan_for(batch)
    assert plan.segmented and not plan._tiled
    rep = plan.replicate(2)
    addr = rep.tgt_perm.data_ptr()
    flipped = batch.edge_index.flip(0).contiguous()                 # still block diagonal, same per-graph counts
    plan.rebuild(flipped)
    assert plan.replicate(2) is rep and rep.tgt_perm.data_ptr() == addr
    n = int(batch.x.shape[0])
    fresh = ops.GraphPlan(flipped, n, batch.ptr, batch.edge_ptr, batch._max_nodes, batch._max_edges)
    want = ops.GraphPlan(torch.cat([flipped, flipped + n], dim=1), 2 * n)
    for name in ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge"):
        assert torch.equal(getattr(plan, name), getattr(fresh, name)), name
        assert torch.equal(getattr(rep, name), getattr(want, name)), "replica " + name


def test_mask_regulariser(ops):
    from oracle import sgcn_img_snp as OS
    rng = np.random.default_rng(3)
    prob = torch.from_numpy(rng.standard_normal((90, 3))).float()
    snps = torch.from_numpy(rng.standard_normal((1, 54))).float()
    e = torch.from_numpy(rng.random(5000) * 0.98 + 0.01).float()
    ref_in = [t.double().requires_grad_(True) for t in (prob, e, snps)]
    hp = OS.HP
    parts = [OS._bin_entropy_and_l1(torch.sigmoid(ref_in[0]), 1e-6), OS._bin_entropy_and_l1(ref_in[1], 1e-6),
             OS._bin_entropy_and_l1(torch.sigmoid(ref_in[2]), 1e-6)]
    ref = hp.lamda_x_l1 * parts[0][0] + hp.lamda_e_l1 * parts[1][0] + hp.lamda_x_l1 * parts[2][0] \
        + hp.lamda_x_ent * parts[0][1] + hp.lamda_e_ent * parts[1][1] + hp.lamda_x_ent * parts[2][1]
    g_ref = torch.autograd.grad(ref * 1.7, ref_in)
    dev = [t.cuda().requires_grad_(True) for t in (prob, e, snps)]
    got = ops.MaskRegulariser.apply(dev[0], dev[1], dev[2], hp.lamda_x_l1, hp.lamda_x_ent, hp.lamda_e_l1,
                                    hp.lamda_e_ent, 1e-6)
    g = torch.autograd.grad(got * 1.7, dev)
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref))
    for a, b, nm in zip(g, g_ref, ("dprob", "de", "dsnps")):
        assert_matches(a, b.numpy(), TOL, nm)


@pytest.mark.parametrize("bsz,rois,h0,deg,with_snps", [(8, 90, 3, 3, True), (3, 10, 3, 6, False), (3, 70, 3, 40, True),
                                                       (2, 17, 5, 6, True)])
def test_stacked_masks_carry_the_regulariser(ops, bsz, rois, h0, deg, with_snps):
    """ops.EdgeMaskStacked with reg_hp: loss_probability (kernel/sgcn_img_snp.py:153-181) rides in the mask launch of
    the stacked (plain | masked) sweep and its gradient in the mask's backward.  Against the fp64 oracle: both halves of
    the stacked tensors, the regulariser (sum of the partials), and the gradients of a loss that uses all of them — low
    and high degree graphs (4 lanes per node / a wave per node / the tiled walks), with and without SNP logits."""
    from oracle import sgcn_img_snp as OS
    rng = np.random.default_rng(bsz + rois)
    n = bsz * rois
    ei, ew = _rand_graph(rng, n, deg * n)
    ne = ei.shape[1]
    mk = lambda *sh: torch.from_numpy(rng.standard_normal(sh)).float()                 # noqa: E731
    x = torch.from_numpy(rng.random((n, h0))).float()
    prob, pb, snps = mk(rois, h0), mk(2 * h0, 1), mk(1, 54)
    cx, cw, reg_w = mk(2 * n, h0), mk(2 * ne), 1.7
    hp = OS.HP
    ref_in = [t.double().requires_grad_(True) for t in (x, prob, pb, snps)]
    xm, ewm, e = OS.edge_and_region_masks({"prob": ref_in[1], "prob_bias": ref_in[2]}, ref_in[0], ei, ew.double(), rois)
    parts = [OS._bin_entropy_and_l1(torch.sigmoid(ref_in[1]), 1e-6), OS._bin_entropy_and_l1(e, 1e-6)]
    if with_snps:
        parts.append(OS._bin_entropy_and_l1(torch.sigmoid(ref_in[3]), 1e-6))
    l1w, entw = (hp.lamda_x_l1, hp.lamda_e_l1, hp.lamda_x_l1), (hp.lamda_x_ent, hp.lamda_e_ent, hp.lamda_x_ent)
    reg = sum(l1w[i] * pt[0] + entw[i] * pt[1] for i, pt in enumerate(parts))
    total = (torch.cat([ref_in[0], xm]) * cx.double()).sum() + (torch.cat([ew.double(), ewm]) * cw.double()).sum() + reg_w * reg
    feat, cs = torch.from_numpy(rng.random((bsz, 54))).float(), mk(2 * bsz, 54)
    masked = feat.double() * torch.sigmoid(ref_in[3])
    if with_snps:
        total = total + (torch.cat([feat.double(), masked]) * cs.double()).sum()
    g_ref = torch.autograd.grad(total, ref_in, allow_unused=True)
    dev = [t.cuda().requires_grad_(True) for t in (x, prob, pb, snps)]
    plan = ops.GraphPlan(ei.cuda(), n)
    reg_hp = (hp.lamda_x_l1, hp.lamda_x_ent, hp.lamda_e_l1, hp.lamda_e_ent, 1e-6)
    extra = 0.0
    if with_snps:                                            # ... and the SNP mask of the stacked sweep (:147-151)
        outs = ops.EdgeMaskStacked.apply(dev[0], dev[1], dev[2], ew.cuda(), plan, rois, dev[3], reg_hp, feat.cuda())
        x_in, ew_in, e_g, regp, full = outs
        assert_matches(full, torch.cat([feat.double(), masked.detach()]).numpy(), TOL, "snps (plain | masked)")
        extra = (full * cs.cuda()).sum()
    else:
        x_in, ew_in, e_g, regp = ops.EdgeMaskStacked.apply(dev[0], dev[1], dev[2], ew.cuda(), plan, rois, None, reg_hp)
    assert_matches(x_in, torch.cat([x.double(), xm.detach()]).numpy(), TOL, "x_in")
    assert_matches(ew_in, torch.cat([ew.double(), ewm.detach()]).numpy(), TOL, "ew_in")
    assert_matches(e_g, e.detach().numpy(), TOL, "e")
    assert abs(float(regp.sum()) - float(reg)) <= 1e-5 * abs(float(reg))
    got = torch.autograd.grad((x_in * cx.cuda()).sum() + (ew_in * cw.cuda()).sum() + reg_w * regp.sum() + extra, dev,
                              allow_unused=True)
    for a, b, nm in zip(got, g_ref, ("dx", "dprob", "dprob_bias", "dsnps")):
        if b is None:
            assert a is None, nm
            continue
        assert_matches(a, b.numpy(), TOL, nm)


@pytest.mark.parametrize("bsz,rd,soft", [(7, 40, True), (64, 2880, True), (32, 300, False)])
def test_gram_losses(ops, bsz, rd, soft):
    from oracle import sgcn_img_snp as OS
    rng = np.random.default_rng(bsz)
    s = torch.from_numpy(rng.standard_normal((bsz, rd)) + 0.3).float()
    tsne = torch.from_numpy(rng.random((bsz, 16)) * 3).float()
    sr = s.double().requires_grad_(True)
    c_ref = OS.consist_loss(sr, tsne.double() if soft else None, 0.01, soft=soft)
    o_ref = OS.orthogonal_constraint(sr)
    g_ref = torch.autograd.grad(0.7 * c_ref + 0.3 * o_ref, sr)[0]
    sg = s.cuda().requires_grad_(True)
    lap = ops.rbf_laplacian(tsne.cuda() if soft else None, bsz, 0.01, "cuda")
    c, o = ops.GramLosses.apply(sg, lap, 1)
    c, o = c[0], o[0]
    g = torch.autograd.grad(0.7 * c + 0.3 * o, sg)[0]
    assert abs(float(c) - float(c_ref)) <= 1e-4 * max(abs(float(c_ref)), 1e-6)
    assert abs(float(o) - float(o_ref)) <= 1e-4 * max(abs(float(o_ref)), 1e-6)
    assert_matches(g, g_ref.numpy(), 2e-4, "ds")
    # the train step's form: the Laplacian built INSIDE the loss kernel (igcn_gram_loss_fwd_rbf), no launch in front
    sg2 = s.cuda().requires_grad_(True)
    c2, o2 = ops.GramLosses.apply(sg2, None, 1, False, (tsne.cuda() if soft else None, 0.01))
    g2 = torch.autograd.grad(0.7 * c2[0] + 0.3 * o2[0], sg2)[0]
    assert abs(float(c2[0]) - float(c_ref)) <= 1e-4 * max(abs(float(c_ref)), 1e-6)
    assert torch.equal(o2, torch.stack([o]))
    assert_matches(g2, g_ref.numpy(), 2e-4, "ds (Laplacian built in the kernel)")


def test_gram_losses_grouped(ops):
    from oracle import sgcn_img_snp as OS
    rng = np.random.default_rng(1)
    bsz, rd = 16, 96
    s = torch.from_numpy(rng.standard_normal((2 * bsz, rd)) + 0.3).float()
    tsne = torch.from_numpy(rng.random((bsz, 8)) * 3).float()
    sr = s.double().requires_grad_(True)
    ref = [0.7 * OS.consist_loss(sr[k * bsz:(k + 1) * bsz], tsne.double(), 0.01) * (k + 1)
           + 0.3 * OS.orthogonal_constraint(sr[k * bsz:(k + 1) * bsz]) for k in range(2)]
    g_ref = torch.autograd.grad(ref[0] + ref[1], sr)[0]
    sg = s.cuda().requires_grad_(True)
    lap = ops.rbf_laplacian(tsne.cuda(), bsz, 0.01, "cuda")
    c, o = ops.GramLosses.apply(sg, lap, 2)
    w = torch.tensor([1.0, 2.0], device="cuda")
    g = torch.autograd.grad((0.7 * c * w).sum() + 0.3 * o.sum(), sg)[0]
    assert_matches(g, g_ref.numpy(), 2e-4, "ds")


def test_gram_loss_forward_prepares_the_backward_for_the_announced_upstream(ops):
    """``GramLosses(expect=...)`` (the train step: d loss / d (consist, orth) = the loss weights, known at the forward):
    the forward kernel writes the backward's S itself; a backward whose upstream was announced through UNIT_DGRAM takes
    it (no igcn_gram_loss_bwd launch) and returns the SAME bytes as the regular backward; any other upstream — other
    values, or none announced — runs the regular backward."""
    rng = np.random.default_rng(3)
    bsz, rd = 48, 160
    s = torch.from_numpy(rng.standard_normal((2 * bsz, rd)) + 0.3).float().cuda()
    tsne = torch.from_numpy(rng.random((bsz, 16)) * 3).float().cuda()
    lam = [0.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]
    unit = ops.unit_dgram(lam)
    up = torch.tensor(unit, device="cuda")

    def run(expect, announce):
        sg = s.clone().requires_grad_(True)
        part = ops.GramLosses.apply(sg, None, 2, "partials", (tsne, 0.01), expect)
        g = up.view(1, 4).expand(part.shape[0], 4)
        ops.UNIT_DGRAM.clear()
        if announce is not None:
            ops.UNIT_DGRAM[up.data_ptr()] = announce
        before = _gram_bwd_calls[0]
        ds = torch.autograd.grad(part, sg, g)[0]
        return part.detach().clone(), ds, _gram_bwd_calls[0] - before

    from igcn_amd import _lib
    _gram_bwd_calls = [0]
    real = _lib.call

    def counting(name, *a):
        if name == "igcn_gram_loss_bwd":
            _gram_bwd_calls[0] += 1
        return real(name, *a)
    ops.call = counting
    try:
        p0, d0, n0 = run(None, None)                     # regular
        p1, d1, n1 = run(unit, unit)                     # prepared and announced: no backward launch
        p2, d2, n2 = run(unit, None)                     # prepared, nobody announced the upstream: regular backward
        p3, d3, n3 = run(unit, (9.0, 9.0, 9.0, 9.0))     # announced with other values: regular backward
    finally:
        ops.call = real
    assert (n0, n1, n2, n3) == (1, 0, 1, 1)
    for p, d in ((p1, d1), (p2, d2), (p3, d3)):
        assert torch.equal(p, p0) and torch.equal(d, d0)
    assert not ops.UNIT_DGRAM or n3 == 1


@pytest.mark.parametrize("n_graphs,seed", [(1, 0), (8, 1), (256, 2)])
def test_segmented_plan_is_bit_identical_to_the_sorted_plan(ops, n_graphs, seed):
    from igcn_amd import synth
    batch = synth.brain_batch(n_graphs, seed=seed, rois=90).to("cuda")
    seg = ops.plan_for(batch)
    assert seg.status is not None          # the one-workgroup-per-graph path was taken
    seg.check()
    ref = ops.GraphPlan(batch.edge_index, batch.x.shape[0])          # device-wide stable radix sort
    for name in ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge"):
        assert torch.equal(getattr(seg, name), getattr(ref, name)), name


def test_segmented_plan_ragged_graphs_and_loops(ops):
    from igcn_amd.data import Batch, Data
    rng = np.random.default_rng(5)
    graphs = []
    for n in (1, 7, 90, 33, 2):
        e = int(rng.integers(0, 6 * n + 1))
        ei = torch.from_numpy(rng.integers(0, n, (2, e))).long()
        graphs.append(Data(x=torch.zeros(n, 3), edge_index=ei, edge_attr=torch.ones(e)))
    batch = Batch.from_data_list(graphs).to("cuda")
    seg = ops.plan_for(batch)
    seg.check()
    ref = ops.GraphPlan(batch.edge_index, batch.x.shape[0])
    for name in ("tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge"):
        assert torch.equal(getattr(seg, name), getattr(ref, name)), name


def test_segmented_plan_refuses_offsets_outside_the_batch(ops):
    """The per-graph build reads ptr / edge_ptr from DEVICE memory (GraphedTrainStep.load copies a loader's tensors
    there): offsets that leave the batch set the status word; nothing is read or written out of bounds."""
    from igcn_amd import _lib, synth
    batch = synth.brain_batch(6, seed=3, rois=30).to("cuda")
    plan = ops.plan_for(batch)
    plan.check()
    good = plan._seg[1].clone()
    bad = good.clone()
    bad[2] = 1 << 40                                                  # graph 1 "ends" (and graph 2 starts) far away
    plan._seg[1].copy_(bad)
    plan.rebuild(batch.edge_index)
    with pytest.raises(_lib.IgcnError):
        plan.check()
    plan.status.zero_()
    plan._seg[1].copy_(good)
    plan.rebuild(batch.edge_index)
    plan.check()


@pytest.mark.parametrize("sizes,edges", [((512, 512, 512), (262144, 262144, 262144)),     # dense stress-shape graphs
                                         ((300, 1024, 5, 77), (9000, 40000, 0, 4097)),     # ragged, an empty graph
                                         ((90, 90), (270, 4097))])
def test_tiled_plan_is_bit_identical_to_a_stable_sort(ops, sizes, edges, monkeypatch):
    """igcn_graph_plan_build_tiled (graphs with more than 4096 edges: one-pass counting sort per graph) against
    numpy's stable argsort and the general multi-pass build."""
    from igcn_amd.data import Batch, Data
    # complete row-major graphs would make this a dense-block plan, whose rebuild only re-verifies the structure
    # (tests/test_gpu_dense.py); this test is about the sorting builders
    monkeypatch.setenv("IGCN_NO_DENSE_BLOCKS", "1")
    rng = np.random.default_rng(11)
    graphs = []
    for n, e in zip(sizes, edges):
        if e == n * n:                                              # all pairs in row-major order, like dense_graph()
            ei = torch.stack([torch.arange(n).repeat_interleave(n), torch.arange(n).repeat(n)])
        else:
            ei = torch.from_numpy(rng.integers(0, n, (2, e))).long()
        graphs.append(Data(x=torch.zeros(n, 3), edge_index=ei, edge_attr=torch.ones(ei.shape[1])))
    batch = Batch.from_data_list(graphs).to("cuda")
    plan = ops.plan_for(batch)
    assert plan._tiled
    plan.check()
    ei = batch.edge_index.cpu().numpy()
    n = int(batch.x.shape[0])
    for key, ptr_t, perm_t in ((ei[1], plan.tgt_ptr, plan.tgt_perm), (ei[0], plan.src_ptr, plan.src_perm)):
        perm = np.argsort(key, kind="stable").astype(np.int32)
        assert np.array_equal(perm_t.cpu().numpy(), perm)
        assert np.array_equal(ptr_t.cpu().numpy(), np.searchsorted(key[perm], np.arange(n + 1)).astype(np.int32))
    ref = ops.GraphPlan(batch.edge_index, n)                        # general LSD build
    for name in ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge"):
        assert torch.equal(getattr(plan, name), getattr(ref, name)), name
    # rebuilt in place inside a hipGraph: same result after a replay on a permuted batch of the same sizes
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.rebuild(batch.edge_index)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        plan.rebuild(batch.edge_index)
        ref.rebuild(batch.edge_index)
    flipped = batch.edge_index.flip(0).contiguous()                 # still block diagonal, same per-graph counts
    batch.edge_index.copy_(flipped)
    g.replay()
    torch.cuda.synchronize()
    key = flipped[1].cpu().numpy()
    assert np.array_equal(plan.tgt_perm.cpu().numpy(), np.argsort(key, kind="stable").astype(np.int32))
    assert torch.equal(plan.tgt_perm, ref.tgt_perm) and torch.equal(plan.src_ptr, ref.src_ptr)


@pytest.mark.parametrize("hint", [0, 64])
def test_propagate_dense_graph_variants(ops, hint):
    """High in-degree launch shape of the scatter-aggregate (one wave per target, 16 B per lane)."""
    from oracle import pyg_ops
    rng = np.random.default_rng(hint)
    g, r, f = 3, 64, 16
    rr = torch.arange(r).repeat_interleave(r)
    cc = torch.arange(r).repeat(r)
    ei = torch.cat([torch.stack([rr, cc]) + k * r for k in range(g)], dim=1)
    ew = torch.from_numpy(rng.random(ei.shape[1]) / r + 0.01).float()
    x = torch.from_numpy(rng.standard_normal((g * r, f))).float()
    w = torch.eye(f)
    b = torch.from_numpy(rng.standard_normal(f)).float()
    want = torch.relu(pyg_ops.gcn_conv(x.double(), ei, ew.double(), w.double(), b.double()))
    plan = ops.GraphPlan(ei.cuda(), g * r)
    plan.nodes_per_graph = hint
    coef = ops.GcnNorm.apply(ew.cuda(), plan)
    out = ops.GcnPropagate.apply(x.cuda(), coef[0], coef[1], b.cuda(), plan, True, coef[2], coef[3])
    assert_matches(out, want.numpy(), TOL, "out")


@pytest.mark.parametrize("g,r,f,density,cross", [(3, 64, 16, 1.0, False), (2, 200, 16, 0.9, False),
                                                 (2, 130, 8, 1.0, False), (2, 70, 64, 1.0, False),
                                                 (3, 96, 4, 0.8, False), (2, 128, 32, 1.0, True)])
def test_propagate_lds_staged_dense_fwd_bwd(ops, monkeypatch, g, r, f, density, cross):
    """The LDS-staged scatter-aggregate for dense uniform batches (igcn_gcn_propagate_{fwd,bwd} with the
    nodes_per_graph hint, average in-degree >= 64): forward, dh, dbias and the coefficient gradients against the fp64
    oracle and against the wave-per-target kernels (IGCN_PROPAGATE_NO_LDS=1).  `density` < 1 gives ragged lists (odd
    record offsets: the 16-byte record loads start on an even position and mask the neighbour's record); `cross`
    adds edges BETWEEN graphs, which the staged rows do not cover (global-row fallback)."""
    from oracle import pyg_ops
    rng = np.random.default_rng(g * 1000 + r + f)
    src, dst = [], []
    for k in range(g):
        m = rng.random((r, r)) < density
        rr, cc = np.nonzero(m)
        src.append(rr + k * r)
        dst.append(cc + k * r)
    if cross:
        src.append(rng.integers(0, r, 300)), dst.append(rng.integers(r, 2 * r, 300))
    ei = torch.from_numpy(np.vstack([np.concatenate(src), np.concatenate(dst)])).long()
    ew = torch.from_numpy(rng.random(ei.shape[1]) / r + 0.01).float()
    x = torch.from_numpy(rng.standard_normal((g * r, f))).float()
    b = torch.from_numpy(rng.standard_normal(f)).float()
    cot = torch.from_numpy(rng.standard_normal((g * r, f))).float()
    xd, ewd, bd = x.double().requires_grad_(True), ew.double().requires_grad_(True), b.double().requires_grad_(True)
    want = torch.relu(pyg_ops.gcn_conv(xd, ei, ewd, torch.eye(f, dtype=torch.float64), bd))
    (want * cot.double()).sum().backward()

    def run():
        plan = ops.GraphPlan(ei.cuda(), g * r)
        plan.nodes_per_graph = r
        xg, ewg, bg = x.cuda().requires_grad_(True), ew.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        coef = ops.GcnNorm.apply(ewg, plan)
        out = ops.GcnPropagate.apply(xg, coef[0], coef[1], bg, plan, True, coef[2], coef[3])
        (out * cot.cuda()).sum().backward()
        return out.detach(), xg.grad, ewg.grad, bg.grad

    assert ei.shape[1] >= 64 * g * r                     # the shape that selects the LDS-staged kernels
    got = run()
    monkeypatch.setenv("IGCN_PROPAGATE_NO_LDS", "1")
    ref = run()
    for name, a, c, w in zip(("out", "dh", "dew", "dbias"), got, ref, (want.detach(), xd.grad, ewd.grad, bd.grad)):
        assert_matches(a, w.numpy(), TOL, name + " vs oracle")
        assert_matches(a, c.cpu().numpy(), 2e-5, name + " vs wave-per-target kernels")


@pytest.mark.parametrize("bsz,lq,lk,d", [(3, 10, 9, 8), (4, 90, 400, 32), (2, 130, 77, 16), (5, 90, 45, 32),
                                          (2, 300, 130, 32), (3, 7, 5, 32), (2, 16, 16, 32),
                                          # head_dim 16 with <= 8 query tiles and >= 8 key tiles: the shared-score-tile
                                          # backward (ragged tiles, chunks that split key tiles between waves)
                                          (3, 128, 400, 32), (2, 17, 130, 32), (2, 90, 136, 32), (2, 48, 300, 32),
                                          (1, 1, 128, 32), (2, 113, 777, 32),
                                          # head dims off the fast path of round 1: 10, 15, 24, 12, 32, 5 (padded in LDS)
                                          (4, 90, 400, 20), (3, 90, 140, 30), (4, 90, 200, 48), (2, 33, 50, 24),
                                          (2, 40, 70, 64), (3, 21, 19, 10),
                                          # K, V too large for LDS: the chunked (streamed) kernels
                                          (2, 512, 1300, 32), (1, 300, 2500, 20), (2, 1100, 90, 32)])
def test_attention_core(ops, bsz, lq, lk, d):
    rng = np.random.default_rng(lq * lk)
    h = 2
    q = torch.from_numpy(rng.standard_normal((bsz, lq, d))).float()
    kv = torch.from_numpy(rng.standard_normal((bsz, lk, 2 * d))).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, lq, d))).float()
    rq, rkv = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = rq.view(bsz, lq, h, d // h).transpose(1, 2)
    kvh = rkv.view(bsz, lk, 2, h, d // h)
    k_, v_ = kvh[:, :, 0].transpose(1, 2), kvh[:, :, 1].transpose(1, 2)
    att = torch.softmax(qh @ k_.transpose(-1, -2) / (d // h) ** 0.5, dim=-1)
    o_ref = (att @ v_).transpose(1, 2).reshape(bsz, lq, d)
    g_ref = torch.autograd.grad((o_ref * cot.double()).sum(), [rq, rkv])
    assert ops.attn_core_supported(d, h, lq, lk)
    dq_, dkv_ = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    o = ops.AttentionCore.apply(dq_, dkv_, h)
    g = torch.autograd.grad((o * cot.cuda()).sum(), [dq_, dkv_])
    assert_matches(o, o_ref.detach().numpy(), TOL, "o")
    assert_matches(g[0], g_ref[0].numpy(), 2e-4, "dq", floor=1e-6)
    assert_matches(g[1], g_ref[1].numpy(), 2e-4, "dkv", floor=1e-6)


@pytest.mark.parametrize("bsz,lq,lk", [(4, 90, 400), (2, 512, 1300), (3, 50, 37), (2, 16, 32), (1, 7, 5),
                                        (1, 300, 2500), (2, 1100, 90)])
def test_attention_core_bf16_operands(ops, bsz, lq, lk):
    """igcn_attn_core_bf16_*: the same attention with bf16 OPERANDS on the matrix cores (fp32 accumulation, fp32 softmax).
    Against fp64 torch: o within 1e-2, the gradients within 3e-2 of their scale (bf16 has 8 mantissa bits: 4e-3 per
    operand, averaged down over the reductions); and NOT identical to the fp32 core, so the test sees the kernels it
    names.  Ragged tile edges (50, 37, 7, 5), one chunk and several chunks on either side."""
    rng = np.random.default_rng(lq * lk + 1)
    h, d = 2, 32
    q = torch.from_numpy(rng.standard_normal((bsz, lq, d))).float()
    kv = torch.from_numpy(rng.standard_normal((bsz, lk, 2 * d))).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, lq, d))).float()
    rq, rkv = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = rq.view(bsz, lq, h, d // h).transpose(1, 2)
    kvh = rkv.view(bsz, lk, 2, h, d // h)
    k_, v_ = kvh[:, :, 0].transpose(1, 2), kvh[:, :, 1].transpose(1, 2)
    att = torch.softmax(qh @ k_.transpose(-1, -2) / (d // h) ** 0.5, dim=-1)
    o_ref = (att @ v_).transpose(1, 2).reshape(bsz, lq, d)
    g_ref = torch.autograd.grad((o_ref * cot.double()).sum(), [rq, rkv])
    assert ops.attn_core_bf16(d, h, lq, lk)
    dq_, dkv_ = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    o = ops.AttentionCore.apply(dq_, dkv_, h, True)
    g = torch.autograd.grad((o * cot.cuda()).sum(), [dq_, dkv_])
    o32 = ops.AttentionCore.apply(dq_, dkv_, h)
    assert not torch.equal(o, o32)
    assert_matches(o, o_ref.detach().numpy(), 1e-2, "o")
    assert_matches(g[0], g_ref[0].numpy(), 3e-2, "dq")
    assert_matches(g[1], g_ref[1].numpy(), 3e-2, "dkv")
    # rms error well under the bound (a systematic layout slip shows up here long before it reaches the max)
    rel = lambda a, w: float((a.detach().cpu().double() - w).pow(2).mean().sqrt() / w.pow(2).mean().sqrt())   # noqa: E731
    assert rel(o, o_ref.detach()) < 5e-3 and rel(g[0], g_ref[0]) < 1e-2 and rel(g[1], g_ref[1]) < 1e-2


def test_attention_cores_are_deterministic_and_agree(ops):
    """Three kernels serve head_dim 16 at the model's shape: the split-bf16 core (csrc/attn_split.hip: the default), the
    exact-fp32 core with shared score tiles (IGCN_ATTN_EXACT_FP32=1: k_attn_mfma_fwd / k_attn_mfma_bwd_shared) and its
    two-orientation backward (+ IGCN_ATTN_BWD_TWICE=1).  Each gives the same bits on every run (static splits, no
    atomics); the two exact kernels agree to fp32 rounding (2e-5), the split core with them at the model's stated
    bounds (1e-4 output, 1e-3 gradients; measured 6e-6 / 2.5e-5, tools/attn_error.py) without being identical — each in a
    child process, because the switches are read once when the library loads."""
    import subprocess
    import sys
    code = """
import os, sys, torch
sys.path.insert(0, %r)
import igcn_amd
from igcn_amd import ops
torch.manual_seed(3)
q = torch.randn(6, 90, 32, device="cuda", requires_grad=True)
kv = torch.randn(6, 400, 64, device="cuda", requires_grad=True)
cot = torch.randn(6, 90, 32, device="cuda")
gs = []
for _ in range(3):
    o = ops.AttentionCore.apply(q, kv, 2)
    gs.append([o.detach()] + list(torch.autograd.grad((o * cot).sum(), [q, kv])))
assert all(torch.equal(a, b) for g in gs[1:] for a, b in zip(gs[0], g)), "run-to-run bits differ"
torch.save([t.cpu() for t in gs[0]], sys.argv[1])
"""
    import tempfile
    from conftest import ROOT
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, extra in (("split", {}), ("exact", {"IGCN_ATTN_EXACT_FP32": "1"}),
                           ("twice", {"IGCN_ATTN_EXACT_FP32": "1", "IGCN_ATTN_BWD_TWICE": "1"})):
            env = dict(os.environ)
            for k in ("IGCN_ATTN_BWD_TWICE", "IGCN_ATTN_EXACT_FP32"):
                env.pop(k, None)
            env.update(extra)
            path = os.path.join(d, f"g_{tag}.pt")
            r = subprocess.run([sys.executable, "-c", code % ROOT, path], env=env, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, (tag, r.stderr[-2000:])
            outs[tag] = torch.load(path)
    assert torch.equal(outs["exact"][0], outs["twice"][0])                      # (the switch is about the backward)
    for a, b, nm in zip(outs["exact"][1:], outs["twice"][1:], ("dq", "dkv")):
        assert not torch.equal(a, b), nm + ": the switch did not change the kernel"
        assert_matches(a, b.numpy(), 2e-5, nm + " shared vs two-pass", floor=1e-6)
    for a, b, nm, tol_ in zip(outs["split"], outs["exact"], ("o", "dq", "dkv"), (1e-4, 1e-3, 1e-3)):
        assert not torch.equal(a, b), nm + ": the split core did not run"
        assert_matches(a, b.numpy(), tol_, nm + " split vs exact", floor=1e-6)


@pytest.mark.parametrize("bsz,lq,lk,amp", [(8, 90, 400, 1.0), (4, 90, 400, 3.0), (3, 37, 203, 1.0), (2, 128, 129, 1.0),
                                            (2, 512, 1300, 1.0), (3, 50, 37, 2.0), (1, 7, 5, 1.0)])
def test_attention_core_split_bf16_vs_fp64(ops, bsz, lq, lk, amp):
    """igcn_attn_core_split_*: every product of the attention on v_mfma_f32_16x16x32_bf16 with operands carried as a bf16
    head + a bf16 remainder.  Against fp64 torch the output holds 1e-4 and the gradients 1e-3 of their scale — the
    bounds of the exact-fp32 core — also with a sharp softmax (``amp`` 3: scores of +-20 before the softmax), ragged tiles,
    one and several key chunks; the backward where its one-workgroup form applies (<= 128 queries, >= 128 keys)."""
    from igcn_amd import _lib
    from igcn_amd._lib import call, stream_ptr
    rng = np.random.default_rng(lq * lk + 2)
    h, d = 2, 32
    q = torch.from_numpy(rng.standard_normal((bsz, lq, d)) * amp).float()
    kv = torch.from_numpy(rng.standard_normal((bsz, lk, 2 * d)) * amp).float()
    cot = torch.from_numpy(rng.standard_normal((bsz, lq, d))).float()
    rq, rkv = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    qh = rq.view(bsz, lq, h, d // h).transpose(1, 2)
    kvh = rkv.view(bsz, lk, 2, h, d // h)
    k_, v_ = kvh[:, :, 0].transpose(1, 2), kvh[:, :, 1].transpose(1, 2)
    att = torch.softmax(qh @ k_.transpose(-1, -2) / (d // h) ** 0.5, dim=-1)
    o_ref = (att @ v_).transpose(1, 2).reshape(bsz, lq, d)
    g_ref = torch.autograd.grad((o_ref * cot.double()).sum(), [rq, rkv])
    lib = _lib.load()
    assert lib.igcn_attn_core_split_supported(d, h, lq, lk)
    qg, kvg, cg = q.cuda(), kv.cuda(), cot.cuda()
    o = torch.empty_like(qg)
    lse = torch.empty(bsz, h, lq, device="cuda")
    call("igcn_attn_core_split_fwd", bsz, d, h, lq, lk, qg.data_ptr(), kvg.data_ptr(), o.data_ptr(), lse.data_ptr(),
         stream_ptr())
    assert_matches(o, o_ref.detach().numpy(), 1e-4, "o")
    lse_ref = torch.logsumexp(qh @ k_.transpose(-1, -2) / (d // h) ** 0.5, dim=-1).detach()
    assert float((lse.cpu().double() - lse_ref).abs().max()) <= 1e-4 * max(1.0, float(lse_ref.abs().max()))
    if lib.igcn_attn_core_split_bwd_supported(d, h, lq, lk):
        dq, dkv = torch.empty_like(qg), torch.empty_like(kvg)
        call("igcn_attn_core_split_bwd", bsz, d, h, lq, lk, qg.data_ptr(), kvg.data_ptr(), o.data_ptr(), lse.data_ptr(),
             cg.data_ptr(), dq.data_ptr(), dkv.data_ptr(), None, stream_ptr())
        assert_matches(dq, g_ref[0].numpy(), 1e-3, "dq", floor=1e-6)
        assert_matches(dkv, g_ref[1].numpy(), 1e-3, "dkv", floor=1e-6)
    else:
        assert lq > 128 or lk < 128


def test_loss_head_matches_composite():
    """igcn_loss_head_* against the term-by-term composition of train() :525-543 (fp64 torch), values and gradients."""
    import torch.nn.functional as F
    from igcn_amd import ops
    torch.manual_seed(0)
    b, c, nr, s = 37, 3, 3, 54
    dev = "cuda"
    logp = torch.log_softmax(torch.randn(2 * b, c, device=dev), -1).requires_grad_(True)
    y = torch.randint(0, c, (b,), device=dev)
    reg = torch.randn(2 * b, nr, device=dev, requires_grad=True)
    clin = torch.rand(b * nr, device=dev)
    x_hat = torch.randn(2 * b, s, device=dev, requires_grad=True)
    snps = torch.rand(b, s, device=dev)
    gram = torch.rand(2, 2, device=dev, requires_grad=True)
    prob = torch.rand((), device=dev, requires_grad=True)
    lam, hp_ce, hp_mi = [0.7, 1.0, 0.5, 1.5e-3, 0.1, 0.2], 1.3, 0.8
    loss, terms = ops.LossHead.apply(logp, y, reg, clin, x_hat, snps, gram, prob, lam, hp_ce, hp_mi)
    (loss * 1.7).backward()
    got = [t.grad.clone() for t in (logp, reg, x_hat, gram, prob)]
    d = lambda t: t.detach().double().requires_grad_(True)          # noqa: E731
    logp2, reg2, x2, gram2, prob2 = d(logp), d(reg), d(x_hat), d(gram), d(prob)
    t = [lam[0] * F.nll_loss(logp2[:b], y), lam[0] * F.nll_loss(logp2[b:], y),
         lam[1] * (F.mse_loss(reg2[:b].reshape(-1), clin.double()) + F.mse_loss(reg2[b:].reshape(-1), clin.double())) / 2,
         lam[2] * prob2,
         lam[3] * (((x2[:b] - snps.double()) ** 2).sum() + ((x2[b:] - snps.double()) ** 2).sum()) / 2,
         lam[4] * (gram2[0, 0] + gram2[1, 0]) / 2, lam[5] * gram2[0, 1]]
    want = hp_ce * t[0] + hp_mi * t[1] + sum(t[2:])
    (want * 1.7).backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    for k in range(7):
        assert abs(float(terms[k]) - float(t[k])) <= 1e-5 * max(1e-3, abs(float(t[k]))), k
    for g, w, name in zip(got, (logp2, reg2, x2, gram2, prob2), ("logp", "reg", "x_hat", "gram", "prob")):
        assert_matches(g, w.grad.float().cpu().numpy(), 1e-5, "grad " + name)


@pytest.mark.parametrize("b,k,c,nr,drop,upstream,lazy", [(37, 64, 3, 3, True, 1.0, False), (256, 64, 3, 3, True, 1.0, True),
                                                          (5, 32, 2, 4, False, 1.0, False), (40, 64, 3, 3, True, 1.7, False),
                                                          (16, 16, 4, 1, False, 1.0, True)])
def test_head_loss_one_launch_equals_heads_plus_loss_head(b, k, c, nr, drop, upstream, lazy):
    """ops.HeadLoss (igcn_head_loss_fwd + igcn_loss_final): lin2 | lin2_regr, log_softmax, the loss head and the backward
    of all of it in one multi-workgroup launch, against the chain it replaces — ops.small_linear_pair -> ops.LossHead
    (unit-gradient route) -> their backward launches — on the same inputs: loss, the seven terms, log_softmax, regression
    outputs and every gradient (features, both layers' weights and biases, x_hat, Gram terms, regulariser), with and without
    dropout factors, for the unit upstream gradient of a train step and a general one, with the loss value issued at once
    and lazily (inside ``deferred_reductions``: as an entry of the flush)."""
    from igcn_amd import ops
    from igcn_amd.train import _unit_grad
    torch.manual_seed(b + k)
    dev, s = "cuda", 54
    mk = lambda *sh: torch.randn(*sh, device=dev)                        # noqa: E731
    hf, hr = mk(2 * b, k).relu(), mk(2 * b, k).relu()
    keep1 = (torch.rand(2 * b, k, device=dev) > 0.5).float() * 2.0 if drop else None
    keep2 = (torch.rand(2 * b, k, device=dev) > 0.3).float() / 0.7 if drop else None
    w2, b2, w2r, b2r = mk(c, k) * 0.3, mk(c) * 0.1, mk(nr, k) * 0.3, mk(nr) * 0.1
    y = torch.randint(0, c, (b,), device=dev)
    clin, x_hat, snps = torch.rand(b * nr, device=dev), mk(2 * b, s), torch.rand(b, s, device=dev)
    gram, prob = torch.rand(7, 4, device=dev), torch.rand(11, device=dev)
    lam, hp_ce, hp_mi = [0.7, 1.0, 0.5, 1.5e-3, 0.1, 0.2], 1.3, 0.8
    unit = _unit_grad(torch.zeros((), device=dev))                   # registers the cached d loss / d loss = 1
    assert unit is not None and ops.UNIT_GRAD_PTRS

    def leaves():
        return [t.clone().requires_grad_(True) for t in (hf, w2, b2, hr, w2r, b2r, x_hat, gram, prob)]

    def finish(loss, ls):
        go = unit if upstream == 1.0 else torch.full((), upstream, device=dev)
        if lazy:
            with ops.deferred_reductions():
                g = torch.autograd.grad(loss, ls, grad_outputs=go)
        else:
            g = torch.autograd.grad(loss, ls, grad_outputs=go)
        torch.cuda.synchronize()
        return g
    # the chain
    ls0 = leaves()
    logits, reg_o = ops.small_linear_pair(ls0[0], ls0[1], ls0[2], keep1, ls0[3], ls0[4], ls0[5], keep2)
    loss0, terms0, logp0 = ops.LossHead.apply(logits, y, reg_o, clin, ls0[6], snps, ls0[7], ls0[8], lam, hp_ce, hp_mi, True)
    g0 = finish(loss0, ls0)
    # the one launch
    ls1 = leaves()
    assert ops.head_loss_supported(ls1[0], ls1[1], ls1[3], ls1[4], keep1, keep2)
    loss1, terms1, logp1, reg1 = ops.HeadLoss.apply(ls1[0], keep1, ls1[1], ls1[2], ls1[3], keep2, ls1[4], ls1[5], y, clin,
                                                    ls1[6], snps, ls1[7], ls1[8], lam, hp_ce, hp_mi, lazy)
    g1 = finish(loss1, ls1)
    assert abs(float(loss1) - float(loss0)) <= 2e-6 * max(1.0, abs(float(loss0)))
    for j in range(7):
        assert abs(float(terms1[j]) - float(terms0[j])) <= 2e-6 * max(1e-3, abs(float(terms0[j]))), j
    assert_matches(logp1, logp0.detach().cpu().numpy(), 2e-6, "log_softmax")
    assert_matches(reg1, reg_o.detach().cpu().numpy(), 2e-6, "regression outputs")
    for a_, w_, nm in zip(g1, g0, ("features", "lin2.weight", "lin2.bias", "features_regr", "lin2_regr.weight",
                                    "lin2_regr.bias", "x_hat", "gram", "prob")):
        assert_matches(a_, w_.detach().cpu().numpy(), 5e-6, "grad " + nm, floor=1e-7)


@pytest.mark.parametrize("m,n,k,batch,sk,form", [(256, 256, 2880, 2, 4, "nt"), (256, 2880, 256, 2, 1, "nn"),
                                                 (70, 33, 131, 3, 2, "nt"), (5, 7, 32, 4, 1, "nn")])
def test_gemm_f32_batched(m, n, k, batch, sk, form):
    """igcn_gemm_f32_batched: `batch` independent products in one launch (the Gram matrices of the two passes and their
    backward products) against fp64 matmul, split and unsplit, both operand layouts, ragged sizes."""
    from igcn_amd._lib import call, ptr, stream_ptr
    torch.manual_seed(m + n + k)
    a = torch.randn(batch, m, k, device="cuda")
    b = torch.randn(batch, n, k, device="cuda") if form == "nt" else torch.randn(batch, k, n, device="cuda")
    c = torch.full((batch, m, n), float("nan"), device="cuda")
    scr = torch.empty(batch * sk * m * n, device="cuda") if sk > 1 else None
    if form == "nt":
        call("igcn_gemm_f32_batched", m, n, k, batch, ptr(a), k, 1, m * k, ptr(b), k, 1, n * k, ptr(c), m * n, n, sk,
             ptr(scr), stream_ptr())
        want = a.double() @ b.double().transpose(1, 2)
    else:
        call("igcn_gemm_f32_batched", m, n, k, batch, ptr(a), k, 1, m * k, ptr(b), 1, n, k * n, ptr(c), m * n, n, sk,
             ptr(scr), stream_ptr())
        want = a.double() @ b.double()
    assert_matches(c, want.float().cpu().numpy(), 2e-6, "batched product")


@pytest.mark.parametrize("rows,fin,fout", [(512, 64, 32), (4608, 32, 32), (300, 54, 7), (70, 33, 5)])
def test_gemm_group_matches_single_products(rows, fin, fout):
    """ops.gemm_group (igcn_gemm_f32_grouped): the dX / dW pair of a linear layer and a forward pair with bias in one
    launch each — the same bits as the one-product launches (same tiling and K split per problem), also on shapes
    whose rows allow only 8-byte loads, or none (there the entry point falls back to one launch per product)."""
    from igcn_amd import ops
    torch.manual_seed(rows + fin)
    dy = torch.randn(rows, fout, device="cuda")
    x = torch.randn(rows, fin, device="cuda")
    w = torch.randn(fout, fin, device="cuda")
    bias = torch.randn(fout, device="cuda")
    dx, dw = ops.gemm_group([("nn", dy, w, None, None, False), ("tn", dy, x, None, None, False)])
    assert torch.equal(dx, ops.gemm_nn(dy, w)) and torch.equal(dw, ops.gemm_tn(dy, x))
    assert_matches(dx, (dy.double() @ w.double()).float().cpu().numpy(), 2e-6, "dx")
    assert_matches(dw, (dy.double().t() @ x.double()).float().cpu().numpy(), 2e-6, "dw")
    y1, y2 = ops.gemm_group([("nt", x, w, None, bias, False), ("nt", x[: rows // 2], w[: max(1, fout // 2)], None, None, False)])
    assert torch.equal(y1, ops.gemm_nt(x, w, bias, 0))
    assert torch.equal(y2, ops.gemm_nt(x[: rows // 2], w[: max(1, fout // 2)], None, 0))
    # bf16-operand members (configs[4]): the same bits as igcn_gemm_bf16 product by product
    dxb, dwb = ops.gemm_group([("nn", dy, w, None, None, False), ("tn", dy, x, None, None, False)], bf16=True)
    assert torch.equal(dxb, ops.gemm_nn(dy, w, bf16=True)) and torch.equal(dwb, ops.gemm_tn(dy, x, bf16=True))
    yb, = ops.gemm_group([("nt", x, w, None, bias, False, 1)], bf16=True)
    assert torch.equal(yb, ops.gemm_nt(x, w, bias, 1, bf16=True))


@pytest.mark.parametrize("groups,training", [(2, True), (1, True), (2, False)])
def test_node_linear_bn_pair_matches_two_readouts(groups, training):
    """ops.NodeLinearBNPair (conc_for_attention + conc read-outs of one input in paired launches) against two
    ops.NodeLinearBN calls: outputs, running statistics and every gradient bit for bit (same kernel bodies)."""
    from igcn_amd import ops
    torch.manual_seed(11)
    b, f, n, d1 = 64, 5, 403, 32
    x = torch.randn(b, f, n, device="cuda", requires_grad=True)
    keep = (torch.rand(b, n, device="cuda") > 0.5).float() * 2
    par = lambda *s_: torch.randn(*s_, device="cuda", requires_grad=True)          # noqa: E731
    w1, g1, b1, w2, g2, b2 = par(d1, f), par(n), par(n), par(1, f), par(n), par(n)
    assert ops.node_linear_bn_pair_supported(x, w1, w2, None)
    stats = lambda: [torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")]   # noqa: E731
    (rm1, rv1), (rm2, rv2), (qm1, qv1), (qm2, qv2) = stats(), stats(), stats(), stats()
    c1, c2 = torch.randn(b, n, d1, device="cuda"), torch.randn(b, n, 1, device="cuda")
    leaves = [x, w1, g1, b1, w2, g2, b2]
    o1, o2 = ops.NodeLinearBNPair.apply(x, None, w1, g1, b1, rm1, rv1, 0.1, 1e-5, w2, g2, b2, rm2, rv2, 0.1, 1e-5, keep,
                                        training, groups)
    got = torch.autograd.grad((o1 * c1).sum() + (o2 * c2).sum(), leaves)
    z1 = ops.NodeLinearBN.apply(x, w1, g1, b1, qm1, qv1, training, 0.1, 1e-5, groups, None)
    z2 = ops.NodeLinearBN.apply(x, w2, g2, b2, qm2, qv2, training, 0.1, 1e-5, groups, keep)
    want = torch.autograd.grad((z1 * c1).sum() + (z2 * c2).sum(), leaves)
    assert torch.equal(o1, z1) and torch.equal(o2, z2)
    for a_, b_ in ((rm1, qm1), (rv1, qv1), (rm2, qm2), (rv2, qv2)):
        assert torch.equal(a_, b_)
    for g, w_, nm in zip(got, want, ("dx", "dw1", "dg1", "db1", "dw2", "dg2", "db2")):
        if nm == "dx":
            assert_matches(g, w_.cpu().numpy(), 1e-6, nm)       # one sum of the two branches either way
        else:
            assert torch.equal(g, w_), nm


def test_small_linear_pair_matches_two_layers():
    """ops.small_linear_pair (lin2 + lin2_regr in one launch each way) against two ops.linear calls: same bits."""
    from igcn_amd import ops
    torch.manual_seed(9)
    r, k = 512, 64
    x1 = torch.randn(r, k, device="cuda", requires_grad=True)
    x2 = torch.randn(r, k, device="cuda", requires_grad=True)
    w1 = torch.randn(3, k, device="cuda", requires_grad=True)
    w2 = torch.randn(1, k, device="cuda", requires_grad=True)
    b1 = torch.randn(3, device="cuda", requires_grad=True)
    b2 = torch.randn(1, device="cuda", requires_grad=True)
    keep1 = (torch.rand(r, k, device="cuda") > 0.5).float() * 2
    c1, c2 = torch.randn(r, 3, device="cuda"), torch.randn(r, 1, device="cuda")
    leaves = [x1, w1, b1, x2, w2, b2]
    y1, y2 = ops.small_linear_pair(x1, w1, b1, keep1, x2, w2, b2, None)
    got = torch.autograd.grad((y1 * c1).sum() + (y2 * c2).sum(), leaves)
    z1, z2 = ops.linear(x1, w1, b1, keep=keep1), ops.linear(x2, w2, b2)
    want = torch.autograd.grad((z1 * c1).sum() + (z2 * c2).sum(), leaves)
    assert torch.equal(y1, z1) and torch.equal(y2, z2)
    for g, w_, nm in zip(got, want, ("dx1", "dw1", "db1", "dx2", "dw2", "db2")):
        assert torch.equal(g, w_), nm
    ref = (x1 * keep1).double() @ w1.double().t() + b1.double()
    assert_matches(y1, ref.detach().float().cpu().numpy(), 2e-6, "y1 vs fp64")


@pytest.mark.parametrize("rows,k1,k2,relu", [(512, 2912, 3182, True), (37, 34, 70, True), (300, 96, 54, False),
                                             (256, 32, 2, True), (700, 130, 66, True)])
def test_head_bwd_one_pass(ops, rows, k1, k2, relu):
    """igcn_head_bwd_pair behind ops.linear_pair's backward: ReLU mask, bias gradients, input gradients and weight
    gradients of both first layers from one launch, against fp64 — the bench shape, one / two / three row splits with
    ragged ends, column blocks cut by the matrix edge (34, 70, 54, 130, 66, 2 columns), no ReLU; and the same bits when the
    row-split sums are deferred."""
    lib = __import__("igcn_amd")._lib.load()
    assert lib.igcn_head_bwd_supported(rows, 64, k1) and not lib.igcn_head_bwd_supported(rows, 64, 33) \
        and not lib.igcn_head_bwd_supported(rows, 32, k1)
    rng = np.random.default_rng(rows + k1)
    mk = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))     # noqa: E731
    host = [mk(rows, k1), mk(64, k1) / 8, mk(64), mk(rows, k2), mk(64, k2) / 8, mk(64)]
    c1, c2 = mk(rows, 64), mk(rows, 64)

    def run(defer):
        leaves = [t.cuda().requires_grad_(True) for t in host]
        y1, y2 = ops.linear_pair(*leaves, relu=relu)
        if defer:
            with ops.deferred_reductions():
                return torch.autograd.grad((y1 * c1.cuda()).sum() + (y2 * c2.cuda()).sum(), leaves)
        return torch.autograd.grad((y1 * c1.cuda()).sum() + (y2 * c2.cuda()).sum(), leaves)

    got, got_deferred = run(False), run(True)
    ref = [t.double().requires_grad_(True) for t in host]
    act = torch.relu if relu else (lambda t: t)
    z1, z2 = act(ref[0] @ ref[1].t() + ref[2]), act(ref[3] @ ref[4].t() + ref[5])
    want = torch.autograd.grad((z1 * c1.double()).sum() + (z2 * c2.double()).sum(), ref)
    for g, gd, w_, nm in zip(got, got_deferred, want, ("dx1", "dw1", "db1", "dx2", "dw2", "db2")):
        assert_matches(g, w_.numpy(), 5e-6, nm)
        assert torch.equal(g, gd), nm + " deferred"


def test_linear_pair_matches_two_linears(monkeypatch):
    """ops.linear_pair (the heads' first layers as one op, grouped launches both ways) against two ops.linear calls:
    the same bits forward and backward (same kernels, same tiling per product), also with 8-byte-only rows.  [The
    grouped-GEMM backward: the one-pass kernel of test_head_bwd_one_pass is switched off.]"""
    from igcn_amd import ops
    monkeypatch.setenv("IGCN_NO_HEAD_FUSED", "1")
    torch.manual_seed(5)
    for k1, k2 in ((2912, 3182), (96, 54)):
        x1 = torch.randn(512, k1, device="cuda", requires_grad=True)
        x2 = torch.randn(512, k2, device="cuda", requires_grad=True)
        w1 = torch.randn(64, k1, device="cuda", requires_grad=True)
        w2 = torch.randn(64, k2, device="cuda", requires_grad=True)
        b1 = torch.randn(64, device="cuda", requires_grad=True)
        b2 = torch.randn(64, device="cuda", requires_grad=True)
        c1, c2 = torch.randn(512, 64, device="cuda"), torch.randn(512, 64, device="cuda")
        leaves = [x1, w1, b1, x2, w2, b2]
        y1, y2 = ops.linear_pair(x1, w1, b1, x2, w2, b2, relu=True)
        got = torch.autograd.grad((y1 * c1).sum() + (y2 * c2).sum(), leaves)
        z1, z2 = ops.linear(x1, w1, b1, relu=True), ops.linear(x2, w2, b2, relu=True)
        want = torch.autograd.grad((z1 * c1).sum() + (z2 * c2).sum(), leaves)
        assert torch.equal(y1, z1) and torch.equal(y2, z2)
        for g, w_, nm in zip(got, want, ("dx1", "dw1", "db1", "dx2", "dw2", "db2")):
            assert torch.equal(g, w_), nm
        ref = torch.relu(x1.double() @ w1.double().t() + b1.double())
        assert_matches(y1, ref.detach().float().cpu().numpy(), 2e-6, "y1 vs fp64")


@pytest.mark.gpu
def test_loss_head_from_scores_and_partials():
    """The step's form of the loss head: raw class scores (log_softmax inside the kernel, returned as an output),
    un-reduced Gram / regulariser partials as inputs — against F.log_softmax + the term-by-term composite in fp64."""
    import torch.nn.functional as F
    from igcn_amd import ops
    torch.manual_seed(1)
    b, c, nr, s = 37, 3, 3, 54
    dev = "cuda"
    scores = (3 * torch.randn(2 * b, c, device=dev)).requires_grad_(True)
    y = torch.randint(0, c, (b,), device=dev)
    reg = torch.randn(2 * b, nr, device=dev, requires_grad=True)
    clin = torch.rand(b * nr, device=dev)
    x_hat = torch.randn(2 * b, s, device=dev, requires_grad=True)
    snps = torch.rand(b, s, device=dev)
    gram_p = torch.rand(b, 4, device=dev, requires_grad=True)           # rows sum to (consist0, orth0, consist1, orth1)
    prob_p = torch.rand(19, device=dev, requires_grad=True)
    lam, hp_ce, hp_mi = [0.7, 1.0, 0.5, 1.5e-3, 0.1, 0.2], 1.3, 0.8
    loss, terms, logp = ops.LossHead.apply(scores, y, reg, clin, x_hat, snps, gram_p, prob_p, lam, hp_ce, hp_mi, True)
    (loss * 1.7).backward()
    got = [t.grad.clone() for t in (scores, reg, x_hat, gram_p, prob_p)]
    d = lambda t: t.detach().double().requires_grad_(True)          # noqa: E731
    sc2, reg2, x2, gram2, prob2 = d(scores), d(reg), d(x_hat), d(gram_p), d(prob_p)
    logp2 = F.log_softmax(sc2, dim=-1)
    gsum = gram2.sum(0)
    t = [lam[0] * F.nll_loss(logp2[:b], y), lam[0] * F.nll_loss(logp2[b:], y),
         lam[1] * (F.mse_loss(reg2[:b].reshape(-1), clin.double()) + F.mse_loss(reg2[b:].reshape(-1), clin.double())) / 2,
         lam[2] * prob2.sum(),
         lam[3] * (((x2[:b] - snps.double()) ** 2).sum() + ((x2[b:] - snps.double()) ** 2).sum()) / 2,
         lam[4] * (gsum[0] + gsum[2]) / 2, lam[5] * gsum[1]]
    want = hp_ce * t[0] + hp_mi * t[1] + sum(t[2:])
    (want * 1.7).backward()
    assert_matches(logp, logp2.detach().float().cpu().numpy(), 1e-5, "log_softmax")
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    for k in range(7):
        assert abs(float(terms[k]) - float(t[k])) <= 1e-5 * max(1e-3, abs(float(t[k]))), k
    for g, w, name in zip(got, (sc2, reg2, x2, gram2, prob2), ("scores", "reg", "x_hat", "gram", "prob")):
        assert_matches(g, w.grad.float().cpu().numpy(), 1e-5, "grad " + name, floor=1e-7)
    # lam[0] == 0: the class scores are not read for the loss and their gradient is exactly zero, even when non-finite
    bad = scores.detach().clone()
    bad[0, 0] = float("inf")
    bad.requires_grad_(True)
    lam0 = [0.0] + lam[1:]
    loss0, _, _ = ops.LossHead.apply(bad, y, reg.detach(), clin, x_hat.detach(), snps, gram_p.detach(), prob_p.detach(),
                                     lam0, hp_ce, hp_mi, True)
    loss0.backward()
    assert bool(torch.isfinite(loss0)) and float(bad.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("from_logits", [True, False])
def test_loss_head_forward_writes_the_unit_upstream_gradients(monkeypatch, from_logits):
    """With the train step's cached d loss / d loss = 1 as the upstream gradient, ops.LossHead returns the gradients its
    FORWARD wrote (igcn_loss_head_fwd_grads) and launches no backward kernel: the same bits as igcn_loss_head_bwd, and
    any other upstream (here 1.7, or a fresh ones tensor) still takes the backward kernel."""
    from igcn_amd import _lib, ops
    from igcn_amd.train import _unit_grad
    torch.manual_seed(2)
    b, c, nr, s = 37, 3, 3, 54
    dev = "cuda"
    scores = 3 * torch.randn(2 * b, c, device=dev)
    if not from_logits:
        scores = torch.log_softmax(scores, -1)
    y = torch.randint(0, c, (b,), device=dev)
    ins = [scores, torch.randn(2 * b, nr, device=dev), torch.randn(2 * b, s, device=dev), torch.rand(b, 4, device=dev),
           torch.rand(19, device=dev)]
    clin, snps = torch.rand(b * nr, device=dev), torch.rand(b, s, device=dev)
    lam, hp_ce, hp_mi = [0.7, 1.0, 0.5, 1.5e-3, 0.1, 0.2], 1.3, 0.8

    def grads(upstream):
        leaves = [t.clone().requires_grad_(True) for t in ins]
        out = ops.LossHead.apply(leaves[0], y, leaves[1], clin, leaves[2], snps, leaves[3], leaves[4], lam, hp_ce, hp_mi,
                                 from_logits)
        return torch.autograd.grad(out[0], leaves, grad_outputs=upstream)

    unit = _unit_grad(torch.zeros((), device=dev))
    assert unit.data_ptr() in ops.UNIT_GRAD_PTRS
    calls = []
    real = ops.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    monkeypatch.setattr(ops, "call", spy)
    fused = grads(unit)
    assert "igcn_loss_head_fwd_grads" in calls and "igcn_loss_head_bwd" not in calls
    calls.clear()
    fresh = grads(torch.ones((), device=dev))                      # same value, another tensor: the backward kernel
    assert "igcn_loss_head_bwd" in calls
    scaled = grads(1.7 * torch.ones((), device=dev))
    monkeypatch.setenv("IGCN_NO_LOSS_HEAD_FUSED", "1")
    calls.clear()
    plain = grads(unit)
    assert "igcn_loss_head_fwd_grads" not in calls and "igcn_loss_head_bwd" in calls
    for a, f, p_, sc, nm in zip(fused, fresh, plain, scaled, ("dscores", "dreg", "dxhat", "dgram", "dprob")):
        assert torch.equal(a, f) and torch.equal(a, p_), nm
        assert_matches(sc, (1.7 * a.double()).cpu().numpy(), 1e-6, nm + " x 1.7")


_AB_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import igcn_amd
from igcn_amd import ops, synth
rng = np.random.default_rng(3)
# GO attention layer (2 -> 5 features) on a small hierarchy
go_snps, adj, pool_dim = synth.go_hierarchy((60, 30, 20, 9, 1), seed=4)
a = torch.from_numpy(adj.T.copy()).to_sparse().coalesce()
idx = a.indices()
csr = ops.Csr(idx[0], idx[1], adj.shape[0], adj.shape[0], "cuda")
n = adj.shape[0]
x = torch.from_numpy(rng.standard_normal((5, 2, n))).float().cuda().requires_grad_(True)
par = [torch.from_numpy(rng.standard_normal(s)).float().cuda().requires_grad_(True) for s in ((5, 2), (5, 2), (10,), (5,))]
y = ops.GoAttention.apply(x, par[0], par[1], par[2], par[3], csr)
cot = torch.from_numpy(rng.standard_normal(tuple(y.shape))).float().cuda()
g = torch.autograd.grad((y * cot).sum(), [x] + par)
# GO decoder layer (5 -> 2) on the same hierarchy: rows = the 120 nodes, cols = the top 60 (LDS-resident by default,
# global-memory kernels under IGCN_GO_ATTN_CM=1)
ad = torch.from_numpy(adj[:, 60:].copy()).to_sparse().coalesce()
di = ad.indices()
dcsr = ops.Csr(di[0], di[1], n, n - 60, "cuda")
xd = torch.from_numpy(rng.standard_normal((5, 5, n - 60))).float().cuda().requires_grad_(True)
pd = [torch.from_numpy(rng.standard_normal((2, 5))).float().cuda().requires_grad_(True) for _ in range(2)]
yd = ops.GoDecode.apply(xd, pd[0], pd[1], dcsr)
cd = torch.from_numpy(rng.standard_normal(tuple(yd.shape))).float().cuda()
gd = torch.autograd.grad((yd * cd).sum(), [xd] + pd)
# SNP -> GO map (C = 2) on the CSR kernels (LDS-tiled by default, the first kernels under IGCN_SPMM_NO_LDS=1)
ops.SparseMap.DENSE_LIMIT = 0
gidx = torch.from_numpy((rng.random((n, 54)) < 0.06).astype(np.float32)).to_sparse().coalesce().indices()
gcsr = ops.Csr(gidx[0], gidx[1], n, 54, "cuda")
xs = torch.from_numpy(rng.random((5, 54))).float().cuda().requires_grad_(True)
vs = torch.from_numpy(rng.standard_normal((2, gidx.shape[1]))).float().cuda().requires_grad_(True)
ys = ops.SparseMap.apply(xs, gcsr, vs)
cs = torch.from_numpy(rng.standard_normal(tuple(ys.shape))).float().cuda()
gs = torch.autograd.grad((ys * cs).sum(), [xs, vs])
# attention core, head_dim 16
q = torch.from_numpy(rng.standard_normal((3, 40, 32))).float().cuda().requires_grad_(True)
kv = torch.from_numpy(rng.standard_normal((3, 70, 64))).float().cuda().requires_grad_(True)
o = ops.AttentionCore.apply(q, kv, 2)
co = torch.from_numpy(rng.standard_normal((3, 40, 32))).float().cuda()
ga = torch.autograd.grad((o * co).sum(), [q, kv])
# dense but NOT complete graphs (2 x 70 nodes, ~90 % of the pairs): edge mask + gcn_norm + scatter-aggregate on the tiled
# record-stream kernels of csrc/sgcn.hip (lanes-over-nodes tiled walks, LDS-staged aggregation by default; the wave-per-list
# / wave-per-target kernels under IGCN_NO_TILED_LISTS=1 / IGCN_PROPAGATE_NO_LDS=1)
gq, rq = 2, 70
src, dst = [], []
for k in range(gq):
    rr, cc = np.nonzero(rng.random((rq, rq)) < 0.9)
    src.append(rr + k * rq), dst.append(cc + k * rq)
ei = torch.from_numpy(np.vstack([np.concatenate(src), np.concatenate(dst)])).long().cuda()
plan = ops.GraphPlan(ei, gq * rq)
plan.nodes_per_graph = rq
xq = torch.from_numpy(rng.random((gq * rq, 3))).float().cuda().requires_grad_(True)
pq = torch.from_numpy(rng.standard_normal((rq, 3))).float().cuda().requires_grad_(True)
pbq = torch.from_numpy(rng.standard_normal((6, 1))).float().cuda().requires_grad_(True)
ewq = torch.from_numpy(rng.random(ei.shape[1]) + 0.05).float().cuda().requires_grad_(True)
hq = torch.from_numpy(rng.standard_normal((gq * rq, 16))).float().cuda().requires_grad_(True)
bq = torch.from_numpy(rng.standard_normal(16)).float().cuda().requires_grad_(True)
xm, ewm, e = ops.EdgeMask.apply(xq, pq, pbq, ewq, plan, rq)
coef = ops.GcnNorm.apply(ewm, plan)
oq = ops.GcnPropagate.apply(hq, coef[0], coef[1], bq, plan, True, coef[2], coef[3])
cq = torch.from_numpy(rng.standard_normal(tuple(oq.shape))).float().cuda()
gq_ = torch.autograd.grad((oq * cq).sum() + (xm * xm).sum(), [xq, pq, pbq, ewq, hq, bq], allow_unused=True)
gq_ = [t if t is not None else torch.zeros(1, device="cuda") for t in gq_]
np.savez(sys.argv[2], y=y.detach().cpu().numpy(), o=o.detach().cpu().numpy(), yd=yd.detach().cpu().numpy(),
         oq=oq.detach().cpu().numpy(), **{f"q{i}": t.cpu().numpy() for i, t in enumerate(gq_)},
         **{f"g{i}": t.cpu().numpy() for i, t in enumerate(g)}, **{f"a{i}": t.cpu().numpy() for i, t in enumerate(ga)},
         **{f"d{i}": t.cpu().numpy() for i, t in enumerate(gd)}, ys=ys.detach().cpu().numpy(),
         **{f"s{i}": t.cpu().numpy() for i, t in enumerate(gs)})
"""


def test_alternative_kernel_variants_agree(tmp_path):
    """The A/B switches stay honest: the global-memory GO kernels (attention backward, decoder forward / backward:
    IGCN_GO_ATTN_CM=1 — also the fallbacks for hierarchies too large for LDS), the first CSR map kernels
    (IGCN_SPMM_NO_LDS=1 — the fallback for structures whose operand rows do not fit LDS) and, on a dense-but-not-complete
    batch, the wave-per-list / wave-per-target forms of the record-stream kernels (IGCN_NO_TILED_LISTS=1,
    IGCN_PROPAGATE_NO_LDS=1) give the numbers of the default kernels.  The switches are read once per process, so each variant runs in a short child process (one
    at a time)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "ab.py"
    script.write_text(_AB_SCRIPT)
    outs = {}
    for tag, env in (("default", {}), ("alt", {"IGCN_GO_ATTN_CM": "1", "IGCN_SPMM_NO_LDS": "1", "IGCN_NO_TILED_LISTS": "1",
                                               "IGCN_PROPAGATE_NO_LDS": "1"})):
        out = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, str(script), ROOT, str(out)], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    for k in outs["default"].files:
        a, b = outs["default"][k], outs["alt"][k]
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), k


# ------------------------------------------------------------------------------------------------ graph pool
@pytest.mark.parametrize("g,r,d,seed", [(3, 10, 8, 0), (32, 90, 32, 1), (5, 1, 4, 2), (7, 13, 5, 3)])
def test_graph_pool_mean_max_add(ops, g, r, d, seed):
    """igcn_graph_pool_* against torch on the [G,R,D] view (kernel/sgcn_img_snp.py:230-235): ties in the max (ReLU
    zeros) send the gradient to the first maximal node, as torch-scatter's CPU scatter_max does."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((g * r, d))).float()
    x = torch.relu(x)                                   # exact zeros => ties
    x[:r, 0] = 0.0                                      # an all-tied column
    xg = x.cuda().requires_grad_(True)
    out = ops.GraphPool.apply(xg, r)
    xr = x.double().requires_grad_(True)
    v = xr.view(g, r, d)
    mx = v.max(dim=1).values
    first = (v == mx.unsqueeze(1)).double().argmax(dim=1)
    ref = torch.cat([v.mean(1), v.gather(1, first.unsqueeze(1)).squeeze(1), v.sum(1)], dim=1)
    assert_matches(out, ref.detach().float().numpy(), 1e-6, "pool")
    cot = torch.from_numpy(rng.standard_normal((g, 3 * d))).float()
    (out * cot.cuda()).sum().backward()
    (ref * cot.double()).sum().backward()
    assert_matches(xg.grad, xr.grad.float().numpy(), 1e-6, "pool grad")


# ------------------------------------------------------------------------------------------------ fused SGCN stack
@pytest.mark.parametrize("g,r,h0,f,layers,deg,loops", [(4, 90, 3, 16, 2, 3, "all"), (3, 10, 3, 4, 2, 3, "some"),
                                                       (2, 33, 5, 8, 3, 6, "none"), (5, 17, 2, 32, 1, 4, "multi"),
                                                       (2, 90, 3, 16, 4, 5, "some"),
                                                       # fewer nodes than features: the weight-gradient partials of a
                                                       # layer do not fit its (dead) transform buffer
                                                       (3, 9, 3, 16, 2, 3, "some"), (2, 20, 3, 32, 3, 4, "all")])
def test_fused_sgcn_stack_fwd_bwd(ops, g, r, h0, f, layers, deg, loops):
    """igcn_sgcn_stack_* (gcn_norm + L x GCNConv + ReLU + concatenation, LDS-resident, one workgroup per graph) against
    the fp64 PyG restatement: outputs, dx, d(edge weight), every dW / db.  Stored self-loops (kept weight), nodes
    without a stored loop (added with weight 1), several stored loops on one node (last wins), isolated nodes."""
    from igcn_amd.data import Batch, Data
    from oracle import pyg_ops
    rng = np.random.default_rng(g * 100 + r + f)
    graphs = []
    for _ in range(g):
        src = rng.integers(0, r, deg * r)
        dst = rng.integers(0, r, deg * r)
        keep = src != dst
        src, dst = src[keep], dst[keep]
        src, dst = src[dst != r - 1], dst[dst != r - 1]            # node r-1 has no incoming edge
        if loops in ("all", "some", "multi"):
            nodes = np.arange(r) if loops == "all" else rng.choice(r, r // 2, replace=False)
            src, dst = np.concatenate([src, nodes]), np.concatenate([dst, nodes])
        if loops == "multi":
            src, dst = np.concatenate([src, [2, 2]]), np.concatenate([dst, [2, 2]])
        order = rng.permutation(src.size)
        ei = torch.from_numpy(np.vstack([src[order], dst[order]])).long()
        graphs.append(Data(x=torch.from_numpy(rng.random((r, h0))).float(), edge_index=ei,
                           edge_attr=torch.from_numpy(rng.random(ei.shape[1]) + 0.05).float()))
    batch = Batch.from_data_list(graphs).to("cuda")
    plan = ops.plan_for(batch)
    plan.check()
    assert ops.sgcn_stack_supported(plan, r, h0, f, layers)
    ws = [torch.from_numpy(rng.standard_normal((f, h0 if l == 0 else f)) / np.sqrt(h0 if l == 0 else f)).float()
          for l in range(layers)]
    bs = [torch.from_numpy(0.3 * rng.standard_normal(f)).float() for _ in range(layers)]
    cot = torch.from_numpy(rng.standard_normal((g * r, layers * f))).float()
    # oracle, fp64
    xd = batch.x.cpu().double().requires_grad_(True)
    ewd = batch.edge_attr.cpu().double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    bd = [b.double().requires_grad_(True) for b in bs]
    ei_c = batch.edge_index.cpu()
    hs, hcur = [], xd
    for l in range(layers):
        hcur = torch.relu(pyg_ops.gcn_conv(hcur, ei_c, ewd, wd[l], bd[l]))
        hs.append(hcur)
    want = torch.cat(hs, dim=1)
    (want * cot.double()).sum().backward()
    # HIP
    xg = batch.x.clone().requires_grad_(True)
    ewg = batch.edge_attr.clone().requires_grad_(True)
    wg = [w.cuda().requires_grad_(True) for w in ws]
    bg = [b.cuda().requires_grad_(True) for b in bs]
    out = ops.SgcnStack.apply(xg, ewg, plan, r, *[t for pair in zip(wg, bg) for t in pair])
    (out * cot.cuda()).sum().backward()
    assert_matches(out, want.detach().numpy(), TOL, "xcat")
    assert_matches(xg.grad, xd.grad.numpy(), TOL, "dx")
    if loops != "multi":          # several stored loops on one node: index_put's duplicate-index gradient is undefined
        assert_matches(ewg.grad, ewd.grad.numpy(), TOL, "dew")
    for l in range(layers):
        assert_matches(wg[l].grad, wd[l].grad.numpy(), TOL, f"dW{l}", floor=1e-6)
        assert_matches(bg[l].grad, bd[l].grad.numpy(), TOL, f"db{l}", floor=1e-6)


@pytest.mark.parametrize("g,r,h0,f,layers,deg,loops", [(5, 90, 3, 16, 2, 3, "some"), (3, 24, 1, 8, 3, 4, "multi"),
                                                      (4, 270, 1, 4, 3, 3, "none"), (6, 40, 3, 32, 1, 2, "all")])
def test_front_kernel_equals_plan_build_mask_and_stack(ops, g, r, h0, f, layers, deg, loops):
    """igcn_sgcn_front_fwd — the plan of the batch and of its 2-copy replica, the stacked (plain | masked) inputs, the edge
    mask, loss_probability, the SNP mask and the GCNConv stack of both passes in ONE launch — against the three launches it
    replaces (igcn_graph_plan_build_segmented_rep, igcn_edge_mask_fwd_reg, igcn_sgcn_stack_fwd): every plan array
    bit-exact (edge_index indexing), x_in / ew_in / e / the SNP mask / xcat bit-exact (same arithmetic in the same order),
    the regulariser to fp32 rounding (its partial sums are grouped per graph instead of per 256 edges); then the gradients
    of the combined op against those of the chain (same two backward launches on the plan the front kernel wrote)."""
    from igcn_amd.data import Batch, Data
    rng = np.random.default_rng(g * 1000 + r + f)
    graphs = []
    for _ in range(g):
        src = rng.integers(0, r, deg * r)
        dst = rng.integers(0, r, deg * r)
        keep = src != dst
        src, dst = src[keep], dst[keep]
        if loops in ("all", "some", "multi"):
            nodes = np.arange(r) if loops == "all" else rng.choice(r, r // 2, replace=False)
            src, dst = np.concatenate([src, nodes]), np.concatenate([dst, nodes])
        if loops == "multi":
            src, dst = np.concatenate([src, [2, 2]]), np.concatenate([dst, [2, 2]])
        order = rng.permutation(src.size)
        ei = torch.from_numpy(np.vstack([src[order], dst[order]])).long()
        graphs.append(Data(x=torch.from_numpy(rng.random((r, h0))).float(), edge_index=ei,
                           edge_attr=torch.from_numpy(rng.random(ei.shape[1]) + 0.05).float()))
    batch = Batch.from_data_list(graphs).to("cuda")
    n, ne = batch.x.shape[0], batch.edge_index.shape[1]
    hp = (0.1, 0.2, 0.15, 0.05, 1e-6)
    t = lambda a: torch.from_numpy(np.asarray(a)).float().cuda()                     # noqa: E731
    prob, pb, snps = t(rng.standard_normal((r, h0))), t(rng.standard_normal((2 * h0, 1))), t(rng.standard_normal((1, 54)))
    feat = t(rng.random((g, 54)))
    ws = [t(rng.standard_normal((f, h0 if l == 0 else f)) / np.sqrt(h0 if l == 0 else f)) for l in range(layers)]
    bs = [t(0.3 * rng.standard_normal(f)) for _ in range(layers)]
    cot = t(rng.standard_normal((2 * n, layers * f)))
    cot_e, cot_s = t(rng.standard_normal(ne)), t(rng.standard_normal((2 * g, 54)))
    names = ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge")

    def run(front):
        b = Batch.from_data_list(graphs).to("cuda")
        plan = ops.plan_for(b)
        plan_g = plan.replicate(2)
        if front:                                        # the plan arrays are the front kernel's to fill: poison them
            for p_ in (plan, plan_g):
                for k in names:
                    getattr(p_, k).fill_(-7)
            plan.rebuild(b.edge_index, lazy=True)
            assert plan._pending_build is not None
        leaves = [v.clone().requires_grad_(True) for v in (b.x, prob, pb, b.edge_attr, snps, *ws, *bs)]
        x, pr, pbv, ew, sn = leaves[:5]
        wl, bl = leaves[5:5 + layers], leaves[5 + layers:]
        wb = [v for pair in zip(wl, bl) for v in pair]
        if front:
            assert ops.sgcn_front_supported(plan, r, h0, f, layers, feat, sn)
            xcat, _, e, regp, full = ops.SgcnFront.apply(x, pr, pbv, ew, plan, r, sn, hp, feat, b.edge_index, *wb)
            assert plan._pending_build is None
        else:
            x_in, ew_in, e, regp, full = ops.EdgeMaskStacked.apply(x, pr, pbv, ew, plan, r, sn, hp, feat)
            xcat = ops.SgcnStack.apply(x_in, ew_in, plan_g, r, *wb)
        loss = (xcat * cot).sum() + (e * cot_e).sum() + 3.0 * regp.sum() + (full * cot_s).sum()
        loss.backward()
        plan.check()
        arrays = {("p", k): getattr(plan, k).clone() for k in names}
        arrays.update({("r", k): getattr(plan_g, k).clone() for k in names})
        return (xcat.detach(), e.detach(), regp.detach().sum(), full.detach()), [v.grad for v in leaves], arrays

    (xc0, e0, reg0, full0), g0, a0 = run(False)
    (xc1, e1, reg1, full1), g1, a1 = run(True)
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k
    assert torch.equal(xc0, xc1) and torch.equal(e0, e1) and torch.equal(full0, full1)
    assert abs(float(reg0) - float(reg1)) <= 1e-6 * max(1.0, abs(float(reg0)))
    for i, (a, b_) in enumerate(zip(g0, g1)):
        assert (a is None) == (b_ is None), i
        if a is not None:
            assert_matches(b_, a.cpu().numpy(), 2e-6, f"grad of leaf {i}", floor=1e-6)


# ------------------------------------------------------------------------------------------------ fused dropout
def test_dropout_masks_one_launch(ops):
    """igcn_dropout_masks: factors in {0, 1/(1-p)} per site, keep rate ~ 1-p, fresh masks on every launch AND on every
    replay of a captured launch (the kernel advances its own device-side counter)."""
    torch.manual_seed(123)
    state = ops.DropoutState(torch.device("cuda"))
    sites = [((64, 3000), 0.4), ((64, 401), 0.5), ((7,), 0.3), ((128, 64), 0.3)]
    m1 = ops.dropout_masks(sites, state)
    m2 = ops.dropout_masks(sites, state)
    for (shape, p), a, b in zip(sites, m1, m2):
        assert tuple(a.shape) == tuple(shape)
        vals = torch.unique(a)
        assert all(abs(float(v)) < 1e-12 or abs(float(v) - 1.0 / (1.0 - p)) < 1e-6 for v in vals), (p, vals)
        if a.numel() > 1000:
            rate = float((a > 0).float().mean())
            assert abs(rate - (1.0 - p)) < 4.0 * (p * (1 - p) / a.numel()) ** 0.5 + 1e-3, (p, rate)
            assert float((a != b).float().mean()) > 0.2          # a second launch draws different masks
            # no visible structure along rows / columns
            assert float(((a > 0).float().mean(0) - (1 - p)).abs().max()) < 0.35
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        mg = ops.dropout_masks(sites, state)
    g.replay()
    first = mg[0].clone()
    g.replay()
    torch.cuda.synchronize()
    assert float((first != mg[0]).float().mean()) > 0.2          # replays advance the counter
    assert int(state.state[1:].abs().sum().item()) == 0          # every arrival word is back at 0
    big = ops.dropout_masks([((2048, 3000), 0.4)], state)[0]     # more workgroups than the launch cap: grid-stride
    assert abs(float((big > 0).float().mean()) - 0.6) < 2e-3 and int(state.state[1:].abs().sum().item()) == 0
    # more sites than one launch takes (a GO hierarchy with many levels): split over launches, same contract per site
    many = [((33, 40 + k), 0.1 + 0.02 * k) for k in range(40)]
    mm = ops.dropout_masks(many, state)
    assert len(mm) == 40
    for (shape, p), a in zip(many, mm):
        assert tuple(a.shape) == tuple(shape)
        assert all(abs(float(v)) < 1e-12 or abs(float(v) - 1.0 / (1.0 - p)) < 1e-5 for v in torch.unique(a)), p
        assert abs(float((a > 0).float().mean()) - (1 - p)) < 0.12


def test_dropout_masks_are_a_pure_function_of_counter_and_index(ops):
    """The mask generator's contract (first half): a factor depends on (stream counter, flat element index,
    the site's p) only — ``oracle/dropout.py`` rebuilds the arrays of any launch on the host, bit for bit: a launch of its
    own, the next launch (counter + 1), and a launch that RIDES in another kernel's grid."""
    from oracle import dropout as OD
    state = ops.DropoutState(torch.device("cuda"))
    sites = [((64, 3001), 0.4), ((9, 401), 0.5), ((7,), 0.3), ((128, 64), 0.3), ((2, 5, 33), 0.1)]
    for _ in range(2):
        c = int(state.state[0].item())
        got = ops.dropout_masks(sites, state)
        torch.cuda.synchronize()
        assert int(state.state[0].item()) == c + 1
        for g, w in zip(got, OD.masks(sites, c)):
            assert np.array_equal(g.cpu().numpy(), w)
    state.state[0] = (1 << 40) + 12345                           # the high word of the counter takes part
    c = int(state.state[0].item())
    got = ops.dropout_masks(sites[:2], state)
    for g, w in zip(got, OD.masks(sites[:2], c)):
        assert np.array_equal(g.cpu().numpy(), w)
    c = int(state.state[0].item())
    got = ops.dropout_masks(sites, state, ride=True)             # queued: nothing launched yet
    ops.call("igcn_rider_flush", ops.stream_ptr())
    torch.cuda.synchronize()
    for g, w in zip(got, OD.masks(sites, c)):
        assert np.array_equal(g.cpu().numpy(), w)


def test_dropout_launch_advances_the_batch_counters(ops):
    """igcn_dropout_masks with counters: BatchNorm's num_batches_tracked words advance by `inc` per launch — eagerly and
    on every replay of a captured launch — and nothing else of the contract changes."""
    state = ops.DropoutState(torch.device("cuda"))
    cnt = [torch.tensor(v, dtype=torch.int64, device="cuda") for v in (0, 7, 100)]
    sites = [((16, 50), 0.5), ((33,), 0.25)]
    m = ops.dropout_masks(sites, state, cnt, 2)
    assert [int(c) for c in cnt] == [2, 9, 102] and tuple(m[0].shape) == (16, 50)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.dropout_masks(sites, state, cnt, 2)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert [int(c) for c in cnt] == [8, 15, 108]


def test_dropout_masks_ride_in_the_plan_builds_launch(ops):
    """``dropout_masks(ride=True)``: the job is queued for the stream and drawn by extra workgroups of the next per-graph
    plan build's grid (k_plan_segmented_ride) — the SAME masks a launch of its own draws from the same counter, the
    counters bumped once, the plan the same bits; a job nobody carried is launched by igcn_rider_flush; a second job on
    a stream that still holds one is refused."""
    from igcn_amd import _lib, synth
    from igcn_amd._lib import call, stream_ptr
    batch = synth.brain_batch(24, seed=3, rois=90).to("cuda")
    sites = [((48, 3000), 0.4), ((48, 401), 0.5), ((7,), 0.3), ((96, 64), 0.3)]
    torch.manual_seed(9)
    st_a, st_b = ops.DropoutState(torch.device("cuda")), ops.DropoutState(torch.device("cuda"))
    st_b.state.copy_(st_a.state)                                       # the same counter: the same masks
    cnt_a = [torch.tensor(5, dtype=torch.int64, device="cuda")]
    cnt_b = [torch.tensor(5, dtype=torch.int64, device="cuda")]
    want = ops.dropout_masks(sites, st_a, cnt_a, 2)                    # a launch of its own
    plan0 = ops.plan_for(batch)
    ref = {n: getattr(plan0, n).clone() for n in ("src32", "dst32", "tgt_ptr", "tgt_perm", "src_ptr", "src_perm", "loop_edge")}
    got = ops.dropout_masks(sites, st_b, cnt_b, 2, ride=True)          # queued ...
    torch.cuda.synchronize()
    assert int(cnt_b[0]) == 5                                          # ... nothing has run yet
    plan0.rebuild(batch.edge_index)                                    # ... carried by this build
    torch.cuda.synchronize()
    assert int(cnt_b[0]) == 7 == int(cnt_a[0])
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert torch.equal(st_a.state, st_b.state)                         # counter advanced once, arrival words back at 0
    for n, t in ref.items():
        assert torch.equal(getattr(plan0, n), t), n
    plan0.check()
    # nobody carries it: the flush launches it
    w2 = ops.dropout_masks(sites, st_a)
    g2 = ops.dropout_masks(sites, st_b, ride=True)
    call("igcn_rider_flush", stream_ptr())
    call("igcn_rider_flush", stream_ptr())                             # nothing waiting: nothing happens
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(w2, g2))
    ops.dropout_masks(sites, st_b, ride=True)
    with pytest.raises(_lib.IgcnError):
        ops.dropout_masks(sites, st_b, ride=True)                      # one job per stream
    call("igcn_rider_flush", stream_ptr())
    torch.cuda.synchronize()
    # a job of a step that failed half way is FORGOTTEN, not launched (its buffers may be gone)
    before = st_b.state.clone()
    ops.dropout_masks(sites, st_b, cnt_b, 2, ride=True)
    call("igcn_rider_cancel", stream_ptr())
    plan0.rebuild(batch.edge_index)
    call("igcn_rider_flush", stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(st_b.state, before) and int(cnt_b[0]) == 7
    ops.dropout_masks(sites, st_b, ride=True)                          # (the stream takes a new job)
    call("igcn_rider_flush", stream_ptr())
    torch.cuda.synchronize()


def test_grad_fan_sums_the_consumers_gradients_in_one_launch(ops):
    """ops.GradFan: k aliases of a tensor, the k incoming gradients summed by igcn_sum_n (16-byte path and scalar tail),
    a missing consumer gradient skipped, unaligned gradients summed by the fallback — all equal to autograd's own adds."""
    rng = np.random.default_rng(3)
    for shape in ((90, 3), (512, 5, 400), (1, 54), (7,)):
        t = torch.from_numpy(rng.standard_normal(shape)).float().cuda().requires_grad_(True)
        w = [torch.from_numpy(rng.standard_normal(shape)).float().cuda() for _ in range(3)]
        a, b, c = ops.GradFan.apply(t, 3)
        ((a * w[0]).sum() + (b * w[1]).sum() + (c * w[2]).sum()).backward()
        assert_matches(t.grad, (w[0].double() + w[1].double() + w[2].double()).cpu().numpy(), 1e-6, "three consumers")
        t.grad = None
        a, b, c = ops.GradFan.apply(t, 3)
        ((a * w[0]).sum() + (c * w[2]).sum()).backward()                  # the second alias is never used
        assert_matches(t.grad, (w[0].double() + w[2].double()).cpu().numpy(), 1e-6, "two of three")
        t.grad = None
    # a gradient that is a misaligned view: the fallback adds
    t = torch.zeros(8, device="cuda", requires_grad=True)
    a, b = ops.GradFan.apply(t, 2)
    base = torch.arange(9, dtype=torch.float32, device="cuda")
    torch.autograd.backward([a, b], [base[1:], torch.ones(8, device="cuda")])
    assert torch.equal(t.grad, base[1:] + 1)


def test_fused_stack_adds_a_second_consumers_gradient_on_load(ops):
    """ops.SgcnStack with the dual output (negative rois): two autograd handles of one buffer, whose gradients the
    backward kernel adds while staging — equal to the single-output op fed the sum."""
    from igcn_amd import synth
    rng = np.random.default_rng(8)
    b = synth.brain_batch(5, seed=2, rois=90).to("cuda")
    plan = ops.plan_for(b)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()             # noqa: E731
    ws = [t(16, 3) * 0.5, t(16) * 0.1, t(16, 16) * 0.3, t(16) * 0.1]
    g1, g2 = t(450, 32), t(450, 32)
    res = []
    for dual in (True, False):
        x = b.x.clone().requires_grad_(True)
        par = [w.clone().requires_grad_(True) for w in ws]
        if dual:
            y1, y2 = ops.SgcnStack.apply(x, b.edge_attr, plan, -90, *par)
            assert y1.data_ptr() == y2.data_ptr()
            torch.autograd.backward([y1, y2], [g1, g2])
        else:
            y = ops.SgcnStack.apply(x, b.edge_attr, plan, 90, *par)
            y.backward(g1 + g2)
        res.append([x.grad] + [p.grad for p in par])
    for a, c in zip(*res):
        assert_matches(a, c.cpu().numpy(), 2e-6, "dual vs summed")


def test_consumers_apply_the_dropout_factors(ops):
    """`keep` inside igcn_small_linear_*, igcn_bn1d_* and igcn_node_linear_bn_* (D = 1) equals a multiply in front of /
    behind the unfused op, forward and backward."""
    rng = np.random.default_rng(5)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()          # noqa: E731
    keepf = lambda *s: (torch.from_numpy((rng.random(s) > 0.5) * 2.0).float().cuda())   # noqa: E731
    # --- narrow linear: y = (x * keep) W^T + b
    x, w, b, k = t(512, 64).requires_grad_(True), t(3, 64).requires_grad_(True), t(3).requires_grad_(True), keepf(512, 64)
    y = ops.linear(x, w, b, keep=k)
    cot = t(512, 3)
    g = torch.autograd.grad((y * cot).sum(), (x, w, b))
    xr, wr, br = (v.detach().double().requires_grad_(True) for v in (x, w, b))
    yr = (xr * k.double()) @ wr.t() + br
    gr = torch.autograd.grad((yr * cot.double()).sum(), (xr, wr, br))
    assert_matches(y, yr.detach().cpu().numpy(), TOL, "small_linear keep")
    for a, c, nm in zip(g, gr, ("dx", "dW", "db")):
        assert_matches(a, c.cpu().numpy(), TOL, "small_linear keep " + nm)
    # --- BatchNorm1d + ReLU + dropout
    x, ga, be = t(256, 32).requires_grad_(True), (1 + 0.2 * t(32)).requires_grad_(True), t(32).requires_grad_(True)
    k = keepf(256, 32)
    rm, rv = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
    y = ops.BatchNorm1dGrouped.apply(x, ga, be, rm, rv, True, 0.1, 1e-5, True, 2, k)
    y0 = ops.BatchNorm1dGrouped.apply(x, ga, be, rm.clone(), rv.clone(), True, 0.1, 1e-5, True, 2, None)
    cot = t(256, 32)
    g = torch.autograd.grad((y * cot).sum(), (x, ga, be))
    g0 = torch.autograd.grad((y0 * (cot * k)).sum(), (x, ga, be))
    assert_matches(y, (y0 * k).detach().cpu().numpy(), 1e-6, "bn1d keep")
    for a, c, nm in zip(g, g0, ("dx", "dgamma", "dbeta")):
        assert_matches(a, c.cpu().numpy(), 1e-5, "bn1d keep " + nm)
    # --- node-wise linear + BatchNorm(#nodes) + ReLU + dropout, D = 1
    bsz, f, n = 64, 5, 400
    x, w = t(bsz, f, n).requires_grad_(True), t(1, f).requires_grad_(True)
    ga, be = (1 + 0.2 * t(n)).requires_grad_(True), (0.1 * t(n)).requires_grad_(True)
    k = keepf(bsz, n)
    rm, rv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
    y = ops.NodeLinearBN.apply(x, w, ga, be, rm, rv, True, 0.1, 1e-5, 2, k)
    y0 = ops.NodeLinearBN.apply(x, w, ga, be, rm.clone(), rv.clone(), True, 0.1, 1e-5, 2, None)
    cot = t(bsz, n, 1)
    g = torch.autograd.grad((y * cot).sum(), (x, w, ga, be))
    g0 = torch.autograd.grad((y0 * (cot * k.unsqueeze(2))).sum(), (x, w, ga, be))
    assert_matches(y, (y0 * k.unsqueeze(2)).detach().cpu().numpy(), 1e-6, "nlbn keep")
    for a, c, nm in zip(g, g0, ("dx", "dW", "dgamma", "dbeta")):
        assert_matches(a, c.cpu().numpy(), 1e-5, "nlbn keep " + nm, floor=1e-6)
