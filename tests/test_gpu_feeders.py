"""The feeders in front of the train step on the GPU: every batch they hand over is, bit for bit, what the
reference's loop would have built — ``Batch.from_data_list([dataset[i] for i in idx])`` (batch.py:24-123) for the subjects
``idx`` of the epoch order (``DataLoader(shuffle=True)`` semantics: every subject once per epoch), and for the dense
feeder the per-subject ``preprocess_diffusion_imgs_snps`` (util_gdc.py:71-101) in front of it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y", "batch", "ptr", "edge_ptr")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from igcn_amd import _lib
    _lib.load()


@pytest.fixture(scope="module")
def subjects():
    from igcn_amd import synth
    from igcn_amd.data import Data
    graphs = synth.brain_graph_list(40, seed=5, rois=90, tsne_dim=16)
    slim = [Data(**{k: v for k, v in g.__dict__.items() if k != "A"}) for g in graphs]
    return graphs, slim


def _same(got, want, keys=KEYS):
    for k in keys:
        a, b = getattr(got, k), getattr(want, k)
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert torch.equal(a.cpu().view(torch.uint8), b.cpu().view(torch.uint8)), k


def test_one_launch_gather_equals_from_data_list(subjects):
    from igcn_amd.data import Batch
    from igcn_amd.loader import UniformGraphStore, _like
    _, slim = subjects
    store = UniformGraphStore(slim, "cuda")
    assert store.one_launch()
    idx = torch.tensor([7, 0, 39, 7, 21, 3, 3, 12], device="cuda")
    slot = _like(store.batch(torch.arange(8, device="cuda")), "cuda")
    store.gather_into(idx, slot)
    torch.cuda.synchronize()
    want = Batch.from_data_list([slim[int(i)] for i in idx])
    _same(slot, want, KEYS + ("sbjID", "demographics"))
    assert slot.num_graphs == 8


@pytest.mark.parametrize("kind", ["device", "host"])
def test_feeder_hands_over_every_subject_once_per_epoch(subjects, kind):
    """Two epochs of 3 batches of 12 out of 40 subjects (ragged tail of 4 dropped, DataLoader(drop_last=True)): the
    batches are the collation of the seeded epoch order, slot reuse and side-stream hand-over included."""
    from igcn_amd.data import Batch
    from igcn_amd.loader import DeviceFeeder, EpochIndex, HostFeeder, UniformGraphStore
    _, slim = subjects
    if kind == "device":
        feeder = DeviceFeeder(UniformGraphStore(slim, "cuda"), 12, steps=6, seed=11)
        order = EpochIndex(40, 12, seed=11, device="cuda")
    else:
        feeder = HostFeeder(UniformGraphStore(slim, "cpu", pin=True), 12, torch.device("cuda", 0), steps=6, seed=11)
        order = EpochIndex(40, 12, seed=11)
    seen = []
    for batch in feeder:
        torch.cuda.current_stream().wait_event(batch.ready)
        idx = order.next().cpu()
        _same(batch, Batch.from_data_list([slim[int(i)] for i in idx]))
        seen.append(batch.sbjID.cpu().clone())
        batch.release()
    for ep in (seen[:3], seen[3:]):
        ids = torch.cat(ep)
        assert ids.numel() == 36 and ids.unique().numel() == 36
    assert not torch.equal(torch.cat(seen[:3]), torch.cat(seen[3:]))          # reshuffled


def test_device_feeder_writes_a_captured_steps_inputs_in_place(subjects):
    """``DeviceFeeder(into=step.data)``: the gather lands in the static inputs of a captured step (no slot, no copy;
    ``load`` of the step's own batch is a no-op) and the fed step equals the eager step on the same subjects."""
    import copy
    from test_gpu_epoch import _small_model as small_model, LAM
    from igcn_amd.data import Batch
    from igcn_amd.loader import DeviceFeeder, EpochIndex, UniformGraphStore
    from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
    _, slim = subjects
    m1 = small_model()
    m2 = copy.deepcopy(m1)
    o1, o2 = FlatAdam(m1.parameters(), lr=1e-3), FlatAdam(m2.parameters(), lr=1e-3)
    static = Batch.from_data_list(slim[:12]).to("cuda")
    static.x.requires_grad_(True)
    step = GraphedTrainStep(m1, o1, static, LAM, warmup=1)
    order = EpochIndex(40, 12, seed=2, device="cuda")
    for batch in DeviceFeeder(UniformGraphStore(slim, "cuda"), 12, steps=4, seed=2, into=step.data):
        assert batch is step.data
        torch.cuda.current_stream().wait_event(batch.ready)
        step.load(batch)
        batch.release()
        idx = order.next().cpu()
        want = Batch.from_data_list([slim[int(i)] for i in idx])
        _same(batch, want)
        l1 = float(step())
        l2 = float(train_step(m2, o2, want.to("cuda"), LAM))
        assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l2)), (l1, l2)


def test_dense_feeder_equals_the_pre_transform_of_the_drawn_subjects(subjects):
    """DeviceGdcFeeder (matrices read in place by igcn_gdc_topk_of, attributes by igcn_gather_batch) against
    ``batch_from_dense`` of the gathered matrices — itself pinned to util_gdc.py by tests/test_gpu_gdc.py."""
    from igcn_amd.gdc import batch_from_dense
    from igcn_amd.loader import DeviceGdcFeeder, EpochIndex, UniformGraphStore
    graphs, slim = subjects
    adj = torch.stack([g.A for g in graphs]).cuda()
    store = UniformGraphStore(slim, "cuda")
    cols = {k: store.cols[k] for k in ("x", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y")}
    feeder = DeviceGdcFeeder(adj, cols, 12, steps=5, top_k=3, alpha=0.05, seed=3)
    order = EpochIndex(40, 12, seed=3, device="cuda")
    for batch in feeder:
        torch.cuda.current_stream().wait_event(batch.ready)
        idx = order.next()
        sel = {k: torch.index_select(v, 0, idx) for k, v in cols.items()}
        x = sel.pop("x")
        per = {k: (v.reshape(12, -1) if k in ("snps_feat", "tsne_fdim", "clini_score") else v.reshape(-1))
               for k, v in sel.items()}
        want = batch_from_dense(torch.index_select(adj, 0, idx), x, top_k=3, alpha=0.05, check=True, **per)
        _same(batch, want)
        batch.release()


def test_gather_rejects_a_destination_of_the_wrong_size(subjects):
    from igcn_amd.loader import UniformGraphStore, _like
    _, slim = subjects
    store = UniformGraphStore(slim, "cuda")
    slot = _like(store.batch(torch.arange(8, device="cuda")), "cuda")
    with pytest.raises(ValueError):
        store.gather_into(torch.arange(6, device="cuda"), slot)


def test_an_index_outside_the_dataset_is_not_read(subjects):
    """A subject index >= the dataset size (or negative) must not turn into an out-of-bounds read on the device: the
    gather hands over zero rows and -1 index entries, the dense transform an empty graph (count 0, padding slots)."""
    from igcn_amd._lib import call, ptr, stream_ptr
    from igcn_amd.loader import UniformGraphStore, _like
    graphs, slim = subjects
    store = UniformGraphStore(slim, "cuda")
    idx = torch.tensor([3, 40, 5, -1], device="cuda")              # 40 subjects: 40 and -1 are outside
    slot = _like(store.batch(torch.arange(4, device="cuda")), "cuda")
    store.gather_into(idx, slot)
    good = store.batch(torch.tensor([3, 5], device="cuda"))
    x = slot.x.view(4, 90, -1)
    assert torch.equal(x[0], good.x.view(2, 90, -1)[0]) and torch.equal(x[2], good.x.view(2, 90, -1)[1])
    assert float(x[1].abs().max()) == 0.0 and float(x[3].abs().max()) == 0.0
    e = store.edges
    ei = slot.edge_index.view(2, 4, e)
    assert bool((ei[:, 1] == -1).all()) and bool((ei[:, 3] == -1).all()) and bool((ei[:, 0] >= 0).all())
    adj = torch.stack([g.A for g in graphs]).cuda()
    out_ei = torch.full((2, 4 * 90 * 3), 7, dtype=torch.int64, device="cuda")
    out_ew = torch.full((4 * 90 * 3,), 7.0, device="cuda")
    counts = torch.full((4,), 7, dtype=torch.int32, device="cuda")
    call("igcn_gdc_topk_of", 4, 90, 3, 0.05, ptr(adj), 40, ptr(idx), ptr(out_ei), ptr(out_ew), ptr(counts), stream_ptr())
    assert counts.tolist()[1] == 0 and counts.tolist()[3] == 0 and counts.tolist()[0] == 270
    assert bool((out_ei.view(2, 4, 270)[:, 1] == -1).all()) and float(out_ew.view(4, 270)[3].abs().max()) == 0.0
