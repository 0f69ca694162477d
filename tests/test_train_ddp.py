"""SURVEY 8 f-3 / f-4 on CPU: the DDP train step under gloo (world_size 2), the WGAN-GP step with its second-order
gradient penalty, the checkpoint format of the reference (models/netinterface.py:405-448), and the GenRe model's
state_dict keys."""
import json
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(encode_dims=8, nf=16, d_nf=2)             # ShapeHD with narrow 3-D networks: seconds on one core


def _make(seed=7):
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.models.shapehd import ShapeHDNet
    torch.manual_seed(seed)
    return ShapeHDNet(**SMALL)


def _grads(net):
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd import train as T, dist_utils
    dist = dist_utils.init_from_env("gloo")
    net = _make()
    model = T.ddp(net, torch.device("cpu"), dist)
    optim = torch.optim.SGD(net.marrnet2.parameters(), lr=0.0)          # lr 0: the step leaves the gradients in place
    inputs, voxel = T.sketch_batch(2, "cpu", seed=100 + rank, size=64)
    loss, _ = T.shapehd_train_step(model, optim, inputs, voxel, w_gan_loss=0.5)
    ret[rank] = (loss.item(), {k: v.numpy() for k, v in _grads(net).items()})
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_equal_the_mean_of_the_per_rank_gradients():
    """two gloo ranks, different half-batches: after the DDP step every rank holds the MEAN of the two ranks' local
    gradients -- the single-process gradient of the mean of the two half-batch losses (BatchNorm statistics stay per
    rank, as in the reference, which has no SyncBN); only MarrNet-2 receives gradients (critic and copy are frozen)"""
    sys.path.insert(0, ROOT)
    from genre_shapehd_amd import train as T
    world, port = 2, 29600 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    torch.set_num_threads(2)
    want = None
    for r in range(world):
        net = _make()
        inputs, voxel = T.sketch_batch(2, "cpu", seed=100 + r, size=64)
        optim = torch.optim.SGD(net.marrnet2.parameters(), lr=0.0)
        T.shapehd_train_step(net, optim, inputs, voxel, w_gan_loss=0.5)
        g = _grads(net)
        want = g if want is None else {k: want[k] + g[k] for k in g}
    want = {k: v / world for k, v in want.items()}
    assert all(k.startswith("marrnet2.") for k in want) and len(want) > 50
    for r in range(world):
        got = ret[r][1]
        assert set(got) == set(want)
        for k in want:
            scale = max(1e-6, want[k].abs().max().item())
            assert (torch.from_numpy(got[k]) - want[k]).abs().max().item() <= 1e-5 * scale + 1e-7, k


def test_wgangp_step_differentiates_the_gradient_penalty():
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.models.shapehd import WGANGP
    from genre_shapehd_amd.networks import VoxelGenerator, VoxelDiscriminator
    torch.manual_seed(0)
    gan = WGANGP(VoxelGenerator(nz=8, nf=2, res=64), VoxelDiscriminator(nf=2, res=64), nz=8)
    real = (torch.rand(2, 1, 64, 64, 64) > 0.7).float()
    # the penalty alone: its gradient w.r.t. the critic's weights exists only through create_graph=True
    fake = gan.sample(2, "cpu").detach()
    gp = gan.grad_penalty(real, fake)
    gp.backward()
    w = gan.net_d.main[0].weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and w.grad.abs().max().item() > 0
    before = [p.detach().clone() for p in gan.net_d.parameters()], [p.detach().clone() for p in gan.net_g.parameters()]
    log = gan.train_on_batch(0, real)
    assert all(torch.isfinite(torch.as_tensor(v)).all() for v in log.values())
    assert any((a != b).any() for a, b in zip(before[0], gan.net_d.parameters()))
    assert any((a != b).any() for a, b in zip(before[1], gan.net_g.parameters()))


def test_checkpoint_round_trip_keeps_the_training_hyperparameters(tmp_path):
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.models import checkpoint as C
    from genre_shapehd_amd import train as T
    net = _make(1)
    optim = torch.optim.Adam(net.marrnet2.parameters(), lr=1e-3, betas=(0.5, 0.9))
    inputs, voxel = T.sketch_batch(1, "cpu", seed=3, size=64)
    T.shapehd_train_step(net, optim, inputs, voxel, 0.1)
    path = str(tmp_path / "checkpoint.pt")
    C.save_state_dict(path, [net], [optim], epoch=7, loss_eval=0.25)
    raw = torch.load(path)
    assert set(raw) == {"nets", "optimizers", "epoch", "loss_eval"} and len(raw["nets"]) == 1     # netinterface.py:405-412
    net2 = _make(2)
    optim2 = torch.optim.Adam(net2.marrnet2.parameters(), lr=5e-5, betas=(0.9, 0.999))
    extra = C.load_state_dict(path, [net2], [optim2])
    assert extra == {"epoch": 7, "loss_eval": 0.25}
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    g = optim2.param_groups[0]
    assert g["lr"] == 5e-5 and tuple(g["betas"]) == (0.9, 0.999)        # current options win (netinterface.py:439-448)
    s1, s2 = optim.state_dict()["state"], optim2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)
    C.save_state_dict(path, [net])                                       # nets only: loads without optimizers
    assert C.load_state_dict(path, [net2], [optim2]) == {}


def test_genre_model_state_dict_has_the_reference_keys():
    """models/genre_full_model.py:104-113 + depth_pred_with_sph_inpaint.py:97-105 + marrnet1.py:137-154: the key set is
    the composition of the per-network fixtures (generated from the reference's network classes) -- and, as a whole, the
    fixture generated from the reference's own GenRe model class (tests/test_reference_checkpoint.py)"""
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.models import GenReNet
    with open(os.path.join(ROOT, "tests", "golden", "networks_keys.json")) as f:
        nk = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "genre_reference_keys.json")) as f:
        full = json.load(f)
    got = {k: list(v.shape) for k, v in GenReNet().state_dict().items()}
    assert got == full
    for prefix, name in (("depth_and_inpaint.net1.", "uresnet_net"), ("depth_and_inpaint.net2.", "uresnet_inpaint"),
                         ("refine_net.", "unet3d")):
        for k, v in nk[name].items():
            assert full[prefix + k] == v, prefix + k


def _gan_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd import train as T, dist_utils
    from genre_shapehd_amd.models.shapehd import WGANGP
    from genre_shapehd_amd.networks import VoxelGenerator, VoxelDiscriminator
    dist = dist_utils.init_from_env("gloo")
    torch.manual_seed(0)                                                # identical initial weights on every rank
    raw_g, raw_d = VoxelGenerator(nz=8, nf=2, res=64), VoxelDiscriminator(nf=2, res=64)
    gan = WGANGP(raw_g, raw_d, nz=8, lr=1e-3, generator=torch.Generator().manual_seed(100 + rank))
    gan.net_g, gan.net_d = T.ddp(raw_g, torch.device("cpu"), dist), T.ddp(raw_d, torch.device("cpu"), dist)
    rng = torch.Generator().manual_seed(200 + rank)                     # different data on every rank
    for step in range(2):
        real = (torch.rand(2, 1, 64, 64, 64, generator=rng) > 0.7).float()
        gan.train_on_batch(step, real)
    out = {"g": [p.detach().numpy().copy() for p in raw_g.parameters()],
           "d": [p.detach().numpy().copy() for p in raw_d.parameters()],
           "bn": [b.detach().numpy().copy() for k, b in raw_g.named_buffers() if k.endswith("running_mean")]}
    with torch.no_grad():                                               # what the launcher does before rank 0 saves
        dist_utils.average_float_buffers(dist, [raw_g, raw_d])
    out["bn_avg"] = [b.detach().numpy().copy() for k, b in raw_g.named_buffers() if k.endswith("running_mean")]
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_wgangp_under_ddp_keeps_parameters_in_sync_and_batchnorm_statistics_per_rank():
    """models/wgangp.py:77-164 under DDP: the critic is called three times (real, fake, interpolate) before its one
    backward and its penalty is a second-order gradient; after two batches of different data per rank the parameters
    of both networks are identical on the two ranks (all-reduced gradients), while the generator's BatchNorm running
    statistics differ -- buffers are NOT broadcast (train.ddp: broadcast_buffers=False; the reference has no SyncBN)"""
    import numpy as np
    world, port = 2, 31600 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_gan_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    for net in ("g", "d"):
        assert len(a[net]) == len(b[net]) > 0
        for x, y in zip(a[net], b[net]):
            assert np.array_equal(x, y), net
    assert len(a["bn"]) > 0 and any(not np.array_equal(x, y) for x, y in zip(a["bn"], b["bn"]))
    # ... and a checkpoint carries their mean over the ranks, the same on every rank (dist_utils.average_float_buffers)
    for x, y, m, n in zip(a["bn"], b["bn"], a["bn_avg"], b["bn_avg"]):
        assert np.array_equal(m, n) and np.allclose(m, (x + y) / 2, rtol=1e-6, atol=1e-7)


def test_back_projection_layer_constants_survive_a_change_of_batch_size():
    """a captured HIP graph holds the raw pointer of the layer's cached fl / cam_dist tensors: a tensor handed out for
    one batch size must stay alive (and unchanged) when another batch size is seen"""
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.modules.camera_backprojection_module import Camera_back_projection_layer
    layer = Camera_back_projection_layer()
    t1 = layer._const(418.3, 1, "cpu")
    t8 = layer._const(418.3, 8, "cpu")
    again = layer._const(418.3, 1, "cpu")
    assert again is t1 and t8.shape == (8, 1) and t1.shape == (1, 1)
    assert layer._const(2.2, 1, "cpu") is not t1
    for n in range(2, 2 + 3 * layer._MAX_CONSTS):                       # bounded
        layer._const(418.3, n, "cpu")
    assert len(layer._consts) <= layer._MAX_CONSTS
