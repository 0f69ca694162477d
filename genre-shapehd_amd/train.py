"""DDP launcher + train steps for BASELINE.json configs #4 (ShapeHD fine-tuning, 3-D WGAN-GP) and #5 (GenRe joint
fine-tuning through the differentiable projections, optional Chamfer term) -- SURVEY section 8 f-3.

The reference has no distributed code (train.py:137-214 builds one model on one GPU and loops over a DataLoader);
this is the MI355X-first replacement for that entry: one process per GPU,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        genre-shapehd_amd/train.py --config shapehd --batch 8 --steps 100

`torch.distributed` backend "nccl" (= RCCL over xGMI); every trainable network is wrapped in
DistributedDataParallel, whose bucketed all-reduce overlaps the Conv3d backward; the geometric ops need no
collective (batch items are independent) and BatchNorm statistics stay per rank, as in the reference.
Data are synthetic and seeded per rank (no dataset ships with the reference, F7); checkpoints are written by rank 0
in the reference's format (models/checkpoint.py).

The step functions are importable on their own (tests/test_train_ddp.py runs them under gloo on CPU)."""
import argparse
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn.functional as F

try:
    import genre_shapehd_amd                          # noqa: F401  (the alias module of the hyphenated package directory)
except ImportError:                                   # run as a script from anywhere: put the repo root on the path
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import genre_shapehd_amd                          # noqa: F401
from genre_shapehd_amd import dist_utils              # noqa: E402
from genre_shapehd_amd.models import shapehd as M_shapehd, checkpoint     # noqa: E402


# ---- synthetic batches (shapes of datasets/shapenet.py after preprocessing) ---------------------------------
def sketch_batch(n, device, seed, size=256, res=128):
    """MarrNet-2 / ShapeHD inputs: depth [n,1,S,S], normal [n,3,S,S], silhou [n,1,S,S] (x100 scale) + gt voxel [n,1,res^3]"""
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, size)
    r2 = ax[:, None] ** 2 + ax[None, :] ** 2
    rad = 0.3 + 0.4 * torch.rand(n, 1, 1, 1, generator=g)
    sil = (r2[None, None] < rad ** 2).float()
    depth = (1 - r2[None, None] / rad ** 2).clamp(min=0).sqrt() * sil * 100
    normal = torch.randn(n, 3, size, size, generator=g) * sil * 100
    av = torch.linspace(-1, 1, res)
    v2 = av[:, None, None] ** 2 + av[None, :, None] ** 2 + av[None, None, :] ** 2
    voxel = (v2[None, None] < rad[..., None] ** 2).float()
    return (SimpleNamespace(depth=depth.to(device), normal=normal.to(device), silhou=(sil * 100).to(device)),
            voxel.to(device))


def genre_batch(n, device, seed, margin=16):
    """GenRe inputs rgb [n,3,256,256], silhou (x100) + the ground truths of the joint loss"""
    g = torch.Generator().manual_seed(seed)
    sk, voxel = sketch_batch(n, "cpu", seed)
    rgb = torch.rand(n, 3, 256, 256, generator=g)
    side = 128 + 2 * margin
    gt = SimpleNamespace(voxel=voxel.to(device), spherical_object=torch.rand(n, 1, side, side, generator=g).to(device),
                         depth=sk.depth.to(device), normal=sk.normal.to(device), silhou=sk.silhou.to(device),
                         depth_minmax=torch.tensor([[1.8, 2.6]]).repeat(n, 1).to(device),
                         cloud=(torch.rand(n, 2048, 3, generator=g) - 0.5).mul(0.8).to(device))
    return SimpleNamespace(rgb=rgb.to(device), silhou=sk.silhou.to(device)), gt


# ---- train steps ----------------------------------------------------------------------------------------------
def unwrap(net):
    return net.module if hasattr(net, "module") else net


def shapehd_train_step(net, optimizer, inputs, gt_voxel, w_gan_loss=0.0):
    """marrnet2.py:46-54 with shapehd.py's loss: one Adam step on the fine-tuned MarrNet-2 (the frozen critic and the
    frozen copy take no gradient, so DDP all-reduces MarrNet-2's gradients only)"""
    optimizer.zero_grad(set_to_none=True)
    pred = net(inputs)
    loss, parts = M_shapehd.shapehd_loss(pred, gt_voxel, w_gan_loss)
    loss.backward()
    optimizer.step()
    return loss.detach(), parts


def depth_to_points(abs_depth, silhou_t, k=2048, fl=418.3, cam_dist=2.2, generator=None, idx=None):
    """back-project k foreground pixels of a ray-depth map [n,1,H,W] (the layout get_abs_depth produces) to 3-D with
    cam_bp's camera (back_projection_kernel.cu:231-242): differentiable w.r.t. the depth.  -> [n,k,3].  idx [n,k]
    (int64 pixel indices) fixes the sample instead of drawing it (reproducible steps)."""
    n, _, H, W = abs_depth.shape
    if idx is None:
        fg = (abs_depth.detach().flatten(1) > 0).float() + 1e-6
        idx = torch.multinomial(fg, k, replacement=True, generator=generator)
    d = abs_depth.flatten(1).gather(1, idx)
    u = (idx // W).float() - (H - 1) / 2.0
    v = (idx % W).float() - (W - 1) / 2.0
    dz = d * fl / torch.sqrt(u * u + v * v + fl * fl)
    return torch.stack((dz - cam_dist, -dz * v / fl, -dz * u / fl), -1)


def genre_train_step(net, optimizer, inputs, gt, opt, chamfer_weight=0.0, chamfer_idx=None):
    """joint fine-tuning of all three GenRe modules (--joint_train, depth_pred_with_sph_inpaint.py:114-118,
    genre_full_model.py:117-121).  What reaches MarrNet-1 through the geometry: the voxel loss through Unet_3D's second input
    channel clamp(proj_depth / 50) <- cam_bp <- get_abs_depth (genre_full_model.py:126).  The other branch -- spherical
    back-projection <- net2 <- render_spherical <- clamp(proj * 50) <- cam_bp -- trains net2 and the refiner but carries an
    IDENTICALLY ZERO gradient into MarrNet-1, in the reference as here: the x50 clamp saturates every occupied voxel and
    blocks every empty one (depth_pred_with_sph_inpaint.py:124; DESIGN 3.4d,
    tests/test_gpu_models.py::test_projection_gradients_of_the_joint_step_against_the_cpu_chain).  chamfer_weight > 0
    adds a Chamfer term (toolbox/nndistance; the reference ships the op but no loss uses it) between the
    back-projected predicted depth and a ground-truth surface cloud."""
    from genre_shapehd_amd.models.genre import genre_loss, SCALE_25D
    from genre_shapehd_amd.callers import AbsDepth
    optimizer.zero_grad(set_to_none=True)
    pred = net(inputs)
    loss = genre_loss(pred, gt, opt, joint=opt.joint_train)
    if chamfer_weight > 0:
        from genre_shapehd_amd.toolbox.nndistance.functions.nnd import nndistance
        depth = AbsDepth.apply(pred["depth"], pred["depth_minmax"], inputs.silhou, SCALE_25D)
        pts = depth_to_points(depth, inputs.silhou, idx=chamfer_idx).contiguous()
        d1, d2 = nndistance(pts, gt.cloud.contiguous())
        loss = loss + chamfer_weight * (d1.mean() + d2.mean())
    loss.backward()
    optimizer.step()
    return loss.detach()


def ddp(net, device, dist):
    if dist is None:
        return net
    from torch.nn.parallel import DistributedDataParallel as DDP
    # broadcast_buffers=False: BatchNorm running statistics stay per rank (the reference has no SyncBN and no buffer
    # broadcast); DDP's default would overwrite every rank's buffers with rank 0's on each forward
    if device.type == "cuda":
        return DDP(net, device_ids=[device.index], bucket_cap_mb=64,     # ~4 buckets for Unet_3D's 214 MB of gradients
                   broadcast_buffers=False)
    return DDP(net, broadcast_buffers=False)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["shapehd", "wgangp", "genre"], default="shapehd")
    ap.add_argument("--batch", type=int, default=8, help="per GPU")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--w_gan_loss", type=float, default=1e-3)
    ap.add_argument("--chamfer_weight", type=float, default=0.0)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--save", default=None)
    ap.add_argument("--resume", default=None)
    args = ap.parse_args(argv)
    rank, local, world = dist_utils.env_rank_world()
    cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(device)
    # util/util_loadlib.py:11 of the reference: MIOpen times its solvers per convolution shape instead of trusting its
    # heuristics (measured on MI355X, round 4: ShapeHD step 49.7 -> 35.8 ms, GenRe joint 75.5 -> 64.5 ms; the results land in
    # the user find-db, which tools/warm_miopen.py ships in the tree)
    torch.backends.cudnn.benchmark = True
    dist = dist_utils.init_from_env(args.backend or ("nccl" if cuda else "gloo"), device if cuda else None)
    torch.manual_seed(1234)                                             # identical initial weights on every rank
    start = 0
    if args.config == "shapehd":
        net = M_shapehd.ShapeHDNet().to(device)
        model = ddp(net, device, dist)
        optim = torch.optim.Adam(net.marrnet2.parameters(), lr=args.lr, betas=(0.5, 0.9))
        nets, optims = [net], [optim]
    elif args.config == "wgangp":
        gan = M_shapehd.WGANGP(lr=args.lr)
        gan.net_g.to(device), gan.net_d.to(device)
        raw_g, raw_d = gan.net_g, gan.net_d
        gan.net_g, gan.net_d = ddp(raw_g, device, dist), ddp(raw_d, device, dist)
        nets, optims = [raw_g, raw_d], [gan.opt_g, gan.opt_d]
    else:
        from genre_shapehd_amd.models.genre import GenReNet, GenReOptions
        gopt = GenReOptions(joint_train=True)
        net = GenReNet(gopt).to(device)
        model = ddp(net, device, dist)
        optim = torch.optim.Adam(net.parameters(), lr=args.lr, betas=(0.5, 0.9))
        nets, optims = [net], [optim]
    if args.resume:
        extra = checkpoint.load_state_dict(args.resume, nets, optims)
        start = int(extra.get("epoch", 0))
    for step in range(start, start + args.steps):
        seed = 1000 * step + rank
        if args.config == "shapehd":
            inputs, voxel = sketch_batch(args.batch, device, seed)
            loss, _ = shapehd_train_step(model, optim, inputs, voxel, args.w_gan_loss)
        elif args.config == "wgangp":
            _, voxel = sketch_batch(args.batch, device, seed)
            loss = gan.train_on_batch(step, voxel)["err_d"]
        else:
            inputs, gt = genre_batch(args.batch, device, seed)
            loss = genre_train_step(model, optim, inputs, gt, gopt, args.chamfer_weight)
        if dist is not None:                                            # global mean for the log line only
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss = loss / world
        if rank == 0:
            print("step %d loss %.6f" % (step, loss.item()), flush=True)
    if args.save:
        # per-rank BatchNorm statistics (ddp(): broadcast_buffers=False) -> their mean over the ranks, on every rank, before
        # rank 0 writes the checkpoint (dist_utils.average_float_buffers)
        with torch.no_grad():
            dist_utils.average_float_buffers(dist, nets)
    if args.save and rank == 0:
        checkpoint.save_state_dict(args.save, nets, optims, epoch=start + args.steps, loss_eval=float(loss))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
