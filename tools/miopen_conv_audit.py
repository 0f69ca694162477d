"""Which MIOpen solver is inaccurate?  Collects every convolution configuration a train step of this repository runs
(module class, channels, kernel, stride, padding, input shape), replays each one alone on the GPU in fp32 -- forward,
data gradient, weight gradient -- and compares with the same convolution in float64 on the CPU.  Printed per
configuration: max |gpu - f64| / max |f64| for y, dx, dw, and the CPU's own fp32 error beside it.  Run it under the
MIOpen database configuration in question (MIOPEN_USER_DB_PATH / MIOPEN_FIND_MODE / MIOPEN_DEBUG_* in the environment);
the convolution keys printed are the ones of the user find-db (genre-shapehd_amd/.miopen/db/*.ufdb.txt).

usage (GPU box): python tools/miopen_conv_audit.py [shapehd|wgangp|genre] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import miopen_cache  # noqa: E402

if os.environ.get("GENRE_MIOPEN_DIR") != "none":
    miopen_cache.use(os.environ.get("GENRE_MIOPEN_DIR"))
import torch  # noqa: E402
from torch import nn  # noqa: E402
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd import train as T  # noqa: E402
from genre_shapehd_amd.models import shapehd as MS  # noqa: E402

CONVS = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)


def collect(which, batch):
    """-> ordered dict config -> first module name, from one CPU forward with hooks (shapes only; tiny cost next to the audit)"""
    seen = {}

    def hook(name):
        def fn(mod, inp, out):
            key = (type(mod).__name__, mod.in_channels, mod.out_channels, tuple(mod.kernel_size), tuple(mod.stride),
                   tuple(mod.padding), tuple(getattr(mod, "output_padding", ())), mod.bias is not None,
                   tuple(inp[0].shape))
            seen.setdefault(key, name)
        return fn

    if which == "shapehd":
        net = MS.ShapeHDNet().train()
        ins, vox = T.sketch_batch(batch, "cpu", seed=21)
        run = lambda: net(ins)                                            # noqa: E731
    elif which == "wgangp":
        gan = MS.WGANGP()
        net = nn.ModuleList([gan.net_g, gan.net_d])
        run = lambda: gan.net_d(gan.net_g(torch.randn(batch, 200, 1, 1, 1)))  # noqa: E731
    else:
        from genre_shapehd_amd.models.genre import GenReNet, GenReOptions
        raise SystemExit("genre: needs the GPU ops for its forward; audit shapehd / wgangp")
    for name, mod in net.named_modules():
        if isinstance(mod, CONVS):
            mod.register_forward_hook(hook(name))
    with torch.no_grad():
        run()
    return seen


def audit(key, name, dev):
    cls, cin, cout, k, s, p, op, bias, shape = key
    kw = dict(kernel_size=k, stride=s, padding=p, bias=False)
    if "Transpose" in cls:
        kw["output_padding"] = op
    torch.manual_seed(hash((cin, cout, k, shape)) & 0xFFFF)
    m64 = getattr(nn, cls)(cin, cout, **kw).double()
    x64 = torch.randn(shape, dtype=torch.float64)
    y64 = m64(x64.requires_grad_(True))
    g64 = torch.randn_like(y64)
    y64.backward(g64)
    ref = (y64.detach(), x64.grad, m64.weight.grad)
    out = []
    for d in ("cpu", dev):
        m = getattr(nn, cls)(cin, cout, **kw)
        m.weight.data.copy_(m64.weight.data.float())
        m.to(d)
        x = x64.detach().float().to(d).requires_grad_(True)
        y = m(x)
        y.backward(g64.float().to(d))
        got = (y.detach(), x.grad, m.weight.grad)
        out.append([((a.double().cpu() - b).abs().max() / b.abs().max()).item() for a, b in zip(got, ref)])
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "shapehd"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    print("MIOpen db: %s  find mode: %s" % (miopen_cache.current(), os.environ.get("MIOPEN_FIND_MODE")), flush=True)
    dev = torch.device("cuda:0")
    cfgs = collect(which, batch)
    print("%d distinct convolutions in %s (batch %d)" % (len(cfgs), which, batch), flush=True)
    bad = 0
    t0 = time.time()
    for key, name in cfgs.items():
        cpu, gpu = audit(key, name, dev)
        flag = "  <-- " + ",".join(n for n, e in zip(("y", "dx", "dw"), gpu) if e > 1e-5) if max(gpu) > 1e-5 else ""
        bad += bool(flag)
        print("%-44s %-16s cin %4d cout %4d k %s s %s in %s | gpu y %.1e dx %.1e dw %.1e | cpu32 y %.1e dx %.1e dw %.1e%s"
              % (name[-44:], key[0], key[1], key[2], "x".join(map(str, key[3])), key[4][0], "x".join(map(str, key[8])),
                 gpu[0], gpu[1], gpu[2], cpu[0], cpu[1], cpu[2], flag), flush=True)
    print("%d of %d configurations beyond 1e-5 of the float64 result; %.0f s" % (bad, len(cfgs), time.time() - t0))


if __name__ == "__main__":
    main()
