"""configs[0] on the CPU: the product's host entry points of nndistance (csrc/nnd_host.hip -- the reference's
my_lib.nnd_forward / nnd_backward, toolbox/nndistance/src/my_lib.c:6-118) through NNDFunction with CPU tensors,
bit for bit against the reference's own my_lib.c compiled for the host (oracle/_ref) and against the C oracle."""
import numpy as np
import pytest
import torch

import inputs


@pytest.mark.parametrize("b,n,m", [(1, 2048, 2048), (3, 257, 100), (2, 1, 5), (1, 64, 1)])
def test_cpu_tensors_take_the_host_entry_points(b, n, m, genre, oracle):
    x1, x2 = inputs.clouds(b, n, m, seed1=10 + n, seed2=20 + m)
    if n >= 64 and m >= 5:
        x2[:, 3] = x2[:, 1]                                             # exact ties: the first of equal minima wins
    a = torch.from_numpy(x1).requires_grad_(True)
    c = torch.from_numpy(x2).requires_grad_(True)
    d1, d2, i1, i2 = genre.nndistance_w_idx(a, c)
    assert not d1.is_cuda and i1.dtype == torch.int32
    rd1, rd2, ri1, ri2 = oracle.nnd_forward(x1, x2)
    assert np.array_equal(i1.numpy(), ri1) and np.array_equal(i2.numpy(), ri2)
    assert np.array_equal(d1.detach().numpy(), rd1) and np.array_equal(d2.detach().numpy(), rd2)
    rng = np.random.default_rng(5)
    g1, g2 = rng.standard_normal((b, n)).astype(np.float32), rng.standard_normal((b, m)).astype(np.float32)
    ((d1 * torch.from_numpy(g1)).sum() + (d2 * torch.from_numpy(g2)).sum()).backward()     # one nnd_backward call
    ga, gc = oracle.nnd_backward(x1, x2, g1, g2, ri1, ri2)
    assert np.array_equal(a.grad.numpy(), ga) and np.array_equal(c.grad.numpy(), gc)


def test_host_entry_matches_the_reference_build(genre, reference):
    x1, x2 = inputs.clouds(2, 300, 211, seed1=1, seed2=2)
    d1, d2, i1, i2 = genre.nndistance_w_idx(torch.from_numpy(x1), torch.from_numpy(x2))
    rd1, rd2, ri1, ri2 = reference.nnd_forward(x1, x2)
    assert np.array_equal(i1.numpy(), ri1) and np.array_equal(i2.numpy(), ri2)
    assert np.array_equal(d1.numpy(), rd1) and np.array_equal(d2.numpy(), rd2)
    score = genre.nndistance_score(torch.from_numpy(x1), torch.from_numpy(x2))
    assert score.shape == (2,) and torch.isfinite(score).all()


@pytest.mark.parametrize("isa", ["scalar", "avx2", "avx512"])
def test_every_vector_width_of_the_host_search_is_bit_identical(isa):
    """csrc/nnd_host.hip runs 16 / 8 / 1 queries per vector (picked from the CPU's features at run time, pinned here through
    GENRE_NND_HOST_ISA -- read once per process, hence the subprocess): dist and idx bit-identical to the C oracle on ragged
    sizes (query counts that are not a multiple of the width, one query, one target), exact ties (the first of equal minima
    wins in every lane) and a NaN coordinate (never `<` anything: it stays where the scalar loop leaves it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path[:0] = [%r, %r]
import numpy as np, torch
import inputs
import genre_shapehd_amd as G
from oracle.oracle import Oracle
O = Oracle()
for b, n, m in ((1, 2048, 2048), (2, 37, 100), (3, 1, 5), (1, 64, 1), (2, 17, 16), (1, 255, 33)):
    x1, x2 = inputs.clouds(b, n, m, seed1=30 + n, seed2=40 + m)
    if m >= 5:
        x2[:, 3] = x2[:, 1]
    if n >= 17:
        x1[0, 5, 1] = np.nan
    d1, d2, i1, i2 = G.nndistance_w_idx(torch.from_numpy(x1), torch.from_numpy(x2))
    rd1, rd2, ri1, ri2 = O.nnd_forward(x1, x2)
    assert np.array_equal(i1.numpy(), ri1) and np.array_equal(i2.numpy(), ri2), (b, n, m)
    assert np.array_equal(d1.numpy(), rd1, equal_nan=True) and np.array_equal(d2.numpy(), rd2, equal_nan=True), (b, n, m)
import ctypes
lib = ctypes.CDLL(%r)
lib.genre_nnd_host_isa.restype = ctypes.c_char_p
print('ok isa=' + lib.genre_nnd_host_isa().decode())
""" % (root, os.path.join(root, "tests"), os.path.join(root, "genre-shapehd_amd", "csrc", "libgenre_hip.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GENRE_NND_HOST_ISA=isa))
    assert out.returncode == 0 and "ok isa=" in out.stdout, (out.stdout[-300:], out.stderr[-1500:])
    ran = out.stdout.rsplit("ok isa=", 1)[1].strip()
    if ran != isa:          # this CPU lacks the requested width: the run above exercised `ran`, not `isa` -- say so
        pytest.skip("the host CPU has no %s: the search ran its %s path (bit-identical, but not the width asked for)" % (isa, ran))
