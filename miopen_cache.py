"""Where MIOpen keeps its compiled convolution kernels for this repository.

On a fresh box MIOpen compiles one HIP kernel per convolution configuration it meets (several seconds each; the three GenRe
networks, MarrNet-2 and the 3-D GAN have ~300 between forward, data- and weight-gradient) and caches them under ~/.cache and
~/.config -- which a fresh box does not have.  `use()` points MIOpen at a directory INSIDE the working tree instead
(genre-shapehd_amd/.miopen: a build artefact like libgenre_hip.so -- git-ignored, travels with the tree; filled by
tools/warm_miopen.py), so that bench.py's `train` and `m1` sections and the network tests start from compiled kernels.
Nothing here touches the hot path or any timed region; without the directory MIOpen behaves as always.
Must be called before MIOpen is first used (bench.py, tests/conftest.py and __graft_entry__ call it before importing torch)."""
import os

ROOT = os.path.dirname(os.path.abspath(__file__))
DEFAULT = os.path.join(ROOT, "genre-shapehd_amd", ".miopen")


def use(path=None, create=False):
    path = os.path.abspath(path or DEFAULT)
    if not os.path.isdir(path):
        if not create:
            return None
        os.makedirs(path, exist_ok=True)
    for sub, var in (("db", "MIOPEN_USER_DB_PATH"), ("cache", "MIOPEN_CUSTOM_CACHE_DIR")):
        d = os.path.join(path, sub)
        os.makedirs(d, exist_ok=True)
        os.environ.setdefault(var, d)
    return path


def current():
    return os.environ.get("MIOPEN_CUSTOM_CACHE_DIR"), os.environ.get("MIOPEN_USER_DB_PATH")
