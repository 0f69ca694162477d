"""Callers of the geometric hot path (SURVEY section 8 f-2 ... f-4): the GenRe and ShapeHD networks composed around
the native ops, a checkpoint reader/writer in the reference's format, and the inference entry.  Everything here is
stock PyTorch-ROCm; the MI355X-specific work stays in csrc/ and is reached through toolbox/ and callers.py."""
from .genre import MarrNet1Net, DepthInpaintNet, GenReNet, GenReOptions, Inputs, GenReInference, genre_loss   # noqa: F401
from .shapehd import MarrNet2Net, ShapeHDNet, shapehd_loss, WGANGP                                  # noqa: F401
from .checkpoint import save_state_dict, load_state_dict, optimizer_load_state_dict                 # noqa: F401
