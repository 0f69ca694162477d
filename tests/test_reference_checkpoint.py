"""checkpoints of the reference load into this repo's GenRe model key for key.

  * always: GenReNet's state_dict keys and shapes == tests/golden/genre_reference_keys.json, which
    tests/golden/make_genre_reference_golden.py wrote from the reference's OWN classes (models/genre_full_model.py:104-113,
    depth_pred_with_sph_inpaint.py:97-105, marrnet1.py:137-154 incl. the depth min/max head);
  * where /root/reference exists (this container, not the GPU box): that script is re-run with --check in a subprocess --
    it builds the reference's Net, writes a checkpoint with the reference's NetInterface.save_state_dict
    (netinterface.py:405-412), loads it through GenReInference.load() (tensor for tensor equal), runs the reference
    class's forward on the CPU with the oracle's ops behind the reference's toolbox interfaces, and compares keys and
    forward digests with the committed fixtures (the ones tests/test_gpu_reference_checkpoint.py holds the GPU to)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_genre_model_keys_equal_the_reference_classes():
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.models import GenReNet
    with open(os.path.join(GOLD, "genre_reference_keys.json")) as f:
        want = json.load(f)
    got = {k: list(v.shape) for k, v in GenReNet().state_dict().items()}
    assert got == want
    assert "depth_and_inpaint.net1.decoder_minmax.9.bias" in want and want["grid"] == [1, 1, 128, 128, 3]


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference tree is not on this machine")
def test_checkpoint_of_the_reference_classes_loads_and_fixtures_are_fresh():
    out = subprocess.run([sys.executable, os.path.join(GOLD, "make_genre_reference_golden.py"), "--check"],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "loads through GenReInference.load()" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
