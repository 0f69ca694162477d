"""How profiles/pmc_targets.py orders the renderers' launches, shared by the three summarisers (summarize_rocpd.py --phases,
pmc_traffic_table.py, pmc_sq_table.py): a kernel's dispatches, in launch order, fall into equal consecutive PHASES -- a leading
remainder is set-up (the constants of an occupancy hint are built by rendering a constant volume once) and is dropped -- and are
reported as "name@phase":
  genre  GenRe's own volume as the step's layer hands it over (with its occupancy words): what the timed steps render
  dense  the same volume without the words (segment forward only)
  nosave the image-minor forward on GenRe's volume without saved state (what the step runs: the clamp provably blocks every voxel)
  soft   a volume whose every sample passes the clamps: a gradient everywhere"""
TWO = ("genre", "soft")
PHASES = {
    "seg_combine_kernel": ("genre", "dense", "soft"),
    "bm_combine_fwd_kernel": ("genre", "nosave", "soft"), "bm_combine_bwd_kernel": TWO, "bm_scatter_kernel": TWO,
    "bm_zero_shared_kernel": TWO, "seg_combine_bwd_kernel": TWO, "seg_scatter_kernel": TWO, "seg_halo_kernel": TWO,
    "render_bwd_brick_kernel": TWO, "zero_shared_bricks_kernel": TWO,
}


def split(name, vals):
    """[(suffix, values)]: ("", vals) for an un-phased kernel, else one ("@phase", values) per phase"""
    if "bm_sample_kernel<" in name:          # its HINT template argument names the phase: <PS, SAVE, HINT, NT>
        args = name.split("bm_sample_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        if args[:2] == ["true", "true"]:
            return [("@genre" if args[2] == "true" else "@soft", vals)]
        return [("", vals)]
    if "seg_sample_kernel<" in name:         # <VEC, SPEC, SAVE_V>: pmc_targets.py saves sample values in the genre and soft phases
        args = name.split("seg_sample_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        if args[2] == "true":
            per = len(vals) // 2
            vals = vals[len(vals) - 2 * per:]
            return [("@genre", vals[:per]), ("@soft", vals[per:])]
        return [("@dense", vals)]
    for key, phases in PHASES.items():
        if key in name and len(vals) >= len(phases):
            per = len(vals) // len(phases)
            vals = vals[len(vals) - per * len(phases):]
            return [("@" + ph, vals[i * per:(i + 1) * per]) for i, ph in enumerate(phases)]
    return [("", vals)]
