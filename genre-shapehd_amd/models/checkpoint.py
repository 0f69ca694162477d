"""Checkpoints in the reference's format (models/netinterface.py:405-448):

    {'nets': [state_dict, ...], 'optimizers': [state_dict, ...] (optional), 'epoch': ..., 'loss_eval': ..., ...}

Reloading an optimizer keeps the CURRENT training hyper-parameters (learning rate, betas, ...) and takes only the
moments and step counts from the file, as the reference does when a run is resumed with new options."""
import torch


def save_state_dict(path, nets, optimizers=None, **additional_values):
    state = {"nets": [n.state_dict() for n in nets]}
    if optimizers is not None:
        state["optimizers"] = [o.state_dict() for o in optimizers]
    state.update(additional_values)
    torch.save(state, path)


def optimizer_load_state_dict(optimizer, state, keep_training_params=False):
    if keep_training_params:
        current = optimizer.state_dict()["param_groups"]
        assert len(current) == len(state["param_groups"])
        state = dict(state, param_groups=[dict(g, **{k: v for k, v in cur.items() if k != "params"})
                                         for cur, g in zip(current, state["param_groups"])])
    optimizer.load_state_dict(state)


def load_state_dict(path, nets, optimizers=None, load_optimizer="auto", map_location="cpu"):
    """-> the additional values stored beside 'nets' / 'optimizers' (epoch, loss_eval, ...)"""
    state = torch.load(path, map_location=map_location)
    if load_optimizer == "auto":
        load_optimizer = "optimizers" in state and optimizers is not None
    assert len(nets) == len(state["nets"]), "checkpoint holds %d networks, expected %d" % (len(state["nets"]), len(nets))
    for net, sd in zip(nets, state["nets"]):
        net.load_state_dict(sd)
    if load_optimizer:
        assert len(optimizers) == len(state["optimizers"])
        for opt, sd in zip(optimizers, state["optimizers"]):
            optimizer_load_state_dict(opt, sd, keep_training_params=True)
    return {k: v for k, v in state.items() if k not in ("nets", "optimizers")}
