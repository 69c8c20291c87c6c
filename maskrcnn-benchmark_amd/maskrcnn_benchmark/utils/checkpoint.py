"""Minimal checkpointing for the training loop (reference utils/checkpoint.py:13-139): save / load
model + optimizer + scheduler + iteration on rank 0.  The model-zoo catalog and Caffe2 weight
conversion of the reference are outside the hot path (no network, random init)."""
import logging
import os

import torch

from .comm import is_main_process


class DetectronCheckpointer(object):
    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.cfg = cfg
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.save_dir = save_dir
        self.save_to_disk = is_main_process() if save_to_disk is None else save_to_disk
        self.logger = logger or logging.getLogger(__name__)

    def _module(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self._module().state_dict()}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        data.update(kwargs)
        path = os.path.join(self.save_dir, "{}.pth".format(name))
        self.logger.info("Saving checkpoint to {}".format(path))
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(path)

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def load(self, f=None, use_latest=True):
        """reference utils/checkpoint.py:52-73: `last_checkpoint` overrides `f` unless use_latest=False.
        Keys saved from a DDP-wrapped model (leading "module.", which is what the reference writes) are
        accepted: the prefix is stripped like the reference's strip_prefix_if_present.  The file format is
        otherwise this package's own ({"model": unwrapped state_dict, ...})."""
        if self.has_checkpoint() and use_latest:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as fh:
                f = fh.read().strip()
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from {}".format(f))
        data = torch.load(f, map_location="cpu")
        state = data.pop("model")
        if state and all(k.startswith("module.") for k in state):
            state = {k[len("module."):]: v for k, v in state.items()}
        self._module().load_state_dict(state)
        if "optimizer" in data and self.optimizer is not None:
            self.optimizer.load_state_dict(data.pop("optimizer"))
        if "scheduler" in data and self.scheduler is not None:
            self.scheduler.load_state_dict(data.pop("scheduler"))
        return data
