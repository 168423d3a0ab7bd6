"""Checkpoint I/O with the reference's on-disk layout (src/utils/path/ckpt.py:13-93):
`{"model": state_dict, "iteration": epoch, "optimizer": optimizer.state_dict(), "learning_rate": lr}` in
`<out>/logs/{G,D}_<step|latest>.pth`, written to a temp file then moved into place."""
import glob
import logging
import os
import shutil
import time

import torch

logger = logging.getLogger("easevoice")


def load_checkpoint(checkpoint_path, model, optimizer=None, skip_optimizer=False):
    assert os.path.isfile(checkpoint_path)
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    iteration, learning_rate = ck["iteration"], ck["learning_rate"]
    if optimizer is not None and not skip_optimizer and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    saved = ck["model"]
    target = model.module if hasattr(model, "module") else model
    new_sd = {}
    for k, v in target.state_dict().items():
        if k in saved and saved[k].shape == v.shape:     # shape-checked key copy, ckpt.py:33-46
            new_sd[k] = saved[k]
        else:
            logger.error(f"error, {k} is not in the checkpoint")
            new_sd[k] = v
    target.load_state_dict(new_sd)
    logger.info(f"Loaded checkpoint '{checkpoint_path}' (iteration {iteration})")
    return model, optimizer, learning_rate, iteration


def latest_checkpoint_path(dir_path, regex="G_*.pth"):
    f_list = glob.glob(os.path.join(dir_path, regex))
    latest = [x for x in f_list if "latest" in x]
    if latest:
        return latest[0]
    f_list.sort(key=lambda f: int("".join(filter(str.isdigit, f))))
    return f_list[-1]       # IndexError when empty, like the reference (caught by the caller -> pretrained path)


def save_with_torch(obj, path):
    d, name = os.path.dirname(path), os.path.basename(path)
    tmp = os.path.join(d or ".", f".{time.time()}.{os.getpid()}.tmp")
    torch.save(obj, tmp)
    shutil.move(tmp, os.path.join(d, name))


def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    logger.info(f"Saving model and optimizer state at iteration {iteration} to {checkpoint_path}")
    target = model.module if hasattr(model, "module") else model
    sd = {k: v.detach().cpu().clone() for k, v in target.state_dict().items()}
    save_with_torch({"model": sd, "iteration": iteration, "optimizer": optimizer.state_dict(),
                     "learning_rate": learning_rate}, checkpoint_path)
