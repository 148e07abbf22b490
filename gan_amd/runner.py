"""Run-directory, logging, metric-file, figure and epoch-loop plumbing shared by the Pix2Pix and CycleGAN drivers.

What is kept from the reference is the CONTRACT: `<output>/<YYYY-MM-DD-HHhMM>/` with
`logs/{Log.txt,config.json,train_metrics.json,val_metrics.json}`, `training_checkpoints/`, `test_images/epoch_N.png`,
`final_test_imgs/imgK.png`, `prediction_images/imgK.png`, `figs/<Model> <Loss name>.png`, the checkpoint cadence
(every 5th epoch and the last, pix2pix.py:308-317) and the `.` tick per 100 mini-batches.  The code is ours: one
`Run` object owns the directory, one epoch driver serves both models, losses stay on the device and are drained once per
epoch (the reference pulls `.numpy()` on every loss of every step, pix2pix.py:276-279).
"""
from __future__ import annotations

import json
import os
import sys
import time
from datetime import datetime

import numpy as np
import torch


class Run:
    """One timestamped run directory.  `strict_logs`: Pix2Pix refuses to reuse an existing logs/ directory
    (pix2pix.py:392), CycleGAN does not (cycle_gan.py:428)."""

    def __init__(self, output_root: str, log_to_file: bool, strict_logs: bool, writer: bool = True):
        """writer=False: a non-zero rank of a data-parallel run - it creates nothing on disk (rank 0 alone owns the run
        directory, logs, checkpoints and figures) and keeps its console."""
        self.writer = writer
        self.root = os.path.join(output_root, datetime.now().strftime("%Y-%m-%d-%Hh%M"))
        self.logs = os.path.join(self.root, 'logs')
        self._saved = None
        if not writer:
            return
        os.makedirs(self.root, exist_ok=True)
        os.makedirs(self.logs, exist_ok=not strict_logs)
        if log_to_file:
            self._saved = (sys.stdout, sys.stderr)
            sys.stdout = sys.stderr = open(os.path.join(self.logs, "Log.txt"), "w")

    def dir(self, name: str, fresh: bool = False) -> str:
        d = os.path.join(self.root, name)
        if self.writer:
            os.makedirs(d, exist_ok=not fresh)
        return d

    def write_json(self, name: str, obj) -> None:
        if not self.writer:
            return
        with open(os.path.join(self.logs, name), 'w') as f:
            json.dump(obj, f)

    def close(self) -> None:
        if self._saved is not None:
            log = sys.stdout
            sys.stdout, sys.stderr = self._saved
            log.close()
            self._saved = None


def save_panels(path: str, panels, gray: bool) -> None:
    """Side-by-side image panels [(title, HWC array in [-1, 1])] -> one PNG at dpi 200."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(1, len(panels), figsize=(6 * len(panels) - (3 if len(panels) == 3 else 0), 6), squeeze=False)
    for ax, (title, img) in zip(axes[0], panels):
        img = np.clip(np.asarray(img, dtype=np.float32) * 0.5 + 0.5, 0.0, 1.0)
        ax.imshow(img[..., 0], cmap='gray', vmin=0.0, vmax=1.0) if gray else ax.imshow(img)
        ax.set_title(title)
        ax.set_axis_off()
    fig.tight_layout()
    fig.savefig(path, dpi=200)
    plt.close(fig)


def plot_loss_curves(train: dict, val: dict, model_name: str, out_dir: str) -> None:
    """One figure per loss key: training and validation means by epoch (epochs counted from 1), `<Model> <key>.png`."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    os.makedirs(out_dir, exist_ok=True)
    for key, tr in train.items():
        epochs = np.arange(1, len(tr) + 1)
        fig, ax = plt.subplots(figsize=(10, 8), dpi=80)
        ax.plot(epochs, tr, alpha=0.7, label='Training')
        ax.plot(epochs[:len(val[key])], val[key], alpha=0.7, label='Validation')
        ax.set(xlabel='Epoch', ylabel='Loss', title=f'{model_name} {key}')
        ax.legend()
        fig.tight_layout()
        fig.savefig(os.path.join(out_dir, f'{model_name} {key}.png'), dpi=200)
        plt.close(fig)


def run_epochs(epochs: int, keys, train_batches, val_batches, step, on_checkpoint, on_sample, headline, epoch_mean=None, after_pass=None):
    """Epoch driver.  train_batches()/val_batches() yield the positional arguments of `step(*args, training)`, which
    returns the per-step loss tensors (device); `headline` = (train key, val key) printed per epoch.
    epoch_mean(acc, n) -> mean loss vector of a pass (data-parallel runs: over all ranks' steps, gan_amd.ddp.mean_over_ranks).
    Returns (train_cost_functions, val_cost_functions): {key: [epoch mean, ...]}."""
    hist = {k: [] for k in keys}, {k: [] for k in keys}
    t0 = time.time()
    for epoch in range(1, epochs + 1):
        sums = []
        for batches, training in ((train_batches, True), (val_batches, False)):
            acc, n = None, 0
            for args in batches():
                losses = torch.stack(tuple(step(*args, training)))
                acc = losses if acc is None else acc + losses           # stays on the device: no per-step sync
                n += 1
                if training and n % 100 == 0:
                    print('.', end='', flush=True)
            mean = epoch_mean(acc, n) if epoch_mean is not None else ((acc / n) if n else None)
            sums.append(mean.cpu().tolist() if mean is not None else [float('nan')] * len(keys))      # one drain per pass
            if after_pass is not None:
                after_pass()          # device-side error flags are read where the pass has been drained anyway (layer stacks: grid-barrier timeout)
        for h, means in zip(hist, sums):
            for k, v in zip(keys, means):
                h[k].append(v)
        last = epoch == epochs
        if epoch % 5 == 0 or last:
            on_checkpoint()
            if not last:
                on_sample(epoch)
        print(f'\nCumulative training duration at end of epoch {epoch}: {(time.time() - t0) / 60:.2f} min')
        print(f"Train {headline[0]}: {hist[0][headline[0]][-1]:.2f}, {headline[1]}: {hist[0][headline[1]][-1]:.2f}; "
              f"val {headline[0]}: {hist[1][headline[0]][-1]:.2f}, {headline[1]}: {hist[1][headline[1]][-1]:.2f}\n", flush=True)
    return hist
