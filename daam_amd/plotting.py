"""Overlay plot of a word heat map on its image (reference ``daam/heatmap.py:20-53``).
Presentation only; matplotlib is imported lazily."""
from __future__ import annotations

import numpy as np
import torch

__all__ = ['plot_overlay_heat_map']


def plot_overlay_heat_map(im, heat_map: torch.Tensor, word=None, out_file=None, crop=None, color_normalize=True, ax=None):
    import matplotlib
    matplotlib.use('Agg', force=False)
    from matplotlib import pyplot as plt
    if ax is None:
        plt.clf()
        plt.rcParams.update({'font.size': 24})
        target = plt
    else:
        target = ax
    im = np.array(im)
    heat_map = heat_map.squeeze().float().cpu()
    if crop is not None:
        heat_map = heat_map[crop:-crop, crop:-crop]
        im = im[crop:-crop, crop:-crop]
    if color_normalize:
        target.imshow(heat_map.numpy(), cmap='jet')
    else:
        heat_map = heat_map.clamp(min=0, max=1)
        target.imshow(heat_map.numpy(), cmap='jet', vmin=0.0, vmax=1.0)
    rgba = torch.cat((torch.from_numpy(im).float() / 255, (1 - heat_map.unsqueeze(-1))), dim=-1)
    target.imshow(rgba)
    if word is not None:
        if ax is None:
            plt.title(word)
        else:
            ax.set_title(word)
    if out_file is not None:
        plt.savefig(out_file)
