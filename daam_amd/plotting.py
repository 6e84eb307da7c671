"""Overlay of a word heat map on its image: what ``WordHeatMap.plot_overlay`` draws (the reference's figure,
``daam/heatmap.py:20-53``: the heat map in the 'jet' colour map underneath, the image on top with opacity
``1 - heat``).  Presentation only, downstream of the extraction path; matplotlib is imported lazily."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

__all__ = ['plot_overlay_heat_map', 'overlay_layers']


def overlay_layers(image, heat_map: torch.Tensor, crop: Optional[int] = None, color_normalize: bool = True):
    """The two raster layers of the figure as numpy arrays: ``(heat [h, w] float32, rgba [h, w, 4] float32)`` -- the
    heat values (clamped to [0, 1] unless ``color_normalize``) and the image with alpha ``1 - heat``."""
    pixels = np.asarray(image)
    heat = heat_map.detach().squeeze().to(torch.float32).cpu().numpy()
    if crop is not None:
        window = (slice(crop, -crop), slice(crop, -crop))
        heat, pixels = heat[window], pixels[window]
    if not color_normalize:
        heat = np.clip(heat, 0.0, 1.0)
    rgb = pixels.astype(np.float32) / 255.0
    alpha = (1.0 - heat)[..., None].astype(np.float32)
    return heat, np.concatenate([rgb, alpha], axis=-1)


def plot_overlay_heat_map(im, heat_map: torch.Tensor, word=None, out_file=None, crop=None, color_normalize=True, ax=None):
    import matplotlib
    matplotlib.use('Agg', force=False)
    from matplotlib import pyplot as plt
    heat, rgba = overlay_layers(im, heat_map, crop, color_normalize)
    standalone = ax is None
    if standalone:
        plt.clf()
        plt.rcParams.update({'font.size': 24})
        ax = plt.gca()
    limits = {} if color_normalize else dict(vmin=0.0, vmax=1.0)
    ax.imshow(heat, cmap='jet', **limits)
    ax.imshow(rgba)
    if word is not None:
        ax.set_title(word)
    if out_file is not None:
        plt.savefig(out_file)
