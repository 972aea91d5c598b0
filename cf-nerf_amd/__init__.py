"""cfnerf_amd - MI355X-native (gfx950) CF-NeRF ray-batch hot path behind the reference's Python interface.

Import as ``import cfnerf_amd`` (repo-root shim; the directory name ``cf-nerf_amd`` is not an identifier).
"""
from . import _lib  # noqa: F401
from . import train  # noqa: F401
from . import evaluate  # noqa: F401
from .evaluate import gather_rows, render_path_train, render_uncertainty, row_shard, sparsification_plot  # noqa: F401
from . import data  # noqa: F401
from .data import RayPool  # noqa: F401
from .api import default_args, save_checkpoint  # noqa: F401
from .api import (Embedder, NeRF_Flows, batchify, batchify_rays, create_nerf, get_embedder, get_rays, img2mse,  # noqa: F401
                  mse2psnr, ndc_rays, param_layout, raw2outputs, render, render_rays, run_network, t_vals_table)

__all__ = ["Embedder", "NeRF_Flows", "batchify", "batchify_rays", "create_nerf", "get_embedder", "get_rays",
           "img2mse", "mse2psnr", "ndc_rays", "param_layout", "raw2outputs", "render", "render_rays", "run_network",
           "t_vals_table", "render_path_train", "render_uncertainty", "gather_rows", "row_shard", "sparsification_plot", "RayPool", "save_checkpoint", "default_args"]
