"""Input pipeline of the VAE training path: TIFF files -> device batches."""
from .loader import (DeviceImageLoader, create_regression_dataloaders, create_vae_dataloaders, list_tif_paths,
                     shard_indices, split_paths)
from .tiff import read_tiff, write_tiff

__all__ = ["DeviceImageLoader", "create_regression_dataloaders", "create_vae_dataloaders", "list_tif_paths", "shard_indices", "split_paths", "read_tiff",
           "write_tiff"]
