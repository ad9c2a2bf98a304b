"""MI355X-native conv-VAE train path (drop-in for moving-mnist-vae's model.VAE / main.train).

The directory name carries a hyphen (it mirrors the upstream repository name), so import it
with ``importlib.import_module("moving-mnist-vae_amd")``.  Importing the package does not need
a GPU; constructing / running the model does, and fails loudly when the HIP library is absent.
"""
from .main import (MovingMNISTClips, checkpoint_variant, clips_from_npz_array, generate, generate_only_pixelcnn, load_checkpoint, quantise_frames, save_checkpoint, select_model,  # noqa: F401
                   train)


def __getattr__(name):
    if name in ("VAE", "FusedAdam", "GradSync", "Communicator", "MmvaeError"):
        from . import model as _m
        return getattr(_m, name)
    raise AttributeError(name)
