"""Host-side helpers mirrored from the reference's footprints/utils.py (only what the hot path needs)."""
import os

from PIL import Image

MODEL_DIR = "models"


def sigmoid_to_depth(disp, min_depth=0.1, max_depth=100):
    """utils.py:36-42 -- convert sigmoid disparity to depth."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return 1 / scaled_disp


def pil_loader(path):
    """utils.py:91-94."""
    with open(path, "rb") as fh:
        with Image.open(fh) as img:
            return img.convert("RGB")


def model_folder(model_name):
    """The reference downloads <MODEL_DIR>/<name>.zip from GCS (utils.py:105-141); there is no network here,
    so the weights must already be unpacked at models/<name>/model.pth."""
    path = os.path.join(MODEL_DIR, model_name)
    if not os.path.exists(os.path.join(path, "model.pth")):
        raise FileNotFoundError(
            "pretrained weights not found at %s/model.pth -- this build has no network access; unpack the released "
            "'%s' zip there (the checkpoint format is the reference's)" % (path, model_name))
    return path
