"""Import the reference's own hot-path modules (THIS container only).

Test infrastructure (see oracle/__init__.py).  /root/reference does not exist
on the GPU box; everything here returns None there and callers skip.
The reference needs ``torchvision`` (network.py:10) and ``cv2`` (utils.py:14),
neither of which is installed: ``sys.modules`` gets (a) oracle.standin_resnet
as ``torchvision.models.resnet34`` and (b) a ``cv2`` object exposing only
``setNumThreads`` (all utils.py touches at import).  Nothing is written to
/root/reference.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "footprints"))


def load_reference():
    """-> (network_module, losses_module) of the reference, or None when absent."""
    if not available():
        return None
    from . import standin_resnet
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvm.resnet34 = standin_resnet.resnet34
        tv.models = tvm
        tv.transforms = types.ModuleType("torchvision.transforms")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
        sys.modules["torchvision.transforms"] = tv.transforms
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.setNumThreads = lambda n: None
        sys.modules["cv2"] = cv2
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    keep = {k: os.environ.get(k) for k in ("MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "OMP_NUM_THREADS")}
    sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
    network = importlib.import_module("footprints.network")
    losses = importlib.import_module("footprints.training.losses")
    for k, v in keep.items():               # utils.py:16-18 pins threads to 1 at import; undo
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return network, losses


def load_reference_metrics():
    """-> the reference's footprints.evaluation.evaluate_model module, or None when /root/reference is absent.
    Its import needs cv2 (stand-in above), tqdm (installed) and skimage.morphology.convex_hull_image (only used by the
    Matterport convex-hull helper, never by evaluate_mask / evaluate_depth): a module object that raises if called."""
    if load_reference() is None:
        return None
    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        skm = types.ModuleType("skimage.morphology")

        def convex_hull_image(im):
            raise RuntimeError("skimage is not installed")
        skm.convex_hull_image = convex_hull_image
        sk.morphology = skm
        sys.modules["skimage"] = sk
        sys.modules["skimage.morphology"] = skm
    return importlib.import_module("footprints.evaluation.evaluate_model")
