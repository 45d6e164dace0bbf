"""predict_simple -- drop-in CLI of footprints/predict_simple.py:29-141 on the HIP engine.

    python -m footprints_amd.predict_simple --image P --model {kitti,matterport,handheld}
                                            [--no_save_vis] [--save_dir D] [--weights W]

Same artefacts as the reference: <save_dir>/outputs/<stem>.npy = float32 [4,H,W] (channels 0,1 LOGITS,
2,3 sigmoid disparities, predict_simple.py:67-73) and <save_dir>/visualisations/<stem>.jpg.  Host plumbing is
restated without the libraries this image lacks: torchvision's Resize(ANTIALIAS)+ToTensor become PIL LANCZOS +
numpy (identical arithmetic), cv2.resize(bilinear) becomes PIL BILINEAR, cv2.imwrite becomes PIL save.
The reference quirk of thresholding the *logit* at 0.5 for the visualisation mask (predict_simple.py:77) is
kept.  `--no_cuda` is accepted for CLI compatibility but raises: this package has no CPU compute path.
"""
import argparse
import os

import numpy as np
import torch
from PIL import Image

from .model_manager import ModelManager
from .utils import MODEL_DIR, model_folder, pil_loader, sigmoid_to_depth

MODEL_HEIGHT_WIDTH = {"kitti": (192, 640), "matterport": (512, 640), "handheld": (256, 448)}   # predict_simple.py:21-25
IMAGE_EXTENSIONS = {".jpg", ".jpeg", ".png"}


def preprocess(pil_image, height_width):
    """Resize((H,W), ANTIALIAS) + ToTensor + [None] (predict_simple.py:51-60) -> float32 [1,3,H,W] in [0,1]."""
    h, w = height_width
    resized = pil_image.resize((w, h), Image.LANCZOS)
    arr = np.asarray(resized, dtype=np.uint8).astype(np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1)[None].contiguous()


class InferenceManager:
    def __init__(self, model_name, save_dir, use_cuda=True, save_visualisations=True, weights_path=None, model_manager=None):
        if not use_cuda or not torch.cuda.is_available():
            raise RuntimeError("footprints_amd.predict_simple needs a MI355X: the package has no CPU compute path "
                               "(--no_cuda is accepted for CLI compatibility only)")
        self.model_name = model_name
        self.height_width = MODEL_HEIGHT_WIDTH[model_name]
        if model_manager is None:
            model_manager = ModelManager(is_inference=True, use_cuda=True)
            model_manager.load_model(weights_path=weights_path or model_folder(model_name))
        self.model_manager = model_manager
        self.model_manager.model.eval()
        self.model_manager.model.inference_scales = ("1/1",)      # only the full-resolution prediction is consumed below
        self.save_dir = save_dir
        os.makedirs(os.path.join(save_dir, "outputs"), exist_ok=True)
        self.save_visualisations = save_visualisations
        if save_visualisations:
            os.makedirs(os.path.join(save_dir, "visualisations"), exist_ok=True)

    def predict_array(self, pil_image):
        x = preprocess(pil_image, self.height_width).cuda()
        with torch.no_grad():
            pred = self.model_manager.model(x)
        return pred["1/1"].cpu().numpy().squeeze(0)          # [4,H,W]

    def predict_for_single_image(self, image_path):
        print("Predicting for {}".format(image_path))
        original = pil_loader(image_path)
        pred = self.predict_array(original)
        filename, _ = os.path.splitext(os.path.basename(image_path))
        npy_save_path = os.path.join(self.save_dir, "outputs", filename + ".npy")
        print("└> Saving predictions to {}".format(npy_save_path))
        np.save(npy_save_path, pred)
        if self.save_visualisations:
            vis = self.visualise(pred, original)
            vis_save_path = os.path.join(self.save_dir, "visualisations", filename + ".jpg")
            print("└> Saving visualisation to {}".format(vis_save_path))
            Image.fromarray(vis).save(vis_save_path, quality=95)
        return pred

    @staticmethod
    def visualise(pred, original):
        """predict_simple.py:75-92: mask = resized LOGIT > 0.5 (quirk), plasma colormap of the hidden depth."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        size = original.size
        resize = lambda a: np.asarray(Image.fromarray(a.astype(np.float32), mode="F").resize(size, Image.BILINEAR))
        hidden_ground = resize(pred[1]) > 0.5
        hidden_depth = resize(sigmoid_to_depth(pred[3]))
        rgb = np.array(original) / 255.0
        if hidden_ground.any():
            _max, _min = hidden_depth[hidden_ground].max(), hidden_depth[hidden_ground].min()
            hidden_depth = (hidden_depth - _min) / max(_max - _min, 1e-12)
        cmap = plt.get_cmap("plasma", 256)(np.clip(hidden_depth, 0, 1))[:, :, :3]
        m = hidden_ground[:, :, None]
        vis = rgb * (1 - m) + cmap * m
        return (vis * 255).astype(np.uint8)

    def _image_files(self, folder):
        """image files directly inside `folder` (same extension filter as predict_simple.py:21-25), in sorted order"""
        return [os.path.join(folder, name) for name in sorted(os.listdir(folder))
                if os.path.splitext(name)[1].lower() in IMAGE_EXTENSIONS]

    def predict(self, image_path):
        """--image may name one file or a folder of images (predict_simple.py:94-110); returns the number of images written"""
        if os.path.isdir(image_path):
            targets = self._image_files(image_path)
        elif os.path.isfile(image_path):
            targets = [image_path]
        else:
            raise FileNotFoundError("--image: no such file or folder: %r" % (image_path,))
        for path in targets:
            self.predict_for_single_image(path)
        return len(targets)


def parse_args(argv=None):
    """the reference's flag names and defaults (predict_simple.py:113-131) -- they are the CLI contract -- plus --weights"""
    ap = argparse.ArgumentParser(prog="footprints_amd.predict_simple",
                                 description="Footprints prediction for one image or a folder of images on the HIP engine.")
    ap.add_argument("--image", required=True, type=str, help="image file, or a folder whose image files are all predicted")
    ap.add_argument("--model", type=str, choices=sorted(MODEL_HEIGHT_WIDTH), help="which released model: fixes the input resolution")
    ap.add_argument("--no_cuda", action="store_true", help="accepted for compatibility; raises (no CPU compute path here)")
    ap.add_argument("--no_save_vis", action="store_true", help="write only the .npy predictions, no .jpg overlays")
    ap.add_argument("--save_dir", type=str, default="predictions", help="output root (outputs/ and visualisations/ below it)")
    ap.add_argument("--weights", type=str, default=None,
                    help="folder holding model.pth (default: %s/<model>; this build cannot download)" % MODEL_DIR)
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    manager = InferenceManager(model_name=args.model, use_cuda=torch.cuda.is_available() and not args.no_cuda,
                               save_visualisations=not args.no_save_vis, save_dir=args.save_dir, weights_path=args.weights)
    manager.predict(image_path=args.image)


if __name__ == "__main__":
    main()
