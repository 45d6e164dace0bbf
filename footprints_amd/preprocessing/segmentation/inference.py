"""Ground-segmentation inference -- counterpart of footprints/preprocessing/segmentation/inference.py:60-84 (`test_batch`): the full
resolution logit map through a sigmoid, as a numpy array [B,1,H,W].  Dataset readers / file writers stay the reference's."""
import torch

from .network import Segmentor


class InferenceManager:
    def __init__(self, load_path=None, use_PSP=False, model=None):
        self.model = model if model is not None else Segmentor(pretrained=False, use_PSP=use_PSP)
        if load_path is not None:
            self.model.load_state_dict(torch.load(load_path, map_location="cpu"))        # inference.py:93-101
        self.model.cuda().eval()

    def test_batch(self, inputs):
        with torch.no_grad():
            preds = self.model(inputs["image"].cuda(non_blocking=True))
            return torch.sigmoid(preds[3][:, 0:1]).cpu().numpy()                          # "just take max resolution prediction"
