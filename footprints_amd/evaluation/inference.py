"""Test-set inference -- counterpart of the reference's footprints/evaluation/inference.py (`InferenceManager.test_batch`
:99-123 and `InferenceDataset.save_result`, datasets/inference_dataset.py:35-43), SURVEY.md section 8(f) N1/N2.

The reference takes `model(image)['1/1']`, applies the sigmoid to the two mask channels on the host side of the graph, copies
fp32 to the CPU and lets every `save_result` cast to float16.  Here the network evaluates only the full-resolution heads
(`inference_scales`), the encoder BatchNorm is folded into the convolutions, and one kernel (`fp_pack_pred_fp16`) applies the
sigmoid and rounds to float16 on the device, so the D2H copy is half the bytes and `save_result` writes the array as is.
The dataset / dataloader side (KITTI, Matterport, handheld readers) is outside this build's scope.
"""
import os

import numpy as np
import torch

from .. import ops
from ..model_manager import ModelManager


class InferenceManager:
    def __init__(self, load_path=None, model_manager=None, save_path=None):
        if model_manager is None:
            model_manager = ModelManager(use_cuda=True, is_inference=True)
            if load_path is not None:
                model_manager.load_model(weights_path=load_path, load_optimiser=False)
        self.model_manager = model_manager
        self.model = model_manager.model
        self.model.eval()
        self.model.inference_scales = ("1/1",)          # "just take max resolution prediction" (inference.py:104)
        self.savepath = save_path

    def test_batch(self, inputs):
        """inputs['image']: [B,3,H,W] float tensor.  Returns a float16 numpy array [B,4,H,W]: sigmoid(mask logits), depth."""
        image = inputs["image"].cuda(non_blocking=True)
        with torch.no_grad():
            pred = self.model(image)["1/1"]
            return ops.pack_pred_fp16(pred).cpu().numpy()

    def save_result(self, filename, prediction, savepath=None):
        savepath = savepath or self.savepath
        os.makedirs(savepath, exist_ok=True)
        np.save(os.path.join(savepath, "{}.npy".format(filename)), np.asarray(prediction, dtype=np.float16))

    def run(self, loader):
        """loader yields dicts with 'image' and 'idx' (file stems), like the reference's InferenceDataset batches"""
        for inputs in loader:
            preds = self.test_batch(inputs)
            for i, pred in enumerate(preds):
                self.save_result(inputs["idx"][i], pred)
