"""ModelManager -- drop-in for footprints/model_manager.py:14-56 (construction, device move, optimiser,
scheduler, `model.pth` / `optimiser.pth` checkpoints in the reference's format).

Differences, all deliberate: Adam is the fused HIP optimiser (same math, same checkpoint format); the
reference's CPU `load_state_dict(..., map_location=...)` TypeError (model_manager.py:37) is fixed by mapping at
`torch.load`; `pretrained=True` cannot download ImageNet weights here and degrades to random init.
"""
import os

import torch

from .network import FootprintNetwork
from .optim import FusedAdam


class ModelManager:
    def __init__(self, save_folder=None, use_cuda=True, is_inference=False, learning_rate=1e-4, lr_step_size=10):
        self.save_folder = save_folder
        self.use_cuda = use_cuda
        self.model = FootprintNetwork(pretrained=True)
        if self.use_cuda:
            self.model.cuda()
        if not is_inference:
            self.optimiser = FusedAdam(self.model, lr=learning_rate)
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimiser, step_size=lr_step_size)

    def load_model(self, weights_path, load_optimiser=False):
        print("loading model weights from {}...".format(weights_path))
        dev = next(self.model.parameters()).device
        weights = torch.load(os.path.join(weights_path, "model.pth"), map_location=dev)
        self.model.load_state_dict(weights)
        print("successfully loaded weights!")
        if load_optimiser:
            print("loading optimiser...")
            weights = torch.load(os.path.join(weights_path, "optimiser.pth"), map_location=dev)
            self.optimiser.load_state_dict(weights)
            print("successfully loaded optimiser!")

    def save_model(self, folder_name):
        save_path = os.path.join(self.save_folder, folder_name)
        print("saving weights to {}...".format(save_path))
        os.makedirs(save_path, exist_ok=True)
        # clone: parameters are views of one flat buffer; a checkpoint must hold plain per-tensor storages
        state = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        torch.save(state, os.path.join(save_path, "model.pth"))
        torch.save(self.optimiser.state_dict(), os.path.join(save_path, "optimiser.pth"))
        print("success!")
