"""Command-line surface -- the flags of the reference's footprints/options.py:13-128 (same names, types, choices and
defaults), so `python -m footprints_amd.main <reference flags>` parses unchanged.  Two additions for this build:
`--synthetic_steps N` (batches per epoch from the synthetic loader when no dataset is mounted: there is no KITTI /
Matterport data on the GPU box) and `--device_augment` (row N3: flip / colour jitter / label assembly on the GPU)."""
import argparse


class Options:
    def __init__(self):
        self.options = None
        p = self.parser = argparse.ArgumentParser()
        # universal (options.py:13-35)
        p.add_argument("--mode", type=str, choices=["train", "inference"], default="train", help="training or inference mode")
        p.add_argument("--height", type=int, default=192, help="height of input images")
        p.add_argument("--width", type=int, default=640, help="width of input images")
        p.add_argument("--depth_range", nargs="+", type=float, default=[0.1, 100], help="range of depth values")
        # training (options.py:37-112)
        p.add_argument("--training_dataset", type=str, choices=["kitti", "matterport"], default="kitti")
        p.add_argument("--epochs", type=int, default=10)
        p.add_argument("--log_freq", type=int, default=250, help="steps between validation passes / logs")
        p.add_argument("--val_batches", type=int, default=10, help="validation batches to run and average over")
        p.add_argument("--batch_size", type=int, default=12)
        p.add_argument("--lr", type=float, default=1e-4)
        p.add_argument("--use_footprint_prior", action="store_true", help="dead flag in the reference too (options.py:66-69)")
        p.add_argument("--footprint_prior", type=float, default=0.25, help="weight of negative hidden-footprint labels")
        p.add_argument("--no_depth_mask", action="store_true")
        p.add_argument("--moving_objects_method", type=str, choices=["none", "ours"], default="ours")
        p.add_argument("--project_down_baseline", action="store_true")
        p.add_argument("--num_workers", type=int, default=8)
        p.add_argument("--config_path", type=str, default="paths.yaml")
        p.add_argument("--model_name", type=str, default="model")
        p.add_argument("--log_path", type=str, default="./logs")
        # test (options.py:114-128)
        p.add_argument("--inference_data_type", choices=["kitti", "matterport"], default="kitti")
        p.add_argument("--load_path", type=str, help="the model path to load from")
        p.add_argument("--inference_save_path", default=None)
        p.add_argument("--save_test_visualisations", action="store_true")
        # this build
        p.add_argument("--synthetic_steps", type=int, default=0, help="batches per epoch of synthetic data (no dataset mounted)")
        p.add_argument("--device_augment", action="store_true", help="flip / colour jitter / label assembly on the GPU (row N3)")

    def parse(self, argv=None):
        self.options = self.parser.parse_args(argv)
        return self.options
