"""Entry point -- counterpart of footprints/main.py:8-26: `python -m footprints_amd.main --mode train ...`.

The KITTI / Matterport dataset readers are outside this build's scope: training runs over any
iterable of batches with the reference schema; with `--synthetic_steps N` it runs over the synthetic loader the benchmark
uses.  Inference mode needs a loader of {'image', 'idx'} batches and is therefore only reachable programmatically
(`footprints_amd.evaluation.inference.InferenceManager(...).run(loader)`)."""
from .options import Options
from .parallel import DistContext
from .training.train import SyntheticLoader, TrainManager


def main(argv=None):
    opt = Options().parse(argv)
    if opt.mode == "train":
        # FIRST thing under a launcher: select this rank's GPU (LOCAL_RANK) and join the host group before ANY loader, assembler,
        # pinned buffer or stream exists -- a DeviceBatchAssembler built earlier would put its slots and its copy stream on GPU 0 on every rank
        ctx = DistContext.from_env()
        if ctx.is_main:
            print("In training mode!")
        if opt.synthetic_steps <= 0:
            raise SystemExit("no dataset readers in this build: pass --synthetic_steps N, or construct TrainManager(options, "
                             "train_loader=..., val_loader=...) with your own loaders")
        if opt.device_augment and opt.training_dataset == "kitti":
            # row N3: host samples as the file readers deliver them -> pinned double-buffered H2D -> flip / jitter / label kernels
            from .datasets import DeviceBatchAssembler, DeviceLoader, SyntheticSampleSource
            asm = DeviceBatchAssembler(opt.batch_size, opt.height, opt.width, dataset="kitti", no_depth_mask=opt.no_depth_mask,
                                       project_down_baseline=opt.project_down_baseline, moving_objects_method=opt.moving_objects_method)
            train = DeviceLoader(SyntheticSampleSource(opt.batch_size, opt.height, opt.width, opt.synthetic_steps), asm, is_train=True)
        else:
            train = SyntheticLoader(opt.batch_size, opt.height, opt.width, opt.synthetic_steps)
        val = SyntheticLoader(opt.batch_size, opt.height, opt.width, max(1, opt.val_batches), seed=11)
        # under torch.distributed.run (WORLD_SIZE > 1) TrainManager turns into one replica of a data-parallel run: --batch_size is the
        # per-GPU batch, --synthetic_steps the number of GLOBAL batches per epoch (each rank trains on every world-th one)
        TrainManager(opt, train_loader=train, val_loader=val, dist_context=ctx).train()
    elif opt.mode == "inference":
        raise SystemExit("inference mode needs a dataset loader: use footprints_amd.evaluation.inference.InferenceManager")
    else:
        raise NotImplementedError


if __name__ == "__main__":
    main()
