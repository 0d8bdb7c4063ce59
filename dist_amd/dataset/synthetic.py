"""Synthetic clip loader (there are no datasets in this environment): yields the reference loader's batch
structure (dataset/base/base_dataset.py:441): (inputs {"video": [b,3,T,H,W]}, labels {"supervised": [b]},
index, meta), already on the GPU (DATA.SYNTHETIC_HOST: in host memory, as the reference's loader yields them), sharded by rank."""
import torch


class SyntheticLoader:
    def __init__(self, cfg, split, rank=0, world=1):
        self.cfg = cfg
        n = max(1, world)
        self.batch = (cfg.TRAIN.BATCH_SIZE if split == "train" else cfg.TEST.BATCH_SIZE) // n
        self.iters = int(getattr(cfg.DATA, "SYNTHETIC_ITERS", 8))
        self.T, self.res = cfg.DATA.NUM_INPUT_FRAMES, cfg.DATA.TRAIN_CROP_SIZE
        self.K = cfg.VIDEO.HEAD.NUM_CLASSES
        self.gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
        self.split, self.rank, self.world = split, rank, n
        self.views = int(cfg.TEST.NUM_ENSEMBLE_VIEWS) * int(cfg.TEST.NUM_SPATIAL_CROPS) if split == "test" else 1
        self.dataset = range(self.iters * self.batch * n)

    def __len__(self):
        return self.iters

    def __iter__(self):
        for it in range(self.iters):
            video = torch.randn(self.batch, 3, self.T, self.res, self.res, device="cuda", generator=self.gen)
            idx = (it * self.world + self.rank) * self.batch + torch.arange(self.batch, device="cuda")
            if self.split == "test":          # every view of a video carries the video's label (utils/meters.py:96-100)
                labels = ((idx // self.views) * 7919) % self.K
            else:
                labels = torch.randint(0, self.K, (self.batch,), device="cuda", generator=self.gen)
            if getattr(self.cfg.DATA, "SYNTHETIC_HOST", False):      # the reference loader's hand-over: host tensors (the same values as the resident form)
                video, labels, idx = video.cpu(), labels.cpu(), idx.cpu()
            yield {"video": video}, {"supervised": labels}, idx, {}


def build_loader(cfg, split):
    from ..utils import distributed as du
    if not getattr(cfg.DATA, "SYNTHETIC", False):
        raise NotImplementedError("video decoding (decord) is outside the hot path; set DATA.SYNTHETIC true")
    return SyntheticLoader(cfg, split, du.get_rank(), du.get_world_size())


def label_texts(cfg, device="cuda", vocab=49408):
    """Stand-in for the CLIP BPE tokenisation of the class names (dataset/utils/simple_tokenizer.py:138-179):
    deterministic token ids [K,77] with an EOT maximum per row."""
    K = cfg.VIDEO.HEAD.NUM_CLASSES
    g = torch.Generator().manual_seed(7)
    t = torch.zeros(K, 77, dtype=torch.long)
    t[:, 0] = vocab - 2
    t[:, 1:5] = torch.randint(1, vocab - 3, (K, 4), generator=g)
    t[:, 5] = vocab - 1
    return t.to(device)
