"""YAML option files with the reference's keys (basicsr/utils/options.py:6-95)."""
from collections import OrderedDict
from os import path as osp

import yaml


def ordered_yaml():
    try:
        from yaml import CDumper as Dumper
        from yaml import CLoader as Loader
    except ImportError:
        from yaml import Dumper, Loader
    tag = yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG
    Dumper.add_representer(OrderedDict, lambda d, data: d.represent_dict(data.items()))
    Loader.add_constructor(tag, lambda l, node: OrderedDict(l.construct_pairs(node)))
    return Loader, Dumper


def parse(opt_path, is_train=True):
    """Same derived keys as the reference parser: is_train, datasets.*.phase/scale, path.*,
    debug-mode overrides for names containing 'debug' (options.py:31-95)."""
    with open(opt_path, mode="r") as f:
        Loader, _ = ordered_yaml()
        opt = yaml.load(f, Loader=Loader)
    opt["is_train"] = is_train
    if "datasets" in opt:
        for phase, dataset in opt["datasets"].items():
            phase = phase.split("_")[0]
            dataset["phase"] = phase
            if "scale" in opt:
                dataset["scale"] = opt["scale"]
            for key in ("dataroot_gt", "dataroot_lq"):
                if dataset.get(key) is not None:
                    dataset[key] = osp.expanduser(dataset[key])
    opt.setdefault("path", OrderedDict())
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)
    opt["path"]["root"] = opt["path"].get("root", osp.abspath(osp.join(__file__, osp.pardir, osp.pardir)))
    if is_train:
        root = osp.join(opt["path"]["root"], "experiments", opt["name"])
        opt["path"]["experiments_root"] = root
        opt["path"]["models"] = osp.join(root, "models")
        opt["path"]["training_states"] = osp.join(root, "training_states")
        opt["path"]["log"] = root
        opt["path"]["visualization"] = osp.join(root, "visualization")
        if "debug" in opt["name"]:
            if "val" in opt:
                opt["val"]["val_freq"] = 8
            opt.setdefault("logger", OrderedDict())
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        root = osp.join(opt["path"]["root"], "results", opt["name"])
        opt["path"]["results_root"] = root
        opt["path"]["log"] = root
        opt["path"]["visualization"] = osp.join(root, "visualization")
    return opt


def shapes_from_dataset_opt(ds):
    """(T, img_chn) implied by a dataset block: image_npy_dataset.py:48,211-232 (blur-VFI,
    return_deblur_voxel) and image_sharp_npy_dataset.py:52 (sharp-VFI).  SURVEY.md 3.4."""
    m = int(ds.get("num_end_interpolation", 1))
    n = int(ds.get("num_inter_interpolation", 1))
    if ds.get("return_deblur_voxel", False):
        return 2 * m + n, 2 * 3 + 2 * (m - 1)
    return n, 6
