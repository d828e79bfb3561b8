#!/usr/bin/env python3
"""Record the ``state_dict`` contract of the reference's model factory, per ``configs/*.yaml``.

Run in the build container only (the reference checkout is at /root/reference there):

    python tests/golden/make_state_dict_keys.py

For every (model name, YAML) pair that ``train.py:create_model`` (train.py:81-326) can build, the script calls that
function of the reference itself, unchanged, and writes ``tests/golden/state_dict_keys.json``:

    {case: {"model": name, "yaml": file name, "config": the parsed YAML, "keys": {key: [shape, dtype]},
            "standin_prefixes": [...]}}

The YAML dictionaries are stored so that the test needs nothing from the reference checkout.  torchvision, timm and cv2
are not installed here: empty in-process stub modules are registered under those names (nothing is fetched), and
where the reference builds a third-party backbone at construction (``models.resnet152``,
``timm.create_model('swin_tiny_patch4_window7_224')``) the oracle's restatement of that architecture stands in.  Keys
under ``standin_prefixes`` therefore come from that restatement, not from torchvision / timm themselves: they pin the
reference's wrapping (prefixes, ``nn.Sequential`` indices of ``children()[:-1]``), not the third-party key names.
``checkpoint.pretrained`` is forced to false for CRNN (true would select the VGG16 branch, which needs torchvision's
weights); data paths are placeholders as in the YAML files.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # nothing is written into the reference checkout

import torch
import yaml

REF = os.environ.get("VC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)

CASES = [  # (case id, --model name, YAML under configs/)
    ("Conv3D", "Conv3D", "Conv3D_train.yaml"),
    ("CRNN", "CRNN", "CRNN_train.yaml"),
    ("ResNetCRNN", "ResNetCRNN", "ResNetCRNN_train.yaml"),
    ("swintransformer-RNN", "swintransformer-RNN", "swintransformer-RNN_train.yaml"),
    ("swintransformer-RNN/baseline", "swintransformer-RNN", "swintransformer_RNN_baseline.yaml"),
    ("clusterSwin_TCGlstm", "clusterSwin_TCGlstm", "clusterSwin_TCGlstm.yaml"),
]


def _stub(name):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
    return sys.modules[name]


def main():
    from oracle import torch_oracle as O
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.transforms = _stub("torchvision.transforms")
    _stub("cv2")

    def resnet152(weights=None):  # torchvision's child layout around the oracle's restatement of the trunk
        trunk = O.make_resnet_trunk("resnet152")
        m = torch.nn.Module()
        for n, child in zip(("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool"), trunk):
            m.add_module(n, child)
        m.add_module("fc", torch.nn.Linear(2048, 1000))
        return m
    tv.models.resnet152 = resnet152
    tv.models.ResNet152_Weights = types.SimpleNamespace(IMAGENET1K_V1=None)

    os.chdir(REF)  # train.py appends the relative path 'utils' to sys.path
    sys.path.insert(0, os.path.join(REF, "utils"))
    sys.path.insert(0, os.path.join(REF, "models", "swintransformer_RNN"))
    sys.path.insert(0, REF)
    import train as ref_train  # noqa: E402  (the reference's train.py)
    import utils.common.model_components as mc  # noqa: E402
    from utils.config_loader import ConfigLoader  # noqa: E402
    swin_t = lambda name, pretrained=True, num_classes=0: O.SwinT()  # noqa: E731
    mc.create_model = swin_t                    # ClusterSwinEncoder (model_components.py:412)
    _stub("timm").create_model = swin_t         # SwinTransformerEncoder imports timm inside its constructor (:253-254)

    out = {}
    for case, model_name, yml in CASES:
        path = os.path.join(REF, "configs", yml)
        loader = ConfigLoader(path)
        cfg = loader.config
        if model_name == "CRNN":
            cfg.setdefault("checkpoint", {})["pretrained"] = False
        if model_name == "swintransformer-RNN":
            # its Config insists on the four paths being present (UCF101_SwinCRNN_fixed.py:166-173)
            cfg.setdefault("data", {}).setdefault("data_path", "/path/to/your/dataset")
            cfg["data"].setdefault("action_name_path", "/path/to/action_names.txt")
            cfg.setdefault("checkpoint", {}).setdefault("save_dir", "/path/to/save/models")
            cfg.setdefault("logging", {}).setdefault("log_dir", "/path/to/logs")
        torch.manual_seed(0)
        model = ref_train.create_model(model_name, loader)
        sd = model.state_dict()
        standin = []
        if model_name == "ResNetCRNN":
            standin = ["encoder.resnet."]
        elif model_name == "swintransformer-RNN":
            standin = ["encoder.swin."]
        elif model_name == "clusterSwin_TCGlstm":
            standin = ["encoder.backbone."]
        out[case] = {"model": model_name, "yaml": yml, "config": cfg, "standin_prefixes": standin,
                     "keys": {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}}
        print(f"{case}: {len(sd)} tensors, {sum(v.numel() for v in sd.values()) / 1e6:.2f} M elements")
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(OUT, "state_dict_keys.json"))


if __name__ == "__main__":
    main()
