"""The model factory over the HIP modules (video_classification_amd/factory.py) against the state_dict contract of
the reference's own factory: tests/golden/state_dict_keys.json holds, per configs/*.yaml, the parsed YAML and the
key -> (shape, dtype) list of the model that the reference's train.py:create_model built from it
(tests/golden/make_state_dict_keys.py).  Building from the same YAML values here must give exactly those keys, shapes
and dtypes, in the same order.  Constructing the modules needs no GPU."""
import json
import os

import pytest
import torch

from video_classification_amd import factory

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_keys.json")
with open(FIXTURE) as _f:
    CASES = json.load(_f)


@pytest.mark.parametrize("case", sorted(CASES))
def test_state_dict_contract(case):
    ref = CASES[case]
    model = factory.create_model(ref["model"], ref["config"])
    sd = model.state_dict()
    want = ref["keys"]
    assert sorted(sd.keys()) == sorted(want.keys()), (
        sorted(set(want) - set(sd))[:5], sorted(set(sd) - set(want))[:5])
    for k, v in sd.items():
        shape, dtype = want[k]
        assert list(v.shape) == shape, k
        assert str(v.dtype).replace("torch.", "") == dtype, k
    # a state_dict of that layout loads strictly, with and without the DataParallel prefix
    from video_classification_amd.trainer import strip_module_prefix
    blank = {k: torch.zeros(s, dtype=getattr(torch, d)) for k, (s, d) in want.items()}
    model.load_state_dict(strip_module_prefix({"module." + k: v for k, v in blank.items()}), strict=True)


def test_factory_accepts_config_loader_objects_and_rejects_unknown_models():
    class Loader:  # the reference's ConfigLoader keeps the YAML dictionary in .config (utils/config_loader.py:21)
        config = CASES["Conv3D"]["config"]
    assert type(factory.create_model("Conv3D", Loader())).__name__ == "Conv3DModel"
    with pytest.raises(ValueError, match="Unsupported model"):
        factory.create_model("VGG", {})
    with pytest.raises(NotImplementedError):
        factory.create_model("ResNetCRNN_varylength", {})
