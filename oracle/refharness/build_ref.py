"""Build the REAL reference model (imported from /root/reference, never copied) on CPU with the
build's synthetic weights.  Only usable in the build container; test infrastructure only.
Recipe: SURVEY.md Appendix D."""
import copy
import os
import tempfile

import torch
import yaml

from . import stubs


def build_reference_xpoint(cfg: dict, state_dict=None, strict=True):
    """cfg: the `model:` dict (xpoint_amd.synth.xpoint_exp1_config).  Mirrors reference
    benchmark.py:50-127 (params.yaml override -> H/W patch -> XPoint(cfg) -> load_state_dict)."""
    stubs.install()
    import xpoint.models as ref_models  # noqa: the real reference package

    cfg = copy.deepcopy(cfg)
    vssm = cfg["use_attention"]["model_parameters"]["MODEL"]["VSSM"]
    # reference train.py:36-38: model_parameters is a dump of the (absent) vssm_tiny.yaml, so
    # re-synthesise that yaml from it.
    tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump({"MODEL": {"TYPE": "vssm", "NAME": "vssm_tiny_segmentation", "DROP_PATH_RATE": 0.2,
                              "VSSM": dict(vssm)},
                    "DATA": {"IMG_SIZE": 512}}, tmp)
    tmp.close()
    cfg["use_attention"]["pretrained"]["yaml_file"] = tmp.name
    try:
        net = ref_models.XPoint(cfg)
    finally:
        os.unlink(tmp.name)
    net.eval()
    if state_dict is not None:
        sd = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in state_dict.items()}
        missing, unexpected = net.load_state_dict(sd, strict=strict)
        assert not missing and not unexpected, (missing, unexpected)
    return net


def build_reference_superpoint(state_dict=None):
    stubs.install()
    import xpoint.models as ref_models
    net = ref_models.SuperPointMagicLeap({})
    net.eval()
    if state_dict is not None:
        net.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return net


def build_reference_conv_xpoint(cfg: dict, state_dict=None):
    """Conv-encoder XPoint (model_weights/multipoint/params.yaml): no VMamba yaml needed."""
    stubs.install()
    import xpoint.models as ref_models
    net = ref_models.XPoint(copy.deepcopy(cfg))
    net.eval()
    if state_dict is not None:
        missing, unexpected = net.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
        assert not missing and not unexpected
    return net
