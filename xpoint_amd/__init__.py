"""xpoint_amd — MI355X (gfx950) native XPoint inference hot path behind the reference's interfaces.

    from xpoint_amd import models, utils
    net = models.XPoint(cfg['model']); net.load_state_dict(weights); net.to('cuda').eval()
    out_optical, out_thermal, hm = net(data)
    prob = utils.box_nms(out_optical['prob'] * mask, 8, 0.015)

Compute is hand-written HIP behind a C ABI (include/xpoint_hip.h, libxpoint_hip.so); there is no
CPU fallback.  Build: `python -m xpoint_amd.build`.
"""
__version__ = "0.1.0"
