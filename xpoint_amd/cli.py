"""Thin command-line front ends of the two prediction flows, with the reference scripts' arguments
(predict_align_image_pair.py:24-37, predict_keypoints.py: the same -y / -m / -v / -i / -s options):

    python -m xpoint_amd.cli align     -y configs/cipdp.yaml -m model_weights/XPoint-EXP1 -v latest [-i 0] [-n 1] [-s 0] [-e] [-o out.npz]
    python -m xpoint_amd.cli keypoints -y configs/cipdp.yaml -m model_weights/XPoint-EXP1 -v latest [-i 0] [-n 1] [-s 0] [-o out.npz]

What is kept from the scripts: the YAML handling (model params from <model-dir>/params.yaml overwrite config['model'], the
`use_attention` height / width patch of predict_align_image_pair.py:50-54), `<model-dir>/<version>.model` loaded with
`strict=False` after `fix_model_weigth_keys`, the seeds, the per-sample flow.  What is not: plotting (-p, -r) and the HDF5
datasets — the dataset must be a folder dataset (`dataset.foldername`, xpoint_amd/datasets.py).  `-e` adds the registration
step (robust homography per pair) and prints the inlier counts.  Prints one line per sample and a timing summary."""
from __future__ import annotations

import argparse
import os
import random
import time

import numpy as np
import torch
import yaml


def load_config(yaml_config: str, model_dir: str) -> dict:
    with open(yaml_config, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    with open(os.path.join(model_dir, 'params.yaml'), 'r') as f:
        config['model'] = yaml.load(f, Loader=yaml.FullLoader)['model']          # overwrite the model params
    ua = config['model'].get('use_attention')
    if ua and ua.get('check'):                                                    # predict_align_image_pair.py:50-54
        ph, pw = ua['height'], ua['width']
        ua.setdefault('model_parameters', {}).setdefault('DATA', {})['IMG_SIZE'] = (ph, pw)
        ua['height'] = config['dataset']['height']
        ua['width'] = config['dataset']['width']
    return config


def build(config: dict, model_dir: str, version: str, device: str):
    from . import datasets, models, utils
    ds_cfg = dict(config['dataset'])
    ds_cfg.pop('type', None)
    ds_cfg['single_image'] = False
    dataset = datasets.ImagePairDataset(ds_cfg)
    net = getattr(models, config['model']['type'])(config['model'])
    if version != 'none':
        weights = torch.load(os.path.join(model_dir, version + '.model'), map_location=torch.device('cpu'))
        weights = utils.fix_model_weigth_keys(weights)
        missing, unexpected = net.load_state_dict(weights, strict=False)
        loaded = set(weights.keys()) - set(missing)
        print(f"Successfully loaded {len(loaded)} keys.\nMissing keys: {len(missing)}\nUnexpected keys: {len(unexpected)}")
        if len(loaded) < 1:
            raise ValueError("No weights were loaded correctly! Please check the model and weights file.")
    net.to(device)
    net.eval()
    return dataset, net


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m xpoint_amd.cli", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('flow', choices=['align', 'keypoints'])
    ap.add_argument('-y', '--yaml-config', default='configs/cipdp.yaml', help='YAML config file')
    ap.add_argument('-m', '--model-dir', default='model_weights/xpoint', help='Directory of the model')
    ap.add_argument('-v', '--version', default='latest', help='Model version (name of the param file), none for no weights')
    ap.add_argument('-i', '--index', default=0, type=int, help='Index of the first sample')
    ap.add_argument('-n', '--count', default=1, type=int, help='Number of consecutive samples')
    ap.add_argument('-e', dest='evaluation', action='store_true', help='align: also estimate the homography of every pair and warp the optical image with it')
    ap.add_argument('--save-warped', default=None, metavar='DIR', help='align -e: write the warped optical image of every sample to DIR/<name>_warped.png')
    ap.add_argument('-s', '--seed', default=0, type=int, help='Seed of the random generators')
    ap.add_argument('-o', '--output', default=None, help='write keypoints / matches of the samples to this .npz')
    ap.add_argument('--device', default='cuda:0')
    args = ap.parse_args(argv)

    random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)
    config = load_config(args.yaml_config, args.model_dir)
    if not torch.cuda.is_available():
        raise SystemExit("xpoint_amd runs on the GPU only (no CPU fallback)")
    print('Predicting on device: {}'.format(args.device))
    dataset, net = build(config, args.model_dir, args.version, args.device)
    from . import predict
    pred_cfg = {k: v for k, v in config.get('prediction', {}).items() if k in predict.DEFAULT_PREDICTION or k == 'reprojection_threshold'}
    out = {}
    t_total = 0.0
    with torch.no_grad():
        for idx in range(args.index, min(args.index + args.count, len(dataset))):
            data = dataset.load_batch([idx], args.device)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if args.flow == 'align':
                _, _, res = predict.predict_align_image_pair(net, data, pred_cfg, estimate_homography=args.evaluation)
            else:
                ko, kt = predict.predict_keypoints(net, data, pred_cfg)
                res = [dict(kp_optical=ko[0], kp_thermal=kt[0])]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            t_total += dt
            r = res[0]
            name = data.get('name', [str(idx)])[0]
            line = f"{name}: {len(r['kp_optical'])} / {len(r['kp_thermal'])} keypoints"
            if 'matches' in r:
                line += f", {len(r['matches'])} matches"
            if 'matchesMask' in r:
                line += f", {int(np.sum(r['matchesMask']))} inliers"
            print(line + f"  ({dt * 1e3:.1f} ms)")
            out[f"{idx}/kp_optical"] = r['kp_optical'].cpu().numpy(); out[f"{idx}/kp_thermal"] = r['kp_thermal'].cpu().numpy()
            if 'matches' in r:
                out[f"{idx}/matches"] = np.array([(m.queryIdx, m.trainIdx) for m in r['matches']], dtype=np.int64).reshape(-1, 2)
            if 'H_est' in r:
                out[f"{idx}/H_est"] = np.asarray(r['H_est'])
            if 'warped_optical' in r:
                out[f"{idx}/warped_optical"] = r['warped_optical'].cpu().numpy()
                if args.save_warped:
                    import os
                    from PIL import Image
                    os.makedirs(args.save_warped, exist_ok=True)
                    Image.fromarray(out[f"{idx}/warped_optical"]).save(os.path.join(args.save_warped, f"{name}_warped.png"))
    print(f"{args.flow}: {sum(1 for k in out if k.endswith('/kp_optical'))} sample(s), {t_total * 1e3:.1f} ms")
    if args.output:
        np.savez(args.output, **out)
    return out


if __name__ == "__main__":
    main()
