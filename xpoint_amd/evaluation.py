"""Evaluation-harness metrics on the MI355X path — SURVEY.md §8(f) rank 1.

Mirrors the reference's `xpoint/utils/benchmark_evaluation.py` (same function names, arguments and return
structures, including its per-sample overwrite quirks, see below) and `xpoint/utils/homographies.py:479-526`
(`warp_keypoints`, `filter_points`):

  compute_repeatability_for_sample   benchmark_evaluation.py:396-467
  compute_descriptor_for_sample      benchmark_evaluation.py:588-751   (NN-mAP inputs, M-score)
  compute_mAP / compute_desc_dict    benchmark_evaluation.py:469-558
  compute_pts_dist_for_sample        benchmark_evaluation.py:755-830   (corner error of the estimated homography;
  compute_homography_dict            benchmark_evaluation.py:560-586    the estimator is utils.find_homography, the
                                                                        device stand-in for cv2.findHomography MAGSAC:
                                                                        same contract, not bit-comparable — §8(f) rank 2)
  compute_metrics                    benchmark_evaluation.py:832-963

Device work goes through the C ABI: keypoints and descriptors stay on the GPU, descriptor sampling is
`xp_sample_descriptors`, both match directions come from ONE `xp_match_mnn` call (mutual nearest neighbours are
the same pairs seen from either side), and the N x M "distance to the nearest keypoint" matrices of the reference
(`np.linalg.norm(warped[:, None] - kp[None])`, `torch.norm(dist.float(), dim=-1) <= th`) are never materialised:
`xp_points_min_dist` returns the per-row minimum, which is all the metrics use.  The per-match bookkeeping (tp lists,
sorting, precision / recall) is host numpy on a few thousand numbers, exactly as in the reference.

Reference behaviour kept on purpose (the goldens in tests/golden/g13 come from the reference's own code):
  * inside a batch the repeatability dict and the `n_gt_*` counts are OVERWRITTEN per sample, so they hold the last
    sample's value (benchmark_evaluation.py:447-461, 662-663) — which is why the reference's NN-mAP can exceed 1;
  * `warp_keypoints(..., return_type=int)` truncates toward zero (`astype(int)`);
  * the repeatability keypoints use `prob > thr` times the valid mask, the descriptor metrics use `prob > thr` of the
    already masked heat map.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, utils
from ._lib import ptr


def div0(a, b):
    """reference utils.div0: a / b with 0 where b == 0."""
    with np.errstate(divide='ignore', invalid='ignore'):
        c = np.true_divide(a, b)
        c[~np.isfinite(c)] = 0
    return c


def warp_keypoints(keypoints, homography, return_type=int):
    """homographies.py:479-497.  keypoints (N,2) as (y, x); homography 3x3 acting on (x, y, 1) — the projective map of
    cv2.perspectiveTransform evaluated in float64."""
    keypoints = np.asarray(keypoints)
    if len(keypoints) == 0:
        return keypoints
    h = np.asarray(homography, dtype=np.float64)
    xy = keypoints[:, ::-1].astype(np.float64)
    hom = np.concatenate([xy, np.ones((xy.shape[0], 1))], 1) @ h.T
    out = hom[:, :2] / hom[:, 2:3]
    return out[:, ::-1].astype(return_type)


def filter_points(points, shape):
    """homographies.py:511-526."""
    points = points[points[:, 0] >= 0]
    points = points[points[:, 1] >= 0]
    points = points[points[:, 0] < shape[0]]
    points = points[points[:, 1] < shape[1]]
    return points


def _min_dist(a_np, b_dev_f32):
    """min_j |a_i - b_j| on the device; a: host array (N,2) (int or float), b: device float32 (M,2).  Returns numpy f32."""
    na, nb = int(a_np.shape[0]), int(b_dev_f32.shape[0])
    if na == 0:
        return np.zeros((0,), np.float32)
    a = torch.from_numpy(np.ascontiguousarray(a_np, dtype=np.float64)).to(b_dev_f32.device)
    out = torch.empty((na,), dtype=torch.float32, device=b_dev_f32.device)
    b = b_dev_f32.contiguous()
    _lib.call("xp_points_min_dist", ctypes.c_void_p(a.data_ptr()), na, ctypes.c_void_p(b.data_ptr()) if nb else None, nb,
              ctypes.c_void_p(out.data_ptr()), _lib.current_stream())
    return out.cpu().numpy()


def compute_repeatability_for_sample(out_optical, out_thermal, data, H_optical, H_thermal, detection_threshold, distance_thresh):
    """benchmark_evaluation.py:396-467; same returns: ({th: [repeatability]}, n_kp_optical list, n_kp_thermal list)."""
    n_kp_optical_member, n_kp_thermal_member = [], []
    repeatability_member_dict = {}
    ths = distance_thresh if type(distance_thresh) is list else [distance_thresh]
    for (prob_o, prob_t, mask_o, mask_t, h_o, h_t) in zip(out_optical['prob'].split(1), out_thermal['prob'].split(1),
                                                          data['optical']['valid_mask'].split(1),
                                                          data['thermal']['valid_mask'].split(1),
                                                          H_optical.split(1), H_thermal.split(1)):
        kp_optical_d = torch.nonzero((prob_o.squeeze() > detection_threshold).float() * mask_o.squeeze().to(prob_o.device))
        kp_thermal_d = torch.nonzero((prob_t.squeeze() > detection_threshold).float() * mask_t.squeeze().to(prob_t.device))
        n_kp_optical_member.append(kp_optical_d.shape[0])
        n_kp_thermal_member.append(kp_thermal_d.shape[0])
        kp_optical, kp_thermal = kp_optical_d.cpu().numpy(), kp_thermal_d.cpu().numpy()
        image_shape = tuple(prob_o.squeeze().shape)
        h_o = h_o.squeeze().double().cpu(); h_t = h_t.squeeze().double().cpu()
        h_o32, h_t32 = h_o.float(), h_t.float()       # the reference inverts the float32 matrices (torch .inverse())
        warped_optical = warp_keypoints(kp_optical, h_o32.inverse().numpy())
        warped_optical = warp_keypoints(warped_optical, h_t32.numpy())
        warped_optical = filter_points(warped_optical, image_shape)
        warped_thermal = warp_keypoints(kp_thermal, h_t32.inverse().numpy())
        warped_thermal = warp_keypoints(warped_thermal, h_o32.numpy())
        warped_thermal = filter_points(warped_thermal, image_shape)
        N_thermal, N_optical = warped_thermal.shape[0], warped_optical.shape[0]
        # min over the other image's keypoints of the pixel distance (reference: all-pairs norm, then np.min(axis=1))
        d1 = _min_dist(warped_thermal, kp_optical_d.float())
        d2 = _min_dist(warped_optical, kp_thermal_d.float())
        for th in ths:
            repeatability_member = []
            count1 = int(np.sum(d1 <= th)) if kp_optical.shape[0] != 0 else 0
            count2 = int(np.sum(d2 <= th)) if kp_thermal.shape[0] != 0 else 0
            if N_thermal + N_optical > 0:
                repeatability_member.append((count1 + count2) / (N_thermal + N_optical))
            repeatability_member_dict[th] = repeatability_member
    return repeatability_member_dict, n_kp_optical_member, n_kp_thermal_member


def compute_mAP(precision, recall):
    """benchmark_evaluation.py:469-473."""
    return np.sum(precision[1:] * (recall[1:] - recall[:-1]))


def _mutual_matches(desc_o, desc_t):
    """Both reference calls get_matches(desc_o, desc_t) and get_matches(desc_t, desc_o) (bfmatcher, crossCheck) from one
    device call: lists of (queryIdx, trainIdx, distance) ordered by queryIdx, as BFMatcher.match returns them."""
    if desc_o.shape[0] == 0 or desc_t.shape[0] == 0:
        return [], []
    res = utils.match_descriptors(desc_o[None].contiguous(), desc_t[None].contiguous(), None, "strict_mnn")
    n = int(res["match_count"][0].item())
    q = res["match_q"][0, :n].cpu().numpy(); t = res["match_t"][0, :n].cpu().numpy(); d = res["match_d"][0, :n].cpu().numpy()
    m_o = [(int(a), int(b), float(c)) for a, b, c in zip(q, t, d)]                    # query = optical (ascending q)
    order = np.argsort(t, kind="stable")
    m_t = [(int(t[i]), int(q[i]), float(d[i])) for i in order]                        # query = thermal (ascending t)
    return m_o, m_t


def compute_descriptor_for_sample(prob_optical, prob_thermal, desc_optical, desc_thermal, data, config,
                                  keypoint_detection_threshold, threshold_keypoints):
    """benchmark_evaluation.py:588-751; same returns ({th: {tp_*, distance_*, m_score_*, matching_kp_numbers, n_gt_*}}).
    desc_* are the network's dense descriptor maps (B, D, Hc, Wc) on the device."""
    method = config['prediction']['matching']['method']
    if method != 'bfmatcher' or config['prediction']['matching'].get('knn_matches', False) or \
            not config['prediction']['matching'].get('method_kwargs', {}).get('crossCheck', False):
        raise NotImplementedError("xpoint_amd.evaluation: the device matcher implements bfmatcher + crossCheck (the reference's configured mode)")
    descriptor_dict = {th: {} for th in threshold_keypoints} if type(threshold_keypoints) is list else {threshold_keypoints: {}}
    H_o, W_o = data['optical']['image'].shape[2:]
    H_t, W_t = data['thermal']['image'].shape[2:]
    per_sample = []
    for (prob_o, prob_t, h_o, h_t, desc_o, desc_t) in zip(prob_optical, prob_thermal, data['optical']['homography'],
                                                          data['thermal']['homography'], desc_optical, desc_thermal):
        h_o, h_t = h_o.float().cpu(), h_t.float().cpu()
        gt_homography = torch.mm(h_t, h_o.inverse())
        pred_optical = torch.nonzero((prob_o.squeeze() > keypoint_detection_threshold).float())
        pred_thermal = torch.nonzero((prob_t.squeeze() > keypoint_detection_threshold).float())
        d_o = utils.interpolate_descriptors(pred_optical, desc_o, H_o, W_o)
        d_t = utils.interpolate_descriptors(pred_thermal, desc_t, H_t, W_t)
        matches_optical, matches_thermal = _mutual_matches(d_o, d_t)
        matches_optical = sorted(matches_optical, key=lambda x: x[2])
        matches_thermal = sorted(matches_thermal, key=lambda x: x[2])
        warped_optical = warp_keypoints(pred_optical.cpu().float().numpy(), gt_homography.numpy(), float)
        warped_thermal = warp_keypoints(pred_thermal.cpu().float().numpy(), gt_homography.inverse().numpy(), float)
        # rows of the reference's correct_* matrices that contain a True <=> the nearest keypoint is within th
        near_o = _min_dist(warped_optical, pred_thermal.float())
        near_t = _min_dist(warped_thermal, pred_optical.float())
        po, pt = pred_optical.cpu().numpy().astype(np.float64), pred_thermal.cpu().numpy().astype(np.float64)

        def pair_dist(w, other, matches):       # |warped[q] - other[t]| for the matched pairs, in the reference's f32 arithmetic
            if not matches:
                return np.zeros((0,), np.float32)
            qi = np.array([m[0] for m in matches]); ti = np.array([m[1] for m in matches])
            diff = (w[qi].astype(np.float64) - other[ti]).astype(np.float32)
            return np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
        image_shape = tuple(prob_o.squeeze().shape)
        per_sample.append(dict(mo=matches_optical, mt=matches_thermal, near_o=near_o, near_t=near_t,
                               pd_o=pair_dist(warped_optical, pt, matches_optical), pd_t=pair_dist(warped_thermal, po, matches_thermal),
                               N_optical=filter_points(warped_optical, image_shape).shape[0],
                               N_thermal=filter_points(warped_thermal, image_shape).shape[0]))
    for th in descriptor_dict.keys():
        tp_o, tp_t, dist_o, dist_t, ms_o, ms_t, mk = [], [], [], [], [], [], []
        n_gt_o = n_gt_t = 0
        for s in per_sample:
            n_gt_o = int(np.sum(s["near_o"] <= th))          # overwritten per sample, as in the reference
            n_gt_t = int(np.sum(s["near_t"] <= th))
            c_o = (s["pd_o"] <= th); c_t = (s["pd_t"] <= th)
            tp_o += [bool(v) for v in c_o]; dist_o += [m[2] for m in s["mo"]]
            tp_t += [bool(v) for v in c_t]; dist_t += [m[2] for m in s["mt"]]
            nm_o, nm_t = int(c_o.sum()), int(c_t.sum())
            ms_o.append(float(nm_o) / s["N_optical"] if s["N_optical"] > 0 else 0.0)
            ms_t.append(float(nm_t) / s["N_thermal"] if s["N_thermal"] > 0 else 0.0)
            mk.append((nm_o + nm_t) // 2)
        descriptor_dict[th] = dict(tp_optical=tp_o, tp_thermal=tp_t, distance_optical=dist_o, distance_thermal=dist_t,
                                   m_score_optical=ms_o, m_score_thermal=ms_t, matching_kp_numbers=mk,
                                   n_gt_optical=n_gt_o, n_gt_thermal=n_gt_t)
    return descriptor_dict


def compute_desc_dict(descriptor_metrics_dict):
    """benchmark_evaluation.py:476-558 (host numpy on the per-match lists)."""
    results = {}
    for th, d in descriptor_metrics_dict.items():
        out = {}
        maps = {}
        for spec in ("optical", "thermal"):
            tp = np.array(d[f'tp_{spec}']); dist = np.array(d[f'distance_{spec}'])
            idx = np.argsort(dist)
            tp = tp[idx]; fp = np.logical_not(tp); dist = dist[idx]
            tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
            recall = div0(tp_cum, d[f'n_gt_{spec}'])
            precision = div0(tp_cum, tp_cum + fp_cum)
            recall = np.concatenate([[0], recall, [1]])
            precision = np.concatenate([[0], precision, [0]])
            precision = np.maximum.accumulate(precision[::-1])[::-1]
            maps[spec] = compute_mAP(precision, recall)
            out.update({f'tp_{spec}': tp, f'fp_{spec}': fp, f'distance_{spec}': dist, f'recall_{spec}': recall,
                        f'precision_{spec}': precision, f'nn_map_{spec}': maps[spec], f'm_score_{spec}': np.array(d[f'm_score_{spec}'])})
        out['nn_map'] = (maps['optical'] + maps['thermal']) * 0.5
        out['m_score'] = (out['m_score_optical'].mean() + out['m_score_thermal'].mean()) * 0.5
        results[th] = out
    return results


def compute_pts_dist_for_sample(prob_optical, prob_thermal, desc_optical, desc_thermal, data, config, keypoint_detection_threshold,
                                ransac_reporjection_thresholds):
    """benchmark_evaluation.py:755-830: mean distance of the four image "corners" (the reference's own point list,
    [[0,0],[H,0],[0,W],[H,H]]) warped by the ground-truth and by the estimated optical->thermal homography; 999.0 when
    no model could be estimated.  Returns {threshold: [distance per sample]}."""
    ths = ransac_reporjection_thresholds if type(ransac_reporjection_thresholds) is list else [ransac_reporjection_thresholds]
    H_o, W_o = data['optical']['image'].shape[2:]
    H_t, W_t = data['thermal']['image'].shape[2:]
    samples = []
    for (prob_o, prob_t, h_o, h_t, desc_o, desc_t) in zip(prob_optical, prob_thermal, data['optical']['homography'],
                                                          data['thermal']['homography'], desc_optical, desc_thermal):
        h_o, h_t = h_o.float().cpu(), h_t.float().cpu()
        gt_homography = torch.mm(h_t, h_o.inverse())
        pred_optical = torch.nonzero((prob_o.squeeze() > keypoint_detection_threshold).float())
        pred_thermal = torch.nonzero((prob_t.squeeze() > keypoint_detection_threshold).float())
        d_o = utils.interpolate_descriptors(pred_optical, desc_o, H_o, W_o)
        d_t = utils.interpolate_descriptors(pred_thermal, desc_t, H_t, W_t)
        m_o, _ = _mutual_matches(d_o, d_t)
        if m_o:
            qi = torch.tensor([m[0] for m in m_o], device=pred_optical.device); ti = torch.tensor([m[1] for m in m_o], device=pred_optical.device)
            optical_pts = pred_optical[qi].flip(-1).float()       # cv2.KeyPoint(x = col, y = row)
            thermal_pts = pred_thermal[ti].flip(-1).float()
        else:
            optical_pts = thermal_pts = torch.zeros((0, 2), device=pred_optical.device)
        samples.append((gt_homography, optical_pts, thermal_pts))
    out = {}
    for th in ths:
        member = []
        for gt_homography, optical_pts, thermal_pts in samples:
            H_est, _ = utils.find_homography(optical_pts, thermal_pts, th) if optical_pts.shape[0] >= 4 else (None, None)
            if H_est is not None:
                pts = np.array([[0, 0], [H_o, 0], [0, W_o], [H_o, H_o]])
                gt = warp_keypoints(pts, gt_homography.numpy(), float)
                est = warp_keypoints(pts, H_est, float)
                member.append(np.linalg.norm(est - gt, axis=1).sum() / 4)
            else:
                member.append(999.0)
        out[th] = member
    return out


def compute_homography_dict(overall_pts_dist_dict, threshold_warp):
    """benchmark_evaluation.py:560-586."""
    results = {}
    for th_ransac, dists in overall_pts_dist_dict.items():
        pts_dist = np.array(dists)
        out = {'average_h_error': pts_dist.mean(), 'h_correctness': {}}
        for th_warp in threshold_warp:
            out['h_correctness']['epsilon_warp_th' + str(th_warp)] = (pts_dist < th_warp).sum() / len(pts_dist)
        results[th_ransac] = out
    return results


def compute_metrics(net, dataloader, device, config, keypoint_detection_threshold=0.015, thresh_repeatability=3, thresh_keypoints=2,
                    thresh_warp=2, ransac_reproj_thresholds=3):
    """benchmark_evaluation.py:832-963.  `dataloader` is any iterable of reference-style `data` dicts; returns
    {'repeatability': ..., 'descriptor': ..., 'homography': ...} (the last one from this build's estimator)."""
    repeatability = {th: [] for th in thresh_repeatability} if type(thresh_repeatability) is list else {thresh_repeatability: []}
    n_kp_optical, n_kp_thermal = [], []
    descriptor_metrics_dict = {th: {} for th in thresh_keypoints} if type(thresh_keypoints) is list else {thresh_keypoints: {}}
    rts = ransac_reproj_thresholds if type(ransac_reproj_thresholds) is list else [ransac_reproj_thresholds]
    overall_pts_dist_dict = {th: [] for th in rts}
    pred = config['prediction']
    for data in dataloader:
        B = data['optical']['image'].shape[0]
        for spec in ('optical', 'thermal'):
            if 'homography' not in data[spec]:
                data[spec]['homography'] = torch.eye(3, dtype=torch.float32).repeat(B, 1, 1)
        H_optical, H_thermal = data['optical']['homography'], data['thermal']['homography']
        data = utils.data_to_device(data, device)
        if not net.takes_pair():
            out_optical, out_thermal = net(data['optical']), net(data['thermal'])
        else:
            out_optical, out_thermal, _ = net(data)
        prob_optical = out_optical['prob'] * data['optical']['valid_mask']
        prob_thermal = out_thermal['prob'] * data['thermal']['valid_mask']
        if pred['nms'] > 0:
            kw = dict(keep_top_k=pred['topk'], on_cpu=pred.get('cpu_nms', False))
            prob_thermal = utils.box_nms(prob_thermal, pred['nms'], keypoint_detection_threshold, **kw)
            prob_optical = utils.box_nms(prob_optical, pred['nms'], keypoint_detection_threshold, **kw)
            out_optical['prob'] = utils.box_nms(out_optical['prob'], pred['nms'], keypoint_detection_threshold, **kw)
            out_thermal['prob'] = utils.box_nms(out_thermal['prob'], pred['nms'], keypoint_detection_threshold, **kw)
        rep, nko, nkt = compute_repeatability_for_sample(out_optical, out_thermal, data, H_optical, H_thermal,
                                                         keypoint_detection_threshold, thresh_repeatability)
        for key, value in rep.items():
            repeatability[key].extend(value)
        n_kp_optical += nko; n_kp_thermal += nkt
        dd = compute_descriptor_for_sample(prob_optical, prob_thermal, out_optical['desc'], out_thermal['desc'], data, config,
                                           keypoint_detection_threshold, thresh_keypoints)
        for key, value in dd.items():
            for key2, value2 in value.items():
                if key2.startswith("n_gt"):
                    descriptor_metrics_dict[key][key2] = descriptor_metrics_dict[key].get(key2, 0) + value2
                else:
                    descriptor_metrics_dict[key][key2] = descriptor_metrics_dict[key].get(key2, []) + value2
        pd = compute_pts_dist_for_sample(prob_optical, prob_thermal, out_optical['desc'], out_thermal['desc'], data, config,
                                         keypoint_detection_threshold, rts)
        for key, value in pd.items():
            overall_pts_dist_dict[key] += value
    out_rep = {'repeatability_mean': {k: np.mean(v) for k, v in repeatability.items()},
               'n_kp_optical': np.mean(n_kp_optical), 'n_kp_thermal': np.mean(n_kp_thermal)}
    out_rep['n_kp_avg'] = (out_rep['n_kp_optical'] + out_rep['n_kp_thermal']) / 2.0
    tw = thresh_warp if type(thresh_warp) is list else [thresh_warp]
    return {"repeatability": out_rep, "descriptor": compute_desc_dict(descriptor_metrics_dict),
            "homography": compute_homography_dict(overall_pts_dist_dict, tw)}


# ------------------------------------------------------------------------------------------------
# Timing harness: reference benchmark_evaluation.py:16-142 (`desc_process_and_display_sample`) — the callable benchmark.py:149-172
# drives to print "Two forward passes / Box nms / interpolate" rates.  Same signature, same loop, same `time_dict_seconds` keys.
# ------------------------------------------------------------------------------------------------
class _EventSpan:
    """One timed span on the current HIP stream: hipEvents around the enqueued work (the reference brackets with
    torch.cuda.synchronize() + time.time(), which on a GPU also counts the host's launch latency of an idle queue)."""

    def __init__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        self.e0.record()
        return self

    def __exit__(self, *exc):
        self.e1.record()
        return False

    def seconds(self):
        self.e1.synchronize()
        return self.e0.elapsed_time(self.e1) * 1e-3


def desc_process_and_display_sample(net, dataset, device, config, args):
    """reference benchmark_evaluation.py:16-227.  For every index in args.index: sample -> device -> forward (both spectra) ->
    box_nms(prob * valid_mask) -> per pair nonzero + interpolate_descriptors; returns
        time_dict_seconds = {"two_forward": [...], "nms": [...], "interpolate": [...]}      (one entry per sample / per pair)
    measured with HIP events on the stream the kernels run on.  With args.plot the reference goes on to match, estimate the homography
    (cv2 MAGSAC), mask the matches by the ground-truth homography and WRITE A PNG (cv2.drawMatches / imwrite, :129-218); cv2 is not
    part of this path, so the same quantities — keypoints, matches, H_est, the ground-truth matches mask — are computed on the device and
    written next to where the PNG would go, as `<same stem>.npz`."""
    import os
    time_dict_seconds = {"two_forward": [], "nms": [], "interpolate": []}
    pred = config['prediction']
    thr = pred['detection_threshold']
    for index in args.index:
        data = dataset[index]
        data = utils.data_to_device(data, device)
        data = utils.data_unsqueeze(data, 0)
        with _EventSpan() as t_fwd:
            if not net.takes_pair():
                out_optical = net(data['optical'])
                out_thermal = net(data['thermal'])
            else:
                out_optical, out_thermal, out_hm = net(data)
        time_dict_seconds["two_forward"].append(t_fwd.seconds())
        with _EventSpan() as t_nms:
            if pred['nms'] > 0:
                out_optical['prob'] = utils.box_nms(out_optical['prob'] * data['optical']['valid_mask'], pred['nms'], thr,
                                                    keep_top_k=pred['topk'], on_cpu=pred['cpu_nms'])
                out_thermal['prob'] = utils.box_nms(out_thermal['prob'] * data['thermal']['valid_mask'], pred['nms'], thr,
                                                    keep_top_k=pred['topk'], on_cpu=pred['cpu_nms'])
        time_dict_seconds["nms"].append(t_nms.seconds())
        n = data['optical']['image'].shape[0]
        for spec in ('optical', 'thermal'):                 # add homography to data if not available (reference :88-92)
            if 'homography' not in data[spec].keys():
                data[spec]['homography'] = torch.eye(3, dtype=torch.float32).to(device).view(1, 3, 3).repeat(n, 1, 1)
        H, W = data['optical']['image'].shape[2:]
        for i in range(n):
            prob_optical, prob_thermal = out_optical['prob'][i], out_thermal['prob'][i]
            pred_optical = torch.nonzero((prob_optical.squeeze() > thr).float())
            pred_thermal = torch.nonzero((prob_thermal.squeeze() > thr).float())
            with _EventSpan() as t_int:
                if out_optical['desc'][i].shape[1:] == prob_optical.shape[1:]:
                    # classic descriptors, directly take values (reference :117-120)
                    do, dt = out_optical['desc'][i], out_thermal['desc'][i]
                    desc_optical_sampled = do[:, pred_optical[:, 0], pred_optical[:, 1]].transpose(0, 1)
                    desc_thermal_sampled = dt[:, pred_thermal[:, 0], pred_thermal[:, 1]].transpose(0, 1)
                elif 'desc_nhwc' in out_optical:
                    desc_optical_sampled = utils.interpolate_descriptors_nhwc(pred_optical, out_optical['desc_nhwc'][i], H, W)
                    desc_thermal_sampled = utils.interpolate_descriptors_nhwc(pred_thermal, out_thermal['desc_nhwc'][i], H, W)
                else:
                    desc_optical_sampled = utils.interpolate_descriptors(pred_optical, out_optical['desc'][i], H, W)
                    desc_thermal_sampled = utils.interpolate_descriptors(pred_thermal, out_thermal['desc'][i], H, W)
            time_dict_seconds["interpolate"].append(t_int.seconds())
            if getattr(args, "plot", False):
                matches = utils.get_matches(desc_optical_sampled, desc_thermal_sampled, pred['matching']['method'], pred['matching']['knn_matches'],
                                            **pred['matching']['method_kwargs'])
                kpo, kpt = pred_optical.cpu().numpy().astype(np.float32), pred_thermal.cpu().numpy().astype(np.float32)
                optical_pts = np.float32([kpo[m.queryIdx][::-1] for m in matches]).reshape(-1, 1, 2)       # cv2.KeyPoint(c[1], c[0]).pt = (x, y)
                thermal_pts = np.float32([kpt[m.trainIdx][::-1] for m in matches]).reshape(-1, 1, 2)
                if optical_pts.shape[0] < 4 or thermal_pts.shape[0] < 4:
                    H_est = np.eye(3, 3)
                else:
                    H_est, _ = utils.find_homography(optical_pts, thermal_pts, float(pred['reprojection_threshold']), 10000)
                # correct matches mask (reference :189-193): matches whose ground-truth reprojection error is below the threshold
                H_gt = np.matmul(data['thermal']['homography'][i].cpu().numpy(), np.linalg.inv(data['optical']['homography'][i].cpu().numpy()))
                if len(matches):
                    warped_optical = warp_keypoints(optical_pts.squeeze(1)[:, ::-1], H_gt)[:, ::-1]
                    diff = np.linalg.norm(thermal_pts.squeeze(1) - warped_optical, axis=1)
                    matchesMask = (diff < pred['reprojection_threshold'])
                else:
                    matchesMask = np.zeros((0,), bool)
                save_dir = os.path.join(args.output_dir, 'images', 'supp', 'i' + str(index))
                os.makedirs(save_dir, exist_ok=True)
                stem = "{}_{}_i{}_s{}".format(os.path.join(save_dir, args.model_dir.split("/")[-1]), args.version, index, args.seed)
                np.savez(stem + ".npz", kp_optical=kpo, kp_thermal=kpt, matches=np.array([[m.queryIdx, m.trainIdx] for m in matches], np.int32).reshape(-1, 2),
                         H_est=np.eye(3) if H_est is None else H_est, matches_mask=matchesMask)
    return time_dict_seconds
