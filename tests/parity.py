"""Near-tie attribution for end-to-end index parity (VERDICT r1 "weak" 1).

The forward agrees with the reference to ~1e-5, not bit for bit, so a keypoint or match list can differ from the
reference's where — and only where — a decision sat inside that budget.  These helpers take a difference and either
EXPLAIN every element of it by a measured near-tie (score within `tol` of the detection threshold, of a competing
overlapping candidate, or a descriptor-distance gap below `tol`), following NMS cascades back to such a root, or
return it as unexplained.  Tests fail on any unexplained element and print the report.

Test infrastructure only (numpy, host side); nothing in xpoint_amd imports this."""
import numpy as np


def _overlaps(dy, dx, size):
    """box_nms' IoU > 0.1 test for two size x size boxes at integer offsets, as the integer predicate of SURVEY.md a12
    (reference utils/utils.py:148-192 + torchvision nms): inter / (2 size^2 - inter) > 0.1."""
    ay, ax = abs(int(dy)), abs(int(dx))
    if ay >= size or ax >= size:
        return False
    inter = (size - ay) * (size - ax)
    return inter / (2.0 * size * size - inter) > 0.1


def explain_keypoint_diff(kp_mine, kp_ref, prob_mine, thr, size=8, tol=1e-4, topk_cut=None):
    """kp_* (N,2) integer (y,x); prob_mine (H,W) float: the map MY keypoints came from (before NMS).
    topk_cut: with keep_top_k, the score of the last survivor kept (a survivor whose score is within tol of it may fall on either side of
    the cut).  Returns (report, unexplained): report = list of dicts for every keypoint in the symmetric difference."""
    mine = {tuple(int(v) for v in p) for p in np.asarray(kp_mine).reshape(-1, 2)}
    ref = {tuple(int(v) for v in p) for p in np.asarray(kp_ref).reshape(-1, 2)}
    diff = sorted(mine ^ ref)
    if not diff:
        return [], []
    H, W = prob_mine.shape
    P = np.asarray(prob_mine, dtype=np.float64)
    info = {}
    for (y, x) in diff:
        s = P[y, x]
        why = None
        if abs(s - thr) <= tol:
            why = f"score {s:.7f} within {tol:g} of the threshold {thr}"
        elif topk_cut is not None and abs(s - topk_cut) <= 2 * tol:
            why = f"score {s:.7f} within {2 * tol:g} of the top-k cut score {topk_cut:.7f}"
        else:
            # a competing overlapping candidate whose score is within 2 tol (each side carries up to tol), or whose own
            # candidacy is within tol of the threshold
            best = None
            for yy in range(max(0, y - size + 1), min(H, y + size)):
                for xx in range(max(0, x - size + 1), min(W, x + size)):
                    if (yy, xx) == (y, x) or not _overlaps(yy - y, xx - x, size):
                        continue
                    c = P[yy, xx]
                    if c <= thr - tol:
                        continue
                    if abs(c - s) <= 2 * tol or abs(c - thr) <= tol:
                        if best is None or abs(c - s) < abs(best[2] - s):
                            best = (yy, xx, c)
            if best is not None:
                why = (f"score {s:.7f} vs overlapping candidate ({best[0]},{best[1]}) score {best[2]:.7f} "
                       f"(|diff| {abs(best[2] - s):.2e})")
        info[(y, x)] = why
    # cascades: a flipped decision changes who suppresses whom next to it
    changed = True
    while changed:
        changed = False
        for p in diff:
            if info[p] is not None:
                continue
            for q in diff:
                if q != p and info[q] is not None and _overlaps(q[0] - p[0], q[1] - p[1], size):
                    info[p] = f"cascade of ({q[0]},{q[1]}) [{info[q].split(' [')[0][:60]}]"
                    changed = True
                    break
    report = [dict(kp=p, side="mine only" if p in mine else "reference only", score=float(P[p]), why=info[p]) for p in diff]
    return report, [r for r in report if r["why"] is None]


def explain_match_diff(kp_o_mine, kp_t_mine, kp_o_ref, kp_t_ref, matches_mine, matches_ref, desc_of, tol=1e-4):
    """matches_* (M,2) index pairs into the respective keypoint lists; desc_of(side, points (n,2)) -> (n,D) float array:
    MY descriptor for arbitrary (y,x) points of side 'optical' / 'thermal'.
    A pair is in a mutual-NN list iff each is the other's nearest neighbour, so a pair present on one side only is explained
    by (a) one of its keypoints existing on one side only, (b) its row / column nearest neighbour being such a keypoint, or
    (c) a best-vs-second distance gap below tol in its row or column.  Returns (report, unexplained)."""
    def as_pairs(kpo, kpt, m):
        kpo = np.asarray(kpo).reshape(-1, 2); kpt = np.asarray(kpt).reshape(-1, 2); m = np.asarray(m).reshape(-1, 2)
        return {(tuple(int(v) for v in kpo[q]), tuple(int(v) for v in kpt[t])) for q, t in m}
    Mm = as_pairs(kp_o_mine, kp_t_mine, matches_mine)
    Mr = as_pairs(kp_o_ref, kp_t_ref, matches_ref)
    diff = sorted(Mm ^ Mr)
    if not diff:
        return [], []
    sets = {}
    for side, a, b in (("optical", kp_o_mine, kp_o_ref), ("thermal", kp_t_mine, kp_t_ref)):
        A = {tuple(int(v) for v in p) for p in np.asarray(a).reshape(-1, 2)}
        B = {tuple(int(v) for v in p) for p in np.asarray(b).reshape(-1, 2)}
        U = sorted(A | B)
        pts = np.array(U, dtype=np.int64).reshape(-1, 2)
        sets[side] = dict(U=U, index={p: i for i, p in enumerate(U)}, in_m=np.array([p in A for p in U]), in_r=np.array([p in B for p in U]),
                          D=np.asarray(desc_of(side, pts), dtype=np.float64))

    def line(src_side, p, dst_side):
        s, d = sets[src_side], sets[dst_side]
        v = s["D"][s["index"][p]]
        dist = np.sqrt(((d["D"] - v) ** 2).sum(1))
        out = []
        arg = {}
        for tag, mask in (("mine", d["in_m"]), ("reference", d["in_r"])):
            dd = np.where(mask, dist, np.inf)
            o = np.argsort(dd, kind="stable")[:2]
            arg[tag] = int(o[0])
            gap = dd[o[1]] - dd[o[0]] if len(o) > 1 else np.inf
            if gap < tol:
                out.append(f"{dst_side} NN gap {gap:.2e} ({tag} keypoint set)")
        if arg["mine"] != arg["reference"]:
            out.append(f"nearest {dst_side} keypoint differs between the keypoint sets ({d['U'][arg['mine']]} vs {d['U'][arg['reference']]})")
        return out
    report = []
    for (q, t) in diff:
        why = []
        so, st = sets["optical"], sets["thermal"]
        iq, it = so["index"][q], st["index"][t]
        if not (so["in_m"][iq] and so["in_r"][iq]):
            why.append(f"optical keypoint {q} exists on one side only")
        if not (st["in_m"][it] and st["in_r"][it]):
            why.append(f"thermal keypoint {t} exists on one side only")
        why += line("optical", q, "thermal") + line("thermal", t, "optical")
        report.append(dict(pair=(q, t), side="mine only" if (q, t) in Mm else "reference only", why="; ".join(why) if why else None))
    return report, [r for r in report if r["why"] is None]


def format_report(title, report):
    lines = [f"{title}: {len(report)} differing element(s)"]
    for r in report:
        key = r.get("kp", r.get("pair"))
        lines.append(f"  {key} [{r['side']}] <- {r['why'] if r['why'] else 'UNEXPLAINED'}")
    return "\n".join(lines)
