"""KITTI / SlopedKITTI detection evaluation with the per-pair and per-(frame, threshold) work on the GPU
(SURVEY.md §8 f3).

Mirror of core/pcdet/datasets/kitti/kitti_object_eval_python/eval.py (get_thresholds, clean_data,
eval_class, get_mAP, get_mAP_R40, do_eval, get_official_eval_result) and of the slopedkitti copy
(difficulty 3 = "all", metric 3 = centre matching score, ATS / ASS / AOS / ODS, do_eval_slopedkitti,
get_slopedkitti_eval_result): same arguments, same tables, same report text.

What changed underneath (see csrc/kitti_eval.hip): the reference runs only the rotated IoU on the GPU,
over "parts" of 50-100 frames whose cross-frame pairs are computed and discarded, and does the greedy
matching on the host under numba.jit, frame by frame and threshold by threshold.  Here one launch
computes exactly the per-frame overlap blocks of the whole split, one launch matches every frame (pass A)
and one launch matches every (frame, score threshold) pair (pass B); the host keeps only the string
handling (`clean_data`) and the 41-point precision / recall bookkeeping.

`backend` (tests inject a CPU-oracle one) must provide overlaps / pass_a / pass_b like
de6d_amd.ops.kitti_eval.DeviceEvalBackend; the default is that class and needs the GPU.
"""
import io as sysio

import numpy as np

CLASS_NAMES = ['car', 'pedestrian', 'cyclist', 'van', 'person_sitting', 'truck']
CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting', 5: 'Truck'}
N_SAMPLE_PTS = 41
# difficulty tables of clean_data; the fourth column is SlopedKITTI's "all" level
MIN_HEIGHT = [40, 25, 25, -1]
MAX_OCCLUSION = [0, 1, 2, 10000]
MAX_TRUNCATION = [0.15, 0.3, 0.5, 10000]


def get_thresholds(scores, num_gt, num_sample_pts=41):
    """score thresholds at which recall crosses the num_sample_pts sampling positions (eval.py:9-26)"""
    ordered = np.sort(np.asarray(scores))[::-1]
    picked, current_recall, last = [], 0.0, len(ordered) - 1
    for i, score in enumerate(ordered):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < last else l_recall
        if i < last and (r_recall - current_recall) < (current_recall - l_recall):
            continue
        picked.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return picked


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """which ground truths / detections count for (class, difficulty): 0 evaluate, 1 ignore, -1 other class;
    DontCare boxes (eval.py:29-75), vectorised"""
    cls = CLASS_NAMES[current_class]
    gt_names = np.char.lower(np.asarray(gt_anno['name'], dtype=str)) if len(gt_anno['name']) else np.zeros(0, dtype=str)
    gt_bbox = np.asarray(gt_anno['bbox']).reshape(-1, 4)
    same = gt_names == cls
    neighbour = ((cls == 'pedestrian') & (gt_names == 'person_sitting')) | ((cls == 'car') & (gt_names == 'van'))
    hard = (np.asarray(gt_anno['occluded']) > MAX_OCCLUSION[difficulty]) | \
           (np.asarray(gt_anno['truncated']) > MAX_TRUNCATION[difficulty]) | \
           ((gt_bbox[:, 3] - gt_bbox[:, 1]) <= MIN_HEIGHT[difficulty])
    ignored_gt = np.full(len(gt_names), -1, np.int32)
    ignored_gt[neighbour | (same & hard)] = 1
    ignored_gt[same & ~hard] = 0
    dc_bboxes = gt_bbox[np.asarray(gt_anno['name'], dtype=str) == 'DontCare'] if len(gt_names) else np.zeros((0, 4))

    dt_names = np.char.lower(np.asarray(dt_anno['name'], dtype=str)) if len(dt_anno['name']) else np.zeros(0, dtype=str)
    dt_bbox = np.asarray(dt_anno['bbox']).reshape(-1, 4)
    ignored_dt = np.where(dt_names == cls, 0, -1).astype(np.int32)
    ignored_dt[np.abs(dt_bbox[:, 3] - dt_bbox[:, 1]) < MIN_HEIGHT[difficulty]] = 1
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc_bboxes


def _cat(annos, key, width):
    """concatenate a per-frame field to (total, width) float64 (exact for float32 input); empty frames allowed"""
    parts = [np.asarray(a[key], np.float64).reshape(-1, width) for a in annos if len(a['name'])]
    return np.concatenate(parts, 0) if parts else np.zeros((0, width))


def _metric_boxes(annos, metric):
    if metric == 0:
        return _cat(annos, 'bbox', 4)
    loc, dims, ry = _cat(annos, 'location', 3), _cat(annos, 'dimensions', 3), _cat(annos, 'rotation_y', 1)
    if metric == 1:
        return np.concatenate([loc[:, [0, 2]], dims[:, [0, 2]], ry], 1)
    if metric == 2:
        return np.concatenate([loc, dims, ry], 1)
    return np.concatenate([loc, dims, ry, _cat(annos, 'pitch', 1), _cat(annos, 'roll', 1)], 1)


class SplitLayout(object):
    """the ragged layout of a split: per-frame offsets into the concatenated detections / ground truths and
    into the per-frame (n_dt x n_gt) overlap blocks, plus the box tables of every metric"""

    def __init__(self, gt_annos, dt_annos, metrics=(0, 1, 2)):
        assert len(gt_annos) == len(dt_annos)
        self.n_frames = len(gt_annos)
        n_gt = np.array([len(a['name']) for a in gt_annos], np.int64)
        n_dt = np.array([len(a['name']) for a in dt_annos], np.int64)
        self.gt_off = np.concatenate([[0], np.cumsum(n_gt)]).astype(np.int32)
        self.dt_off = np.concatenate([[0], np.cumsum(n_dt)]).astype(np.int32)
        self.pair_off = np.concatenate([[0], np.cumsum(n_gt * n_dt)]).astype(np.int64)
        # detections written by generate_prediction_dicts are float32; numba / NumPy type the arithmetic on them as such
        self.dt_f32 = any(np.asarray(a['bbox']).dtype == np.float32 for a in dt_annos if len(a['name']))
        self.gt_boxes = {m: _metric_boxes(gt_annos, m) for m in metrics}
        self.dt_boxes = {m: _metric_boxes(dt_annos, m) for m in metrics}
        self.gt_alpha, self.dt_alpha = _cat(gt_annos, 'alpha', 1).reshape(-1), _cat(dt_annos, 'alpha', 1).reshape(-1)
        self.dt_score = _cat(dt_annos, 'score', 1).reshape(-1)


def _default_backend(layout):
    from .....ops.kitti_eval import DeviceEvalBackend
    return DeviceEvalBackend(layout)


def _running_max_from_right(table, n):
    for i in range(n):
        table[i] = np.max(table[i:], axis=-1)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=100,
               backend=None, layout=None):
    """precision / recall / orientation tables [class, difficulty, min_overlap, 41] of one metric
    (eval.py:448-553; metric 3 additionally returns the SlopedKITTI true-positive error sums)"""
    layout = layout or SplitLayout(gt_annos, dt_annos, metrics=(metric,))
    backend = backend or _default_backend(layout)
    shape = [len(current_classes), len(difficultys), len(min_overlaps)]
    precision, recall, aos = (np.zeros(shape + [N_SAMPLE_PTS]) for _ in range(3))
    ate, ase, num_tp = (np.zeros(shape) for _ in range(3))
    aoe = np.zeros(shape + [3])
    with np.errstate(invalid='ignore', divide='ignore'):
        for m, current_class in enumerate(current_classes):
            for l, difficulty in enumerate(difficultys):
                cleaned = [clean_data(g, d, current_class, difficulty) for g, d in zip(gt_annos, dt_annos)]
                total_num_valid_gt = sum(c[0] for c in cleaned)
                ignored_gt = np.concatenate([c[1] for c in cleaned]) if cleaned else np.zeros(0, np.int32)
                ignored_dt = np.concatenate([c[2] for c in cleaned]) if cleaned else np.zeros(0, np.int32)
                dc_off = np.concatenate([[0], np.cumsum([len(c[3]) for c in cleaned])]).astype(np.int32)
                dc_bbox = np.concatenate([np.asarray(c[3], np.float64).reshape(-1, 4) for c in cleaned]) if cleaned else np.zeros((0, 4))
                for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                    tp_scores, tp_count, gt_of_tp = backend.pass_a(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap,
                                                                   want_gt_of_tp=metric == 3)
                    matched = np.concatenate([tp_scores[layout.gt_off[f]:layout.gt_off[f] + tp_count[f]]
                                              for f in range(layout.n_frames)]) if layout.n_frames else np.zeros(0)
                    thresholds = np.array(get_thresholds(matched, total_num_valid_gt))
                    pr = backend.pass_b(metric, ignored_gt, ignored_dt, dc_off, dc_bbox, min_overlap, thresholds, compute_aos)
                    n = len(thresholds)
                    recall[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    _running_max_from_right(precision[m, l, k], n)
                    _running_max_from_right(recall[m, l, k], n)
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                        _running_max_from_right(aos[m, l, k], n)
                    if metric == 3:
                        errs = true_positive_errors(gt_annos, dt_annos, layout, gt_of_tp)
                        ate[m, l, k], ase[m, l, k], aoe[m, l, k], num_tp[m, l, k] = errs
    ret = {'recall': recall, 'precision': precision, 'orientation': aos}
    if metric == 3:
        ret.update(ate=ate, ase=ase, aoe=aoe, num_tp=num_tp)
    return ret


def true_positive_errors(gt_annos, dt_annos, layout, gt_of_tp):
    """translation / scale / orientation error sums over the matched pairs (slopedkitti eval.py:613-646)"""
    ate = ase = 0.0
    aoe, n_tp = np.zeros(3), 0

    def angles(anno, idx):
        return np.stack([np.asarray(anno[key])[idx] for key in ('rotation_y', 'pitch', 'roll')], -1) % (np.pi * 2)

    for f, (g, d) in enumerate(zip(gt_annos, dt_annos)):
        gt_idx = gt_of_tp[layout.dt_off[f]:layout.dt_off[f + 1]]
        hit = gt_idx > -1
        if hit.sum() == 0:
            continue
        sel = gt_idx[hit]
        dim_gt, dim_dt = np.asarray(g['dimensions'])[sel, :], np.asarray(d['dimensions'])[hit, :]
        ate += np.linalg.norm(np.asarray(g['location'])[sel, :] - np.asarray(d['location'])[hit, :], axis=-1).sum()
        inter = np.min(np.array([dim_gt, dim_dt]), axis=0).prod(axis=1)
        ase += (1 - inter / (dim_dt.prod(axis=1) + dim_gt.prod(axis=1) - inter)).sum()
        gap = np.abs(angles(d, hit) - angles(g, sel))
        gap[gap > np.pi] = 2 * np.pi - gap[gap > np.pi]
        aoe += gap.sum(axis=0)
        n_tp += hit.sum()
    return ate, ase, aoe, n_tp


def get_mAP(prec):
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100


def get_mAP_R40(prec):
    return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100


def get_tp_score(ate, ase, aoe, num_tp):
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.array([np.clip(1 - err / num_tp, a_min=0, a_max=1) for err in (ate, ase, aoe.sum(axis=-1))])


def get_ods(mAP, tp_score_list):
    return mAP / 100 / 2.0 + (tp_score_list / (tp_score_list.shape[0] * 2)).sum(axis=0)


def _line(text):
    buf = sysio.StringIO()
    print(text, file=buf)
    return buf.getvalue()


def _class_ids(current_classes):
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    name_to_class = {v: k for k, v in CLASS_TO_NAME.items()}
    return [name_to_class[c] if isinstance(c, str) else c for c in current_classes]


def _alpha_is_valid(dt_annos):
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            return bool(anno['alpha'][0] != -10)
    return False


def _eval_metrics(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, difficultys, metrics, PR_detail_dict, backend):
    layout = SplitLayout(gt_annos, dt_annos, metrics=metrics)
    backend = backend or _default_backend(layout)
    out = {}
    for metric, key in zip(metrics, ('bbox', 'bev', '3d', '3dctr')):
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps,
                         compute_aos and metric == 0, backend=backend, layout=layout)
        out[key] = ret
        if PR_detail_dict is not None:
            PR_detail_dict[key] = ret['precision']
            if metric == 0 and compute_aos:
                PR_detail_dict['aos'] = ret['orientation']
    return out


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, PR_detail_dict=None, backend=None):
    r = _eval_metrics(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, [0, 1, 2], (0, 1, 2), PR_detail_dict, backend)
    aos = (get_mAP(r['bbox']['orientation']), get_mAP_R40(r['bbox']['orientation'])) if compute_aos else (None, None)
    p = {k: r[k]['precision'] for k in ('bbox', 'bev', '3d')}
    return (get_mAP(p['bbox']), get_mAP(p['bev']), get_mAP(p['3d']), aos[0],
            get_mAP_R40(p['bbox']), get_mAP_R40(p['bev']), get_mAP_R40(p['3d']), aos[1])


def do_eval_slopedkitti(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, PR_detail_dict=None, backend=None):
    r = _eval_metrics(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, [0, 1, 2, 3], (0, 1, 2, 3), PR_detail_dict, backend)
    aos = (get_mAP(r['bbox']['orientation']), get_mAP_R40(r['bbox']['orientation'])) if compute_aos else (None, None)
    p = {k: r[k]['precision'] for k in ('bbox', 'bev', '3d', '3dctr')}
    ctr = r['3dctr']
    tp_score_list = get_tp_score(ctr['ate'], ctr['ase'], ctr['aoe'], ctr['num_tp'])
    map_ctr, map_ctr_r40 = get_mAP(p['3dctr']), get_mAP_R40(p['3dctr'])
    return (get_mAP(p['bbox']), get_mAP(p['bev']), get_mAP(p['3d']), aos[0], map_ctr, get_ods(map_ctr, tp_score_list),
            get_mAP_R40(p['bbox']), get_mAP_R40(p['bev']), get_mAP_R40(p['3d']), aos[1], map_ctr_r40,
            get_ods(map_ctr_r40, tp_score_list), tp_score_list)


def _triple(label, table, j, i, digits=4):
    return _line('%s%s' % (label, ', '.join('%.*f' % (digits, table[j, d, i]) for d in range(3))))


def _r40_entries(ret_dict, name, tables, j):
    for key, table in tables:
        for level, tag in enumerate(('easy', 'moderate', 'hard')):
            ret_dict['%s_%s/%s_R40' % (name, key, tag)] = table[j, level, 0]


def get_official_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None, backend=None):
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3)
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])
    current_classes = _class_ids(current_classes)
    min_overlaps = np.stack([overlap_0_7, overlap_0_5], axis=0)[:, :, current_classes]   # [overlap set, metric, class]
    compute_aos = _alpha_is_valid(dt_annos)
    bbox, bev, d3, aos, bbox40, bev40, d340, aos40 = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos,
                                                              PR_detail_dict=PR_detail_dict, backend=backend)
    result, ret_dict = '', {}
    for j, curcls in enumerate(current_classes):
        name = CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            for head, (t_bbox, t_bev, t_3d, t_aos) in (('AP', (bbox, bev, d3, aos)), ('AP_R40', (bbox40, bev40, d340, aos40))):
                result += _line('%s %s@%s:' % (name, head, ', '.join('%.2f' % v for v in min_overlaps[i, :, j])))
                result += _triple('bbox AP:', t_bbox, j, i) + _triple('bev  AP:', t_bev, j, i) + _triple('3d   AP:', t_3d, j, i)
                if compute_aos:
                    result += _triple('aos  AP:', t_aos, j, i, digits=2)
            if i == 0:
                if compute_aos:
                    _r40_entries(ret_dict, name, (('aos', aos40),), j)
                _r40_entries(ret_dict, name, (('3d', d340), ('bev', bev40), ('image', bbox40)), j)
    return result, ret_dict


def get_slopedkitti_eval_result(gt_annos, dt_annos, current_classes, PR_detail_dict=None, backend=None):
    overlap_0_7 = np.array([[0.70, 0.50, 0.50, 0.70, 0.50, 0.70]] * 3 + [[0.53] * 6])   # image / bev / 3d IoU, centre score 2 - 2 sigmoid(1)
    overlap_0_5 = np.array([[0.70, 0.50, 0.50, 0.70, 0.50, 0.50], [0.50, 0.25, 0.25, 0.50, 0.25, 0.50],
                            [0.50, 0.25, 0.25, 0.50, 0.25, 0.50], [0.20] * 6])
    current_classes = _class_ids(current_classes)
    min_overlaps = np.stack([overlap_0_7, overlap_0_5], axis=0)[:, :, current_classes]
    compute_aos = _alpha_is_valid(dt_annos)
    (bbox, bev, d3, aos, ctr, ods, bbox40, bev40, d340, aos40, ctr40, ods40, tp_scores) = do_eval_slopedkitti(
        gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, PR_detail_dict=PR_detail_dict, backend=backend)
    result, ret_dict = '\n', {}
    pad = ' ' * 27
    for j, curcls in enumerate(current_classes):
        name = CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            for head, (t_bbox, t_bev, t_3d, t_aos, t_ctr, t_ods) in (('AP', (bbox, bev, d3, aos, ctr, ods)),
                                                                     ('AP_R40', (bbox40, bev40, d340, aos40, ctr40, ods40))):
                result += _line('%s %s@%s:' % (name, head, ', '.join('%.2f' % v for v in min_overlaps[i, :, j])))
                result += _line('level  :  easy     mode    hard      all')
                result += _triple('bbox AP:', t_bbox, j, i) + _triple('bev  AP:', t_bev, j, i) + _triple('3d   AP:', t_3d, j, i)
                if compute_aos:
                    result += _triple('aos  AP:', t_aos, j, i, digits=2)
                for label, value in (('CAP', t_ctr[j, 3, i]), ('ATS', tp_scores[0][j, 3, i]), ('ASS', tp_scores[1][j, 3, i]),
                                     ('AOS', tp_scores[2][j, 3, i]), ('ODS', t_ods[j, 3, i])):
                    result += _line('3d  %s:%s%.4f' % (label, pad, value))
                if head == 'AP':
                    result += _line(' ')
            if i == 0:
                if compute_aos:
                    _r40_entries(ret_dict, name, (('aos', aos40),), j)
                _r40_entries(ret_dict, name, (('3d', d340), ('bev', bev40), ('image', bbox40)), j)
            result += _line(' ')
    return result, ret_dict
