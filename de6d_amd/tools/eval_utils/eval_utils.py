"""Evaluation loop of the hot path (mirror of core/tools/eval_utils/eval_utils.py:12-121:
`statistics_info`, `eval_one_epoch` — same arguments, same `result.pkl`, same log lines and return
dictionary).

Differences that follow from the MI355X design: the model is NOT wrapped in DistributedDataParallel
for testing (inference has no collective; every rank runs its shard of scenes), results are merged
with one `all_gather_object` (de6d_amd.parallel.gather_detections) instead of pickle files and two
barriers (common_utils.merge_results_dist), and `generate_prediction_dicts` converts a batch with one
kernel launch and one device-to-host copy.
"""
import pickle
import time

import torch
import torch.distributed as dist

from ...pcdet.models import load_data_to_gpu
from ... import parallel


def statistics_info(cfg, ret_dict, metric, disp_dict):
    thresholds = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST
    for t in thresholds:
        for stage in ('roi', 'rcnn'):
            metric['recall_%s_%s' % (stage, t)] += ret_dict.get('%s_%s' % (stage, t), 0)
    metric['gt_num'] += ret_dict.get('gt', 0)
    t0 = thresholds[0]
    disp_dict['recall_%s' % t0] = '(%d, %d) / %d' % (metric['recall_roi_%s' % t0], metric['recall_rcnn_%s' % t0],
                                                     metric['gt_num'])


def _rank_world(dist_test):
    if dist_test and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def eval_one_epoch(cfg, model, dataloader, epoch_id, logger, dist_test=False, save_to_file=False, result_dir=None):
    result_dir.mkdir(parents=True, exist_ok=True)
    final_output_dir = result_dir / 'final_result' / 'data'
    if save_to_file:
        final_output_dir.mkdir(parents=True, exist_ok=True)

    thresholds = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST
    metric = {'gt_num': 0}
    for t in thresholds:
        metric['recall_roi_%s' % t] = 0
        metric['recall_rcnn_%s' % t] = 0

    dataset = dataloader.dataset
    class_names = dataset.class_names
    det_annos = []
    rank, world = _rank_world(dist_test)

    logger.info('*************** EPOCH %s EVALUATION *****************' % epoch_id)
    model.eval()
    start_time = time.time()
    for batch_dict in dataloader:
        load_data_to_gpu(batch_dict)
        with torch.no_grad():
            pred_dicts, ret_dict = model(batch_dict)
        disp_dict = {}
        statistics_info(cfg, ret_dict, metric, disp_dict)
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names,
                                                       output_path=final_output_dir if save_to_file else None)

    if world > 1:
        det_annos = parallel.gather_detections(det_annos, len(dataset))
        all_metrics = [None] * world
        dist.all_gather_object(all_metrics, metric)
        metric = {k: sum(m[k] for m in all_metrics) for k in metric}

    logger.info('*************** Performance of EPOCH %s *****************' % epoch_id)
    sec_per_example = (time.time() - start_time) / max(len(dataset), 1)
    logger.info('Generate label finished(sec_per_example: %.4f second).' % sec_per_example)
    if rank != 0:
        return {}

    ret = {}
    gt_num = max(metric['gt_num'], 1)
    for t in thresholds:
        for stage in ('roi', 'rcnn'):
            value = metric['recall_%s_%s' % (stage, t)] / gt_num
            logger.info('recall_%s_%s: %f' % (stage, t, value))
            ret['recall/%s_%s' % (stage, t)] = value

    total_pred_objects = sum(len(anno['name']) for anno in det_annos)
    logger.info('Average predicted number of objects(%d samples): %.3f'
                % (len(det_annos), total_pred_objects / max(1, len(det_annos))))
    with open(result_dir / 'result.pkl', 'wb') as f:
        pickle.dump(det_annos, f)

    try:
        result_str, result_dict = dataset.evaluation(det_annos, class_names,
                                                     eval_metric=cfg.MODEL.POST_PROCESSING.EVAL_METRIC,
                                                     output_path=final_output_dir)
    except NotImplementedError as err:   # a dataset without ground truth
        result_str, result_dict = 'evaluation skipped: %s' % err, {}
    if result_str is None:
        result_str = 'evaluation skipped: no ground-truth annotations'
    logger.info(result_str)
    ret.update(result_dict)
    logger.info('Result is save to %s' % result_dir)
    logger.info('****************Evaluation done.*****************')
    return ret
