"""CPU restatement of the Det6D inference forward pass — TEST INFRASTRUCTURE ONLY.

Follows the reference's Python hot loop in the reference's own tensor layouts
(xyz (B,N,3), features (B,C,N)), using oracle.ops for every compiled op:

  PointNet2FSMSG.forward              core/pcdet/models/backbones_3d/pointnet2_backbone.py:199-263
  _PointnetSAModuleFSBase.forward     core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:358-494
  PointHeadBox6DVote.forward (eval)   core/pcdet/models/dense_heads/point_head_box6d_vote.py:794-903
  Detector3DTemplate.post_processing  core/pcdet/models/detectors/detector3d_template.py:178-284

Inputs are a plain config dict (the MODEL section of the YAML) and a state dict of numpy arrays
under the reference's parameter names.  Conv+BatchNorm(eval) pairs are folded as
W' = W * gamma/sqrt(var+eps), shift = beta - mean * gamma/sqrt(var+eps) in float32, and every
output channel is one ascending-k fmaf chain (oracle/det6d_oracle.c: det6d_oracle_linear).
"""
import numpy as np

from . import ops

F32 = np.float32


def _r4(v):
    return (v + 3) // 4 * 4


def _fold(sd, prefix, conv_i, bn_i):
    """folded (Cin, Cout) weight + (Cout,) shift of `prefix.conv_i` [+ BatchNorm `prefix.bn_i`]"""
    w = sd['%s.%d.weight' % (prefix, conv_i)].astype(F32)
    w = w.reshape(w.shape[0], -1)
    bias = sd.get('%s.%d.bias' % (prefix, conv_i), None)
    if bn_i is not None:
        g = sd['%s.%d.weight' % (prefix, bn_i)].astype(F32)
        b = sd['%s.%d.bias' % (prefix, bn_i)].astype(F32)
        mean = sd['%s.%d.running_mean' % (prefix, bn_i)].astype(F32)
        var = sd['%s.%d.running_var' % (prefix, bn_i)].astype(F32)
        scale = g / np.sqrt(var + F32(1e-5))
        shift = b - mean * scale
        w = w * scale[:, None]
        if bias is not None:
            shift = shift + bias.astype(F32) * scale
    else:
        shift = bias.astype(F32) if bias is not None else np.zeros(w.shape[0], F32)
    return np.ascontiguousarray(w.T), shift.astype(F32)


def _stack(sd, prefix, n_bn_layers, final_bias_conv=False):
    """layers of a Sequential(Conv,BN,ReLU)*n [+ Conv(bias)]: [(W, shift, act)]"""
    layers = []
    for i in range(n_bn_layers):
        w, s = _fold(sd, prefix, 3 * i, 3 * i + 1)
        layers.append((w, s, 1))
    if final_bias_conv:
        w, s = _fold(sd, prefix, 3 * n_bn_layers, None)
        layers.append((w, s, 0))
    return layers


def _pad_rows(w, k_rows, k_offset=0):
    out = np.zeros((k_rows, w.shape[1]), F32)
    out[k_offset:k_offset + w.shape[0]] = w
    return out


def _chain(x, layers):
    """(R, C) -> (R, C') through plain pointwise layers"""
    for w, s, act in layers:
        k = x.shape[1]
        x = ops.linear(np.ascontiguousarray(x), _pad_rows(w, k), s, act)
    return x


def _rows(xyz, feats):
    """[xyz | features | pad] rows (B,N,ld) — cat([grouped_xyz, grouped_features]) order"""
    b, n, _ = xyz.shape
    c = 0 if feats is None else feats.shape[1]
    rows = np.zeros((b, n, _r4(3 + c)), F32)
    rows[:, :, :3] = xyz
    if c:
        rows[:, :, 3:3 + c] = feats.transpose(0, 2, 1)
    return rows


def sa_layer(sd, prefix, spec, xyz, feats, scores=None, new_xyz=None):
    """One PointnetSAModuleFSMSG.  spec: dict(npoint_list, sample_range_list, sample_method_list,
    radii, nsamples, n_mlp_layers, dilated, gamma, agg (int layers), conf (int BN layers or None))."""
    b, n, _ = xyz.shape
    sample_idx = None
    if new_xyz is None:
        idx_list = []
        for (lo, hi), method, npoint in zip(spec['sample_range_list'], spec['sample_method_list'], spec['npoint_list']):
            hi = n if hi == -1 else hi
            sl = np.ascontiguousarray(xyz[:, lo:hi])
            if method == 'd-fps':
                idx = ops.fps(sl, npoint)
            elif method == 's-fps':
                w = ops.sigmoid_pow(np.ascontiguousarray(scores[:, lo:hi]), spec['gamma'])
                idx = ops.fps_weights(sl, w, npoint)
            else:
                raise NotImplementedError(method)
            idx_list.append(idx + lo)
        sample_idx = np.concatenate(idx_list, axis=-1).astype(np.int32)
        new_xyz = ops.gather_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), sample_idx).transpose(0, 2, 1)
        new_xyz = np.ascontiguousarray(new_xyz)
    m = new_xyz.shape[1]
    rows = _rows(xyz, feats)
    pooled, former = [], 0.0
    aux = {'sample_idx': sample_idx, 'idx_cnt': [], 'idx': []}
    for gi, (radius, ns) in enumerate(zip(spec['radii'], spec['nsamples'])):
        if spec['dilated']:
            cnt, idx = ops.ball_query_dilated(former, radius, ns, xyz, new_xyz)
        else:
            cnt, idx = ops.ball_query_cnt(radius, ns, xyz, new_xyz)
        former = radius
        aux['idx_cnt'].append(cnt)
        aux['idx'].append(idx)
        layers = _stack(sd, '%s.mlps.%d' % (prefix, gi), spec['n_mlp_layers'])
        w0, s0, a0 = layers[0]
        x = ops.linear(rows, _pad_rows(w0, rows.shape[-1]), s0, a0, idx=idx, ctr=new_xyz)
        for li, (w, s, a) in enumerate(layers[1:]):
            last = li == len(layers) - 2
            if last:
                x = ops.linear(x, w, s, a, cnt=cnt, pool=ns)  # mask, then max over nsample
            else:
                x = ops.linear(x, w, s, a)
        pooled.append(x)
    feat = np.concatenate(pooled, axis=1)  # (B*M, sum C)
    new_scores = None
    if spec['agg']:
        feat = _chain(feat, _stack(sd, prefix + '.aggregation_mlp', spec['agg']))
    if spec['conf'] is not None:
        new_scores = _chain(feat, _stack(sd, prefix + '.confidence_mlp', spec['conf'], final_bias_conv=True))
        new_scores = new_scores[:, 0].reshape(b, m)
    new_feats = np.ascontiguousarray(feat.reshape(b, m, -1).transpose(0, 2, 1))
    return new_xyz, new_feats, new_scores, aux


def fp_module(sd, prefix, n_layers, unknown, known, unknown_feats, known_feats):
    """PointnetFPModule.forward (core/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:144-174):
    three_nn -> 1 / (dist + 1e-8) normalised over the three neighbours -> three_interpolate -> cat with the skip features
    -> shared Conv2d/BN/ReLU stack (folded, ascending-k fma chains).  unknown (B,n,3), known (B,m,3),
    unknown_feats (B,C1,n) | None, known_feats (B,C2,m) -> (B,C',n)"""
    d2, idx = ops.three_nn(unknown, known)
    dist = np.sqrt(d2).astype(F32)                                   # ThreeNN.forward returns sqrt(dist2) (pointnet2_utils.py:206)
    recip = (F32(1.0) / (dist + F32(1e-8))).astype(F32)
    norm = ((recip[:, :, 0] + recip[:, :, 1]) + recip[:, :, 2]).astype(F32)[:, :, None]
    weight = (recip / norm).astype(F32)
    interpolated = ops.three_interpolate(np.ascontiguousarray(known_feats), idx, np.ascontiguousarray(weight))
    feats = interpolated if unknown_feats is None else np.concatenate([interpolated, unknown_feats], axis=1)
    b, c, n = feats.shape
    x = np.ascontiguousarray(feats.transpose(0, 2, 1)).reshape(b * n, c)
    y = _chain(x, _stack(sd, prefix + '.mlp', n_layers))
    return np.ascontiguousarray(y.reshape(b, n, -1).transpose(0, 2, 1))


def backbone_forward(backbone_cfg, sd, points, batch_size, prefix='backbone_3d'):
    """PointNet2FSMSG.forward incl. the feature-propagation branch (pointnet2_backbone.py:199-263): returns
    dict(l_xyz, l_features, l_scores, point_features (B*N', C), point_xyz (B, N', 3))"""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    pts = np.asarray(points, F32)
    n = pts.shape[0] // batch_size
    xyz = np.ascontiguousarray(pts[:, 1:4].reshape(batch_size, n, 3))
    feats = np.ascontiguousarray(pts[:, 4:].reshape(batch_size, n, -1).transpose(0, 2, 1)) if pts.shape[1] > 4 else None
    l_xyz, l_feats, l_scores = [xyz], [feats], [None]
    pre = prefix + '.' if prefix else ''
    for k, spec in enumerate(backbone_specs({'BACKBONE_3D': backbone_cfg})):
        nx, nf, ns, _ = sa_layer(sd, '%sSA_modules.%d' % (pre, k), spec, l_xyz[-1], l_feats[-1], l_scores[-1])
        l_xyz.append(nx); l_feats.append(nf); l_scores.append(ns)
    fp = backbone_cfg.get('FP_MLPS', None)
    i = 0
    if fp is not None:
        for i in range(-1, -(len(fp) + 1), -1):
            l_feats[i - 1] = fp_module(sd, '%sFP_modules.%d' % (pre, i + len(fp)), len(fp[i + len(fp)]), l_xyz[i - 1], l_xyz[i],
                                       l_feats[i - 1], l_feats[i])
    out_feats = l_feats[i - 1]
    b, c, m = out_feats.shape
    return dict(l_xyz=l_xyz, l_features=l_feats, l_scores=l_scores, point_xyz=l_xyz[i - 1],
                point_features=np.ascontiguousarray(out_feats.transpose(0, 2, 1)).reshape(b * m, c))


def backbone_specs(model_cfg):
    sa = model_cfg['BACKBONE_3D']['SA_CONFIG']
    agg = sa.get('AGGREGATION_MLPS', None)
    conf = sa.get('CONFIDENCE_MLPS', None)
    specs = []
    for k in range(len(sa['NPOINT_LIST'])):
        specs.append(dict(
            npoint_list=sa['NPOINT_LIST'][k], sample_range_list=sa['SAMPLE_RANGE_LIST'][k],
            sample_method_list=sa['SAMPLE_METHOD_LIST'][k], radii=sa['RADIUS'][k], nsamples=sa['NSAMPLE'][k],
            n_mlp_layers=len(sa['MLPS'][k][0]), dilated=sa.get('DILATED_RADIUS_GROUP', False),
            gamma=sa.get('WEIGHT_GAMMA', 1.0),
            agg=len(agg[k]) if agg and agg[k] else 0,
            conf=len(conf[k]) if conf and conf[k] else None))
    return specs


def forward(model_cfg, sd, points, batch_size):
    """points (B*N, 5) [b,x,y,z,i] -> dict with every intermediate the parity tests compare"""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    pts = np.asarray(points, F32)
    n = pts.shape[0] // batch_size
    xyz = np.ascontiguousarray(pts[:, 1:4].reshape(batch_size, n, 3))
    feats = np.ascontiguousarray(pts[:, 4:].reshape(batch_size, n, -1).transpose(0, 2, 1)) if pts.shape[1] > 4 else None
    out = {'l_xyz': [], 'l_scores': [], 'l_features': [], 'sample_idx': [], 'idx_cnt': []}
    scores = None
    for k, spec in enumerate(backbone_specs(model_cfg)):
        xyz, feats, scores, aux = sa_layer(sd, 'backbone_3d.SA_modules.%d' % k, spec, xyz, feats, scores)
        out['l_xyz'].append(xyz)
        out['l_scores'].append(scores)
        out['l_features'].append(feats)
        out['sample_idx'].append(aux['sample_idx'])
        out['idx_cnt'].append(aux['idx_cnt'])
    b, m = batch_size, xyz.shape[1]
    out['point_features'] = np.ascontiguousarray(feats.transpose(0, 2, 1)).reshape(b * m, -1)

    head = model_cfg['POINT_HEAD']
    lo, hi = head['SAMPLE_RANGE']
    cand_xyz = np.ascontiguousarray(xyz[:, lo:hi])
    cand_feats = np.ascontiguousarray(feats[:, :, lo:hi].transpose(0, 2, 1))  # (B,P,C)
    p = cand_xyz.shape[1]
    vote_layers = _stack(sd, 'point_head.vote_layers', len(head['VOTE_CONFIG']['VOTE_FC']), final_bias_conv=True)
    off = _chain(cand_feats.reshape(b * p, -1), vote_layers)
    vote, off_clamped = ops.vote_points(off, cand_xyz.reshape(b * p, 3), head['VOTE_CONFIG']['MAX_TRANSLATION_RANGE'])
    vote_xyz = vote.reshape(b, p, 3)
    spec = dict(radii=head['SA_CONFIG']['RADIUS'], nsamples=head['SA_CONFIG']['NSAMPLE'],
                n_mlp_layers=len(head['SA_CONFIG']['MLPS'][0]), dilated=False, gamma=1.0, agg=0, conf=None)
    _, vfeat, _, aux = sa_layer(sd, 'point_head.SA_module', spec, xyz, feats, new_xyz=vote_xyz)
    x = np.ascontiguousarray(vfeat.transpose(0, 2, 1)).reshape(b * p, -1)
    shared = _chain(x, _stack(sd, 'point_head.shared_fc_layer', len(head['SHARED_FC'])))
    cls = _chain(shared, _stack(sd, 'point_head.cls_layers', len(head['CLS_FC']), final_bias_conv=True))
    reg = _chain(shared, _stack(sd, 'point_head.reg_layers', len(head['REG_FC']), final_bias_conv=True))
    bc = head['TARGET_CONFIG']['BOX_CODER_CONFIG']
    boxes = ops.decode_boxes(reg, vote, nbin=bc.get('angle_bin_num', 12), ground_aware=bc.get('ground_aware', True),
                             minus=bc.get('minus', False), threshold_deg=bc.get('threshold', 10),
                             factor_deg=bc.get('factor', 45))
    out.update(point_candidate_coords=cand_xyz.reshape(b * p, 3), point_vote_coords=vote, vote_offsets=off_clamped,
               head_idx_cnt=aux['idx_cnt'], batch_cls_preds=cls, point_reg_preds=reg, batch_box_preds=boxes)
    pp = model_cfg['POST_PROCESSING']
    nc = pp['NMS_CONFIG']
    ob, osc, ol, oi, oc = ops.postprocess(cls, boxes, b, pp['SCORE_THRESH'], nc['NMS_PRE_MAXSIZE'],
                                          nc['NMS_POST_MAXSIZE'], nc['NMS_THRESH'])
    out['pred_dicts'] = [dict(pred_boxes=ob[i, :oc[i]], pred_scores=osc[i, :oc[i]], pred_labels=ol[i, :oc[i]].astype(np.int64),
                              pred_index=oi[i, :oc[i]]) for i in range(b)]
    return out
