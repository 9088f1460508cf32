"""Replacement for the reference extension module ``pointnet2_batch_cuda``
(core/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:11-30).

Same names, same positional arguments, caller-allocated outputs, `int` return (1) like the
reference wrappers (sampling.cpp:40-50, ball_query.cpp:31-75, group_points.cpp, interpolate.cpp).
Differences by design: kernels run on torch's CURRENT stream (the reference uses the legacy
default stream), bad inputs raise instead of exit(-1).
"""
from .. import _lib as L


def _s():
    return L.stream_ptr()


def farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx):
    L.require_cuda(xyz, temp, idx)
    L.call("det6d_fps", b, n, m, L.ptr(xyz), L.ptr(temp), L.ptr(idx), _s())
    return 1


def furthest_point_sampling_weights_wrapper(b, n, m, xyz, weights, temp, idx):
    L.require_cuda(xyz, weights, temp, idx)
    L.call("det6d_fps_weights", b, n, m, L.ptr(xyz), L.ptr(weights), L.ptr(temp), L.ptr(idx), _s())
    return 1


def furthest_point_sampling_matrix_wrapper(b, n, m, matrix, temp, idx):
    raise NotImplementedError(
        "f-fps (furthest_point_sampling_matrix, sampling_gpu.cu:268-373) is outside the Det6D "
        "hot path (SURVEY.md 2.2) and is not provided by libdet6d_hip")


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    L.require_cuda(points, idx, out)
    L.call("det6d_gather_points", b, c, n, npoints, L.ptr(points), L.ptr(idx), L.ptr(out), _s())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    L.require_cuda(grad_out, idx, grad_points)
    L.call("det6d_gather_points_grad", b, c, n, npoints, L.ptr(grad_out), L.ptr(idx), L.ptr(grad_points), _s())
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    L.require_cuda(new_xyz, xyz, idx)
    L.call("det6d_ball_query", b, n, m, radius, nsample, L.ptr(new_xyz), L.ptr(xyz), L.ptr(idx), _s())
    return 1


#: from this many points on, the cnt / dilated queries go through the grid-hashed kernel — when it takes the shape
#: (det6d_ball_query_grid_supported: nsample <= 64, n <= 98304); every other shape runs the brute-force kernels, which
#: have no limit, like the reference's (ball_query_gpu.cu:53-130)
GRID_QUERY_MIN_N = 2048


def _grid_takes(n, nsample):
    return n >= GRID_QUERY_MIN_N and bool(L.lib().det6d_ball_query_grid_supported(n, nsample, 1))


def _grid_shell(b, n, m, radius_in, radius_out, nsample, new_xyz, xyz, idx_cnt, idx):
    """single shell through det6d_ball_query_pair_grid (second shell empty, outputs discarded)"""
    import torch
    dev = xyz.device
    ws = torch.empty((int(L.lib().det6d_ball_query_grid_workspace_bytes(b, n)),), dtype=torch.uint8, device=dev)
    junk_cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
    junk_idx = torch.empty((b, m, 1), dtype=torch.int32, device=dev)
    L.call("det6d_ball_query_pair_grid", b, n, m, radius_in, radius_out, nsample, 0.0, 0.0, 1, L.ptr(new_xyz),
           L.ptr(xyz), L.ptr(ws), L.ptr(idx_cnt), L.ptr(idx), L.ptr(junk_cnt), L.ptr(junk_idx), _s())
    return 1


def ball_query_cnt_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx_cnt, idx):
    L.require_cuda(new_xyz, xyz, idx_cnt, idx)
    if _grid_takes(n, nsample):
        return _grid_shell(b, n, m, 0.0, radius, nsample, new_xyz, xyz, idx_cnt, idx)
    L.call("det6d_ball_query_cnt", b, n, m, radius, nsample, L.ptr(new_xyz), L.ptr(xyz), L.ptr(idx_cnt),
           L.ptr(idx), _s())
    return 1


def ball_query_dilated_wrapper(b, n, m, radius_in, radius_out, nsample, new_xyz, xyz, idx_cnt, idx):
    L.require_cuda(new_xyz, xyz, idx_cnt, idx)
    if _grid_takes(n, nsample):
        return _grid_shell(b, n, m, radius_in, radius_out, nsample, new_xyz, xyz, idx_cnt, idx)
    L.call("det6d_ball_query_dilated", b, n, m, radius_in, radius_out, nsample, L.ptr(new_xyz), L.ptr(xyz),
           L.ptr(idx_cnt), L.ptr(idx), _s())
    return 1


def grid_query_wrapper(out_nebidx, out_nebidxmsk, out_cent, out_centmsk, out_actual_centnum, in_data, in_actual_numpoints,
                       param_coord_shift, param_grid_size, param_voxel_size, param_kernel_size):
    raise NotImplementedError(
        "grid_query_wrapper (gridify.h:13-25) is exported by the reference's module but called by nothing on the Det6D path "
        "(SURVEY.md 8b: dead); libdet6d_hip does not provide it")


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    L.require_cuda(points, idx, out)
    L.call("det6d_group_points", b, c, n, npoints, nsample, L.ptr(points), L.ptr(idx), L.ptr(out), _s())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    L.require_cuda(grad_out, idx, grad_points)
    L.call("det6d_group_points_grad", b, c, n, npoints, nsample, L.ptr(grad_out), L.ptr(idx),
           L.ptr(grad_points), _s())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    L.require_cuda(unknown, known, dist2, idx)
    L.call("det6d_three_nn", b, n, m, L.ptr(unknown), L.ptr(known), L.ptr(dist2), L.ptr(idx), _s())


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    L.require_cuda(points, idx, weight, out)
    L.call("det6d_three_interpolate", b, c, m, n, L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(out), _s())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    L.require_cuda(grad_out, idx, weight, grad_points)
    L.call("det6d_three_interpolate_grad", b, c, n, m, L.ptr(grad_out), L.ptr(idx), L.ptr(weight),
           L.ptr(grad_points), _s())
