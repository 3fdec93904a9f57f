"""In front of the filter, on the GPU: bounding boxes -> obstacle slots, lane points -> lane slots.

ingest_bbox   what ObstacleList2D.update_by_bounding_box builds per box (reference
              cbf/obstacles.py:833-858 with Ellipse2D.from_bounding_box 320-331 and
              CollisionCone2D.from_bounding_box 534-543), for a whole batch.
lane_fit      PolyLane.fit_polynomial_curve (cbf/obstacles.py:715-773) for every agent's lane points.

All arithmetic is in libsccbf.so; tensors hold device memory and the current stream.
"""
import ctypes as C

import torch

from . import _lib
from .batched import _ptr, obs_layout, obs_shape


def _need_gpu(t, what):
    if not t.is_cuda:
        raise _lib.SccbfError("%s needs tensors on the GPU (no CPU fallback)" % what)
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError("%s works on float32 or float64 tensors" % what)


def _check(name, t, shape, dtype, dev):
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous():
        raise ValueError("%s: expected contiguous %s %s on %s, got %s %s on %s" % (
            name, tuple(shape), dtype, dev, tuple(t.shape), t.dtype, t.device))


def _layout_flag(tiled, agent_major):
    if tiled and agent_major:
        raise ValueError("one layout: tiled or agent_major")
    return _lib.FLAG_OBS_TILED if tiled else (_lib.FLAG_OBS_AGENT_MAJOR if agent_major else 0)


def ingest_bbox(bbox, kind, buffer=0.5, tiled=False, out=None, agent_major=False):
    """bbox (M,2,A,4) = location x, y, extent x, y | yaw, speed, -, -; kind uint8 (M,) or (M,A) with
    OBS_ELLIPSE / OBS_CONE (anything else: an all-zero slot).  Returns obs in the planes layout
    (M,2,A,4) or, with tiled=True, in 64-agent tiles, or, with agent_major=True, as per-agent lists
    (A,M,2,4); `kind` is the obs_type of the solve."""
    _need_gpu(bbox, "ingest_bbox")
    M, A = bbox.shape[0], bbox.shape[2]
    dev, ty = bbox.device, bbox.dtype
    _check("bbox", bbox, (M, 2, A, 4), ty, dev)
    flags = lflag = _layout_flag(tiled, agent_major)
    if kind.dim() == 2:
        flags |= _lib.FLAG_TYPE_PER_AGENT
        _check("kind", kind, (M, A), torch.uint8, dev)
    else:
        _check("kind", kind, (M,), torch.uint8, dev)
    if out is None:
        out = torch.zeros(obs_shape(lflag, A, M), dtype=ty, device=dev)
    _check("out", out, obs_shape(lflag, A, M), ty, dev)
    lib = _lib.load()
    fn = lib.sccbf_ingest_bbox_f32 if ty == torch.float32 else lib.sccbf_ingest_bbox_f64
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = fn(A, M, _ptr(bbox), _ptr(kind), float(buffer), flags, _ptr(out), C.c_void_p(stream))
    _lib.check(rc)
    return out


def bbox_as_obs(bbox, kind, buffer=0.5):
    """No ingestion pass at all: the box records themselves become the filter's obstacle slots.  Writes
    `buffer` into component [m, 1, :, 2] of bbox (M,2,A,4) = location x, y, extent x, y | yaw, speed, -, -
    IN PLACE and returns (bbox, obs_type) with the ellipse kinds of `kind` (uint8 (M,) or (M,A)) turned into
    OBS_ELLIPSE_BOX and the cone kinds into OBS_CONE_BOX: the filter then makes
    Ellipse2D.from_bounding_box / CollisionCone2D.from_bounding_box (reference cbf/obstacles.py:320-331,
    534-543) of every box on every tick, as ObstacleList2D.update_by_bounding_box does (833-858), with the
    answers of ingest_bbox(...) followed by the solve.  Other kinds keep their slot as it is."""
    M, A = bbox.shape[0], bbox.shape[2]
    _check("bbox", bbox, (M, 2, A, 4), bbox.dtype, bbox.device)
    ty = kind.clone()
    ell = (kind == _lib.OBS_ELLIPSE) | (kind == _lib.OBS_ELLIPSE_PREP)
    cone = (kind == _lib.OBS_CONE) | (kind == _lib.OBS_CONE_PREP)
    ty[ell] = _lib.OBS_ELLIPSE_BOX
    ty[cone] = _lib.OBS_CONE_BOX
    boxed = (ell | cone)
    boxed = boxed if boxed.dim() == 2 else boxed[:, None].expand(M, A)
    bbox[:, 1, :, 2] = torch.where(boxed, torch.as_tensor(buffer, dtype=bbox.dtype, device=bbox.device), bbox[:, 1, :, 2])
    return bbox, ty


def lane_fit(px, py, obs, slot, sigma=None, degree=3, buffer=1.5):
    """px, py (K,A): lane points of every agent; sigma (K,A) or None (10 everywhere; a fixed point
    is a point with a small sigma, a point with sigma <= 0 is skipped).  Writes lane slot `slot`
    of obs (planes (M,2,A,4), tiles (ceil(A/64),M,2,64,4) or per-agent lists (A,M,2,4)) in place and returns obs."""
    _need_gpu(px, "lane_fit")
    K, A = px.shape
    dev, ty = px.device, px.dtype
    _check("px", px, (K, A), ty, dev)
    _check("py", py, (K, A), ty, dev)
    if sigma is not None:
        _check("sigma", sigma, (K, A), ty, dev)
    lflag, M, exp = obs_layout(obs, A)
    if lflag == _lib.FLAG_OBS_SHARED:
        raise ValueError("lane_fit writes per-agent slots: planes, tiles or per-agent lists")
    _check("obs", obs, exp, ty, dev)
    lib = _lib.load()
    fn = lib.sccbf_lane_fit_f32 if ty == torch.float32 else lib.sccbf_lane_fit_f64
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = fn(A, K, int(degree), _ptr(px), _ptr(py), _ptr(sigma), float(buffer), int(slot), M,
                lflag, _ptr(obs), C.c_void_p(stream))
    _lib.check(rc)
    return obs
