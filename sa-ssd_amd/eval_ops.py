"""Mirror of the device part of the reference's KITTI evaluation: `rotate_iou_gpu_eval`
(mmdet/core/post_processing/rotate_nms_gpu.py:594-627, used by mmdet/core/evaluation/kitti_eval.py for the BEV and 3-D
overlap matrices) on the HIP kernel sassd_rotate_iou_eval.  Same signature: numpy in, numpy out."""
import numpy as np
import torch

from . import kernels as K


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5]: (cx, cy, x_dim, y_dim, angle) -> [N,K] in boxes.dtype."""
    dtype = boxes.dtype
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=np.float32).astype(dtype)
    if not torch.cuda.is_available():
        raise RuntimeError("sassd.eval_ops needs an MI355X (no CPU fallback; the CPU oracle lives in oracle/)")
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, np.float32)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, np.float32)).to(dev)
    with torch.cuda.device(dev):
        out = K.rotate_iou_eval(b, q, criterion)
    return out.cpu().numpy().astype(dtype)
