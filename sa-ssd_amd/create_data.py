"""Offline data preparation: the three artefacts tools/create_data.py derives from a raw KITTI tree,
in the same on-disk formats, with the per-point work on the GPU.

  create_kitti_info_file        kitti_infos_{train,val,trainval,test}.pkl  -- per-frame records incl. the number of lidar
                                points inside every labelled box                          (create_data.py:16-104)
  create_reduced_point_cloud    training/velodyne_reduced/%06d.bin -- the sweep cut to the camera-2 frustum, what
                                KittiLiDAR reads                                          (create_data.py:107-165)
  create_groundtruth_database   gt_database/<frame>_<class>_<k>.bin + kitti_dbinfos_train.pkl -- the objects
                                PointAugmentor pastes                                     (create_data.py:168-266)

Per frame the sweep is uploaded once; the frustum test and the point-in-box masks are sassd_points_in_polytopes launches
(the reference: numba CPU loops, ~120 k points x boxes per frame).  Needs an MI355X, like everything per-point here."""
import pathlib
import pickle

import numpy as np
import torch

from . import geometry as G
from . import kitti_common as kitti

_CLASSES = ('Car', 'Pedestrian', 'Cyclist', 'Van', 'Person_sitting', 'Truck', 'Tram', 'Misc')   # kitti_common.py:222-237


def _read_imageset_file(path):
    with open(path, 'r') as f:
        return [int(line) for line in f.readlines()]


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("sassd.create_data needs an MI355X for the per-point work (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _load_points(path, num_features=4, device=None):
    pts = np.fromfile(str(path), dtype=np.float32, count=-1).reshape([-1, num_features])
    return torch.from_numpy(pts).to(_device() if device is None else device)


def _in_image(points_dev, info):
    """the rows of the sweep inside the camera-2 frustum (remove_outside_points, on the GPU)."""
    return G.remove_outside_points(points_dev, info['calib/R0_rect'], info['calib/Tr_velo_to_cam'], info['calib/P2'],
                                   info["img_shape"])


def _boxes_lidar(info, num_obj):
    annos = info['annos']
    cam = np.concatenate([annos['location'][:num_obj], annos['dimensions'][:num_obj],
                          annos['rotation_y'][:num_obj, np.newaxis]], axis=1)
    return G.box_camera_to_lidar(cam, info['calib/R0_rect'], info['calib/Tr_velo_to_cam'])


def _calculate_num_points_in_gt(data_path, infos, relative_path, remove_outside=True, num_features=4, device=None):
    for info in infos:
        v_path = pathlib.Path(data_path) / info["velodyne_path"] if relative_path else info["velodyne_path"]
        pts = _load_points(v_path, num_features, device)
        if remove_outside:
            pts = _in_image(pts, info)
        annos = info['annos']
        num_obj = len([n for n in annos['name'] if n != 'DontCare'])
        inside = G.points_in_rbbox(pts, _boxes_lidar(info, num_obj))
        counts = inside.sum(0).cpu().numpy()
        ignored = len(annos['dimensions']) - num_obj
        annos["num_points_in_gt"] = np.concatenate([counts, -np.ones([ignored])]).astype(np.int32)


def create_kitti_info_file(data_path, save_path=None, relative_path=True, device=None):
    sets = {k: _read_imageset_file(str(pathlib.Path(data_path) / "ImageSets" / (k + ".txt")))
            for k in ("train", "val", "test")}
    save_path = pathlib.Path(data_path if save_path is None else save_path)
    infos = {}
    for k in ("train", "val"):
        infos[k] = kitti.get_kitti_image_info(data_path, training=True, velodyne=True, calib=True, image_ids=sets[k],
                                              relative_path=relative_path)
        _calculate_num_points_in_gt(data_path, infos[k], relative_path, device=device)
    infos["trainval"] = infos["train"] + infos["val"]
    infos["test"] = kitti.get_kitti_image_info(data_path, training=False, label_info=False, velodyne=True, calib=True,
                                               image_ids=sets["test"], relative_path=relative_path)
    for k, v in infos.items():
        with open(save_path / ('kitti_infos_%s.pkl' % k), 'wb') as f:
            pickle.dump(v, f)
    return infos


def _create_reduced_point_cloud(data_path, info_path, save_path=None, back=False, device=None):
    with open(info_path, 'rb') as f:
        kitti_infos = pickle.load(f)
    for info in kitti_infos:
        v_path = pathlib.Path(data_path) / info['velodyne_path']
        pts = _load_points(v_path, 4, device)
        if back:
            pts[:, 0] = -pts[:, 0]
        pts = _in_image(pts, info)
        if save_path is None:
            out_dir = v_path.parent.parent / (v_path.parent.stem + "_reduced")
        else:
            out_dir = pathlib.Path(save_path)
        out_dir.mkdir(parents=True, exist_ok=True)
        pts.cpu().numpy().tofile(str(out_dir / (v_path.name + ("_back" if back else ""))))


def create_reduced_point_cloud(data_path, train_info_path=None, val_info_path=None, test_info_path=None, save_path=None,
                               with_back=False, device=None):
    root = pathlib.Path(data_path)
    paths = [root / 'kitti_infos_train.pkl' if train_info_path is None else train_info_path,
             root / 'kitti_infos_val.pkl' if val_info_path is None else val_info_path,
             root / 'kitti_infos_test.pkl' if test_info_path is None else test_info_path]
    for back in ([False, True] if with_back else [False]):
        for p in paths:
            _create_reduced_point_cloud(data_path, p, save_path, back=back, device=device)


def create_groundtruth_database(data_path, info_path=None, used_classes=None, database_save_path=None,
                                db_info_save_path=None, relative_path=True, lidar_only=False, bev_only=False,
                                coors_range=None, device=None):
    root = pathlib.Path(data_path)
    info_path = root / 'kitti_infos_train.pkl' if info_path is None else info_path
    db_dir = root / 'gt_database' if database_save_path is None else pathlib.Path(database_save_path)
    db_info_save_path = root / "kitti_dbinfos_train.pkl" if db_info_save_path is None else db_info_save_path
    db_dir.mkdir(parents=True, exist_ok=True)
    with open(info_path, 'rb') as f:
        kitti_infos = pickle.load(f)
    used_classes = list(_CLASSES) if used_classes is None else used_classes
    all_db_infos = {name: [] for name in used_classes}
    group_counter = 0
    for info in kitti_infos:
        v_path = str(root / info['velodyne_path']) if relative_path else info['velodyne_path']
        pts = _load_points(v_path, info.get('pointcloud_num_features', 4), device)
        if not lidar_only:
            pts = _in_image(pts, info)
        annos = info["annos"]
        names, gt_idxes, difficulty = annos["name"], annos["index"], annos["difficulty"]
        num_obj = int(np.sum(annos["index"] >= 0))
        rbbox_lidar = _boxes_lidar(info, num_obj)
        if bev_only:
            assert coors_range is not None
            rbbox_lidar[:, 2] = coors_range[2]
            rbbox_lidar[:, 5] = coors_range[5] - coors_range[2]
        group_ids = annos["group_ids"] if "group_ids" in annos else np.arange(annos["bbox"].shape[0], dtype=np.int64)
        inside = G.points_in_rbbox(pts, rbbox_lidar)                        # [points, objects] on the GPU
        group_dict = {}
        for i in range(num_obj):
            filename = f"{info['image_idx']}_{names[i]}_{gt_idxes[i]}.bin"
            obj = pts[inside[:, i]]
            center = torch.from_numpy(rbbox_lidar[i, :3]).to(obj.device)
            obj[:, :3] = (obj[:, :3].double() - center).float()            # float32 - float64, rounded once
            obj.cpu().numpy().tofile(str(db_dir / filename))
            if names[i] in used_classes:
                db_info = {"name": names[i], "path": str(db_dir.stem + "/" + filename) if relative_path
                           else str(db_dir / filename), "image_idx": info["image_idx"], "gt_idx": gt_idxes[i],
                           "box3d_lidar": rbbox_lidar[i], "num_points_in_gt": obj.shape[0], "difficulty": difficulty[i]}
                if group_ids[i] not in group_dict:
                    group_dict[group_ids[i]] = group_counter
                    group_counter += 1
                db_info["group_id"] = group_dict[group_ids[i]]
                if "score" in annos:
                    db_info["score"] = annos["score"][i]
                all_db_infos[names[i]].append(db_info)
    with open(db_info_save_path, 'wb') as f:
        pickle.dump(all_db_infos, f)
    return all_db_infos
