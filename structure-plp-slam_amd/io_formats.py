"""Host-side data formats either side of the device path: the dataset sequence readers of
the reference's example drivers and the map-database wire format of key points, key lines and descriptors.  Plain Python,
no device work -- the counterpart of host C++ in the reference (example/util/*.cc, src/PLPSLAM/data/common.cc:56-203)."""
import os
from collections import namedtuple

import numpy as np

from . import KL_DTYPE, KP_DTYPE

rgbd_frame = namedtuple("rgbd_frame", "rgb_img_path depth_img_path timestamp")
stereo_frame = namedtuple("stereo_frame", "left_img_path right_img_path timestamp")
mono_frame = namedtuple("mono_frame", "img_path timestamp")


def _open_lines(path):
    try:
        with open(path) as f:
            return f.read().split("\n")
    except OSError:
        raise RuntimeError("Could not load a timestamp file from " + path)


class tum_rgbd_sequence:
    """example/util/tum_rgbd_util.cc:33-129: rgb.txt / depth.txt (three header rows, then 'timestamp file'), every RGB
    frame paired with the depth frame nearest in time (the first one on ties), dropped when the gap exceeds
    min_timediff_thr; the frame's timestamp is the mean of the two."""

    def __init__(self, seq_dir_path, min_timediff_thr=0.1):
        rgb = self._acquire(seq_dir_path, seq_dir_path + "/rgb.txt")
        depth = self._acquire(seq_dir_path, seq_dir_path + "/depth.txt")
        self.timestamps_, self.rgb_img_file_paths_, self.depth_img_file_paths_ = [], [], []
        if not depth:
            if rgb:
                raise RuntimeError("depth.txt lists no frames")      # the reference dereferences begin() of an empty vector here
            return
        dts = np.array([t for t, _ in depth], np.float64)
        for t, path in rgb:
            diff = np.abs(t - dts)
            j = int(np.argmin(diff))                                 # first minimum, like the strict '<' scan
            if min_timediff_thr < diff[j]:
                continue
            self.timestamps_.append((t + dts[j]) / 2.0)
            self.rgb_img_file_paths_.append(path)
            self.depth_img_file_paths_.append(depth[j][1])

    @staticmethod
    def _acquire(seq_dir_path, timestamp_file_path):
        out = []
        for s in _open_lines(timestamp_file_path)[3:]:
            if s:
                tok = s.split()
                out.append((float(tok[0]), seq_dir_path + "/" + (tok[1] if len(tok) > 1 else "")))
        return out

    def get_frames(self):
        return [rgbd_frame(r, d, t) for r, d, t in zip(self.rgb_img_file_paths_, self.depth_img_file_paths_, self.timestamps_)]


class euroc_sequence:
    """example/util/euroc_util.cc:33-78: cam0/data.csv (one header row, 'timestamp_ns,file'), images named by the timestamp"""

    def __init__(self, seq_dir_path):
        self.timestamps_, self.left_img_file_paths_, self.right_img_file_paths_ = [], [], []
        for s in _open_lines(seq_dir_path + "/cam0/data.csv")[1:]:
            s = s.replace(",", " ")
            if s:
                ts = int(s.split()[0])
                self.timestamps_.append(ts / 1E9)
                self.left_img_file_paths_.append(f"{seq_dir_path}/cam0/data/{ts}.png")
                self.right_img_file_paths_.append(f"{seq_dir_path}/cam1/data/{ts}.png")

    def get_frames(self):
        return [stereo_frame(l, r, t) for l, r, t in zip(self.left_img_file_paths_, self.right_img_file_paths_, self.timestamps_)]


class kitti_sequence:
    """example/util/kitti_util.cc:32-74: times.txt, image_0 / image_1 with six-digit frame numbers"""

    def __init__(self, seq_dir_path):
        self.timestamps_ = [float(s.split()[0]) for s in _open_lines(seq_dir_path + "/times.txt") if s]
        self.left_img_file_paths_ = [f"{seq_dir_path}/image_0/{i:06d}.png" for i in range(len(self.timestamps_))]
        self.right_img_file_paths_ = [f"{seq_dir_path}/image_1/{i:06d}.png" for i in range(len(self.timestamps_))]

    def get_frames(self):
        return [stereo_frame(l, r, t) for l, r, t in zip(self.left_img_file_paths_, self.right_img_file_paths_, self.timestamps_)]


class image_sequence:
    """example/util/image_util.cc:30-62: every directory entry, sorted by path, stamped i / fps"""

    def __init__(self, img_dir_path, fps):
        if not os.path.isdir(img_dir_path):
            raise RuntimeError("directory " + img_dir_path + " does not exist")
        self.fps_ = fps
        self.img_file_paths_ = sorted(img_dir_path + "/" + name for name in os.listdir(img_dir_path))

    def get_frames(self):
        return [mono_frame(p, (1.0 / self.fps_) * i) for i, p in enumerate(self.img_file_paths_)]


def read_image(path, grayscale=False):
    """cv::imread(path, IMREAD_UNCHANGED) for the PNG files of the sequences above: uint8 [rows, cols(, 3 in BGR order)],
    or uint16 for the depth maps.  Needs Pillow on the host; decoding is not a device step."""
    from PIL import Image
    with Image.open(path) as im:
        if grayscale:
            im = im.convert("L")
        a = np.array(im)
    if a.ndim == 3:
        a = a[:, :, 2::-1] if a.shape[2] >= 3 else a[:, :, 0]       # RGB(A) -> BGR, what cv::imread hands out
    return np.ascontiguousarray(a)


# ---- map database wire format (src/PLPSLAM/data/common.cc) -------------------------------------------------------------
def convert_keypoints_to_json(keypts):
    """:56-66: {"pt": [x, y], "ang": angle, "oct": unsigned octave}"""
    k = np.asarray(keypts, KP_DTYPE)
    return [{"pt": [float(a["x"]), float(a["y"])], "ang": float(a["angle"]), "oct": int(np.uint32(a["octave"]))} for a in k]


def convert_json_to_keypoints(json_keypts):
    """:82-97: cv::KeyPoint(x, y, size 0, angle, response 0, octave, class_id -1)"""
    k = np.zeros(len(json_keypts), KP_DTYPE)
    for i, j in enumerate(json_keypts):
        k[i] = (np.float32(j["pt"][0]), np.float32(j["pt"][1]), 0.0, np.float32(j["ang"]), 0.0, np.int32(np.uint32(j["oct"])), -1)
    return k


def convert_keylines_to_json(keylines):
    """:69-80"""
    k = np.asarray(keylines, KL_DTYPE)
    return [{"pt_s": [float(a["startPointX"]), float(a["startPointY"])], "pt_e": [float(a["endPointX"]), float(a["endPointY"])],
             "ang": float(a["angle"]), "oct": int(np.uint32(a["octave"]))} for a in k]


def convert_json_to_keylines(json_keylines):
    """:100-113 with the six-argument KeyLine constructor (descriptor_custom.hpp:121-136): angle, class_id -1, octave, the
    mid point in double precision, the two end points; the other fields stay as they were (zero here)."""
    k = np.zeros(len(json_keylines), KL_DTYPE)
    for i, j in enumerate(json_keylines):
        xs, ys, xe, ye = (np.float32(v) for v in (*j["pt_s"], *j["pt_e"]))
        k[i]["angle"] = np.float32(j["ang"]); k[i]["class_id"] = -1; k[i]["octave"] = np.int32(np.uint32(j["oct"]))
        k[i]["pt_x"] = np.float32(0.5 * (float(xs) + float(xe))); k[i]["pt_y"] = np.float32(0.5 * (float(ys) + float(ye)))
        k[i]["startPointX"], k[i]["startPointY"], k[i]["endPointX"], k[i]["endPointY"] = xs, ys, xe, ye
    return k


def convert_undistorted_to_json(undist_keypts):
    """:115-123"""
    k = np.asarray(undist_keypts, KP_DTYPE)
    return [[float(a["x"]), float(a["y"])] for a in k]


def convert_json_to_undistorted(json_undist_keypts, keypts=None):
    """:125-137: positions overwrite a copy of keypts (or default key points when none are given)"""
    if keypts is None or len(keypts) == 0:
        k = np.zeros(len(json_undist_keypts), KP_DTYPE)
        k["size"] = 0.0; k["angle"] = -1.0; k["class_id"] = -1      # cv::KeyPoint()
    else:
        k = np.array(keypts, KP_DTYPE)
        if len(k) != len(json_undist_keypts):
            raise ValueError("key point count differs from the json array")
    for i, j in enumerate(json_undist_keypts):
        k[i]["x"] = np.float32(j[0]); k[i]["y"] = np.float32(j[1])
    return k


def convert_descriptors_to_json(descriptors):
    """:139-155 (ORB) and :158-174 (LBD): each 32-byte row as eight uint32 read from memory (little endian)"""
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
    return d.view("<u4").tolist()


def convert_json_to_descriptors(json_descriptors):
    """:176-189 / :192-205"""
    if len(json_descriptors) == 0:
        return np.zeros((0, 32), np.uint8)
    return np.ascontiguousarray(np.array(json_descriptors, dtype="<u4").reshape(-1, 8)).view(np.uint8).reshape(-1, 32)


convert_lbd_descriptors_to_json = convert_descriptors_to_json
convert_json_to_lbd_descriptors = convert_json_to_descriptors
