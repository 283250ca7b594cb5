"""Reader of the backend's binary state file (badslam_amd/host/rgbd_io.h: SaveState / LoadState), written from the
documented layout alone: the sections "RGBDVideo (frame poses)" and "Direct BA" of the reference's state file
(applications/badslam/src/badslam/io.cc:112-180) behind the header "BADSLAM" + version byte 101.  Test infrastructure."""
import struct

import numpy as np

SURFEL_DATA_ROWS = 8      # kSurfelDataAttributeCount: x, y, z, packed normal, radius^2, packed colour, descriptor 1, descriptor 2


class _Cursor:
    def __init__(self, blob):
        self.blob, self.pos = blob, 0

    def take(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.blob):
            raise ValueError("unexpected end of file")
        values = struct.unpack_from(fmt, self.blob, self.pos)
        self.pos += size
        return values if len(values) > 1 else values[0]

    def array(self, dtype, count):
        a = np.frombuffer(self.blob, dtype=dtype, count=count, offset=self.pos).copy()
        self.pos += a.nbytes
        return a


def _camera(c):
    type_int, width, height, parameter_count = c.take("<4i")
    return dict(type_int=type_int, width=width, height=height, parameters=c.array("<f4", parameter_count))


def read_state(path):
    blob = open(path, "rb").read()
    c = _Cursor(blob)
    if c.take("<7s") != b"BADSLAM":
        raise ValueError("file identifier does not match")
    s = dict(version=c.take("<B"))
    if s["version"] != 101:
        raise ValueError("unknown file format version")
    frame_count = c.take("<I")
    s["frame_poses"] = c.array("<f4", 7 * frame_count).reshape(frame_count, 7)      # Sophus order: qx qy qz qw tx ty tz
    s["color_camera"] = _camera(c)
    s["pyramid_level_for_color"] = c.take("<i")
    s["depth_camera"] = _camera(c)
    cf_width, cf_height, cf_stride = c.take("<3i")
    rows = c.array(np.uint8, cf_height * cf_stride).reshape(cf_height, cf_stride)
    s["cfactor"] = rows[:, :4 * cf_width].copy().view("<f4")
    s["a"], s["raw_to_float_depth"], s["baseline_fx"] = c.take("<3f")
    s["sparse_surfel_cell_size"] = c.take("<i")
    s["keyframes"] = []
    for _ in range(c.take("<i")):
        kf_id = c.take("<i")
        if kf_id < 0:
            s["keyframes"].append(None)
            continue
        frame_index, activation, last_active, last_covis = c.take("<4i")
        s["keyframes"].append(dict(id=kf_id, frame_index=frame_index, activation=activation,
                                   last_active_in_ba_iteration=last_active, last_covis_in_ba_iteration=last_covis))
    s["surfel_count"], s["surfels_size"] = c.take("<2i")
    s["surfels"] = c.array("<f4", SURFEL_DATA_ROWS * s["surfels_size"]).reshape(SURFEL_DATA_ROWS, s["surfels_size"])
    s["ba_iteration_count"], s["last_ba_iteration_count"] = c.take("<2i")
    s["use_depth_residuals"], s["use_descriptor_residuals"] = (bool(v) for v in c.take("<2B"))
    s["min_observation_count_while_bootstrapping_1"], s["min_observation_count_while_bootstrapping_2"], s["min_observation_count"] = c.take("<3i")
    s["surfel_merge_dist_factor"] = c.take("<f")
    s["bytes_consumed"], s["file_size"] = c.pos, len(blob)
    return s
