"""Procedural sensors of the reference (python/util.py:84-143) as plain structs.

`Sensor` mirrors what the hot path needs from a Mitsuba `perspective` sensor with
an `hdrfilm` (rgb, gaussian filter, sample_border): look_at frame, x-fov, film
size.  `set_sensor_res` (util.py:146-150) becomes `Sensor.set_res`.
"""
import math

import numpy as np

from ._lib import DsdfCamera


class Sensor:
    def __init__(self, origin, target=(0.5, 0.5, 0.5), up=(0.0, 1.0, 0.0), fov=39.0, resx=128, resy=128):
        self.origin = np.asarray(origin, np.float64)
        self.target = np.asarray(target, np.float64)
        self.up = np.asarray(up, np.float64)
        self.fov = float(fov)
        self.resx, self.resy = int(resx), int(resy)

    def set_res(self, res):
        self.resx, self.resy = int(res[0]), int(res[1])

    def film_size(self):
        return self.resx, self.resy

    def frame(self):
        """Mitsuba `Transform4f.look_at`: dir, left = up x dir, up' = dir x left."""
        d = self.target - self.origin
        d = d / np.linalg.norm(d)
        left = np.cross(self.up, d)
        left = left / np.linalg.norm(left)
        return left, np.cross(d, left), d

    def to_struct(self):
        left, up, d = self.frame()
        c = DsdfCamera()
        for i in range(3):
            c.origin[i] = self.origin[i]
            c.left[i] = left[i]
            c.up[i] = up[i]
            c.dir[i] = d[i]
        c.tan_half_fov = math.tan(math.radians(self.fov) * 0.5)
        return c


def get_regular_camera_positions(n_sensors, angle_shift=0.0, radius=2.0, height_scale=1.0):
    """python/util.py:84-112 for the branch `get_regular_cameras` reaches
    (height_steps = int(n > 1) <= 1): a ring with sinusoidally varying elevation."""
    k = np.arange(n_sensors, dtype=np.float64)
    ang = (k / n_sensors + angle_shift / n_sensors) * (2.0 * np.pi)
    elev = np.clip(1.15 / height_scale + 0.5 * np.sin(ang * n_sensors / 4.0), 0.0, np.pi / 2 + 0.05)
    ring = np.stack([np.cos(ang) * np.sin(elev), np.cos(elev), np.sin(ang) * np.sin(elev)], -1) * radius
    return ring + np.array([0.5, 0.0, 0.5])


def get_regular_cameras(n_sensors, angle_shift=0.0, resx=128, resy=128, radius=2.0, height_scale=1.0):
    """python/util.py:115-138."""
    return [Sensor(o, resx=resx, resy=resy)
            for o in get_regular_camera_positions(n_sensors, angle_shift, radius, height_scale)]


def get_regular_cameras_top(n_sensors, angle_shift=0.0, resx=128, resy=128, radius=2.0):
    """python/util.py:141-143."""
    return get_regular_cameras(n_sensors, angle_shift, resx, resy, radius, height_scale=1.3)
