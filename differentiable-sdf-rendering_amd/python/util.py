"""Utilities of the optimisation harness (counterparts of python/util.py): procedural sensors,
sensor resolution, `.vol` grid IO, image IO, metadata, checkpoints."""
import glob
import json
import os
import struct
import sys

import numpy as np
import torch

from dsdf.cameras import Sensor, get_regular_camera_positions, get_regular_cameras, get_regular_cameras_top  # noqa: F401


def default_device():
    """Device of the optimisation state: the HIP device; plain torch bookkeeping (config tables,
    box constraint, .vol IO) also runs on a GPU-less host for the CPU test-suite."""
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def set_sensor_res(sensor, res):
    """python/util.py:146-150."""
    sensor.set_res((int(res[0]), int(res[1])))


def atleast_4d(tensor):
    return tensor[..., None] if tensor.dim() == 3 else tensor


# ---- Mitsuba VolumeGrid file: 'VOL' 3, int32 type=1, xres,yres,zres, channels, 6 x float bbox, data
def write_vol(path, data):
    a = data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else np.asarray(data)
    a = a[..., None] if a.ndim == 3 else a
    z, y, x, c = a.shape
    with open(path, 'wb') as f:
        f.write(b'VOL' + bytes([3]))
        f.write(struct.pack('<iiiii', 1, x, y, z, c))
        f.write(struct.pack('<6f', 0, 0, 0, 1, 1, 1))
        f.write(np.ascontiguousarray(a, '<f4').tobytes())


def read_vol(path, device=None):
    device = device or default_device()
    with open(path, 'rb') as f:
        head = f.read(4)
        if head[:3] != b'VOL' or head[3] != 3:
            raise ValueError(f"{path}: not a version-3 Mitsuba volume file")
        typ, x, y, z, c = struct.unpack('<iiiii', f.read(20))
        if typ != 1:
            raise ValueError(f"{path}: only float32 volumes are supported")
        f.read(24)
        a = np.frombuffer(f.read(4 * x * y * z * c), '<f4').reshape(z, y, x, c)
    t = torch.from_numpy(a.copy()).to(device)
    return t[..., 0] if c == 1 else t


def write_image(path, img):
    """PNG (8-bit sRGB) via PIL when available, else .npy; `.exr` requests are stored as .npy
    (no OpenEXR writer in this environment)."""
    a = img.detach().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    if path.endswith('.png'):
        try:
            from PIL import Image
            srgb = np.where(a <= 0.0031308, 12.92 * a, 1.055 * np.clip(a, 0, None) ** (1 / 2.4) - 0.055)
            Image.fromarray((np.clip(srgb, 0, 1) * 255 + 0.5).astype(np.uint8)).save(path)
            return path
        except ImportError:
            path = path[:-4] + '.npy'
    elif not path.endswith('.npy'):
        path = os.path.splitext(path)[0] + '.npy'
    np.save(path, a.astype(np.float32))
    return path


def read_image(path, device=None):
    device = device or default_device()
    return torch.from_numpy(np.load(path)).to(device)


def resize_img(img, target_res, smooth=False):
    """Box-filter resize of an (H,W,C) tensor to target_res=(H',W') (python/util.py:14-23)."""
    h, w = int(target_res[0]), int(target_res[1])
    if img.shape[0] == h and img.shape[1] == w:
        return img
    t = img.permute(2, 0, 1)[None]
    return torch.nn.functional.interpolate(t, size=(h, w), mode='area')[0].permute(1, 2, 0)


def dump_metadata(config, opt_config, extra=None, fn='test.json'):
    """python/util.py:152-186."""
    def conv(o):
        if hasattr(o, 'name') and isinstance(getattr(o, 'name'), str):
            return o.name
        if callable(o):
            return getattr(o, '__name__', str(o))
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (tuple, list)):
            return [conv(v) for v in o]
        return o if isinstance(o, (int, float, str, bool, type(None))) else str(o)
    d = {'config': {k: conv(v) for k, v in vars(config).items()},
         'opt_config': {k: conv(v) for k, v in vars(opt_config).items() if k not in ('sensors', 'sensors_reordered', 'variables')},
         'cmd': ' '.join(sys.argv)}
    d.update(extra or {})
    with open(fn, 'wt') as f:
        json.dump(d, f, indent=4)


def optimization_result_exists(output_dir, config, opt_config, scene_name):
    opt_name = opt_config if isinstance(opt_config, str) else opt_config.name
    return os.path.isfile(os.path.join(output_dir, scene_name, opt_name, config.name, 'loss.png'))


def get_checkpoint_path_and_suffix(output_dir, scene_name, opt_name, config_name):
    """python/util.py:202-216."""
    p = os.path.realpath(os.path.join(output_dir, scene_name, opt_name, config_name))
    fn = sorted(glob.glob(os.path.join(p, 'params', '*sdf*.vol')))[-1]
    suffix = os.path.splitext(os.path.basename(fn))[0].split('-')[-1]
    return p, int(suffix) if suffix.isdigit() else suffix
