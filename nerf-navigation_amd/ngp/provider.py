"""Dataset side of the hot path: the `transforms*.json` loader and camera conventions of
nerf/provider.py, feeding `ngp.nav.get_rays` and the renderer.

  nerf_matrix_to_ngp   provider.py:19-27   Blender/NeRF camera-to-world -> the NGP frame (axes cycled, y/z flipped, scaled)
  rand_poses           provider.py:57-91   look-at poses on an orbit sphere
  NeRFDataset          provider.py:94-337  'colmap' (transforms.json) / 'blender' (transforms_{train,val,test}.json) layouts,
                                           image decode + optional downscale, intrinsics, slerped test path, collate -> rays

Differences from the reference, none of which changes a ray:
  * options are keyword arguments instead of an argparse namespace (`opt.path` -> `path`, ...); `from_opt` accepts one;
  * images are decoded with PIL (cv2 is not in this image): same 8-bit RGB / RGBA values divided by 255; a downscale by
    an integer factor is computed as the block mean cv2.INTER_AREA takes; other ratios use PIL's box filter (not
    bit-identical to cv2: up to one 8-bit step);
  * random choices take an explicit `generator` / numpy `rng` so that tests are repeatable.
The data (images, poses) stays host-side unless `preload=True`, exactly as in the reference; rays are generated on `device`.
"""
import glob
import json
import math
import os

import numpy as np
import torch

from .nav import get_rays


def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0, 0, 0)):
    """provider.py:19-27.  Row r of the result is row (r + 1) % 3 of `pose` with columns 1 and 2 negated; the translation
    is scaled and offset.  float32 [4, 4]."""
    p = np.asarray(pose)
    out = np.zeros((4, 4), dtype=np.float32)
    for r, src in enumerate((1, 2, 0)):
        out[r, 0] = p[src, 0]
        out[r, 1] = -p[src, 1]
        out[r, 2] = -p[src, 2]
        out[r, 3] = p[src, 3] * scale + offset[r]
    out[3, 3] = 1
    return out


def rand_poses(size, device, radius=1, theta_range=(math.pi / 3, 2 * math.pi / 3), phi_range=(0, 2 * math.pi), generator=None):
    """provider.py:57-91: `size` cameras on a sphere of `radius`, looking at the origin, up = -y.  [size, 4, 4] float32."""
    def unit(v):
        return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)

    thetas = torch.rand(size, device=device, generator=generator) * (theta_range[1] - theta_range[0]) + theta_range[0]
    phis = torch.rand(size, device=device, generator=generator) * (phi_range[1] - phi_range[0]) + phi_range[0]
    centers = torch.stack([radius * torch.sin(thetas) * torch.sin(phis),
                           radius * torch.cos(thetas),
                           radius * torch.sin(thetas) * torch.cos(phis)], dim=-1)
    forward = -unit(centers)
    up = torch.tensor([0.0, -1.0, 0.0], device=device).expand(size, 3)
    right = unit(torch.cross(forward, up, dim=-1))
    up = unit(torch.cross(right, forward, dim=-1))
    poses = torch.eye(4, dtype=torch.float32, device=device).repeat(size, 1, 1)
    poses[:, :3, :3] = torch.stack((right, up, forward), dim=-1)
    poses[:, :3, 3] = centers
    return poses


def _read_image(path, size_hw):
    """8-bit RGB or RGBA image as float32 [H, W, C] in [0, 1] (provider.py:214-229), resized to `size_hw` if given."""
    try:
        from PIL import Image
    except ImportError as e:                                # no silent fallback: the dataset cannot be read without a decoder
        raise RuntimeError("ngp.provider needs PIL to decode images") from e
    with Image.open(path) as im:
        im = im.convert("RGBA" if im.mode in ("RGBA", "LA", "PA") or "transparency" in im.info else "RGB")
        if size_hw is not None and (im.height, im.width) != tuple(size_hw):
            fh, fw = im.height // size_hw[0], im.width // size_hw[1]
            if fh * size_hw[0] == im.height and fw * size_hw[1] == im.width:
                # integer factors: cv2.INTER_AREA is the plain block mean, rounded half-to-even back to 8 bits
                a = np.asarray(im, dtype=np.float64).reshape(size_hw[0], fh, size_hw[1], fw, -1).mean(axis=(1, 3))
                return (np.rint(a).astype(np.float32)) / 255
            im = im.resize((size_hw[1], size_hw[0]), resample=Image.BOX)
        a = np.asarray(im, dtype=np.uint8)
    return a.astype(np.float32) / 255


def _slerp_matrices(r0, r1, ratio):
    from scipy.spatial.transform import Rotation, Slerp
    return Slerp([0, 1], Rotation.from_matrix(np.stack([r0, r1])))(ratio).as_matrix()


class NeRFDataset:
    """provider.py:94-337.  `type` in train / val / test / trainval / all."""

    def __init__(self, path, device, type="train", downscale=1, n_test=10, *, scale=0.33, offset=(0, 0, 0), bound=2,
                 preload=False, fp16=False, num_rays=4096, rand_pose=-1, error_map=False, color_space="srgb", rng=None):
        self.device, self.type, self.downscale, self.root_path = device, type, downscale, path
        self.preload, self.scale, self.offset, self.bound, self.fp16 = preload, scale, tuple(offset), bound, fp16
        self.training = type in ("train", "all", "trainval")
        self.num_rays = num_rays if self.training else -1
        self.rand_pose = rand_pose
        rng = rng if rng is not None else np.random.default_rng()

        if os.path.exists(os.path.join(path, "transforms.json")):
            self.mode = "colmap"                            # one file, split by hand; the test set is an interpolated path
        elif os.path.exists(os.path.join(path, "transforms_train.json")):
            self.mode = "blender"                           # the split comes with the data
        else:
            raise NotImplementedError(f"[NeRFDataset] Cannot find transforms*.json under {path}")
        transform = self._load_transforms(type)

        if "h" in transform and "w" in transform:
            self.H, self.W = int(transform["h"]) // downscale, int(transform["w"]) // downscale
        else:
            self.H = self.W = None                          # taken from the first image

        frames = transform["frames"]
        to_ngp = lambda f: nerf_matrix_to_ngp(np.array(f["transform_matrix"], dtype=np.float32), scale=scale, offset=self.offset)
        if self.mode == "colmap" and type == "test":
            i0, i1 = rng.choice(len(frames), 2, replace=False)
            p0, p1 = to_ngp(frames[i0]), to_ngp(frames[i1])
            poses, images = [], None
            for i in range(n_test + 1):
                ratio = math.sin((i / n_test - 0.5) * math.pi) * 0.5 + 0.5
                pose = np.eye(4, dtype=np.float32)
                pose[:3, :3] = _slerp_matrices(p0[:3, :3], p1[:3, :3], ratio)
                pose[:3, 3] = (1 - ratio) * p0[:3, 3] + ratio * p1[:3, 3]
                poses.append(pose)
        else:
            if self.mode == "colmap":
                frames = frames[1:] if type == "train" else frames[:1] if type == "val" else frames
            poses, images = [], []
            for f in frames:
                f_path = os.path.join(path, f["file_path"])
                if self.mode == "blender" and "." not in os.path.basename(f_path):
                    f_path += ".png"
                if not os.path.exists(f_path):              # the fox data lists frames that do not exist
                    continue
                image = _read_image(f_path, None if self.H is None else (self.H, self.W))
                if self.H is None:
                    if downscale != 1:
                        self.H, self.W = image.shape[0] // downscale, image.shape[1] // downscale
                        image = _read_image(f_path, (self.H, self.W))
                    else:
                        self.H, self.W = image.shape[:2]
                poses.append(to_ngp(f))
                images.append(image)

        self.poses = torch.from_numpy(np.stack(poses, axis=0))
        self.images = torch.from_numpy(np.stack(images, axis=0)) if images is not None else None
        self.radius = self.poses[:, :3, 3].norm(dim=-1).mean(0).item()
        self.error_map = torch.ones([self.images.shape[0], 128 * 128], dtype=torch.float) if self.training and error_map else None

        if preload:
            self.poses = self.poses.to(device)
            if self.images is not None:
                self.images = self.images.to(torch.half if fp16 and color_space != "linear" else torch.float).to(device)
            if self.error_map is not None:
                self.error_map = self.error_map.to(device)

        if "fl_x" in transform or "fl_y" in transform:
            fl_x = (transform["fl_x"] if "fl_x" in transform else transform["fl_y"]) / downscale
            fl_y = (transform["fl_y"] if "fl_y" in transform else transform["fl_x"]) / downscale
        elif "camera_angle_x" in transform or "camera_angle_y" in transform:
            fl_x = self.W / (2 * np.tan(transform["camera_angle_x"] / 2)) if "camera_angle_x" in transform else None
            fl_y = self.H / (2 * np.tan(transform["camera_angle_y"] / 2)) if "camera_angle_y" in transform else None
            fl_x, fl_y = (fl_y if fl_x is None else fl_x), (fl_x if fl_y is None else fl_y)
        else:
            raise RuntimeError("Failed to load focal length, please check the transforms.json!")
        cx = transform["cx"] / downscale if "cx" in transform else self.W / 2
        cy = transform["cy"] / downscale if "cy" in transform else self.H / 2
        self.intrinsics = np.array([fl_x, fl_y, cx, cy])

    @classmethod
    def from_opt(cls, opt, device, type="train", downscale=1, n_test=10):
        """the reference's constructor signature: `opt` is its argparse namespace (main_nerf.py)"""
        return cls(opt.path, device, type, downscale, n_test, scale=opt.scale, offset=opt.offset, bound=opt.bound,
                   preload=opt.preload, fp16=opt.fp16, num_rays=opt.num_rays, rand_pose=opt.rand_pose,
                   error_map=opt.error_map, color_space=opt.color_space)

    def _load_transforms(self, type):
        def read(name):
            with open(os.path.join(self.root_path, name), "r") as f:
                return json.load(f)
        if self.mode == "colmap":
            return read("transforms.json")
        if type == "all":                                   # every split file in the directory
            merged = None
            for p in sorted(glob.glob(os.path.join(self.root_path, "*.json"))):
                t = read(os.path.basename(p))
                if merged is None:
                    merged = t
                else:
                    merged["frames"].extend(t["frames"])
            return merged
        if type == "trainval":
            t = read("transforms_train.json")
            t["frames"].extend(read("transforms_val.json")["frames"])
            return t
        return read(f"transforms_{type}.json")

    def collate(self, index, generator=None):
        """provider.py:275-325: one batch (a list of frame indices, length 1 in the reference) -> rays (+ ground-truth pixels)."""
        B = len(index)
        if self.rand_pose == 0 or index[0] >= len(self.poses):         # a random pose without ground truth
            poses = rand_poses(B, self.device, radius=self.radius, generator=generator)
            s = np.sqrt(self.H * self.W / self.num_rays)
            rH, rW = int(self.H / s), int(self.W / s)
            rays = get_rays(poses, self.intrinsics / s, rH, rW, -1)
            return {"H": rH, "W": rW, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"]}
        poses = self.poses[index].to(self.device)
        error_map = None if self.error_map is None else self.error_map[index]
        rays = get_rays(poses, self.intrinsics, self.H, self.W, self.num_rays, error_map, generator=generator)
        results = {"H": self.H, "W": self.W, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"]}
        if self.images is not None:
            images = self.images[index].to(self.device)
            if self.training:
                C = images.shape[-1]
                images = torch.gather(images.view(B, -1, C), 1, torch.stack(C * [rays["inds"]], -1))
            results["images"] = images
        if error_map is not None:
            results["index"] = index
            results["inds_coarse"] = rays["inds_coarse"]
        return results

    def dataloader(self):
        from torch.utils.data import DataLoader
        size = len(self.poses)
        if self.training and self.rand_pose > 0:
            size += size // self.rand_pose                  # indices past the end ask for a random pose
        loader = DataLoader(list(range(size)), batch_size=1, collate_fn=self.collate, shuffle=self.training, num_workers=0)
        loader._data = self                                 # the trainer reaches error_map and poses through the loader
        loader.has_gt = self.images is not None
        return loader
