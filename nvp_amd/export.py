"""Grid export for image / video codecs: the on-disk representation the reference's
`experiment_scripts/compression.py` produces and `eval_compression.py` reads back.

* keyframes (compression.py:17-77): per feature dimension d and level l, the level's `res x res` cells quantised to 8 bits
  with that (d, l)'s own min / max -> one grayscale image `keyframes/<plane>/dim<d>/<ll>.png`;
* sparse grid (compression.py:80-106): per feature dimension d, the `[T, X, Y]` volume quantised with that dimension's min / max
  -> T grayscale frames `sparsegrid/dim<d>/<fffff>.png` (the reference then feeds them to ffmpeg/HEVC, the keyframe images
  to JPEG; those external codecs are out of scope - any codec that returns 8-bit images plugs into `*_from_planes`);
* decode (eval_compression.py:22-124): `img / 255 * (max - min) + min`, levels concatenated, dimensions interleaved.

Quantisation is the reference's expression, `uint8(255 * (x - min) / (max - min) + 0.5)` in float32.  The reference recomputes
min / max from the uncompressed checkpoint when decoding; `export_model` stores them in `side_info.json` so the directory is
self-contained.  PNG files are written / read with the standard library (8-bit grayscale, no interlace).  CPU-side tooling:
tensors are moved to the host, nothing here touches the HIP kernels.
"""
from __future__ import annotations

import json
import math
import os
import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

UNIT = 2.0 ** 8 - 1.0          # compression.py:127


def level_geometry(cfg: dict) -> Tuple[List[int], List[int]]:
    """(resolution per level, cell offset per level + total): compression.py:26-33."""
    res, off, total = [], [], 0
    for i in range(int(cfg["n_levels"])):
        b = int(math.ceil(math.exp(i * math.log(cfg["per_level_scale"])) * 16 - 1) + 1)
        res.append(b)
        off.append(total)
        total += b * b
    off.append(total)
    return res, off


def _q8(x: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    q = UNIT * ((x - lo) / (hi - lo))
    return torch.clamp((q + 0.5).to(torch.uint8), 0, 255)


def keyframe_planes(params: torch.Tensor, cfg: dict):
    """-> (images[d][l] uint8 [res, res], mins[d][l], maxs[d][l]) for one 2D keyframe grid (flat `params`)."""
    dim = int(cfg["n_features_per_level"])
    res, off = level_geometry(cfg)
    feats = params.detach().float().reshape(-1, dim)          # quantised where the parameters live; only the uint8 planes travel to the host
    if feats.shape[0] != off[-1]:
        raise ValueError("params length does not match the encoding_config")
    images, mins, maxs = [], [], []
    for d in range(dim):
        images.append([]); mins.append([]); maxs.append([])
        for l, r in enumerate(res):
            x = feats[off[l]:off[l + 1], d]
            lo, hi = torch.min(x), torch.max(x)
            images[d].append(_q8(x, lo, hi).reshape(r, r).cpu().numpy())
            mins[d].append(float(lo)); maxs[d].append(float(hi))
    return images, mins, maxs


def keyframes_from_planes(images, mins, maxs) -> torch.Tensor:
    """Inverse (eval_compression.py:73-93): 8-bit images (possibly after a lossy codec) -> flat fp32 params."""
    per_dim = []
    for d in range(len(images)):
        lv = [torch.from_numpy(np.asarray(img, dtype=np.float32)).reshape(-1) / UNIT * (maxs[d][l] - mins[d][l]) + mins[d][l]
              for l, img in enumerate(images[d])]
        per_dim.append(torch.cat(lv))
    return torch.stack(per_dim, dim=-1).reshape(-1)


def sparse_planes(emb: torch.Tensor):
    """-> (frames[d] uint8 [T, X, Y], mins[d], maxs[d]) for the sparse grid `[T, X, Y, F]`."""
    e = emb.detach().float()
    frames, mins, maxs = [], [], []
    for d in range(e.shape[3]):
        x = e[:, :, :, d]
        lo, hi = torch.min(x), torch.max(x)
        frames.append(_q8(x, lo, hi).cpu().numpy())
        mins.append(float(lo)); maxs.append(float(hi))
    return frames, mins, maxs


def sparse_from_planes(frames, mins, maxs) -> torch.Tensor:
    """Inverse (eval_compression.py:96-124): per-dimension 8-bit frame stacks -> fp32 `[T, X, Y, F]`."""
    vols = [torch.from_numpy(np.asarray(f, dtype=np.float32)) / UNIT * (maxs[d] - mins[d]) + mins[d] for d, f in enumerate(frames)]
    return torch.stack(vols, dim=3)


# ---- minimal 8-bit grayscale PNG (stdlib only) ---------------------------------------------------------------
def write_png(path: str, img: np.ndarray) -> None:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    h, w = img.shape
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))           # filter type 0 on every scanline

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def read_png(path: str) -> np.ndarray:
    """Reads the files write_png produces (8-bit gray, non-interlaced, filter 0-4 handled)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, inter = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, inter) != (8, 0, 0):
                raise ValueError("only 8-bit grayscale non-interlaced PNGs are supported")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    out = np.zeros((h, w), dtype=np.uint8)
    prev = np.zeros(w, dtype=np.int32)
    for r in range(h):
        ft = raw[r * (w + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=w, offset=r * (w + 1) + 1).astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                                   # sub / average / paeth need the running left neighbour
            cur = np.zeros(w, dtype=np.int32)
            for c in range(w):
                a = cur[c - 1] if c else 0
                b, cc = prev[c], (prev[c - 1] if c else 0)
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[c] = (line[c] + pred) & 255
        out[r] = cur
        prev = cur
    return out


PLANES = (("xy", "keyframes_xy", "2d_encoding_xy"), ("xt", "keyframes_xt", "2d_encoding_xt"), ("yt", "keyframes_yt", "2d_encoding_yt"))


def export_model(model, cfg: dict, out_dir: str) -> Dict[str, int]:
    """Write compression.py's `compression/src` tree for an NVP model (+ side_info.json).  Returns file / byte counts."""
    side, files, nbytes = {"keyframes": {}, "sparsegrid": {}}, 0, 0
    for short, attr, key in PLANES:
        images, mins, maxs = keyframe_planes(getattr(model, attr).params, cfg[key])
        side["keyframes"][short] = {"min": mins, "max": maxs}
        for d, lv in enumerate(images):
            ddir = os.path.join(out_dir, "keyframes", short, f"dim{d}")
            os.makedirs(ddir, exist_ok=True)
            for l, img in enumerate(lv):
                path = os.path.join(ddir, f"{l:02d}.png")
                write_png(path, img); files += 1; nbytes += os.path.getsize(path)
    frames, mins, maxs = sparse_planes(model.sparse_grid.embeddings)
    side["sparsegrid"] = {"min": mins, "max": maxs}
    for d, vol in enumerate(frames):
        ddir = os.path.join(out_dir, "sparsegrid", f"dim{d}")
        os.makedirs(ddir, exist_ok=True)
        for t in range(vol.shape[0]):
            path = os.path.join(ddir, f"{t:05d}.png")
            write_png(path, vol[t]); files += 1; nbytes += os.path.getsize(path)
    with open(os.path.join(out_dir, "side_info.json"), "w") as f:
        json.dump(side, f)
    return {"files": files, "bytes": nbytes}


def import_model(model, cfg: dict, out_dir: str) -> None:
    """Read an `export_model` tree back and rebind the de-quantised grids (what eval_compression.py:176-190 does)."""
    side = json.load(open(os.path.join(out_dir, "side_info.json")))
    dev = model.sparse_grid.embeddings.device
    with torch.no_grad():
        for short, attr, key in PLANES:
            dim, nl = int(cfg[key]["n_features_per_level"]), int(cfg[key]["n_levels"])
            images = [[read_png(os.path.join(out_dir, "keyframes", short, f"dim{d}", f"{l:02d}.png")) for l in range(nl)] for d in range(dim)]
            enc = getattr(model, attr)
            enc.params = torch.nn.Parameter(keyframes_from_planes(images, side["keyframes"][short]["min"], side["keyframes"][short]["max"]).to(dev))
        T, F = model.sparse_grid.embeddings.shape[0], model.sparse_grid.embeddings.shape[3]
        frames = [np.stack([read_png(os.path.join(out_dir, "sparsegrid", f"dim{d}", f"{t:05d}.png")) for t in range(T)]) for d in range(F)]
        model.sparse_grid.embeddings = torch.nn.Parameter(sparse_from_planes(frames, side["sparsegrid"]["min"], side["sparsegrid"]["max"]).to(dev))
