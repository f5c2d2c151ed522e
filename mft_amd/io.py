"""FlowCache -- the (left_id, right_id) -> (flow, occlusion, sigma) cache plugin of
``MFT/utils/io.py:618-751``, with the interface the tracker uses
(``MFT/MFT.py:214-228``): ``read(left_id, right_id) -> (flow, occl, sigma) | (None, None, None)``
and ``write(left_id, right_id, flow, occl, sigma)``.

Tiers, filled in this order like the reference: device memory (HBM, default budget 5 GB --
raise it freely, an MI355X has 288 GB), host RAM, then one file per pair in ``cache_dir``.
The on-disk format is the reference's ``<left>--<right>.flowouX16.pkl`` (uint16-quantised
PNG-in-pickle, ``mft_amd/flowou_codec.py``): a cache directory written by the
reference can be loaded here and vice versa.  Like in the reference, an entry that went through
the disk tier comes back quantised (16 bits over the channel's range), entries served from HBM or
RAM are exact.
"""
from __future__ import annotations

import shutil
from pathlib import Path

import torch

SUFFIX = ".flowouX16.pkl"


def _nbytes(tensors):
    return sum(t.numel() * t.element_size() for t in tensors)


class FlowCache:
    def __init__(self, cache_dir=None, max_RAM_MB=10000, max_GPU_RAM_MB=5000, device="cuda"):
        self.cache_dir = Path(cache_dir) if cache_dir is not None else None
        self.max_RAM_MB, self.max_GPU_RAM_MB = max_RAM_MB, max_GPU_RAM_MB
        self.device = device
        self.ram_cache, self.gpu_ram_cache = {}, {}
        self.bytes_used = self.gpu_ram_bytes_used = 0
        self.n_saved = 0
        if self.cache_dir is not None:
            self.cache_dir.mkdir(parents=True, exist_ok=True)

    def _path(self, left_id, right_id):
        return self.cache_dir / f"{left_id}--{right_id}{SUFFIX}"

    def ram_space_left(self):
        return max(self.max_RAM_MB * 1000000 - self.bytes_used, 0)

    def gpu_ram_space_left(self):
        return max(self.max_GPU_RAM_MB * 1000000 - self.gpu_ram_bytes_used, 0)

    def read(self, left_id, right_id):
        key = (left_id, right_id)
        if key in self.gpu_ram_cache:
            return self.gpu_ram_cache[key]
        if key in self.ram_cache:
            return tuple(t.to(self.device) for t in self.ram_cache[key])
        if self.cache_dir is not None and self._path(*key).exists():
            try:
                val = self._load(self._path(*key))
                self.write(left_id, right_id, *val)      # promote to the faster tiers
                return val
            except Exception:
                pass
        return None, None, None

    def write(self, left_id, right_id, flow_left_to_right, occlusions, sigmas):
        key = (left_id, right_id)
        val = (flow_left_to_right, occlusions, sigmas)
        if self.gpu_ram_space_left() > 0:
            if key not in self.gpu_ram_cache:
                self.gpu_ram_bytes_used += _nbytes(val)
            self.gpu_ram_cache[key] = val
        elif self.ram_space_left() > 0:
            val = tuple(t.cpu() for t in val)
            if key not in self.ram_cache:
                self.bytes_used += _nbytes(val)
            self.ram_cache[key] = val
        elif self.cache_dir is not None and not self._path(*key).exists():
            self._save(self._path(*key), val)
        self.n_saved += 1

    def _save(self, path, val):
        from .flowou_codec import write_flowou_X16
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        write_flowou_X16(path, *(t.to(self.device) for t in val))

    def _load(self, path):
        from .flowou_codec import read_flowou_X16
        return read_flowou_X16(path, device=self.device)

    def clear(self, clear_disk=True):
        self.gpu_ram_cache.clear()
        self.ram_cache.clear()
        self.bytes_used = self.gpu_ram_bytes_used = 0
        self.n_saved = 0
        if clear_disk and self.cache_dir is not None:
            shutil.rmtree(self.cache_dir, ignore_errors=True)

    def backup_to_disk(self):
        """Save every cached pair to ``cache_dir``."""
        assert self.cache_dir is not None
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        n = 0
        for cache in (self.ram_cache, self.gpu_ram_cache):
            for key, val in cache.items():
                if not self._path(*key).exists():
                    self._save(self._path(*key), val)
                    n += 1
        return n

    def load_from_disk(self):
        assert self.cache_dir is not None
        n = 0
        for path in sorted(self.cache_dir.glob("*" + SUFFIX)):
            left_id, right_id = (int(x) for x in path.name[: -len(SUFFIX)].split("--"))
            try:
                val = self._load(path)
                self.write(left_id, right_id, *val)
                n += 1
            except Exception:
                pass
        return n
